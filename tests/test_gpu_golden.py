"""GPU: the HIP path, through the C-ABI (mex_api), reproduces the committed golden vectors bit for bit:
EXACT_ORDER mode against the 'lex' outputs, RED_BLACK mode against the 'colour' outputs."""
import pytest

import golden_util as gu
import problems as pb

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", gu.names())
def test_hip_matches_golden(pdeip, name):
    meta, inputs, outs = gu.load(name)
    fn, args, kw = gu.call(pdeip.mex_api, meta, inputs, single=True)
    for tag, want in outs.items():
        pdeip.mex_api.set_mode({"lex": 0, "colour": 1, "any": 0}[tag])
        got = fn(*args, **kw)
        got = got if isinstance(got, tuple) else (got,)
        assert len(got) == len(want)
        for k, (g, w) in enumerate(zip(got, want)):
            assert pb.bit_equal(g, w), "%s[%s] output %d: %s" % (name, tag, k, pb.describe_mismatch(g, w))
    pdeip.mex_api.set_mode(0)
