"""CPU: the oracle reproduces the committed golden vectors bit for bit (guards the oracle, and the
fixtures the GPU tests compare against, from accidental change)."""
import pytest

import golden_util as gu
import problems as pb


@pytest.mark.parametrize("name", gu.names())
def test_oracle_matches_golden(oracle, name):
    meta, inputs, outs = gu.load(name)
    fn, args, kw = gu.call(oracle, meta, inputs, single=False)
    for tag, want in outs.items():
        if meta["gateway"] in gu.ORDERED:
            got = fn(*args, order={"lex": oracle.LEX, "colour": oracle.COLOUR}[tag], **kw)
        else:
            got = fn(*args)
        got = got if isinstance(got, tuple) else (got,)
        assert len(got) == len(want)
        for k, (g, w) in enumerate(zip(got, want)):
            assert pb.bit_equal(g, w), "%s[%s] output %d: %s" % (name, tag, k, pb.describe_mismatch(g, w))


def test_fixture_inventory():
    have = set(m["gateway"] for m, _, _ in (gu.load(n) for n in gu.names()))
    want = {"Oflow_sor_elin4_2d", "Oflow_sor_llin4_2d", "Oflow_sor_llin8_2d", "Oflow_lhs_elin4_2d", "Oflow_lhs_llin4_2d",
            "Disp_sor_llin4_2d", "Disp_sor_llin_sym4_2d", "PDEsolver4", "PDEsolver8", "DdiffWeights", "BilinInterp_2d", "FstDerivatives5",
            "SndDerivatives5"}
    assert want <= have, "gateways without a golden fixture: %s" % (want - have)
