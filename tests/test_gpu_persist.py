"""GPU: the persistent exact-order kernel (one launch, progress counters, write-through hand-off) against the
oracle, bit for bit, forced on with PDEIP_EXACT_PERSIST=1 -- small/odd/large frames, NaN-laced data, every
5-point model, multi-frame -- and its abort word stays clear.  It is the default form; the launch-per-front form
(PDEIP_EXACT_PERSIST=0) is kept and checked against the same answers."""
import os

import numpy as np
import pytest

import problems as pb

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["persist", "walk64", "walk48", "walk32"])
def persist_on(request):
    """The one-launch form, through round 2's kernel (k_sor_exact_persist, the default) and through round 3's opt-in walker
    (k_sor_walk: PDEIP_EXACT_WALK=1) at each of its strip widths."""
    keys = ("PDEIP_EXACT_PERSIST", "PDEIP_EXACT_WALK", "PDEIP_WALK_W")
    old = {k: os.environ.get(k) for k in keys}
    os.environ["PDEIP_EXACT_PERSIST"] = "1"
    if request.param.startswith("walk"):
        os.environ["PDEIP_EXACT_WALK"] = "1"
        os.environ["PDEIP_WALK_W"] = request.param[4:]
    else:
        os.environ["PDEIP_EXACT_WALK"] = "0"
    yield
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _ok(pdeip):
    assert pdeip.capi.load().pdeip_persist_error() == 0, pdeip.capi.last_error()


@pytest.mark.parametrize("shape", [(32, 48), (97, 131), (64, 200), (131, 70), (3, 3), (5, 300), (260, 7), (388, 584), (1080, 1920)])
def test_persistent_elin4(pdeip, oracle, persist_on, shape):
    pdeip.mex_api.set_mode(0)
    for it in (1, 4, 9):
        p = pb.elin4(811, *shape, nan_frac=0.01)
        got = pdeip.mex_api.Oflow_sor_elin4_2d(*p.values(), np.float32(it), np.float32(1.9), np.float32(1))
        assert pdeip.capi.load().pdeip_last_launch_count() <= 3  # derive + one persistent launch + border fill
        _ok(pdeip)
        want = oracle.Oflow_sor_elin4_2d(*p.values(), it, 1.9)
        for g, w in zip(got, want):
            assert pb.bit_equal(g, w), "elin4 %s it=%d: %s" % (shape, it, pb.describe_mismatch(g, w))


@pytest.mark.parametrize("shape", [(97, 131), (388, 584), (70, 1000)])
def test_persistent_other_models(pdeip, oracle, persist_on, shape):
    api = pdeip.mex_api
    api.set_mode(0)
    q = pb.llin4(821, *shape, nan_frac=0.02)
    for g, w in zip(api.Oflow_sor_llin4_2d(*q.values(), np.float32(5), np.float32(1.9), np.float32(1)), oracle.Oflow_sor_llin4_2d(*q.values(), 5, 1.9)):
        assert pb.bit_equal(g, w)
    d = pb.disp4(822, *shape, nan_frac=0.02)
    assert pb.bit_equal(api.Disp_sor_llin4_2d(*d.values(), np.float32(6), np.float32(1.9), np.float32(1)), oracle.Disp_sor_llin4_2d(*d.values(), 6, 1.9))
    e = pb.pde4(823, *shape, nframes=3, nan_frac=0.02)
    assert pb.bit_equal(api.PDEsolver4(*e.values(), np.float32(5), np.float32(1.75), np.float32(1)), oracle.PDEsolver4(*e.values(), 5, 1.75))
    _ok(pdeip)


def test_auto_policy_uses_persistent_for_long_calls(pdeip, oracle):
    """Without the override every call takes the persistent kernel (few launches)."""
    os.environ.pop("PDEIP_EXACT_PERSIST", None)
    pdeip.mex_api.set_mode(0)
    p = pb.elin4(831, 388, 584)
    lib = pdeip.capi.load()
    got = pdeip.mex_api.Oflow_sor_elin4_2d(*p.values(), np.float32(20), np.float32(1.9), np.float32(1))
    assert lib.pdeip_last_launch_count() <= 3
    _ok(pdeip)
    for g, w in zip(got, oracle.Oflow_sor_elin4_2d(*p.values(), 20, 1.9)):
        assert pb.bit_equal(g, w)
    got = pdeip.mex_api.Oflow_sor_elin4_2d(*p.values(), np.float32(4), np.float32(1.9), np.float32(1))
    assert lib.pdeip_last_launch_count() <= 3
    for g, w in zip(got, oracle.Oflow_sor_elin4_2d(*p.values(), 4, 1.9)):
        assert pb.bit_equal(g, w)


@pytest.fixture()
def persist_off():
    old = os.environ.get("PDEIP_EXACT_PERSIST")
    os.environ["PDEIP_EXACT_PERSIST"] = "0"
    yield
    if old is None:
        os.environ.pop("PDEIP_EXACT_PERSIST", None)
    else:
        os.environ["PDEIP_EXACT_PERSIST"] = old


@pytest.mark.parametrize("shape", [(32, 48), (97, 131), (5, 300), (260, 7), (388, 584)])
def test_launch_per_front_form(pdeip, oracle, persist_off, shape):
    """The older schedule, one launch per wavefront: same bits, many launches."""
    api = pdeip.mex_api
    api.set_mode(0)
    for it in (1, 4, 9):
        p = pb.elin4(841, *shape, nan_frac=0.01)
        got = api.Oflow_sor_elin4_2d(*p.values(), np.float32(it), np.float32(1.9), np.float32(1))
        if min(shape) > 64:
            assert pdeip.capi.load().pdeip_last_launch_count() > 3
        for g, w in zip(got, oracle.Oflow_sor_elin4_2d(*p.values(), it, 1.9)):
            assert pb.bit_equal(g, w), "elin4 %s it=%d: %s" % (shape, it, pb.describe_mismatch(g, w))
    q = pb.llin4(842, *shape, nan_frac=0.02)
    for g, w in zip(api.Oflow_sor_llin4_2d(*q.values(), np.float32(5), np.float32(1.9), np.float32(1)), oracle.Oflow_sor_llin4_2d(*q.values(), 5, 1.9)):
        assert pb.bit_equal(g, w)
    d = pb.disp4(843, *shape, nan_frac=0.02)
    assert pb.bit_equal(api.Disp_sor_llin4_2d(*d.values(), np.float32(6), np.float32(1.9), np.float32(1)), oracle.Disp_sor_llin4_2d(*d.values(), 6, 1.9))
    e = pb.pde4(844, *shape, nframes=2, nan_frac=0.02)
    assert pb.bit_equal(api.PDEsolver4(*e.values(), np.float32(5), np.float32(1.75), np.float32(1)), oracle.PDEsolver4(*e.values(), 5, 1.75))


def test_release_then_same_shape_again(pdeip, oracle, persist_on):
    """pdeip_release() frees the schedule table of the persistent kernel; the cached (B, iter) shape must go with it, or
    the next call of the same shape reads an uninitialised table (round-1 advisor finding)."""
    api = pdeip.mex_api
    api.set_mode(0)
    p = pb.elin4(831, 97, 260, nan_frac=0.01)
    want = oracle.Oflow_sor_elin4_2d(*p.values(), 4, 1.9)
    for _ in range(2):
        got = api.Oflow_sor_elin4_2d(*p.values(), np.float32(4), np.float32(1.9), np.float32(1))
        _ok(pdeip)
        for g, w in zip(got, want):
            assert pb.bit_equal(g, w), pb.describe_mismatch(g, w)
        assert pdeip.capi.load().pdeip_release() == 0
    # the device-pointer callers share the same per-device cache
    import importlib

    dev = importlib.import_module("pde-based-image-processing_amd.device")
    d = {k: dev.to_device(v) for k, v in p.items()}
    dev.oflow_sor_elin4(*d.values(), 4, 1.9, 0)
    assert pdeip.capi.load().pdeip_release() == 0
    d = {k: dev.to_device(v) for k, v in p.items()}
    dev.oflow_sor_elin4(*d.values(), 4, 1.9, 0)
    assert pb.bit_equal(dev.to_matlab(d["U"]), want[0]) and pb.bit_equal(dev.to_matlab(d["V"]), want[1])
    _ok(pdeip)


def test_set_device_keeps_working(pdeip, oracle):
    """pdeip_set_device() to the same / an invalid device, then a solve (per-device caches, not process-wide statics)."""
    api = pdeip.mex_api
    api.set_mode(1)
    capi = pdeip.capi
    capi.call("pdeip_set_device", 0)
    assert capi.get_devices() == [0]
    with pytest.raises(capi.PdeipError):
        capi.call("pdeip_set_device", 99)
    p = pb.elin4(832, 64, 80)
    got = api.Oflow_sor_elin4_2d(*p.values(), np.float32(4), np.float32(1.9), np.float32(1))
    want = oracle.Oflow_sor_elin4_2d(*p.values(), 4, 1.9, order=oracle.COLOUR)
    for g, w in zip(got, want):
        assert pb.bit_equal(g, w)
    api.set_mode(0)


@pytest.mark.parametrize("knob", ["0", "1"])
def test_schedule_kinds_give_the_same_bits(pdeip, oracle, knob):
    """The walkers take their (strip, sweep) from per-XCD lists when every workgroup of the call is resident (PDEIP_PERSIST_XCD=1,
    the default) and from one list in key order otherwise (more workgroups than compute units, or the knob at 0): placement changes
    speed only.  5-point and 9-point walkers, a grid below the CU count (affine unless the knob says no) and one far above it
    (iter = 40 x 6 strips x 2 frames: one list whatever the knob)."""
    os.environ.pop("PDEIP_EXACT_PERSIST", None)
    old = os.environ.get("PDEIP_PERSIST_XCD")
    os.environ["PDEIP_PERSIST_XCD"] = knob
    try:
        api = pdeip.mex_api
        api.set_mode(0)
        for it in (3, 40):
            p = pb.elin4(831, 150, 330, nan_frac=0.01)
            got = api.Oflow_sor_elin4_2d(*p.values(), np.float32(it), np.float32(1.9), np.float32(1))
            want = oracle.Oflow_sor_elin4_2d(*p.values(), it, 1.9)
            for g, w in zip(got, want):
                assert pb.bit_equal(g, w), "elin4 it=%d knob=%s: %s" % (it, knob, pb.describe_mismatch(g, w))
            e = pb.pde8(832, 90, 330, nframes=2)
            assert pb.bit_equal(api.PDEsolver8(*e.values(), np.float32(it), np.float32(1.75), np.float32(1)), oracle.PDEsolver8(*e.values(), it, 1.75, order=0)), \
                "pde8 it=%d knob=%s" % (it, knob)
            _ok(pdeip)
    finally:
        if old is None:
            os.environ.pop("PDEIP_PERSIST_XCD", None)
        else:
            os.environ["PDEIP_PERSIST_XCD"] = old


def _host_order(B, T, affine):
    """The schedule table as round 2 built it on the host: eight lists, each in (key = b + 2t, t) order."""
    table, items = [0] * 16, []
    for x in range(8):
        table[x] = len(items)
        if not affine and x > 0:
            continue
        for key in range((B - 1) + 2 * (T - 1) + 1):
            for t in range(T):
                b = key - 2 * t
                if 0 <= b < B and (not affine or (b & 7) == x):
                    items.append(b | (t << 16))
    table[8] = len(items)
    return table[:9], items


@pytest.mark.parametrize("B,T", [(1, 1), (1, 7), (5, 3), (8, 4), (9, 4), (60, 4), (60, 20), (23, 40), (200, 2)])
def test_schedule_table_built_on_the_device(pdeip, B, T):
    """k_persist_order writes each item at its rank; the table equals the host-built one (both list kinds), so every dependency
    of an item -- (b-1,t), (b,t-1), (b+1,t-1) -- comes before it in its list order's key."""
    import ctypes
    lib = pdeip.capi.load()
    for affine in (0, 1):
        buf = (ctypes.c_int * (16 + B * T))()
        assert lib.pdeip_debug_persist_order(B, T, affine, buf) == 0, pdeip.capi.last_error()
        hdr, items = _host_order(B, T, affine)
        assert list(buf[:9]) == hdr, (B, T, affine, list(buf[:9]), hdr)
        assert list(buf[16:]) == items, (B, T, affine)


def test_abort_word_is_reported_once_and_survives_a_regrown_control_block(pdeip, oracle):
    """The reporting path of a timed-out dependency wait, without the half second: pdeip_debug_raise_abort sets the sticky word
    as a walker would.  A call made while it is set still terminates (every wait falls through); the host entry point then
    fails loudly instead of handing back invalid planes; pdeip_persist_error reports once and clears; a control block that is
    regrown (a larger call) in between does not lose the report; afterwards results are exact again."""
    lib, api = pdeip.capi.load(), pdeip.mex_api
    api.set_mode(0)
    p = pb.elin4(841, 70, 150)
    run = lambda q=p, it=3: api.Oflow_sor_elin4_2d(*q.values(), np.float32(it), np.float32(1.9), np.float32(1))
    want = oracle.Oflow_sor_elin4_2d(*p.values(), 3, 1.9)
    for g, w in zip(run(), want):
        assert pb.bit_equal(g, w)
    assert lib.pdeip_debug_raise_abort() == 0
    with pytest.raises(api.MexError):          # the gateway checks pdeip_persist_error() before it returns planes
        run()
    assert lib.pdeip_persist_error() == 0      # reported by the failed call: cleared
    for g, w in zip(run(), want):
        assert pb.bit_equal(g, w)
    # latch: raise it, then make a call that needs a larger control block (more strips x sweeps) -- the old block is read before it goes
    assert lib.pdeip_debug_raise_abort() == 0
    big = pb.elin4(842, 40, 2300)
    with pytest.raises(api.MexError):
        run(big, 9)
    assert lib.pdeip_persist_error() == 0
    for g, w in zip(run(), want):
        assert pb.bit_equal(g, w)
