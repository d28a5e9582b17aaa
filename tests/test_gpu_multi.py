"""GPU: the in-process column-slab split behind the host entry points (csrc/pdeip_multi.hip).

One GPU is all the test box has, so the split is exercised with virtual slabs (PDEIP_VIRTUAL_SLABS = n: n slabs dealt over the
device group, here all on device 0): every red-black point solver through the C-ABI's host-pointer gateways must return,
bit for bit, what the single-domain call and the oracle's colour order return -- 2*iter halo columns per cut, colour parity
by the slab's global column, owned columns copied back.  With real devices the same code runs one thread per device."""
import os

import numpy as np
import pytest

import problems as pb

pytestmark = pytest.mark.gpu


@pytest.fixture()
def slabs():
    def set_slabs(n):
        if n:
            os.environ["PDEIP_VIRTUAL_SLABS"] = str(n)
        else:
            os.environ.pop("PDEIP_VIRTUAL_SLABS", None)
    yield set_slabs
    os.environ.pop("PDEIP_VIRTUAL_SLABS", None)


@pytest.mark.parametrize("nslabs", [2, 3, 5])
@pytest.mark.parametrize("shape,it", [((96, 131), 4), ((64, 200), 1), ((128, 260), 7), ((33, 180), 20)])
def test_virtual_slabs_match_single_domain(pdeip, oracle, slabs, nslabs, shape, it):
    api = pdeip.mex_api
    api.set_mode(1)
    try:
        f = lambda v: np.float32(v)
        cases = [
            ("Oflow_sor_elin4_2d", pb.elin4(951, *shape, nan_frac=0.01), 1.9),
            ("Oflow_sor_llin4_2d", pb.llin4(952, *shape, nan_frac=0.01), 1.9),
            ("Disp_sor_llin4_2d", pb.disp4(953, *shape, nan_frac=0.01), 1.9),
            ("PDEsolver4", pb.pde4(954, *shape, nframes=2, nan_frac=0.01), 1.75),
            ("PDEsolver8", pb.pde8(955, *shape, nframes=2, nan_frac=0.01), 1.75),
        ]
        for name, p, omega in cases:
            slabs(0)
            single = getattr(api, name)(*p.values(), f(it), f(omega), f(1))
            slabs(nslabs)
            split = getattr(api, name)(*p.values(), f(it), f(omega), f(1))
            want = getattr(oracle, name)(*p.values(), it, omega, order=oracle.COLOUR)
            single, split, want = [x if isinstance(x, tuple) else (x,) for x in (single, split, want)]
            for a, b, w in zip(single, split, want):
                assert pb.bit_equal(b, a), "%s %s it=%d slabs=%d: split differs from single domain: %s" % (name, shape, it, nslabs, pb.describe_mismatch(b, a))
                assert pb.bit_equal(b, w), "%s: split differs from the oracle" % name
    finally:
        slabs(0)
        api.set_mode(0)


def test_exact_order_and_line_relaxation_are_not_split(pdeip, oracle, slabs):
    """Orderings whose dependency front crosses the frame run on one device whatever the group says."""
    api = pdeip.mex_api
    p = pb.elin4(956, 64, 120)
    slabs(4)
    api.set_mode(0)
    got = api.Oflow_sor_elin4_2d(*p.values(), np.float32(4), np.float32(1.9), np.float32(1))
    want = oracle.Oflow_sor_elin4_2d(*p.values(), 4, 1.9)
    assert pb.bit_equal(got[0], want[0]) and pb.bit_equal(got[1], want[1])
    got = api.Oflow_sor_elin4_2d(*p.values(), np.float32(2), np.float32(1.5), np.float32(2))
    want = oracle.Oflow_sor_elin4_2d(*p.values(), 2, 1.5, solver=2)
    assert pb.bit_equal(got[0], want[0]) and pb.bit_equal(got[1], want[1])
    slabs(0)


def test_set_devices_on_a_one_gpu_box(pdeip):
    capi = pdeip.capi
    capi.set_devices([0])
    assert capi.get_devices() == [0]
    with pytest.raises(capi.PdeipError):
        capi.set_devices([0, 0])
    with pytest.raises(capi.PdeipError):
        capi.set_devices([0, 63])


@pytest.mark.parametrize("nthreads", [2, 3])
def test_virtual_slabs_on_worker_threads(pdeip, oracle, slabs, nthreads):
    """PDEIP_VIRTUAL_THREADS: the slabs of a one-device group are dealt over n worker threads, each with a device-state slot of
    its own -- the std::thread branch a real device group takes, with everything the workers share (error text and launch
    counter per thread, atomic workspace generation, locked profiling slots).  Same bits as the single domain, the launch
    counter is the sum over the workers, and a worker's failure reaches the caller with its own message."""
    api, lib = pdeip.mex_api, pdeip.capi.load()
    api.set_mode(1)
    os.environ["PDEIP_VIRTUAL_THREADS"] = str(nthreads)
    try:
        f = lambda v: np.float32(v)
        for name, p, omega in [("Oflow_sor_elin4_2d", pb.elin4(961, 128, 300, nan_frac=0.01), 1.9),
                               ("Disp_sor_llin4_2d", pb.disp4(962, 96, 260, nan_frac=0.01), 1.9),
                               ("PDEsolver8", pb.pde8(963, 64, 240, nframes=2), 1.75)]:
            for rep in range(3):  # repeated: the workers' slots keep their workspace between calls
                slabs(0)
                single = getattr(api, name)(*p.values(), f(4), f(omega), f(1))
                slabs(5)
                lib.pdeip_profile_enable(1)
                split = getattr(api, name)(*p.values(), f(4), f(omega), f(1))
                lib.pdeip_profile_enable(0)
                assert lib.pdeip_last_launch_count() >= 5, "five slabs, at least one launch each"
                single, split = [x if isinstance(x, tuple) else (x,) for x in (single, split)]
                for a, b in zip(single, split):
                    assert pb.bit_equal(b, a), "%s threads=%d: %s" % (name, nthreads, pb.describe_mismatch(b, a))
    finally:
        os.environ.pop("PDEIP_VIRTUAL_THREADS", None)
        slabs(0)
        api.set_mode(0)
        lib.pdeip_release()
