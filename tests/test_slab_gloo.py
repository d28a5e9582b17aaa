"""CPU, world_size 2 and 3 over gloo: the column-slab decomposition + wide-halo exchange (slab.py) gives,
on the columns each rank owns, bit for bit what the single-domain red-black sweep gives.

The local relaxation is the injected oracle sweep (no GPU here); on the GPU box the same SlabSolver runs
the HIP kernels (tests/test_gpu_slab.py, bench.py --gpus N)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NROWS, NCOLS = 37, 61   # odd sizes: slabs of unequal width, odd col0 on some ranks


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_sweep(kind):
    import oracle_lib as orc

    def np_view(t):  # torch [ncols_local, nrows] C-order  ==  numpy [nrows, ncols_local] F-order (zero copy)
        return t.numpy().T

    def fn(iterate, coef, k, omega, col0):
        order = orc.COLOUR | ((col0 & 1) << 1)
        it = [np_view(t) for t in iterate]
        cf = [np_view(t) for t in coef]
        if kind == "elin4":
            U, V = orc.oflow_sor_elin4(it[0], it[1], *cf, k, omega, order)
            it[0][...], it[1][...] = U, V
        elif kind == "llin4":
            dU, dV = orc.oflow_sor_llin4(cf[0], cf[1], it[0], it[1], *cf[2:], k, omega, order)
            it[0][...], it[1][...] = dU, dV
        elif kind == "disp4":
            it[0][...] = orc.disp_sor_llin4(cf[0], it[0], *cf[1:], k, omega, order)
        elif kind == "pde4":
            it[0][...] = orc.pde_sor4(it[0], *cf, k, omega, order)
        elif kind == "pde8":
            it[0][...] = orc.pde_sor8(it[0], *cf, k, omega, order)
    return fn


def _problem(kind):
    import problems as pb

    if kind == "elin4":
        p = pb.elin4(401, NROWS, NCOLS, nan_frac=0.03)
        return [p["U"], p["V"]], [p[k] for k in ("M", "Cu", "Cv", "Du", "Dv", "wW", "wN", "wE", "wS")]
    if kind == "llin4":
        p = pb.llin4(402, NROWS, NCOLS)
        return [p["dU"], p["dV"]], [p[k] for k in ("U", "V", "M", "Cu", "Cv", "Du", "Dv", "wW", "wN", "wE", "wS")]
    if kind == "disp4":
        p = pb.disp4(403, NROWS, NCOLS)
        return [p["dU"]], [p[k] for k in ("U", "Cu", "Du", "wW", "wN", "wE", "wS")]
    if kind == "pde8":
        p = pb.pde8(405, NROWS, NCOLS)
        return [p["X"]], [p[k] for k in ("TRACE", "B", "wW", "wNW", "wN", "wNE", "wE", "wSE", "wS", "wSW")]
    p = pb.pde4(404, NROWS, NCOLS)
    return [p["X"]], [p[k] for k in ("TRACE", "B", "wW", "wN", "wE", "wS")]


def _worker(rank, world, port, kind, iters, k, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        slab = importlib.import_module("pde-based-image-processing_amd.slab")
        iterate, coef = _problem(kind)
        to_t = lambda a: torch.from_numpy(np.ascontiguousarray(a.T))  # MATLAB [nrows,ncols] -> [ncols,nrows]
        dom = slab.SlabDomain(NCOLS, NROWS, rank, world, halo=2 * k)
        it_l = [dom.slice_local(to_t(a)) for a in iterate]
        cf_l = [dom.slice_local(to_t(a)) for a in coef]
        slab.SlabSolver(dom, kind, sweeps_per_exchange=k, sweep_fn=_oracle_sweep(kind)).solve(it_l, cf_l, iters, 1.7)
        gathered = [dom.gather_owned(t) for t in it_l]
        if rank == 0:
            np.savez(out_path, *[g.numpy().T for g in gathered])
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("kind,iters,k", [("elin4", 4, 4), ("elin4", 7, 3), ("llin4", 4, 2), ("disp4", 5, 4), ("pde4", 4, 4), ("pde8", 5, 2)])
def test_slabs_match_single_domain(tmp_path, oracle, world, kind, iters, k):
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker, args=(world, _free_port(), kind, iters, k, out), nprocs=world, join=True)
    got = np.load(out)
    iterate, coef = _problem(kind)
    single = [t.clone() for t in map(lambda a: torch.from_numpy(np.ascontiguousarray(a.T)), iterate)]
    _oracle_sweep(kind)(single, [torch.from_numpy(np.ascontiguousarray(a.T)) for a in coef], iters, 1.7, 0)
    import problems as pb
    for f, s in enumerate(single):
        assert pb.bit_equal(got["arr_%d" % f], s.numpy().T), "%s field %d: %s" % (kind, f, pb.describe_mismatch(got["arr_%d" % f], s.numpy().T))


def _worker_calls(rank, world, port, kind, calls, k, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        slab = importlib.import_module("pde-based-image-processing_amd.slab")
        iterate, coef = _problem(kind)
        to_t = lambda a: torch.from_numpy(np.ascontiguousarray(a.T))
        dom = slab.SlabDomain(NCOLS, NROWS, rank, world, halo=2 * k)
        it_l = [dom.slice_local(to_t(a)) for a in iterate]
        cf_l = [dom.slice_local(to_t(a)) for a in coef]
        exchanges = [0]
        real = dom.exchange
        dom.exchange = lambda fields: (exchanges.__setitem__(0, exchanges[0] + 1), real(fields))[1]
        solver = slab.SlabSolver(dom, kind, sweeps_per_exchange=k, sweep_fn=_oracle_sweep(kind))
        for n in calls:
            solver.solve(it_l, cf_l, n, 1.7)
        gathered = [dom.gather_owned(t) for t in it_l]
        if rank == 0:
            np.savez(out_path, *[g.numpy().T for g in gathered], exchanges=np.array(exchanges))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind,calls,k,want_exchanges", [("elin4", (2, 2, 2, 2), 4, 2), ("elin4", (4, 4, 4), 8, 2), ("pde4", (3, 3, 3), 4, 3)])
def test_halo_budget_carries_across_calls(tmp_path, oracle, kind, calls, k, want_exchanges):
    """Several solver calls between two exchanges (what bench.py does at N > 1): same bits as one domain, fewer exchanges."""
    out = str(tmp_path / "gathered.npz")
    mp.spawn(_worker_calls, args=(2, _free_port(), kind, calls, k, out), nprocs=2, join=True)
    got = np.load(out)
    assert int(got["exchanges"][0]) == want_exchanges
    iterate, coef = _problem(kind)
    single = [torch.from_numpy(np.ascontiguousarray(a.T)).clone() for a in iterate]
    cf = [torch.from_numpy(np.ascontiguousarray(a.T)) for a in coef]
    for n in calls:
        _oracle_sweep(kind)(single, cf, n, 1.7, 0)
    import problems as pb
    for f, s in enumerate(single):
        assert pb.bit_equal(got["arr_%d" % f], s.numpy().T), "%s field %d: %s" % (kind, f, pb.describe_mismatch(got["arr_%d" % f], s.numpy().T))


def _worker_c5(rank, world, port, outer, k, out_path):
    """BASELINE config C5's inner loop on slabs: outer x [DdiffWeights(U + dU) -- a radius-1 stencil stage --, k disparity sweeps]."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib as orc
        slab = importlib.import_module("pde-based-image-processing_amd.slab")
        iterate, coef = _problem("disp4")
        to_t = lambda a: torch.from_numpy(np.ascontiguousarray(a.T))
        dom = slab.SlabDomain(NCOLS, NROWS, rank, world, halo=2 * k + 1)
        it_l = [dom.slice_local(to_t(a)) for a in iterate]
        cf_l = [dom.slice_local(to_t(a)) for a in coef]          # U, Cu, Du, wW, wN, wE, wS
        exchanges = [0]
        real = dom.exchange
        dom.exchange = lambda fields: (exchanges.__setitem__(0, exchanges[0] + 1), real(fields))[1]
        solver = slab.SlabSolver(dom, "disp4", sweeps_per_exchange=k, sweep_fn=_oracle_sweep("disp4"))

        def weights():
            D = (cf_l[0] + it_l[0]).numpy().T                   # U + dU on the local slab
            for dst, w in zip(cf_l[3:], _c5_weights(orc, D)):
                dst.numpy().T[...] = w
        for _ in range(outer):
            solver.stage(it_l, 1, weights)
            solver.solve(it_l, cf_l, k, 1.7)
        gathered = [dom.gather_owned(t) for t in it_l]
        if rank == 0:
            np.savez(out_path, gathered[0].numpy().T, exchanges=np.array(exchanges))
    finally:
        dist.destroy_process_group()


def _c5_weights(orc, D):
    return orc.diffweights6(np.asfortranarray(D), 0.001)   # wW, wN, wE, wS


@pytest.mark.parametrize("world", [2, 3])
def test_weights_plus_sweeps_loop_on_slabs(tmp_path, oracle, world):
    """stage(): a stencil evaluated locally (diffusion weights from the iterate, radius 1) draws on the same halo budget as the
    sweeps, so the loop `weights, 4 sweeps` runs sliced with one exchange per pass of halo 9 and gives the single-domain bits."""
    import oracle_lib as orc
    import problems as pb
    outer, k = 3, 4
    out = str(tmp_path / "c5.npz")
    mp.spawn(_worker_c5, args=(world, _free_port(), outer, k, out), nprocs=world, join=True)
    got = np.load(out)
    assert int(got["exchanges"][0]) == outer
    iterate, coef = _problem("disp4")
    dU = np.asfortranarray(iterate[0].copy())
    cf = [np.asfortranarray(c.copy()) for c in coef]
    for _ in range(outer):
        cf[3:] = [np.asfortranarray(w) for w in _c5_weights(orc, cf[0] + dU)]
        dU = orc.disp_sor_llin4(cf[0], dU, *cf[1:], k, 1.7, orc.COLOUR)
    assert pb.bit_equal(got["arr_0"], dU), pb.describe_mismatch(got["arr_0"], dU)


def test_split_columns_and_halo_checks():
    slab = importlib.import_module("pde-based-image-processing_amd.slab")
    assert slab.split_columns(10, 3) == [(0, 4), (4, 7), (7, 10)]
    d = slab.SlabDomain(3840, 2160, 3, 8, halo=8)
    assert (d.c0, d.c1, d.lo, d.hi, d.col0) == (1440, 1920, 1432, 1928, 1432)
    with pytest.raises(ValueError):
        slab.SlabDomain(40, 10, 0, 8, halo=8)
    with pytest.raises(ValueError):
        slab.SlabSolver(slab.SlabDomain(3840, 2160, 1, 2, halo=4), sweeps_per_exchange=4)
    # a stencil stage wider than the halo can never be covered by an exchange
    sv = slab.SlabSolver(slab.SlabDomain(3840, 2160, 1, 2, halo=8), sweeps_per_exchange=4, sweep_fn=lambda *a: None)
    with pytest.raises(ValueError):
        sv.stage([], 9, lambda: None)
    # one rank: no halo, no exchange, the stage just runs
    one = slab.SlabSolver(slab.SlabDomain(64, 32, 0, 1, halo=0), sweeps_per_exchange=4, sweep_fn=lambda *a: None)
    assert one.stage([], 3, lambda: 7) == 7


def test_single_rank_without_halo_solves_in_one_run():
    """world == 1, halo == 0 (what bench.py builds at N = 1): solve() and solve_pingpong() run all sweeps at once -- the halo
    budget only exists where a cut does (round 2 derived the sweep room from the halo and never advanced here)."""
    slab = importlib.import_module("pde-based-image-processing_amd.slab")
    calls = []

    def fn(it, coef, k, omega, col0, out=None):
        calls.append((k, out is not None))
        if out is not None:
            for a, b in zip(it, out):
                b.copy_(a + k)
    fn.out_of_place = True
    t = torch.zeros(64, 32)
    sv = slab.SlabSolver(slab.SlabDomain(64, 32, 0, 1, halo=0), sweeps_per_exchange=4, sweep_fn=fn)
    sv.solve([t], [], 6, 1.7)
    assert calls == [(6, False)]
    cur = sv.solve_pingpong([t], [], 5, 1.7)
    assert calls[-1] == (5, True) and float(cur[0][0, 0]) == 5.0
    # with a cut, a halo below two columns covers no sweep: refuse instead of looping
    sv2 = slab.SlabSolver.__new__(slab.SlabSolver)
    sv2.dom, sv2.k, sv2.spent, sv2.sweep_fn = slab.SlabDomain(64, 32, 0, 2, halo=1), 1, 0, fn
    with pytest.raises(ValueError):
        sv2.solve([t], [], 1, 1.7)
