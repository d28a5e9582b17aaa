#!/usr/bin/env python3
"""bench.py -- SOR sweeps ("iterations")/s and achieved HBM GB/s on a 3840x2160 flow problem.

Contract (driver):  python bench.py --gpus N --steps K --warmup W   (N>1: launched by
torch.distributed.run, one rank per GPU).  Prints ONE JSON line on rank 0.

Workload (BASELINE.md row "M"/"C4"; BASELINE.json metric): the early-linearization (Horn-Schunck
type) point solver `Oflow_sor_elin4_2d` / GS_SOR_elin4_2d on one 2160 x 3840 float32 frame
(nrows=2160, ncols=3840), synthetic motion-tensor coefficients, symmetric weights U(0.5,5) (wE of a
pixel = wW of its east neighbour, as the drivers' diffusion weights are: a convergent problem), omega 1.9.
One STEP = one solver call of iter=4 sweeps on buffers resident in HBM (what the FMG smoother
issues per call, FlowEminNDFASFMG_elin_2D_v10.m:55-56,398-411), so value = 4*K / time.

  * value        RED_BLACK ordering (the ordering that decomposes across GPUs; the same kernel
                 at every N so the driver's scaling series is one algorithm).  N>1: the frame is
                 cut into column slabs with a 2*iter-column halo, one RCCL exchange per step.
  * exact_order  (N=1 only) the same workload in the reference's lexicographic order, the
                 library's default and the bit-parity mode, reported beside it.
  * roofline     algorithmic bytes (52 B/pixel/sweep: 11 planes read + 2 written) / average
                 sweep-kernel launch duration, measured with HIP events on the launch stream
                 inside this run (pdeip_profile_*), vs 8 TB/s HBM3E peak; beside it frac_hbm =
                 measured HBM traffic of the same launch (PMC, profiles/traffic.json) / time / peak
                 and overfetch = traffic / the bytes a perfectly fused launch needs.
  * parity_rb    how far the timed ordering is from the reference's result: RMS / max of RED_BLACK
                 vs the CPU oracle's lexicographic result after the first iter=4 call, and the
                 sweep count from which the two orderings agree to 1e-4 RMS on this frame.
  * host_call    wall time of pdeip_oflow_sor_elin4 with HOST pointers (what a MEX stub calls):
                 PCIe-inclusive, both orderings; never `value`.
  * cpu_baseline the CPU oracle (plain-C port of the reference loop, lexicographic order) on
                 the same frame on ONE host core, a bounded number of calls; plus the oracle's
                 red-black order on all host cores (OpenMP), the like-for-like comparator of value.
"""
import argparse
import ctypes
import importlib
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NROWS, NCOLS = 2160, 3840
ITER, OMEGA = 4, 1.9
BYTES_PER_PIXEL_SWEEP = 52.0  # 11 R + 2 W float32 planes
HBM_PEAK_GBS = 8000.0         # MI355X_MICROARCH.md: 8 TB/s spec


def env_int_py(name, default):
    try:
        return int(os.environ.get(name, default))
    except ValueError:
        return default


def make_planes(torch, device, nrows, ncols, seed=0):
    """Motion-tensor structured coefficients (positive semi-definite data term), U(0.5,5) weights."""
    g = torch.Generator(device=device).manual_seed(seed)

    def u(lo, hi):
        return torch.empty((ncols, nrows), device=device, dtype=torch.float32).uniform_(lo, hi, generator=g)

    # image-gradient sized data terms (|Ix|,|Iy| <= 0.5) against weights in [0.5, 5]
    a, b, c = u(-0.5, 0.5), u(-0.5, 0.5), u(-1.0, 1.0)
    wW, wN, wE, wS = [u(0.5, 5.0) for _ in range(4)]
    # a symmetric operator, as OPdiffWeights / DdiffWeights produce: wE(i,j) = wW(i,j+1), wS(i,j) = wN(i+1,j)
    # (tensor layout [ncols, nrows]: dim 0 = image column j, dim 1 = image row i).  With independently drawn
    # weights SOR at omega = 1.9 diverges in either ordering; this problem converges (tests/test_oracle_math.py
    # checks the same recipe on the CPU) and main() asserts the iterate is finite after the timed loop.
    wE[:-1, :] = wW[1:, :]
    wS[:, :-1] = wN[:, 1:]
    coef = [a * b, -a * c, -b * c, a * a, b * b, wW, wN, wE, wS]  # M,Cu,Cv,Du,Dv,wW,wN,wE,wS
    U, V = u(-1.0, 1.0), u(-1.0, 1.0)
    return U, V, [t.contiguous() for t in coef]


def cpu_baseline(U, V, coef, calls):
    """Time the CPU oracle (reference order, 1 thread) on the same frame.  Returns (sweeps/s, result U,V)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    lib = oracle_lib.lib()
    hu, hv = U.cpu().numpy().copy(), V.cpu().numpy().copy()   # [ncols, nrows] C-order == MATLAB column-major
    hc = [t.cpu().numpy() for t in coef]
    first = None
    t0 = time.perf_counter()
    for _ in range(calls):
        lib.orc_oflow_sor_elin4(hu.ctypes.data, hv.ctypes.data, *[a.ctypes.data for a in hc], NROWS, NCOLS, ITER,
                                ctypes.c_float(OMEGA), 0)
        if first is None:
            first = (hu.copy(), hv.copy())
    dt = time.perf_counter() - t0
    return calls * ITER / dt, first


def cpu_red_black_all_cores(U, V, coef, calls):
    """The oracle's red-black order on every host core (OpenMP): the like-for-like CPU comparator of `value`."""
    import oracle_lib

    lib = oracle_lib.lib()

    def spread(t):
        """Host copy whose pages are first touched by the threads that will sweep them (NUMA placement)."""
        src = t.cpu().numpy()
        dst = np.empty_like(src)
        lib.orc_plane_copy_omp(dst.ctypes.data, src.ctypes.data, NROWS, NCOLS, 0)
        return dst

    hu, hv = spread(U), spread(V)
    hc = [spread(t) for t in coef]
    args = [hu.ctypes.data, hv.ctypes.data] + [a.ctypes.data for a in hc] + [NROWS, NCOLS, ITER, ctypes.c_float(OMEGA), 0]
    used = lib.orc_oflow_sor_elin4_rb_omp(*args)  # warm-up: thread pool, page placement
    t0 = time.perf_counter()
    for _ in range(calls):
        used = lib.orc_oflow_sor_elin4_rb_omp(*args)
    return calls * ITER / (time.perf_counter() - t0), int(used)


def host_call(capi, U, V, coef, mode, reps=7):
    """Median wall time of pdeip_oflow_sor_elin4 (host pointers in, host pointers out) for one iter=4 call at 4K."""
    lib = capi.load()
    hu, hv = U.cpu().numpy().copy(), V.cpu().numpy().copy()
    hc = [t.cpu().numpy() for t in coef]
    ou, ov = np.empty_like(hu), np.empty_like(hv)
    old = capi.get_mode()
    capi.set_mode(mode)
    times = []
    try:
        for k in range(reps + 1):
            t0 = time.perf_counter()
            rc = lib.pdeip_oflow_sor_elin4(hu.ctypes.data, hv.ctypes.data, *[a.ctypes.data for a in hc], NROWS, NCOLS, 1, ITER,
                                           ctypes.c_float(OMEGA), 1, ou.ctypes.data, ov.ctypes.data, None, None)
            capi.check(rc)
            if k:
                times.append(time.perf_counter() - t0)
    finally:
        capi.set_mode(old)
    return sorted(times)[len(times) // 2] * 1e3, ou, ov


def main():
    # The bench owns the device: the exact-order walkers may take the per-XCD lists (an opt-in because they are live only while
    # the call's whole grid is resident: DESIGN.md 5.2).  Reported in config.knobs; PDEIP_PERSIST_XCD=0 in the environment wins.
    os.environ.setdefault("PDEIP_PERSIST_XCD", "1")
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cpu-calls", type=int, default=40,
                    help="solver calls timed for the CPU baseline (0 = skip); 40 calls x 4 sweeps take ~9 s on one core")
    ap.add_argument("--headline-only", action="store_true",
                    help="only the timed red-black workload (what the rocprofv3 summary under profiles/ is taken from)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("PDEIP_DIST_BACKEND", "nccl") != "nccl":
        local_rank = 0  # rehearsal: every rank shares GPU 0
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP library is the product, there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("PDEIP_DIST_BACKEND", "nccl")  # "gloo": rehearsal of the N>1 path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    pkg = importlib.import_module("pde-based-image-processing_amd")
    dev = importlib.import_module("pde-based-image-processing_amd.device")
    slab = importlib.import_module("pde-based-image-processing_amd.slab")
    capi = pkg.capi
    capi.load()

    U0, V0, coef_full = make_planes(torch, device, NROWS, NCOLS)
    N = NROWS * NCOLS

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps, warmup):
        for _ in range(warmup):
            step()
        barrier()
        capi.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        dt = time.perf_counter() - t0
        ms, nl = capi.profile_read()
        capi.profile_enable(False)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, ms, nl

    # ---- RED_BLACK ordering: value ----------------------------------------------------------------
    # N > 1: a 2k-column halo lets k sweeps run between two halo exchanges; k = 16 = four solver calls (the step is 0.2 ms of
    # kernel time at N = 1 and ~55 us split eight ways, an RCCL exchange is latency-bound: fewer, wider ones; 64 extra columns
    # of work per interior slab).  Owned columns stay bit-exact.
    k_ex = max(ITER, env_int_py("PDEIP_SLAB_SWEEPS_PER_EXCHANGE", 4 * ITER))
    dom = slab.SlabDomain(NCOLS, NROWS, rank, world, halo=2 * k_ex)
    U, V = dom.slice_local(U0), dom.slice_local(V0)
    coef = [dom.slice_local(t) for t in coef_full]
    solver = slab.SlabSolver(dom, "elin4", sweeps_per_exchange=k_ex)

    if world == 1:
        # one GPU: every step relaxes the current iterate into the other of two plane sets (pdeip_oflow_sor_elin4_dev_to: what a
        # gateway does -- input read, output written -- and no device-to-device copy of the iterate after the launch)
        sets, cur = [(U, V), (torch.empty_like(U), torch.empty_like(V))], [0]

        def step_rb():
            a, b = sets[cur[0]], sets[1 - cur[0]]
            dev.oflow_sor_elin4(a[0], a[1], *coef, ITER, OMEGA, capi.MODE_RED_BLACK, out=b)
            cur[0] ^= 1
    else:
        state = [[U, V]]

        def step_rb():  # the slab form of the same thing: the solver returns the plane set that holds the iterate now
            state[0] = solver.solve_pingpong(state[0], coef, ITER, OMEGA)

    # The driver's own warm-up (W steps) and K timed steps first, on a device that has just been handed over: `value_cold`.
    # Then an untimed pre-warm brings the device to its steady clocks (measured on one box, K = 50: W = 5 26.9 k, W = 50 29.2 k,
    # W = 200 32.9 k sweeps/s): the metric is a steady-state throughput, so `value` is W more warm-up steps + K timed steps AFTER
    # PREWARM untimed calls (reported as config.prewarm_steps), and the same K-step loop is repeated REPEATS - 1 more times: `value_repeats`
    # holds min / median / max of all of them (the timed region is a few ms: single runs move by 2-4 %).
    calls_done = [0]
    inner_step = step_rb

    def step_counted():
        inner_step()
        calls_done[0] += 1

    step_rb = step_counted
    cold_dt, _, _ = timed(step_rb, args.steps, args.warmup)
    value_cold = args.steps * ITER / cold_dt
    PREWARM = env_int_py("PDEIP_BENCH_PREWARM", 300)
    for _ in range(PREWARM):
        step_rb()
    barrier()
    dt, ms, nl = timed(step_rb, args.steps, args.warmup)
    value = args.steps * ITER / dt
    REPEATS = max(1, env_int_py("PDEIP_BENCH_REPEATS", 5))
    rates = [value]
    for _ in range(REPEATS - 1):
        rdt, _, _ = timed(step_rb, args.steps, 0)
        rates.append(args.steps * ITER / rdt)
    if world == 1:
        U, V = sets[cur[0]]
    else:
        U, V = state[0]
    if not (bool(torch.isfinite(U).all()) and bool(torch.isfinite(V).all())):
        raise SystemExit("bench.py: the iterate is not finite after the timed loop (the workload must converge)")
    # N > 1: what the slabs hold must be, bit for bit, what ONE domain holds after the same number of red-black sweeps.  Rank 0
    # gathers the owned columns and relaxes the whole frame by the same number of solver calls (outside every timed region).
    parity_slabs = None
    if world > 1:
        gU, gV = dom.gather_owned(U), dom.gather_owned(V)
        if rank == 0:
            a, b2 = (U0.clone(), V0.clone()), (torch.empty_like(U0), torch.empty_like(V0))
            for _ in range(calls_done[0]):
                dev.oflow_sor_elin4(a[0], a[1], *coef_full, ITER, OMEGA, capi.MODE_RED_BLACK, out=b2)
                a, b2 = b2, a
            torch.cuda.synchronize()
            du, dv = (gU - a[0]).abs().max().item(), (gV - a[1]).abs().max().item()
            parity_slabs = {"max_abs": float(max(du, dv)), "bit_identical": bool(torch.equal(gU, a[0]) and torch.equal(gV, a[1])),
                            "sweeps": calls_done[0] * ITER,
                            "what": "owned columns of all ranks gathered on rank 0 vs the single-domain red-black solver after the same %d calls" % calls_done[0]}
            del a, b2
        del gU, gV
    # per launch: this rank's pixels (owned + halo columns are all relaxed by the launch; count owned only)
    own_px = (dom.c1 - dom.c0) * NROWS
    launch_s = ms * 1e-3 / max(nl, 1)
    # the library fuses up to four sweeps into one launch (k_sor_rbp): a launch of k sweeps has to move every plane ONCE --
    # 9 coefficient planes + the iterate in + the iterate out = 13 planes -- whatever k is.  That is the byte count the launch
    # is priced with (`achieved`, `frac`); SURVEY 8(d)'s per-sweep figure (52 B x pixels x sweeps in the launch) is the
    # EFFECTIVE rate beside it, which exceeds the HBM peak by construction once sweeps are fused.
    sweeps_per_launch = args.steps * ITER / max(nl, 1)
    fused_min_bytes = (9 + 2 + 2) * 4.0 * own_px
    effective_bytes = BYTES_PER_PIXEL_SWEEP * own_px * sweeps_per_launch
    achieved = fused_min_bytes / launch_s / 1e9
    effective = effective_bytes / launch_s / 1e9
    k_fused = int(round(sweeps_per_launch))
    kernel_name = ("k_sor_rbp<ModelElin4, %d sweeps per launch>" % k_fused) if k_fused >= 4 else ("k_sor_rb<ModelElin4, sweeps per launch = %d>" % k_fused)

    out = {
        "metric": "SOR iterations/sec (3840x2160 flow)", "value": round(value, 2), "unit": "iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 5),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "value_cold": round(value_cold, 2),
        "value_repeats": {"n": len(rates), "min": round(min(rates), 2), "median": round(sorted(rates)[len(rates) // 2], 2), "max": round(max(rates), 2)},
        "config": {"workload": "Oflow_sor_elin4_2d point SOR, 2160x3840 f32 frame, iter=4/call, omega=1.9, resident in HBM",
                   "ordering": "red_black", "prewarm_steps": PREWARM, "decomposition": "column slabs, %d-column halo, 1 RCCL exchange per %d sweeps" % (2 * k_ex, k_ex)
                   if world > 1 else "single GPU",
                   "knobs": {k: os.environ[k] for k in sorted(os.environ) if k.startswith("PDEIP_")}},
        "roofline": {"bound": "hbm", "kernel": kernel_name,
                     "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                     "launch_us": round(launch_s * 1e6, 2), "bytes_per_launch": fused_min_bytes,
                     "sweeps_per_launch": round(sweeps_per_launch, 3),
                     "effective_achieved": round(effective, 1), "effective_frac": round(effective / HBM_PEAK_GBS, 4),
                     "effective_bytes_per_launch": effective_bytes,
                     "limiter": "the compute units' memory pipelines: 443 MB of column pieces (eleven planes in, halo included) + 66 MB out per "
                                "launch at 22 GB/s per unit = 5.6-5.7 TB/s chip-wide, the rate the library's plain streaming kernels reach; the same "
                                "launch with every workgroup barrier removed (waves free-running, results invalid) takes 92 us against 96 us; "
                                "fewer HBM bytes (L2 hits), a deeper DMA lead and a second loader wave change nothing (DESIGN.md 5.1, profiles/NOTES.md R3.4)",
                     "note": "achieved/frac = the bytes a launch of k fused sweeps must move (13 planes x 4 B x pixels) / launch time / peak; "
                             "effective_* = SURVEY 8(d)'s per-sweep figure (52 B x pixels x k sweeps) / launch time -- an effective rate that can "
                             "exceed what HBM delivers; frac_hbm = measured traffic / time / peak is the HBM utilisation, overfetch = traffic / bytes_per_launch"},
    }
    if parity_slabs is not None:
        out["parity_slabs"] = parity_slabs
    tr = os.path.join(ROOT, "profiles", "traffic.json")
    if world == 1 and os.path.exists(tr):
        try:
            table = json.load(open(tr))
            key = "k_sor_rb%s_elin4_2160x3840_bytes_per_launch" % ("" if k_fused == 1 else k_fused)
            traffic = table.get(key)
            out["roofline"]["traffic"] = traffic
            if traffic:
                # what the memory system actually moved for this launch (PMC FETCH_SIZE x 2 + WRITE_SIZE, separate passes)
                fused_min = fused_min_bytes
                out["roofline"]["hbm_gbs_from_traffic"] = round(traffic / launch_s / 1e9, 1)
                out["roofline"]["frac_hbm"] = round(traffic / launch_s / 1e9 / HBM_PEAK_GBS, 4)
                out["roofline"]["overfetch"] = round(traffic / fused_min, 3)
                out["roofline"]["traffic_source"] = table.get("source", "profiles/traffic.json")
        except (ValueError, OSError):
            pass

    if world == 1 and not args.headline_only:
        # ---- exact (reference) ordering on the same workload -------------------------------------
        Ue, Ve = U0.clone(), V0.clone()

        def step_exact():
            dev.oflow_sor_elin4(Ue, Ve, *coef_full, ITER, OMEGA, capi.MODE_EXACT_ORDER)

        e_steps = max(20, args.steps // 10)   # every reported leg times at least 20 calls
        edt, ems, enl = timed(step_exact, e_steps, max(3, args.warmup // 10))
        e_launch_s = ems * 1e-3 / max(enl, 1)
        e_bytes = BYTES_PER_PIXEL_SWEEP * N * ITER * e_steps / max(enl, 1)  # tiles of one front per launch
        e_kernel = "k_sor_walk<ModelElin4>" if os.environ.get("PDEIP_EXACT_WALK", "0") not in ("", "0") else "k_sor_exact_persist<ModelElin4>"
        e_traffic = None
        try:   # PMC FETCH_SIZE x 2 + WRITE_SIZE of the walkers' launch, per schedule kind (profiles/traffic.json)
            tkey = "k_sor_exact_persist_elin4_2160x3840_exact_" + ("xcd_affine" if os.environ.get("PDEIP_PERSIST_XCD", "0") not in ("", "0") else "single_list")
            e_traffic = json.load(open(tr)).get(tkey, {}).get("bytes_per_launch") if e_kernel.startswith("k_sor_exact") else None
        except (ValueError, OSError):
            pass
        out["exact_order"] = {
            "value": round(e_steps * ITER / edt, 2), "unit": "iterations/s", "ms_per_step": round(edt / e_steps * 1e3, 4),
            "launches_per_step": enl // e_steps, "timed_calls": e_steps,
            "schedule": "per-XCD lists (PDEIP_PERSIST_XCD=1: needs the grid resident)" if os.environ.get("PDEIP_PERSIST_XCD", "0") not in ("", "0") else "single key-ordered list (default)",
            "roofline": {"bound": "hbm", "kernel": e_kernel, "achieved": round(e_bytes / e_launch_s / 1e9, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(e_bytes / e_launch_s / 1e9 / HBM_PEAK_GBS, 4),
                         "traffic": e_traffic, "launch_us": round(e_launch_s * 1e6, 2),
                         "frac_hbm": round(e_traffic / e_launch_s / 1e9 / HBM_PEAK_GBS, 4) if e_traffic else None,
                         "limiter": "one compute unit's memory pipeline per 16-row chunk (46 KB in + 8 KB out per walker) and the strips' start-up "
                                    "chain: DESIGN.md 5.2 (stamped model), profiles/NOTES.md R3.1"},
        }
        # the same frame with iter=20 per call (the Horn-Schunck driver's setting): the persistent kernel
        def step_exact20():
            dev.oflow_sor_elin4(Ue, Ve, *coef_full, 20, OMEGA, capi.MODE_EXACT_ORDER)

        l_steps = max(20, args.steps // 40)
        ldt, _, _ = timed(step_exact20, l_steps, 2)
        out["exact_order"]["iter20_per_call"] = {"value": round(l_steps * 20 / ldt, 2), "unit": "iterations/s",
                                                 "ms_per_call": round(ldt / l_steps * 1e3, 4), "kernel": "k_sor_exact_persist<ModelElin4>"}
        capi.call("pdeip_persist_error")
        # ---- CPU baseline + parity of the first call -----------------------------------------------
        if args.cpu_calls > 0:
            cpu_rate, first = cpu_baseline(U0, V0, coef_full, args.cpu_calls)
            Up, Vp = U0.clone(), V0.clone()
            dev.oflow_sor_elin4(Up, Vp, *coef_full, ITER, OMEGA, capi.MODE_EXACT_ORDER)
            torch.cuda.synchronize()
            du = Up.cpu().numpy().astype(np.float64) - first[0]
            dv = Vp.cpu().numpy().astype(np.float64) - first[1]
            out["cpu_baseline"] = {"value": round(cpu_rate, 3), "unit": "iterations/s", "cores": 1, "kind": "port",
                                   "sample": "%d calls x iter=%d of the same 2160x3840 frame, reference (lexicographic) order, "
                                             "oracle/pdeip_oracle.c built -O2 -ffp-contract=off" % (args.cpu_calls, ITER),
                                   "host_cpus": os.cpu_count()}
            out["parity"] = {"mode": "exact_order vs cpu oracle, first call", "rms_u": float(np.sqrt((du * du).mean())),
                             "rms_v": float(np.sqrt((dv * dv).mean())), "max_abs": float(max(np.abs(du).max(), np.abs(dv).max()))}
            # the like-for-like CPU comparator of `value`: the same red-black order on every host core
            omp_rate, omp_threads = cpu_red_black_all_cores(U0, V0, coef_full, max(20, args.cpu_calls))
            out["cpu_baseline"]["red_black_all_cores"] = {"value": round(omp_rate, 2), "unit": "iterations/s", "cores": omp_threads,
                                                          "kind": "port", "sample": "%d calls x iter=%d, oracle red-black order, OpenMP over "
                                                          "the columns of a colour pass, pages first-touched by the sweeping threads; not in the reference (its "
                                                          "flow solvers are single-threaded)" % (max(20, args.cpu_calls), ITER)}
            # ---- how far the timed ordering is from the reference's result ---------------------------------
            Ur, Vr = U0.clone(), V0.clone()
            dev.oflow_sor_elin4(Ur, Vr, *coef_full, ITER, OMEGA, capi.MODE_RED_BLACK)
            torch.cuda.synchronize()
            du = Ur.cpu().numpy().astype(np.float64) - first[0]
            dv = Vr.cpu().numpy().astype(np.float64) - first[1]
            prb = {"mode": "red_black (the timed ordering) vs cpu oracle in the reference's lexicographic order, after the first iter=%d call" % ITER,
                   "rms_u": float(np.sqrt((du * du).mean())), "rms_v": float(np.sqrt((dv * dv).mean())),
                   "max_abs": float(max(np.abs(du).max(), np.abs(dv).max()))}
            # sweeps until the two orderings agree to 1e-4 RMS on this frame (exact order on the GPU is the oracle bit for bit)
            Ux, Vx = U0.clone(), V0.clone()
            Ur, Vr = U0.clone(), V0.clone()
            done, trace = 0, {}
            for target in (4, 8, 16, 32, 64, 128, 256, 512, 1024):
                dev.oflow_sor_elin4(Ux, Vx, *coef_full, target - done, OMEGA, capi.MODE_EXACT_ORDER)
                dev.oflow_sor_elin4(Ur, Vr, *coef_full, target - done, OMEGA, capi.MODE_RED_BLACK)
                done = target
                r = float(torch.sqrt(((Ux.double() - Ur.double()) ** 2).mean()).item())
                trace[str(target)] = r
                if r < 1e-4:
                    break
            prb["rms_u_by_sweeps"] = trace
            prb["sweeps_to_1e-4_rms"] = done if trace[str(done)] < 1e-4 else None
            capi.call("pdeip_persist_error")
            out["parity_rb"] = prb
            # ---- the host-pointer call a MEX stub makes: H2D of 13 planes, solve, D2H of 2 (PCIe-inclusive) -------
            hc = {}
            for name, mode in (("exact_order", capi.MODE_EXACT_ORDER), ("red_black", capi.MODE_RED_BLACK)):
                ms_call, ou, ov = host_call(capi, U0, V0, coef_full, mode)
                hc[name + "_ms"] = round(ms_call, 3)
                hc[name + "_iterations_per_s"] = round(ITER / (ms_call * 1e-3), 1)
                if mode == capi.MODE_EXACT_ORDER:
                    hc["exact_order_max_abs_vs_cpu"] = float(max(np.abs(ou.astype(np.float64) - first[0]).max(), np.abs(ov.astype(np.float64) - first[1]).max()))
            hc["bytes_over_pcie"] = (13 + 2) * 4 * N
            # the link as this process sees it: one pageable 33 MB plane up, timed alone (what the entry point's own copies do)
            probe_h, probe_d = U0.cpu(), torch.empty_like(U0)
            for _ in range(3):
                probe_d.copy_(probe_h)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                probe_d.copy_(probe_h)
            torch.cuda.synchronize()
            up_s = (time.perf_counter() - t0) / 10
            hc["h2d_gbs_one_plane"] = round(4 * N / up_s / 1e9, 1)
            hc["link_bound_ms"] = {"13_planes_up": round(13 * up_s * 1e3, 3), "13_up_plus_2_down_serial": round(15 * up_s * 1e3, 3)}
            hc["overlap"] = "red-black call cut into 6 column slabs on two worker threads (upload | sweeps | download of different slabs), PDEIP_HOST_OVERLAP"
            del probe_h, probe_d
            hc["workload"] = "pdeip_oflow_sor_elin4, host pointers, 2160x3840, iter=4, median of 7 calls"
            out["host_call"] = hc
            del Ur, Vr, Ux, Vx
            # ---- solver 2 (alternating line relaxation, the MATLAB drivers' default), same frame ----------
            # one ALR iteration = column lines of U,V then row lines of V,U.  exact = reference line order
            # (bit-identical, serial by construction), zebra = even/odd lines concurrently.
            lib = sys.modules["oracle_lib"].lib()
            hu, hv = U0.cpu().numpy().copy(), V0.cpu().numpy().copy()
            hc = [t.cpu().numpy() for t in coef_full]
            t0 = time.perf_counter()
            lib.orc_oflow_alr_elin4(hu.ctypes.data, hv.ctypes.data, *[a.ctypes.data for a in hc], NROWS, NCOLS, 1, ctypes.c_float(1.5), 0)
            alr_cpu = time.perf_counter() - t0
            alr = {"unit": "iterations/s", "cpu_oracle": round(1.0 / alr_cpu, 3), "omega": 1.5}
            for name, mode, reps in (("exact_order", capi.MODE_EXACT_ORDER, 1), ("zebra", capi.MODE_RED_BLACK, 20)):
                Ua, Va = U0.clone(), V0.clone()
                dev.oflow_alr_elin4(Ua, Va, *coef_full, 1, 1.5, mode)
                torch.cuda.synchronize()
                if mode == capi.MODE_EXACT_ORDER:
                    alr["exact_order_max_abs_vs_cpu"] = float(max(np.abs(Ua.cpu().numpy().astype(np.float64) - hu).max(),
                                                                   np.abs(Va.cpu().numpy().astype(np.float64) - hv).max()))
                t0 = time.perf_counter()
                dev.oflow_alr_elin4(Ua, Va, *coef_full, reps, 1.5, mode)
                torch.cuda.synchronize()
                alr[name] = round(reps / (time.perf_counter() - t0), 3)
            out["line_relaxation"] = alr
            # ---- one late-linearisation pyramid level resident in HBM (BASELINE config C2: 1080x1920, 3 channels) --
            # firstLoop body = warp, derivatives, 4 x [robust assembly, diffusion weights, Oflow_sor_llin4_2d iter=4], median
            fl = importlib.import_module("pde-based-image-processing_amd.flow_level")
            spec = importlib.util.spec_from_file_location("matlab_side", os.path.join(ROOT, "oracle", "matlab_side.py"))
            ms = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(ms)
            rng = np.random.default_rng(2)
            jj, ii = np.meshgrid(np.arange(1920), np.arange(1080))
            tex = lambda di, dj, c: (np.sin(0.021 * (ii + di) + c) * np.cos(0.017 * (jj + dj) - c) + 0.3 * np.sin(0.11 * (ii + di) + 0.07 * (jj + dj))).astype(np.float32)
            I0 = np.asfortranarray(np.stack([tex(0, 0, c) for c in range(3)], axis=2))
            I1 = np.asfortranarray(np.stack([tex(0.7, -0.4, c) for c in range(3)], axis=2))
            Z = np.zeros((1080, 1920), dtype=np.float32, order="F")
            # runme.m:44 runs the driver with ('grad', 'gradmag'): first term = rgb2grad (6 planes), second = gradient magnitude (3 planes)
            lp = dict(firstLoop=1, secondLoop=4, iter=4, omega=1.9, solver=1, alpha=0.042, b1=1.4843, b2=0.2915, sndTerm="gradmag", order=0)
            dI0, dI1, dZ = dev.to_device(I0), dev.to_device(I1), dev.to_device(Z)
            dG0, dG1 = dev.rgb2grad(dI0), dev.rgb2grad(dI1)
            level = {}
            for name, prm, mode in (("exact_order", lp, capi.MODE_EXACT_ORDER), ("red_black", lp, capi.MODE_RED_BLACK),
                                    ("zebra_alr", dict(lp, solver=2, omega=1.5), capi.MODE_RED_BLACK)):
                lv = fl.FlowLlinLevel(prm, mode=mode)
                gU, gV = lv.run(dG0, dG1, dZ, dZ, dI0, dI1)
                torch.cuda.synchronize()
                laps = []  # median of 21 runs: one run in ~20 catches an allocator stall of tens of ms (seen twice in a mean of three)
                for _ in range(21):
                    t0 = time.perf_counter()
                    gU, gV = lv.run(dG0, dG1, dZ, dZ, dI0, dI1)
                    torch.cuda.synchronize()
                    laps.append(time.perf_counter() - t0)
                level[name + "_ms"] = round(sorted(laps)[len(laps) // 2] * 1e3, 3)
                if name == "exact_order":
                    t0 = time.perf_counter()
                    wU, wV = ms.flow_level(sys.modules["oracle_lib"], ms.rgb2grad(I0), ms.rgb2grad(I1), Z, Z, lp, I2t0=I0, I2t1=I1)
                    level["cpu_statement_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
                    level["exact_order_max_abs_vs_cpu"] = float(max(np.abs(dev.to_matlab(gU).astype(np.float64) - wU).max(),
                                                                    np.abs(dev.to_matlab(gV).astype(np.float64) - wV).max()))
            level["workload"] = ("FlowEminND_llin_2D_v10 firstLoop body as runme.m configures it ('grad', 'gradmag'): 1080x1920, 6 + 3 planes, "
                                 "secondLoop=4, iter=4; solver 1 (exact / red-black) and solver 2 zebra")
            del dG0, dG1
            out["flow_level"] = level
            # ---- BASELINE config C5: one level of the disparity drivers at 1988x2880x3, as runme.m:20/28 configures them ----
            jj, ii = np.meshgrid(np.arange(2880), np.arange(1988))
            tex5 = lambda dj, c: (np.sin(0.021 * ii + c) * np.cos(0.017 * (jj + dj) - c) + 0.3 * np.sin(0.11 * ii + 0.07 * (jj + dj))).astype(np.float32)
            dL = dev.to_device(np.stack([tex5(0, c) for c in range(3)], axis=2))
            dR = dev.to_device(np.stack([tex5(1.3, c) for c in range(3)], axis=2))
            gL, gR = dev.rgb2grad(dL), dev.rgb2grad(dR)
            dZ5 = torch.zeros((2880, 1988), device=device)
            dp = dict(firstLoop=1, secondLoop=4, iter=4, omega=1.9, alpha=0.15, b1=0.25, b2=0.72, beta=0.4, sndTerm="gradmag")
            dl = {"workload": "firstLoop body at 1988x2880x3, secondLoop=4, iter=4: DispEminND_llin_2D ('grad','gradmag': 6 + 3 planes) and "
                              "DispEminND_llin_sym_2D (both views)", "unit": "ms"}
            for name, prm, mode in (("red_black", dict(dp, solver=1, omega=1.5), capi.MODE_RED_BLACK), ("zebra_alr", dict(dp, solver=2, omega=1.5), capi.MODE_RED_BLACK)):
                for tag, run in (("disparity_", lambda lv=fl.DispLlinLevel(prm, mode=mode): lv.run(gL, gR, dZ5, dL, dR)),
                                 ("symmetric_", lambda lv=fl.DispSymLevel(prm, mode=mode): lv.run(dL, dR, dZ5, dZ5, 2.0))):
                    run(); torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(20):
                        run()
                    torch.cuda.synchronize()
                    dl[tag + name] = round((time.perf_counter() - t0) / 20 * 1e3, 3)
            out["disparity_level"] = dl
            del dL, dR, gL, gR, dZ5
            # ---- whole drivers (pyramid + every level resident; host numpy frames in, flow out), as runme.m calls them --------
            drivers = importlib.import_module("pde-based-image-processing_amd.drivers")
            Iseq = np.concatenate([(I0 + 1.3) * 98.0, (I1 + 1.3) * 98.0], axis=2).astype(np.float32)   # 0..255, 1080x1920x(3+3)
            dr = {"workload": "FlowEminND_llin_2D_v10(Iseq 1080x1920x3 pair, 'grad', 'gradmag'), 16 scales x firstLoop 4 x secondLoop 4 x iter 4; "
                              "incl. H2D of the frames and D2H of the flow", "unit": "ms"}
            for name, kw in (("red_black_sor", dict(mode=capi.MODE_RED_BLACK, solver=1, omega=1.5)), ("zebra_alr", dict(mode=capi.MODE_RED_BLACK, omega=1.5))):
                drivers.FlowEminND_llin_2D_v10(Iseq, 3, "grad", "gradmag", **kw)
                torch.cuda.synchronize()
                nrep = 20 if name == "red_black_sor" else 5   # the zebra run is 0.2 s a piece
                t0 = time.perf_counter()
                for _ in range(nrep):
                    drivers.FlowEminND_llin_2D_v10(Iseq, 3, "grad", "gradmag", **kw)
                torch.cuda.synchronize()
                dr[name] = round((time.perf_counter() - t0) / nrep * 1e3, 1)
                # graph=True: the resident part replayed from a HIP graph captured on the first call for this frame size
                ref = drivers.FlowEminND_llin_2D_v10(Iseq, 3, "grad", "gradmag", **kw)
                drivers.FlowEminND_llin_2D_v10(Iseq, 3, "grad", "gradmag", graph=True, **kw)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(nrep):
                    got = drivers.FlowEminND_llin_2D_v10(Iseq, 3, "grad", "gradmag", graph=True, **kw)
                torch.cuda.synchronize()
                dr[name + "_graph"] = round((time.perf_counter() - t0) / nrep * 1e3, 1)
                dr[name + "_graph_same_bits"] = bool(all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref, got)))
            # the same driver as ONE C-ABI call (pdeip_flow_nd_llin, the level loop in C++: what the MEX stub of a MATLAB session calls)
            kw = dict(mode=capi.MODE_RED_BLACK, solver=1, omega=1.5)
            ref = drivers.FlowEminND_llin_2D_v10(Iseq, 3, "grad", "gradmag", **kw)
            got = drivers.capi_FlowEminND_llin_2D_v10(Iseq, 3, "grad", "gradmag", **kw)
            t0 = time.perf_counter()
            for _ in range(20):
                drivers.capi_FlowEminND_llin_2D_v10(Iseq, 3, "grad", "gradmag", **kw)
            dr["red_black_sor_cxx"] = round((time.perf_counter() - t0) / 20 * 1e3, 1)
            dr["red_black_sor_cxx_same_bits"] = bool(all(np.array_equal(a, b, equal_nan=True) for a, b in zip(ref, got)))
            out["driver_nd_1080p"] = dr
            del Iseq
            # ---- the lagged-diffusivity loop of the TV denoiser resident in HBM (BASELINE config C3: 2160x3840 gray) ----
            tvp = dict(alpha=500.0, omega=1.75, outer_iter=20, inner_iter=4, solver=1)
            gI = torch.empty((NCOLS, NROWS), device=device, dtype=torch.float32).uniform_(0, 1)
            tv = {"workload": "TVdenoise8 loop, 2160x3840, outer_iter=20 (21 x [ADdiffWeights+quantile, PsiData/TRACE/B, PDEsolver8 inner_iter=4])"}
            for name, prm, mode in (("exact_order", tvp, capi.MODE_EXACT_ORDER), ("red_black", tvp, capi.MODE_RED_BLACK),
                                    ("zebra_alr", dict(tvp, solver=2), capi.MODE_RED_BLACK)):   # solver 2 is TVdenoise8's default
                lv = fl.TvLevel(prm, mode=mode)
                lv.run(gI, gI)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    lv.run(gI, gI)
                torch.cuda.synchronize()
                tv[name + "_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 2)
            out["tv_level"] = tv
            del gI
            # ---- BASELINE config C4 with its outer structure: the FAS full-multigrid driver at 2160x3840, resident ------
            # pyramid + per-scale constants, then per scale one V-cycle (smoother = firstLoop 4 x [weights, Oflow_sor_elin4_2d
            # iter 4], residual, restriction, Oflow_lhs_elin4_2d, prolongation); CPU statement on a bounded 540x960 frame.
            fas = importlib.import_module("pde-based-image-processing_amd.fas")
            jj, ii = np.meshgrid(np.arange(NCOLS), np.arange(NROWS))
            big = lambda di, dj: ((np.sin(0.021 * (ii + di)) * np.cos(0.017 * (jj + dj)) + 0.3 * np.sin(0.11 * (ii + di) + 0.07 * (jj + dj)) + 1.5) * 80).astype(np.float32)
            F0, F1 = np.asfortranarray(big(0, 0)[:, :, None]), np.asfortranarray(big(0.7, -0.4)[:, :, None])
            d0, d1 = dev.to_device(F0), dev.to_device(F1)
            fmg = {"workload": "FlowEminNDFASFMG_elin_2D_v10 whole driver, 2160x3840x1, 9 scales, V-cycle, firstLoop=4, iter=4", "unit": "ms"}
            for name, prm, mode in (("red_black_sor", dict(solver=1, omega=1.0), capi.MODE_RED_BLACK),
                                    ("zebra_alr", dict(solver=2, omega=1.5), capi.MODE_RED_BLACK),
                                    ("exact_order_sor", dict(solver=1, omega=1.0), capi.MODE_EXACT_ORDER)):
                drv = fas.FasFmgFlow(prm, mode=mode)
                drv.run(d0, d1)
                torch.cuda.synchronize()
                nrep = 20 if name == "red_black_sor" else 5   # zebra: 0.16 s a run, exact order 57 ms
                t0 = time.perf_counter()
                for _ in range(nrep):
                    drv.run(d0, d1)
                torch.cuda.synchronize()
                fmg[name] = round((time.perf_counter() - t0) / nrep * 1e3, 2)
                if True:
                    # the same launches replayed from a captured HIP graph (graphs.py): bit-identical, no per-launch host work;
                    # the eager figure above is bound by the host enqueueing ~1 700 launches
                    ref = [t.clone() for t in drv.run(d0, d1)]
                    drv.run_graph(d0, d1)
                    torch.cuda.synchronize()
                    reps = nrep
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        got = drv.run_graph(d0, d1)
                    torch.cuda.synchronize()
                    fmg[name + "_graph"] = round((time.perf_counter() - t0) / reps * 1e3, 2)
                    fmg[name + "_graph_same_bits"] = bool(all(torch.equal(a, b) for a, b in zip(ref, got)))
            # the same driver as ONE C-ABI call from host frames (pdeip_flow_fas_fmg_elin: 66 MB up, 66 MB down included)
            Ifmg = np.asfortranarray(np.concatenate([F0, F1], axis=2))   # MATLAB's layout: no host-side copy inside the timed call
            kwf = dict(mode=capi.MODE_RED_BLACK, solver=1, omega=1.0)
            drivers.capi_FlowEminNDFASFMG_elin_2D_v10(Ifmg, 1, **kwf)
            t0 = time.perf_counter()
            for _ in range(10):
                drivers.capi_FlowEminNDFASFMG_elin_2D_v10(Ifmg, 1, **kwf)
            fmg["red_black_sor_cxx_host_frames"] = round((time.perf_counter() - t0) / 10 * 1e3, 2)
            del Ifmg
            s0, s1 = np.asfortranarray(F0[::4, ::4]), np.asfortranarray(F1[::4, ::4])
            sp = dict(fas.DEFAULTS, solver=1, omega=1.0, order=0)
            t0 = time.perf_counter()
            wU, wV = ms.fas_fmg(sys.modules["oracle_lib"], s0, s1, sp)
            fmg["cpu_statement_540x960"] = round((time.perf_counter() - t0) * 1e3, 1)
            drv = fas.FasFmgFlow(dict(solver=1, omega=1.0), mode=capi.MODE_EXACT_ORDER)
            gU, gV = drv.run(dev.to_device(s0), dev.to_device(s1))
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            gU, gV = drv.run(dev.to_device(s0), dev.to_device(s1))
            torch.cuda.synchronize()
            fmg["exact_order_sor_540x960"] = round((time.perf_counter() - t0) * 1e3, 2)
            fmg["exact_order_max_abs_vs_cpu_540x960"] = float(max(np.abs(dev.to_matlab(gU).astype(np.float64) - wU).max(),
                                                                  np.abs(dev.to_matlab(gV).astype(np.float64) - wV).max()))
            out["fmg"] = fmg
            del d0, d1, drv
            # ---- the solver call of each BASELINE config at its own frame size (sweeps/s, both orderings) ----------
            gen = torch.Generator(device=device).manual_seed(7)

            def planes(nr, nc, k, lo=0.5, hi=5.0):
                return [torch.empty((nc, nr), device=device, dtype=torch.float32).uniform_(lo, hi, generator=gen) for _ in range(k)]

            def rate(fn, it, reps=20):
                fn(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                return round(it * reps / (time.perf_counter() - t0), 1)

            cfgs = {}
            for mode, tag in ((capi.MODE_RED_BLACK, "red_black"), (capi.MODE_EXACT_ORDER, "exact_order")):
                nr, nc = 388, 584      # C1 Horn-Schunck, iter = 20
                a, b, c = planes(nr, nc, 3, -0.5, 0.5); w = planes(nr, nc, 4); Uc, Vc = planes(nr, nc, 2, -1, 1)
                dat = [a * b, -a * c, -b * c, a * a, b * b]  # M, Cu, Cv, Du, Dv: built once, not inside the timed call
                cfgs.setdefault("C1_elin4_388x584_iter20", {})[tag] = rate(lambda: dev.oflow_sor_elin4(Uc, Vc, *dat, *w, 20, OMEGA, mode), 20)
                nr, nc = 1080, 1920    # C2 late linearisation, iter = 4
                a, b, c = planes(nr, nc, 3, -0.5, 0.5); w = planes(nr, nc, 4); Uc, Vc = planes(nr, nc, 2, -1, 1); dUc, dVc = planes(nr, nc, 2, -0.1, 0.1)
                dat = [a * b, -a * c, -b * c, a * a, b * b]
                cfgs.setdefault("C2_llin4_1080x1920_iter4", {})[tag] = rate(lambda: dev.oflow_sor_llin4(Uc, Vc, dUc, dVc, *dat, *w, 4, OMEGA, mode), 4)
                nr, nc = 2160, 3840    # C3 TV denoising, 8 neighbours, inner_iter = 4
                w8 = planes(nr, nc, 8); Xc, Bc = planes(nr, nc, 2, 0, 1); TR = 1 + sum(w8)
                cfgs.setdefault("C3_pde8_2160x3840_iter4", {})[tag] = rate(lambda: dev.pde_sor8(Xc, TR, Bc, *w8, 4, 1.75, mode), 4)
                del w8, TR
                nr, nc = 1988, 2880    # C5 disparity, iter = 4
                w = planes(nr, nc, 4); Ud, = planes(nr, nc, 1, -3, 3); dUd, Cud = planes(nr, nc, 2, -0.5, 0.5); Dud, = planes(nr, nc, 1, 0.05, 2)
                cfgs.setdefault("C5_disp4_1988x2880_iter4", {})[tag] = rate(lambda: dev.disp_sor_llin4(Ud, dUd, Cud, Dud, *w, 4, OMEGA, mode), 4)
            # red-black roofline per config: planes a launch must move (coefficients + read-only + iterate in + out) x sweeps fused per launch
            for key, (npx, planes, it, fuse) in {"C1_elin4_388x584_iter20": (388 * 584, 13, 20, 4), "C2_llin4_1080x1920_iter4": (1080 * 1920, 15, 4, 4),
                                                 "C3_pde8_2160x3840_iter4": (2160 * 3840, 12, 4, 2), "C5_disp4_1988x2880_iter4": (1988 * 2880, 9, 4, 4)}.items():
                per_call_s = it / cfgs[key]["red_black"]
                launches = -(-it // fuse)
                fused_min = launches * planes * npx * 4
                cfgs[key]["roofline_red_black"] = {"bound": "hbm", "us_per_call": round(per_call_s * 1e6, 1), "launches_per_call": launches,
                                                   "bytes_per_call": fused_min, "frac": round(fused_min / per_call_s / 1e9 / HBM_PEAK_GBS, 4),
                                                   "effective_frac": round(it * planes * npx * 4 / per_call_s / 1e9 / HBM_PEAK_GBS, 4)}
            cfgs["roofline_note"] = ("frac = launches x planes x 4 B x pixels (what the fused launches of a call must move) / call time / 8 TB/s; effective_frac = the "
                                     "per-sweep figure (planes x 4 B x pixels x iter); C1 is a 0.9 MB-per-plane frame relaxed out of LDS: a latency chain, not a stream")
            cfgs["unit"] = "iterations/s"
            cfgs["note"] = "C4 (elin4 2160x3840 iter=4) is the headline workload above"
            out["configs"] = cfgs
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
