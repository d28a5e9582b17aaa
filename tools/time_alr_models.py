"""Zebra line relaxation per model and size (GPU only; the CPU oracle's times are in DESIGN.md 5.5)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from pdeip_amd import device as dev
import problems as pb


def run(name, fn, p, skip, reps, per):
    d = {k: dev.to_device(v) for k, v in p.items()}
    args = list(d.values())
    fn(*args, reps, 1.5, 1); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        fn(*args, reps, 1.5, 1)
    torch.cuda.synchronize()
    print("%-28s %8.3f ms per %s" % (name, (time.perf_counter() - t0) / 3 / (reps if per == "iteration" else 1) * 1e3, per), flush=True)


for nr, nc in ((2160, 3840), (1080, 1920), (388, 584)):
    run("elin4 %dx%d" % (nr, nc), dev.oflow_alr_elin4, pb.elin4(7, nr, nc), 2, 8, "iteration")
run("llin4 1080x1920", dev.oflow_alr_llin4, pb.llin4(7, 1080, 1920), 4, 8, "iteration")
run("llin8 1080x1920", dev.oflow_alr_llin8, pb.llin8(7, 1080, 1920), 4, 8, "iteration")
run("disp4 1988x2880", dev.disp_alr_llin4, pb.disp4(7, 1988, 2880), 2, 8, "iteration")
run("pde4 2160x3840", dev.pde_alr4, pb.pde4(7, 2160, 3840), 1, 8, "iteration")
run("pde8 2160x3840 (one call)", dev.pde_alr8, pb.pde8(7, 2160, 3840), 1, 1, "call")
