"""Exact-order elin4, one sweep, at a cache-resident and a DRAM-resident frame size, for rocprofv3 --pmc passes (address translation and
read latency of k_sor_exact_persist)."""
import importlib, sys
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
dev = importlib.import_module("pde-based-image-processing_amd.device")
for nr, nc in ((2160, 2050), (2160, 3840)):
    g = torch.Generator(device="cuda").manual_seed(1)
    P = lambda lo, hi: torch.empty((nc, nr), device="cuda").uniform_(lo, hi, generator=g)
    U, V = P(-1, 1), P(-1, 1)
    coef = [P(-0.5, 0.5) for _ in range(3)] + [P(0.1, 1), P(0.1, 1)] + [P(0.5, 5) for _ in range(4)]
    for _ in range(4):
        dev.oflow_sor_elin4(U, V, *coef, 1, 1.0, 0)
    torch.cuda.synchronize()
