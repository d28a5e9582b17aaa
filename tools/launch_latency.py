"""Per-kernel cost of a chain of tiny dependent launches of the library, on torch's default stream and on a side stream."""
import sys, time, importlib
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); import torch
dev = importlib.import_module("pde-based-image-processing_amd.device")
A = torch.zeros((15, 9), device="cuda"); B = torch.ones_like(A); C = torch.empty_like(A)
N = 2000


def chain():
    for _ in range(N // 2):
        dev.add(A, B, C)
        dev.add(C, B, A)


for name, stream in (("default stream", None), ("side stream", torch.cuda.Stream())):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        chain(); torch.cuda.synchronize()
        t0 = time.perf_counter(); chain(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("%-15s enqueue %.2f us/launch, end-to-end %.2f us/launch" % (name, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6), flush=True)
        g = torch.cuda.CUDAGraph()
        s2 = torch.cuda.Stream(); s2.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=s2):
            chain()
        g.replay(); torch.cuda.synchronize()
        t0 = time.perf_counter(); g.replay(); torch.cuda.synchronize()
        print("%-15s graph replay %.2f us/kernel" % (name, (time.perf_counter() - t0) / N * 1e6), flush=True)
