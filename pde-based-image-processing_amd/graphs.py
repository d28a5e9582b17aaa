"""HIP-graph replay of a resident driver run.

A driver run at a fixed frame size is a fixed sequence of a few thousand small launches (pyramid, per-scale terms, weights,
solver calls); below ~270x480 every one of them is shorter than what the host needs to enqueue it, so the run is bound by
the host.  Captured once (torch.cuda.graph around the same Python code: the library only enqueues on the stream it is
given), the sequence replays from the device's command processor with no host work in between.

The library keeps scratch buffers whose addresses end up in the captured kernel arguments; `pdeip_workspace_generation()`
changes when one of them is freed or regrown, and the graph is captured again then.  The first call runs eagerly (it
allocates those buffers and opts kernels into their LDS sizes, neither of which may happen during capture).
"""
import torch

from . import capi


class GraphedRun:
    """fn(*tensors) -> tensor or tuple of tensors, all device-resident, shapes fixed.  Call with new input VALUES (same shapes):
    they are copied into the captured input buffers, the graph replays, and the captured output tensors are returned (valid until
    the next call)."""

    def __init__(self, fn):
        self.fn = fn
        self.graph = None
        self.static_in = None
        self.static_out = None
        self.generation = None
        self.failed = False

    def _capture(self, inputs):
        self.static_in = [t.clone() for t in inputs]
        self.fn(*self.static_in)  # warm-up: workspace, LDS opt-ins, schedule tables
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            out = self.fn(*self.static_in)
        torch.cuda.current_stream().wait_stream(side)
        self.graph, self.static_out = g, out
        self.generation = capi.load().pdeip_workspace_generation()

    def __call__(self, *inputs):
        if self.failed:
            return self.fn(*inputs)
        stale = self.graph is None or self.generation != capi.load().pdeip_workspace_generation() or \
            any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(self.static_in, inputs))
        if stale:
            self.graph = None
            try:
                self._capture(inputs)
            except (RuntimeError, capi.PdeipError):
                # something in fn cannot be captured (a host read-back, a schedule upload): run it eagerly from now on
                self.failed = True
                torch.cuda.synchronize()
                return self.fn(*inputs)
        for dst, src in zip(self.static_in, inputs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.graph.replay()
        return self.static_out
