"""One frame across several GPUs: column slabs + wide-halo exchange (RED_BLACK ordering).

Why column slabs.  The buffers are MATLAB column-major, so a block of consecutive MATLAB columns
is one contiguous byte range of every plane: a slab is a zero-copy slice, and a halo of H
columns is one contiguous message of H*nrows floats per field ("row tiles with a one-row halo"
in BASELINE.json's wording, seen from the transposed image the kernels work on).

Why a wide halo.  A red-black sweep moves information by at most two columns, so a slab that
carries H = 2k halo columns on each cut side can run k sweeps with no communication and still
have bit-exact values in the columns it owns; the halo's outer columns go stale and are simply
refreshed by the next exchange.  One exchange per k sweeps (k = iter = 4, or a multiple: the budget
carries across solver calls) replaces the 2*iter per-colour exchanges a one-column halo would need:
the messages (H*nrows*4 B per field, 69 KB at 2160 rows and H = 8) stay latency-bound either way, so
fewer is better on point-to-point xGMI; the price is 2H extra columns of work per interior slab.  The exact-order (lexicographic) mode does not decompose this way -- its
dependency front crosses the whole frame -- so it is single-GPU (replicas) only.

Exchange = torch.distributed batched isend/irecv (backend "nccl" is RCCL on ROCm; "gloo" in the
CPU tests).  There is no reduction collective on the path: iteration counts are fixed.

The local relaxation is injected (`sweep_fn`) so the decomposition logic can be tested on CPU
with world_size 2 against the single-domain oracle; the product default is the HIP library.
"""
import torch
import torch.distributed as dist


def split_columns(ncols, world):
    """Contiguous, near-equal column ranges [c0, c1) per rank."""
    base, rem = divmod(ncols, world)
    bounds, c = [], 0
    for r in range(world):
        w = base + (1 if r < rem else 0)
        bounds.append((c, c + w))
        c += w
    return bounds


class SlabDomain:
    """Geometry of this rank's slab of a [ncols, nrows] frame and its halo exchange."""

    def __init__(self, ncols, nrows, rank, world, halo, group=None):
        if world > 1 and min(c1 - c0 for c0, c1 in split_columns(ncols, world)) < halo:
            raise ValueError("slabs of %d columns are narrower than the halo (%d)" % (ncols // world, halo))
        self.ncols, self.nrows, self.rank, self.world, self.halo, self.group = ncols, nrows, rank, world, halo, group
        self.c0, self.c1 = split_columns(ncols, world)[rank]
        self.lo = max(0, self.c0 - halo)          # first global column held locally
        self.hi = min(ncols, self.c1 + halo)      # one past the last
        self.left = rank - 1 if rank > 0 else None
        self.right = rank + 1 if rank < world - 1 else None
        self._ops_cache = {}  # exchange(): P2P op lists per set of planes

    @property
    def col0(self):
        """Global index of local column 0 (colour parity for the kernels)."""
        return self.lo

    @property
    def ncols_local(self):
        return self.hi - self.lo

    def slice_local(self, plane):
        """Global plane [(F,) ncols, nrows] -> this rank's slab incl. halo (a copy: it becomes resident)."""
        return plane[..., self.lo:self.hi, :].clone(memory_format=torch.contiguous_format)

    def owned(self, local):
        """View of the columns this rank owns inside a local plane."""
        return local[..., self.c0 - self.lo:self.c1 - self.lo, :]

    def exchange(self, fields):
        """Refresh the halo columns of every local 2-D plane in `fields` from the neighbours' owned columns."""
        if self.world == 1:
            return
        H, staged = self.halo, []
        # gloo moves host memory only: a rehearsal of the multi-rank path on CUDA tensors over gloo stages
        # the halo columns through the host (never the case with the nccl/RCCL backend)
        via_host = fields[0].is_cuda and dist.get_backend(self.group) == "gloo"
        # the op list only depends on which buffers are exchanged: built once per set of planes (a solver
        # call per step would otherwise spend more host time here than the sweeps take on the GPU)
        key = tuple(t.data_ptr() for t in fields)
        ops = None if via_host else self._ops_cache.get(key)
        if ops is None:
            ops = []
            own0, own1 = self.c0 - self.lo, self.c1 - self.lo
            for t in fields:
                pairs = []
                if self.left is not None:
                    pairs.append((t[own0:own0 + H], t[own0 - H:own0], self.left))
                if self.right is not None:
                    pairs.append((t[own1 - H:own1], t[own1:own1 + H], self.right))
                for src, dst, peer in pairs:
                    if via_host:
                        hsrc, hdst = src.cpu(), torch.empty(dst.shape, dtype=dst.dtype)
                        staged.append((dst, hdst))
                        src, dst = hsrc, hdst
                    ops.append(dist.P2POp(dist.isend, src, peer, self.group))
                    ops.append(dist.P2POp(dist.irecv, dst, peer, self.group))
            if not via_host:
                if len(self._ops_cache) > 16:
                    self._ops_cache.clear()
                self._ops_cache[key] = ops
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for dst, hdst in staged:
            dst.copy_(hdst)

    def gather_owned(self, local, dst=0):
        """Rank `dst` gets the assembled [ncols, nrows] plane (others get None)."""
        mine = self.owned(local).contiguous()
        if self.world == 1:
            return mine
        # gloo moves host memory only (rehearsal of the multi-rank path on CUDA tensors): stage through the host there
        via_host = mine.is_cuda and dist.get_backend(self.group) == "gloo"
        wire = mine.cpu() if via_host else mine
        if self.rank == dst:
            parts = [torch.empty((c1 - c0, self.nrows), dtype=wire.dtype, device=wire.device)
                     for c0, c1 in split_columns(self.ncols, self.world)]
            parts[dst] = wire
            reqs = [dist.irecv(parts[r], r, self.group) for r in range(self.world) if r != dst]
            for q in reqs:
                q.wait()
            out = torch.cat(parts, dim=0)
            return out.to(mine.device) if via_host else out
        dist.send(wire, dst, self.group)
        return None


def _hip_elin4(it, coef, n_sweeps, omega, col0, out=None):
    from . import capi, device
    device.oflow_sor_elin4(it[0], it[1], *coef, n_sweeps, omega, mode=capi.MODE_RED_BLACK, col0=col0, out=out)


_hip_elin4.out_of_place = True  # relaxes `it` into `out` when given: no device-to-device copy after an odd number of launches


def _hip_llin4(it, coef, n_sweeps, omega, col0):
    from . import capi, device
    device.oflow_sor_llin4(coef[0], coef[1], it[0], it[1], *coef[2:], n_sweeps, omega, mode=capi.MODE_RED_BLACK,
                           col0=col0)


def _hip_disp4(it, coef, n_sweeps, omega, col0):
    from . import capi, device
    device.disp_sor_llin4(coef[0], it[0], *coef[1:], n_sweeps, omega, mode=capi.MODE_RED_BLACK, col0=col0)


def _hip_pde4(it, coef, n_sweeps, omega, col0):
    from . import capi, device
    device.pde_sor4(it[0], *coef, n_sweeps, omega, mode=capi.MODE_RED_BLACK, col0=col0)


def _hip_pde8(it, coef, n_sweeps, omega, col0):
    from . import capi, device
    device.pde_sor8(it[0], *coef, n_sweeps, omega, mode=capi.MODE_RED_BLACK, col0=col0)


HIP_SWEEPS = {"elin4": _hip_elin4, "llin4": _hip_llin4, "disp4": _hip_disp4, "pde4": _hip_pde4, "pde8": _hip_pde8}


class SlabSolver:
    """`iter` red-black SOR sweeps of one point solver on this rank's slab, and the stencil stages around them.

    iterate : list of local planes relaxed in place (U,V / dU,dV / dU / X), each [ncols_local, nrows]
    coef    : list of local read-only planes in the order the kernel takes them
              elin4: M,Cu,Cv,Du,Dv,wW,wN,wE,wS   llin4: U,V,M,Cu,Cv,Du,Dv,wW,wN,wE,wS
              disp4: U,Cu,Du,wW,wN,wE,wS          pde4 : TRACE,B,wW,wN,wE,wS
              pde8 : TRACE,B,wW,wNW,wN,wNE,wE,wSE,wS,wSW   (four-colour 9-point sweep: two columns per sweep as well)
    sweeps_per_exchange : k; the domain's halo must be >= 2k.

    The halo is a budget of columns: after an exchange all `halo` columns next to a cut are exact; a sweep lets two of them go
    stale, a stencil stage computed locally from the iterate (diffusion weights: radius 1; `stage`) as many as its radius, and
    planes derived that way need no exchange of their own.  The budget carries across calls; an exchange happens when the next
    step would overdraw it."""

    def __init__(self, domain, kind="elin4", sweeps_per_exchange=4, sweep_fn=None):
        if domain.world > 1 and domain.halo < 2 * sweeps_per_exchange:
            raise ValueError("halo %d < 2 * sweeps_per_exchange (%d)" % (domain.halo, sweeps_per_exchange))
        self.dom, self.k = domain, sweeps_per_exchange
        self.sweep_fn = sweep_fn if sweep_fn is not None else HIP_SWEEPS[kind]
        self.spent = None  # halo columns gone stale since the last refresh; None: unknown, exchange first
        self._alt = None   # second plane set of the out-of-place form (solve_pingpong)

    def invalidate(self):
        """The iterate planes were changed from outside (or are different tensors): refresh the halo before the next step."""
        self.spent = None

    def _room(self, iterate):
        """Sweeps the budget still covers; refreshes the halo when it covers none.  A single slab has no cut and so no
        budget to keep: any number of sweeps runs at once."""
        if self.dom.world == 1:
            self.spent = 0
            return 1 << 30
        cap = self.dom.halo
        if cap < 2:
            raise ValueError("a halo of %d columns covers no sweep" % cap)
        if self.spent is None or cap - self.spent < 2:
            self.dom.exchange(iterate)
            self.spent = 0
        return (cap - self.spent) // 2

    def stage(self, iterate, radius, fn):
        """Runs fn() -- a stencil of `radius` columns evaluated on the local planes from the iterate (weights, a residual, a
        warp of bounded reach) -- after making sure the halo covers it: what fn produces is exact in the owned columns and as far
        into the halo as the iterate was, less `radius`; the sweeps that follow draw on the same budget."""
        if self.dom.world > 1:
            if radius > self.dom.halo:
                raise ValueError("stage radius %d exceeds the halo (%d)" % (radius, self.dom.halo))
            if self.spent is None or self.spent + radius > self.dom.halo:
                self.dom.exchange(iterate)
                self.spent = 0
            self.spent += radius
        return fn()

    def solve(self, iterate, coef, iters, omega):
        """The halo budget carries over from call to call: with sweeps_per_exchange = 8 and iters = 4 every second call
        exchanges (the owned columns are exact as long as no more than halo/2 sweeps ran since the last refresh)."""
        done = 0
        while done < iters:
            k = min(self._room(iterate), iters - done)
            self.sweep_fn(iterate, coef, k, omega, self.dom.col0)
            self.spent += 2 * k
            done += k

    def solve_pingpong(self, iterate, coef, iters, omega):
        """solve() for sweep functions that can write their result to a second plane set (`out_of_place`): every run of sweeps
        relaxes the current planes into the other set, so no launch sequence ends with a device-to-device copy.  Returns the
        planes that hold the iterate now -- the caller carries them into the next call (the halo budget goes with them)."""
        if not getattr(self.sweep_fn, "out_of_place", False):
            self.solve(iterate, coef, iters, omega)
            return iterate
        if self._alt is None or any(a.shape != b.shape for a, b in zip(self._alt, iterate)) or any(a is b for a in self._alt for b in iterate):
            self._alt = [torch.empty_like(t) for t in iterate]
        cur, other = list(iterate), self._alt
        done = 0
        while done < iters:
            k = min(self._room(cur), iters - done)
            self.sweep_fn(cur, coef, k, omega, self.dom.col0, out=other)
            cur, other = other, cur
            self.spent += 2 * k
            done += k
        self._alt = other
        return cur
