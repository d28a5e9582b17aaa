"""pde-based-image-processing_amd -- MI355X (gfx950) implementation of the MEX-side stencil hot path
of JediZ/PDE-based-image-processing.

  csrc/        hand-written HIP kernels + the C-ABI (libpdeip.so, include/pdeip.h)
  capi.py      ctypes binding of the C-ABI
  mex_api.py   mirror of the reference's MEX gateways (same names, arguments, error behaviour)
  device.py    device-resident calls on torch tensors (streams, no host copies)
  slab.py      column-slab decomposition of one frame across GPUs + RCCL halo exchange

The directory name is not a Python identifier; import it with
    importlib.import_module("pde-based-image-processing_amd")
or through the alias module `pdeip_amd` at the repository root.
"""
from . import capi  # noqa: F401
from . import mex_api  # noqa: F401
from .capi import MODE_EXACT_ORDER, MODE_RED_BLACK, PdeipError  # noqa: F401

__all__ = ["capi", "mex_api", "MODE_EXACT_ORDER", "MODE_RED_BLACK", "PdeipError"]
