"""ctypes binding of libpdeip.so (include/pdeip.h).  Plumbing only: no arithmetic lives here.

The library is the product.  If it is missing or does not load, every entry point raises
PdeipError -- there is no CPU fallback of any kind.
"""
import ctypes
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PDEIP_LIB") or os.path.join(HERE, "libpdeip.so")  # PDEIP_LIB: a diagnostic build of the same library (tools/)

PDEIP_OK = 0
PDEIP_ERR_ARG = 1
PDEIP_ERR_SOLVER = 2
PDEIP_ERR_UNSUPPORTED = 3
PDEIP_ERR_DEVICE = 4
PDEIP_ERR_NOMEM = 5
MODE_EXACT_ORDER = 0
MODE_RED_BLACK = 1

_P = ctypes.c_void_p  # float* (host or device), passed as an address
_I = ctypes.c_int
_F = ctypes.c_float


class PdeipError(RuntimeError):
    """An error reported by libpdeip.so (the text a MEX stub would hand to mexErrMsgTxt)."""

    def __init__(self, code, message):
        super().__init__(message)
        self.code = code


def _sig(n_ptr_before, tail):
    return [_P] * n_ptr_before + tail


# name -> argtypes, in the order of include/pdeip.h
SIGNATURES = {
    # host-pointer drop-in entry points
    "pdeip_oflow_sor_elin4": _sig(11, [_I, _I, _I, _I, _F, _I, _P, _P, _P, _P]),
    "pdeip_oflow_sor_llin4": _sig(13, [_I, _I, _I, _I, _F, _I, _P, _P, _P, _P]),
    "pdeip_oflow_sor_llin8": _sig(17, [_I, _I, _I, _I, _F, _I, _P, _P, _P, _P]),
    "pdeip_oflow_lhs_elin4": _sig(9, [_I, _I, _I, _P, _P]),
    "pdeip_oflow_lhs_llin4": _sig(11, [_I, _I, _I, _P, _P]),
    "pdeip_disp_sor_llin4": _sig(8, [_I, _I, _I, _F, _I, _P, _P]),
    "pdeip_disp_sor_llin_sym4": _sig(16, [_I, _I, _I, _F, _I, _P, _P]),
    "pdeip_pde_sor4": _sig(7, [_I, _I, _I, _I, _F, _I, _P]),
    "pdeip_pde_sor8": _sig(11, [_I, _I, _I, _I, _F, _I, _P]),
    "pdeip_diffweights6": [_P, _I, _I, _I, _F, _P, _P, _P, _P],
    "pdeip_warp_bilinear": [_P, _P, _P, _I, _I, _I, _P],
    "pdeip_fst_derivatives5": [_P, _P, _I, _I, _I, _P, _P, _P],
    "pdeip_snd_derivatives5": [_P, _P, _I, _I, _I, _P, _P, _P, _P, _P],
    # device-pointer entry points (first argument: hipStream_t)
    "pdeip_oflow_sor_elin4_dev": _sig(1 + 11, [_I, _I, _I, _F, _I, _I]),
    "pdeip_oflow_sor_elin4_dev_to": _sig(1 + 13, [_I, _I, _I, _F, _I, _I]),
    "pdeip_oflow_sor_llin4_dev_to": _sig(1 + 15, [_I, _I, _I, _F, _I, _I]),
    "pdeip_disp_sor_llin4_dev_to": _sig(1 + 9, [_I, _I, _I, _F, _I, _I]),
    "pdeip_pde_sor4_dev_to": _sig(1 + 8, [_I, _I, _I, _I, _F, _I, _I]),
    "pdeip_oflow_sor_llin4_dev": _sig(1 + 13, [_I, _I, _I, _F, _I, _I]),
    "pdeip_disp_sor_llin4_dev": _sig(1 + 8, [_I, _I, _I, _F, _I, _I]),
    "pdeip_disp_sor_llin_sym4_dev": _sig(1 + 16, [_I, _I, _I, _F, _I, _I, _I]),
    "pdeip_pde_sor4_dev": _sig(1 + 7, [_I, _I, _I, _I, _F, _I, _I]),
    "pdeip_pde_sor8_dev": _sig(1 + 11, [_I, _I, _I, _I, _F, _I, _I]),
    "pdeip_oflow_alr_elin4_dev": _sig(1 + 11, [_I, _I, _I, _F, _I]),
    "pdeip_oflow_alr_llin4_dev": _sig(1 + 13, [_I, _I, _I, _F, _I]),
    "pdeip_oflow_alr_llin8_dev": _sig(1 + 17, [_I, _I, _I, _F, _I]),
    "pdeip_disp_alr_llin4_dev": _sig(1 + 8, [_I, _I, _I, _F, _I]),
    "pdeip_pde_alr4_dev": _sig(1 + 7, [_I, _I, _I, _I, _F, _I]),
    "pdeip_pde_alr8_dev": _sig(1 + 11, [_I, _I, _I, _I, _F, _I]),
    "pdeip_flow_warp_dev": [_P, _P, _P, _P, _I, _P, _I, _I, _I, _P, _P],
    "pdeip_flow_coords_dev": [_P, _P, _P, _I, _I, _P, _P],
    "pdeip_flow_assemble_dev": [_P, _P, _P, _P, _I, _F, _P, _P, _P, _I, _F, _P, _P, _F, _I, _I, _P, _P, _P, _P, _P],
    "pdeip_disp_assemble_dev": [_P, _P, _P, _I, _F, _P, _P, _I, _F, _P, _F, _I, _I, _P, _P],
    "pdeip_add_dev": [_P, _P, _P, _I, _I, _P],
    "pdeip_hs_assemble_dev": [_P, _P, _P, _I, _F, _F, _I, _I, _P, _P, _P, _P, _P],
    "pdeip_fas_gauss5_dev": [_P, _P, _I, _I, _I, _P, _P],
    "pdeip_fas_down_dev": [_P, _P, _I, _I, _I, _P],
    "pdeip_fas_prepare_dev": [_P, _P, _P, _I, _I, _I, _F, _F, _P],
    "pdeip_fas_assemble_dev": [_P] * 6 + [_I, _I, _I, _F, _F, _F, _I] + [_P] * 6,
    "pdeip_fas_assemble_weights_dev": [_P] * 6 + [_I, _I, _I, _F, _F, _F] + [_P] * 9,
    "pdeip_fas_restrict_dev": [_P, _P, _I, _I, _I, _F, _P],
    "pdeip_fas_rhs_dev": [_P, _P, _P, _P, _I, _I, _I, _P],
    "pdeip_fas_prolong_add_dev": [_P, _P, _I, _I, _P, _P, _I, _I, _F],
    "pdeip_fas_upscale_dev": [_P, _P, _I, _I, _F, _I, _I, _P],
    "pdeip_ad_weights_dev": [_P, _P, _I, _I, _I, ctypes.c_double] + [_P] * 8,
    "pdeip_tv4_assemble_dev": [_P, _P, _P, _I, _I, _I, _F] + [_P] * 6,
    "pdeip_flow_assemble_weights_dev": [_P, _P, _P, _P, _I, _F, _P, _P, _P, _P, _P, _I, _F, _P, _P, _P, _P, _F, _I, _I] + [_P] * 9,
    "pdeip_flow_assemble_gradmag_dev": [_P, _P, _P, _P, _I, _F, _P, _P, _P, _P, _P, _I, _F, _P, _P, _F, _I, _I, _P, _P, _P, _P, _P],
    "pdeip_disp_assemble_gradmag_dev": [_P, _P, _P, _I, _F, _P, _P, _P, _P, _I, _F, _P, _F, _I, _I, _P, _P],
    "pdeip_rgb2grad_dev": [_P, _P, _I, _I, _I, _P],
    "pdeip_sym_warp_flow_dev": [_P, _P, _P, _I, _I, _P],
    "pdeip_sym_flow_terms_dev": [_P, _P, _P, _I, _I, _P, _P, _P, _P],
    "pdeip_sym_assemble_dev": [_P] * 7 + [_I] + [_P] * 5 + [_F, _F, _F, ctypes.c_double, ctypes.c_double, _I, _I, _I, _P, _P],
    "pdeip_flow_apriori_dev": [_P, _P, _P, _P, ctypes.c_double, ctypes.c_double, ctypes.c_double, _I, _I, _I, _I, _P, _P],
    "pdeip_disp_apriori_dev": [_P, _P, _P, _P, ctypes.c_double, ctypes.c_double, ctypes.c_double, _I, _I, _I, _I, _P, _P],
    "pdeip_pyr_resize_dev": [_P, _P, _I, _I, _I, _I, _I, _I, _P],
    "pdeip_pyr_smooth_dev": [_P, _P, _I, _I, _I, _P, _I, _P],
    "pdeip_flow_opdiffweights_dev": [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P],
    "pdeip_selftest_inv_sqrt": [_I, ctypes.c_uint],
    "pdeip_tv_assemble_dev": [_P, _P, _P, _I, _I, _I, _F] + [_P] * 10,
    "pdeip_median3_pair_dev": [_P, _P, _P, _P, _P, _I, _I, _P, _P],
    "pdeip_median3_dev": [_P, _P, _P, _I, _I, _P],
    "pdeip_oflow_res_elin4_dev": _sig(1 + 13, [_I, _I, _I]),
    "pdeip_oflow_lhs_elin4_dev": _sig(1 + 11, [_I, _I, _I]),
    "pdeip_oflow_res_llin4_dev": _sig(1 + 15, [_I, _I, _I]),
    "pdeip_oflow_lhs_llin4_dev": _sig(1 + 13, [_I, _I, _I]),
    "pdeip_diffweights6_dev": [_P, _P, _I, _I, _I, _F, _P, _P, _P, _P],
    "pdeip_warp_bilinear_dev": [_P, _P, _P, _P, _I, _I, _I, _P],
    "pdeip_fst_derivatives5_dev": [_P, _P, _P, _I, _I, _I, _P, _P, _P],
    "pdeip_snd_derivatives5_dev": [_P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P],
    # whole drivers, resident (csrc/pdeip_drivers.hip)
    "pdeip_flow_nd_llin": [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    "pdeip_disp_nd_llin": [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P],
    "pdeip_flow_hs_elin": [_P, _I, _I, _I, _P, _P, _P],
    "pdeip_disp_nd_llin_sym": [_P, _P, _I, _I, _I, _P, _P],
    "pdeip_flow_ad_llin": [_P, _I, _I, _I, _I, _I, _P, ctypes.c_double, _I, _P, _P, _P, _P],
    "pdeip_flow_fas_fmg_elin": [_P, _I, _I, _I, _P, _P, _P],
    "pdeip_tvdenoise8": [_P, _I, _I, _I, _P, _P],
    "pdeip_tvdenoise4": [_P, _I, _I, _I, _P, _P],
    # library state
    "pdeip_set_mode": [_I],
    "pdeip_get_mode": [],
    "pdeip_set_device": [_I],
    "pdeip_set_devices": [_I, ctypes.POINTER(ctypes.c_int)],
    "pdeip_get_devices": [ctypes.POINTER(ctypes.c_int), _I],
    "pdeip_release": [],
    "pdeip_last_launch_count": [],
    "pdeip_workspace_generation": [],
    "pdeip_persist_error": [],
    "pdeip_debug_persist_order": [_I, _I, _I, ctypes.POINTER(ctypes.c_int)],
    "pdeip_debug_raise_abort": [],
    "pdeip_debug_rcp_check": [ctypes.POINTER(ctypes.c_ulonglong)],
    "pdeip_profile_enable": [_I],
    "pdeip_profile_read": [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)],
}
STRING_FUNCS = ("pdeip_version", "pdeip_last_error")

_lib = None


def load():
    """Load libpdeip.so once and declare every prototype.  Raises PdeipError if it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PdeipError(PDEIP_ERR_DEVICE,
                         "libpdeip.so is not built (%s). Run `python __graft_entry__.py build`; "
                         "there is no CPU fallback." % LIB_PATH)
    # Load order: the torch wheel carries its own copy of the HIP runtime.  If libpdeip.so (linked against
    # /opt/rocm's) initialises the GPU first and torch is imported afterwards, torch's runtime finds "No HIP GPUs";
    # with torch first both share one runtime.  device.py / slab.py need torch anyway; mex_api alone does not, so a
    # process without torch installed simply skips this.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as exc:  # e.g. libamdhip64 missing
        raise PdeipError(PDEIP_ERR_DEVICE, "cannot load %s: %s" % (LIB_PATH, exc)) from exc
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    for name in STRING_FUNCS:
        fn = getattr(lib, name)
        fn.argtypes = []
        fn.restype = ctypes.c_char_p
    _lib = lib
    return lib


def last_error():
    return load().pdeip_last_error().decode("utf-8", "replace")


def check(rc):
    if rc != PDEIP_OK:
        raise PdeipError(rc, last_error())


_fn_cache = {}


def call(name, *args):
    """Call an int-returning entry point and raise PdeipError on a non-zero status."""
    fn = _fn_cache.get(name)
    if fn is None:
        fn = _fn_cache[name] = getattr(load(), name)
    rc = fn(*args)
    if rc != PDEIP_OK:
        raise PdeipError(rc, last_error())


def set_mode(mode):
    call("pdeip_set_mode", int(mode))


def get_mode():
    return load().pdeip_get_mode()


def set_devices(ids):
    ids = [int(i) for i in ids]
    call("pdeip_set_devices", len(ids), (ctypes.c_int * len(ids))(*ids))


def get_devices():
    buf = (ctypes.c_int * 16)()
    n = load().pdeip_get_devices(buf, 16)
    return [buf[k] for k in range(n)]


def profile_enable(on=True):
    call("pdeip_profile_enable", 1 if on else 0)


def profile_read():
    """(elapsed milliseconds of the bracketed sweep launches, number of sweep launches) since the last read."""
    ms, n = ctypes.c_double(0.0), ctypes.c_int(0)
    call("pdeip_profile_read", ctypes.byref(ms), ctypes.byref(n))
    return ms.value, n.value


def version():
    return load().pdeip_version().decode()
