"""CPU: libpdeip.so loads and exports every symbol include/pdeip.h declares; the ctypes table covers them.
(No compute calls: there is no GPU here.)"""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "pdeip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pdeip_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for gateway in ("pdeip_oflow_sor_elin4", "pdeip_oflow_sor_llin4", "pdeip_oflow_sor_llin8", "pdeip_oflow_lhs_elin4",
                    "pdeip_oflow_lhs_llin4", "pdeip_disp_sor_llin4", "pdeip_pde_sor4", "pdeip_pde_sor8",
                    "pdeip_diffweights6", "pdeip_warp_bilinear", "pdeip_fst_derivatives5", "pdeip_snd_derivatives5"):
        assert gateway in syms and (gateway + "_dev" in syms or gateway.endswith(("llin8", "lhs_elin4", "lhs_llin4"))), gateway


def test_library_exports_every_declared_symbol(pdeip):
    lib = ctypes.CDLL(pdeip.capi.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, "declared in pdeip.h but not exported: %s" % missing


def test_ctypes_table_matches_header(pdeip):
    table = set(pdeip.capi.SIGNATURES) | set(pdeip.capi.STRING_FUNCS)
    assert table == set(declared_symbols())


def test_state_calls_without_gpu(pdeip):
    capi = pdeip.capi
    assert "gfx950" in capi.version()
    capi.set_mode(capi.MODE_RED_BLACK)
    assert capi.get_mode() == capi.MODE_RED_BLACK
    capi.set_mode(capi.MODE_EXACT_ORDER)
    try:
        capi.set_mode(7)
        raise AssertionError("unknown mode accepted")
    except capi.PdeipError as exc:
        assert exc.code == capi.PDEIP_ERR_ARG and "ordering" in str(exc)
