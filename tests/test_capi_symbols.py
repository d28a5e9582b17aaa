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


def _child(code, env):
    import subprocess
    import sys

    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.returncode == 0, out.stderr
    return out.stdout.strip().splitlines()[-1]


_READ_MODE = ("import importlib, sys; sys.path.insert(0, '.'); capi = importlib.import_module('pde-based-image-processing_amd').capi; "
              "%s print(capi.get_mode(), capi.get_devices())")


def test_environment_knobs_reach_the_library(pdeip):
    """PDEIP_MODE / PDEIP_DEVICE(S): what an unchanged MATLAB session sets before starting."""
    assert _child(_READ_MODE % "", {"PDEIP_MODE": "red_black"}).startswith("1 ")
    assert _child(_READ_MODE % "", {"PDEIP_MODE": "exact"}).startswith("0 ")
    assert _child(_READ_MODE % "", {}).startswith("0 ")  # default: the reference's order
    # an explicit call made before the first use wins over the environment
    assert _child(_READ_MODE % "capi.set_mode(0);", {"PDEIP_MODE": "red_black"}).startswith("0 ")
    # device ids that do not exist are refused with a message and the default group stays
    assert _child(_READ_MODE % "", {"PDEIP_DEVICES": "97,98"}).endswith("[0]")
