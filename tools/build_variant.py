"""A/B aid: link a variant of libpdeip.so whose pdeip_sor5.hip (the 5-point solvers) is compiled with extra flags.
    python tools/build_variant.py NAME -DFOO=1 ...   ->  pde-based-image-processing_amd/libpdeip_NAME.so
Run the variant with PDEIP_LIB=<that file>; built here (no GPU needed), the .so travels with the snapshot."""
import importlib.util, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pde-based-image-processing_amd")
spec = importlib.util.spec_from_file_location("pdeip_build", os.path.join(PKG, "build.py"))
b = importlib.util.module_from_spec(spec); spec.loader.exec_module(b)
name, flags = sys.argv[1], sys.argv[2:]
units = [u for u in flags if u.endswith(".hip")] or ["pdeip_sor5.hip"]
flags = [f for f in flags if not f.endswith(".hip")]
b.build(verbose=False)
objs = []
for u in b._units():
    if u in units:
        o = os.path.join(b.OBJ, u.replace(".hip", "_%s.o" % name))
        subprocess.run(["/opt/rocm/bin/hipcc"] + b.CFLAGS + flags + ["-c", "-o", o, os.path.join(b.CSRC, u)], check=True, cwd=b.CSRC)
        objs.append(o)
    else:
        objs.append(b._obj(u))
out = os.path.join(PKG, "libpdeip_%s.so" % name)
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True, cwd=b.CSRC)
print(out)
