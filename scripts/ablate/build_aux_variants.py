"""Cache-policy experiment on the implicit-GEMM kernel's DMA streams: the `aux` immediate of raw_ptr_buffer_load_lds (bit0 sc0/glc, bit1 nt/slc, ...) for the
activation (A) and weight (W) loads.  Builds yolov7-tracker_amd/lib/aux_<name>.so; run `Y7T_LIB=... python scripts/bench_conv.py 32 10`."""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolov7_tracker_amd import build as b
b.build()
src = open(os.path.join(b.CSRC, "y7t_conv.hip")).read()
def variant(a_aux, w_aux):
    s = src
    # A loads: calls whose first argument is xr / xr2; W loads: wr
    def repl(m):
        call = m.group(0)
        aux = a_aux if ("(xr," in call or "(xr2," in call) else w_aux
        return re.sub(r",\s*0\s*,\s*0\s*\)\s*;$", ", 0, %d);" % aux, call)
    s2 = re.sub(r"__builtin_amdgcn_raw_ptr_buffer_load_lds\((?:xr2|xr|wr),[^;]*;", repl, s, flags=re.S)
    return s2
tmp = os.path.join(b.OBJ, "ablate"); os.makedirs(tmp, exist_ok=True)
objs = [os.path.join(b.OBJ, f[:-4] + ".o") for f in b._sources() if f != "y7t_conv.hip"]
for name, a, w in (("a1", 1, 0), ("a2", 2, 0), ("w2", 0, 2), ("a2w2", 2, 2)):
    p = os.path.join(tmp, "y7t_conv_aux_%s.hip" % name); open(p, "w").write(variant(a, w))
    n = open(p).read().count(", 0, %d);" % (a or w))
    o = p[:-4] + ".o"
    subprocess.check_call([b.HIPCC] + b.FLAGS + ["-ffp-contract=fast", "-I", b.CSRC, "-c", p, "-o", o])
    lib = os.path.join(b.LIBDIR, "aux_%s.so" % name)
    subprocess.check_call([b.HIPCC, "--offload-arch=" + b.ARCH, "-shared", "-fPIC", "-o", lib, o] + objs)
    print("built", lib, "patched calls:", n)
