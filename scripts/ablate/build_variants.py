"""Bottleneck ablations of the implicit-GEMM conv kernel (results are WRONG by construction; only the timing is used):
   1 = no global->LDS DMA after the prologue, 2 = operand fragments read from LDS only in the first K-step,
   3 = both (MFMA + barriers only), 4 = DMA + LDS reads but no MFMA.
Each variant is a patched copy of csrc/y7t_conv.hip linked with the product's other objects into yolov7-tracker_amd/lib/ablate_<n>.so;
run `Y7T_LIB=.../ablate_<n>.so python scripts/bench_conv.py 32` on the GPU box."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolov7_tracker_amd import build as b
b.build()
src = open(os.path.join(b.CSRC, "y7t_conv.hip")).read()
DECL = "        half8 wf[2][TN], xf[2][TM];\n"
LOOP = "    for (int kt = 0; kt < nk; ++kt) {\n"
MFMA = "for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cb][i], xf[cb][j], acc[i][j], 0, 0, 0);"
assert src.count(DECL) == 1 and src.count(LOOP) == 1 and src.count(MFMA) == 1
def variant(n):
    s = src
    if n & 1:
        s = s.replace("const bool do_load = kt + NST - 1 < nk;", "const bool do_load = false;")
    if n & 2:
        s = s.replace(DECL, "").replace(LOOP, "    half8 wf[2][TN], xf[2][TM];\n" + LOOP)
        s = s.replace("        read_frags(0, 0);\n", "        if (kt == 0) read_frags(0, 0);\n")
        s = s.replace("if (ks + 1 < KS) read_frags(ks + 1, cb ^ 1);", "if (kt == 0 && ks + 1 < KS) read_frags(ks + 1, cb ^ 1);")
    if n & 4:
        s = s.replace(MFMA, 'for (int j = 0; j < TM; ++j) asm volatile("" ::"v"(wf[cb][i]), "v"(xf[cb][j]));')
    return s
tmp = os.path.join(b.OBJ, "ablate"); os.makedirs(tmp, exist_ok=True)
objs = [os.path.join(b.OBJ, f[:-4] + ".o") for f in b._sources() if f != "y7t_conv.hip"]
for n in [int(a) for a in sys.argv[1:]] or (1, 2, 3, 4):
    p = os.path.join(tmp, "y7t_conv_%d.hip" % n); open(p, "w").write(variant(n))
    o = p[:-4] + ".o"
    subprocess.check_call([b.HIPCC] + b.FLAGS + ["-ffp-contract=fast", "-I", b.CSRC, "-c", p, "-o", o])
    lib = os.path.join(b.LIBDIR, "ablate_%d.so" % n)
    subprocess.check_call([b.HIPCC, "--offload-arch=" + b.ARCH, "-shared", "-fPIC", "-o", lib, o] + objs)
    print("built", lib)
