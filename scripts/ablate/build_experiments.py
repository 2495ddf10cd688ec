"""Experimental builds of liby7t.so for the next round's measurements -- macro variants of csrc/y7t_conv.hip linked with the product's other objects:

   nw8    -DY7T_IGEMM_NW=8     (superseded: the same 8-wave kernels are template instances inside liby7t.so now, Y7T_CONV_NW8=1 -- csrc/y7t_conv.hip)
                               512-thread workgroups (2 x 4 waves); tiles 256x256x64 (default), 256x128x64 (Y7T_CONV_VARIANT=6), 128x128x64 (=7);
                               plain layers with Cin % 64 == 0 and Cout % 128 == 0 only: layer-level tests / timing
                               (Y7T_LIB=.../exp_nw8.so python scripts/bench_conv.py 32)
   fixup  -DY7T_SPLITK_FIXUP   the last workgroup of a tile to arrive reduces the split-K slabs (no k_splitk_reduce launch) when
                               Y7T_CONV_SPLITK=2; batch-1 latency mode (Y7T_LIB=.../exp_fixup.so Y7T_CONV_SPLITK=2 python scripts/latency_mode.py)

   next   -DY7T_NEXT_TRACKER   tracker step: candidate lists on a run-time row stride so that they sit in LDS at 500 objects (the per-component solves are
                               42 % of that frame step and every access of theirs is an L2 round trip today); parity of the macro build is tested on the CPU
                               (tests/test_hostsim_next.py, scripts/parity_sweep.py); measure with Y7T_LIB=.../exp_next.so python scripts/time_tracker.py

Unlike y7t_conv.hip itself these libraries have not run on a GPU yet; the default build's device code is unaffected by the macros
(checked byte for byte against the tested build when they were added).
(y7t_det.h changes layout under Y7T_SPLITK_FIXUP, so that variant recompiles every translation unit.)"""
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from yolov7_tracker_amd import build as b  # noqa: E402

VARIANTS = {"nw8": (["-DY7T_IGEMM_NW=8"], ["y7t_conv.hip"]), "fixup": (["-DY7T_SPLITK_FIXUP=1"], None), "next": (["-DY7T_NEXT_TRACKER=1"], ["y7t_tracker.hip"])}
b.build()
tmp = os.path.join(b.OBJ, "exp")
os.makedirs(tmp, exist_ok=True)
for name in sys.argv[1:] or sorted(VARIANTS):
    macros, only = VARIANTS[name]
    objs = []
    for f in b._sources():
        if only is not None and f not in only:
            objs.append(os.path.join(b.OBJ, f[:-4] + ".o"))
            continue
        o = os.path.join(tmp, "%s_%s.o" % (f[:-4], name))
        subprocess.check_call([b.HIPCC] + b.FLAGS + b.FILE_FLAGS.get(f, []) + macros + ["-c", os.path.join(b.CSRC, f), "-o", o])
        objs.append(o)
    lib = os.path.join(b.LIBDIR, "exp_%s.so" % name)
    subprocess.check_call([b.HIPCC, "--offload-arch=" + b.ARCH, "-shared", "-fPIC", "-o", lib] + objs)
    print("built", lib)
