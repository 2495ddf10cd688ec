"""Build liby7t.so (all HIP translation units of csrc/) for gfx950 with hipcc, in-tree -- and liby7t_ablate.so, the same sources with -DY7T_ABLATE_BUILD:
the measuring build that reads the experiment switches from the environment and carries the timing-ablation / tile-variant instances ("wrong results" kernels
among them), which the product library does not (csrc/y7t_common.h; load it with Y7T_LIB=<path>).

    python -m yolov7_tracker_amd.build          # incremental, both libraries
    python -m yolov7_tracker_amd.build --force
    python -m yolov7_tracker_amd.build --product-only

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container; the
resulting yolov7-tracker_amd/lib/liby7t.so travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "liby7t.so")
OBJ_ABLATE = os.path.join(HERE, "build_ablate")
LIB_ABLATE = os.path.join(LIBDIR, "liby7t_ablate.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
# -ffp-contract=off: the float64 tracker arithmetic must round like the plain-C oracle
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-I", os.path.join(os.path.dirname(HERE), "include")]
# per-file overrides (the conv kernels want contraction: fp32 accumulate of fp16 products)
# -fno-slp-vectorize: the SLP vectoriser pairs the epilogues' fp32 multiplies / adds into v_pk_mul_f32 / v_pk_add_f32, and ONE packed-fp32 instruction holds the matrix
# pipe off for ~9 ns -- from either wave of a SIMD -- where two plain fp32 instructions per MFMA cost nothing (scripts/ubench/issue_classes.hip, profiles/r03_issue_classes.txt)
_CONV = ["-ffp-contract=fast", "-fno-slp-vectorize"]
FILE_FLAGS = {"y7t_conv.hip": _CONV, "y7t_conv_patch.hip": _CONV, "y7t_conv_patch_s2.hip": _CONV, "y7t_stem.hip": ["-fno-slp-vectorize"],
              "y7t_conv_ws.hip": _CONV + ["-mllvm", "-pragma-unroll-threshold=10000000"], "y7t_conv_ws128.hip": _CONV + ["-mllvm", "-pragma-unroll-threshold=10000000"], "y7t_conv_p8.hip": _CONV, "y7t_conv_ws_s2.hip": _CONV + ["-mllvm", "-pragma-unroll-threshold=10000000"], "y7t_post.hip": []}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "y7t.h"))
    return hs


def _compile(src, force, ablate=False):
    obj = os.path.join(OBJ_ABLATE if ablate else OBJ, src[:-4] + ".o")
    spath = os.path.join(CSRC, src)
    newest = max([os.path.getmtime(spath)] + [os.path.getmtime(h) for h in _headers()])
    if not force and os.path.exists(obj) and os.path.getmtime(obj) >= newest:
        return obj, False
    cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(src, []) + (["-DY7T_ABLATE_BUILD"] if ablate else []) + ["-c", spath, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    return obj, True


def _build_one(force, verbose, ablate):
    obj_dir, lib = (OBJ_ABLATE, LIB_ABLATE) if ablate else (OBJ, LIB)
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = _sources()
    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, ablate), srcs))
    objs = [o for o, _ in res]
    if verbose:      # which translation units hipcc actually compiled in this call (the build check must show that it built)
        done = [s_ for s_, (_, c) in zip(srcs, res) if c]
        print("%s: compiled %d of %d translation units for %s%s" % (os.path.basename(lib), len(done), len(srcs), ARCH, (": " + " ".join(done)) if done else " (objects newer than every source and header)"))
    if force or any(c for _, c in res) or not os.path.exists(lib):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", lib] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("built", lib)
    return lib


def build(force=False, verbose=False, ablate=True):
    """-> path of the product library; ablate=True also (re)builds lib/liby7t_ablate.so"""
    lib = _build_one(force, verbose, False)
    if ablate:
        _build_one(force, verbose, True)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, ablate="--product-only" not in sys.argv))
