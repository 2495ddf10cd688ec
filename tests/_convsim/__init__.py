"""tests/_convsim -- TEST INFRASTRUCTURE ONLY: the implicit-GEMM convolution kernel of csrc/y7t_conv.hip compiled for the CPU from its real source and run
work-item by work-item (fake/hip/hip_runtime.h models the gfx950 builtins it uses; runtime.inc runs workgroups as OS threads).  Checks the kernel's index
arithmetic against a plain convolution where no GPU exists, for the shipped 4-wave build and for experimental builds (macros).  Never imported by the product."""
import ctypes
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(os.path.dirname(_HERE))
_CSRC = os.path.join(_ROOT, "yolov7-tracker_amd", "csrc")
_CLANG = os.environ.get("Y7T_HOST_CLANG", "/opt/rocm/lib/llvm/bin/clang++")      # ext_vector_type / _Float16: clang, compiling for the host
_libs = {}


def build(defs=()):
    tag = "".join(c for c in "".join(defs) if c.isalnum()) or "default"
    bdir = os.path.join(_HERE, "build_" + tag)
    so = os.path.join(bdir, "libconvsim.so")
    srcs = [os.path.join(_CSRC, f) for f in ("y7t_conv.hip", "y7t_conv_common.h", "y7t_det.h", "y7t_common.h", "y7t_conv_patch.hip", "y7t_conv_patch_s2.hip", "y7t_conv_ws.hip", "y7t_conv_p8.hip", "y7t_conv_ws_s2.hip", "y7t_conv_ws128.hip")]
    deps = srcs + [os.path.join(_HERE, "runtime.inc"), os.path.join(_HERE, "fake", "hip", "hip_runtime.h"), os.path.abspath(__file__)]
    if os.path.exists(so) and os.path.getmtime(so) >= max(os.path.getmtime(d) for d in deps):
        return so
    os.makedirs(bdir, exist_ok=True)
    text = open(srcs[0]).read()
    text, n = re.subn(r"asm volatile\([^;]*\);", ";", text)          # the s_waitcnt statements: loads complete at once here
    assert n == 3, n
    common = open(srcs[1]).read()
    common = common.replace("#define GLOBAL_AS __attribute__((address_space(1)))", "#define GLOBAL_AS").replace(
        "#define LDS_AS __attribute__((address_space(3)))", "#define LDS_AS")
    assert "address_space" not in common
    open(os.path.join(bdir, "y7t_conv_common.h"), "w").write(common)
    open(os.path.join(bdir, "convsim.cpp"), "w").write(text + "\n" + open(os.path.join(_HERE, "runtime.inc")).read())
    patch = re.sub(r"asm volatile\([^;]*\);", ";", open(srcs[4]).read())      # waits, and the register pins of the ablation instances
    assert "asm" not in patch
    # a block-scope `extern` inside the file's anonymous namespace would declare (anonymous)::smem: let the name find the global LDS array instead
    patch, n = re.subn(r"extern __shared__ __attribute__\(\(aligned\(16\)\)\) char smem\[\];", "", patch)
    assert n >= 1
    open(os.path.join(bdir, "convsim_patch.cpp"), "w").write(patch)
    s2 = re.sub(r"asm volatile\([^;]*\);", ";", open(srcs[5]).read())          # the stride-2 patch kernel (opt-in experiment): same treatment
    assert "asm" not in s2
    s2, n = re.subn(r"extern __shared__ __attribute__\(\(aligned\(16\)\)\) char smem\[\];", "", s2)
    assert n == 1
    open(os.path.join(bdir, "convsim_patch_s2.cpp"), "w").write(s2)
    ws = re.sub(r"asm volatile\([^;]*\);", ";", open(srcs[6]).read())          # the weights-stationary 64 -> 64 kernel: same treatment
    assert "asm volatile" not in ws          # (the spelled-out MFMA -- plain asm -- sits behind `#if defined(Y7T_CONVSIM) ... #else`: not compiled here)
    ws, n = re.subn(r"extern __shared__ __attribute__\(\(aligned\(16\)\)\) char smem\[\];", "", ws)
    assert n == 1
    open(os.path.join(bdir, "convsim_ws.cpp"), "w").write(ws)
    p8 = open(srcs[7]).read()                                                     # the 256 x 256 x 64 ping-pong kernel: its waits are a macro that becomes the DMA model's cs_vmcnt
    assert "asm volatile" in p8 and "P8_VMCNT" in p8
    p8, n = re.subn(r"extern __shared__ __attribute__\(\(aligned\(16\)\)\) char smem\[\];", "", p8)
    assert n == 2      # (the one-tile kernel and the persistent form, round 6)
    open(os.path.join(bdir, "convsim_p8.cpp"), "w").write(p8)
    ws2 = open(srcs[8]).read()                                                    # the stride-2 weights-stationary kernel (64 -> 128): waits as a macro, like p8
    assert "WS2_VMCNT" in ws2
    ws2, n = re.subn(r"extern __shared__ __attribute__\(\(aligned\(16\)\)\) char smem\[\];", "", ws2)
    assert n == 1
    open(os.path.join(bdir, "convsim_ws_s2.cpp"), "w").write(ws2)
    ws128 = re.sub(r"asm volatile\([^;]*\);", ";", open(srcs[9]).read())    # the 128-channel weights-stationary kernel: same treatment as the 64-channel one
    assert "asm volatile" not in ws128
    ws128, n = re.subn(r"extern __shared__ __attribute__\(\(aligned\(16\)\)\) char smem\[\];", "", ws128)
    assert n == 1
    open(os.path.join(bdir, "convsim_ws128.cpp"), "w").write(ws128)
    cmd = [_CLANG, "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-w", "-ffp-contract=off", "-I", os.path.join(_HERE, "fake"), "-I", _CSRC,
           "-I", os.path.join(_ROOT, "include")] + list(defs) + ["-o", so, os.path.join(bdir, "convsim.cpp"), os.path.join(bdir, "convsim_patch.cpp"),
                                                                          os.path.join(bdir, "convsim_patch_s2.cpp"), os.path.join(bdir, "convsim_ws.cpp"),
                                                                          os.path.join(bdir, "convsim_p8.cpp"), os.path.join(bdir, "convsim_ws_s2.cpp"), os.path.join(bdir, "convsim_ws128.cpp")]
    subprocess.check_call(cmd)
    return so


def lib(defs=()):
    key = tuple(defs)
    if key not in _libs:
        L = ctypes.CDLL(build(defs))
        L.cs_last_error.restype = ctypes.c_char_p
        L.cs_last_kernel.restype = ctypes.c_char_p
        L.cs_conv.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                              ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_int] * 11
        L.cs_conv_dual.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p] + [ctypes.c_int] * 8 + [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p] + [ctypes.c_int] * 7
        L.cs_set_dma_deferred.argtypes = [ctypes.c_int]
        _libs[key] = L
    return _libs[key]
