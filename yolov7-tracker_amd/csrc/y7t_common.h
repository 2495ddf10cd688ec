// y7t_common.h -- host-side helpers shared by the C-ABI translation units of liby7t.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include "../../include/y7t.h"

void y7t_set_error(const char* fmt, ...);
void y7t_note_kernel(const char* fmt, ...);   // see y7t_last_kernel()

#define Y7T_HIP_CHECK(expr)                                                                 \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            y7t_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return Y7T_E_HIP;                                                               \
        }                                                                                   \
    } while (0)

#define Y7T_ARG_CHECK(cond)                                                  \
    do {                                                                     \
        if (!(cond)) {                                                       \
            y7t_set_error("bad argument: %s (%s:%d)", #cond, __FILE__, __LINE__); \
            return Y7T_E_ARG;                                                \
        }                                                                    \
    } while (0)

#define Y7T_LAUNCH_CHECK() Y7T_HIP_CHECK(hipGetLastError())

// Run-time switches.  The PRODUCT library (lib/liby7t.so) answers every EXPERIMENT switch with its measured default -- it does not read the environment for them and
// carries none of the timing-ablation ("wrong results") or tile-variant instances; the same sources built with -DY7T_ABLATE_BUILD give lib/liby7t_ablate.so, which
// does (load it with Y7T_LIB=<path>; scripts/*ablat*.sh, scripts/sweep_conv.py).  y7t_switch(): the handful of switches the product itself keeps (README).
#include <stdlib.h>
static inline int y7t_switch(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#if defined(Y7T_ABLATE_BUILD)
#define Y7T_ABLATE 1
static inline int y7t_exp_switch(const char* name, int dflt) { return y7t_switch(name, dflt); }
#else
#define Y7T_ABLATE 0
static inline int y7t_exp_switch(const char*, int dflt) { return dflt; }
#endif

// One-time set-up that is PER DEVICE (kernel attributes such as hipFuncAttributeMaxDynamicSharedMemorySize; the CU count): a done-bit per device, set after the
// call succeeded, so that a second device of the process -- or two threads racing -- never skip it (ADVICE r3 for the tracker, r4 for the conv / stem kernels).
#include <atomic>
struct Y7TOncePerDevice { std::atomic<unsigned long long> done{0}; };
template <class F>
static inline int y7t_once_per_device(Y7TOncePerDevice& o, F&& setup) {
    int dev = 0;
    Y7T_HIP_CHECK(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (o.done.load(std::memory_order_acquire) & bit) return 0;
    if (int e = setup()) return e;
    o.done.fetch_or(bit, std::memory_order_release);
    return 0;
}
// compute units of the current device (cached per device)
static inline int y7t_num_cus() {
    static std::atomic<int> ncu[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int v = ncu[dev & 63].load(std::memory_order_relaxed);
    if (v > 0) return v;
    hipDeviceProp_t prop;
    v = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ncu[dev & 63].store(v, std::memory_order_relaxed);
    return v;
}
