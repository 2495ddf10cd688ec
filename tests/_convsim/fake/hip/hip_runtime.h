// tests/_convsim/fake/hip/hip_runtime.h -- TEST INFRASTRUCTURE ONLY.
// A host stand-in for the HIP runtime and the gfx950 builtins that csrc/y7t_conv.hip uses, so that the REAL kernel source can be compiled for the CPU and
// run thread by thread (one OS thread per work-item, real barriers): the kernel's index arithmetic -- load geometry, LDS swizzle, fragment mapping, epilogue
// transposition, tile order -- is checked against a plain convolution without a GPU (tests/test_convsim.py).  What it does NOT model: timing, the lgkmcnt waits, LDS bank
// conflicts, register pressure.  The buffer->LDS DMAs land at once, or (cs_dma_deferred, for kernels that mark their vmcnt waits) as late as their waits allow.
#pragma once
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <thread>
#include <vector>

#define __HIP_DEVICE_COMPILE__ 1
#define Y7T_CONVSIM 1
#define __builtin_amdgcn_sched_barrier(m) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(m, n, id) ((void)0)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__
#define __restrict__

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 0 };
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n); return *p ? hipSuccess : 1; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
struct hipDeviceProp_t { int multiProcessorCount; };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 3; return hipSuccess; }   // (persistent kernels: a few workgroups, several tiles each)
static inline const char* hipGetErrorString(hipError_t) { return "convsim"; }

// ---- the running workgroup ------------------------------------------------------------------------------------
extern thread_local dim3 threadIdx;
extern dim3 blockIdx, gridDim, blockDim;
extern char smem[];                                   // the dynamic LDS allocation (`extern __shared__ char smem[]` in the kernel binds to it)
void cs_wg_barrier();
void cs_wave_barrier(int wave);
extern void* cs_xchg[];                               // per-wave exchange slots of the cross-lane builtins
void cs_launch(const std::function<void()>& kernel, dim3 grid, dim3 block);
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) cs_launch([&]() { kernel(__VA_ARGS__); }, grid, block)
#define __syncthreads() cs_wg_barrier()
#define __builtin_amdgcn_s_barrier() cs_wg_barrier()
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_sleep(x) ((void)0)
#define __builtin_amdgcn_readfirstlane(x) (x)          /* (only applied to wave-uniform values) */
#define __threadfence() ((void)0)
#define __builtin_amdgcn_fence(order, scope) ((void)0)   /* (workgroups run one after the other here) */
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))
#define __builtin_amdgcn_exp2f(x) exp2f(x)
template <class T> static inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

// ---- buffer resources and the buffer -> LDS DMA ------------------------------------------------------------------
struct cs_rsrc { const char* base; unsigned bytes; };
#define __amdgpu_buffer_rsrc_t cs_rsrc
static inline cs_rsrc cs_make_rsrc(void* p, int, unsigned bytes, unsigned) { return cs_rsrc{(const char*)p, bytes}; }
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, bytes, flags) cs_make_rsrc((void*)(p), stride, bytes, flags)
// every lane fetches `size` bytes at base + voffset + soffset + imm (dwords past num_records read as zero -- the hardware range check) and the wave writes
// them to LDS at ldsbase + lane * size (M0-based, lane order)
// cs_dma_deferred = 0: the bytes land at issue (the EARLIEST a DMA can land: a re-staged buffer that is still being read shows up as wrong results);
// cs_dma_deferred = 1: they land when the issuing lane's own counted wait forces them -- cs_vmcnt(n), what `s_waitcnt vmcnt(n)` stands for in a kernel that marks its
// waits with a macro -- the LATEST they can land: a fragment read that is not covered by a wait + barrier reads the 0xAB fill.  Together the two runs bracket every
// landing time the hardware can produce for a schedule whose LDS accesses are ordered by its barriers.
struct cs_dma { cs_rsrc r; char* dst; int size; unsigned off; };
extern int cs_dma_deferred;
extern thread_local std::vector<cs_dma> cs_dmaq;
static inline void cs_dma_land(const cs_dma& q) {
    for (int d = 0; d < q.size; d += 4) {
        const unsigned long long o = (unsigned long long)q.off + d;
        uint32_t w = 0;
        if (o + 4 <= q.r.bytes) memcpy(&w, q.r.base + o, 4);
        memcpy(q.dst + d, &w, 4);
    }
}
static inline void cs_buffer_load_lds(cs_rsrc r, void* ldsbase, int size, int voffset, int soffset, int imm, int) {
    const int lane = threadIdx.x & 63;
    const cs_dma q{r, (char*)ldsbase + lane * size, size, (unsigned)voffset + (unsigned)soffset + (unsigned)imm};
    if (cs_dma_deferred) cs_dmaq.push_back(q);
    else cs_dma_land(q);
}
static inline void cs_vmcnt(int n) {      // at most n of this lane's DMAs stay in flight: the older ones land now, in issue order
    while ((int)cs_dmaq.size() > n) { cs_dma_land(cs_dmaq.front()); cs_dmaq.erase(cs_dmaq.begin()); }
}
#define __builtin_amdgcn_raw_ptr_buffer_load_lds(r, lds, size, voff, soff, imm, aux) cs_buffer_load_lds(r, (void*)(lds), size, voff, soff, imm, aux)

// ---- cross-lane builtins -----------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(8))) _Float16 cs_half8;
typedef __attribute__((ext_vector_type(16))) float cs_floatx16;
typedef __attribute__((ext_vector_type(2))) unsigned cs_uint2;
// v_mfma_f32_32x32x16_f16: D(32 x 32) += A(32 x 16) B(16 x 32).  Lane l holds A[l % 32][8 (l / 32) .. +7] and B[8 (l / 32) .. +7][l % 32];
// D element g * 4 + e of lane l is D[8 g + 4 (l / 32) + e][l % 32]
struct cs_mfma_slot { cs_half8 a, b; };
static inline cs_floatx16 cs_mfma(cs_half8 a, cs_half8 b, cs_floatx16 c, int, int, int) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    cs_mfma_slot* s = (cs_mfma_slot*)cs_xchg[wave];
    s[lane].a = a; s[lane].b = b;
    cs_wave_barrier(wave);
    const int col = lane & 31, hi = lane >> 5;
    for (int g = 0; g < 4; ++g)
        for (int e = 0; e < 4; ++e) {
            const int row = 8 * g + 4 * hi + e;
            float acc = c[g * 4 + e];
            for (int k = 0; k < 16; ++k) acc += (float)s[row + 32 * (k >> 3)].a[k & 7] * (float)s[col + 32 * (k >> 3)].b[k & 7];
            c[g * 4 + e] = acc;
        }
    cs_wave_barrier(wave);
    return c;
}
#define __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z) cs_mfma(a, b, c, x, y, z)
// v_permlane32_swap(a, b): the upper half of a and the lower half of b trade places -> lanes 0-31 get {a[l], a[l + 32]}, lanes 32-63 {b[l - 32], b[l]}
static inline cs_uint2 cs_permlane32_swap(unsigned a, unsigned b, bool, bool) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned* s = (unsigned*)cs_xchg[wave];
    s[lane] = a; s[64 + lane] = b;
    cs_wave_barrier(wave);
    cs_uint2 r;
    if (lane < 32) { r[0] = s[lane]; r[1] = s[lane + 32]; }
    else { r[0] = s[64 + lane - 32]; r[1] = s[64 + lane]; }
    cs_wave_barrier(wave);
    return r;
}
#define __builtin_amdgcn_permlane32_swap(a, b, x, y) cs_permlane32_swap(a, b, x, y)
