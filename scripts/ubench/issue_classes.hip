// issue_classes.hip -- which instruction classes a wave (or its SIMD neighbour) can run in the shadow of an MFMA on gfx950 (diagnostics, not part of the product).
// Like issue_model.hip: one workgroup per CU, 256 threads (one wave per SIMD) or 512 (two); a loop of slots = [one v_mfma_f32_32x32x16_f16 (MF = 1) or none] + N
// instructions of ONE class, all asm volatile, independent registers.  Reported: ns per slot per wave.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float floatx16;
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(4))) unsigned u4;
typedef __attribute__((ext_vector_type(2))) unsigned u2;

enum { PK_MUL, MUL, FMA, ADDU, MOV, CVTPK, EXP, RCP, SALU, NOP, LDS128, LDS64, PK_ADD, PK_MULH, ACCRD, PK_FMA, NKIND };
static const char* kname[] = {"v_pk_mul_f32", "v_mul_f32", "v_fma_f32", "v_add_u32", "v_mov_b32", "v_cvt_pk_f16_f32", "v_exp_f32", "v_rcp_f32", "s_add_u32", "s_nop 0",
                              "ds_read_b128", "ds_read_b64", "v_pk_add_f32", "v_pk_mul_f16", "v_accvgpr_read", "v_pk_fma_f32"};

template <int KIND>
__device__ __forceinline__ void op(f2& x, const f2& y, u4& l, unsigned addr, unsigned& sreg, float& acc_src) {
    if (KIND == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (KIND == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(x) : "v"(y));
    if (KIND == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
    if (KIND == PK_MULH) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
    if (KIND == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
    if (KIND == FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x[0]) : "v"(y[0]));
    if (KIND == ADDU) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[0]) : "v"(y[0]));
    if (KIND == MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(x[0]) : "v"(y[0]));
    if (KIND == CVTPK) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(x[0]) : "v"(y[0]), "v"(y[1]));
    if (KIND == EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(x[0]));
    if (KIND == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(x[0]));
    if (KIND == SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sreg));
    if (KIND == NOP) asm volatile("s_nop 0");
    if (KIND == LDS128) asm volatile("ds_read_b128 %0, %1" : "=v"(l) : "v"(addr));
    if (KIND == LDS64) asm volatile("ds_read_b64 %0, %1" : "=v"(x) : "v"(addr));
    if (KIND == ACCRD) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(x[0]) : "a"(acc_src));
}

template <int MF, int KIND, int N, int THREADS>
__global__ void __launch_bounds__(THREADS, 1) k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += THREADS) ((float*)smem)[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    floatx16 c[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) c[j][e] = 0.f;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)0.01f; b[e] = (_Float16)0.02f; }
    f2 x[4][8], y = {1.0001f, 0.9999f};
    u4 l[4][8];
    unsigned sreg = 0;
    float accsrc = 1.f;
    for (int j = 0; j < 4; ++j) for (int i = 0; i < 8; ++i) { x[j][i] = f2{1.f + i, 2.f + j}; l[j][i] = u4{0, 0, 0, 0}; }
    const unsigned addr = (unsigned)((tid & 63) * 16 + (tid >> 6) * 1024);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (MF) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[j]) : "v"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < N; ++i) op<KIND>(x[j][i], y, l[j][i], addr, sreg, accsrc);
        }
        if (KIND == LDS128 || KIND == LDS64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    float s = (float)sreg;
    for (int j = 0; j < 4; ++j) {
        for (int e = 0; e < 16; ++e) s += c[j][e];
        for (int i = 0; i < 8; ++i) s += x[j][i][0] + x[j][i][1] + (float)l[j][i][0];
    }
    out[blockIdx.x * THREADS + tid] = s;
}

static float* g_out;
static int g_cus;
template <int MF, int KIND, int N, int THREADS>
static double run() {
    const int iters = 10000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MF, KIND, N, THREADS>), dim3(g_cus), dim3(THREADS), 65536, 0, g_out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MF, KIND, N, THREADS>), dim3(g_cus), dim3(THREADS), 65536, 0, g_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / (iters * 4.0);
}
template <int KIND>
static void row() {
    // ns per slot per wave: without MFMA (N = 4, 8: the class's own issue cost), with one MFMA per slot (N = 1, 2, 4, 8), and with two waves per SIMD (N = 2, 4)
    const double v4 = run<0, KIND, 4, 256>(), v8 = run<0, KIND, 8, 256>();
    const double m1 = run<1, KIND, 1, 256>(), m2 = run<1, KIND, 2, 256>(), m4 = run<1, KIND, 4, 256>(), m8 = run<1, KIND, 8, 256>();
    const double w2 = run<1, KIND, 2, 512>(), w4 = run<1, KIND, 4, 512>();
    printf("%-18s | alone: %5.2f ns/instr | +MFMA, 1 wave/SIMD: N=1 %6.2f  N=2 %6.2f  N=4 %6.2f  N=8 %6.2f | 2 waves/SIMD (per wave): N=2 %6.2f  N=4 %6.2f\n", kname[KIND], (v8 - v4) / 4.0,
           m1, m2, m4, m8, w2, w4);
    fflush(stdout);
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    g_cus = p.multiProcessorCount;
    hipMalloc(&g_out, (size_t)g_cus * 512 * 4);
    printf("%s, %d CUs; ns per slot per wave; bare MFMA slot: 1 wave/SIMD %.2f ns, 2 waves/SIMD %.2f ns per wave\n", p.gcnArchName, g_cus, run<1, NOP, 0, 256>(), run<1, NOP, 0, 512>());
    row<PK_MUL>(); row<PK_ADD>(); row<PK_FMA>(); row<PK_MULH>(); row<MUL>(); row<FMA>(); row<ADDU>(); row<MOV>(); row<CVTPK>(); row<EXP>(); row<RCP>(); row<SALU>(); row<NOP>();
    row<LDS128>(); row<LDS64>(); row<ACCRD>();
    return 0;
}
