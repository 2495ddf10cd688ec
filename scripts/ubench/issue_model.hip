// issue_model.hip -- what one wave can issue in the shadow of its own MFMAs on gfx950 (diagnostics, not part of the product).
// A workgroup of 256 (one wave per SIMD) or 512 threads (two), one workgroup per CU, runs a long loop of "slots": one v_mfma_f32_32x32x16_f16 on one of four
// independent accumulators + NV packed-fp32 multiplies + NT transcendentals (v_exp_f32) + NL ds_read_b128, every instruction an asm volatile statement on
// its own registers (no dependences inside a slot; each register is reused 4 slots later).  ACC selects where the accumulators live (0 = arch VGPRs, 1 = ACC
// registers); WA where the A operand lives.  Reported: ns per MFMA per wave, and the ratio to the bare-MFMA loop of the same launch shape.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float floatx16;
typedef __attribute__((ext_vector_type(2))) float f2;
typedef __attribute__((ext_vector_type(4))) unsigned u4;

template <int ACC, int WA, int NV, int NT, int NL, int THREADS>
__global__ void __launch_bounds__(THREADS, 1) k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += THREADS) ((float*)smem)[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    floatx16 c[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) c[j][e] = 0.f;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)0.01f; b[e] = (_Float16)0.02f; }
    f2 x[4][6], y = {1.0001f, 0.9999f};
    float t[4][4];
    u4 l[4][2];
    for (int j = 0; j < 4; ++j) { for (int i = 0; i < 6; ++i) x[j][i] = f2{1.f + i, 2.f + j}; for (int i = 0; i < 4; ++i) t[j][i] = 0.001f * (i + j); for (int i = 0; i < 2; ++i) l[j][i] = u4{0, 0, 0, 0}; }
    const unsigned addr = (unsigned)((tid & 63) * 16 + (tid >> 6) * 1024);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ACC == 0 && WA == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b));
            if (ACC == 0 && WA == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[j]) : "a"(a), "v"(b));
            if (ACC == 1 && WA == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[j]) : "v"(a), "v"(b));
            if (ACC == 1 && WA == 1) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[j]) : "a"(a), "v"(b));
#pragma unroll
            for (int i = 0; i < NL; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(l[j][i]) : "v"(addr), "n"(i * 4096));
#pragma unroll
            for (int i = 0; i < NT; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(t[j][i]));
#pragma unroll
            for (int i = 0; i < NV; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x[j][i]) : "v"(y));
        }
        if (NL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (once per 4 slots: the reads of this iteration are reused by the next)
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0.f;
    for (int j = 0; j < 4; ++j) {
        for (int e = 0; e < 16; ++e) s += c[j][e];
        for (int i = 0; i < 6; ++i) s += x[j][i][0] + x[j][i][1];
        for (int i = 0; i < 4; ++i) s += t[j][i];
        for (int i = 0; i < 2; ++i) s += (float)l[j][i][0];
    }
    out[blockIdx.x * THREADS + tid] = s;
}

static float* g_out;
static int g_cus;
template <int ACC, int WA, int NV, int NT, int NL, int THREADS>
static double run(const char* name, double base) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<ACC, WA, NV, NT, NL, THREADS>), dim3(g_cus), dim3(THREADS), 65536, 0, g_out, 200);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<ACC, WA, NV, NT, NL, THREADS>), dim3(g_cus), dim3(THREADS), 65536, 0, g_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / (iters * 4.0);          // per MFMA of one wave
    const double tf = 32768.0 * (THREADS / 64) * g_cus / ns / 1e3;
    printf("%-44s %7.2f ns/MFMA/wave  %7.1f TFLOP/s  x%.2f\n", name, ns, tf, base > 0 ? ns / base : 1.0);
    fflush(stdout);
    return ns;
}

#define RUN(ACC, WA, NV, NT, NL, TH, base) run<ACC, WA, NV, NT, NL, TH>("acc=" #ACC " wA=" #WA " pk=" #NV " trans=" #NT " lds=" #NL " threads=" #TH, base)
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    g_cus = p.multiProcessorCount;
    hipMalloc(&g_out, (size_t)g_cus * 512 * 4);
    printf("%s, %d CUs; one workgroup per CU\n", p.gcnArchName, g_cus);
    printf("-- one wave per SIMD, accumulators in arch VGPRs (acc=0) / ACC registers (acc=1), A operand in VGPR (wA=0) / ACC (wA=1)\n");
    const double b00 = RUN(0, 0, 0, 0, 0, 256, 0), b01 = RUN(0, 1, 0, 0, 0, 256, 0), b10 = RUN(1, 0, 0, 0, 0, 256, 0), b11 = RUN(1, 1, 0, 0, 0, 256, 0);
    (void)b11;
    printf("-- packed fp32 multiplies per MFMA\n");
    RUN(0, 1, 2, 0, 0, 256, b01); RUN(0, 1, 4, 0, 0, 256, b01); RUN(0, 1, 6, 0, 0, 256, b01);
    RUN(1, 0, 2, 0, 0, 256, b10); RUN(1, 0, 4, 0, 0, 256, b10); RUN(1, 0, 6, 0, 0, 256, b10);
    printf("-- transcendentals per MFMA\n");
    RUN(0, 1, 0, 1, 0, 256, b01); RUN(0, 1, 0, 2, 0, 256, b01); RUN(0, 1, 0, 4, 0, 256, b01);
    RUN(1, 0, 0, 1, 0, 256, b10); RUN(1, 0, 0, 2, 0, 256, b10); RUN(1, 0, 0, 4, 0, 256, b10);
    printf("-- ds_read_b128 per MFMA\n");
    RUN(0, 1, 0, 0, 1, 256, b01); RUN(0, 1, 0, 0, 2, 256, b01);
    RUN(1, 0, 0, 0, 1, 256, b10); RUN(1, 0, 0, 0, 2, 256, b10);
    printf("-- the slot of the weights-stationary kernel: 1 read + 1 transcendental + 1-2 packed\n");
    RUN(0, 1, 1, 1, 1, 256, b01); RUN(0, 1, 2, 1, 1, 256, b01);
    RUN(1, 0, 1, 1, 1, 256, b10); RUN(1, 0, 2, 1, 1, 256, b10); RUN(0, 0, 2, 1, 1, 256, b00);
    printf("-- two waves per SIMD (512 threads): per-wave ns, so x2.00 = the matrix pipe is shared perfectly\n");
    const double c01 = RUN(0, 1, 0, 0, 0, 512, 0), c10 = RUN(1, 0, 0, 0, 0, 512, 0);
    RUN(0, 1, 2, 1, 1, 512, c01); RUN(1, 0, 2, 1, 1, 512, c10); RUN(1, 0, 4, 2, 1, 512, c10); RUN(1, 0, 6, 0, 0, 512, c10); RUN(1, 0, 0, 4, 0, 512, c10);
    return 0;
}
