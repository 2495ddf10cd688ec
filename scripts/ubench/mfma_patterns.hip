// mfma_patterns.hip -- does the ORDER in which operands meet in the matrix pipe change what it sustains under the power limit? (diagnostics, not part of the product)
// Register-only loops of v_mfma_f32_32x32x16_f16 on random fp16 operands (8 A fragments, 8 B fragments, 8 accumulators per wave; 2 workgroups x 4 waves per CU, ~40 ms
// per case), the same instruction count in every pattern:
//   0: one A, one B fragment for every MFMA (nothing toggles on the operand side)        1: A fixed, B cycles through 4 fragments
//   2: A and B both change with every MFMA (4 x 4 visited diagonally)                      3: A changes every 4th MFMA, B cycles through 4 (the weights-stationary kernel)
//   4: 2 A x 4 B, A inner (A toggles every MFMA, B every 2nd)                              5: 2 A x 4 B, B inner (the patch kernels: A every 4th, B every MFMA)
//   6: all-zero operands (reference: the clock without operand power)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float floatx16;

template <int PAT>
__global__ void __launch_bounds__(256, 2) k(float* out, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    half8 a[4], b[4];
    unsigned h = tid * 2654435761u + 12345u;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            h = h * 1664525u + 1013904223u; const float u = (float)(h >> 8) / 16777216.0f - 0.5f;
            h = h * 1664525u + 1013904223u; const float v = (float)(h >> 8) / 16777216.0f - 0.5f;
            a[i][e] = (_Float16)(PAT == 6 ? 0.f : 2.f * u);
            b[i][e] = (_Float16)(PAT == 6 ? 0.f : 2.f * v);
        }
    floatx16 acc[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#define MF(A, B, C) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[C]) : "v"(a[A]), "v"(b[B]))
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) {          // 16 MFMAs per iteration; accumulator j (4 independent chains)
                if (PAT == 0 || PAT == 6) MF(0, 0, j);
                if (PAT == 1) MF(0, j, j);
                if (PAT == 2) MF((r + j) & 3, j, j);
                if (PAT == 3) MF(r, j, j);
                if (PAT == 4) MF(j & 1, (r * 2 + (j >> 1)) & 3, j);
                if (PAT == 5) MF(r & 1, j, j);
            }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[tid] = s;
}

template <int PAT>
static void run(float* out, int cus, const char* what) {
    const int iters = 60000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<PAT>, dim3(cus * 2), dim3(256), 0, 0, out, 2000);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<PAT>, dim3(cus * 2), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-78s %8.1f TFLOP/s  (%.1f ms)\n", what, 32768.0 * 16 * iters * 8.0 * cus / ms / 1e9, ms);
    fflush(stdout);
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    float* out;
    hipMalloc(&out, (size_t)p.multiProcessorCount * 512 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run<6>(out, p.multiProcessorCount, "6: all-zero operands");
        run<0>(out, p.multiProcessorCount, "0: random operands, the same A and B fragment in every MFMA");
        run<1>(out, p.multiProcessorCount, "1: A fixed, B cycles through 4 fragments");
        run<3>(out, p.multiProcessorCount, "3: A changes every 4th MFMA, B cycles through 4 (weights-stationary kernel)");
        run<5>(out, p.multiProcessorCount, "5: 2 A x 4 B, B inner (patch kernels)");
        run<4>(out, p.multiProcessorCount, "4: 2 A x 4 B, A inner");
        run<2>(out, p.multiProcessorCount, "2: A and B both change with every MFMA");
    }
    return 0;
}
