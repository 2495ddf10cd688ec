// micro-benchmark: what the matrix pipe sustains as a function of OPERAND VALUES (diagnostics; not part of the product).
// A register-only loop of v_mfma_f32_32x32x16_f16 (4 independent accumulators per wave, 2 workgroups x 4 waves per CU, no memory
// traffic at all) timed for ~10 ms per case with (a) all-zero operands, (b) one constant, (c) uniformly random fp16 bit patterns
// of unit scale.  The instruction stream is identical; the difference is the power the multipliers draw and the clock the GPU
// can hold under its power limit.       hipcc --offload-arch=gfx950 -O3 -w mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float floatx16;

__global__ void __launch_bounds__(256, 2) k(float* out, int iters, int mode) {
    const int tid = threadIdx.x + blockIdx.x * 256;
    half8 a[2], b[2];
    unsigned h = tid * 2654435761u + 12345u;
    for (int i = 0; i < 2; ++i)
        for (int e = 0; e < 8; ++e) {
            h = h * 1664525u + 1013904223u; const float u = (float)(h >> 8) / 16777216.0f - 0.5f;
            h = h * 1664525u + 1013904223u; const float v = (float)(h >> 8) / 16777216.0f - 0.5f;
            a[i][e] = (_Float16)(mode == 0 ? 0.f : mode == 1 ? 0.5f : 2.f * u);
            b[i][e] = (_Float16)(mode == 0 ? 0.f : mode == 1 ? 0.5f : 2.f * v);
        }
    floatx16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        if (mode == 2) {   // keep the accumulators bounded without touching the MFMA count: flip the sign of one operand set
#pragma unroll
            for (int e = 0; e < 8; ++e) a[0][e] = -a[0][e], a[1][e] = -a[1][e];
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[tid] = s;
}

int main() {
    const int blocks = 512;
    float* out; hipMalloc(&out, blocks * 256 * 4);
    const char* names[3] = {"all-zero operands", "constant 0.5 operands", "random operands in [-1, 1)"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            k<<<blocks, 256>>>(out, 1000, mode);
            hipDeviceSynchronize();
            const int iters = 40000;   // ~10-20 ms
            hipEventRecord(e0);
            k<<<blocks, 256>>>(out, iters, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flops = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
            printf("%-28s %8.1f TFLOP/s  (%.2f ms)\n", names[mode], flops / ms / 1e9, ms);
        }
    return 0;
}
