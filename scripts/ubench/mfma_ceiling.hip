// micro-benchmarks of the building blocks of k_conv_igemm on gfx950 (diagnostics, not part of the product):
//   mode 0: MFMA only (4 independent 32x32x16 f16 accumulators per wave)
//   mode 1: + one ds_read_b128 fragment per MFMA (same 1:1 ratio as the conv kernel), no barrier
//   mode 2: + one workgroup barrier per 16 MFMAs
//   mode 3: mode 2 with 128x128 tile sharing (4 waves read overlapping rows like the conv kernel)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float floatx16;

template <int MODE>
__global__ void __launch_bounds__(256, 2) k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += 256) ((float*)smem)[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    floatx16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    half8 wf[2], xf[2];
    for (int e = 0; e < 8; ++e) { wf[0][e] = (_Float16)0.01f; wf[1][e] = (_Float16)0.02f; xf[0][e] = (_Float16)0.03f; xf[1][e] = (_Float16)0.04f; }
    const int l31 = lane & 31, hi32 = lane >> 5, wn = wave >> 1, wm = wave & 1;
    for (int it = 0; it < iters; ++it) {
        const char* base = smem + (it & 1) * 32768;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            if (MODE >= 1) {
                const int q = ks * 2 + hi32;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = (MODE == 3 ? wn * 64 : 0) + i * 32 + l31;
                    wf[i] = *(const half8*)(base + 16384 + row * 128 + ((q ^ ((row >> 1) & 7)) << 4));
                    const int row2 = (MODE == 3 ? wm * 64 : 0) + i * 32 + l31;
                    xf[i] = *(const half8*)(base + row2 * 128 + ((q ^ ((row2 >> 1) & 7)) << 4));
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + tid] = s;
}

// mode 4/5/6: the full 2-stage structure of the conv kernel: every K-step each wave issues 8 x 1 KiB buffer->LDS DMAs for the
// next stage (source: 4 = one 32 KiB region per block that stays in L1/L2; 5 = a streaming window of a 256 MiB buffer; 6 = like 4
// but 4 DMAs per K-step (half the bytes)), vmcnt(0) + barrier, 16 ds_read_b128 + 16 MFMA.
template <int MODE>
__global__ void __launch_bounds__(256, 2) kd(float* out, const _Float16* src, unsigned src_bytes, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += 256) ((float*)smem)[i] = (float)(i & 7) * 0.001f;
    __syncthreads();
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
    floatx16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    half8 wf[2], xf[2];
    const int l31 = lane & 31, hi32 = lane >> 5, wn = wave >> 1, wm = wave & 1;
    constexpr int NL = (MODE == 6) ? 4 : 8;
    unsigned base_off = (MODE == 5) ? (blockIdx.x * 32768u) % (src_bytes - (1u << 22)) : (blockIdx.x & 63) * 32768u;
    for (int it = 0; it < iters; ++it) {
        const int cur = it & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        char* dst = smem + (cur ^ 1) * 32768;
        const unsigned o = base_off + ((MODE == 5) ? (unsigned)(it & 63) * 65536u : 0u);
#pragma unroll
        for (int l = 0; l < NL; ++l)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(dst + (l * 4 + wave) * 1024), 16,
                                                     (int)(o + (l * 4 + wave) * 1024 + lane * 16), 0, 0, 0);
        const char* base = smem + cur * 32768;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int q = ks * 2 + hi32;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int row = wn * 64 + i * 32 + l31;
                wf[i] = *(const half8*)(base + 16384 + row * 128 + ((q ^ ((row >> 1) & 7)) << 4));
                const int row2 = wm * 64 + i * 32 + l31;
                xf[i] = *(const half8*)(base + row2 * 128 + ((q ^ ((row2 >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc[i][j][e];
    out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void rund(const char* name, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 4000;
    float* out; hipMalloc(&out, blocks * 256 * 4);
    _Float16* src; const unsigned sb = 256u << 20; hipMalloc(&src, sb); hipMemset(src, 0, sb);
    hipFuncSetAttribute((const void*)kd<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kd<MODE><<<blocks, 256, 65536>>>(out, src, sb, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    kd<MODE><<<blocks, 256, 65536>>>(out, src, sb, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
    printf("%-44s %d blocks/CU: %7.1f TFLOP/s (%.3f ms)\n", name, blocks_per_cu, flops / ms / 1e9, ms);
    hipFree(out); hipFree(src);
}

template <int MODE>
void run(const char* name, int blocks_per_cu) {
    const int blocks = 256 * blocks_per_cu, iters = 4000;
    float* out; hipMalloc(&out, blocks * 256 * 4);
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256, 65536>>>(out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256, 65536>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
    printf("%-44s %d blocks/CU: %7.1f TFLOP/s (%.3f ms)\n", name, blocks_per_cu, flops / ms / 1e9, ms);
    hipFree(out);
}

// short-loop experiment: the same kernel as mode 4 but launched like the 80x80 256->256 conv layer: 800 blocks x 36 K-steps
void run_short(int blocks, int iters, int bpc_hint) {
    float* out; hipMalloc(&out, (size_t)blocks * 256 * 4);
    _Float16* src; const unsigned sb = 256u << 20; hipMalloc(&src, sb); hipMemset(src, 0, sb);
    hipFuncSetAttribute((const void*)kd<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kd<4><<<blocks, 256, 65536>>>(out, src, sb, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) kd<4><<<blocks, 256, 65536>>>(out, src, sb, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    const double flops = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
    printf("short loops: %5d blocks x %3d K-steps: %7.1f TFLOP/s (%.1f us per launch)\n", blocks, iters, flops / ms / 1e9, ms * 1e3);
    hipFree(out); hipFree(src);
}

int main() {
    run_short(800, 36, 2); run_short(1024, 36, 2); run_short(512, 36, 2); run_short(3200, 36, 2); run_short(800, 72, 2); run_short(800, 288, 2);
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<0>("MFMA only", bpc);
        run<1>("MFMA + 1 ds_read_b128 per MFMA", bpc);
        run<2>("  + barrier per 16 MFMAs", bpc);
        run<3>("  + shared 128x128 tile rows (conv pattern)", bpc);
        rund<4>("  + 32 KiB/step DMA, L1/L2-resident source", bpc);
        rund<6>("  + 16 KiB/step DMA, L1/L2-resident source", bpc);
        rund<5>("  + 32 KiB/step DMA, streaming 256 MiB", bpc);
    }
    return 0;
}
