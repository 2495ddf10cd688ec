// issue_vmem.hip -- what a vector-memory instruction costs a wave that is busy with MFMAs on gfx950 (diagnostics, not part of the product).
// One workgroup per CU, 256 threads (one wave per SIMD) or 512; a loop of 16 MFMA slots (v_mfma_f32_32x32x16_f16, 4 accumulators) with ONE memory instruction
// after slot 0 (and optionally after slot 8: PER16 = 2), all asm volatile.  Kinds:
//   0 none   1 global_store_dwordx4, 64 lanes x 16 B contiguous (8 full 128-B lines)   2 the same, 32 B per 128-B line (32 lines touched: the ws64 epilogue's pattern)
//   3 global_store_dwordx4, one line per lane (64 lines)   4 buffer_load_dwordx4 ... lds (1 KiB contiguous)   5 global_load_dwordx4 contiguous (+ s_waitcnt vmcnt(2))
//   6 global_store_dwordx2 32 B per line   7 global_store_dword 4 B x 64 contiguous
// Every wave works on its own few KiB (cache resident): the cost measured is the instruction's, not DRAM's.  Reported: ns per 16-slot iteration per wave and the
// difference to kind 0 = ns per memory instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(16))) float floatx16;
typedef __attribute__((ext_vector_type(4))) unsigned u4;
typedef __attribute__((ext_vector_type(2))) unsigned u2;
#define LDS_AS __attribute__((address_space(3)))

template <int KIND, int PER16, int THREADS>
__global__ void __launch_bounds__(THREADS, 1) k(char* buf, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    floatx16 c[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) c[j][e] = 0.f;
    half8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)0.01f; b[e] = (_Float16)0.02f; }
    char* base = buf + ((size_t)blockIdx.x * 8 + wave) * 16384;
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 16384, 0x00020000);
    u4 d = {1u, 2u, 3u, 4u}, ld[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    const unsigned off_contig = lane * 16, off_32 = (lane & 31) * 128 + (lane >> 5) * 16, off_line = lane * 128;
    char* p_contig = base + off_contig; char* p_32 = base + off_32; char* p_line = base + off_line; char* p_d2 = base + (lane & 15) * 128 + (lane >> 4) * 8; char* p_d1 = base + lane * 4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int sl = 0; sl < 16; ++sl) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c[sl & 3]) : "v"(a), "v"(b));
            if (sl == 0 || (PER16 == 2 && sl == 8)) {
                if (KIND == 1) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p_contig), "v"(d) : "memory");
                if (KIND == 2) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p_32), "v"(d) : "memory");
                if (KIND == 3) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p_line), "v"(d) : "memory");
                if (KIND == 4) __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (LDS_AS void*)(smem + wave * 1024), 16, off_contig, 0, 0, 0);
                if (KIND == 5) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(ld[sl >> 3]) : "v"(p_contig) : "memory"); asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }
                if (KIND == 6) asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p_d2), "v"(u2{d[0], d[1]}) : "memory");
                if (KIND == 7) asm volatile("global_store_dword %0, %1, off" ::"v"(p_d1), "v"(d[0]) : "memory");
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    float s = (float)(ld[0][0] + ld[1][0]);
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += c[j][e];
    out[blockIdx.x * THREADS + tid] = s;
}

static char* g_buf; static float* g_out; static int g_cus;
template <int KIND, int PER16, int THREADS>
static double run() {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<KIND, PER16, THREADS>), dim3(g_cus), dim3(THREADS), 16384, 0, g_buf, g_out, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, PER16, THREADS>), dim3(g_cus), dim3(THREADS), 16384, 0, g_buf, g_out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6 / iters;
}
template <int KIND>
static void row(const char* name, double b1, double b2) {
    const double a1 = run<KIND, 1, 256>(), a2 = run<KIND, 2, 256>(), w1 = run<KIND, 1, 512>(), w2 = run<KIND, 2, 512>();
    printf("%-58s | 1 wave/SIMD: 1 per 16 MFMAs +%6.1f ns, 2 per 16: +%6.1f ns each | 2 waves/SIMD: +%6.1f, +%6.1f ns each (per wave)\n", name, a1 - b1, (a2 - b1) / 2, w1 - b2, (w2 - b2) / 2);
    fflush(stdout);
}
int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    g_cus = p.multiProcessorCount;
    hipMalloc(&g_buf, (size_t)g_cus * 8 * 16384);
    hipMalloc(&g_out, (size_t)g_cus * 512 * 4);
    const double b1 = run<0, 1, 256>(), b2 = run<0, 1, 512>();
    printf("%s, %d CUs; 16 MFMA slots: %.1f ns (1 wave/SIMD), %.1f ns per wave (2 waves/SIMD)\n", p.gcnArchName, g_cus, b1, b2);
    row<1>("global_store_dwordx4, 1 KiB contiguous (8 lines)", b1, b2);
    row<2>("global_store_dwordx4, 32 B per line (32 lines)", b1, b2);
    row<3>("global_store_dwordx4, one line per lane (64 lines)", b1, b2);
    row<6>("global_store_dwordx2, 32 B per line (16 lines)", b1, b2);
    row<7>("global_store_dword, 256 B contiguous (2 lines)", b1, b2);
    row<4>("buffer_load_dwordx4 ... lds, 1 KiB contiguous", b1, b2);
    row<5>("global_load_dwordx4, 1 KiB contiguous (+ vmcnt(2))", b1, b2);
    return 0;
}
