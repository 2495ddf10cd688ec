// micro-benchmark: cost of one wave-wide `buffer_load_dwordx4 ... lds` (1 KiB into LDS) as a function of its SOURCE address pattern
// (diagnostics for the conv kernels' DMA streams; not part of the product).  2 workgroups x 4 waves per CU, every wave keeps 8 DMAs
// in flight.   hipcc --offload-arch=gfx950 -O3 dma_patterns.hip -o dma_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
#define LDS_AS __attribute__((address_space(3)))

template <int MODE>
__global__ void __launch_bounds__(256, 2) k(const char* src, unsigned src_bytes, int iters, int stride, float* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, src_bytes, 0x00020000);
    unsigned voff;
    const unsigned blk = (blockIdx.x % 64) * (1u << 20);      // 64 MiB of distinct 1 MiB regions: L2-resident after the first pass
    if (MODE == 0) voff = blk + wave * 1024 + lane * 16;                                       // contiguous 1 KiB
    else if (MODE == 1 || MODE == 2) voff = blk + (wave * 16 + (lane >> 2)) * stride + (lane & 3) * 16;   // 16 segments of 64 B at `stride`
    else if (MODE == 3) {                                                                      // 80-byte LDS pitch: 12.8 pixels, pad lane = OOB
        const int byte = lane * 16, px = byte / 80, cs = (byte - px * 80) >> 4;
        voff = cs < 4 ? blk + (wave * 13 + px) * stride + cs * 16 : 0xFF000000u;
    } else if (MODE == 4) voff = 0xFF000000u;                                                  // all out of range (zero fill)
    else voff = blk + (wave * 64 + lane) * stride;                                             // 5: every lane its own line
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (LDS_AS void*)(smem + (wave * 8 + q) * 1024), 16, voff, (it & 15) * 65536 + q * 128, 0, 0);
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = ((float*)smem)[threadIdx.x];
}

template <int MODE>
void run(const char* name, int stride) {
    const int blocks = 512, iters = 2000;
    float* out; hipMalloc(&out, blocks * 256 * 4);
    char* src; const unsigned sb = 128u << 20; hipMalloc(&src, sb); hipMemset(src, 1, sb);
    hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256, 32768>>>(src, sb, 50, stride, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<blocks, 256, 32768>>>(src, sb, iters, stride, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)blocks * 4 * iters * 8;     // wave-DMAs
    printf("%-58s %7.2f ns per wave-DMA per CU  (%6.1f clk @2.4GHz, %6.2f TB/s chip)\n", name, ms * 1e6 / (n / 256), ms * 1e6 / (n / 256) * 2.4,
           n * 1024 / ms / 1e9);
    hipFree(out); hipFree(src);
}

int main() {
    run<0>("contiguous 1 KiB", 0);
    run<1>("16 x 64 B segments, stride 4608 (weight rows)", 4608);
    run<2>("16 x 64 B segments, stride 512 (pixel rows, ld 256)", 512);
    run<2>("16 x 64 B segments, stride 2048 (pixel rows, ld 1024)", 2048);
    run<2>("16 x 64 B segments, stride 128 (both halves of a line)", 128);
    run<3>("12.8 x 64 B pixels at 80-byte LDS pitch, stride 512", 512);
    run<4>("all lanes out of range (zero fill)", 0);
    run<5>("64 x 16 B, one line per lane, stride 512", 512);
    return 0;
}
