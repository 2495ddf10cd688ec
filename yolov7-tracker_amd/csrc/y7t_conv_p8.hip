// y7t_conv_p8.hip -- 1x1 / stride 1 Conv(+folded BN)+bias+activation as a 256 x 256 x 64 GEMM tile on an 8-wave ping-pong pipeline (two waves per SIMD).
//
// Same math as k_conv_igemm (y7t_conv.hip; /root/reference/models/common.py:99-111 after utils/torch_utils.py:181-201):
//      D[n][m] = sum_k W[n][k] X[m][k],   n = output channel, m = output pixel, k = input channel,
// for the 1x1 layers of yolov7-w6 with Cout a multiple of 256 on the 160^2 / 80^2 / 40^2 maps (/root/reference/cfg/deploy/yolov7-w6.yaml: the twin 1x1 and
// the closing 1x1 of every ELAN block, the lateral 1x1 convs of the head, the three consumers of a nearest-x2 upsample): 14 launches, 2.4 ms of the 15.3 ms
// launch list on k_conv_igemm<128,128,32,2> (profiles/r03_conv_per_layer_b32.txt), which runs at 600-775 TFLOP/s there.
//
// Why another kernel.  k_conv_igemm is a 128 x 128 tile with ONE barrier and a full `s_waitcnt vmcnt(0)` per 32-deep K-step: its DMA-only ablation runs in 85-90 % of
// the full time whatever the ring depth (profiles/r02_vendor_gemm_probe.txt) -- the structure, not the MFMA pipe, is the bound, and 128 x 128 tiles pull 64 B through
// the vector-memory path per kFLOP.  This kernel is the other structure (cdna_hip_programming.md section 5, "the 256^2 8-phase template"):
//   * 256 x 256 tile, K-step 64: half the LDS-fill bytes per flop.  512 threads = 8 waves = two wave GROUPS of four (one wave of each group per SIMD).
//   * each K-tile is staged as FOUR half-tiles of 16 KiB (pixel halves P0 P1, channel halves C0 C1; 128 rows x 128 bytes, XOR-swizzled like k_conv_igemm's rows), two
//     buffers of them (128 KiB), by `buffer_load_dwordx4 ... lds` -- two 1 KiB pieces per wave and half-tile.
//   * a K-tile is FOUR phases, one per quadrant of the wave's 128-pixel x 64-channel output: (P0,C0) (P0,C1) (P1,C1) (P1,C0).  The wave's rows are spread over BOTH
//     halves of each operand (64 pixels of P0 + 64 of P1, 32 channels of C0 + 32 of C1), so every half-tile is read in ONE phase only (C0 stays in registers for the
//     fourth) and is free for the next-but-one K-tile right after it: P0, C0 after phase 0, C1 after phase 1, P1 after phase 2.
//   * every phase issues ONE half-tile of prefetch (phase 0: C1 of K-tile t+1, 1: P1 of t+1, 2: P0 of t+2, 3: C0 of t+2) and then waits `vmcnt(8)`: four half-tiles stay in
//     flight across the barriers, none is ever drained; a half-tile is staged >= 5 phases (~1.2 us of MFMA time) before its first read and >= 2 phases after its last.
//   * phase = [fragment reads + the prefetch + vmcnt(8)] barrier [8 x v_mfma_f32_32x32x16_f16 at s_setprio 1] barrier.  Group 1 runs ONE barrier behind group 0, so on
//     every SIMD one wave is in its MFMA block while its partner reads fragments and issues DMAs: the matrix pipe of a SIMD always has exactly one wave feeding it.
//   * past the last K-tile the schedule keeps issuing (out-of-range: the hardware writes zeros into half-tiles nobody reads any more), so the wait counts stay uniform.
//   * bias: fetched to LDS by the first DMA of the kernel, the accumulators START at the bias (the epilogue adds nothing); epilogue: activation, fp16, v_permlane32_swap
//     into 16-byte pieces, each wave transposes ITS 128 pixels x 64 channels through its own 16 KiB of the (now idle) ring and stores full 128-byte lines.
//   * upsample-on-read (DUAL): K-tiles whose channels lie in [up_c0, up_c0 + up_C) fetch pixel (y, x) from the half-resolution tensor `in2` at (y >> 1, x >> 1).
// Weights: korder 7 (detector/weights.py::panel_pack_p8): per (channel tile, K-tile) one contiguous 32 KiB block that IS the swizzled LDS image of C0 then C1;
// LDS row r of half h holds channel n0 + (r / 32) * 64 + h * 32 + r % 32, so a wave's 64 channels are CONTIGUOUS in the output (one 128-byte line per pixel).
// Pixels: LDS row r of half h holds pixel m0 + (r / 64) * 128 + h * 64 + r % 64 (a group's 128 pixels are contiguous).
// One workgroup per output tile and per CU (129 KiB of LDS): the hardware dispatcher balances the tiles over the CUs the tracker / NMS streams leave free (a statically
// partitioned persistent kernel lost 0.4 ms to that, profiles/r04_ws128_measurement.txt).
#include "y7t_common.h"
#include "y7t_conv_common.h"
#include <stdlib.h>

namespace {

#if defined(Y7T_CONVSIM)      // tests/_convsim: the DMA queue of the host model (fake/hip/hip_runtime.h); lanes of a wave are separate OS threads there
#define P8_VMCNT(n) cs_vmcnt(n)
#define P8_WAVE_SYNC() cs_wave_barrier(threadIdx.x >> 6)
#define P8_KEEP(a, b) ((void)0)
#else
#define P8_VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")
#define P8_WAVE_SYNC() ((void)0)      // LDS instructions of one wave execute in order
#define P8_KEEP(a, b) asm volatile("" :: "v"(a), "v"(b))      // (timing ablations: keeps the fragment reads alive without the MFMAs)
#endif

struct P8 {
    static constexpr int BM = 256, BN = 256, BK = 64, NT = 512;
    static constexpr int ROWB = 128;              // bytes per LDS row: 64 fp16
    static constexpr int HALF = 128 * ROWB;       // a half-tile: 128 rows, 16 KiB
    static constexpr int BUF = 4 * HALF;          // P0 P1 C0 C1 of one K-tile
    static constexpr int RING = 2 * BUF;          // 128 KiB
    static constexpr int BIAS = RING;             // BN floats behind the ring
    static constexpr int LDS = RING + BN * 4;
    static constexpr unsigned OOB = 0xFF000000u;  // voffset of a zero-filled lane: out of range with or without the (< 16 MiB) scalar offset
};

// (a variant with the two DMA pieces of a phase issued INSIDE the MFMA block and `vmcnt(6)` at its end measured the same: profiles/r04_p8_measurements.txt)
// ABL (timing ablations, wrong results; Y7T_CONV_ABLATE): 1 no prefetch DMAs in the loop, 2 no MFMAs, 4 no fragment reads, 8 no s_setprio, 16 no output stores
template <int ACT, bool DUAL, int ABL = 0>
__global__ void __launch_bounds__(512, 2) k_conv1x1_p8(const Y7TConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C = P8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wq = wave & 3, l31 = lane & 31, hi32 = lane >> 5;

    // XCD-aware order (as k_conv_igemm): workgroup b runs on XCD b % 8, every XCD gets a contiguous range of tiles; channel tiles fastest inside it
    int bid = blockIdx.x;
    if (p.xcd_swizzle) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int n_tiles_n = p.Cout_pad / C::BN;
    const int tile_n = bid % n_tiles_n, tile_m = bid / n_tiles_n;
    const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;
    const int nk = p.Cin / C::BK;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, (unsigned)p.Cout_pad * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr2 = __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? p.in2 : p.in), 0, DUAL ? p.in2_bytes : p.in_bytes, 0x00020000);

    // ---- staging geometry: one DMA instruction of the workgroup fills 64 rows (wave w: rows 8 w .. 8 w + 7, lane l: row l / 8, 16-byte slot l % 8); a half-tile is two
    //      of them.  LDS slot s of row r holds source chunk s ^ ((r >> 1) & 7)  (rows 64 apart have the same swizzle) ----
    const int srow = wave * 8 + (lane >> 3);
    const int gch = (lane & 7) ^ ((srow >> 1) & 7);
    int xoff[2][2], xoff2[DUAL ? 2 : 1][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int rd = 0; rd < 2; ++rd) {
            const int m = m0 + rd * 128 + h * 64 + srow;      // row rd * 64 + srow of pixel half h
            xoff[h][rd] = (int)C::OOB;
            if (DUAL) xoff2[h][rd] = (int)C::OOB;
            if (m < p.M) {
                xoff[h][rd] = (m * p.ldin + p.cin_off + gch * 8) * 2;
                if (DUAL) {
                    const int HW = p.H * p.W, b = m / HW, rem = m - b * HW, y = rem / p.W, x = rem - y * p.W;
                    xoff2[h][rd] = (((b * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (x >> 1)) * p.ldin2 + p.cin2_off + gch * 8) * 2;
                }
            }
        }
    const int wvo = tile_n * nk * (2 * C::HALF) + wave * 1024 + lane * 16;      // this lane's 16 bytes of piece (wave) of round 0 of C0 of K-tile 0

    // piece rd (0 / 1: rows 0-63 / 64-127) of pixel half h of K-tile T -> buffer b   (T >= nk: zeros into a half nobody reads)
    bool in_loop = false;      // (ABL 1: the prologue's DMAs stay)
    auto stage_p1 = [&](int h, int b, int T, int rd) __attribute__((always_inline)) {
        if ((ABL & 1) && in_loop) return;
        char* dst = smem + b * C::BUF + h * C::HALF + wave * 1024 + rd * 8192;
        const bool live = T < nk;
        const int ci = T * C::BK;
        const bool up = DUAL && live && ci >= p.up_c0 && ci < p.up_c0 + p.up_C;    // wave-uniform: a K-tile lies in one source (up_c0, up_C multiples of 64)
        if (up) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr2, (LDS_AS void*)dst, 16, xoff2[DUAL ? h : 0][rd], (ci - p.up_c0) * 2, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (LDS_AS void*)dst, 16, live ? xoff[h][rd] : (int)C::OOB, live ? ci * 2 : 0, 0, 0);
    };
    auto stage_c1 = [&](int h, int b, int T, int rd) __attribute__((always_inline)) {      // ... of channel half h
        if ((ABL & 1) && in_loop) return;
        char* dst = smem + b * C::BUF + (2 + h) * C::HALF + wave * 1024 + rd * 8192;
        const bool live = T < nk;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (LDS_AS void*)dst, 16, live ? wvo + h * C::HALF + rd * 8192 : (int)C::OOB, live ? T * (2 * C::HALF) : 0, 0, 0);
    };
    auto stage_p = [&](int h, int b, int T) __attribute__((always_inline)) { stage_p1(h, b, T, 0); stage_p1(h, b, T, 1); };
    auto stage_c = [&](int h, int b, int T) __attribute__((always_inline)) { stage_c1(h, b, T, 0); stage_c1(h, b, T, 1); };

    // ---- fragment geometry: lane (l31, hi32) reads row base + l31, logical chunk 2 ks + hi32 of k-substep ks ----
    const int swl = (l31 >> 1) & 7;
    const int prow = (grp * 64 + l31) * C::ROWB, crow = (wq * 32 + l31) * C::ROWB;
    half8 pf[2][4], c0[4], c1[4];
    auto read_p = [&](int hP, int b) __attribute__((always_inline)) {
        if (ABL & 4) return;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) pf[j][ks] = *(const half8*)(smem + b * C::BUF + hP * C::HALF + prow + j * 32 * C::ROWB + (((ks * 2 + hi32) ^ swl) << 4));
    };
    auto read_c = [&](half8 (&cf)[4], int hC, int b) __attribute__((always_inline)) {
        if (ABL & 4) return;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) cf[ks] = *(const half8*)(smem + b * C::BUF + (2 + hC) * C::HALF + crow + (((ks * 2 + hi32) ^ swl) << 4));
    };
    floatx16 acc[2][2][2];      // [pixel half][32-pixel block][channel half]
    auto mma = [&](int hP, const half8 (&cf)[4], int hC) __attribute__((always_inline)) {      // a phase's MFMA block
        if (!(ABL & 8)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (ABL & 2) P8_KEEP(cf[ks], pf[j][ks]);
                else acc[hP][j][hC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cf[ks], pf[j][ks], acc[hP][j][hC], 0, 0, 0);
            }
        if (!(ABL & 8)) __builtin_amdgcn_s_setprio(0);
    };

    // ---- prologue: the bias, K-tile 0 and the first two halves of K-tile 1 (what phases (-1, 2) and (-1, 3) of the steady state would have staged) ----
    __builtin_amdgcn_raw_ptr_buffer_load_lds(br, (LDS_AS void*)(smem + C::BIAS + wq * 256), 4, (n0 + wq * 64 + lane) * 4, 0, 0, 0);      // (both groups: same bytes)
    stage_p(0, 0, 0); stage_c(0, 0, 0); stage_c(1, 0, 0); stage_p(1, 0, 0); stage_p(0, 1, 1); stage_c(0, 1, 1);
    P8_VMCNT(8);                          // 13 issued: the bias, P0 and C0 of K-tile 0 have landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();         // ... everybody's
#pragma unroll
    for (int hC = 0; hC < 2; ++hC)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            typedef __attribute__((ext_vector_type(4))) float float4v;
            const float4v bv = *(const float4v*)(smem + C::BIAS + (wq * 64 + hC * 32 + 8 * g + 4 * hi32) * 4);
#pragma unroll
            for (int hP = 0; hP < 2; ++hP)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[hP][j][hC][g * 4 + e] = bv[e];
        }
    if (grp == 1) __builtin_amdgcn_s_barrier();      // group 1 runs one barrier behind: its fragment reads / DMAs fall into group 0's MFMA blocks and vice versa
    in_loop = true;
    if (ABL & 4) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { c0[ks] = c1[ks] = pf[0][ks] = pf[1][ks] = half8{1, 1, 1, 1, 1, 1, 1, 1}; }
    }

    // ---- one K-tile = four phases; b (its buffer) is a compile-time constant of each instance ----
    auto ktile = [&](int T, auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
        // phase 0: quadrant (P0, C0); prefetch C1 of K-tile T + 1 (that half of the other buffer was last read in phase 1 of K-tile T - 1)
        read_p(0, b); read_c(c0, 0, b);
        stage_c(1, b ^ 1, T + 1);
        P8_VMCNT(8);
        __builtin_amdgcn_s_barrier();
        mma(0, c0, 0);
        __builtin_amdgcn_s_barrier();
        // phase 1: (P0, C1); prefetch P1 of T + 1
        read_c(c1, 1, b);
        stage_p(1, b ^ 1, T + 1);
        P8_VMCNT(8);
        __builtin_amdgcn_s_barrier();
        mma(0, c1, 1);
        __builtin_amdgcn_s_barrier();
        // phase 2: (P1, C1); prefetch P0 of T + 2 into THIS buffer (its P0 was read in phase 0)
        read_p(1, b);
        stage_p(0, b, T + 2);
        P8_VMCNT(8);
        __builtin_amdgcn_s_barrier();
        mma(1, c1, 1);
        __builtin_amdgcn_s_barrier();
        // phase 3: (P1, C0) with the C0 fragments of phase 0; prefetch C0 of T + 2
        stage_c(0, b, T + 2);
        P8_VMCNT(8);
        __builtin_amdgcn_s_barrier();
        mma(1, c0, 0);
        __builtin_amdgcn_s_barrier();
    };
    for (int T = 0; T < nk; T += 2) {
        ktile(T, std::integral_constant<int, 0>{});
        if (T + 1 < nk) ktile(T + 1, std::integral_constant<int, 1>{});
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();      // re-align the groups
    P8_VMCNT(0);                                     // the tail's zero-fills have landed too ...
    __builtin_amdgcn_s_barrier();                    // ... everybody's, and every wave is past its last fragment read: the ring is scratch now

    // ---- epilogue: activation, fp16; the wave's 128 pixels x 64 channels through its own 16 KiB (row = pixel, 128 bytes, slot = chunk ^ (pixel & 7)), out as full lines ----
    char* scr = smem + wave * 16384;
    typedef __attribute__((ext_vector_type(4))) unsigned uint4v;
#pragma unroll
    for (int hP = 0; hP < 2; ++hP)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pl = hP * 64 + j * 32 + l31;
#pragma unroll
            for (int hC = 0; hC < 2; ++hC)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    unsigned w[2][2];
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const int g = gp * 2 + gg;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = act_t<ACT>(acc[hP][j][hC][g * 4 + e]);
                        typedef __attribute__((ext_vector_type(2))) _Float16 half2v;
                        half2v h0 = {(half_t)v[0], (half_t)v[1]}, h1 = {(half_t)v[2], (half_t)v[3]};
                        w[gg][0] = __builtin_bit_cast(unsigned, h0);
                        w[gg][1] = __builtin_bit_cast(unsigned, h1);
                    }
                    // v_permlane32_swap: lanes 0-31 end up with channels 16 gp .. + 7, lanes 32-63 with 16 gp + 8 .. + 15 of their pixel (as k_conv_igemm's epilogue)
                    auto r0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                    uint4v pk = {r0[0], r1[0], r0[1], r1[1]};
                    const int ch = hC * 4 + gp * 2 + hi32;
                    *(uint4v*)(scr + pl * 128 + ((ch ^ (pl & 7)) << 4)) = pk;
                }
        }
    P8_WAVE_SYNC();
    half_t* outp = (half_t*)p.out;
    // 16 pieces of 16 bytes per lane, eight at a time: the LDS reads of a group first, then its stores (left alone the compiler waited for every read in front of its
    // store -- and with one workgroup per CU nothing else runs meanwhile; round 4, as y7t_conv_patch.hip)
#pragma unroll 1
    for (int k0 = 0; k0 < 16; k0 += 8) {
        uint4v v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int px = (k0 + k) * 8 + (lane >> 3), ch = lane & 7;
            v[k] = *(const uint4v*)(scr + px * 128 + ((ch ^ (px & 7)) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int px = (k0 + k) * 8 + (lane >> 3), ch = lane & 7;
            const int m = m0 + grp * 128 + px, n = n0 + wq * 64 + ch * 8;
            if (m < p.M && n < p.Cout && !(ABL & 16)) *(uint4v*)(outp + (size_t)m * p.ldout + p.cout_off + n) = v[k];
        }
    }
#endif
}


// ---- round 6: the PERSISTENT form (profiles/r06_batch_80.txt: p8 is the list's largest family at 80 frames, and a tile's fixed cost -- workgroup launch, the prologue's
// first DMA round trip, drain -- is a quarter of it).  One workgroup per CU walks a COLUMN of pixel tiles with one channel tile (tile_m = lane, lane + Gn, ...), and the
// schedule's tail, which the one-tile kernel fills with zero-fills, stages the NEXT tile's K-tiles 0 and 1 instead: K-tile nk of the current tile IS K-tile 0 of the next
// (nk even, so it lands in buffer 0), and when the loop ends the ring is exactly in the state the prologue leaves -- P0 C0 C1 P1 of K-tile 0 and P0 C0 of K-tile 1 issued,
// the last eight pieces in flight under `vmcnt(8)`.  The epilogue cannot use the whole ring as scratch any more: a wave transposes its 128 pixels x 64 channels 32 pixels
// at a time through 4 KiB of the two half-tiles nothing is staged into before the next tile's first phase (P1 and C1 of buffer 1: last read in phases 1-2 of the last
// K-tile, re-staged by phases 0-1 of the next tile's first), a barrier on either side.  The counted waits stay `vmcnt(8)`: with the epilogue's stores in the queue the
// count can only force MORE than it needs (loads return in order among loads), never less.
// DYN: the column is not walked at a fixed stride -- after its first tile a workgroup takes pixel tiles of its channel tile from the op's tile counter (Y7TConvArgs::tile_ctr, one
// counter per channel tile as in y7t_conv_ws128.hip; the ticket of the tile after next is fetched a whole tile ahead, so the prefetch always knows where it goes): what
// round 5 measured for every persistent kernel beside co-running work, and columns of 5-6 tiles (the 40 x 40 layers) stop ending a tile apart.
template <int ACT, bool DUAL, bool DYN>
__global__ void __launch_bounds__(512, 2) k_conv1x1_p8p(const Y7TConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C = P8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, wq = wave & 3, l31 = lane & 31, hi32 = lane >> 5;
    int bid = blockIdx.x;
    if (p.xcd_swizzle) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int n_tiles_n = p.Cout_pad / C::BN, Gn = (int)gridDim.x / n_tiles_n;      // (the launcher: gridDim.x a multiple of n_tiles_n)
    const int tile_n = bid % n_tiles_n, m_tiles = (p.M + C::BM - 1) / C::BM;
    int tile_m = bid / n_tiles_n;
    if (tile_m >= m_tiles) return;
    const int n0 = tile_n * C::BN, nk = p.Cin / C::BK;      // nk even (the launcher)

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc((void*)p.bias, 0, (unsigned)p.Cout_pad * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr2 = __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? p.in2 : p.in), 0, DUAL ? p.in2_bytes : p.in_bytes, 0x00020000);

    const int srow = wave * 8 + (lane >> 3);
    const int gch = (lane & 7) ^ ((srow >> 1) & 7);
    struct Off { int x[2][2]; int x2[DUAL ? 2 : 1][2]; };
    auto offsets = [&](int m0, Off& o) __attribute__((always_inline)) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int rd = 0; rd < 2; ++rd) {
                const int m = m0 + rd * 128 + h * 64 + srow;
                o.x[h][rd] = (int)C::OOB;
                if (DUAL) o.x2[h][rd] = (int)C::OOB;
                if (m < p.M) {
                    o.x[h][rd] = (m * p.ldin + p.cin_off + gch * 8) * 2;
                    if (DUAL) {
                        const int HW = p.H * p.W, b = m / HW, rem = m - b * HW, y = rem / p.W, x = rem - y * p.W;
                        o.x2[h][rd] = (((b * (p.H >> 1) + (y >> 1)) * (p.W >> 1) + (x >> 1)) * p.ldin2 + p.cin2_off + gch * 8) * 2;
                    }
                }
            }
    };
    Off cur, nxt;
    bool has_next = false;
    volatile int* const tkt = (volatile int*)(smem + C::LDS);      // DYN: two ticket slots behind the bias
    int* const ctr = DYN ? p.tile_ctr + tile_n : nullptr;
    if (DYN && tid == 0) tkt[0] = atomicAdd(ctr, 1);      // the ticket of this workgroup's SECOND tile (visible behind the prologue's barrier)
    const int wvo = tile_n * nk * (2 * C::HALF) + wave * 1024 + lane * 16;

    // piece rd of pixel half h of K-tile T of the CURRENT tile -> buffer b; T >= nk: K-tile T - nk of the NEXT tile of this workgroup's column (none left: zeros)
    auto stage_p1 = [&](int h, int b, int T, int rd) __attribute__((always_inline)) {
        char* dst = smem + b * C::BUF + h * C::HALF + wave * 1024 + rd * 8192;
        const bool own = T < nk, live = own || has_next;
        const int ci = (own ? T : T - nk) * C::BK;
        const bool up = DUAL && live && ci >= p.up_c0 && ci < p.up_c0 + p.up_C;
        const int xo = own ? cur.x[h][rd] : nxt.x[h][rd];
        if (up) __builtin_amdgcn_raw_ptr_buffer_load_lds(xr2, (LDS_AS void*)dst, 16, own ? cur.x2[DUAL ? h : 0][rd] : nxt.x2[DUAL ? h : 0][rd], (ci - p.up_c0) * 2, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (LDS_AS void*)dst, 16, live ? xo : (int)C::OOB, live ? ci * 2 : 0, 0, 0);
    };
    auto stage_c1 = [&](int h, int b, int T, int rd) __attribute__((always_inline)) {
        char* dst = smem + b * C::BUF + (2 + h) * C::HALF + wave * 1024 + rd * 8192;
        const bool own = T < nk, live = own || has_next;
        const int Tw = own ? T : T - nk;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (LDS_AS void*)dst, 16, live ? wvo + h * C::HALF + rd * 8192 : (int)C::OOB, live ? Tw * (2 * C::HALF) : 0, 0, 0);
    };
    auto stage_p = [&](int h, int b, int T) __attribute__((always_inline)) { stage_p1(h, b, T, 0); stage_p1(h, b, T, 1); };
    auto stage_c = [&](int h, int b, int T) __attribute__((always_inline)) { stage_c1(h, b, T, 0); stage_c1(h, b, T, 1); };

    const int swl = (l31 >> 1) & 7;
    const int prow = (grp * 64 + l31) * C::ROWB, crow = (wq * 32 + l31) * C::ROWB;
    half8 pf[2][4], c0[4], c1[4];
    auto read_p = [&](int hP, int b) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) pf[j][ks] = *(const half8*)(smem + b * C::BUF + hP * C::HALF + prow + j * 32 * C::ROWB + (((ks * 2 + hi32) ^ swl) << 4));
    };
    auto read_c = [&](half8 (&cf)[4], int hC, int b) __attribute__((always_inline)) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) cf[ks] = *(const half8*)(smem + b * C::BUF + (2 + hC) * C::HALF + crow + (((ks * 2 + hi32) ^ swl) << 4));
    };
    floatx16 acc[2][2][2];
    auto mma = [&](int hP, const half8 (&cf)[4], int hC) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[hP][j][hC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cf[ks], pf[j][ks], acc[hP][j][hC], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    auto ktile = [&](int T, auto bc) __attribute__((always_inline)) {
        constexpr int b = decltype(bc)::value;
        read_p(0, b); read_c(c0, 0, b);
        stage_c(1, b ^ 1, T + 1);
        P8_VMCNT(8);
        __builtin_amdgcn_s_barrier();
        mma(0, c0, 0);
        __builtin_amdgcn_s_barrier();
        read_c(c1, 1, b);
        stage_p(1, b ^ 1, T + 1);
        P8_VMCNT(8);
        __builtin_amdgcn_s_barrier();
        mma(0, c1, 1);
        __builtin_amdgcn_s_barrier();
        read_p(1, b);
        stage_p(0, b, T + 2);
        P8_VMCNT(8);
        __builtin_amdgcn_s_barrier();
        mma(1, c1, 1);
        __builtin_amdgcn_s_barrier();
        stage_c(0, b, T + 2);
        P8_VMCNT(8);
        __builtin_amdgcn_s_barrier();
        mma(1, c0, 0);
        __builtin_amdgcn_s_barrier();
    };

    // ---- prologue of the workgroup's FIRST tile (every later tile finds the ring in this state when the previous tile's loop ends) ----
    offsets(tile_m * C::BM, cur);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(br, (LDS_AS void*)(smem + C::BIAS + wq * 256), 4, (n0 + wq * 64 + lane) * 4, 0, 0, 0);
    stage_p(0, 0, 0); stage_c(0, 0, 0); stage_c(1, 0, 0); stage_p(1, 0, 0); stage_p(0, 1, 1); stage_c(0, 1, 1);
    P8_VMCNT(8);
    __builtin_amdgcn_s_barrier();
    typedef __attribute__((ext_vector_type(4))) unsigned uint4v;
    half_t* outp = (half_t*)p.out;
    char* const scr = smem + C::BUF + (wave < 4 ? 1 : 3) * C::HALF + (wave & 3) * 4096;      // 4 KiB of P1 / C1 of buffer 1 (see above)
    for (int it = 0;; ++it) {
        const int m0 = tile_m * C::BM, tile_next = DYN ? Gn + __builtin_amdgcn_readfirstlane(tkt[it & 1]) : tile_m + Gn;
        has_next = tile_next < m_tiles;
        // the ticket of the tile after next: a returning atomic whose result the COMPILER must not wait for (its s_waitcnt vmcnt(0) here would drain this wave's eight
        // in-flight half-tile pieces and the previous epilogue's stores).  Issued as inline asm: by the time the K loop is over this wave has issued >= 16 younger
        // loads behind >= 8 `vmcnt(8)` waits, and loads return in order -- the register holds the ticket when it is read below.
        int fetched = 0;
        if (DYN && tid == 0 && has_next) {
#if defined(Y7T_CONVSIM)
            fetched = atomicAdd(ctr, 1);
#else
            asm volatile("global_atomic_add %0, %1, %2, off sc0" : "=v"(fetched) : "v"(ctr), "v"(1) : "memory");
#endif
        }
        if (has_next) offsets(tile_next * C::BM, nxt);
#pragma unroll
        for (int hC = 0; hC < 2; ++hC)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                typedef __attribute__((ext_vector_type(4))) float float4v;
                const float4v bv = *(const float4v*)(smem + C::BIAS + (wq * 64 + hC * 32 + 8 * g + 4 * hi32) * 4);
#pragma unroll
                for (int hP = 0; hP < 2; ++hP)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[hP][j][hC][g * 4 + e] = bv[e];
            }
        if (grp == 1) __builtin_amdgcn_s_barrier();
        for (int T = 0; T < nk; T += 2) {
            ktile(T, std::integral_constant<int, 0>{});
            ktile(T + 1, std::integral_constant<int, 1>{});
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();      // re-align the groups: every wave is past its last fragment read
        if (DYN && tid == 0 && has_next) tkt[(it + 1) & 1] = fetched;      // (read at the start of the next tile, behind the barrier at the end of this one)
        // ---- epilogue, 32 pixels x 64 channels at a time through this wave's 4 KiB ----
#pragma unroll
        for (int hP = 0; hP < 2; ++hP)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
#pragma unroll
                for (int hC = 0; hC < 2; ++hC)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        unsigned w[2][2];
#pragma unroll
                        for (int gg = 0; gg < 2; ++gg) {
                            const int g = gp * 2 + gg;
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = act_t<ACT>(acc[hP][j][hC][g * 4 + e]);
                            typedef __attribute__((ext_vector_type(2))) _Float16 half2v;
                            half2v h0 = {(half_t)v[0], (half_t)v[1]}, h1 = {(half_t)v[2], (half_t)v[3]};
                            w[gg][0] = __builtin_bit_cast(unsigned, h0);
                            w[gg][1] = __builtin_bit_cast(unsigned, h1);
                        }
                        auto r0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                        auto r1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                        uint4v pk = {r0[0], r1[0], r0[1], r1[1]};
                        const int ch = hC * 4 + gp * 2 + hi32;
                        *(uint4v*)(scr + l31 * 128 + ((ch ^ (l31 & 7)) << 4)) = pk;
                    }
                P8_WAVE_SYNC();
                uint4v v4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int px = k * 8 + (lane >> 3), ch = lane & 7;
                    v4[k] = *(const uint4v*)(scr + px * 128 + ((ch ^ (px & 7)) << 4));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int px = k * 8 + (lane >> 3), ch = lane & 7;
                    const int m = m0 + grp * 128 + hP * 64 + j * 32 + px, n = n0 + wq * 64 + ch * 8;
                    if (m < p.M && n < p.Cout) *(uint4v*)(outp + (size_t)m * p.ldout + p.cout_off + n) = v4[k];
                }
                P8_WAVE_SYNC();
            }
        if (!has_next) break;
        __builtin_amdgcn_s_barrier();                    // nobody reads its scratch any more: the next tile's first phases stage C1 / P1 of K-tile 1 over it
        cur = nxt;
        tile_m = tile_next;
    }
    P8_VMCNT(0);      // (the last tile's tail: zero-fills into half-tiles nobody reads, landed before the workgroup's LDS is handed on)
    if (DYN && tid == 0) {      // the last workgroup to leave hands every counter of the op back at zero
        if (atomicAdd(p.tile_ctr + Y7T_TILE_CTR_DONE, 1) == (int)gridDim.x - 1) {
            for (int i = 0; i <= Y7T_TILE_CTR_DONE; ++i) p.tile_ctr[i] = 0;
        }
    }
#endif
}

}   // namespace

// korder 7 layers only (detector/graph.py::p8_eligible mirrors the conditions)
int y7t_conv_p8_launch(const Y7TConvArgs& a, hipStream_t s) {
    using C = P8;
    bool ok = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.Cin % C::BK == 0 && a.K_pad == a.Cin && a.Cout_pad % C::BN == 0 && !a.out_f32 && !a.epi &&
              !(a.Cout & 7) && !(a.ldout & 7) && !(a.cout_off & 7) && !(a.ldin & 7) && !(a.cin_off & 7) && a.in_bytes <= C::OOB - (1u << 24) && a.Ho == a.H && a.Wo == a.W;
    if (a.up_C > 0) ok = ok && a.up_c0 % C::BK == 0 && a.up_C % C::BK == 0 && a.up_c0 + a.up_C <= a.Cin && !(a.H & 1) && !(a.W & 1) && !(a.ldin2 & 7) && !(a.cin2_off & 7) &&
                         a.in2_bytes <= C::OOB - (1u << 24);
    if (!ok) {
        y7t_set_error("conv: weights are in the 256 x 64 panel order (korder 7) but the layer is not a 1x1 / stride 1 convolution with Cin %% 64 == 0, Cout_pad %% 256 == 0 "
                      "and an aligned fp16 output (Cin=%d Cout=%d Cout_pad=%d ldout=%d cout_off=%d up=[%d,%d))", a.Cin, a.Cout, a.Cout_pad, a.ldout, a.cout_off, a.up_c0,
                      a.up_c0 + a.up_C);
        return Y7T_E_ARG;
    }
    static Y7TOncePerDevice attr;      // (the attribute is per device: ADVICE r4)
    if (int e_ = y7t_once_per_device(attr, [&]() -> int {
#define P8_ATTR(ACT, DUAL) Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv1x1_p8<ACT, DUAL>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
        P8_ATTR(Y7T_ACT_NONE, false) P8_ATTR(Y7T_ACT_SILU, false) P8_ATTR(Y7T_ACT_LEAKY, false) P8_ATTR(Y7T_ACT_NONE, true) P8_ATTR(Y7T_ACT_SILU, true) P8_ATTR(Y7T_ACT_LEAKY, true)
#undef P8_ATTR
        return 0;
    })) return e_;
    const int grid = ((a.M + C::BM - 1) / C::BM) * (a.Cout_pad / C::BN);
    const bool dual = a.up_C > 0;
    // the persistent form (k_conv1x1_p8p) where a workgroup gets more than one tile and the K-tiles of a tile are an even number (the ring's parity)
    static const int persist = y7t_exp_switch("Y7T_CONV_P8_PERSIST", 1);
    const int n_tiles_n = a.Cout_pad / C::BN, ncu = y7t_num_cus(), gp_ = (ncu / n_tiles_n) * n_tiles_n;
    // (at least two tiles per workgroup on average: with 500 tiles on 256 workgroups -- the 20 x 20 layers at 80 frames -- the persistent form measured 0-3 % behind, r6ax)
    if (persist && !a.ablate && ((a.Cin / C::BK) & 1) == 0 && gp_ > 0 && grid >= 2 * gp_) {
        static Y7TOncePerDevice attrp;
        if (int e_ = y7t_once_per_device(attrp, [&]() -> int {
#define P8P_ATTR(ACT, DUAL) Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv1x1_p8p<ACT, DUAL, false>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS + 16)); \
                            Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv1x1_p8p<ACT, DUAL, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS + 16));
            P8P_ATTR(Y7T_ACT_NONE, false) P8P_ATTR(Y7T_ACT_SILU, false) P8P_ATTR(Y7T_ACT_LEAKY, false) P8P_ATTR(Y7T_ACT_NONE, true) P8P_ATTR(Y7T_ACT_SILU, true) P8P_ATTR(Y7T_ACT_LEAKY, true)
#undef P8P_ATTR
            return 0;
        })) return e_;
        static const int dyn_env = y7t_switch("Y7T_CONV_WS_DYN", 1);      // (the persistent kernels' product switch: =0 the static column)
        const bool dyn = a.tile_ctr && dyn_env && n_tiles_n <= Y7T_TILE_CTR_DONE;
#define P8P_GO(ACT) \
        do { if (dual && dyn) hipLaunchKernelGGL((k_conv1x1_p8p<ACT, true, true>), dim3(gp_), dim3(C::NT), C::LDS + 16, s, a); \
             else if (dual) hipLaunchKernelGGL((k_conv1x1_p8p<ACT, true, false>), dim3(gp_), dim3(C::NT), C::LDS + 16, s, a); \
             else if (dyn) hipLaunchKernelGGL((k_conv1x1_p8p<ACT, false, true>), dim3(gp_), dim3(C::NT), C::LDS + 16, s, a); \
             else hipLaunchKernelGGL((k_conv1x1_p8p<ACT, false, false>), dim3(gp_), dim3(C::NT), C::LDS + 16, s, a); } while (0)
        if (a.act == Y7T_ACT_SILU) P8P_GO(Y7T_ACT_SILU);
        else if (a.act == Y7T_ACT_LEAKY) P8P_GO(Y7T_ACT_LEAKY);
        else P8P_GO(Y7T_ACT_NONE);
#undef P8P_GO
        Y7T_LAUNCH_CHECK();
        y7t_note_kernel("p8<256,256,64> 1x1%s", dual ? " upsample-on-read" : "");
        return 0;
    }
#if Y7T_ABLATE      // liby7t_ablate.so only
    if (a.ablate && a.act == Y7T_ACT_SILU && !dual) {      // timing ablations of the plain SiLU instance (Y7T_CONV_ABLATE=1|2|4|8|16; wrong results)
#define P8_ABL(N) case N: Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv1x1_p8<Y7T_ACT_SILU, false, N>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS)); \
                          hipLaunchKernelGGL((k_conv1x1_p8<Y7T_ACT_SILU, false, N>), dim3(grid), dim3(C::NT), C::LDS, s, a); break;
        switch (a.ablate) { P8_ABL(1) P8_ABL(2) P8_ABL(4) P8_ABL(8) P8_ABL(16) P8_ABL(6) P8_ABL(7) default: y7t_set_error("conv: unknown p8 ablation %d", a.ablate); return Y7T_E_ARG; }
#undef P8_ABL
        Y7T_LAUNCH_CHECK();
        y7t_note_kernel("p8<256,256,64> 1x1 ablated");
        return 0;
    }
#endif
#define P8_GO(ACT) \
    do { if (dual) hipLaunchKernelGGL((k_conv1x1_p8<ACT, true>), dim3(grid), dim3(C::NT), C::LDS, s, a); \
         else hipLaunchKernelGGL((k_conv1x1_p8<ACT, false>), dim3(grid), dim3(C::NT), C::LDS, s, a); } while (0)
    if (a.act == Y7T_ACT_SILU) P8_GO(Y7T_ACT_SILU);
    else if (a.act == Y7T_ACT_LEAKY) P8_GO(Y7T_ACT_LEAKY);
    else P8_GO(Y7T_ACT_NONE);
#undef P8_GO
    Y7T_LAUNCH_CHECK();
    y7t_note_kernel("p8<256,256,64> 1x1%s", dual ? " upsample-on-read" : "");
    return 0;
}
