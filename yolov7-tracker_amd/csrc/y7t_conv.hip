// y7t_conv.hip -- YOLOv7 Conv(+folded BN)+bias+activation as an im2col-free implicit GEMM on the CDNA4 matrix
// cores (v_mfma_f32_32x32x16_f16), NHWC fp16 activations, fp32 accumulate.
//
// Restates /root/reference/models/common.py:99-111 (Conv.fuseforward = act(conv2d(x, W') + b') with autopad :23-27)
// after utils/torch_utils.py:181-201 (fuse_conv_and_bn) has folded the BatchNorm into W', b'.
//
// GEMM view:  D[n][m] = sum_k Wp[n][k] * X[m][k]      n = output channel, m = output pixel (b, ho, wo),
//             k = (kh*KW + kw)*Cin + ci  -- the im2col matrix X is never built: each 16-byte K-chunk of a
//             pixel row is fetched straight from the NHWC tensor (8 consecutive channels of one tap); a chunk whose tap
//             falls into the padding / whose row is past M / whose k is past K is an out-of-range buffer offset.
// Tiling:     256 threads = 4 waves (2 along n x 2 along m); block tile BN x BM, K-step 64 (128-byte rows);
//             both operands staged through LDS with `buffer_load_dwordx4 ... lds` (16 B per lane DMA, lane-linear
//             destination, hardware range check supplies the zeros of padding taps / ragged edges),
//             double-buffered: the loads of K-step t+1 are in flight while the MFMAs of step t run; one barrier
//             per K-step.  LDS rows are XOR-swizzled at 16-byte granularity (slot = chunk ^ ((row >> 1) & 7): two 128-byte rows
//             fill one 256-byte bank row, so every 16-lane service group of ds_read_b128 touches 16 distinct slots) -- applied
//             to the per-lane SOURCE address on the way in and to the ds_read address on the way out.
//             The weight operand goes to MFMA's A side so that each lane ends up holding 4 consecutive output
//             channels of ONE pixel: the epilogue (bias + SiLU/LeakyReLU) packs them into one 8-byte NHWC store.
// Concat elimination: input and output are channel SLICES of wider NHWC buffers (ldin/cin_off, ldout/cout_off),
//             so producers write straight into their slot of a concat buffer and consumers read slices.
#include "y7t_common.h"
#include "y7t_det.h"
#include <stdlib.h>

#include "y7t_conv_common.h"
#include <algorithm>
#include <vector>

// KM = 1: 1x1 / stride 1 / pad 0 with Cin % 64 == 0 -- no taps, no padding, no K tail: every pixel DMA is `row offset (or out of range)
// + scalar channel offset`, so the K loop carries no address arithmetic and no control flow besides its own counter.
// EPI = 1 (Detect 1x1 convs in a fused forward): the epilogue decodes + filters (models/yolo.py:49-56, utils/general.py:629-662) instead
// of storing the tile.  DUAL (KM = 1 only): upsample-on-read -- K-steps whose channels lie in [up_c0, up_c0 + up_C) fetch pixel (y, x)
// from the half-resolution tensor `in2` at (y >> 1, x >> 1) (nn.Upsample(None, 2, 'nearest') folded into the consumer's loader).
constexpr int kNW = 4;      // waves per workgroup (NW below stays a template parameter: 8-wave instances with 256 x 256 x 64 / 256 x 128 x 64 tiles at two waves per
                            // SIMD were built and measured in round 3 -- 1x1 layers 5-15 % SLOWER, stride-2 3x3 layers on a par with the stride-2 patch kernel,
                            // profiles/r03_conv_variants.txt -- and are no longer instantiated)
template <int BM, int BN, int BK, int NST, bool UT, int KM = 0, int EPI = 0, bool DUAL = false, int NW = kNW>
__global__ void __launch_bounds__(64 * NW, (NW == 8 || BM * BN >= 256 * 256 ? 1 : 2)) k_conv_igemm(const Y7TConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource builtins do not exist in the host pass (it only needs the stub)
    constexpr int ROWB = BK * 2;                 // bytes per LDS row (128 or 64)
    constexpr int CPR = BK / 8;                  // 16-byte chunks per row (8 or 4)
    constexpr int RPW = 64 / CPR;                // rows covered by one wave-wide 1 KiB DMA (8 or 16)
    constexpr int kNT = 64 * NW, kNWM = NW / 2;  // threads, waves along the pixel dimension
    constexpr int RPR = RPW * NW;                // rows per load round over the waves (32 or 64; twice that with 8 waves)
    constexpr int WTN = BN / 2, WTM = BM / kNWM; // wave tile
    static_assert(NW == 4 || EPI == 0, "the 8-wave instances cover the plain convolution and the upsample-on-read loader, not the Detect epilogue");
    constexpr int TN = WTN / 32, TM = WTM / 32;  // 32x32 MFMA tiles per wave
    constexpr int RM = BM / RPR, RN = BN / RPR;  // load rounds per operand
    constexpr int NLD = RM + RN;                 // DMA instructions per thread per stage
    constexpr int STAGE = (BM + BN) * ROWB;      // bytes per LDS stage
    constexpr int LDS_EPI_BYTES = EPI == 1 ? BM * (BN + 1) * 4 : BM * (BN * 2 + 16);
    constexpr int BIAS_OFF = NST * STAGE > LDS_EPI_BYTES ? NST * STAGE : LDS_EPI_BYTES;   // BN biases behind everything else
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / kNWM, wm = wave % kNWM;
    const int n_tiles_m = (p.M + BM - 1) / BM;
    // XCD-aware order: workgroup b runs on XCD b % 8 (observed dispatch order), so give every XCD a CONTIGUOUS range of
    // tiles -- neighbouring pixel tiles share halo rows and the same weight panel in that XCD's private 4 MiB L2
    int bid = blockIdx.x;
    // split-K: `splitk` consecutive workgroups share one output tile, each reduces a contiguous range of K-steps into its own
    // fp32 slab (k_splitk_reduce sums the slabs in a fixed order -> deterministic); used when a layer has too few tiles to fill
    // 256 CUs (the 20x20 / 40x40 maps, batch-1 latency mode)
    if (p.xcd_swizzle) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    // channel tiles fastest: the blocks that share one pixel tile (and therefore all its input lines) run next to each
    // other on the same XCD; the weight panels are small and stay cached anyway
    const int n_tiles_n = p.Cout_pad / BN;
    const int split = bid % p.splitk;
    bid /= p.splitk;
    const int tile_n = p.tile_order ? bid % n_tiles_n : bid / n_tiles_m, tile_m = p.tile_order ? bid / n_tiles_n : bid % n_tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    // the epilogue's biases: fetched now into an LDS corner no stage touches (a global load at the end of a short workgroup is exposed latency)
    if (tid < BN) ((float*)(smem + BIAS_OFF))[tid] = p.bias[n0 + tid];
    const float* lbias = (const float*)(smem + BIAS_OFF);

    // Buffer descriptors (raw, 32-bit byte offsets): a lane whose tap falls into the padding, whose row is past M
    // or whose k is past K gets offset 0xffffffff -> the hardware range check returns zeros into LDS, so there is
    // no zero page and no 64-bit address arithmetic in the K loop.
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xr2 = __builtin_amdgcn_make_buffer_rsrc((void*)(DUAL ? p.in2 : p.in), 0, DUAL ? p.in2_bytes : p.in_bytes, 0x00020000);

    // ---- per-thread load geometry.  LDS slot of logical chunk c in row r:  c ^ sw(r),
    //      sw(r) = (r >> 1) & 7 for 128-byte rows, (r >> 2) & 3 for 64-byte rows (conflict-free ds_read_b128) ----
    const int lrow = wave * RPW + lane / CPR;      // row inside a load round
    const int gchunk = (lane % CPR) ^ (BK == 64 ? ((lrow >> 1) & 7) : ((lrow >> 2) & 3));
    int xoff[RM];          // byte offset of (b, hi0, wi0, cin_off [+ this lane's chunk]) -- may be negative before the tap is added
    int xoff2[DUAL ? RM : 1];   // DUAL: the same pixel in the half-resolution source
    unsigned vmask[RM];    // bit t: tap t of this row is inside the image
    const int HoWo = p.Ho * p.Wo;
    const float inv_howo = 1.0f / (float)HoWo, inv_wo = 1.0f / (float)p.Wo;
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        const int m = m0 + r * RPR + lrow;
        vmask[r] = 0;
        xoff[r] = KM == 1 ? (int)0xFF000000u : 0;      // KM 1: an offset that stays out of range with the (< 16 MiB) scalar channel offset added
        if (DUAL) xoff2[r] = (int)0xFF000000u;
        if (m < p.M) {
            // m -> (b, ho, wo) with a float reciprocal + one-step fix-up (M < 2^24), no integer division
            int b = (int)((float)m * inv_howo);
            b += ((b + 1) * HoWo <= m) - (b * HoWo > m);
            const int rem = m - b * HoWo;
            int ho = (int)((float)rem * inv_wo);
            ho += ((ho + 1) * p.Wo <= rem) - (ho * p.Wo > rem);
            const int wo = rem - ho * p.Wo;
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            xoff[r] = ((((b * p.H + hi0) * p.W + wi0) * p.ldin + p.cin_off) + (UT ? gchunk * 8 : 0)) * 2;
            if (DUAL) xoff2[r] = ((((b * (p.H >> 1) + (ho >> 1)) * (p.W >> 1) + (wo >> 1)) * p.ldin2 + p.cin2_off) + gchunk * 8) * 2;
            if (KM == 1) continue;      // (m past M keeps the out-of-range sentinel set below)
            // taps inside the image: kh in [klo, khi], kw in [wlo, whi]
            const int klo = hi0 < 0 ? -hi0 : 0, khi = (p.H - 1 - hi0) < (p.KH - 1) ? (p.H - 1 - hi0) : (p.KH - 1);
            const int wlo = wi0 < 0 ? -wi0 : 0, whi = (p.W - 1 - wi0) < (p.KW - 1) ? (p.W - 1 - wi0) : (p.KW - 1);
            const unsigned colbits = (whi >= wlo) ? (((2u << whi) - 1u) & ~((1u << wlo) - 1u)) : 0u;
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
                if (kh >= klo && kh <= khi) vmask[r] |= colbits << (kh * p.KW);
        }
    }
    // korder 3 (1x1 layers, BK = 32; detector/weights.py::panel_pack_linear): the BN x 32 weight panel of every K-step is stored as one
    // contiguous block that already is the swizzled LDS image -> a wave's DMA reads 1 KiB of consecutive bytes (8 full cache lines)
    // instead of 16 half lines from 16 weight rows (17 vs 46 clocks in the vector L1, scripts/ubench/dma_patterns.hip)
    const bool wpanel = (p.korder == 3);
    int woff[RN];
#pragma unroll
    for (int r = 0; r < RN; ++r)
        woff[r] = wpanel ? tile_n * (p.K_pad / BK) * (BN * ROWB) + (r * RPR + wave * RPW) * ROWB + lane * 16
                         : ((n0 + r * RPR + lrow) * p.K_pad + gchunk * 8) * 2;

    // k bookkeeping of the next stage to load: uniform (scalar) when every K-step lies inside one tap (Cin % BK == 0)
    const int nk_all = p.K_pad / BK;
    const int kt0 = split * p.ksteps, nk = (kt0 + p.ksteps < nk_all ? kt0 + p.ksteps : nk_all) - kt0;   // this split's K-steps
    int k = (UT ? 0 : gchunk * 8) + kt0 * BK;
    int tap = k / p.Cin, ci = k - tap * p.Cin;

    // one DMA of the stage being filled: idx in [0, NLD): first the RM pixel-row rounds, then the RN weight rounds.
    // tap/ci/k describe that stage; advance_k() moves them to the following stage.
    auto issue_one = [&](int stage, int kt, int idx) {
        char* xs = smem + stage * STAGE;
        char* ws = xs + BM * ROWB;
        if (idx < RM && KM == 1) {
            if (DUAL && ci >= p.up_c0 && ci < p.up_c0 + p.up_C)      // wave-uniform: a K-step lies in one source (up_c0, up_C multiples of BK)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr2, (LDS_AS void*)(xs + (idx * RPR + wave * RPW) * ROWB), 16, xoff2[DUAL ? idx : 0],
                                                         (ci - p.up_c0) * 2, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (LDS_AS void*)(xs + (idx * RPR + wave * RPW) * ROWB), 16, xoff[idx], ci * 2, 0, 0);
        } else if (idx < RM) {
            const int r = idx;
            const int kh = (p.KW == 1) ? tap : (tap * 43) >> 7;   // tap / 3 for tap < 128
            const int kw = tap - kh * p.KW;
            const int tapoff = ((kh * p.W + kw) * p.ldin + ci) * 2;
            const bool kvalid = UT ? true : (k < p.K);            // UT: a tap index past KH*KW has no mask bit
            const bool ok = kvalid && ((vmask[r] >> tap) & 1u);
            const int voff = ok ? xoff[r] + tapoff : -1;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (LDS_AS void*)(xs + (r * RPR + wave * RPW) * ROWB), 16, voff, 0, 0, 0);
        } else {
            const int r = idx - RM;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (LDS_AS void*)(ws + (r * RPR + wave * RPW) * ROWB), 16, woff[r],
                                                     wpanel ? (kt0 + kt) * (BN * ROWB) : (kt0 + kt) * BK * 2, 0, 0);
        }
    };
    // K order.  Default: k = tap*Cin + ci (taps outermost).  korder = 1 (3x3, Cin % 64 == 0, weights packed to match):
    // (kh, 64-channel chunk, kw) -- the three kw taps of one chunk are consecutive K-steps and touch the same input lines
    // shifted by one pixel, so they hit in L1/L2 instead of coming back from the Infinity Cache a dozen steps later.
    const int nchunk = p.Cin >> 6;
    int o_kw = 0, o_c = 0, o_kh = 0, o_sub = 0;
    if (KM == 1) { tap = 0; ci = kt0 * BK; }
    if (KM != 1 && UT && p.korder == 1) {   // decode the (kh, chunk, kw) odometer at this split's first K-step
        const int g = (kt0 * BK) >> 6;
        o_kw = g % p.KW; o_c = (g / p.KW) % nchunk; o_kh = g / (p.KW * nchunk); o_sub = ((kt0 * BK) & 63) / BK;
        tap = o_kh * p.KW + o_kw; ci = (o_c << 6) + o_sub * BK;
    }
    auto advance_k = [&]() {
        if (KM == 1) { ci += BK; return; }
        if (UT && p.korder == 1) {
            if (BK < 64 && ++o_sub < 64 / BK) { ci += BK; return; }
            o_sub = 0;
            if (++o_kw == p.KW) { o_kw = 0; if (++o_c == nchunk) { o_c = 0; ++o_kh; } }
            tap = o_kh * p.KW + o_kw; ci = o_c << 6;
            return;
        }
        k += BK; ci += BK;
        while (ci >= p.Cin) { ci -= p.Cin; ++tap; }
    };
    auto issue_loads = [&](int stage, int kt) {
#pragma unroll
        for (int idx = 0; idx < NLD; ++idx) issue_one(stage, kt, idx);
        advance_k();
    };

    floatx16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // prologue: NST-1 stages in flight
#pragma unroll
    for (int s = 0; s < NST - 1; ++s) if (s < nk) issue_loads(s, s);
    const int l31 = lane & 31, hi32 = lane >> 5;
    int cur = 0, nxt = NST - 1;      // ring positions of stage kt and of stage kt+NST-1
    for (int kt = 0; kt < nk; ++kt) {
        // wait until THIS wave's DMAs of stage kt have landed (the younger stages stay in flight), then barrier:
        // after it every wave's part of stage kt is visible and nobody still reads the buffer we are about to refill
        const int ahead = (nk - 1 - kt) < (NST - 2) ? (nk - 1 - kt) : (NST - 2);
        if (NST >= 4 && ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NLD) : "memory");
        else if (NST >= 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NLD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const bool do_load = kt + NST - 1 < nk;
        const char* xs = smem + cur * STAGE;
        const char* ws = xs + BM * ROWB;
        constexpr int KS = BK / 16;                 // MFMA k-substeps per stage
        constexpr int LPK = (NLD + KS - 1) / KS;    // DMAs issued behind each substep's MFMAs (spreads them over the stage)
        // fragments are double-buffered in registers: the ds_reads of substep ks+1 are issued before the MFMAs of substep ks
        half8 wf[2][TN], xf[2][TM];
        auto read_frags = [&](int ks, int buf) {
            const int q = ks * 2 + hi32;   // logical 16-byte chunk of this lane group
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int row = wn * WTN + i * 32 + l31;
                const int sl = q ^ (BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3));
                wf[buf][i] = *(const half8*)(ws + row * ROWB + (sl << 4));
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int row = wm * WTM + j * 32 + l31;
                const int sl = q ^ (BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3));
                xf[buf][j] = *(const half8*)(xs + row * ROWB + (sl << 4));
            }
        };
        read_frags(0, 0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cb = ks & 1;
            if (ks + 1 < KS) read_frags(ks + 1, cb ^ 1);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cb][i], xf[cb][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
            if (do_load) {
#pragma unroll
                for (int t = 0; t < LPK; ++t)
                    if (ks * LPK + t < NLD) issue_one(nxt, kt + NST - 1, ks * LPK + t);
            }
        }
        if (do_load) advance_k();
        cur = (cur + 1 == NST) ? 0 : cur + 1;
        nxt = (nxt + 1 == NST) ? 0 : nxt + 1;
    }

    if (EPI == 1) {
        // ---- Detect: stage the fp32 tile (bias added) in LDS, then one thread per (pixel, anchor): sigmoid, candidate filter, decode,
        // atomic append.  Row stride Cout | 1 dwords: odd, so the per-pixel rows do not collide on banks. ----
        const Y7TDecode& q = p.dec;
        const int LD = p.Cout | 1;
        float* tile = (float*)smem;
        __syncthreads();          // every wave is done with the staging buffers
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int pix = wm * WTM + j * 32 + l31;
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int nl = wn * WTN + i * 32 + 8 * g + 4 * hi32 + e;
                        if (nl < p.Cout) tile[pix * LD + nl] = acc[i][j][g * 4 + e] + lbias[nl];
                    }
        }
        __syncthreads();
        // Slots: a workgroup reserves ONE block per image with a single global atomic (every candidate bumping count[b] itself
        // serialised ~200 same-address atomics per workgroup in L2: 470 us on the 40x40 level); ranks inside the block come from LDS atomics.
        const int HoWo2 = q.ny * q.nx, nc = q.no - 5;
        int* lcount = (int*)(smem + BM * LD * 4);        // [BM] candidates of image b0 + i found by this workgroup
        int* lbase = lcount + BM;                        // [BM] first slot of that block
        const int b0 = m0 / HoWo2;
        for (int i = tid; i < BM; i += 256) lcount[i] = 0;
        __syncthreads();
        constexpr int MAXT = (BM * 3 + 255) / 256;       // tasks per thread (na <= 3)
        int t_img[MAXT], t_rank[MAXT], t_cidx[MAXT];
        float t_box[MAXT][4], t_score[MAXT], t_cls[MAXT];
#pragma unroll
        for (int it = 0; it < MAXT; ++it) {
#pragma clang fp contract(off)      // the decode below must round like k_decode_filter (y7t_post.hip is built without contraction): bit-equal candidates
            t_img[it] = -1;
            const int t = tid + it * 256;
            if (t >= BM * q.na) continue;
            const int pix = t / q.na, an = t - pix * q.na;
            const int m = m0 + pix;
            if (m >= p.M) continue;
            const float* v = tile + pix * LD + an * q.no;
            const float obj = 1.0f / (1.0f + expf(-v[4]));
            if (!(obj > q.conf_thres)) continue;                  // xc = prediction[..., 4] > conf_thres
            float best = -1.f; int bj = 0;
            for (int c = 0; c < nc; ++c) {                        // x[:, 5:] *= x[:, 4:5]; conf, j = x[:, 5:].max(1)
                const float sc = (1.0f / (1.0f + expf(-v[5 + c]))) * obj;
                if (sc > best) { best = sc; bj = c; }
            }
            if (!(best > q.conf_thres)) continue;
            const int b = m / HoWo2, rem = m - b * HoWo2, y = rem / q.nx, x = rem - y * q.nx;
            const float sx = 1.0f / (1.0f + expf(-v[0])), sy = 1.0f / (1.0f + expf(-v[1]));
            const float sw = 1.0f / (1.0f + expf(-v[2])), sh = 1.0f / (1.0f + expf(-v[3]));
            const float cx = (sx * 2.f - 0.5f + (float)x) * q.stride, cy = (sy * 2.f - 0.5f + (float)y) * q.stride;
            const float w = (sw * 2.f) * (sw * 2.f) * q.aw[an], h = (sh * 2.f) * (sh * 2.f) * q.ah[an];
            t_box[it][0] = cx - w / 2; t_box[it][1] = cy - h / 2; t_box[it][2] = cx + w / 2; t_box[it][3] = cy + h / 2;   // xywh2xyxy
            t_score[it] = best; t_cls[it] = (float)bj;
            t_cidx[it] = q.row0 + (an * q.ny + y) * q.nx + x;
            t_img[it] = b - b0;
            t_rank[it] = atomicAdd(lcount + (b - b0), 1);
        }
        __syncthreads();
        for (int i = tid; i < BM; i += 256) {
            const int c = lcount[i];
            lbase[i] = c > 0 ? atomicAdd(q.count + b0 + i, c) : 0;
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < MAXT; ++it) {
            if (t_img[it] < 0) continue;
            const int b = b0 + t_img[it], slot = lbase[t_img[it]] + t_rank[it];
            if (slot >= q.cap) continue;
            float* bo = q.cbox + ((size_t)b * q.cap + slot) * 4;
            bo[0] = t_box[it][0]; bo[1] = t_box[it][1]; bo[2] = t_box[it][2]; bo[3] = t_box[it][3];
            q.cscore[(size_t)b * q.cap + slot] = t_score[it];
            q.ccls[(size_t)b * q.cap + slot] = t_cls[it];
            q.cidx[(size_t)b * q.cap + slot] = t_cidx[it];
        }
        return;
    }
    if (p.splitk > 1) {   // raw fp32 partial sums of this K range: slab[split][m][Cout_pad]
        float* slab = p.partial + (size_t)split * p.M * p.Cout_pad;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = m0 + wm * WTM + j * 32 + l31;
            if (m >= p.M) continue;
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = n0 + wn * WTN + i * 32 + 8 * g + 4 * hi32;
                    typedef __attribute__((ext_vector_type(4))) float float4v;
                    float4v v = {acc[i][j][g * 4], acc[i][j][g * 4 + 1], acc[i][j][g * 4 + 2], acc[i][j][g * 4 + 3]};
                    *(float4v*)(slab + (size_t)m * p.Cout_pad + n) = v;
                }
        }
        return;
    }
    // ---- epilogue: bias + activation.  A lane holds channels n..n+3 of its pixel for each group g (n = 8g + 4*hi32);
    // v_permlane32_swap between groups g and g+1 gives lanes 0-31 channels 8g..8g+7 and lanes 32-63 channels
    // 8(g+1)..8(g+1)+7 of their pixel -> one 16-byte NHWC store per lane per group pair ----
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * WTM + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int nb = n0 + wn * WTN + i * 32;
            if (p.out_f32 || (p.Cout & 7) || (p.ldout & 7) || (p.cout_off & 7)) {   // direct path (Detect heads: fp32, Cout = 3*(5+nc))
                if (m >= p.M) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int n = nb + 8 * g + 4 * hi32;
                    if (n >= p.Cout) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_fn(acc[i][j][g * 4 + e] + lbias[n - n0 + e], p.act);
                    const size_t o = (size_t)m * p.ldout + p.cout_off + n;
                    if (p.out_f32) {
                        float* op = (float*)p.out + o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < p.Cout) op[e] = v[e];
                    } else {
                        half_t* op = (half_t*)p.out + o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < p.Cout) op[e] = (half_t)v[e];
                    }
                }
            }
        }
    }
    if (p.out_f32 || (p.Cout & 7) || (p.ldout & 7) || (p.cout_off & 7)) return;
    // fp16 path: transpose the tile through LDS so that every pixel's BN channels leave as full 128/256-byte lines.
    // Row stride BN*2 + 16 bytes: the 8-lane service groups of ds_write_b128 land on disjoint banks.
    constexpr int OROW = BN * 2 + 16;
    __syncthreads();          // every wave is done with the staging buffers
    typedef __attribute__((ext_vector_type(4))) float float4v;
    act_dispatch(p.act, [&](auto act_c) {       // one branch on the activation, specialised bodies
    constexpr int ACT = decltype(act_c)::value;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int pix = wm * WTM + j * 32 + l31;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
            const int nl = wn * WTN + i * 32;
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                unsigned w[2][2];   // [group of the pair][2 packed half2]
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int g = gp * 2 + gg;
                    const int n = n0 + nl + 8 * g + 4 * hi32;
                    float v[4];
                    const float4v bv = *(const float4v*)(lbias + (n - n0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_t<ACT>(acc[i][j][g * 4 + e] + bv[e]);
                    typedef __attribute__((ext_vector_type(2))) _Float16 half2v;
                    half2v h0 = {(half_t)v[0], (half_t)v[1]}, h1 = {(half_t)v[2], (half_t)v[3]};
                    w[gg][0] = __builtin_bit_cast(unsigned, h0);
                    w[gg][1] = __builtin_bit_cast(unsigned, h1);
                }
                // v_permlane32_swap: lanes 0-31 end up with channels 8gA..8gA+7, lanes 32-63 with 8gB..8gB+7 of their pixel
                auto r0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                typedef __attribute__((ext_vector_type(4))) unsigned uint4v;
                uint4v pk = {r0[0], r1[0], r0[1], r1[1]};
                *(uint4v*)(smem + pix * OROW + (nl + 8 * (gp * 2 + hi32)) * 2) = pk;
            }
        }
    }
    });
    __syncthreads();
    {
        typedef __attribute__((ext_vector_type(4))) unsigned uint4v;
        constexpr int CPP = BN / 8;                 // 16-byte chunks per pixel
        constexpr int NCH = BM * CPP;
        half_t* outp = (half_t*)p.out;
        // NCH / kNT pieces of 16 bytes per thread, in groups: a group's LDS reads first (every address is inside the tile), then its stores -- left alone the compiler
        // put every read right in front of its store behind s_waitcnt lgkmcnt(0) (round 4: y7t_conv_patch.hip, same change, -0.6 % of its layers)
        constexpr int PER = NCH / kNT, GRP = PER % 8 == 0 ? 8 : PER % 4 == 0 ? 4 : 1;
        static_assert(NCH % kNT == 0, "whole pieces per thread");
#pragma unroll 1
        for (int g0 = 0; g0 < PER; g0 += GRP) {
            uint4v v[GRP];
#pragma unroll
            for (int k = 0; k < GRP; ++k) {
                const int c = tid + (g0 + k) * kNT, pix = c / CPP, ch = c - pix * CPP;
                v[k] = *(const uint4v*)(smem + pix * OROW + ch * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < GRP; ++k) {
                const int c = tid + (g0 + k) * kNT, pix = c / CPP, ch = c - pix * CPP;
                const int m = m0 + pix, n = n0 + ch * 8;
                if (m < p.M && n < p.Cout) *(uint4v*)(outp + (size_t)m * p.ldout + p.cout_off + n) = v[k];
            }
        }
    }
#endif
}

// sum the split-K slabs in split order, add bias, activate, store (fp16 NHWC slice or fp32)
__global__ void __launch_bounds__(256) k_splitk_reduce(const float* __restrict__ partial, int S, int M, int Cout_pad, int Cout, const float* __restrict__ bias,
                                                       int act, void* out, int ldout, int cout_off, int out_f32) {
    const int c4 = Cout_pad / 4;
    const long long tot = (long long)M * c4;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < tot; t += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(t / c4), n = (int)(t - (long long)m * c4) * 4;
        if (n >= Cout) continue;
        typedef __attribute__((ext_vector_type(4))) float float4v;
        // the slabs are summed in split order (deterministic), but their loads are issued four at a time: with one load per trip the loop paid a memory round trip per
        // slab -- S = 4 .. 16 of them in a row in a kernel that has nothing else to do (round 6: 56 of these launches are 13 % of the batch-1 frame)
        const float* p0 = partial + (size_t)m * Cout_pad + n;
        const size_t slab = (size_t)M * Cout_pad;
        float4v a = *(const float4v*)p0;
        int s = 1;
        for (; s + 4 <= S; s += 4) {
            const float4v b0 = *(const float4v*)(p0 + (size_t)s * slab), b1 = *(const float4v*)(p0 + (size_t)(s + 1) * slab);
            const float4v b2 = *(const float4v*)(p0 + (size_t)(s + 2) * slab), b3 = *(const float4v*)(p0 + (size_t)(s + 3) * slab);
            a += b0; a += b1; a += b2; a += b3;
        }
        for (; s < S; ++s) a += *(const float4v*)(p0 + (size_t)s * slab);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = act_fn(a[e] + bias[n + e], act);
        const size_t o = (size_t)m * ldout + cout_off + n;
        if (out_f32) { for (int e = 0; e < 4; ++e) if (n + e < Cout) ((float*)out)[o + e] = v[e]; }
        else if (n + 3 < Cout) { half4 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]}; *(half4*)((half_t*)out + o) = h; }
        else { for (int e = 0; e < 4; ++e) if (n + e < Cout) ((half_t*)out)[o + e] = (half_t)v[e]; }
    }
}

static float* g_splitk_ws = nullptr;
static const size_t kSplitKWsBytes = Y7T_SPLITK_WS_BYTES;

template <int BM, int BN, int BK, int NST, bool UT, int KM = 0, int EPI = 0, bool DUAL = false, int NW = kNW>
static int launch_conv_ut(const Y7TConvArgs& a, hipStream_t s) {
    constexpr unsigned lds_stage = NST * (BM + BN) * BK * 2, lds_epi = EPI == 1 ? BM * (BN + 1) * 4 : BM * (BN * 2 + 16);
    constexpr unsigned lds = (lds_stage > lds_epi ? lds_stage : lds_epi) + BN * 4;   // + the bias corner
    static Y7TOncePerDevice attr;      // (the attribute is per device: ADVICE r4)
    if (int e_ = y7t_once_per_device(attr, [&]() -> int {
        Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv_igemm<BM, BN, BK, NST, UT, KM, EPI, DUAL, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        return 0;
    })) return e_;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.Cout_pad / BN, tiles = tiles_m * tiles_n;
    Y7TConvArgs b = a;
    const int nk = a.K_pad / BK;
    int S = 1;
    if (a.allow_splitk && EPI == 0 && tiles < 256 && nk >= 8) {      // (below 512 tiles, measured in round 4: the 20x20 1x1 layers 34 -> 57 us, 61 -> 67 us)
        S = (512 + tiles - 1) / tiles;
        if (S > nk / 4) S = nk / 4;
        if (S > 16) S = 16;
        while (S > 1 && (size_t)S * a.M * a.Cout_pad * 4 > kSplitKWsBytes) --S;
    }
    b.splitk = S; b.ksteps = (nk + S - 1) / S; b.partial = nullptr;
    if (S > 1) {
        if (!a.splitk_ws && !g_splitk_ws) Y7T_HIP_CHECK(hipMalloc((void**)&g_splitk_ws, kSplitKWsBytes));
        b.partial = a.splitk_ws ? a.splitk_ws : g_splitk_ws;
        b.splitk = (nk + b.ksteps - 1) / b.ksteps;      // no empty splits
    }
    hipLaunchKernelGGL((k_conv_igemm<BM, BN, BK, NST, UT, KM, EPI, DUAL, NW>), dim3(tiles * b.splitk), dim3(64 * NW), lds, s, b);
    Y7T_LAUNCH_CHECK();
    y7t_note_kernel("igemm<%d,%d,%d,%d>%s%s%s%s%s", BM, BN, BK, NST, KM == 1 ? " 1x1" : UT ? "" : " ragged-K", b.splitk > 1 ? " splitK" : "",
                    EPI == 1 ? " detect-decode" : "", DUAL ? " upsample-on-read" : "", NW != kNW ? " 8-wave" : "");
    if (b.splitk > 1) {
        const long long tot = (long long)a.M * (a.Cout_pad / 4);
        int blocks = (int)((tot + 255) / 256); if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(k_splitk_reduce, dim3(blocks), dim3(256), 0, s, (const float*)b.partial, b.splitk, a.M, a.Cout_pad, a.Cout, a.bias, a.act, a.out,
                           a.ldout, a.cout_off, a.out_f32);
        Y7T_LAUNCH_CHECK();
    }
    return 0;
}

template <int BM, int BN, int BK, int NST, int NW = kNW>
static int launch_conv(const Y7TConvArgs& a, hipStream_t s) {
    // uniform-tap specialisation: every K-step of BK channels lies inside one filter tap; 1x1 fast path on top of it
    if (a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.Cin % 64 == 0 && a.in_bytes <= 0xFF000000u - (1u << 24))
        return launch_conv_ut<BM, BN, BK, NST, true, 1, 0, false, NW>(a, s);
    return (a.Cin % BK == 0) ? launch_conv_ut<BM, BN, BK, NST, true>(a, s) : launch_conv_ut<BM, BN, BK, NST, false>(a, s);
}

static int conv_variant() {
    static int v = -1;
    if (v < 0) v = y7t_exp_switch("Y7T_CONV_VARIANT", 0);
    return v;
}

static int conv_dispatch(const Y7TConvArgs& a, hipStream_t s);

int y7t_conv_launch(const Y7TConvArgs& a, hipStream_t s) {
    if (a.Cin % 8 || a.ldin % 8 || a.cin_off % 8 || a.K_pad % 64 || a.Cout_pad % 64 || (!a.out_f32 && (a.ldout % 4 || a.cout_off % 4))) {
        y7t_set_error("conv: unsupported alignment (Cin=%d ldin=%d cin_off=%d K_pad=%d Cout_pad=%d ldout=%d cout_off=%d)", a.Cin, a.ldin,
                      a.cin_off, a.K_pad, a.Cout_pad, a.ldout, a.cout_off);
        return Y7T_E_ARG;
    }
    if (a.KW != 1 && a.KW != 3) { y7t_set_error("conv: kernel width %d unsupported (1 or 3)", a.KW); return Y7T_E_ARG; }
    if ((long long)a.B * a.H * a.W * a.ldin * 2 >= (1ll << 31) || (long long)a.Cout_pad * a.K_pad * 2 >= (1ll << 31)) {
        y7t_set_error("conv: tensor exceeds the 2 GiB range of 32-bit buffer offsets (B=%d H=%d W=%d ld=%d)", a.B, a.H, a.W, a.ldin);
        return Y7T_E_ARG;
    }
    Y7TConvArgs b = a;
    b.in_bytes = (unsigned)((long long)a.B * a.H * a.W * a.ldin * 2);
    b.w_bytes = (unsigned)((long long)a.Cout_pad * a.K_pad * 2);
    b.in2_bytes = a.up_C > 0 ? (unsigned)((long long)a.B * (a.H / 2) * (a.W / 2) * a.ldin2 * 2) : 0u;
    { static int xs = -1; if (xs < 0) xs = y7t_exp_switch("Y7T_CONV_XCD", 1); b.xcd_swizzle = xs; }
    { static int sk = -1; if (sk < 0) sk = y7t_exp_switch("Y7T_CONV_SPLITK", 1); b.allow_splitk = sk; }
    b.splitk = 1; b.ksteps = b.K_pad; b.partial = nullptr;
    { static int to = -1; if (to < 0) to = y7t_exp_switch("Y7T_CONV_TILE_ORDER", 1); b.tile_order = to; }
    { static int ab = -1; if (ab < 0) ab = y7t_exp_switch("Y7T_CONV_ABLATE", 0); b.ablate = ab; }      // (always 0 in the product library)
    return conv_dispatch(b, s);
}

int y7t_conv_patch_try(const Y7TConvArgs& a, hipStream_t s);   // y7t_conv_patch.hip
int y7t_conv_patch_s2_launch(const Y7TConvArgs& a, hipStream_t s);   // y7t_conv_patch_s2.hip (korder 4: opt-in experiment, Y7T_CONV_PATCH_S2=1)
int y7t_conv_ws_launch(const Y7TConvArgs& a, hipStream_t s);         // y7t_conv_ws.hip (korder 5: 64 -> 64 3x3 layers, weights stationary in registers)
int y7t_conv_ws128_launch(const Y7TConvArgs& a, hipStream_t s);      // y7t_conv_ws128.hip (korder 6: 128 -> 128 k 3x3 layers, the same idea on the tile counter)
int y7t_conv_ws_s2_launch(const Y7TConvArgs& a, hipStream_t s);      // y7t_conv_ws_s2.hip (korder 8: the 64 -> 128 3x3 / stride 2 layer, weights stationary in registers)
int y7t_conv_p8_launch(const Y7TConvArgs& a, hipStream_t s);         // y7t_conv_p8.hip (korder 7: 1x1 layers with Cout % 256 == 0 on the 256 x 256 x 64 ping-pong pipeline)

static int conv_dispatch(const Y7TConvArgs& a0, hipStream_t s) {
    Y7TConvArgs a = a0;
    if (a.korder == 9) { a.korder = 2; a.panel64 = 1; }          // the patch kernel's panel order with 64-row panels
    if (a.korder == 10) { a.korder = 3; a.panel64 = 1; }         // the 1x1 panel order with 64-row panels (small maps: twice the workgroups)
    if (a.korder == 8 || a.korder == 11) return y7t_conv_ws_s2_launch(a, s);      // stride-2 register-fragment order (11: with the twin 1x1 behind it): only that kernel reads it
    if (a.korder == 7) return y7t_conv_p8_launch(a, s);         // 256 x 64 weight panels: only that kernel reads them (plain and upsample-on-read)
    if (a.epi || a.up_C > 0) {   // fused Detect epilogue / upsample-on-read loader: instances of the 1x1 fast path only
        if (a.panel64) { y7t_set_error("conv: korder 10 (64-row 1x1 panels) is for plain 1x1 layers, not Detect-decode / upsample-on-read"); return Y7T_E_ARG; }
        const bool fast = a.KH == 1 && a.KW == 1 && a.stride == 1 && a.pad == 0 && a.Cin % 64 == 0 && a.in_bytes <= 0xFF000000u - (1u << 24) &&
                          (a.korder == 3 || a.korder == 0);
        if (!fast) { y7t_set_error("conv: Detect-decode / upsample-on-read need a 1x1 stride-1 layer with Cin %% 64 == 0"); return Y7T_E_ARG; }
        if (a.epi) {
            if (a.Cout_pad != 64 || a.up_C > 0) { y7t_set_error("conv: fused Detect decode needs na * (5 + nc) <= 64 output channels (got %d)", a.Cout); return Y7T_E_ARG; }
            // round 6: a Detect conv has ONE channel tile and no split-K (its epilogue decodes), so on the small maps a workgroup walks the whole K = 768 / 1024 with one
            // K-step in flight per round trip -- 33 us for 0.04 GFLOP-per-frame layers at 40 frames, 28 us at one frame.  A four-stage ring keeps three K-steps in flight
            // (48 KiB of LDS: three workgroups per CU instead of four, which these grids of <= 2000 workgroups do not miss).  The 160 x 160 level at 40 frames (8000
            // tiles, HBM-bound) keeps the two-stage ring.  Measuring build: Y7T_CONV_DETECT_NST=2 is the old form.
            static const int deep = y7t_exp_switch("Y7T_CONV_DETECT_NST", 4);
            if (deep == 4 && (a.M + 127) / 128 <= 4096) return launch_conv_ut<128, 64, 32, 4, true, 1, 1, false>(a, s);
            return launch_conv_ut<128, 64, 32, 2, true, 1, 1, false>(a, s);
        }
        if (a.up_c0 % 32 || a.up_C % 32 || a.up_c0 + a.up_C > a.Cin || (a.H & 1) || (a.W & 1) || a.ldin2 % 8 || a.cin2_off % 8) {
            y7t_set_error("conv: upsample-on-read channel range [%d, %d) / map %dx%d not supported", a.up_c0, a.up_c0 + a.up_C, a.H, a.W);
            return Y7T_E_ARG;
        }
        return a.Cout_pad % 128 == 0 ? launch_conv_ut<128, 128, 32, 2, true, 1, 0, true>(a, s) : launch_conv_ut<128, 64, 32, 2, true, 1, 0, true>(a, s);
    }
    if (a.korder == 4) return y7t_conv_patch_s2_launch(a, s);   // stride-2 LDS-patch kernel's panels: only that kernel reads them
    if (a.korder == 5) return y7t_conv_ws_launch(a, s);         // register-fragment order: only the weights-stationary kernel reads it
    if (a.korder == 6) return y7t_conv_ws128_launch(a, s);      // ... and its 128-channel sibling
    if (a.korder == 3) {   // panel-packed 1x1 weights: only the 32-deep generic kernel reads that layout
        if (a.KH != 1 || a.KW != 1 || a.Cin % 32) { y7t_set_error("conv: korder 3 (panel-packed weights) needs a 1x1 layer with Cin %% 32 == 0"); return Y7T_E_ARG; }
        return a.Cout_pad % 128 == 0 && !a.panel64 ? launch_conv<128, 128, 32, 2>(a, s) : launch_conv<128, 64, 32, 2>(a, s);
    }
    if ((conv_variant() == 0 && !a.no_patch) || a.korder == 2) {   // 3x3 / stride 1 on a large map: LDS-resident patch kernel
        const int rc = y7t_conv_patch_try(a, s);
        if (rc) return rc < 0 ? rc : 0;
    }
    const bool wide = a.Cout_pad % 128 == 0;
    const int var = conv_variant();
    switch (var > 8 ? 0 : var) {
#if Y7T_ABLATE      // the tile / ring variants of the generic kernel (rounds 1-3 sweeps, scripts/sweep_conv.py): liby7t_ablate.so only
    case 1: return wide ? launch_conv<128, 128, 64, 3>(a, s) : launch_conv<128, 64, 64, 3>(a, s);
    case 2: return wide ? launch_conv<128, 128, 32, 3>(a, s) : launch_conv<128, 64, 32, 3>(a, s);
    case 3: return wide ? launch_conv<128, 128, 32, 4>(a, s) : launch_conv<128, 64, 32, 4>(a, s);
    case 4: return wide ? launch_conv<128, 128, 32, 2>(a, s) : launch_conv<128, 64, 32, 2>(a, s);
    case 5: return wide ? launch_conv<256, 128, 32, 2>(a, s) : launch_conv<256, 64, 32, 2>(a, s);
    case 6: return wide ? launch_conv<256, 128, 64, 2>(a, s) : launch_conv<256, 64, 64, 2>(a, s);
    case 8: return a.Cout_pad % 256 == 0 ? launch_conv<256, 256, 64, 2>(a, s) : wide ? launch_conv<256, 128, 64, 2>(a, s) : launch_conv<256, 64, 64, 2>(a, s);
    case 7: return wide ? launch_conv<128, 128, 64, 2>(a, s) : launch_conv<128, 64, 64, 2>(a, s);
#endif
    default: {
        const bool big256 = (long long)(a.M / 256) * (a.Cout_pad / (wide ? 128 : 64)) >= 2048;   // enough 256-pixel tiles to fill the chip 4x
        // measured per-layer (scripts/bench_conv.py): the HBM-bound 1x1 layers prefer the lighter 32-deep stages (4 blocks/CU),
        // the 3x3 layers the 64-deep ones
        if (a.KH == 1) return wide ? launch_conv<128, 128, 32, 2>(a, s) : launch_conv<128, 64, 32, 2>(a, s);
        // (both rules were measured at 32 frames per forward; at small batch the 128-pixel tiles' larger grid wins)
        // stem (Cin = 16, K = 144): three K-steps per tile -- 256-pixel tiles halve the per-tile set-up and epilogue count
        if (a.Cin <= 16 && !wide && big256) return launch_conv<256, 64, 32, 2>(a, s);
        // stride-2 3x3 on the large maps: 256-pixel tiles with 32-deep stages measured 5-7 % ahead (profiles/r01_conv_variants.txt)
        if (a.stride == 2 && big256) return wide ? launch_conv<256, 128, 32, 2>(a, s) : launch_conv<256, 64, 32, 2>(a, s);
        // small maps whose 128-channel tiles do not fill the chip but whose 64-channel tiles do (the 20 x 20 256-channel 3x3 layers at 32 frames: 200 vs 400 tiles):
        // twice the workgroups instead of split-K partial sums + a reduce launch: 229 -> 191 us for those four launches (profiles/r04_small_experiments.txt;
        // Y7T_CONV_NARROW=0: the split-K path)
        static const int narrow = y7t_exp_switch("Y7T_CONV_NARROW", 1);
        const int tm = (a.M + 127) / 128;
        if (narrow && wide && tm * (a.Cout_pad / 128) < 256 && tm * (a.Cout_pad / 64) >= 256) return launch_conv<128, 64, 64, 2>(a, s);
        return wide ? launch_conv<128, 128, 64, 2>(a, s) : launch_conv<128, 64, 64, 2>(a, s);
    }
    }
}
