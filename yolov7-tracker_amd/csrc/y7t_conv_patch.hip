// y7t_conv_patch.hip -- 3x3 / stride 1 / pad 1 Conv(+folded BN)+bias+activation with an LDS-resident input PATCH.
//
// Same math as k_conv_igemm (y7t_conv.hip; /root/reference/models/common.py:99-111 after utils/torch_utils.py:181-201), other
// data movement.  The generic kernel fetches every (pixel, tap) row of the implicit im2col matrix separately: 512 bytes of
// buffer->LDS DMA per MFMA, which at full matrix rate is the whole 64 B/clk/CU of the texture addresser.
// Here a workgroup owns a 2-D tile of 256 output pixels (16x16 or 32 wide x 8 high) x BN channels and keeps, per 32-channel
// chunk of the input, the (TH+2) x (TW+2) pixel patch in LDS ONCE; the nine taps are nine K-steps that read the SAME patch at
// shifted addresses.  Per K-step only the BN x 32 weight panel (and 1/9 of the next patch) comes in:
//        DMA bytes per MFMA:  512 -> 208 (BN=128) / 256 (BN=64);   ds_read_b128 per MFMA: 1 -> 0.75 (BN=128)
// and the K loop contains no address arithmetic at all:
//   * LDS rows (one pixel's 32 channels, 64 B) are PADDED to 80 B instead of XOR-swizzled, so every fragment address is
//     lane_base + compile-time immediate (tap shift, tile row, k-substep all fold into the ds_read offset field).  80-byte rows
//     are conflict-free for ds_read_b128 (slot = 5*pixel + chunk mod 16 is a bijection over the 16 lanes of a service group);
//     for the 16-wide tile the patch-row pitch is a multiple of 256 B so that the two image rows of one MFMA tile interleave.
//   * DMA sources are per-lane byte offsets computed once per workgroup (image border and the 16 pad bytes of each row =
//     out-of-range offset -> hardware zero fill); chunk and tap advance through the instruction's SCALAR offset.
// K order inside the kernel: 32-channel chunk outermost, tap innermost (weights are addressed in place in either packing,
// Y7TConvArgs::korder).  Two chunks (18 K-steps) are unrolled so that ring positions are compile-time.
#include "y7t_common.h"
#include "y7t_conv_common.h"
#include <stdlib.h>

namespace {

constexpr unsigned kOOB = 0xFF000000u;   // voffset of a zero-filled lane: out of range with or without the (< 16 MiB) scalar offset

template <int TW, int TH, int BN>
struct PatchCfg {
    static constexpr int PIXB = 80;                                       // bytes per patch pixel in LDS (64 data + 16 pad)
    // TW == 0: FLAT tiling for narrow maps (W = 20, 40).  The batch is one long strip of zero-padded images, (H+2) rows of
    // PW = W+2 positions each; a workgroup owns 256 CONSECUTIVE positions of the strip (padding positions are computed and dropped:
    // W*H / ((W+2)*(H+2)) useful, 91 % at 40x40, where 16x16 tiles would be 69 %), its patch is the same strip PW+1 positions
    // longer at both ends, and tap (kh, kw) is the shift kh*PW + kw.  TH carries PW.
    static constexpr bool FLAT = (TW == 0);
    static constexpr int PW = TH;
    static constexpr int NPOS = 256 + 2 * PW + 2;                         // FLAT: patch positions
    static constexpr int RP = FLAT ? PW * PIXB : (TW == 16) ? 1536 : (TW + 2) * PIXB;   // patch row pitch (= tap row shift)
    static constexpr int PATCH_DMA = FLAT ? (NPOS * PIXB + 1023) / 1024 : ((TH + 2) * RP + 1023) / 1024;   // wave-wide 1 KiB DMAs per patch
    static constexpr int NPX = (PATCH_DMA + 1) / 2;                       // patch DMAs per PATCH wave (waves 2, 3; a slot past the patch
    static constexpr int PATCH_BYTES = PATCH_DMA * 1024;                  //  repeats the patch's last KiB: same bytes to the same place)
    static constexpr int PPT = (NPX + 6) / 7;                             // pieces per patch wave per tap (taps 0..6)
    static constexpr int WROWB = 64;                                      // weight rows: 64 B, 16-byte slots XOR-swizzled with (row >> 2) & 3
    static constexpr int W_DMA = BN * WROWB / 1024;                       // 8 (BN=128) / 4 (BN=64)
    static constexpr int NWX = W_DMA / 2;                                 // weight DMAs per WEIGHT wave (waves 0, 1) per K-step
    static constexpr int W_BYTES = W_DMA * 1024;
    static constexpr int NWS = 3;                                         // weight ring: K-step u+2 is in flight while u is multiplied
    static constexpr int W_OFF = 0, P_OFF = NWS * W_BYTES;                // LDS map: W ring | patch A | patch B
    static constexpr int LDS_LOOP = P_OFF + 2 * PATCH_BYTES;
    static constexpr int OROW = BN * 2 + 16;
    static constexpr int LDS_EPI = 256 * OROW;
    static constexpr int BIAS_OFF = LDS_LOOP > LDS_EPI + 1024 ? LDS_LOOP : LDS_EPI + 1024;   // (+ the FLAT epilogue's 256-entry pixel table)
    static constexpr int LDS = BIAS_OFF + BN * 4;                         // this workgroup's BN biases, fetched in the prologue
    static constexpr int WN = BN / 64, WM = 4 / WN;                       // waves along channels / pixels
    static constexpr int TM = 8 / WM;                                     // 32-pixel MFMA tiles per wave (4 or 2)
    static constexpr int RPT = (TW == 16) ? 2 : 1;                        // image rows per MFMA tile
    static constexpr int JOFF = FLAT ? 32 * PIXB : RPT * RP;              // LDS distance between a wave's consecutive 32-pixel MFMA tiles
};

// the DMAs of one K-step (a macro, not a lambda: the ABL = 0 instances must compile to exactly the code that was measured).
// ABL bit 9 (512) is not an ablation but an ORDER experiment with correct results (Y7T_CONV_ABLATE=512; next round's measurement): the step's DMAs go out
// behind its second MFMA half instead of in front of the fragment reads -- a buffer->LDS piece costs its wave most in a phase that also carries ds_reads
// (MI355X_MICROARCH.md), and in that order the matrix pipe has the half queued while the pieces go out.
#define Y7T_PATCH_ISSUE_DMAS() \
            { \
                constexpr bool live = !(ABL & 1); \
                if (wrole) { \
                    const int un = u + 3; \
                    const int ccn = (un / 9), tn = un % 9; \
                    const int cn = c0 + ccn; \
                    if (ABL & 16) issue_w(u % 3, 0, 0, 0, true); \
                    else issue_w(u % 3, ccn == 2 ? cbase + s_pair : cbase + ccn * s_odd, tn / 3, tn % 3, cn < nc32 && live); \
                } else if (t < 7) { \
                _Pragma("unroll") \
                    for (int q = 0; q < C::PPT; ++q) \
                        if (t * C::PPT + q < C::NPX) issue_patch_piece(cc ^ 1, c + 1, t * C::PPT + q, c + 1 < nc32 && live && !(ABL & 32)); \
                } \
            }
// ABL: compile-time ablation bits for scripts/sweep_conv.py (Y7T_CONV_ABLATE): 1 zero-filling DMAs only, 2 no MFMAs, 4 no fragment
// reads, 8 no epilogue.  (Run-time switches inside the K loop change its schedule by tens of percent -- hence template instances.)
template <int TW, int TH, int BN, int ABL>
__global__ void __launch_bounds__(256, 2) k_conv3x3_patch(const Y7TConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C = PatchCfg<TW, TH, BN>;
    static_assert(C::FLAT || TW * TH == 256, "256 output pixels per workgroup");
    constexpr int PIXB = C::PIXB, RP = C::RP, TM = C::TM, RPT = C::RPT;
    constexpr bool FLAT = C::FLAT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave id in an SGPR: DMA destinations are scalar
    const int wn = (C::WN == 2) ? (wave >> 1) : 0, wm = (C::WN == 2) ? (wave & 1) : wave;
    const int l31 = lane & 31, hi32 = lane >> 5;

    // ---- tile decode: channel tiles fastest (workgroups sharing a patch run next to each other), XCD-contiguous ranges ----
    int bid = blockIdx.x;
    if (p.xcd_swizzle) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int n_tiles_n = p.Cout_pad / BN;
    const int tile_n = bid % n_tiles_n;
    int pt = bid / n_tiles_n;
    int b = 0, h0 = 0, w0 = 0;
    const int g0 = pt * 256;                                   // FLAT: first strip position of this workgroup
    const int img_pos = (p.H + 2) * C::PW, strip = p.B * img_pos;
    if (!FLAT) {
        constexpr int TWd = FLAT ? 1 : TW;
        const int tiles_x = (p.W + TWd - 1) / TWd, tiles_y = (p.H + TH - 1) / TH;
        const int txi = pt % tiles_x; pt /= tiles_x;
        const int tyi = pt % tiles_y;
        b = pt / tiles_y; h0 = tyi * TH; w0 = txi * TWd;
    }
    const int n0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    // ---- per-lane DMA sources (computed once).  The DMA streams are split by wave: waves 0/1 fetch the weight panels, waves 2/3 the
    // patch.  vmcnt retires in order, so a wave that mixed both would have to see its (HBM-latency) patch pieces land before it
    // could prove a younger (L2-latency) weight panel complete -- Y7T_CONV_ABLATE=32 showed that costing 20 % of the layer. ----
    const bool wrole = wave < 2;
    constexpr int NOFF = C::NPX > C::NWX ? C::NPX : C::NWX;
    unsigned off[NOFF];
#pragma unroll
    for (int i = 0; i < NOFF; ++i) {
        unsigned v = kOOB;
        if (wrole) {
            if (i < C::NWX) {
                const int byte = (wave * C::NWX + i) * 1024 + lane * 16;
                const int row = byte >> 6, slot = (byte >> 4) & 3;
                if (p.korder == 2) v = (unsigned)(tile_n * (p.K_pad >> 5) * C::W_BYTES + byte);   // panel order: memory image == LDS image
                else v = (unsigned)(((n0 + row) * p.K_pad + ((slot ^ ((row >> 2) & 3)) << 3)) * 2);
            }
        } else if (i < C::NPX) {
            int I = (wave - 2) + 2 * i;
            if (I >= C::PATCH_DMA) I = C::PATCH_DMA - 1;
            const int byte = I * 1024 + lane * 16;
            if (FLAT) {
                const int q = byte / PIXB, cs = (byte - q * PIXB) >> 4;
                const int g = g0 - C::PW - 1 + q;                   // strip position of patch slot q
                if (cs < 4 && q < C::NPOS && g >= 0 && g < strip) {
                    const int bb = g / img_pos, rem = g - bb * img_pos;
                    const int yy = rem / C::PW, xx = rem - yy * C::PW;
                    if (yy >= 1 && yy <= p.H && xx >= 1 && xx <= p.W)
                        v = (unsigned)(((((bb * p.H + yy - 1) * p.W + xx - 1) * p.ldin + p.cin_off) + cs * 8) * 2);
                }
            } else {
            const int r = byte / RP, rb = byte - r * RP;
            const int x = rb / PIXB, cs = (rb - x * PIXB) >> 4;
            const int gy = h0 + r - 1, gx = w0 + x - 1;
            const bool ok = r < TH + 2 && x < TW + 2 && cs < 4 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            if (ok) v = (unsigned)(((((b * p.H + gy) * p.W + gx) * p.ldin + p.cin_off) + cs * 8) * 2);
            }
        }
        off[i] = v;
    }

    // ---- fragment read bases ----
    // weights: row = wn*64 + i*32 + l31, logical chunk q = ks*2 + hi32 sits in slot q ^ ((row >> 2) & 3) = q ^ ((l31 >> 2) & 3): per-lane constant
    const int wsw = (l31 >> 2) & 3;
    const char* wlane0 = smem + C::W_OFF + (wn * 64 + l31) * C::WROWB + (((0 + hi32) ^ wsw) << 4);
    const char* wlane1 = smem + C::W_OFF + (wn * 64 + l31) * C::WROWB + (((2 + hi32) ^ wsw) << 4);
    const char* plane = FLAT ? smem + C::P_OFF + (wm * TM * 32 + l31) * PIXB + hi32 * 16
                             : smem + C::P_OFF + (wm * TM * RPT + (TW == 16 ? (l31 >> 4) : 0)) * RP + (TW == 16 ? (l31 & 15) : l31) * PIXB + hi32 * 16;

    const int nc32 = p.Cin >> 5;
    // byte offset (in a weight row) of K-step (chunk c, tap kh,kw) = kh * s_kh + kw * s_kw + chunk part; the chunk part of an even
    // chunk c0 is `cbase`, of c0 + 1 it is cbase + 64 in both packings, and a chunk pair advances it by s_pair
    // korder 2 (panel order, detector/weights.py): the BN x 32 panel of K-step (c, tap) is one contiguous, pre-swizzled W_BYTES block
    // at [tile_n][c * 9 + tap] -- 8 full cache lines per DMA instead of 16 half lines from 16 different weight rows, which the
    // vector L1 serves almost 3x faster (scripts/ubench/dma_patterns.hip: 17 vs 46 clocks per wave-DMA)
    const int s_kh = p.korder == 2 ? 3 * C::W_BYTES : p.korder ? (p.Cin >> 6) * 3 * 128 : 3 * p.Cin * 2;
    const int s_kw = p.korder == 2 ? C::W_BYTES : p.korder ? 128 : p.Cin * 2;
    const int s_odd = p.korder == 2 ? 9 * C::W_BYTES : 64;                  // odd chunk of a pair relative to the even one
    const int s_pair = p.korder == 2 ? 18 * C::W_BYTES : p.korder ? 3 * 128 : 128;
    int cbase = 0;
    auto issue_w = [&](int slot, int coff, int kh, int kw, bool real) {
        const int so = kh * s_kh + kw * s_kw + coff;
#pragma unroll
        for (int i = 0; i < C::NWX; ++i)   // weight waves only
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (LDS_AS void*)(smem + C::W_OFF + slot * C::W_BYTES + (wave * C::NWX + i) * 1024), 16,
                                                     real ? off[i] : kOOB, real ? so : 0, 0, 0);
    };
    auto issue_patch_piece = [&](int pb, int c, int i, bool real) {   // patch waves only
        const int I = ((wave - 2) + 2 * i < C::PATCH_DMA) ? (wave - 2) + 2 * i : C::PATCH_DMA - 1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (LDS_AS void*)(smem + C::P_OFF + pb * C::PATCH_BYTES + I * 1024), 16,
                                                 real ? off[i] : kOOB, c << 6, 0, 0);
    };

    floatx16 acc[2][TM];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // the epilogue's biases: fetched now, beside the first DMAs, into an LDS corner no stage ever touches (a global load at the
    // end of a workgroup that lives for a few microseconds is pure exposed latency)
    if (tid < BN) ((float*)(smem + C::BIAS_OFF))[tid] = p.bias[n0 + tid];
    // prologue: patch of chunk 0 and the weights of K-steps 0..2, then the fragments of K-step 0
    if (wrole) {
        issue_w(0, 0, 0, 0, true);
        issue_w(1, 0, 0, 1, true);
        issue_w(2, 0, 0, 2, true);
    } else {
#pragma unroll
        for (int i = 0; i < C::NPX; ++i) issue_patch_piece(0, 0, i, true);
    }

    static_assert(C::PPT * 7 >= C::NPX, "patch pieces fit into taps 0..6");
    static_assert(C::LDS <= 81920, "two workgroups per CU");
    half8 wf[2][2][2], xf[2][2][TM];   // [register buffer][k-substep][tile]
    auto read_frags = [&](int buf, int slot, int pb, int kh, int kw) {
        const char* ws0 = wlane0 + slot * C::W_BYTES;
        const char* ws1 = wlane1 + slot * C::W_BYTES;
        const char* ps = plane + pb * C::PATCH_BYTES + kh * RP + kw * PIXB;
        if (ABL & 4) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 2; ++i) asm volatile("" : "=v"(wf[buf][ks][i]));
#pragma unroll
                for (int j = 0; j < TM; ++j) asm volatile("" : "=v"(xf[buf][ks][j]));
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) wf[buf][ks][i] = *(const half8*)((ks ? ws1 : ws0) + i * 32 * C::WROWB);
#pragma unroll
            for (int j = 0; j < TM; ++j) xf[buf][ks][j] = *(const half8*)(ps + j * C::JOFF + ks * 32);
        }
    };
    auto mfma_half = [&](int buf, int ks) {
        if (ABL & 2) {
#pragma unroll
            for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(wf[buf][ks][i]));
#pragma unroll
            for (int j = 0; j < TM; ++j) asm volatile("" ::"v"(xf[buf][ks][j]));
            return;
        }
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[buf][ks][i], xf[buf][ks][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    if (wrole) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * C::NWX) : "memory");   // W(0) landed (W(1), W(2) may be in flight)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             // patch(0) landed
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, 0, 0, 0);

    // Software pipeline, K-step u = (chunk c, tap t); the workgroup barrier sits in the MIDDLE of the step's MFMAs:
    //     MFMAs of k-substep 0 (fragments were read during step u-1)
    //     wait until W(u+1) landed, barrier      -- behind it W(u+1) and, at a chunk's last tap, the next patch are visible to all,
    //                                               and every wave is done reading W(u) and the fragments of step u
    //     issue the DMAs of W(u+3) (ring slot of W(u)) and one piece of patch(c+1); read ALL fragments of step u+1
    //     MFMAs of k-substep 1                   -- they cover the latency of those reads and of the barrier
    // A weight wave always has exactly W(u+2) younger than the panel it waits for (tail steps issue zero-filling DMAs); a patch
    // wave waits only at a chunk's last tap, for everything.
    // The step's fragment reads are SPREAD over its MFMAs instead of issued as one burst of 12 behind the barrier (the form of rounds 1-3a; kept for the timing
    // ablations, whose instances carry other ABL bits, and selectable with Y7T_CONV_ABLATE=2048 for A/B runs).  Measured (profiles/r03_patch_ablations.txt, same session,
    // 32 frames): 160^2 128->128 258.8 -> 250.8 us, 80^2 256->256 225.1 -> 221.8, 40^2 384->384 148.3 -> 143.1, 160^2 128->256 465.5 -> 448.0.  The timing ablations of the current form (scripts/patch_ablations.sh, profiles/r03_patch_ablations.txt)
    // price the fragment reads at 22-25 % of the layer although LDS is only ~40 % busy on average: eight waves x 12 KiB right behind every barrier take ~770 LDS
    // cycles to serve, longer than the second MFMA half that is meant to cover them.  Here the 8 patch fragments of step u+1 go out one per MFMA during the FIRST half
    // of step u (the patch is resident for the whole chunk; only at a chunk's last tap the next patch is not visible before the barrier), the 4 weight fragments
    // behind the first MFMAs of the second half (their panel W(u+1) is visible after the barrier), and the wait in front of the barrier counts the patch reads
    // that may still be in flight (LDS returns in order: everything older -- the reads of W(u), whose ring slot is refilled behind the barrier -- has returned).
    for (int c0 = 0; c0 < nc32; c0 += 2) {
#pragma unroll
        for (int u = 0; u < 18; ++u) {
            const int cc = u / 9, t = u % 9;   // compile-time after unrolling
            const int c = c0 + cc, cur = u & 1;
            if (ABL == 0) {
                const int un = u + 1, tn = un % 9, kh = tn / 3, kw = tn % 3;
                const char* ps = plane + ((un / 9) & 1) * C::PATCH_BYTES + kh * RP + kw * PIXB;
                const char* ws0 = wlane0 + (un % 3) * C::W_BYTES;
                const char* ws1 = wlane1 + (un % 3) * C::W_BYTES;
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int q = 0; q < 2 * TM; ++q) {           // first half: MFMA q, then patch fragment q of the next step
                    const int i = q / TM, j = q % TM;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cur][0][i], xf[cur][0][j], acc[i][j], 0, 0, 0);
                    if (t < 8) xf[cur ^ 1][q / TM][q % TM] = *(const half8*)(ps + (q % TM) * C::JOFF + (q / TM) * 32);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_setprio(0);
                if (t < 8) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * TM) : "memory");
                else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (wrole) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NWX) : "memory");
                else if (t == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                Y7T_PATCH_ISSUE_DMAS()
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int q = 0; q < 2 * TM; ++q) {           // second half: MFMA q, then a weight fragment (and, at a chunk's last tap, the next patch's fragments)
                    const int i = q / TM, j = q % TM;
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cur][1][i], xf[cur][1][j], acc[i][j], 0, 0, 0);
                    if (q < 4) wf[cur ^ 1][q >> 1][q & 1] = *(const half8*)(((q >> 1) ? ws1 : ws0) + (q & 1) * 32 * C::WROWB);
                    if (t == 8) xf[cur ^ 1][q / TM][q % TM] = *(const half8*)(ps + (q % TM) * C::JOFF + (q / TM) * 32);
                    __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_s_setprio(0);
                continue;
            }
            mfma_half(cur, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (wrole) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(C::NWX) : "memory");
            else if (t == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (!(ABL & 512)) Y7T_PATCH_ISSUE_DMAS()
            {
                const int un = u + 1, tn = un % 9;
                read_frags(cur ^ 1, un % 3, (un / 9) & 1, tn / 3, tn % 3);
            }
            mfma_half(cur, 1);
            if (ABL & 512) Y7T_PATCH_ISSUE_DMAS()
        }
        cbase += s_pair;
    }

    if (ABL & 8) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) asm volatile("" ::"v"(acc[i][j]));
        return;
    }
    // ---- epilogue: bias + activation, transpose through LDS, full-line NHWC stores ----
    constexpr int OROW = C::OROW;
    const float* lbias = (const float*)(smem + C::BIAS_OFF);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    typedef __attribute__((ext_vector_type(4))) float float4v;
    act_dispatch(p.act, [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int pix = (wm * TM + j) * 32 + l31;   // tile-local pixel id: TW=16 -> row = pix >> 4, x = pix & 15; TW=32 -> row = pix >> 5
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int nl = wn * 64 + i * 32;
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                unsigned w[2][2];
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
                auto r0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                typedef __attribute__((ext_vector_type(4))) unsigned uint4v;
                uint4v pk = {r0[0], r1[0], r0[1], r1[1]};
                *(uint4v*)(smem + pix * OROW + (nl + 8 * (gp * 2 + hi32)) * 2) = pk;
            }
        }
    }
    });
    int* otab = (int*)(smem + C::LDS_EPI);   // FLAT: output pixel index of each of the 256 positions (-1: padding / past the strip)
    if (FLAT) {
        const int g = g0 + tid;
        int o = -1;
        if (g < strip) {
            const int bb = g / img_pos, rem = g - bb * img_pos;
            const int yy = rem / C::PW, xx = rem - yy * C::PW;
            if (yy >= 1 && yy <= p.H && xx >= 1 && xx <= p.W) o = (bb * p.H + yy - 1) * p.W + xx - 1;
        }
        otab[tid] = o;
    }
    __syncthreads();
    {
        typedef __attribute__((ext_vector_type(4))) unsigned uint4v;
        constexpr int CPP = BN / 8;
        half_t* outp = (half_t*)p.out;
        // CPP pieces of 16 bytes per thread, in groups of GRP: the group's LDS reads first (every address is inside the tile), then its stores -- left alone the
        // compiler put every read right in front of its store behind s_waitcnt lgkmcnt(0): 8 or 16 LDS round trips in a row at the end of every workgroup
        constexpr int GRP = CPP < 8 ? CPP : 8;
        static_assert(CPP % GRP == 0, "whole groups");
#pragma unroll 1
        for (int c0 = 0; c0 < CPP; c0 += GRP) {
            uint4v v[GRP];
#pragma unroll
            for (int k = 0; k < GRP; ++k) {
                const int cidx = tid + (c0 + k) * 256, pix = cidx / CPP, ch = cidx - pix * CPP;
                v[k] = *(const uint4v*)(smem + pix * OROW + ch * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < GRP; ++k) {
                const int cidx = tid + (c0 + k) * 256, pix = cidx / CPP, ch = cidx - pix * CPP;
                const int n = n0 + ch * 8;
                long long opix;
                if (FLAT) opix = otab[pix];
                else {
                    constexpr int TWd = FLAT ? 1 : TW;
                    const int r = pix / TWd, x = pix - r * TWd;
                    const int gy = h0 + r, gx = w0 + x;
                    opix = (gy < p.H && gx < p.W) ? (long long)(b * p.H + gy) * p.W + gx : -1;
                }
                if (opix >= 0 && n < p.Cout) *(uint4v*)(outp + (size_t)opix * p.ldout + p.cout_off + n) = v[k];
            }
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------
// Multi-tile variant for the 64-channel panels (Cout_pad % 128 != 0: the 64 -> 64 layers at 320x320 / 160x160, K = 576).  Such a
// workgroup lives for 18 K-steps; Y7T_CONV_ABLATE on the single-tile kernel shows 38 % of the layer in the epilogue and 27 % in
// launch + index set-up + the first HBM-latency patch, all exposed.  Here a workgroup walks MT consecutive pixel tiles and the
// chunk stream simply continues across the tile boundary: while the last chunk of tile t is multiplied the first patch of tile t+1
// is already landing (its per-lane sources are computed during tile t), the weight ring keeps turning, and the epilogue writes
// its 16-byte pieces straight from registers to memory -- no LDS staging, so nothing it touches is in the way of those DMAs.
// ---------------------------------------------------------------------------------------------------------------------
template <int TW, int TH, int MT>
__global__ void __launch_bounds__(256, 2) k_conv3x3_patch_mt(const Y7TConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int BN = 64;
    using C = PatchCfg<TW, TH, BN>;
    static_assert(!C::FLAT && TW * TH == 256, "256 output pixels per tile");
    constexpr int PIXB = C::PIXB, RP = C::RP, TM = C::TM, RPT = C::RPT;
    // weight ring: NWS slots, K-step u+WD goes out at step u (WD = NWS reuses the slot whose fragments were read a step ago).
    // A 6-slot ring with 5 steps of look-ahead was measured too: no faster (491 vs 460 us on the 320x320 64->64 layer) -- the loop
    // is not waiting for weight panels.
    constexpr int NWS = 3, WD = 3;
    constexpr int P_OFF = NWS * C::W_BYTES, BIAS_OFF = P_OFF + 2 * C::PATCH_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave;                       // BN = 64: the four waves split the pixels
    const int l31 = lane & 31, hi32 = lane >> 5;
    int bid = blockIdx.x;
    if (p.xcd_swizzle) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int n_tiles_n = p.Cout_pad / BN;
    const int tile_n = bid % n_tiles_n, n0 = tile_n * BN;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH, ptiles = p.B * tiles_y * tiles_x;
    const int pt_first = (bid / n_tiles_n) * MT;
    const int nt = (ptiles - pt_first) < MT ? (ptiles - pt_first) : MT;      // pixel tiles of this workgroup

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const bool wrole = wave < 2;               // waves 0/1 stream the weight panels, waves 2/3 the patches

    // per-lane patch sources of pixel tile `pt` (patch waves) -- kOOB outside the image / past the last tile
    auto patch_off = [&](int pt, unsigned (&o)[C::NPX]) {
        int q = pt;
        const int txi = q % tiles_x; q /= tiles_x;
        const int tyi = q % tiles_y, b = q / tiles_y, h0 = tyi * TH, w0 = txi * TW;
#pragma unroll
        for (int i = 0; i < C::NPX; ++i) {
            int I = (wave - 2) + 2 * i;
            if (I >= C::PATCH_DMA) I = C::PATCH_DMA - 1;
            const int byte = I * 1024 + lane * 16;
            const int r = byte / RP, rb = byte - r * RP;
            const int x = rb / PIXB, cs = (rb - x * PIXB) >> 4;
            const int gy = h0 + r - 1, gx = w0 + x - 1;
            const bool ok = pt < ptiles && r < TH + 2 && x < TW + 2 && cs < 4 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            o[i] = ok ? (unsigned)(((((b * p.H + gy) * p.W + gx) * p.ldin + p.cin_off) + cs * 8) * 2) : kOOB;
        }
    };
    unsigned pc[C::NPX], pn[C::NPX], woff[C::NWX];
#pragma unroll
    for (int i = 0; i < C::NPX; ++i) pc[i] = pn[i] = kOOB;
#pragma unroll
    for (int i = 0; i < C::NWX; ++i) woff[i] = kOOB;
    if (wrole) {
#pragma unroll
        for (int i = 0; i < C::NWX; ++i) {
            const int byte = (wave * C::NWX + i) * 1024 + lane * 16;
            const int row = byte >> 6, slot = (byte >> 4) & 3;
            woff[i] = p.korder == 2 ? (unsigned)(tile_n * (p.K_pad >> 5) * C::W_BYTES + byte)
                                    : (unsigned)(((n0 + row) * p.K_pad + ((slot ^ ((row >> 2) & 3)) << 3)) * 2);
        }
    } else {
        patch_off(pt_first, pc);
        patch_off(pt_first + 1, pn);
    }
    const int wsw = (l31 >> 2) & 3;
    const char* wlane0 = smem + C::W_OFF + l31 * C::WROWB + (((0 + hi32) ^ wsw) << 4);
    const char* wlane1 = smem + C::W_OFF + l31 * C::WROWB + (((2 + hi32) ^ wsw) << 4);
    const char* plane = smem + P_OFF + (wm * TM * RPT + (TW == 16 ? (l31 >> 4) : 0)) * RP + (TW == 16 ? (l31 & 15) : l31) * PIXB + hi32 * 16;

    const int nc32 = p.Cin >> 5, total = nt * nc32;     // chunks of this workgroup
    const int s_kh = p.korder == 2 ? 3 * C::W_BYTES : p.korder ? (p.Cin >> 6) * 3 * 128 : 3 * p.Cin * 2;
    const int s_kw = p.korder == 2 ? C::W_BYTES : p.korder ? 128 : p.Cin * 2;
    auto chunk_off = [&](int c) -> int {                // byte offset of chunk c (inside its tile) in a weight row / panel list
        return p.korder == 2 ? c * 9 * C::W_BYTES : p.korder ? (c >> 1) * 3 * 128 + (c & 1) * 64 : c * 64;
    };
    auto issue_w = [&](int slot, int c_in_tile, int kh, int kw, bool real) {
        const int so = kh * s_kh + kw * s_kw + chunk_off(c_in_tile);
#pragma unroll
        for (int i = 0; i < C::NWX; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (LDS_AS void*)(smem + C::W_OFF + slot * C::W_BYTES + (wave * C::NWX + i) * 1024), 16,
                                                     real ? woff[i] : kOOB, real ? so : 0, 0, 0);
    };
    auto issue_piece = [&](int pb, int i, unsigned voff, int so) {
        const int I = ((wave - 2) + 2 * i < C::PATCH_DMA) ? (wave - 2) + 2 * i : C::PATCH_DMA - 1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (LDS_AS void*)(smem + P_OFF + pb * C::PATCH_BYTES + I * 1024), 16, voff, so, 0, 0);
    };

    floatx16 acc[2][TM];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    if (tid < BN) ((float*)(smem + BIAS_OFF))[tid] = p.bias[n0 + tid];
    const float* lbias = (const float*)(smem + BIAS_OFF);
    if (wrole) {
#pragma unroll
        for (int st = 0; st < WD; ++st) issue_w(st, st / 9, (st % 9) / 3, st % 3, st < total * 9);
    } else {
#pragma unroll
        for (int i = 0; i < C::NPX; ++i) issue_piece(0, i, pc[i], 0);
    }
    half8 wf[2][2][2], xf[2][2][TM];
    auto read_frags = [&](int buf, int slot, int pb, int kh, int kw) {
        const char* ws0 = wlane0 + slot * C::W_BYTES;
        const char* ws1 = wlane1 + slot * C::W_BYTES;
        const char* ps = plane + pb * C::PATCH_BYTES + kh * RP + kw * PIXB;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < 2; ++i) wf[buf][ks][i] = *(const half8*)((ks ? ws1 : ws0) + i * 32 * C::WROWB);
#pragma unroll
            for (int j = 0; j < TM; ++j) xf[buf][ks][j] = *(const half8*)(ps + j * C::JOFF + ks * 32);
        }
    };
    auto mfma_half = [&](int buf, int ks) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[buf][ks][i], xf[buf][ks][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    if (wrole) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WD - 1) * C::NWX) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, 0, 0, 0);

    half_t* outp = (half_t*)p.out;
    int ct = 0, tcur = 0;                       // chunk index inside the current tile (even), tile counter
    for (int c0 = 0; c0 < total; c0 += 2) {
#pragma unroll
        for (int u = 0; u < 18; ++u) {
            const int cc = u / 9, t = u % 9, cur = u & 1;
            mfma_half(cur, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (wrole) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((WD - 2) * C::NWX) : "memory");   // W(u+1) landed; W(u+2..u+WD-1) may fly
            else if (t == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (wrole) {
                const int un = u + WD, ccn = un / 9, tn = un % 9;
                int cin = ct + ccn;              // chunk inside its tile of the K-step whose weights go out now
                if (cin >= nc32) cin -= nc32;    // ... of the NEXT tile: the ring never drains
                issue_w(un % NWS, cin, tn / 3, tn % 3, c0 + ccn < total);
            } else if (t < 7) {
                const int cin = ct + cc + 1;     // next chunk: same tile, or chunk 0 of the next tile
                const bool nxt = cin == nc32;
                const bool live = c0 + cc + 1 < total;
#pragma unroll
                for (int q = 0; q < C::PPT; ++q)
                    if (t * C::PPT + q < C::NPX) {
                        const int i = t * C::PPT + q;
                        issue_piece(cc ^ 1, i, live ? (nxt ? pn[i] : pc[i]) : kOOB, nxt ? 0 : cin << 6);
                    }
            }
            {
                const int un = u + 1, tn = un % 9;
                read_frags(cur ^ 1, un % NWS, (un / 9) & 1, tn / 3, tn % 3);
            }
            mfma_half(cur, 1);
        }
        ct += 2;
        if (ct == nc32) {
            // ---- tile done: bias + activation, 16-byte NHWC pieces straight from the registers ----
            int q = pt_first + tcur;
            const int txi = q % tiles_x; q /= tiles_x;
            const int tyi = q % tiles_y, b = q / tiles_y, h0 = tyi * TH, w0 = txi * TW;
            act_dispatch(p.act, [&](auto act_c) {
            constexpr int ACT = decltype(act_c)::value;
            typedef __attribute__((ext_vector_type(4))) float float4v;
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int pix = (wm * TM + j) * 32 + l31;
                const int r = pix / TW, x = pix - r * TW;
                const int gy = h0 + r, gx = w0 + x;
                const bool okp = gy < p.H && gx < p.W;
                half_t* orow = outp + ((size_t)(b * p.H + gy) * p.W + gx) * p.ldout + p.cout_off + n0;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        unsigned w[2][2];
#pragma unroll
                        for (int gg = 0; gg < 2; ++gg) {
                            const int g = gp * 2 + gg;
                            const int nl = i * 32 + 8 * g + 4 * hi32;
                            float v[4];
                            const float4v bv = *(const float4v*)(lbias + nl);
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = act_t<ACT>(acc[i][j][g * 4 + e] + bv[e]);
                            typedef __attribute__((ext_vector_type(2))) _Float16 half2v;
                            half2v h0v = {(half_t)v[0], (half_t)v[1]}, h1v = {(half_t)v[2], (half_t)v[3]};
                            w[gg][0] = __builtin_bit_cast(unsigned, h0v);
                            w[gg][1] = __builtin_bit_cast(unsigned, h1v);
                        }
                        auto r0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                        auto r1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                        typedef __attribute__((ext_vector_type(4))) unsigned uint4v;
                        const uint4v pk = {r0[0], r1[0], r0[1], r1[1]};
                        const int nn = i * 32 + 8 * (gp * 2 + hi32);          // first of this lane's 8 channels
                        if (okp && n0 + nn < p.Cout) *(uint4v*)(orow + nn) = pk;
                    }
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
                }
            }
            });
            ct = 0;
            ++tcur;
            if (!wrole) {
#pragma unroll
                for (int i = 0; i < C::NPX; ++i) pc[i] = pn[i];
                patch_off(pt_first + tcur + 1, pn);
            }
        }
    }
#endif
}

template <int TW, int TH, int MT>
int launch_patch_mt(const Y7TConvArgs& a, hipStream_t s) {
    using C0 = PatchCfg<TW, TH, 64>;
    constexpr int lds = 3 * C0::W_BYTES + 2 * C0::PATCH_BYTES + 64 * 4;       // weight ring | patch A | patch B | biases
    static_assert(lds <= 81920, "two workgroups per CU");
    static Y7TOncePerDevice attr;      // (the attribute is per device: ADVICE r4)
    if (int e_ = y7t_once_per_device(attr, [&]() -> int {
        Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_patch_mt<TW, TH, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        return 0;
    })) return e_;
    const int ptiles = a.B * ((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW);
    hipLaunchKernelGGL((k_conv3x3_patch_mt<TW, TH, MT>), dim3(((ptiles + MT - 1) / MT) * (a.Cout_pad / 64)), dim3(256), lds, s, a);
    Y7T_LAUNCH_CHECK();
    y7t_note_kernel("patch_mt<%d,%d,%d>", TW, TH, MT);
    return 0;
}

template <int TW, int TH, int BN, int ABL = 0>
int launch_patch(const Y7TConvArgs& a, hipStream_t s) {
    using C = PatchCfg<TW, TH, BN>;
    static Y7TOncePerDevice attr;      // (the attribute is per device: ADVICE r4)
    if (int e_ = y7t_once_per_device(attr, [&]() -> int {
        Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_patch<TW, TH, BN, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
        return 0;
    })) return e_;
    const int ptiles = C::FLAT ? (a.B * (a.H + 2) * C::PW + 255) / 256 : a.B * ((a.H + TH - 1) / TH) * ((a.W + (C::FLAT ? 1 : TW) - 1) / (C::FLAT ? 1 : TW));
    const int tiles = ptiles * (a.Cout_pad / BN);
    hipLaunchKernelGGL((k_conv3x3_patch<TW, TH, BN, ABL>), dim3(tiles), dim3(256), C::LDS, s, a);
    Y7T_LAUNCH_CHECK();
    if (C::FLAT) y7t_note_kernel("patch_strip<%d,%d>%s", C::PW, BN, ABL == 2048 ? " burst-reads" : "");
    else y7t_note_kernel("patch<%d,%d,%d>%s", TW, TH, BN, ABL == 2048 ? " burst-reads" : "");
    return 0;
}

}   // namespace

// 1 if the layer was launched on the patch kernel, 0 if it is not eligible (caller falls back to k_conv_igemm), < 0 on error
int y7t_conv_patch_try(const Y7TConvArgs& a, hipStream_t s) {
    static int mode = -1;
    if (mode < 0) mode = y7t_exp_switch("Y7T_CONV_PATCH", 1);
    const bool eligible = a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Cin % 64 == 0 && !a.out_f32 && !(a.Cout & 7) && !(a.ldout & 7) &&
                          !(a.cout_off & 7) && a.Ho == a.H && a.Wo == a.W && a.in_bytes <= kOOB - (1u << 24) && a.w_bytes <= kOOB - (1u << 24);
    if (a.korder == 2 && (!eligible || !mode)) {   // panel-packed weights only make sense to this kernel
        y7t_set_error("conv: weights are in panel order (korder 2) but the layer cannot run on the patch kernel");
        return Y7T_E_ARG;
    }
    if (!mode || !eligible) return 0;
    // tile shape: the one that wastes fewer computed pixels; below 80 % useful pixels the linear-M kernel wins
    auto eff = [&](int tw, int th) {
        const int tx = (a.W + tw - 1) / tw, ty = (a.H + th - 1) / th;
        return (double)(a.W * a.H) / ((double)tx * tw * ty * th);
    };
    const double e16 = eff(16, 16), e32 = eff(32, 8);
    const bool use16 = e16 >= e32;
    const bool wide = a.Cout_pad % 128 == 0 && !a.panel64;
    const double eflat = (a.W == 40 || a.W == 20) ? (double)(a.W * a.H) / ((a.W + 2) * (a.H + 2)) : 0.0;   // strip tiling (instantiated for W = 20, 40)
    // too few workgroups for 256 CUs (batch-1 latency mode): the generic kernel with split-K fills the chip better
    if (!a.force_patch && a.korder != 2 && (long long)a.B * a.H * a.W * (a.Cout_pad / (wide ? 128 : 64)) < 256ll * 256) return 0;
    const int abl = Y7T_ABLATE ? a.ablate : 0;      // (ABL = 512, the step's DMAs behind its MFMAs, was measured in round 3: 15.98 vs 16.02 ms for the list -- not instantiated any more)
    if (eflat > (use16 ? e16 : e32) && eflat >= 0.8 && (!abl || abl == 2048)) {
        int rcf;
#if Y7T_ABLATE
        if (abl == 2048) {
            if (a.W == 40) rcf = wide ? launch_patch<0, 42, 128, 2048>(a, s) : launch_patch<0, 42, 64, 2048>(a, s);
            else rcf = wide ? launch_patch<0, 22, 128, 2048>(a, s) : launch_patch<0, 22, 64, 2048>(a, s);
        } else
#endif
        if (a.W == 40) rcf = wide ? launch_patch<0, 42, 128>(a, s) : launch_patch<0, 42, 64>(a, s);
        else rcf = wide ? launch_patch<0, 22, 128>(a, s) : launch_patch<0, 22, 64>(a, s);
        return rcf ? rcf : 1;
    }
    if (!a.force_patch && a.korder != 2 && (use16 ? e16 : e32) < 0.8) return 0;
    int rc;
#if Y7T_ABLATE      // liby7t_ablate.so only: the burst-read form and the timing ablations ("wrong results") of the 16 x 16 x 128 kernel
    if (abl == 2048) {      // A/B: the burst-read form (rounds 1-3a) of the default kernels, correct results
        if (use16) rc = wide ? launch_patch<16, 16, 128, 2048>(a, s) : launch_patch<16, 16, 64, 2048>(a, s);
        else rc = wide ? launch_patch<32, 8, 128, 2048>(a, s) : launch_patch<32, 8, 64, 2048>(a, s);
        return rc ? rc : 1;
    }
    if (abl && use16 && wide) {   // diagnostics: compile-time ablated instances of the 16x16x128 kernel in its burst-read form (scripts/patch_ablations.sh)
        switch (abl) {
        case 1: return launch_patch<16, 16, 128, 1>(a, s) ? -1 : 1;
        case 2: return launch_patch<16, 16, 128, 2>(a, s) ? -1 : 1;
        case 4: return launch_patch<16, 16, 128, 4>(a, s) ? -1 : 1;
        case 8: return launch_patch<16, 16, 128, 8>(a, s) ? -1 : 1;
        case 15: return launch_patch<16, 16, 128, 15>(a, s) ? -1 : 1;
        case 16: return launch_patch<16, 16, 128, 16>(a, s) ? -1 : 1;
        case 32: return launch_patch<16, 16, 128, 32>(a, s) ? -1 : 1;
        case 48: return launch_patch<16, 16, 128, 48>(a, s) ? -1 : 1;
        default: break;
        }
    }
#endif
    static int mt = -1;
    if (mt < 0) mt = y7t_exp_switch("Y7T_CONV_PATCH_MT", 1);
    const int ptiles = a.B * ((a.H + (use16 ? 15 : 7)) / (use16 ? 16 : 8)) * ((a.W + (use16 ? 15 : 31)) / (use16 ? 16 : 32));
    if (!wide && mt && !abl && (ptiles * (a.Cout_pad / 64) >= 4 * 2048 || a.force_patch)) {   // (force_patch: tests)   // 64-channel panels on big maps: multi-tile workgroups
        rc = use16 ? launch_patch_mt<16, 16, 4>(a, s) : launch_patch_mt<32, 8, 4>(a, s);
        return rc ? rc : 1;
    }
    if (use16) rc = wide ? launch_patch<16, 16, 128>(a, s) : launch_patch<16, 16, 64>(a, s);
    else rc = wide ? launch_patch<32, 8, 128>(a, s) : launch_patch<32, 8, 64>(a, s);
    return rc ? rc : 1;
}
