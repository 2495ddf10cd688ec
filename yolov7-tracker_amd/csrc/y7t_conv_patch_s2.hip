// y7t_conv_patch_s2.hip -- 3x3 / STRIDE 2 / pad 1 Conv(+folded BN)+bias+activation with an LDS-resident input patch whose columns are
// de-interleaved by parity.  EXPERIMENT, opt-in (Y7T_CONV_PATCH_S2=1 in the environment when the plan is lowered: detector/graph.py gives the
// eligible layers weights in this kernel's panel order, korder 4, and only korder 4 reaches this file).  The default launch list does not use it.
//
// Same math as k_conv_igemm (y7t_conv.hip; /root/reference/models/common.py:99-111 after utils/torch_utils.py:181-201); the layers it is for are the
// eight down-sampling convolutions of yolov7-w6 (/root/reference/cfg/deploy/yolov7-w6.yaml:18,27,36,45,54 and the head's :118,131,144), 3.2 ms of the
// 16.2 ms launch list on the generic kernel (profiles/r02_conv_per_layer_b32.txt), which fetches every (pixel, tap) row of the implicit im2col matrix
// separately: each input pixel-chunk crosses the vector-memory path 2.25 times, in 64-byte pieces of lines two pixels apart.
//
// Here a workgroup owns 8 x 16 output pixels x BN channels (BN = 128 or 256) and keeps, per 16-CHANNEL chunk of the input, the 17 x 33 pixel patch
// in LDS once (a 32-channel chunk would need 2 x 45 KB: stride 2 quadruples the input footprint per output pixel).  A patch row is stored as two planes,
//      [ E0 E1 ... E16 | O0 O1 ... O15 ]          E_i = patch column 2i, O_i = patch column 2i + 1, 48 bytes per pixel (32 data + 16 pad)
// which costs nothing on the way in (the buffer->LDS DMA writes lanes to consecutive LDS slots but takes a per-lane SOURCE offset) and makes every
// tap a unit-stride read: output column x needs E[x], O[x], E[x + 1] for kw = 0, 1, 2.  (A plain stride-2 ds_read_b128 is a 2-way bank conflict
// for every pixel pitch that keeps 16-byte alignment; 48-byte pixels at unit stride are conflict-free: 3 x mod 16 is a bijection.)  All fragment
// addresses are lane base + compile-time immediate, as in y7t_conv_patch.hip.
// K order: 16-channel chunk outermost, tap innermost; one K-step = one tap of one chunk = ONE v_mfma_f32_32x32x16_f16 per 32 x 32 tile.
// Weights: panel order (korder 4, detector/weights.py::panel_pack_s2): the BN x 16 panel of K-step (chunk, tap) is one contiguous BN * 32 byte
// block that already is the swizzled LDS image (16-byte half h of row r sits in slot h ^ ((r >> 3) & 1): two rows per 64 bytes, so the 16 lanes
// of a ds_read_b128 service group cover 16 distinct 16-byte bank groups).
// Pipeline: as y7t_conv_patch.hip -- waves 0/1 stream the weight panels through an NWS-slot ring, waves 2/3 the next chunk's patch in pieces
// during taps 0..6; one barrier per K-step, placed between the two halves of the step's MFMAs, the next step's fragments prefetched behind it.
// Not measured on a GPU yet (written at the end of round 2 without GPU minutes): tests/test_convsim.py runs this source on the host
// work-item by work-item against a plain convolution; DESIGN.md section 7 has the cost model it was designed to.
#include "y7t_common.h"
#include "y7t_conv_common.h"
#include <stdlib.h>

namespace {

constexpr unsigned kOOB = 0xFF000000u;   // voffset of a zero-filled lane: out of range with or without the (< 16 MiB) scalar offset

// NW = 4: 256 threads, 8 x 16 output pixels, two workgroups per CU.  NW = 8: 512 threads, 16 x 16 output pixels, ONE workgroup per CU (two waves per SIMD
// either way): twice the pixels per weight panel and a 33 x 33 patch (1.06 input pixels fetched per input pixel used instead of 1.10) -- the fewest
// buffer->LDS pieces per MFMA of the forms that fit in LDS.  Measured in round 3 (profiles/r03_conv_variants.txt): NW = 8 lost or tied on every layer, so only NW = 4 is
// instantiated; the parameter stays so that the configuration struct documents both.
template <int BN, int NW>
struct S2Cfg {
    static constexpr int TW = 16, TH = NW == 8 ? 16 : 8;                  // output tile: 128 / 256 pixels
    static constexpr int NT = 64 * NW, NLW = NW / 2;                      // threads; waves per DMA role (weights: waves 0 .. NLW-1, patch: the rest)
    static constexpr int PIXB = 48;                                       // bytes per patch pixel in LDS: 16 channels + 16 pad
    static constexpr int NE = TW + 1, NO = TW;                            // even / odd patch columns per row
    static constexpr int O_OFF = NE * PIXB;                               // the odd plane inside a row
    static constexpr int RP = (NE + NO) * PIXB;                           // patch row pitch (1584)
    static constexpr int ROWS = 2 * TH + 1;
    static constexpr int PATCH_DMA = (ROWS * RP + 1023) / 1024;           // wave-wide 1 KiB DMAs per patch (27 / 52)
    static constexpr int NPX = (PATCH_DMA + NLW - 1) / NLW;               // per PATCH wave
    static constexpr int PATCH_BYTES = PATCH_DMA * 1024;
    static constexpr int PPT = (NPX + 6) / 7;                             // pieces per patch wave per tap (taps 0..6)
    static constexpr int WROWB = 32;                                      // weight rows: 16 channels
    static constexpr int W_BYTES = BN * WROWB;                            // one K-step's panel: 4 / 8 KiB
    static constexpr int W_DMA = W_BYTES / 1024, NWX = W_DMA / NLW;       // DMAs per panel, per WEIGHT wave
    static constexpr int NWS = (BN == 128 || NW == 8) ? 6 : 3;            // ring slots (NW = 4: 24 KiB either way): K-step u+NWS goes out at step u
    static constexpr int W_OFF = 0, P_OFF = NWS * W_BYTES;                // LDS map: W ring | patch A | patch B
    static constexpr int LDS_LOOP = P_OFF + 2 * PATCH_BYTES;
    static constexpr int OROW = BN * 2 + 16;
    static constexpr int LDS_EPI = TW * TH * OROW;
    static constexpr int BIAS_OFF = LDS_LOOP > LDS_EPI ? LDS_LOOP : LDS_EPI;
    static constexpr int LDS = BIAS_OFF + BN * 4;
    static constexpr int WN = BN / 64, WM = NW / WN;                      // waves along channels / pixels
    static constexpr int TM = (TH / 2) / WM;                              // 32-pixel MFMA tiles (2 output rows x 16) per wave
    static constexpr int LDS_MAX = NW == 8 ? 163840 : 81920;              // one / two workgroups per CU
    static constexpr int JOFF = 4 * RP;                                   // LDS distance between consecutive MFMA tiles of a wave (2 output rows = 4 patch rows)
};

// ORD = 0: behind the barrier the DMAs go out first, then the next step's fragment reads, then the MFMAs (the order y7t_conv_patch.hip was tuned to).
// ORD = 1: fragment reads, the MFMAs, THEN the DMAs -- a buffer->LDS piece costs its wave 60-185 clocks of issue, most when the phase also carries ds_reads
// (MI355X_MICROARCH.md), and in this order the matrix pipe has the step's MFMAs queued while they go out.  Measured equal or slower (round 3): only ORD = 0 is instantiated.
template <int BN, int NW, int ORD>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 1 : 2) k_conv3x3s2_patch(const Y7TConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C = S2Cfg<BN, NW>;
    constexpr int PIXB = C::PIXB, RP = C::RP, TM = C::TM, TW = C::TW, TH = C::TH, NWS = C::NWS, NLW = C::NLW;
    static_assert(C::LDS <= C::LDS_MAX, "LDS budget of the chosen occupancy");
    static_assert(TM >= 1 && C::NWX >= 1 && C::WM * C::WN == NW, "wave layout");
    static_assert(C::PPT * 7 >= C::NPX, "patch pieces fit into taps 0..6");
    static_assert(18 % NWS == 0, "ring positions are compile-time in the 18-step unrolled loop");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / C::WM, wm = wave % C::WM;
    const int l31 = lane & 31, hi32 = lane >> 5;

    // ---- tile decode: channel tiles fastest, XCD-contiguous ranges (as y7t_conv_patch.hip) ----
    int bid = blockIdx.x;
    if (p.xcd_swizzle) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int n_tiles_n = p.Cout_pad / BN;
    const int tile_n = bid % n_tiles_n, n0 = tile_n * BN;
    int pt = bid / n_tiles_n;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int txi = pt % tiles_x; pt /= tiles_x;
    const int tyi = pt % tiles_y, b = pt / tiles_y, h0 = tyi * TH, w0 = txi * TW;      // first OUTPUT row / column of the tile

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);
    const int nc16 = p.Cin >> 4;

    // ---- per-lane DMA sources (computed once): waves 0/1 the weight panels, waves 2/3 the patch ----
    const bool wrole = wave < NLW;
    constexpr int NOFF = C::NPX > C::NWX ? C::NPX : C::NWX;
    unsigned off[NOFF];
#pragma unroll
    for (int i = 0; i < NOFF; ++i) {
        unsigned v = kOOB;
        if (wrole) {
            if (i < C::NWX) v = (unsigned)(tile_n * (nc16 * 9) * C::W_BYTES + (wave * C::NWX + i) * 1024 + lane * 16);   // memory image == LDS image
        } else if (i < C::NPX) {
            int I = (wave - NLW) + NLW * i;
            if (I >= C::PATCH_DMA) I = C::PATCH_DMA - 1;
            const int byte = I * 1024 + lane * 16;
            const int r = byte / RP, rb = byte - r * RP;
            const int px = rb / PIXB, cs = (rb - px * PIXB) >> 4;
            const int pc = px < C::NE ? 2 * px : 2 * (px - C::NE) + 1;         // patch column of this slot (even plane first)
            const int gy = 2 * h0 - 1 + r, gx = 2 * w0 - 1 + pc;
            if (r < C::ROWS && cs < 2 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                v = (unsigned)(((((b * p.H + gy) * p.W + gx) * p.ldin + p.cin_off) + cs * 8) * 2);
        }
        off[i] = v;
    }

    // ---- fragment read bases ----
    // weights: row = wn*64 + i*32 + l31; logical half hi32 sits in slot hi32 ^ ((row >> 3) & 1) = hi32 ^ ((l31 >> 3) & 1)
    const char* wlane = smem + C::W_OFF + (wn * 64 + l31) * C::WROWB + ((hi32 ^ ((l31 >> 3) & 1)) << 4);
    // patch: MFMA tile j of this wave = output rows 2*(wm*TM + j) + (l31 >> 4), column xl -> patch row 2*orow + kh, plane by kw.
    // Lanes 16-31 (the tile's second output row) take their columns ROTATED by 14: ds_read_b128 is served in the lane groups {0-3, 12-15, 20-27},
    // {4-11, 16-19, 28-31} (+32), i.e. half a group from each row; the rows are 2 * RP = 198 sixteen-byte units apart (6 mod 16), and with
    // 3 * 14 + 6 = 0 mod 16 the two halves of a group land on complementary bank groups (tests/test_convsim.py checks the address set).
    const int xl = (l31 & 16) ? (((l31 & 15) + 14) & 15) : (l31 & 15);
    const char* plane = smem + C::P_OFF + ((wm * TM * 2 + (l31 >> 4)) * 2) * RP + xl * PIXB + hi32 * 16;

    auto issue_w = [&](int slot, int so, bool real) {   // weight waves only
#pragma unroll
        for (int i = 0; i < C::NWX; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (LDS_AS void*)(smem + C::W_OFF + slot * C::W_BYTES + (wave * C::NWX + i) * 1024), 16,
                                                     real ? off[i] : kOOB, real ? so : 0, 0, 0);
    };
    auto issue_patch_piece = [&](int pb, int c, int i, bool real) {   // patch waves only
        const int I = ((wave - NLW) + NLW * i < C::PATCH_DMA) ? (wave - NLW) + NLW * i : C::PATCH_DMA - 1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (LDS_AS void*)(smem + C::P_OFF + pb * C::PATCH_BYTES + I * 1024), 16,
                                                 real ? off[i] : kOOB, c << 5, 0, 0);
    };

    floatx16 acc[2][TM];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    if (tid < BN) ((float*)(smem + C::BIAS_OFF))[tid] = p.bias[n0 + tid];
    // prologue: patch of chunk 0 and the weight panels of K-steps 0 .. NWS-1
    if (wrole) {
#pragma unroll
        for (int st = 0; st < NWS; ++st) issue_w(st, st * C::W_BYTES, st < nc16 * 9);
    } else {
#pragma unroll
        for (int i = 0; i < C::NPX; ++i) issue_patch_piece(0, 0, i, true);
    }

    half8 wf[2][2], xf[2][TM];   // [register buffer][tile]
    auto read_frags = [&](int buf, int slot, int pb, int kh, int kw) {
        const char* ws = wlane + slot * C::W_BYTES;
        const char* ps = plane + pb * C::PATCH_BYTES + kh * RP + (kw == 1 ? C::O_OFF : kw == 2 ? PIXB : 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[buf][i] = *(const half8*)(ws + i * 32 * C::WROWB);
#pragma unroll
        for (int j = 0; j < TM; ++j) xf[buf][j] = *(const half8*)(ps + j * C::JOFF);
    };
    auto mfma_half = [&](int buf, int i) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[buf][i], xf[buf][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    if (wrole) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NWS - 1) * C::NWX) : "memory");   // W(0) landed
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                     // patch(0) landed
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, 0, 0, 0);

    // K-step u = (chunk c, tap t), two chunks unrolled:
    //     MFMAs of the first 32 channels' tiles (fragments were read during step u-1)
    //     wait until W(u+1) landed (at a chunk's last tap: the next patch), barrier -- behind it everybody is done reading W(u) and step u's fragments
    //     issue W(u+NWS) into W(u)'s slot and one piece of patch(c+1); read ALL fragments of step u+1
    //     MFMAs of the second 32 channels' tiles
    int cbase = 0;                                      // byte offset of chunk c0's first panel in this tile's panel list
    for (int c0 = 0; c0 < nc16; c0 += 2) {
#pragma unroll
        for (int u = 0; u < 18; ++u) {
            const int cc = u / 9, t = u % 9;   // compile-time after unrolling
            const int c = c0 + cc, cur = u & 1;
            mfma_half(cur, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (wrole) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NWS - 2) * C::NWX) : "memory");
            else if (t == 8) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            auto issue_dmas = [&]() {
                if (wrole) {
                    const int un = u + NWS;                            // K-step whose weights go out now (ccn = 2: first chunk of the next pair)
                    const int ccn = un / 9;
                    issue_w(u % NWS, cbase + un * C::W_BYTES, c0 + ccn < nc16);
                } else if (t < 7) {
#pragma unroll
                    for (int q = 0; q < C::PPT; ++q)
                        if (t * C::PPT + q < C::NPX) issue_patch_piece(cc ^ 1, c + 1, t * C::PPT + q, c + 1 < nc16);
                }
            };
            if (ORD == 0) issue_dmas();
            {
                const int un = u + 1, tn = un % 9;
                read_frags(cur ^ 1, un % NWS, (un / 9) & 1, tn / 3, tn % 3);
            }
            mfma_half(cur, 1);
            if (ORD == 1) issue_dmas();
        }
        cbase += 18 * C::W_BYTES;
    }

    // ---- epilogue: bias + activation, transpose through LDS, full-line NHWC stores (as y7t_conv_patch.hip) ----
    constexpr int OROW = C::OROW;
    const float* lbias = (const float*)(smem + C::BIAS_OFF);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    typedef __attribute__((ext_vector_type(4))) float float4v;
    act_dispatch(p.act, [&](auto act_c) {
    constexpr int ACT = decltype(act_c)::value;
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int pix = (wm * TM + j) * 32 + (l31 & 16) + xl;   // tile-local pixel id of this lane's accumulator column: row = pix >> 4, x = pix & 15
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int nl = wn * 64 + i * 32;
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                unsigned w[2][2];
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int g = gp * 2 + gg;
                    float v[4];
                    const float4v bv = *(const float4v*)(lbias + nl + 8 * g + 4 * hi32);
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
                uint4v pk = {r0[0], r1[0], r0[1], r1[1]};
                *(uint4v*)(smem + pix * OROW + (nl + 8 * (gp * 2 + hi32)) * 2) = pk;
            }
        }
    }
    });
    __syncthreads();
    {
        typedef __attribute__((ext_vector_type(4))) unsigned uint4v;
        constexpr int CPP = BN / 8;
        half_t* outp = (half_t*)p.out;
        // the pieces of 16 bytes of a thread in groups: a group's LDS reads first (every address is inside the tile), then its stores (as y7t_conv_patch.hip, round 4)
        constexpr int PER = TW * TH * CPP / C::NT, GRP = PER % 8 == 0 ? 8 : PER % 4 == 0 ? 4 : 1;
        static_assert(TW * TH * CPP % C::NT == 0, "whole pieces per thread");
#pragma unroll 1
        for (int g0 = 0; g0 < PER; g0 += GRP) {
            uint4v v[GRP];
#pragma unroll
            for (int k = 0; k < GRP; ++k) {
                const int cidx = tid + (g0 + k) * C::NT, pix = cidx / CPP, ch = cidx - pix * CPP;
                v[k] = *(const uint4v*)(smem + pix * OROW + ch * 16);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < GRP; ++k) {
                const int cidx = tid + (g0 + k) * C::NT, pix = cidx / CPP, ch = cidx - pix * CPP;
                const int n = n0 + ch * 8;
                const int gy = h0 + (pix >> 4), gx = w0 + (pix & 15);
                if (gy < p.Ho && gx < p.Wo && n < p.Cout) *(uint4v*)(outp + ((size_t)(b * p.Ho + gy) * p.Wo + gx) * p.ldout + p.cout_off + n) = v[k];
            }
        }
    }
#endif
}

template <int BN, int NW, int ORD>
int launch_s2_ord(const Y7TConvArgs& a, hipStream_t s) {
    using C = S2Cfg<BN, NW>;
    static Y7TOncePerDevice attr;      // (the attribute is per device: ADVICE r4)
    if (int e_ = y7t_once_per_device(attr, [&]() -> int {
        Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3s2_patch<BN, NW, ORD>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
        return 0;
    })) return e_;
    const int ptiles = a.B * ((a.Ho + C::TH - 1) / C::TH) * ((a.Wo + C::TW - 1) / C::TW);
    hipLaunchKernelGGL((k_conv3x3s2_patch<BN, NW, ORD>), dim3(ptiles * (a.Cout_pad / BN)), dim3(C::NT), C::LDS, s, a);
    Y7T_LAUNCH_CHECK();
    if (NW == 4) y7t_note_kernel("patch_s2<%d>%s", BN, ORD ? " dma-late" : "");
    else y7t_note_kernel("patch_s2<%d,%d>%s", BN, NW, ORD ? " dma-late" : "");
    return 0;
}

template <int BN, int NW>
int launch_s2(const Y7TConvArgs& a, hipStream_t s) {
    // (ORD = 1 -- the step's DMAs behind its MFMAs -- and NW = 8 -- 512 threads, 16 x 16 output pixels, one workgroup per CU -- were measured in round 3:
    //  within +-1 % resp. 3-20 % slower than this form, profiles/r03_conv_variants.txt; they are no longer instantiated)
    return launch_s2_ord<BN, NW, 0>(a, s);
}

}   // namespace

// panel width the weights of a korder-4 layer are packed for (detector/weights.py::panel_pack_s2 uses the same rule)
static int s2_bn(int Cout_pad) {
    static int force128 = -1;
    if (force128 < 0) force128 = y7t_exp_switch("Y7T_CONV_PATCH_S2_BN", 0) == 128 ? 1 : 0;
    return (Cout_pad % 256 == 0 && !force128) ? 256 : 128;
}

// korder 4 layers only: 0 on success, < 0 on error (there is no fallback: the weights are in this kernel's panel order)
int y7t_conv_patch_s2_launch(const Y7TConvArgs& a, hipStream_t s) {
    const bool ok = a.KH == 3 && a.KW == 3 && a.stride == 2 && a.pad == 1 && a.Cin % 64 == 0 && a.Cout_pad % 128 == 0 && !a.out_f32 && !(a.Cout & 7) &&
                    !(a.ldout & 7) && !(a.cout_off & 7) && !a.epi && a.up_C == 0 && a.in_bytes <= kOOB - (1u << 24) && a.w_bytes <= kOOB - (1u << 24);
    if (!ok) {
        y7t_set_error("conv: weights are in the stride-2 patch kernel's panel order (korder 4) but the layer cannot run on it");
        return Y7T_E_ARG;
    }
    return s2_bn(a.Cout_pad) == 256 ? launch_s2<256, 4>(a, s) : launch_s2<128, 4>(a, s);
}
