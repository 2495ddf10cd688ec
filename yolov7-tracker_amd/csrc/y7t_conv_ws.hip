// y7t_conv_ws.hip -- 3x3 / stride 1 / pad 1 Conv(+folded BN)+bias+activation for the 64 -> 64 layers, WEIGHTS STATIONARY IN REGISTERS.
//
// Same math as k_conv_igemm / k_conv3x3_patch (/root/reference/models/common.py:99-111 after utils/torch_utils.py:181-201).  These are the ELAN
// branch convolutions on the two largest maps (cfg/deploy/yolov7-w6.yaml:22-25 at 320x320, :108-111 at 160x160): 64 -> 64 channels, K = 576.  Their
// algorithmic intensity sits at the ridge (HBM time ~ MFMA time), and on the LDS-patch kernel they ran at 36 % of either: per K-step a workgroup
// streams a weight panel through the vector-memory path (36 of the 63 buffer->LDS pieces of a tile carry WEIGHTS that are the same for every tile),
// takes a workgroup barrier every 8 MFMAs per wave, and reads one LDS fragment per MFMA.
//
// Here the filter bank of the layer -- 64 x 576 fp16 = 72 KiB as MFMA A-fragments -- is loaded ONCE per workgroup into registers: each of the four waves
// keeps the 36 fragments of its 32 output channels (144 VGPRs per lane; gfx950 has 512 unified VGPR/AGPR per lane at one wave per SIMD), and a persistent
// workgroup per CU walks a contiguous range of 16 x 16-pixel tiles:
//   * vector-memory traffic per tile = the input patch only: 18 x 18 pixels x 64 channels, every pixel one full 128-byte line (50 pieces instead of
//     63 half-line pieces + 36 weight pieces);
//   * the patch sits in a THREE-buffer LDS ring (3 x 50 KiB): while tile T is multiplied, tile T+1 has landed or is landing and tile T+2 is being
//     requested -- up to 100 KiB in flight per CU, which is what keeping HBM busy at ~1 us of loaded latency takes;
//   * ONE workgroup barrier per tile (144 MFMAs per wave) instead of one per 8;
//   * one ds_read_b128 per MFMA (pixel fragments only), every address = lane base + immediate (144-byte pixel pitch: 9 sixteen-byte slots, 8 data +
//     1 pad; 36 x mod 64 is a bijection over the 16 lanes of a service group, the row pitch is a multiple of 256 B so the two image rows of an MFMA
//     tile interleave -- conflict-free);
//   * epilogue straight from registers (bias + activation + permlane32_swap -> 16-byte NHWC stores), of the PREVIOUS tile, interleaved with this tile's
//     MFMAs (two accumulator sets: with one wave per SIMD nothing else would cover it); maps of whole tiles only, so that every store is issued and
//     `s_waitcnt vmcnt` can count them next to the pieces.
// Weight layout (korder 5, detector/weights.py::pack_ws): fragment f = (tap * 4 + ks) * 2 + i is 1 KiB, lane l holds W[i*32 + l%32][tap][ks*16 + 8*(l/32) .. +7].
#include "y7t_common.h"
#include "y7t_conv_common.h"
#include <stdlib.h>

namespace {

constexpr unsigned kOOB = 0xFF000000u;

struct WsCfg {
    static constexpr int TW = 16, TH = 16;
    static constexpr int PIXB = 144;                                  // 64 channels x 2 B + 16 B pad
    static constexpr int RP = 2816;                                   // patch row pitch: 18 x 144 = 2592 rounded up to a multiple of 256 B
    static constexpr int PATCH_DMA = ((TH + 2) * RP + 1023) / 1024;   // 50 wave-wide 1 KiB pieces
    static constexpr int PATCH_BYTES = PATCH_DMA * 1024;
    static constexpr int NPW = (PATCH_DMA + 3) / 4;                   // pieces per wave per tile (13; a slot past the patch repeats its last KiB)
    static constexpr int NBUF = 3;
    static constexpr int BIAS_OFF = NBUF * PATCH_BYTES;
    static constexpr int LDS = BIAS_OFF + 64 * 4;
    static constexpr int NSUB = 36;                                   // k16 substeps per tile: 9 taps x 4
};

template <int ACT>      // the activation is a template parameter: the epilogue is instantiated eight times inside each of the two tile bodies
__global__ void __launch_bounds__(256, 1) k_conv3x3_c64_ws(const Y7TConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C = WsCfg;
    constexpr int TW = C::TW, TH = C::TH, PIXB = C::PIXB, RP = C::RP, NPW = C::NPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi32 = lane >> 5;

    // ---- this workgroup's tiles: a contiguous range (neighbouring tiles share halo lines: they are re-read from this CU's L1 / this XCD's L2) ----
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH, ptiles = p.B * tiles_y * tiles_x;
    int bid = blockIdx.x;
    if (p.xcd_swizzle) {      // workgroup b runs on XCD b % 8: give every XCD a contiguous range of workgroups (hence of tiles)
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int per = (ptiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int pt_first = bid * per;
    const int nt = (ptiles - pt_first) < per ? (ptiles - pt_first) : per;
    if (nt <= 0) return;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);

    // Source of this lane's 16 bytes of piece i of a tile = a per-lane CONSTANT (where its 16-byte slot sits inside the 18 x 18 patch, as a byte offset from
    // the patch's first pixel) + a wave-uniform tile offset: one v_add per piece for the tiles whose patch lies inside the image (81 % of a 320 x 320 map).
    // The kernel is bound by INSTRUCTION ISSUE (one wave per SIMD: ~7 issue slots per MFMA; a first version that decoded the tile and recomputed the slot
    // for every piece and every store ran 695 VALU + 369 SALU per 144 MFMAs -- 52 % of the wave's cycles issuing, the matrix pipe 37 % busy), so everything
    // that does not depend on the tile is computed once.  Slots that are never read (the pad chunk of a pixel, the tail of a patch row) fetch the
    // patch's first bytes.  Tiles on the image border take the per-lane bounds check (zero fill through an out-of-range offset).
    struct TileAt { int org, h0, w0; bool live, inner; };      // org: byte offset of the patch's first pixel (h0 - 1, w0 - 1) (interior tiles)
    auto tile_at = [&](int pt) -> TileAt {
        int q = pt;
        const int txi = q % tiles_x; q /= tiles_x;
        const int tyi = q % tiles_y, b = q / tiles_y;
        const int h0 = tyi * TH, w0 = txi * TW;
        return TileAt{((((b * p.H + h0 - 1) * p.W + w0 - 1) * p.ldin) + p.cin_off) * 2, h0, w0, pt < pt_first + nt,
                      tyi > 0 && tyi < tiles_y - 1 && txi > 0 && txi < tiles_x - 1};
    };
    int pc_r[NPW], pc_x[NPW];      // (border tiles only; the compiler keeps what it needs)
    unsigned pconst[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        int I = wave + 4 * i;
        if (I >= C::PATCH_DMA) I = C::PATCH_DMA - 1;
        const int byte = I * 1024 + lane * 16;
        const int r = byte / RP, rb = byte - r * RP;
        const int x = rb / PIXB, cs = (rb - x * PIXB) >> 4;
        const bool used = r < TH + 2 && x < TW + 2 && cs < 8;
        pconst[i] = used ? (unsigned)(((r * p.W + x) * p.ldin + cs * 8) * 2) : 0u;
        pc_r[i] = used ? r : -0x10000; pc_x[i] = x;
    }
    auto issue_piece = [&](int buf, const TileAt& ta, int i) __attribute__((always_inline)) {
        const int I = (wave + 4 * i < C::PATCH_DMA) ? wave + 4 * i : C::PATCH_DMA - 1;
        unsigned v;
        if (ta.inner && ta.live) v = pconst[i] + (unsigned)ta.org;                       // (wave-uniform branch)
        else {
            const int gy = ta.h0 + pc_r[i] - 1, gx = ta.w0 + pc_x[i] - 1;
            v = (ta.live && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) ? pconst[i] + (unsigned)ta.org : kOOB;
        }
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (LDS_AS void*)(smem + buf * C::PATCH_BYTES + I * 1024), 16, v, 0, 0, 0);
    };

    // ---- the filter bank.  Wave w owns output channels 32 (w & 1) .. +31 of the 128 pixels of tile half (w >> 1): 36 A-fragments per lane (144 VGPRs), straight
    // from memory (a fragment is 1 KiB of consecutive bytes), resident for the whole launch.  (A first version gave each wave all 64 channels of 64 pixels --
    // 288 VGPRs of weights, no room for a second accumulator set: its epilogue, ~3 000 cycles of bias + SiLU + pack per tile, ran with the matrix pipe idle,
    // 312 us per 320^2 layer against 358 us for the multi-tile patch kernel.  Here the previous tile's epilogue is interleaved with this tile's MFMAs.) ----
    const int chh = wave & 1, pxh = wave >> 1;
    half8 wreg[C::NSUB];
    {
        const half8* wp = (const half8*)p.w + chh * 64 + lane;
#pragma unroll
        for (int f = 0; f < C::NSUB; ++f) wreg[f] = wp[f * 128];
    }
    if (tid < 64) ((float*)(smem + C::BIAS_OFF))[tid] = p.bias[tid];
    const float* lbias = (const float*)(smem + C::BIAS_OFF) + chh * 32;

    {
        const TileAt t0 = tile_at(pt_first), t1 = tile_at(pt_first + 1);
#pragma unroll
        for (int i = 0; i < NPW; ++i) issue_piece(0, t0, i);
#pragma unroll
        for (int i = 0; i < NPW; ++i) issue_piece(1, t1, i);
    }

    // fragment base of this lane inside a patch buffer: four 32-pixel MFMA tiles of two image rows each, rows 8 pxh + 2j, + 1
    const int plane_off = (pxh * 8 + (l31 >> 4)) * RP + (l31 & 15) * PIXB + hi32 * 16;
    half_t* outp = (half_t*)p.out;
    typedef __attribute__((ext_vector_type(4))) unsigned uint4v;
    typedef __attribute__((ext_vector_type(4))) float float4v;
    typedef __attribute__((ext_vector_type(2))) _Float16 half2v;
    constexpr int NST = 8;                   // 16-byte stores per lane and tile (4 MFMA tiles x 2 group pairs): every one is issued, for every tile (the launcher
                                             // admits maps of whole 16 x 16 tiles only), so that `s_waitcnt vmcnt` can COUNT them next to the pieces

    // one eighth of a finished tile: group pair gp of MFMA tile j -> bias + activation -> 8 channels of one pixel -> one 16-byte NHWC store at
    // (wave-uniform tile / row-pair base) + (per-lane constant)
    const unsigned ovoff = (unsigned)(((((pxh * 8 + (l31 >> 4)) * p.W + (l31 & 15)) * p.ldout) + chh * 32 + 8 * hi32) * 2);
    auto out_base = [&](int pt) -> char* {
        int q = pt;
        const int txi = q % tiles_x; q /= tiles_x;
        const int tyi = q % tiles_y, b = q / tiles_y;
        return (char*)outp + ((size_t)((b * p.H + tyi * TH) * p.W + txi * TW) * p.ldout + p.cout_off) * 2;
    };
    const int jstep = 2 * p.W * p.ldout * 2;      // bytes between the row pairs of consecutive MFMA tiles
    auto store_group = [&](const floatx16 (&a)[4], char* obase, int j, int gp) __attribute__((always_inline)) {
        unsigned w[2][2];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const int g = gp * 2 + gg;
            const float4v bv = *(const float4v*)(lbias + 8 * g + 4 * hi32);
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = act_t<ACT>(a[j][g * 4 + e] + bv[e]);
            half2v h0v = {(half_t)v[0], (half_t)v[1]}, h1v = {(half_t)v[2], (half_t)v[3]};
            w[gg][0] = __builtin_bit_cast(unsigned, h0v);
            w[gg][1] = __builtin_bit_cast(unsigned, h1v);
        }
        auto r0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
        const uint4v v4 = {r0[0], r1[0], r0[1], r1[1]};
        *(uint4v*)(obase + (size_t)j * jstep + gp * 32 + ovoff) = v4;
    };

    // One tile: its 144 MFMAs into `cur`, the PREVIOUS tile's eight store groups (out of `prev`) spread between them, tile t+2's pieces spread between them.
    // vmcnt at the top: younger than this wave's pieces of tile t are exactly what tile t-1 issued: NPW pieces (tile t+1's) and, if tile t-1 had a
    // predecessor to store, NST stores.
    auto tile_body = [&](int t, int buf, floatx16 (&cur)[4], floatx16 (&prev)[4]) __attribute__((always_inline)) {
        if (t >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW + NST) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        __builtin_amdgcn_s_barrier();          // everybody's pieces of tile t are visible; nobody reads the buffer of tile t-1 any more: it takes tile t+2
        const int nbuf = (buf + 2 >= C::NBUF) ? buf + 2 - C::NBUF : buf + 2;
        const TileAt tn = tile_at(pt_first + t + 2);
        char* const ob = out_base(pt_first + t - 1);
        const char* pb = smem + buf * C::PATCH_BYTES + plane_off;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) cur[j][e] = 0.f;
        half8 xf[2][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) xf[0][j] = *(const half8*)(pb + j * 2 * RP);
#pragma clang loop unroll(full)
        for (int s = 0; s < C::NSUB; ++s) {
            const int cb = s & 1;
            if (s + 1 < C::NSUB) {           // fragments of the next substep: tap (kh, kw), 16-channel group ks
                const int sn = s + 1, tap = sn >> 2, ks = sn & 3, kh = tap / 3, kw = tap - kh * 3;
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[cb ^ 1][j] = *(const half8*)(pb + kh * RP + kw * PIXB + ks * 32 + j * 2 * RP);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s], xf[cb][j], cur[j], 0, 0, 0);      // (one wave per SIMD: no s_setprio)
            if (s % 3 == 0 && s / 3 < NPW) issue_piece(nbuf, tn, s / 3);      // tile t+2's pieces: one every third substep (12) + the last one
            if (s == C::NSUB - 2 && NPW > 12) issue_piece(nbuf, tn, 12);
            if (t > 0 && s % 4 == 1 && s / 4 < NST) store_group(prev, ob, (s / 4) >> 1, (s / 4) & 1);      // substeps 1, 5, ..., 29
        }
    };

    floatx16 accA[4], accB[4];
    int buf = 0;
    for (int t = 0; t < nt; t += 2) {
        tile_body(t, buf, accA, accB);
        buf = (buf + 1 == C::NBUF) ? 0 : buf + 1;
        if (t + 1 < nt) {
            tile_body(t + 1, buf, accB, accA);
            buf = (buf + 1 == C::NBUF) ? 0 : buf + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tail's zero-filling pieces have landed before this workgroup's LDS is handed on
    // the last tile's results (its stores overlap nothing)
    char* const obl = out_base(pt_first + nt - 1);
    if (nt & 1) {
#pragma unroll
        for (int g8 = 0; g8 < NST; ++g8) store_group(accA, obl, g8 >> 1, g8 & 1);
    } else {
#pragma unroll
        for (int g8 = 0; g8 < NST; ++g8) store_group(accB, obl, g8 >> 1, g8 & 1);
    }
#endif
}

}   // namespace

// korder 5 layers only (detector/graph.py::ws_eligible mirrors the conditions): 3x3 / 1 / 1, Cin == 64, Cout_pad == 64, fp16 output in 16-byte pieces
int y7t_conv_ws_launch(const Y7TConvArgs& a, hipStream_t s) {
    using C = WsCfg;
    const bool ok = a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Cin == 64 && a.Cout_pad == 64 && !a.out_f32 && !(a.Cout & 7) && !(a.ldout & 7) &&
                    !(a.cout_off & 7) && !(a.ldin & 7) && !(a.cin_off & 7) && a.Ho == a.H && a.Wo == a.W && a.in_bytes <= kOOB - (1u << 24) && a.Cout == 64 &&
                    a.H % C::TH == 0 && a.W % C::TW == 0;      // whole tiles only: the kernel counts its stores (s_waitcnt vmcnt), none may be predicated off
    if (!ok) {
        y7t_set_error("conv: weights are in register-fragment order (korder 5) but the layer is not a 3x3 / stride 1 / 64 -> 64 convolution on a map of whole 16 x 16 tiles with an aligned fp16 output");
        return Y7T_E_ARG;
    }
    static bool attr = false;
    if (!attr) {
        Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_c64_ws<Y7T_ACT_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
        Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_c64_ws<Y7T_ACT_SILU>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
        Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_c64_ws<Y7T_ACT_LEAKY>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
        attr = true;
    }
    static int ncu = -1;      // one persistent workgroup per compute unit (150 KiB of LDS each)
    if (ncu < 0) {
        const char* e = getenv("Y7T_CONV_WS_WGS");
        int dev = 0; hipDeviceProp_t prop;
        ncu = e ? atoi(e) : (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256);
        if (ncu <= 0) ncu = 256;
    }
    const int ptiles = a.B * ((a.H + C::TH - 1) / C::TH) * ((a.W + C::TW - 1) / C::TW);
    const int grid = ptiles < ncu ? ptiles : ncu;
    if (a.act == Y7T_ACT_SILU) hipLaunchKernelGGL(k_conv3x3_c64_ws<Y7T_ACT_SILU>, dim3(grid), dim3(256), C::LDS, s, a);
    else if (a.act == Y7T_ACT_LEAKY) hipLaunchKernelGGL(k_conv3x3_c64_ws<Y7T_ACT_LEAKY>, dim3(grid), dim3(256), C::LDS, s, a);
    else hipLaunchKernelGGL(k_conv3x3_c64_ws<Y7T_ACT_NONE>, dim3(grid), dim3(256), C::LDS, s, a);
    Y7T_LAUNCH_CHECK();
    y7t_note_kernel("ws64<16,16>");
    return 0;
}
