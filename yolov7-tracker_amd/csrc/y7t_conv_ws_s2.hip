// y7t_conv_ws_s2.hip -- 3x3 / STRIDE 2 / pad 1 Conv(+folded BN)+bias+activation for the 64 -> 128 layer, WEIGHTS STATIONARY IN REGISTERS.
//
// Same math as k_conv_igemm (/root/reference/models/common.py:99-111 after utils/torch_utils.py:181-201).  One layer of yolov7-w6 has this shape -- the first
// down-sampling convolution, 640 x 640 x 64 -> 320 x 320 x 128 (cfg/deploy/yolov7-w6.yaml:18) -- and it is the most expensive launch of the benchmarked list:
// 835 us at 32 frames on k_conv_igemm<256,128,32,2> (578 TFLOP/s, 3.0 TB/s algorithmic; profiles/r03_conv_per_layer_b32.txt) against 460 us of HBM time
// (1.68 GB in, 0.84 GB out).  As an implicit GEMM it pulls every (pixel, tap) row AND a 147 KiB weight panel per 256-pixel tile through the buffer->LDS path:
// 7.5 GB per launch, 9 TB/s -- the rate that path saturates at (profiles/r04_p8_measurements.txt).  The stride-2 LDS-patch kernel fetches the input once but
// still streams the panel per tile and lost on this layer (922 us, profiles/r03_conv_variants.txt).
//
// Here the filter bank -- 128 x 576 fp16 = 144 KiB -- lives in REGISTERS for the whole launch, as in y7t_conv_ws.hip: each of the four waves keeps the 36 A-fragments
// of ITS 32 output channels (144 registers per lane; one wave per SIMD), all four waves work on the same pixels, and a persistent workgroup per CU walks a contiguous
// range of tiles.  The only vector-memory traffic of a tile is its input patch (fetched once: 1.27 input pixels per pixel used) and its output.
//   * tile = 2 output rows x 32 columns = two 32-pixel MFMA tiles (one per output row); patch = 5 input rows x 65 columns x 64 channels, columns DE-INTERLEAVED by
//     parity as in y7t_conv_patch_s2.hip ([E0 .. E32 | O0 .. O31] per row: output column x needs E[x], O[x], E[x + 1] for kw = 0, 1, 2 -- unit-stride reads; the DMA writes
//     lanes to consecutive LDS slots but takes a per-lane SOURCE offset, so the shuffle is free); 144-byte pixels (8 data slots + 1 pad): the 16 lanes of a ds_read_b128
//     service group are 16 pixels of ONE row, 9 x mod 16 is a bijection -> conflict-free without any row-pitch constraint;
//   * THREE patch buffers (3 x 46 KiB): tile t is multiplied while t + 1 has landed or is landing and t + 2 is being requested;
//   * ONE barrier per tile in the loop (72 MFMAs per wave) + one for the output transposition: the four waves' 32-channel pieces of a pixel are gathered in LDS
//     (64 pixels x 256 bytes) and leave as full 128-byte lines;
//   * the bias enters as the C operand of a tile's first MFMA.
// Compiler-scheduled (no micro-programmed epilogue: this layer is bound by HBM, not by issue).  Weight layout: korder 8 (detector/weights.py::pack_ws_s2): fragment
// f = (tap * 4 + ks) * 4 + q is 1 KiB, lane l holds W[q * 32 + l % 32][tap][ks * 16 + 8 * (l / 32) .. + 7].
#include "y7t_common.h"
#include "y7t_conv_common.h"
#include <stdlib.h>

namespace {

// device: the counter holds DMA pieces AND output stores, in issue order; host model (tests/_convsim): the DMA queue only
#if defined(Y7T_CONVSIM)
#define WS2_VMCNT(dev, dma) cs_vmcnt(dma)
#else
#define WS2_VMCNT(dev, dma) asm volatile("s_waitcnt vmcnt(" #dev ")" ::: "memory")
#endif

struct Ws2 {
    static constexpr int TW = 32, TH = 2;                             // output tile
    static constexpr int PIXB = 144;                                  // 64 channels x 2 B + 16 B pad
    static constexpr int NE = TW + 1, NO = TW;                        // even / odd patch columns per row
    static constexpr int O_OFF = NE * PIXB;
    static constexpr int RP = (NE + NO) * PIXB;                       // 9360
    static constexpr int ROWS = 2 * TH + 1;                           // 5
    static constexpr int PATCH_DMA = (ROWS * RP + 1023) / 1024;       // 46 wave-wide 1 KiB pieces
    static constexpr int PATCH_BYTES = PATCH_DMA * 1024;
    static constexpr int NPW = (PATCH_DMA + 3) / 4;                   // pieces per wave per tile (12; a slot past the patch repeats the last KiB)
    static constexpr int NBUF = 3;
    static constexpr int OUT_OFF = NBUF * PATCH_BYTES;                // 64 pixels x 256 B: the tile's output, gathered from the four waves
    static constexpr int LDS = OUT_OFF + TW * TH * 256;
    static constexpr int NSUB = 36;                                   // k16 substeps: 9 taps x 4
    static constexpr unsigned OOB = 0xFF000000u;
};

// FUSE: the layer's only consumers are the two 1x1 convolutions that open the next ELAN block (cfg/deploy/yolov7-w6.yaml:19-21: 128 -> 64 twice, one launch of 128 -> 128
// in this library) -- then the 320 x 320 x 128 tensor between them is never written: a tile's activated fp16 output sits in the LDS gather anyway (64 pixels x 128
// channels = exactly the 1x1 conv's B operand), the 1x1 filter bank is 8 more A-fragments per lane (32 registers), 16 more MFMAs per wave and tile, and the gather is
// reused for the result.  korder 11: the 1x1 bank follows the 3x3 bank in memory (fragment (36 + ks) * 4 + q: lane l holds W2[q * 32 + l % 32][ks * 16 + 8 * (l / 32) .. + 7]),
// its 128 biases follow the 3x3 layer's.  Saves the write and the read of 0.84 GB per 32 frames and one launch.
template <int ACT, bool FUSE>
__global__ void __launch_bounds__(256, 1) k_conv3x3s2_c64_ws(const Y7TConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C = Ws2;
    constexpr int TW = C::TW, TH = C::TH, PIXB = C::PIXB, RP = C::RP, NPW = C::NPW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi32 = lane >> 5;

    // ---- this workgroup's tiles: a contiguous range, x fastest (as y7t_conv_ws.hip) ----
    const int tiles_x = p.Wo / TW, tiles_y = p.Ho / TH, ptiles = p.B * tiles_y * tiles_x;
    int bid = blockIdx.x;
    if (p.xcd_swizzle) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int per = (ptiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int pt_first = bid * per;
    const int nt = (ptiles - pt_first) < per ? (ptiles - pt_first) : per;
    if (nt <= 0) return;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);

    // tile iterator: (b, ty, tx) stepped, not decoded
    struct TileIt { int b, ty, tx, n; };
    auto tile_it = [&](int pt) -> TileIt {
        int q = pt;
        const int txi = q % tiles_x; q /= tiles_x;
        return TileIt{q / tiles_y, q % tiles_y, txi, pt_first + nt - pt};
    };
    auto tile_next = [&](TileIt& it) __attribute__((always_inline)) {
        it.n -= 1;
        if (++it.tx == tiles_x) { it.tx = 0; if (++it.ty == tiles_y) { it.ty = 0; ++it.b; } }
    };
    // Source of this lane's 16 bytes of piece i = per-lane constant (its slot's place in the 5 x 65 patch as a byte offset from input pixel (2 h0 - 1, 2 w0 - 1))
    // + tile origin; pedge: the slot lies in the patch's top row (bit 0) / first column (bit 1) -- the only sides that can fall outside an even-sized image.
    unsigned pconst[NPW], pedge = 0;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        int I = wave + 4 * i;
        if (I >= C::PATCH_DMA) I = C::PATCH_DMA - 1;
        const int byte = I * 1024 + lane * 16;
        const int r = byte / RP, rb = byte - r * RP;
        const int px = rb / PIXB, cs = (rb - px * PIXB) >> 4;
        const int pc = px < C::NE ? 2 * px : 2 * (px - C::NE) + 1;         // patch column of this slot (even plane first)
        const bool used = r < C::ROWS && cs < 8;
        pconst[i] = used ? (unsigned)(((r * p.W + pc) * p.ldin + cs * 8) * 2) : 0u;
        const unsigned e = used ? (unsigned)((r == 0) | ((pc == 0) << 1)) : 0u;
        pedge |= e << (2 * i);
    }
    auto piece_offsets = [&](const TileIt& it, unsigned (&pv)[NPW]) __attribute__((always_inline)) {
        // (unsigned arithmetic: the origin of a tile in the top row / first column lies one row / pixel BEFORE the image; the slots that would read there are masked)
        const unsigned org = ((((unsigned)it.b * (unsigned)p.H + (unsigned)(2 * it.ty * TH) - 1u) * (unsigned)p.W + (unsigned)(2 * it.tx * TW) - 1u) * (unsigned)p.ldin +
                              (unsigned)p.cin_off) * 2u;
        const unsigned tmask = (unsigned)((it.ty == 0) | ((it.tx == 0) << 1));
#pragma unroll
        for (int i = 0; i < NPW; ++i) pv[i] = (it.n <= 0 || ((pedge >> (2 * i)) & tmask)) ? C::OOB : pconst[i] + org;
    };
    auto issue_piece = [&](int buf, unsigned v, int i) __attribute__((always_inline)) {
        const int I = (wave + 4 * i < C::PATCH_DMA) ? wave + 4 * i : C::PATCH_DMA - 1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (LDS_AS void*)(smem + buf * C::PATCH_BYTES + I * 1024), 16, v, 0, 0, 0);
    };

    // ---- the filter bank: wave q owns output channels 32 q .. + 31 -- 36 A-fragments per lane, straight from memory (a fragment is 1 KiB of consecutive bytes) ----
    half8 wreg[C::NSUB];
    {
        const half8* wp = (const half8*)p.w + wave * 64 + lane;
#pragma unroll
        for (int f = 0; f < C::NSUB; ++f) wreg[f] = wp[f * 256];
    }
    floatx16 biasv;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) biasv[g * 4 + e] = p.bias[wave * 32 + 8 * g + 4 * hi32 + e];

    half8 w2reg[FUSE ? 8 : 1];
    floatx16 bias2v;
    if (FUSE) {
        const half8* wp2 = (const half8*)p.w + C::NSUB * 256 + wave * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) w2reg[ks] = wp2[ks * 256];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) bias2v[g * 4 + e] = p.bias[128 + wave * 32 + 8 * g + 4 * hi32 + e];
    }

    unsigned pv[NPW];
    TileIt itn = tile_it(pt_first), ito = itn;      // itn: the tile whose pieces are issued next; ito: the tile being computed
    piece_offsets(itn, pv);
#pragma unroll
    for (int i = 0; i < NPW; ++i) issue_piece(0, pv[i], i);
    tile_next(itn);
    piece_offsets(itn, pv);
#pragma unroll
    for (int i = 0; i < NPW; ++i) issue_piece(1, pv[i], i);
    tile_next(itn);

    // fragment base of this lane inside a patch buffer: MFMA tile j = output row j of the tile, lane = output column l31; tap (kh, kw) -> patch row 2 j + kh, plane by kw
    const int plane_off = l31 * PIXB + hi32 * 16;
    half_t* outp = (half_t*)p.out;
    typedef __attribute__((ext_vector_type(4))) unsigned uint4v;

    // ---- epilogue pieces: activation, fp16, v_permlane32_swap -> 16-byte pieces into the tile's output gather (row = pixel, 256 B, slot = chunk ^ (pixel & 15)) ----
    char* og = smem + C::OUT_OFF;
    // chunk c of 8 ACTIVATED values y[0..7] of a lane (MFMA tile j = c / 2, group pair gp = c % 2) -> fp16, lanes paired, one 16-byte piece into the gather
    auto put_chunk = [&](const float* y, int c) __attribute__((always_inline)) {
        const int j = c >> 1, gp = c & 1, pix = j * 32 + l31;
        typedef __attribute__((ext_vector_type(2))) _Float16 half2v;
        unsigned w[2][2];
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            half2v h0 = {(half_t)y[gg * 4 + 0], (half_t)y[gg * 4 + 1]}, h1 = {(half_t)y[gg * 4 + 2], (half_t)y[gg * 4 + 3]};
            w[gg][0] = __builtin_bit_cast(unsigned, h0);
            w[gg][1] = __builtin_bit_cast(unsigned, h1);
        }
        auto r0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
        auto r1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
        const uint4v pk = {r0[0], r1[0], r0[1], r1[1]};
        const int ch = wave * 4 + gp * 2 + hi32;      // 16-byte chunk (8 channels) of the pixel's 128 channels
        *(uint4v*)(og + pix * 256 + ((ch ^ (pix & 15)) << 4)) = pk;
    };
    auto gather = [&](const floatx16 (&a)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float y[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = act_t<ACT>(a[c >> 1][(c & 1) * 8 + e]);
            put_chunk(y, c);
        }
    };
    auto store_tile = [&](const TileIt& it) __attribute__((always_inline)) {      // the gathered tile out as full lines: 64 pixels x 16 chunks = 1024 pieces of 16 bytes, 4 per thread
        const int oy = it.ty * TH, ox = it.tx * TW;
        uint4v v[4];      // the four reads first, then the four stores (left alone the compiler waits for each read in front of its store)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = k * 256 + tid, pix = c >> 4, ch = c & 15;
            v[k] = *(const uint4v*)(og + pix * 256 + ((ch ^ (pix & 15)) << 4));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = k * 256 + tid, pix = c >> 4, ch = c & 15;
            const int gy = oy + (pix >> 5), gx = ox + (pix & 31);
            *(uint4v*)(outp + ((size_t)(it.b * p.Ho + gy) * p.Wo + gx) * p.ldout + p.cout_off + ch * 8) = v[k];
        }
    };
    floatx16 acc2[2];

    int buf = 0;
    for (int t = 0; t < nt; ++t) {
        const int nbuf = (buf + 2 >= C::NBUF) ? buf + 2 - C::NBUF : buf + 2;
        // this wave's pieces of tile t have landed.  Younger than them, in issue order: the 4 output stores of tile t - 2, the NPW pieces of tile t + 1, the 4 stores
        // of tile t - 1 (every thread issues exactly 4 per tile: maps of whole tiles only)
        static_assert(NPW == 12, "the waits below count the pieces and stores of a tile");
        if (t == 0) WS2_VMCNT(12, 12);
        else if (t == 1) WS2_VMCNT(16, 12);
        else WS2_VMCNT(20, 12);
        __builtin_amdgcn_s_barrier();      // everybody's pieces of tile t are visible; nobody reads the buffer of tile t - 1 or the output gather (as the 1x1 layer's operand) any more
        piece_offsets(itn, pv);            // tile t + 2 goes into the buffer tile t - 1 used
#pragma unroll
        for (int i = 0; i < NPW; ++i) issue_piece(nbuf, pv[i], i);
        tile_next(itn);

        const char* pb = smem + buf * C::PATCH_BYTES + plane_off;
        floatx16 acc[2];
        auto frag = [&](int sn, int j) __attribute__((always_inline)) -> half8 {      // pixel fragment of substep sn = tap * 4 + ks for MFMA tile j
            const int tap = sn >> 2, ks = sn & 3, kh = tap / 3, kw = tap - kh * 3;
            return *(const half8*)(pb + (2 * j + kh) * RP + (kw == 1 ? C::O_OFF : kw == 2 ? PIXB : 0) + ks * 32);
        };
        // The fragment reads run three substeps ahead BY HAND and the order of a slot (MFMA, the read for three substeps on, ...) is pinned with sched_barrier: left
        // to itself the compiler put most reads right in front of their MFMA behind an s_waitcnt lgkmcnt(0) -- one LDS round trip (~100 clocks) per 32-clock MFMA
        // (round 4: the kernel ran at 40 % of its MFMA time for that reason, not for its DMAs)
        half8 xq[3][2];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int j = 0; j < 2; ++j) xq[q][j] = frag(q, j);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < C::NSUB; ++s) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s], xq[s % 3][j], s == 0 ? biasv : acc[j], 0, 0, 0);
                if (s + 3 < C::NSUB) xq[s % 3][j] = frag(s + 3, j);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        gather(acc);
        if (FUSE) {
            __builtin_amdgcn_s_barrier();      // the 64 pixels x 128 channels of the 3x3 layer's output are in the gather: the 1x1 layer's B operand
            auto frag2 = [&](int ks, int j) __attribute__((always_inline)) -> half8 {      // channels 16 ks + 8 (lane / 32) .. + 7 of pixel j * 32 + lane % 32
                const int pix = j * 32 + l31, ch = ks * 2 + hi32;
                return *(const half8*)(og + pix * 256 + ((ch ^ (pix & 15)) << 4));
            };
            half8 x2[3][2];
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int j = 0; j < 2; ++j) x2[q][j] = frag2(q, j);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2reg[ks], x2[ks % 3][j], ks == 0 ? bias2v : acc2[j], 0, 0, 0);
                    if (ks + 3 < 8) x2[ks % 3][j] = frag2(ks + 3, j);
                    __builtin_amdgcn_sched_barrier(0);
                }
            __builtin_amdgcn_s_barrier();      // everybody has read the gather: it takes the 1x1 layer's output now
            gather(acc2);
        }
        __builtin_amdgcn_s_barrier();      // the four waves' pieces of every pixel are in place
        store_tile(ito);
        tile_next(ito);
        buf = (buf + 1 == C::NBUF) ? 0 : buf + 1;
    }
    WS2_VMCNT(0, 0);      // the tail's zero-filling pieces have landed before this workgroup's LDS is handed on
#endif
}

}   // namespace

// korder 8 layers only (detector/graph.py::ws_s2_eligible mirrors the conditions): 3x3 / 2 / 1, Cin == 64, Cout == 128, even input, output map of whole 2 x 32 tiles
int y7t_conv_ws_s2_launch(const Y7TConvArgs& a, hipStream_t s) {
    using C = Ws2;
    const bool ok = a.KH == 3 && a.KW == 3 && a.stride == 2 && a.pad == 1 && a.Cin == 64 && a.Cout == 128 && a.Cout_pad == 128 && !a.out_f32 && !(a.ldout & 7) &&
                    !(a.cout_off & 7) && !(a.ldin & 7) && !(a.cin_off & 7) && !(a.H & 1) && !(a.W & 1) && a.Ho * 2 == a.H && a.Wo * 2 == a.W && a.in_bytes < 0x80000000u &&
                    a.Ho % C::TH == 0 && a.Wo % C::TW == 0 && !a.epi && a.up_C == 0;
    if (!ok) {
        y7t_set_error("conv: weights are in the stride-2 register-fragment order (korder 8) but the layer is not a 3x3 / stride 2 / 64 -> 128 convolution of an even map "
                      "whose output is whole 2 x 32 tiles with an aligned fp16 output");
        return Y7T_E_ARG;
    }
    static Y7TOncePerDevice attr;      // (the attribute is per device: ADVICE r4)
    if (int e_ = y7t_once_per_device(attr, [&]() -> int {
#define WS2_ATTR(ACT) Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3s2_c64_ws<ACT, false>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS)); \
                      Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3s2_c64_ws<ACT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
        WS2_ATTR(Y7T_ACT_NONE) WS2_ATTR(Y7T_ACT_SILU) WS2_ATTR(Y7T_ACT_LEAKY)
#undef WS2_ATTR
        return 0;
    })) return e_;
    static int wgs_env = -2;      // Y7T_CONV_WS_WGS: workgroups of a launch (default: one persistent workgroup per compute unit -- 154 KiB of LDS, 144 + registers of weights per lane)
    if (wgs_env == -2) wgs_env = y7t_exp_switch("Y7T_CONV_WS_WGS", -1);
    const int ncu = wgs_env > 0 ? wgs_env : y7t_num_cus();
    const int ptiles = a.B * (a.Ho / C::TH) * (a.Wo / C::TW);
    const int grid = ptiles < ncu ? ptiles : ncu;
    const bool fuse = a.korder == 11;      // + the twin 1x1 convolution behind it (its bank and biases follow this layer's)
#define WS2_GO(ACT) do { if (fuse) hipLaunchKernelGGL((k_conv3x3s2_c64_ws<ACT, true>), dim3(grid), dim3(256), C::LDS, s, a); \
                         else hipLaunchKernelGGL((k_conv3x3s2_c64_ws<ACT, false>), dim3(grid), dim3(256), C::LDS, s, a); } while (0)
    if (a.act == Y7T_ACT_SILU) WS2_GO(Y7T_ACT_SILU);
    else if (a.act == Y7T_ACT_LEAKY) WS2_GO(Y7T_ACT_LEAKY);
    else WS2_GO(Y7T_ACT_NONE);
#undef WS2_GO
    Y7T_LAUNCH_CHECK();
    y7t_note_kernel(fuse ? "ws_s2<2,32> + 1x1" : "ws_s2<2,32>");
    return 0;
}
