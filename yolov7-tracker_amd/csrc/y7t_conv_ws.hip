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
    static constexpr int RING_OFF = BIAS_OFF + 64 * 4;                // DYN: the chunk ids handed to this workgroup, a ring of four
    static constexpr int LDS = RING_OFF + 64;
    static constexpr int CH = 2;                                      // DYN: tiles per chunk of the tile counter
    static constexpr int NSUB = 36;                                   // k16 substeps per tile: 9 taps x 4
};

// ACT: the activation, a template parameter (the epilogue is instantiated inside each tile body).  ABL: timing ablations with WRONG results (Y7T_WS_ABLATE, scripts/ws_probe.py):
// 1 no interleaved epilogue (no stores either), 2 no pieces in the loop (the ring keeps the prologue's tiles), 4 no fragment reads, 8 no vmcnt wait / barrier per tile,
// 16 the epilogue's arithmetic without its stores.
//
// DYN (round 5): DYNAMIC TILE SCHEDULING.  A statically partitioned persistent workgroup loses in the pipeline whatever it gained alone: the CU that also
// hosts the tracker's workgroup or NMS workgroups finishes its range late and the launch ends with it (profiles/r04_ws128_measurement.txt: the list pays 0.43 ms for
// co-running work with 4-per-CU kernels, 0.81 ms with a persistent one).  Here a workgroup takes CHUNKS of CH = 2 x-adjacent tiles from a counter in memory
// (Y7TConvArgs::tile_ctr): chunks 0 and 1 of a workgroup are static (blockIdx.x, gridDim.x + blockIdx.x: the pipeline is three tiles deep and must be primed without a
// round trip), every later one is 2 gridDim.x + atomicAdd(ctr, 1).  The fetch is software-pipelined like everything else in this kernel: lane 0 of wave 0 issues the atomic
// for chunk k + 2 behind the barrier of chunk k's first tile -- OLDER than that tile body's pieces, so the counted `s_waitcnt vmcnt` at the top of the next body already
// covers it --, writes the id into a four-entry LDS ring in front of that body's barrier, and every wave reads it behind the barrier when its iterators hop (the issue
// iterator three tiles ahead, the store iterator one tile behind).  Chunk ids are decoded with multiply-high by precomputed reciprocals (ids < 2^16).  A chunk id past the
// end makes the tile dead (zero-filling pieces, as the tail of the static form); the loop ends at the first dead tile.  The last workgroup to leave resets the counter
// (ctr[Y7T_TILE_CTR_DONE] counts leavers), so a launch list -- or a captured hipGraph -- needs no memset.
template <int ACT, int ABL = 0, bool DYN = false>
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
    const int pt_first = DYN ? 0 : bid * per;
    int nt = DYN ? 0 : ((ptiles - pt_first) < per ? (ptiles - pt_first) : per);
    if (!DYN && nt <= 0) return;
    constexpr int CH = C::CH;
    const unsigned magic_x = 0xFFFFFFFFu / (unsigned)tiles_x + 1u, magic_y = 0xFFFFFFFFu / (unsigned)tiles_y + 1u;      // exact quotients for numerator x divisor < 2^32 (the launcher checks)
    volatile LDS_AS int* const ring = (volatile LDS_AS int*)(smem + C::RING_OFF);

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);

    // Source of this lane's 16 bytes of piece i of a tile = a per-lane CONSTANT (where its 16-byte slot sits inside the 18 x 18 patch, as a byte offset from
    // the patch's first pixel) + a wave-uniform tile offset: one v_add per piece for the tiles whose patch lies inside the image (81 % of a 320 x 320 map).
    // The kernel is bound by INSTRUCTION ISSUE (one wave per SIMD: ~7 issue slots per MFMA; a first version that decoded the tile and recomputed the slot
    // for every piece and every store ran 695 VALU + 369 SALU per 144 MFMAs -- 52 % of the wave's cycles issuing, the matrix pipe 37 % busy), so everything
    // that does not depend on the tile is computed once.  Slots that are never read (the pad chunk of a pixel, the tail of a patch row) fetch the
    // patch's first bytes.  Tiles on the image border take the per-lane bounds check (zero fill through an out-of-range offset).
    // Tiles are walked in order, so their coordinates are stepped, not decoded (two integer divisions per tile were ~50 scalar instructions with the matrix pipe
    // idle): P = index of the tile's first pixel in the NHWC map, (ty, tx) = its place in the tile grid, n = tiles of this workgroup's range still ahead of it.
    // DYN: n = tiles of the current CHUNK still ahead (this one included).  Every chunk is walked as CH tiles -- the last one may hold fewer: its missing tile, like every
    // tile of a chunk past the end, decodes / steps to a pixel index beyond the batch and is DEAD (zero-filling pieces; the loop ends at the first dead tile) -- so the
    // tile parity the ring's timing rests on (even local tile = first tile of a chunk) always holds and a tile's liveness needs no state of its own.
    struct TileIt { int P, ty, tx, n; };
    auto tile_it = [&](int pt) -> TileIt {
        int q = pt;
        const int txi = q % tiles_x; q /= tiles_x;
        const int tyi = q % tiles_y, b = q / tiles_y;
        return TileIt{(b * p.H + tyi * TH) * p.W + txi * TW, tyi, txi, pt_first + nt - pt};
    };
    auto chunk_it = [&](int id) __attribute__((always_inline)) -> TileIt {      // (id wave-uniform; ids stay below 2^16 + a few hundred: the quotients are exact)
        const unsigned pt = (unsigned)id * CH;
        const unsigned q = tiles_x == 1 ? pt : __umulhi(pt, magic_x), txi = pt - q * (unsigned)tiles_x;      // (the reciprocal of 1 does not fit 32 bits)
        const unsigned b = tiles_y == 1 ? q : __umulhi(q, magic_y), tyi = q - b * (unsigned)tiles_y;
        return TileIt{(int)((b * (unsigned)p.H + tyi * TH) * (unsigned)p.W + txi * TW), (int)tyi, (int)txi, CH};
    };
    // DYN: `hop` = the place in this workgroup's chunk sequence of the chunk that follows when the current one is used up (its id is in the ring: written >= one barrier
    // ago, see tile_body)
    auto tile_next = [&](TileIt& it, int hop) __attribute__((always_inline)) {
        if (DYN && it.n == 1) {
            it = chunk_it(__builtin_amdgcn_readfirstlane(ring[hop & 3]));
            return;
        }
        it.P += TW; it.n -= 1;
        if (++it.tx == tiles_x) { it.tx = 0; it.P += (TH - 1) * p.W; if (++it.ty == tiles_y) it.ty = 0; }      // (maps are whole tiles: the next image follows the last row)
    };
    struct TileAt { int org, ty, tx; bool live; };      // org: byte offset of the patch's first pixel (h0 - 1, w0 - 1)
    const unsigned npix = (unsigned)(p.B * p.H * p.W);
    auto tile_at = [&](const TileIt& it) -> TileAt {      // (unsigned arithmetic: P keeps stepping past the last live tile, and wrap-around must be defined behaviour)
        return TileAt{(int)((((unsigned)it.P - (unsigned)p.W - 1u) * (unsigned)p.ldin + (unsigned)p.cin_off) * 2u), it.ty, it.tx, DYN ? (unsigned)it.P < npix : it.n > 0};
    };
    // per-lane constants of piece i: pconst = where its 16-byte slot sits inside the 18 x 18 patch (byte offset from the patch's first pixel); pedge = which
    // halo sides the slot lies on (bit 0 top row, 1 bottom row, 2 left column, 3 right column), four bits per piece -- maps are whole tiles, so a slot can be
    // outside the image only through the halo of a tile that touches that border.  Slots that are never read (pad chunk, row tail) fetch the patch's first bytes.
    unsigned pconst[NPW], pedge_lo = 0, pedge_hi = 0;
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        int I = wave + 4 * i;
        if (I >= C::PATCH_DMA) I = C::PATCH_DMA - 1;
        const int byte = I * 1024 + lane * 16;
        const int r = byte / RP, rb = byte - r * RP;
        const int x = rb / PIXB, cs = (rb - x * PIXB) >> 4;
        const bool used = r < TH + 2 && x < TW + 2 && cs < 8;
        pconst[i] = used ? (unsigned)(((r * p.W + x) * p.ldin + cs * 8) * 2) : 0u;
        const unsigned e = used ? (unsigned)((r == 0) | ((r == TH + 1) << 1) | ((x == 0) << 2) | ((x == TW + 1) << 3)) : 0u;
        if (i < 8) pedge_lo |= e << (4 * i); else pedge_hi |= e << (4 * (i - 8));
    }
    // the 13 source offsets of a tile's pieces, computed once per tile under ONE wave-uniform branch (the piece issue itself is then M0 + one instruction)
    auto piece_offsets = [&](const TileAt& ta, unsigned (&pv)[NPW]) __attribute__((always_inline)) {
        const unsigned tmask = (unsigned)((ta.ty == 0) | ((ta.ty == tiles_y - 1) << 1) | ((ta.tx == 0) << 2) | ((ta.tx == tiles_x - 1) << 3));
        if (!ta.live) {
#pragma unroll
            for (int i = 0; i < NPW; ++i) pv[i] = kOOB;
        } else if (tmask == 0) {
#pragma unroll
            for (int i = 0; i < NPW; ++i) pv[i] = pconst[i] + (unsigned)ta.org;
        } else {
#pragma unroll
            for (int i = 0; i < NPW; ++i) {
                const unsigned e = ((i < 8 ? pedge_lo >> (4 * i) : pedge_hi >> (4 * (i - 8))) & tmask);
                pv[i] = e ? kOOB : pconst[i] + (unsigned)ta.org;
            }
        }
    };
    auto issue_piece = [&](int buf, unsigned v, int i) __attribute__((always_inline)) {
        const int I = (wave + 4 * i < C::PATCH_DMA) ? wave + 4 * i : C::PATCH_DMA - 1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (LDS_AS void*)(smem + buf * C::PATCH_BYTES + I * 1024), 16, v, 0, 0, 0);
    };

    // ---- the filter bank.  Wave w owns output channels 32 (w & 1) .. +31 of the 128 pixels of tile half (w >> 1): 36 A-fragments per lane (144 registers), straight
    // from memory (a fragment is 1 KiB of consecutive bytes), resident for the whole launch.  (A first version gave each wave all 64 channels of 64 pixels --
    // 288 registers of weights, no room for a second accumulator set: its epilogue, ~3 000 cycles of bias + SiLU + pack per tile, ran with the matrix pipe idle.) ----
    const int chh = wave & 1, pxh = wave >> 1;
    half8 wreg[C::NSUB];
    {
        const half8* wp = (const half8*)p.w + chh * 64 + lane;
#pragma unroll
        for (int f = 0; f < C::NSUB; ++f) wreg[f] = wp[f * 128];
    }
    // the bias enters as the C operand of a tile's first MFMA: row 8 g + 4 (lane / 32) + e of the wave's 32 channels sits in accumulator element 4 g + e
    floatx16 biasv;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) biasv[g * 4 + e] = p.bias[chh * 32 + 8 * g + 4 * hi32 + e];

    unsigned pv[NPW];
    unsigned lv = 0;        // DYN: bit i = tile (next body + i) is live
    int fetched = 0;        // DYN, lane 0 of wave 0: the counter value behind the chunk id in flight
    if (DYN) {              // chunks 0 and 1 of this workgroup are static; both ids go into the ring for the iterators' hops (every wave writes the same two words)
        ring[0] = bid; ring[1] = (int)gridDim.x + bid;
    }
    TileIt itn = DYN ? chunk_it(bid) : tile_it(pt_first), ito = itn;      // itn: the tile whose pieces are issued next (t + 2); ito: the tile whose results are stored next (t - 1)
    {
        const TileAt t0 = tile_at(itn);
        tile_next(itn, 0);
        const TileAt t1 = tile_at(itn);
        tile_next(itn, 1);
        piece_offsets(t0, pv);
#pragma unroll
        for (int i = 0; i < NPW; ++i) issue_piece(0, pv[i], i);
        piece_offsets(t1, pv);
#pragma unroll
        for (int i = 0; i < NPW; ++i) issue_piece(1, pv[i], i);
        const TileAt t2 = tile_at(itn);
        piece_offsets(t2, pv);      // tile 2's, issued by the first tile body; every body leaves the next one's behind (itn stays on the last tile it has priced)
        lv = 1u | ((unsigned)t1.live << 1) | ((unsigned)t2.live << 2);      // (tile 0 is live: the grid has at most one workgroup per chunk)
    }

    // fragment base of this lane inside a patch buffer: four 32-pixel MFMA tiles of two image rows each, rows 8 pxh + 2j, + 1
    const int plane_off = (pxh * 8 + (l31 >> 4)) * RP + (l31 & 15) * PIXB + hi32 * 16;
    half_t* outp = (half_t*)p.out;
    typedef __attribute__((ext_vector_type(4))) unsigned uint4v;
    typedef __attribute__((ext_vector_type(2))) float f2;
    typedef __attribute__((ext_vector_type(2))) _Float16 half2v;
    constexpr int NST = 8;                   // 16-byte stores per lane and tile (4 MFMA tiles x 2 group pairs): every one is issued, for every tile (the launcher
                                             // admits maps of whole 16 x 16 tiles only), so that `s_waitcnt vmcnt` can COUNT them next to the pieces
    const unsigned ovoff = (unsigned)(((((pxh * 8 + (l31 >> 4)) * p.W + (l31 & 15)) * p.ldout) + chh * 32 + 8 * hi32) * 2);
    auto out_base = [&](const TileIt& it) -> char* { return (char*)outp + ((size_t)it.P * p.ldout + p.cout_off) * 2; };
    const int jstep = 2 * p.W * p.ldout * 2;      // bytes between the row pairs of consecutive MFMA tiles

    // ---- the epilogue of a finished tile as a MICRO-PROGRAM, one step per MFMA slot of the next tile.  What a wave can issue in the shadow of its own MFMA
    // (scripts/ubench/issue_classes.hip, profiles/r03_issue_classes.txt; a bare MFMA slot is 15.6 ns): up to two plain fp32 / integer VALU instructions and up
    // to two transcendentals cost nothing; ONE packed-fp32 instruction (v_pk_mul / add / fma_f32) costs +9 ns -- they hold the matrix pipe off, from either
    // wave of a SIMD -- and the compiler's SLP vectoriser had turned every epilogue of this library into them; one ds_read_b128 costs +4.7 ns (four waves x 1 KiB
    // per 32-cycle slot IS the LDS bandwidth of a CU).  The compiler's own schedule had also put the fragment reads one MFMA ahead of their use (s_waitcnt
    // lgkmcnt(1) in front of every MFMA).  So the order is spelled out: slot k = 4 s + j of a tile holds MFMA (s, j), the LDS read of fragment (s + 2, j) into the
    // registers that MFMA just consumed, at most ONE transcendental and TWO plain VALU instructions of the previous tile's epilogue, now and then a piece or a
    // store; __builtin_amdgcn_sched_barrier(0) between slots; the translation unit is compiled with -fno-slp-vectorize.
    // A store group G (8 of them per tile) = the 8 accumulator elements 8 (G & 1) + v of MFMA tile G / 2:
    //   T = x * -log2(e) -> E = exp2(T) [trans] -> D = E + 1 -> R = rcp(D) [trans] -> Y = x * R -> fp16 pairs [cvt_pk] -> permlane32_swap x 2 -> one 16-byte store.
    // Period G = slots 4 + 16 G + i, i = 0 .. 15: E_v at i = v, R_v at i = 8 + v; the plain steps as their inputs appear (table below), T of group G + 1, and the tail
    // of group G - 1 (Y6, Y7, cvt3, swaps, store) in its first five slots.
    float T[8], ED[8], R[8], Y[8];
    unsigned Wd[4];
    decltype(__builtin_amdgcn_permlane32_swap(0u, 0u, false, false)) sw0, sw1;
    constexpr float NL2E = -1.44269504088896f;
    constexpr bool SILU = ACT == Y7T_ACT_SILU;
    auto xval = [&](const floatx16 (&a)[4], int G, int v) __attribute__((always_inline)) -> float { return a[G >> 1][8 * (G & 1) + v]; };
    auto cvt2 = [&](float lo, float hi) __attribute__((always_inline)) -> unsigned {
        const half2v h = {(half_t)lo, (half_t)hi};
        return __builtin_bit_cast(unsigned, h);
    };
    constexpr int EPI_PRE = 4;                      // slots 0 .. 3: T of group 0, two per slot
    constexpr int EPI_SLOTS = EPI_PRE + 16 * 8 + 5; // the last step (group 7's store) is slot 136 of 144
    auto epi_step = [&](const floatx16 (&prev)[4], char* ob, int k) __attribute__((always_inline)) {
        if (k < EPI_PRE) {
            if (SILU) { T[2 * k] = xval(prev, 0, 2 * k) * NL2E; T[2 * k + 1] = xval(prev, 0, 2 * k + 1) * NL2E; }
            return;
        }
        const int kk = k - EPI_PRE, G = kk >> 4, i = kk & 15;
        if (G < 8) {
            if (SILU) {
                if (i < 8) ED[i] = __builtin_amdgcn_exp2f(T[i]);
                else R[i - 8] = __builtin_amdgcn_rcpf(ED[i - 8]);
                if (i >= 1 && i <= 8) ED[i - 1] = ED[i - 1] + 1.0f;                                   // D_v one slot behind E_v
                if (i >= 9 && i <= 14) Y[i - 9] = xval(prev, G, i - 9) * R[i - 9];                    // Y_0 .. Y_5 one slot behind R_v
                if (G < 7) {                                                                           // T of the next group: slots 5 - 10, 12, 14
                    const int tv = (i >= 5 && i <= 10) ? i - 5 : i == 12 ? 6 : i == 14 ? 7 : -1;
                    if (tv >= 0) T[tv] = xval(prev, G + 1, tv) * NL2E;
                }
            } else {
                if (i >= 9 && i <= 14) Y[i - 9] = act_t<ACT>(xval(prev, G, i - 9));
            }
            if (i == 11) Wd[0] = cvt2(Y[0], Y[1]);
            if (i == 13) Wd[1] = cvt2(Y[2], Y[3]);
            if (i == 15) Wd[2] = cvt2(Y[4], Y[5]);
        }
        if (G >= 1 && G <= 8) {      // the tail of group G - 1
            const int Gp = G - 1;
            if (i == 0) {
                Y[6] = SILU ? xval(prev, Gp, 6) * R[6] : act_t<ACT>(xval(prev, Gp, 6));
                Y[7] = SILU ? xval(prev, Gp, 7) * R[7] : act_t<ACT>(xval(prev, Gp, 7));
            }
            if (i == 1) Wd[3] = cvt2(Y[6], Y[7]);
            if (i == 2) sw0 = __builtin_amdgcn_permlane32_swap(Wd[0], Wd[2], false, false);      // pairs (e 0,1) of the two 4-row groups
            if (i == 3) sw1 = __builtin_amdgcn_permlane32_swap(Wd[1], Wd[3], false, false);      // pairs (e 2,3)
            if (i == 4) {
                const uint4v v4 = {sw0[0], sw1[0], sw0[1], sw1[1]};
                if (ABL & 16) asm volatile("" ::"v"(v4));
                else *(uint4v*)(ob + (size_t)(Gp >> 1) * jstep + (Gp & 1) * 32 + ovoff) = v4;
            }
        }
    };

    // One tile: its 144 MFMAs into `cur` (the bias as the C operand of the first), the PREVIOUS tile's epilogue out of `prev`, tile t+2's pieces.
    // vmcnt at the top: younger than this wave's pieces of tile t are what tile t-1 issued -- NPW pieces (tile t+1's) and, if tile t-1 had a predecessor to
    // store, NST stores -- and at most two late stores of tile t-2 (waiting for those as well is harmless).
    // DYN: EVEN = t is even, i.e. the first tile of chunk t / 2 of this workgroup: behind its barrier lane 0 of wave 0 asks for chunk t / 2 + 2; the odd body after it
    // publishes the answer in front of ITS barrier (the counted wait has covered the atomic: it is older than every piece and store of the even body).
    auto tile_body = [&](const bool FIRST, const bool EVEN, int t, int buf, floatx16 (&cur)[4], floatx16 (&prev)[4]) __attribute__((always_inline)) {
        const int nbuf = (buf + 2 >= C::NBUF) ? buf + 2 - C::NBUF : buf + 2;
        char* const ob = FIRST ? nullptr : out_base(ito);
        if (!FIRST) tile_next(ito, t >> 1);            // (DYN: a hop at the top of an even body t, into chunk t / 2)
        const char* pb = smem + buf * C::PATCH_BYTES + plane_off;
        if (!(ABL & 8)) {
#if defined(Y7T_CONVSIM)
        if (DYN && !EVEN && wave == 0 && lane == 0) ring[((t >> 1) + 2) & 3] = 2 * (int)gridDim.x + fetched;
        __builtin_amdgcn_s_barrier();
        if (DYN && EVEN && wave == 0 && lane == 0) fetched = atomicAdd(p.tile_ctr, 1);
#else
        if (FIRST || t < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW + NST) : "memory");
        if (DYN && !EVEN && wave == 0) {
            if (lane == 0) {
                asm volatile("v_accvgpr_read_b32 %0, a255" : "=v"(fetched) : : "memory");      // (volatile asm statements keep their order: behind the counted wait)
                ring[((t >> 1) + 2) & 3] = 2 * (int)gridDim.x + fetched;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        // The fetch as an asm statement: the compiler's atomic optimiser turns atomicAdd on a uniform address into mbcnt + atomic + readfirstlane and WAITS for the result
        // on the spot (s_waitcnt vmcnt(0): every piece in flight + one L2 round trip, every other tile).  Spelled out, the result arrives whenever it arrives -- in a
        // register the compiler must not know about, or it copies it (to make room, or into the operand register of the statement that consumes it) BEFORE the value
        // is there: both were seen in the first build.  So the destination is a255 by name (an atomic's data and destination are both ACC registers or both VGPRs:
        // a254 carries the 1), which the allocator -- lowest numbers first, ~200 ACC registers in use -- never reaches; tests/test_ws_isa.py pins that no other
        // instruction of the kernel names a254 / a255.
        if (DYN && EVEN && wave == 0 && lane == 0)
            asm volatile("v_accvgpr_write_b32 a254, 1\n\ts_nop 4\n\tglobal_atomic_add a255, %0, a254, %1 sc0" : : "v"(0), "s"(p.tile_ctr) : "memory", "a254", "a255");
#endif
        }          // everybody's pieces of tile t are visible; nobody reads the buffer of tile t-1 any more: it takes tile t+2
        half8 xf[2][4];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[s][j] = *(const half8*)(pb + (s & 3) * 32 + j * 2 * RP);
        __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(full)
        for (int s = 0; s < C::NSUB; ++s)
#pragma clang loop unroll(full)
        for (int j = 0; j < 4; ++j) {
            const int k = s * 4 + j;
#if defined(Y7T_CONVSIM)
            cur[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s], xf[s & 1][j], s == 0 ? biasv : cur[j], 0, 0, 0);
#else       // the MFMA spelled out: accumulators in VGPRs (the epilogue's packed-fp32 operations read them directly), the stationary weights in ACC registers
            // (the builtin keeps the weights in VGPRs and the accumulators in ACC registers -- one v_accvgpr_read per epilogue value -- or, in VGPR form,
            // copies every weight fragment out of the ACC file before its substep).  What the compiler cannot see through the asm is covered by construction:
            // an accumulator is rewritten four MFMAs (>= 96 cycles) later and read by VALU code a whole tile later.
            if (s == 0) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(cur[j]) : "a"(wreg[s]), "v"(xf[s & 1][j]), "v"(biasv));
            else asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(cur[j]) : "a"(wreg[s]), "v"(xf[s & 1][j]));
#endif
            if (s + 2 < C::NSUB && !(ABL & 4)) {          // fragment (s + 2, j): tap (kh, kw), 16-channel group ks -- into the registers this MFMA has just read
                const int sn = s + 2, tap = sn >> 2, ks = sn & 3, kh = tap / 3, kw = tap - kh * 3;
                xf[s & 1][j] = *(const half8*)(pb + kh * RP + kw * PIXB + ks * 32 + j * 2 * RP);
            }
            if (!FIRST && !(ABL & 1)) epi_step(prev, ob, k);
            {       // tile t+2's pieces: two per period of the micro-program, in slots that carry one plain instruction
                const int kk = k - EPI_PRE, G = kk >> 4, i = kk & 15;
                if (!(ABL & 2) && k >= EPI_PRE && (i == 11 || i == 15) && G * 2 + (i == 15) < NPW) issue_piece(nbuf, pv[G * 2 + (i == 15)], G * 2 + (i == 15));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // The source offsets of the NEXT body's pieces (tile t + 3), here rather than at its top: >= 13 vector instructions that touch no accumulator, between the
        // tile's last MFMAs and whatever follows the body.  The compiler cannot see the MFMAs inside the asm statements, so it inserts no wait states for them --
        // and at the loop's exit it copies one accumulator set onto the other's registers (VALU writes of registers the last two MFMAs are still writing:
        // a write-after-write race that corrupted the last tile of a workgroup on the device; 8-pass MFMA -> VALU needs 11 wait states).
        tile_next(itn, (t + 3) >> 1);                  // (DYN: a hop at the bottom of an odd body t, into the chunk of tile t + 3)
        {
            const TileAt tn = tile_at(itn);
            piece_offsets(tn, pv);
            if (DYN) lv = (lv >> 1) | ((unsigned)tn.live << 2);
        }
        asm volatile("s_nop 7" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    floatx16 accA[4], accB[4];
    tile_body(true, true, 0, 0, accA, accB);
    int buf = 1;
    if (!DYN) {
    for (int t = 1; t < nt; t += 2) {
        tile_body(false, false, t, buf, accB, accA);
        buf = (buf + 1 == C::NBUF) ? 0 : buf + 1;
        if (t + 1 < nt) {
            tile_body(false, true, t + 1, buf, accA, accB);
            buf = (buf + 1 == C::NBUF) ? 0 : buf + 1;
        }
    }
    } else {      // until the first dead tile (lv bit 0 = the next body's tile is live; dead tiles are never followed by live ones).  Same loop shape as the static form:
                  // the register allocation of this kernel is only as good as its control flow is simple
        nt = 1;
        while (lv & 1) {
            tile_body(false, false, nt, buf, accB, accA);
            buf = (buf + 1 == C::NBUF) ? 0 : buf + 1;
            ++nt;
            if (lv & 1) {
                tile_body(false, true, nt, buf, accA, accB);
                buf = (buf + 1 == C::NBUF) ? 0 : buf + 1;
                ++nt;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");      // the tail's zero-filling pieces have landed before this workgroup's LDS is
                                                                                           // handed on; the last MFMAs' results are readable by VALU code
    // the last tile's results (its stores overlap nothing): the same micro-program, back to back
    char* const obl = out_base(ito);
    if (nt & 1) {
#pragma unroll
        for (int k = 0; k < EPI_SLOTS; ++k) epi_step(accA, obl, k);
    } else {
#pragma unroll
        for (int k = 0; k < EPI_SLOTS; ++k) epi_step(accB, obl, k);
    }
    if (DYN && tid == 0) {      // every fetch of this workgroup has returned (vmcnt(0) above); the last workgroup to leave hands the counter back at zero
        if (atomicAdd(p.tile_ctr + Y7T_TILE_CTR_DONE, 1) == (int)gridDim.x - 1) { p.tile_ctr[0] = 0; p.tile_ctr[Y7T_TILE_CTR_DONE] = 0; }
    }
#endif
}

}   // namespace

// korder 5 layers only (detector/graph.py::ws_eligible mirrors the conditions): 3x3 / 1 / 1, Cin == 64, Cout_pad == 64, fp16 output in 16-byte pieces
int y7t_conv_ws_launch(const Y7TConvArgs& a, hipStream_t s) {
    using C = WsCfg;
    const bool ok = a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.Cin == 64 && a.Cout_pad == 64 && !a.out_f32 && !(a.Cout & 7) && !(a.ldout & 7) &&
                    !(a.cout_off & 7) && !(a.ldin & 7) && !(a.cin_off & 7) && a.Ho == a.H && a.Wo == a.W && a.in_bytes < 0x80000000u && a.Cout == 64 &&      // (32-bit byte offsets: tensors below 2 GiB, as y7t_conv_launch enforces)
                    a.H % C::TH == 0 && a.W % C::TW == 0;      // whole tiles only: the kernel counts its stores (s_waitcnt vmcnt), none may be predicated off
    if (!ok) {
        y7t_set_error("conv: weights are in register-fragment order (korder 5) but the layer is not a 3x3 / stride 1 / 64 -> 64 convolution on a map of whole 16 x 16 tiles with an aligned fp16 output");
        return Y7T_E_ARG;
    }
    static Y7TOncePerDevice attr;
    if (int e = y7t_once_per_device(attr, []() -> int {
            Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_c64_ws<Y7T_ACT_NONE>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
            Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_c64_ws<Y7T_ACT_SILU>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
            Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_c64_ws<Y7T_ACT_LEAKY>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
            Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_c64_ws<Y7T_ACT_NONE, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
            Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_c64_ws<Y7T_ACT_SILU, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
            Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_c64_ws<Y7T_ACT_LEAKY, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
            return 0;
        })) return e;
    static int wgs_env = -2;      // Y7T_CONV_WS_WGS: workgroups of a launch (default: one persistent workgroup per compute unit -- 150 KiB of LDS each)
    if (wgs_env == -2) wgs_env = y7t_exp_switch("Y7T_CONV_WS_WGS", -1);
    const int ncu = wgs_env > 0 ? wgs_env : y7t_num_cus();
    const int ptiles = a.B * ((a.H + C::TH - 1) / C::TH) * ((a.W + C::TW - 1) / C::TW);
    static int dyn_env = -1;      // Y7T_CONV_WS_DYN=0: static partition although the caller supplied a tile counter (A/B)
    if (dyn_env < 0) dyn_env = y7t_switch("Y7T_CONV_WS_DYN", 1);
    // (the kernel's quotients by tiles_x / tiles_y are __umulhi(n, 0xFFFFFFFF / d + 1): exact for n * d < 2^32 -- d <= 1024 here: maps of <= 16384 pixels a side)
    const bool dyn = a.tile_ctr && dyn_env && ptiles < (1 << 22) && a.H <= 16384 && a.W <= 16384;
    const int nchunks = (ptiles + C::CH - 1) / C::CH;
    const int grid = dyn ? (nchunks < ncu ? nchunks : ncu) : (ptiles < ncu ? ptiles : ncu);
#if Y7T_ABLATE      // liby7t_ablate.so only: the timing ablations of the SiLU instance (wrong results; scripts/ws_probe.py)
    static int abl = -1;
    if (abl < 0) abl = y7t_exp_switch("Y7T_WS_ABLATE", 0);
    if (abl && a.act == Y7T_ACT_SILU) {
#define Y7T_WS_ABL_CASE(N) \
        case N: \
            Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_c64_ws<Y7T_ACT_SILU, N>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS)); \
            hipLaunchKernelGGL((k_conv3x3_c64_ws<Y7T_ACT_SILU, N>), dim3(grid), dim3(256), C::LDS, s, a); \
            break;
        switch (abl) {
            Y7T_WS_ABL_CASE(1) Y7T_WS_ABL_CASE(2) Y7T_WS_ABL_CASE(3) Y7T_WS_ABL_CASE(4) Y7T_WS_ABL_CASE(8) Y7T_WS_ABL_CASE(16) Y7T_WS_ABL_CASE(18)
            default: y7t_set_error("conv: unknown Y7T_WS_ABLATE value"); return Y7T_E_ARG;
        }
#undef Y7T_WS_ABL_CASE
        Y7T_LAUNCH_CHECK();
        y7t_note_kernel("ws64<16,16> ablated");
        return 0;
    }
#endif
    if (dyn) {
        if (a.act == Y7T_ACT_SILU) hipLaunchKernelGGL((k_conv3x3_c64_ws<Y7T_ACT_SILU, 0, true>), dim3(grid), dim3(256), C::LDS, s, a);
        else if (a.act == Y7T_ACT_LEAKY) hipLaunchKernelGGL((k_conv3x3_c64_ws<Y7T_ACT_LEAKY, 0, true>), dim3(grid), dim3(256), C::LDS, s, a);
        else hipLaunchKernelGGL((k_conv3x3_c64_ws<Y7T_ACT_NONE, 0, true>), dim3(grid), dim3(256), C::LDS, s, a);
        Y7T_LAUNCH_CHECK();
        y7t_note_kernel("ws64<16,16> dyn");
        return 0;
    }
    if (a.act == Y7T_ACT_SILU) hipLaunchKernelGGL((k_conv3x3_c64_ws<Y7T_ACT_SILU>), dim3(grid), dim3(256), C::LDS, s, a);
    else if (a.act == Y7T_ACT_LEAKY) hipLaunchKernelGGL((k_conv3x3_c64_ws<Y7T_ACT_LEAKY>), dim3(grid), dim3(256), C::LDS, s, a);
    else hipLaunchKernelGGL((k_conv3x3_c64_ws<Y7T_ACT_NONE>), dim3(grid), dim3(256), C::LDS, s, a);
    Y7T_LAUNCH_CHECK();
    y7t_note_kernel("ws64<16,16>");
    return 0;
}
