// y7t_conv_ws128.hip -- 3x3 / stride 1 / pad 1 Conv(+folded BN)+bias+activation for the 128 -> 128 k layers, WEIGHTS STATIONARY IN REGISTERS.
//
// History: written at the end of round 3, measured in round 4 (profiles/r04_ws128_measurement.txt): alone on the chip 5-19 % faster than the LDS-patch kernels on its 11
// launches (-176 us per list), in the pipeline 0.2 ms SLOWER -- a statically partitioned persistent workgroup loses to whatever shares its CU -- and deleted.  Round 5
// brings it back ON THE TILE COUNTER (DYN, see y7t_conv_ws.hip: chunks of two x-adjacent tiles from Y7TConvArgs::tile_ctr, one counter per 128-channel output tile,
// fetched two chunks ahead by an asm atomic whose result lands in a255, published through a four-entry LDS ring).
//
// Same math as k_conv3x3_c64_ws (y7t_conv_ws.hip; /root/reference/models/common.py:99-111 after utils/torch_utils.py:181-201), for the ELAN branch convolutions one
// level down (cfg/deploy/yolov7-w6.yaml: 128 -> 128 at 160x160 and 80x80, 128 -> 256 in the head): K = 1152.  Why it might pay (profiles/r03_ws64_probe_and_ablations.txt,
// r03_patch_ablations.txt): with one wave per SIMD what costs the 64 -> 64 kernel a third of its time is the ISSUE of its vector-memory instructions -- 21 per wave
// and 144 MFMAs -- and the LDS-patch kernel these layers run on spends 14-19 % on its patch pieces, ~22 % on fragment reads behind a barrier every 16 MFMAs, and
// streams 8 KiB of weights per K-step.  Here a wave issues 12 vector-memory instructions per 144 MFMAs (8 pieces + 4 stores) and no weight traffic at all:
//   * the filter bank of a 128-channel output tile -- 128 x 1152 fp16 = 288 KiB as MFMA A-fragments -- lives in the REGISTERS of a persistent workgroup: wave q keeps
//     the 72 fragments of its 32 output channels (288 registers per lane: 256 ACC registers + 32 arch VGPRs; gfx950 has 512 per lane at one wave per SIMD);
//   * all four waves multiply the SAME 64 pixels (a 4 x 16 tile: two 32-pixel MFMA tiles of two image rows each), so the accumulators are 2 x 16 registers per
//     set and a second set fits: the previous tile's epilogue (32 values per lane) runs as a micro-program between this tile's MFMAs, as in the 64 -> 64 kernel;
//   * the 6 x 18 pixel x 128-channel patch (272-byte pixels: 256 data + 16 pad; 5 KiB rows) sits in a three-buffer LDS ring (3 x 30 KiB), one barrier per tile;
//   * one ds_read_b128 per MFMA (pixel fragments), three substeps ahead, every address = lane base + immediate.
// Weight layout (korder 6, detector/weights.py::pack_ws128): per 128-channel output tile n, fragment f = (n * 72 + tap * 8 + ks) * 4 + q is 1 KiB, lane l holds
// W[n*128 + q*32 + l%32][tap][ks*16 + 8*(l/32) .. +7].
#include "y7t_common.h"
#include "y7t_conv_common.h"
#include <stdlib.h>

namespace {

constexpr unsigned kOOB = 0xFF000000u;

// S2 (round 5): the same kernel for the 3x3 / STRIDE 2 layers with 128 input channels (cfg/deploy/yolov7-w6.yaml:29 and :116: 128 -> 256 at 320 x 320 and at 160 x 160): an
// output tile of 2 x 16 pixels (ONE 32-pixel MFMA tile per wave: 72 MFMAs per tile), its 5 x 33-pixel input patch with the columns split by parity (17 even ones, then 16
// odd ones: tap kw reads a unit-stride run of 16 -- the layout of y7t_conv_ws_s2.hip), 45 KiB per buffer.  Everything else -- filter bank in registers, two accumulator
// sets, epilogue micro-program, three-buffer ring, tile counter -- is the stride-1 kernel's.
template <bool S2>
struct Ws128CfgT {
    static constexpr int S = S2 ? 2 : 1;
    static constexpr int TW = 16, TH = S2 ? 2 : 4;                    // OUTPUT tile
    static constexpr int PIXB = 272;                                  // 128 channels x 2 B + 16 B pad (17 sixteen-byte slots: odd -> conflict-free column reads)
    static constexpr int ROWS = S2 ? 5 : TH + 2, COLS = S2 ? 33 : TW + 2;   // input patch
    static constexpr int NE = 17;                                     // S2: even patch columns come first, the 16 odd ones behind them
    static constexpr int O_OFF = NE * PIXB;
    static constexpr int RP = S2 ? 9216 : 5120;                       // patch row pitch: 33 (18) x 272 rounded up to a multiple of 256 B
    static constexpr int PATCH_DMA = (ROWS * RP + 1023) / 1024;       // 45 (30) wave-wide 1 KiB pieces
    static constexpr int PATCH_BYTES = PATCH_DMA * 1024;
    static constexpr int NPW = (PATCH_DMA + 3) / 4;                   // pieces per wave per tile (12 / 8; a slot past the patch repeats its last KiB)
    static constexpr int NBUF = 3;
    static constexpr int RING_OFF = NBUF * PATCH_BYTES;                // DYN: the chunk ids handed to this workgroup, a ring of four
    static constexpr int LDS = RING_OFF + 64;
    static constexpr int CH = 2;                                      // DYN: tiles per chunk of the tile counter
    static constexpr int MAX_NT = Y7T_TILE_CTR_DONE;                  // DYN: one counter per 128-channel output tile; tile_ctr[Y7T_TILE_CTR_DONE] counts the leavers
    static constexpr int NSUB = 72;                                   // k16 substeps per tile: 9 taps x 8
    static constexpr int NJ = S2 ? 1 : 2;                             // 32-pixel MFMA tiles per wave and tile
    static constexpr int NACCW = 63;                                  // weight fragments kept in ACC registers (the other 9 in arch VGPRs; a252 .. a255 stay free: the tile counter's fetch lands in a255)
};
typedef Ws128CfgT<false> Ws128Cfg;

template <int ACT, bool DYN = false, bool S2 = false>
__global__ void __launch_bounds__(256, 1) k_conv3x3_c128_ws(const Y7TConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C = Ws128CfgT<S2>;
    constexpr int TW = C::TW, TH = C::TH, PIXB = C::PIXB, RP = C::RP, NPW = C::NPW, NJ = C::NJ, S = C::S;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi32 = lane >> 5;

    // ---- this workgroup's output-channel tile and its contiguous range of pixel tiles (workgroup b: channel tile b % n_nt, so that the workgroups of an XCD --
    // every eighth -- cover all channel tiles of the same pixel ranges) ----
    const int tiles_x = p.Wo / TW, tiles_y = p.Ho / TH, ptiles = p.B * tiles_y * tiles_x;      // (tiles of the OUTPUT map)
    const int n_nt = p.Cout_pad >> 7;
    const int ntile = (int)blockIdx.x % n_nt, wg = (int)blockIdx.x / n_nt, nwg = ((int)gridDim.x + n_nt - 1 - ntile) / n_nt;
    const int per = (ptiles + nwg - 1) / nwg;
    const int pt_first = DYN ? 0 : wg * per;
    int nt = DYN ? 0 : ((ptiles - pt_first) < per ? (ptiles - pt_first) : per);
    if (!DYN && nt <= 0) return;
    constexpr int CH = C::CH;
    const unsigned magic_x = 0xFFFFFFFFu / (unsigned)tiles_x + 1u, magic_y = 0xFFFFFFFFu / (unsigned)tiles_y + 1u;      // exact quotients for the chunk ids of a launch (the launcher checks)
    volatile LDS_AS int* const ring = (volatile LDS_AS int*)(smem + C::RING_OFF);

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);

    // tile coordinates are stepped, not decoded (as in y7t_conv_ws.hip): P = index of the tile's first pixel in the NHWC map, (ty, tx) its place in the tile grid
    // (Pi: index of the tile's first INPUT pixel (S h0, S w0); the same as P at stride 1)
    struct TileIt { int P, Pi, ty, tx, n; };
    auto tile_it = [&](int pt) -> TileIt {
        int q = pt;
        const int txi = q % tiles_x; q /= tiles_x;
        const int tyi = q % tiles_y, b = q / tiles_y;
        return TileIt{(b * p.Ho + tyi * TH) * p.Wo + txi * TW, (b * p.H + tyi * TH * S) * p.W + txi * TW * S, tyi, txi, pt_first + nt - pt};
    };
    auto chunk_it = [&](int id) __attribute__((always_inline)) -> TileIt {      // (id wave-uniform) -- every chunk is walked as CH tiles; a tile past the batch is dead
        const unsigned pt = (unsigned)id * CH;
        const unsigned q = tiles_x == 1 ? pt : __umulhi(pt, magic_x), txi = pt - q * (unsigned)tiles_x;
        const unsigned b = tiles_y == 1 ? q : __umulhi(q, magic_y), tyi = q - b * (unsigned)tiles_y;
        return TileIt{(int)((b * (unsigned)p.Ho + tyi * TH) * (unsigned)p.Wo + txi * TW), (int)((b * (unsigned)p.H + tyi * (TH * S)) * (unsigned)p.W + txi * (TW * S)), (int)tyi, (int)txi, CH};
    };
    auto tile_next = [&](TileIt& it, int hop) __attribute__((always_inline)) {
        if (DYN && it.n == 1) {      // the chunk is used up: the id of chunk `hop` of this workgroup's sequence is in the ring (written >= one barrier ago)
            it = chunk_it(__builtin_amdgcn_readfirstlane(ring[hop & 3]));
            return;
        }
        it.P += TW; it.Pi += TW * S; it.n -= 1;
        if (++it.tx == tiles_x) { it.tx = 0; it.P += (TH - 1) * p.Wo; it.Pi += (TH * S - 1) * p.W; if (++it.ty == tiles_y) it.ty = 0; }
    };
    struct TileAt { int org, ty, tx; bool live; };      // org: byte offset of the patch's first pixel (S h0 - 1, S w0 - 1)
    const unsigned npix = (unsigned)(p.B * p.Ho * p.Wo);
    auto tile_at = [&](const TileIt& it) -> TileAt {      // (unsigned arithmetic: P keeps stepping past the last live tile)
        return TileAt{(int)((((unsigned)it.Pi - (unsigned)p.W - 1u) * (unsigned)p.ldin + (unsigned)p.cin_off) * 2u), it.ty, it.tx, DYN ? (unsigned)it.P < npix : it.n > 0};
    };
    // per-lane constants of piece i: where its 16-byte slot sits inside the 6 x 18 patch, and which halo sides it lies on (4 bits per piece)
    unsigned pconst[NPW], pedge[NPW];      // (the halo sides of piece i: one word per piece -- twelve pieces at stride 2 do not fit four bits each into one register)
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        int I = wave + 4 * i;
        if (I >= C::PATCH_DMA) I = C::PATCH_DMA - 1;
        const int byte = I * 1024 + lane * 16;
        const int r = byte / RP, rb = byte - r * RP;
        const int x = rb / PIXB, cs = (rb - x * PIXB) >> 4;
        const bool used = r < C::ROWS && x < C::COLS && cs < 16;
        const int pc = !S2 ? x : x < C::NE ? 2 * x : 2 * (x - C::NE) + 1;      // patch column of this slot (S2: the even columns first)
        pconst[i] = used ? (unsigned)(((r * p.W + pc) * p.ldin + cs * 8) * 2) : 0u;
        // halo sides: top row, bottom row, first column, last column -- at stride 2 (even input map, pad 1) only the top row and the first column can lie outside
        const unsigned e = used ? (S2 ? (unsigned)((r == 0) | ((pc == 0) << 2)) : (unsigned)((r == 0) | ((r == TH + 1) << 1) | ((x == 0) << 2) | ((x == TW + 1) << 3))) : 0u;
        pedge[i] = e;
    }
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
            for (int i = 0; i < NPW; ++i) pv[i] = (pedge[i] & tmask) ? kOOB : pconst[i] + (unsigned)ta.org;
        }
    };
    auto issue_piece = [&](int buf, unsigned v, int i) __attribute__((always_inline)) {
        const int I = (wave + 4 * i < C::PATCH_DMA) ? wave + 4 * i : C::PATCH_DMA - 1;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (LDS_AS void*)(smem + buf * C::PATCH_BYTES + I * 1024), 16, v, 0, 0, 0);
    };

    // ---- the filter bank: wave q owns output channels 128 ntile + 32 q .. + 31 -- 72 A-fragments per lane, straight from memory, resident for the whole launch ----
    const int q4 = wave;
    half8 wreg[C::NSUB];
    {
        const half8* wp = (const half8*)p.w + ((size_t)ntile * C::NSUB * 4 + q4) * 64 + lane;
#pragma unroll
        for (int f = 0; f < C::NSUB; ++f) wreg[f] = wp[f * 256];
    }
    // the bias enters as the C operand of a tile's first MFMA: row 8 g + 4 (lane / 32) + e of the wave's 32 channels sits in accumulator element 4 g + e
    floatx16 biasv;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) biasv[g * 4 + e] = p.bias[ntile * 128 + q4 * 32 + 8 * g + 4 * hi32 + e];

    unsigned pv[NPW];
    unsigned lv = 0;        // DYN: bit i = tile (next body + i) is live
    int fetched = 0;        // DYN, lane 0 of wave 0 (host simulator only: on the device the value travels through a255)
    if (DYN) {              // chunks 0 and 1 of this workgroup are static; both ids go into the ring for the iterators' hops (every wave writes the same two words)
        ring[0] = wg; ring[1] = nwg + wg;
    }
    TileIt itn = DYN ? chunk_it(wg) : tile_it(pt_first), ito = itn;      // itn: the last tile whose pieces have been priced (t + 2); ito: the tile whose results are stored next (t - 1)
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
        piece_offsets(t2, pv);      // tile 2's, issued by the first tile body; every body leaves the next one's behind
        lv = 1u | ((unsigned)t1.live << 1) | ((unsigned)t2.live << 2);
    }
    int* const ctr = DYN ? p.tile_ctr + ntile : nullptr;

    // fragment base of this lane inside a patch buffer: two 32-pixel MFMA tiles of two image rows each, rows 2 j, + 1
    const int plane_off = (l31 >> 4) * (S * RP) + (l31 & 15) * PIXB + hi32 * 16;      // (S2: output row r reads patch rows 2 r + kh)
    half_t* outp = (half_t*)p.out;
    typedef __attribute__((ext_vector_type(4))) unsigned uint4v;
    typedef __attribute__((ext_vector_type(2))) _Float16 half2v;
    constexpr int NST = 2 * NJ;              // 16-byte stores per lane and tile (NJ MFMA tiles x 2 group pairs): every one is issued, for every tile (whole tiles only)
    const unsigned ovoff = (unsigned)(((((l31 >> 4) * p.Wo + (l31 & 15)) * p.ldout) + ntile * 128 + q4 * 32 + 8 * hi32) * 2);
    auto out_base = [&](const TileIt& it) -> char* { return (char*)outp + ((size_t)it.P * p.ldout + p.cout_off) * 2; };
    const int jstep = 2 * p.Wo * p.ldout * 2;      // bytes between the row pairs of consecutive MFMA tiles

    // ---- the epilogue micro-program of y7t_conv_ws.hip with four store groups instead of eight: group G = the 8 accumulator elements 8 (G & 1) + v of MFMA tile G / 2;
    // period G = slots 4 + 16 G + i: E_v at i = v, R_v at i = 8 + v, at most two plain fp32 instructions and one transcendental per slot ----
    float T[8], ED[8], R[8], Y[8];
    unsigned Wd[4];
    decltype(__builtin_amdgcn_permlane32_swap(0u, 0u, false, false)) sw0, sw1;
    constexpr float NL2E = -1.44269504088896f;
    constexpr bool SILU = ACT == Y7T_ACT_SILU;
    constexpr int NG = 2 * NJ;
    auto xval = [&](const floatx16 (&a)[NJ], int G, int v) __attribute__((always_inline)) -> float { return a[G >> 1][8 * (G & 1) + v]; };
    auto cvt2 = [&](float lo, float hi) __attribute__((always_inline)) -> unsigned {
        const half2v h = {(half_t)lo, (half_t)hi};
        return __builtin_bit_cast(unsigned, h);
    };
    constexpr int EPI_PRE = 4;                       // slots 0 .. 3: T of group 0, two per slot
    constexpr int EPI_SLOTS = EPI_PRE + 16 * NG + 5; // the last step (group 3's store) is slot 72 of 144
    auto epi_step = [&](const floatx16 (&prev)[NJ], char* ob, int k) __attribute__((always_inline)) {
        if (k >= EPI_SLOTS) return;
        if (k < EPI_PRE) {
            if (SILU) { T[2 * k] = xval(prev, 0, 2 * k) * NL2E; T[2 * k + 1] = xval(prev, 0, 2 * k + 1) * NL2E; }
            return;
        }
        const int kk = k - EPI_PRE, G = kk >> 4, i = kk & 15;
        if (G < NG) {
            if (SILU) {
                if (i < 8) ED[i] = __builtin_amdgcn_exp2f(T[i]);
                else R[i - 8] = __builtin_amdgcn_rcpf(ED[i - 8]);
                if (i >= 1 && i <= 8) ED[i - 1] = ED[i - 1] + 1.0f;
                if (i >= 9 && i <= 14) Y[i - 9] = xval(prev, G, i - 9) * R[i - 9];
                if (G < NG - 1) {
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
        if (G >= 1 && G <= NG) {      // the tail of group G - 1
            const int Gp = G - 1;
            if (i == 0) {
                Y[6] = SILU ? xval(prev, Gp, 6) * R[6] : act_t<ACT>(xval(prev, Gp, 6));
                Y[7] = SILU ? xval(prev, Gp, 7) * R[7] : act_t<ACT>(xval(prev, Gp, 7));
            }
            if (i == 1) Wd[3] = cvt2(Y[6], Y[7]);
            if (i == 2) sw0 = __builtin_amdgcn_permlane32_swap(Wd[0], Wd[2], false, false);
            if (i == 3) sw1 = __builtin_amdgcn_permlane32_swap(Wd[1], Wd[3], false, false);
            if (i == 4) {
                const uint4v v4 = {sw0[0], sw1[0], sw0[1], sw1[1]};
                *(uint4v*)(ob + (size_t)(Gp >> 1) * jstep + (Gp & 1) * 32 + ovoff) = v4;
            }
        }
    };
    constexpr int PIECE_SLOT0 = EPI_SLOTS + 3, PIECE_STRIDE = S2 ? 2 : 8;      // tile t+2's pieces: slots 76, 84, ..., 132 behind the micro-program, which ends at slot 72 (S2: slots 44, 46, ..., 66 of 72)
    static_assert(PIECE_SLOT0 + PIECE_STRIDE * (NPW - 1) < C::NSUB * NJ, "pieces fit behind the epilogue");

    // One tile: its 144 MFMAs into `cur` (the bias as the C operand of the first two), the PREVIOUS tile's epilogue out of `prev`, tile t+2's pieces.
    // vmcnt at the top: younger than this wave's pieces of tile t are exactly what tile t-1 issued: NST stores (if it had a predecessor) and NPW pieces.
    // DYN: EVEN = t is even = the first tile of chunk t / 2: behind its barrier lane 0 of wave 0 asks for chunk t / 2 + 2; the odd body after it publishes the answer in
    // front of ITS barrier (y7t_conv_ws.hip has the derivation)
    auto tile_body = [&](const bool FIRST, const bool EVEN, int t, int buf, floatx16 (&cur)[NJ], floatx16 (&prev)[NJ]) __attribute__((always_inline)) {
        const int nbuf = (buf + 2 >= C::NBUF) ? buf + 2 - C::NBUF : buf + 2;
        char* const ob = FIRST ? nullptr : out_base(ito);
        if (!FIRST) tile_next(ito, t >> 1);
        const char* pb = smem + buf * C::PATCH_BYTES + plane_off;
#if defined(Y7T_CONVSIM)
        if (DYN && !EVEN && wave == 0 && lane == 0) ring[((t >> 1) + 2) & 3] = 2 * nwg + fetched;
        __builtin_amdgcn_s_barrier();
        if (DYN && EVEN && wave == 0 && lane == 0) fetched = atomicAdd(ctr, 1);
#else
        if (FIRST || t < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW + NST) : "memory");
        if (DYN && !EVEN && wave == 0) {
            if (lane == 0) {
                asm volatile("v_accvgpr_read_b32 %0, a255" : "=v"(fetched) : : "memory");
                ring[((t >> 1) + 2) & 3] = 2 * nwg + fetched;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();          // everybody's pieces of tile t are visible; nobody reads the buffer of tile t-1 any more: it takes tile t+2
        if (DYN && EVEN && wave == 0 && lane == 0)
            asm volatile("v_accvgpr_write_b32 a254, 1\n\ts_nop 4\n\tglobal_atomic_add a255, %0, a254, %1 sc0" : : "v"(0), "s"(ctr) : "memory", "a254", "a255");
#endif
        half8 xf[3][NJ];
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int j = 0; j < NJ; ++j) xf[s][j] = *(const half8*)(pb + s * 32 + j * 2 * RP);      // substeps 0 .. 2: tap 0, channel groups 0 .. 2 (S2: NJ = 1)
        __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(full)
        for (int s = 0; s < C::NSUB; ++s)
#pragma clang loop unroll(full)
        for (int j = 0; j < NJ; ++j) {
            const int k = s * NJ + j;
#if defined(Y7T_CONVSIM)
            cur[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wreg[s], xf[s % 3][j], s == 0 ? biasv : cur[j], 0, 0, 0);
#else       // the MFMA spelled out (as in y7t_conv_ws.hip): accumulators in arch VGPRs, the first 64 weight fragments in ACC registers, the last 8 in arch VGPRs
            if (s == 0) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(cur[j]) : "a"(wreg[s]), "v"(xf[s % 3][j]), "v"(biasv));
            else if (s < C::NACCW) asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(cur[j]) : "a"(wreg[s]), "v"(xf[s % 3][j]));
            else asm("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(cur[j]) : "v"(wreg[s]), "v"(xf[s % 3][j]));
#endif
            if (s + 3 < C::NSUB) {          // fragment (s + 3, j): tap (kh, kw), 16-channel group ks -- into the registers this MFMA has just read
                const int sn = s + 3, tap = sn >> 3, ks = sn & 7, kh = tap / 3, kw = tap - kh * 3;
                const int kwoff = !S2 ? kw * PIXB : kw == 1 ? C::O_OFF : kw == 2 ? PIXB : 0;      // (S2: tap kw = 1 reads the odd plane, kw = 2 the even plane one column on)
                xf[s % 3][j] = *(const half8*)(pb + kh * RP + kwoff + ks * 32 + j * 2 * RP);
            }
            if (!FIRST) epi_step(prev, ob, k);
            if (k >= PIECE_SLOT0 && (k - PIECE_SLOT0) % PIECE_STRIDE == 0 && (k - PIECE_SLOT0) / PIECE_STRIDE < NPW)
                issue_piece(nbuf, pv[(k - PIECE_SLOT0) / PIECE_STRIDE], (k - PIECE_SLOT0) / PIECE_STRIDE);
            __builtin_amdgcn_sched_barrier(0);
        }
        // the next body's piece offsets here: >= 8 vector instructions + s_nop that touch no accumulator behind the tile's last MFMAs (the compiler cannot see the
        // MFMAs inside the asm statements and inserts no wait states for them: y7t_conv_ws.hip, tests/test_ws_isa.py)
        tile_next(itn, (t + 3) >> 1);
        {
            const TileAt tn = tile_at(itn);
            piece_offsets(tn, pv);
            if (DYN) lv = (lv >> 1) | ((unsigned)tn.live << 2);
        }
        asm volatile("s_nop 7\n\ts_nop 3" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    };

    floatx16 accA[NJ], accB[NJ];
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
    } else {      // until the first dead tile (same loop shape as the static form)
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
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
    char* const obl = out_base(ito);
    if (nt & 1) {
#pragma unroll
        for (int k = 0; k < EPI_SLOTS; ++k) epi_step(accA, obl, k);
    } else {
#pragma unroll
        for (int k = 0; k < EPI_SLOTS; ++k) epi_step(accB, obl, k);
    }
    if (DYN && tid == 0) {      // the last workgroup to leave hands every counter of the op back at zero
        if (atomicAdd(p.tile_ctr + C::MAX_NT, 1) == (int)gridDim.x - 1) {
            for (int i = 0; i <= C::MAX_NT; ++i) p.tile_ctr[i] = 0;
        }
    }
#endif
}

}   // namespace

// korder 6 layers only (detector/graph.py::ws128_eligible / ws128_s2_eligible mirror the conditions): 3x3 / pad 1, Cin == 128, Cout a multiple of 128; stride 1 on maps of
// whole 4 x 16 tiles, or stride 2 on an even map whose output is whole 2 x 16 tiles
template <bool S2>
static int ws128_go(const Y7TConvArgs& a, hipStream_t s) {
    using C = Ws128CfgT<S2>;
    static Y7TOncePerDevice attr;
    if (int e = y7t_once_per_device(attr, []() -> int {
            Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_c128_ws<Y7T_ACT_NONE, false, S2>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
            Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_c128_ws<Y7T_ACT_SILU, false, S2>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
            Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_c128_ws<Y7T_ACT_LEAKY, false, S2>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
            Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_c128_ws<Y7T_ACT_NONE, true, S2>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
            Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_c128_ws<Y7T_ACT_SILU, true, S2>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
            Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_c128_ws<Y7T_ACT_LEAKY, true, S2>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
            return 0;
        })) return e;
    const int ncu = y7t_num_cus();      // one persistent workgroup per compute unit (90 / 135 KiB of LDS, 512 registers per lane)
    const int n_nt = a.Cout_pad / 128;
    const int ptiles = a.B * (a.Ho / C::TH) * (a.Wo / C::TW);
    static int dyn_env = -1;      // Y7T_CONV_WS_DYN=0: static partition although the caller supplied tile counters (A/B)
    if (dyn_env < 0) dyn_env = y7t_switch("Y7T_CONV_WS_DYN", 1);
    // (the kernel's quotients by tiles_x / tiles_y are __umulhi(n, 0xFFFFFFFF / d + 1): exact for n * d < 2^32 -- d <= 1024 here: maps of <= 4096 pixels a side.  Through
    //  round 5 the bound was 60000 tiles, which the 320^2 stride-2 layer passes at 75 frames)
    const bool dyn = a.tile_ctr && dyn_env && n_nt <= C::MAX_NT && ptiles < (1 << 22) && a.H <= 4096 && a.W <= 4096;
    const int units = dyn ? (ptiles + C::CH - 1) / C::CH : ptiles;      // what a workgroup starts on: a chunk, or a tile
    int grid = units * n_nt < ncu ? units * n_nt : ncu;
    if (grid < n_nt) grid = n_nt;
    if (dyn) {
        if (a.act == Y7T_ACT_SILU) hipLaunchKernelGGL((k_conv3x3_c128_ws<Y7T_ACT_SILU, true, S2>), dim3(grid), dim3(256), C::LDS, s, a);
        else if (a.act == Y7T_ACT_LEAKY) hipLaunchKernelGGL((k_conv3x3_c128_ws<Y7T_ACT_LEAKY, true, S2>), dim3(grid), dim3(256), C::LDS, s, a);
        else hipLaunchKernelGGL((k_conv3x3_c128_ws<Y7T_ACT_NONE, true, S2>), dim3(grid), dim3(256), C::LDS, s, a);
        Y7T_LAUNCH_CHECK();
        y7t_note_kernel(S2 ? "ws128_s2<2,16> dyn" : "ws128<4,16> dyn");
        return 0;
    }
    if (a.act == Y7T_ACT_SILU) hipLaunchKernelGGL((k_conv3x3_c128_ws<Y7T_ACT_SILU, false, S2>), dim3(grid), dim3(256), C::LDS, s, a);
    else if (a.act == Y7T_ACT_LEAKY) hipLaunchKernelGGL((k_conv3x3_c128_ws<Y7T_ACT_LEAKY, false, S2>), dim3(grid), dim3(256), C::LDS, s, a);
    else hipLaunchKernelGGL((k_conv3x3_c128_ws<Y7T_ACT_NONE, false, S2>), dim3(grid), dim3(256), C::LDS, s, a);
    Y7T_LAUNCH_CHECK();
    y7t_note_kernel(S2 ? "ws128_s2<2,16>" : "ws128<4,16>");
    return 0;
}

int y7t_conv_ws128_launch(const Y7TConvArgs& a, hipStream_t s) {
    const bool common = a.KH == 3 && a.KW == 3 && a.pad == 1 && a.Cin == 128 && a.Cout_pad % 128 == 0 && a.Cout == a.Cout_pad && !a.out_f32 && !(a.ldout & 7) &&
                        !(a.cout_off & 7) && !(a.ldin & 7) && !(a.cin_off & 7) && a.in_bytes <= kOOB - (1u << 24) && !a.epi && a.up_C == 0;
    // whole tiles only: the kernel counts its stores (s_waitcnt vmcnt)
    if (common && a.stride == 1 && a.Ho == a.H && a.Wo == a.W && a.H % Ws128CfgT<false>::TH == 0 && a.W % Ws128CfgT<false>::TW == 0) return ws128_go<false>(a, s);
    if (common && a.stride == 2 && !(a.H & 1) && !(a.W & 1) && a.Ho * 2 == a.H && a.Wo * 2 == a.W && a.Ho % Ws128CfgT<true>::TH == 0 && a.Wo % Ws128CfgT<true>::TW == 0)
        return ws128_go<true>(a, s);
    y7t_set_error("conv: weights are in the 128-channel register-fragment order (korder 6) but the layer is neither a 3x3 / stride 1 / 128 -> 128 k convolution on a map of whole "
                  "4 x 16 tiles nor a 3x3 / stride 2 one on an even map whose output is whole 2 x 16 tiles, with an aligned fp16 output");
    return Y7T_E_ARG;
}
