// y7t_stem_block.hip -- the first three convolutions of the YOLOv7-w6 forward as ONE kernel, nothing between them in HBM:
//
//   uint8 BGR frame -> BGR->RGB, /255 -> ReOrg -> Conv 12->64 3x3/1 (stem)  -> Conv 64->128 3x3/2 -> twin Conv 128->[64|64] 1x1 -> fp16 NHWC (320 x 320)
//   tracker_dataloader.py:83-88     common.py:48-53   yaml :17                   yaml :19              yaml :20-21 (the ELAN block's two 1x1 branches)
//   (each Conv = conv + folded BatchNorm + bias + SiLU: /root/reference/models/common.py:99-111 after utils/torch_utils.py:181-201)
//
// Why: the stem's 64-channel 640 x 640 output is the largest tensor of the network (52 MB per frame).  It has ONE consumer (the stride-2 conv), whose
// 128-channel 320 x 320 output (26 MB) again has one consumer (the twin 1x1).  As three launches they cost 0.60 + 0.86 + 0.40 ms per 32 frames, all three
// bound by writing / re-reading those tensors; fused, the frame (4.9 MB) comes in and the twin's output (26 MB) goes out -- 157 MB of the 1217 MB a frame
// moves never exist.
//
// A persistent workgroup (256 threads, ONE per CU: the filter banks live in registers -- 512 VGPR/AGPR per lane) walks 8 x 16 tiles of the 320 x 320 map:
//   1. IN    the 19 x 35 reorg-pixel input patch (16 fp16 channels, 12 real), built in LDS from the uint8 frame by the lanes (the next tile's bytes are
//            fetched into registers while this tile is multiplied);
//   2. STEM  17 x 33 stem-map positions = 18 MFMA pixel tiles over the four waves, nine K = 16 steps each off the IN patch, stem weights in registers;
//            bias + SiLU -> fp16 -> the MID patch in LDS, columns DE-INTERLEAVED by parity (row = [E0 .. E16 | O0 .. O15], 144-byte pixels) so that tap
//            kw = 0 / 1 / 2 of output column x is E[x] / O[x] / E[x+1] at unit stride; positions outside the 640 x 640 map are the stride-2 conv's ZERO padding;
//   3. CONV1 the stride-2 conv off the MID patch: wave w owns output channels 32w .. 32w+31 of all 128 pixels, its 32 x 576 filter slice resident in
//            registers as 36 MFMA A-fragments; every fragment address = lane base + immediate, conflict-free (36 x mod 64 over a service group);
//   4. O1    bias + SiLU -> fp16 -> a 128-pixel x 128-channel tile in LDS (over the dead MID patch; 272-byte pixels);
//   5. TWIN  the 1x1 conv off that tile (wave w: output channels 32w ..; 8 A-fragments in registers) -> bias + SiLU -> 16-byte NHWC stores.
// Four workgroup barriers per tile of ~260 MFMAs per wave.  Weight layouts: detector/weights.py::pack_stem_block.
#include "y7t_common.h"
#include "y7t_conv_common.h"
#include <stdlib.h>

namespace {

constexpr int TH = 8, TW = 16;                   // tile of the conv1 / twin output map
constexpr int MR = 2 * TH + 1, MC = 2 * TW + 1;  // MID patch: stem-map positions (17 x 33)
constexpr int NE = TW + 1;                       // even columns per MID row (17); odd: 16
constexpr int IR = MR + 2, IC = MC + 2;          // IN patch: reorg pixels (19 x 35)
constexpr int IPIX = 48;                         // bytes per IN pixel (32 data + 16 pad: conflict-free ds_read_b128, as k_stem_u8)
constexpr int IN_BYTES = IR * IC * IPIX;         // 31920
constexpr int MPIX = 144;                        // bytes per MID pixel (128 data + 16 pad)
constexpr int MRP = 4864;                        // MID row pitch: 33 x 144 = 4752 rounded up to a multiple of 128 (two rows apart = a multiple of 256)
constexpr int MID_BYTES = MR * MRP;              // 82688
constexpr int O1PIX = 272;                       // bytes per O1 pixel (256 data + 16 pad): 68 p mod 64 = 4 p, distinct over a service group
constexpr int NMQ = MR * MC;                     // 561 MID positions
constexpr int NMT = (NMQ + 31) / 32;             // 18 MFMA pixel tiles
constexpr int NIQ = IR * IC;                     // 665 IN positions
constexpr int NIS = (NIQ + 255) / 256;           // 3 per lane
constexpr int IN_OFF = 0, MID_OFF = 32000, BIAS_OFF = MID_OFF + MID_BYTES;     // LDS map (O1 aliases MID)
constexpr int LDS = BIAS_OFF + (64 + 128 + 128) * 4;
static_assert(128 * O1PIX <= MID_BYTES && IN_BYTES <= MID_OFF && MID_OFF % 256 == 0, "LDS map");

struct StemBlockArgs {
    const uint8_t* img;          // (B, H0, W0, 3) uint8 BGR, the network's geometry (H0 = 2 Hr, W0 = 2 Wr)
    int B, H0, W0;
    const _Float16* w0;          // stem weights [64][K0_pad], k = tap * 16 + ci (the plan's op 0, row-major)
    int K0_pad;
    const _Float16* wfrag;       // conv1 A-fragments [4 waves][36][64 lanes][8] then twin A-fragments [4][8][64][8]
    const float *b0, *b1, *b2;   // biases [64], [128], [128]
    _Float16* out;               // NHWC, the twin's output slice
    int ldout, cout_off;
    int act0, act1, act2;
    int Ho, Wo;                  // conv1 / twin map (Hr / 2, Wr / 2)
    int tiles_x, tiles_y, n_tiles;
};

__global__ void __launch_bounds__(256, 1) k_stem_block_u8(const StemBlockArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi32 = lane >> 5;
    const int Hr = p.H0 >> 1, Wr = p.W0 >> 1;
    typedef __attribute__((ext_vector_type(4))) unsigned uint4v;
    typedef __attribute__((ext_vector_type(4))) float float4v;
    typedef __attribute__((ext_vector_type(2))) _Float16 half2v;

    // ---- this workgroup's tiles: a contiguous range ----
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;      // workgroup b runs on XCD b % 8: contiguous ranges per XCD
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int per = (p.n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int pt_first = bid * per;
    const int nt = (p.n_tiles - pt_first) < per ? (p.n_tiles - pt_first) : per;
    if (nt <= 0) return;

    // ---- filter banks: registers for the whole launch ----
    half8 w0f[9][2];         // stem: output channel i*32 + l31, k half hi32 of tap t
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) w0f[t][i] = *(const half8*)(p.w0 + (size_t)(i * 32 + l31) * p.K0_pad + t * 16 + hi32 * 8);
    half8 w1f[36];           // conv1: channels 32 wave + l31; fragment tap * 4 + ks
    {
        const half8* wp = (const half8*)p.wfrag + (size_t)wave * 36 * 64 + lane;
#pragma unroll
        for (int f = 0; f < 36; ++f) w1f[f] = wp[f * 64];
    }
    half8 w2f[8];            // twin 1x1: channels 32 wave + l31; fragment ks
    {
        const half8* wp = (const half8*)p.wfrag + (size_t)4 * 36 * 64 + (size_t)wave * 8 * 64 + lane;
#pragma unroll
        for (int f = 0; f < 8; ++f) w2f[f] = wp[f * 64];
    }
    float* lb = (float*)(smem + BIAS_OFF);
    if (tid < 64) lb[tid] = p.b0[tid];
    if (tid < 128) { lb[64 + tid] = p.b1[tid]; lb[192 + tid] = p.b2[tid]; }

    // ---- uint8 frame -> IN patch.  A lane owns IN positions tid, tid + 256, tid + 512 (< 665): the 2 x 2 source pixels of a reorg pixel are two runs of 6
    // contiguous, 2-byte aligned bytes -> six unconditional 16-bit loads at clamped coordinates, validity applied at conversion (k_stem_u8's scheme) ----
    unsigned short raw[NIS][2][3];
    int flg[NIS];
    auto fetch_raw = [&](int tile) {
        int tt = tile < pt_first + nt ? tile : pt_first;
        const int txi = tt % p.tiles_x; tt /= p.tiles_x;
        const int tyi = tt % p.tiles_y, b = tt / p.tiles_y;
        const int gy0 = 2 * tyi * TH - 2, gx0 = 2 * txi * TW - 2;      // reorg-map coordinates of IN position (0, 0)
#pragma unroll
        for (int s = 0; s < NIS; ++s) {
            const int q = tid + 256 * s, qq = q < NIQ ? q : 0;
            const int ry = qq / IC, rx = qq - ry * IC;
            const int gy = gy0 + ry, gx = gx0 + rx;
            int f = (q < NIQ && (unsigned)gy < (unsigned)Hr && (unsigned)gx < (unsigned)Wr) ? 1 : 0;
            const int xc = min(max(2 * gx, 0), p.W0 - 2);
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const int yc = min(max(2 * gy + dy, 0), p.H0 - 1);
                const unsigned short* sp = (const unsigned short*)(p.img + (((size_t)b * p.H0 + yc) * p.W0 + xc) * 3);
                raw[s][dy][0] = sp[0]; raw[s][dy][1] = sp[1]; raw[s][dy][2] = sp[2];
            }
            flg[s] = f;
        }
    };
    auto store_in = [&]() {
#pragma unroll
        for (int s = 0; s < NIS; ++s) {
            const int q = tid + 256 * s;
            half_t v[16];
#pragma unroll
            for (int c = 12; c < 16; ++c) v[c] = (half_t)0.f;
#pragma unroll
            for (int dy = 0; dy < 2; ++dy) {
                const unsigned w0 = raw[s][dy][0] | ((unsigned)raw[s][dy][1] << 16), w1 = raw[s][dy][2];
                const unsigned char by[6] = {(unsigned char)w0, (unsigned char)(w0 >> 8), (unsigned char)(w0 >> 16), (unsigned char)(w0 >> 24),
                                             (unsigned char)w1, (unsigned char)(w1 >> 8)};
#pragma unroll
                for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch)      // BGR -> RGB; (half)(u * (1/255)) == (half)(u / 255) for every byte; zeros outside the map (the stem's padding)
                        v[(dy + 2 * dx) * 3 + ch] = flg[s] ? (half_t)((float)by[dx * 3 + (2 - ch)] * 0.00392156862745098f) : (half_t)0.f;
            }
            if (q < NIQ) {
                *(half8*)(smem + IN_OFF + q * IPIX) = *(const half8*)v;
                *(half8*)(smem + IN_OFF + q * IPIX + 16) = *(const half8*)(v + 8);
            }
        }
    };

    fetch_raw(pt_first);
    store_in();
    __syncthreads();

    half_t* outp = p.out;
    for (int t = 0; t < nt; ++t) {
        fetch_raw(pt_first + t + 1);                     // the next tile's bytes: in flight under this tile's MFMAs
        int tt = pt_first + t;
        const int txi = tt % p.tiles_x; tt /= p.tiles_x;
        const int tyi = tt % p.tiles_y, b = tt / p.tiles_y;
        const int oy0 = tyi * TH, ox0 = txi * TW;

        // ---- STEM: MID position q = 32 T + l31 -> (r, c); stem-map coordinates (2 oy0 - 1 + r, 2 ox0 - 1 + c) ----
#pragma unroll 1
        for (int T = wave; T < NMT; T += 4) {
            const int q = T * 32 + l31, qq = q < NMQ ? q : NMQ - 1;
            const int r = qq / MC, c = qq - r * MC;
            const char* ib = smem + IN_OFF + (r * IC + c) * IPIX + hi32 * 16;
            floatx16 a0[2];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) a0[i][e] = 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int kh = tap / 3, kw = tap - kh * 3;
                const half8 xf = *(const half8*)(ib + (kh * IC + kw) * IPIX);
#pragma unroll
                for (int i = 0; i < 2; ++i) a0[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0f[tap][i], xf, a0[i], 0, 0, 0);
            }
            const int gy = 2 * oy0 - 1 + r, gx = 2 * ox0 - 1 + c;
            const bool inmap = (unsigned)gy < (unsigned)Hr && (unsigned)gx < (unsigned)Wr;      // outside: the stride-2 conv's zero padding
            char* mrow = smem + MID_OFF + r * MRP + ((c & 1) ? NE + (c >> 1) : (c >> 1)) * MPIX;
            act_dispatch(p.act0, [&](auto act_c) {
            constexpr int ACT = decltype(act_c)::value;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    unsigned w[2][2];
#pragma unroll
                    for (int gg = 0; gg < 2; ++gg) {
                        const int g = gp * 2 + gg;
                        const float4v bv = *(const float4v*)(lb + i * 32 + 8 * g + 4 * hi32);
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = inmap ? act_t<ACT>(a0[i][g * 4 + e] + bv[e]) : 0.f;
                        half2v h0v = {(half_t)v[0], (half_t)v[1]}, h1v = {(half_t)v[2], (half_t)v[3]};
                        w[gg][0] = __builtin_bit_cast(unsigned, h0v);
                        w[gg][1] = __builtin_bit_cast(unsigned, h1v);
                    }
                    auto r0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                    const uint4v pk = {r0[0], r1[0], r0[1], r1[1]};
                    if (q < NMQ) *(uint4v*)(mrow + (i * 32 + 8 * (gp * 2 + hi32)) * 2) = pk;
                }
            });
        }
        __syncthreads();                                 // MID complete

        // ---- CONV1 (3x3 / stride 2): pixel tile j = output rows 2j, 2j+1 x 16 columns; tap (kh, kw) of output (oy, ox) = MID row 2 oy + kh, column 2 ox + kw ----
        floatx16 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        {
            const char* mb = smem + MID_OFF + (2 * (l31 >> 4)) * MRP + (l31 & 15) * MPIX + hi32 * 16;
            half8 xf[2][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) xf[0][j] = *(const half8*)(mb + j * 4 * MRP);
#pragma unroll
            for (int s = 0; s < 36; ++s) {
                const int cur = s & 1;
                if (s + 1 < 36) {
                    const int sn = s + 1, tap = sn >> 2, ks = sn & 3, kh = tap / 3, kw = tap - kh * 3;
                    const int po = (kw == 1 ? NE : kw == 2 ? 1 : 0) * MPIX;
#pragma unroll
                    for (int j = 0; j < 4; ++j) xf[cur ^ 1][j] = *(const half8*)(mb + (j * 4 + kh) * MRP + po + ks * 32);
                }
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1f[s], xf[cur][j], acc[j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
        }
        __syncthreads();                                 // everybody is done reading MID: the O1 tile takes its place

        // ---- O1: bias + activation -> fp16, pixel-major tile [128 pixels][128 channels] ----
        act_dispatch(p.act1, [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                unsigned w[2][2];
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int g = gp * 2 + gg;
                    const float4v bv = *(const float4v*)(lb + 64 + wave * 32 + 8 * g + 4 * hi32);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_t<ACT>(acc[j][g * 4 + e] + bv[e]);
                    half2v h0v = {(half_t)v[0], (half_t)v[1]}, h1v = {(half_t)v[2], (half_t)v[3]};
                    w[gg][0] = __builtin_bit_cast(unsigned, h0v);
                    w[gg][1] = __builtin_bit_cast(unsigned, h1v);
                }
                auto r0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                const uint4v pk = {r0[0], r1[0], r0[1], r1[1]};
                *(uint4v*)(smem + MID_OFF + (j * 32 + l31) * O1PIX + (wave * 32 + 8 * (gp * 2 + hi32)) * 2) = pk;
            }
        });
        __syncthreads();                                 // O1 complete

        // ---- TWIN 1x1 (128 -> 128 = [64 | 64]) off the O1 tile, then the output ----
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
        {
            const char* ob = smem + MID_OFF + l31 * O1PIX + hi32 * 16;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                half8 xf[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) xf[j] = *(const half8*)(ob + j * 32 * O1PIX + ks * 32);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2f[ks], xf[j], acc[j], 0, 0, 0);
            }
        }
        act_dispatch(p.act2, [&](auto act_c) {
        constexpr int ACT = decltype(act_c)::value;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pix = j * 32 + l31;
            const int oy = oy0 + (pix >> 4), ox = ox0 + (pix & 15);
            const bool okp = oy < p.Ho && ox < p.Wo;
            half_t* orow = outp + ((size_t)(b * p.Ho + oy) * p.Wo + ox) * p.ldout + p.cout_off + wave * 32;
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                unsigned w[2][2];
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int g = gp * 2 + gg;
                    const float4v bv = *(const float4v*)(lb + 192 + wave * 32 + 8 * g + 4 * hi32);
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_t<ACT>(acc[j][g * 4 + e] + bv[e]);
                    half2v h0v = {(half_t)v[0], (half_t)v[1]}, h1v = {(half_t)v[2], (half_t)v[3]};
                    w[gg][0] = __builtin_bit_cast(unsigned, h0v);
                    w[gg][1] = __builtin_bit_cast(unsigned, h1v);
                }
                auto r0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                const uint4v pk = {r0[0], r1[0], r0[1], r1[1]};
                if (okp) *(uint4v*)(orow + 8 * (gp * 2 + hi32)) = pk;
            }
        }
        });
        // ---- the next tile's IN patch (the current one was last read before three barriers) ----
        store_in();
        __syncthreads();
    }
#endif
}

}   // namespace

// ops 0..2 of a plan whose front is ReOrg + stem + 3x3/2 64 -> 128 + twin 1x1 128 -> 128 (checked by the caller, y7t_det_forward_stem_block_u8), on frames that
// already have the network's geometry
int y7t_stem_block_u8_launch(const void* frames_u8, int B, int H0, int W0, const _Float16* w0, int K0_pad, const _Float16* wfrag, const float* b0, const float* b1,
                             const float* b2, _Float16* out, int ldout, int cout_off, int act0, int act1, int act2, hipStream_t s) {
    if ((H0 & 3) || (W0 & 3) || ldout % 8 || cout_off % 8 || W0 < 4 || H0 < 4) { y7t_set_error("stem block: frame %dx%d / output slice not supported", H0, W0); return Y7T_E_ARG; }
    StemBlockArgs a;
    a.img = (const uint8_t*)frames_u8; a.B = B; a.H0 = H0; a.W0 = W0; a.w0 = w0; a.K0_pad = K0_pad; a.wfrag = wfrag; a.b0 = b0; a.b1 = b1; a.b2 = b2;
    a.out = out; a.ldout = ldout; a.cout_off = cout_off; a.act0 = act0; a.act1 = act1; a.act2 = act2;
    a.Ho = H0 / 4; a.Wo = W0 / 4;
    a.tiles_x = (a.Wo + TW - 1) / TW; a.tiles_y = (a.Ho + TH - 1) / TH; a.n_tiles = B * a.tiles_x * a.tiles_y;
    static bool attr = false;
    if (!attr) {
        Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_stem_block_u8, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr = true;
    }
    static int ncu = -1;      // one persistent workgroup per compute unit
    if (ncu < 0) {
        const char* e = getenv("Y7T_STEM_BLOCK_WGS");
        int dev = 0; hipDeviceProp_t prop;
        ncu = e ? atoi(e) : (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256);
        if (ncu <= 0) ncu = 256;
    }
    hipLaunchKernelGGL(k_stem_block_u8, dim3(a.n_tiles < ncu ? a.n_tiles : ncu), dim3(256), LDS, s, a);
    Y7T_LAUNCH_CHECK();
    y7t_note_kernel("stem_block_u8<8,16>");
    return 0;
}
