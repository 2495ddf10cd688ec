// y7t_stem.hip -- the front of the YOLOv7-w6 forward as ONE kernel: uint8 BGR frame in HBM -> [letterbox (resize + pad 114)] -> BGR->RGB, /255
// -> ReOrg (space-to-depth) -> Conv 12->64 3x3 + folded BN + bias + activation -> fp16 NHWC.
//
// Restates /root/reference/tracker/tracker_dataloader.py:83-88,100-130 (letterbox, BGR->RGB, /255), models/common.py:48-53 (ReOrg),
// models/common.py:99-111 after utils/torch_utils.py:181-201 (the stem Conv, cfg/deploy/yolov7-w6.yaml:16-17).  SURVEY.md appendix E: the
// stem has K = 108 and is bound by its 52 MB / frame output; reading the uint8 frame directly removes the 13 MB / frame fp16 layout
// tensor (written by k_input_layout, read back by the conv), one launch, and the generic kernel's per-(pixel, tap) gathers of 32-byte rows.
//
// A workgroup walks 16 x 16-pixel tiles of the 640 x 640 (reorg) map.  Per tile: the 18 x 18 x 16-channel fp16 input patch is BUILT in LDS
// from the frame (a lane per reorg pixel: 2 x 2 source pixels x 3 bytes, or four bilinear taps each when the frame is resized); the nine
// taps are nine K = 16 MFMA steps (v_mfma_f32_32x32x16_f16) reading that patch at shifted addresses; the 64 x 144 weight matrix lives in
// registers for the whole kernel (A operand: 9 taps x 2 channel tiles x 16 B per lane).  48-byte pixel pitch in LDS: conflict-free
// ds_read_b128 for a 32-byte payload.  The patch of tile t+1 is fetched (global loads in flight) while tile t is multiplied and stored.
#include "y7t_common.h"
#include "y7t_conv_common.h"
#include <stdlib.h>

namespace {

constexpr int TS = 16;                 // tile side (reorg pixels)
constexpr int PS = TS + 2;             // patch side
constexpr int PIXB = 48;               // bytes per patch pixel in LDS (32 data + 16 pad)
constexpr int PATCH_BYTES = PS * PS * PIXB;     // 15552

struct StemArgs {
    const uint8_t* img;     // (B, H0, W0, 3) uint8 BGR
    int B, H0, W0;          // source frame
    int H, W;               // letterboxed image (network input); Hr = H / 2, Wr = W / 2 is the conv's map
    int new_h, new_w, top, left;      // letterbox geometry (new == source and top = left = 0: no resampling)
    const _Float16* w;      // [64][K_pad] packed stem weights, k = tap * 16 + ci
    int K_pad;
    const float* bias;      // [64]
    _Float16* out;          // NHWC
    int ldout, cout_off, act;
    int tiles_x, tiles_y, n_tiles;
};

// one source pixel of the letterboxed image as BGR floats in [0, 255] (tracker_dataloader.py:100-130)
template <bool RESIZE>
__device__ __forceinline__ void lb_pixel(const StemArgs& p, int b, int y, int x, float sy, float sx, float* bgr) {
    const int yy = y - p.top, xx = x - p.left;
    bgr[0] = bgr[1] = bgr[2] = 114.f;
    if ((unsigned)yy >= (unsigned)p.new_h || (unsigned)xx >= (unsigned)p.new_w) return;
    if (!RESIZE) {
        const uint8_t* s = p.img + (((size_t)b * p.H0 + yy) * p.W0 + xx) * 3;
        bgr[0] = s[0]; bgr[1] = s[1]; bgr[2] = s[2];
        return;
    }
    const float fy = ((float)yy + 0.5f) * sy - 0.5f, fx = ((float)xx + 0.5f) * sx - 0.5f;      // cv2.INTER_LINEAR: half-pixel centres, clamped taps
    int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const float wy = fy - (float)y0, wx = fx - (float)x0;
    const int y1 = min(max(y0 + 1, 0), p.H0 - 1), x1 = min(max(x0 + 1, 0), p.W0 - 1);
    y0 = min(max(y0, 0), p.H0 - 1); x0 = min(max(x0, 0), p.W0 - 1);
    const uint8_t* r0 = p.img + ((size_t)b * p.H0 + y0) * p.W0 * 3;
    const uint8_t* r1 = p.img + ((size_t)b * p.H0 + y1) * p.W0 * 3;
    for (int c = 0; c < 3; ++c) {
        const float t0 = (1.f - wx) * r0[x0 * 3 + c] + wx * r0[x1 * 3 + c];
        const float t1 = (1.f - wx) * r1[x0 * 3 + c] + wx * r1[x1 * 3 + c];
        bgr[c] = rintf((1.f - wy) * t0 + wy * t1);
    }
}

// LINES: the epilogue transposes a wave's 32 pixels x 64 channels through 4 KiB of its own LDS and stores full 128-byte lines (8 pixels per instruction) instead of four
// 32-byte pieces per pixel straight from the registers
template <bool RESIZE, bool LINES>
__global__ void __launch_bounds__(256, 2) k_stem_u8(const StemArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];      // two patches
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi32 = lane >> 5;
    const int Hr = p.H >> 1, Wr = p.W >> 1;
    const float sy = (float)p.H0 / (float)p.new_h, sx = (float)p.W0 / (float)p.new_w;

    // ---- weights: A operand, resident in registers.  Lane: output channel i*32 + l31, k half hi32 of tap t ----
    half8 wf[9][2];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 2; ++i) wf[t][i] = *(const half8*)(p.w + (size_t)(i * 32 + l31) * p.K_pad + t * 16 + hi32 * 8);
    float bv[2][4][4];      // bias of the 4 channels this lane owns per (channel tile, group)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) bv[i][g][e] = p.bias[i * 32 + 8 * g + 4 * hi32 + e];

    // A lane builds patch pixels tid and tid + 256 (< 324).  RESIZE = false (the frame already has the network's geometry up to padding,
    // W0 and `left` even): the 2 x 2 source pixels of a reorg pixel are two runs of 6 contiguous, 2-byte aligned bytes -> six UNCONDITIONAL
    // 16-bit loads at clamped coordinates (validity is a mask applied at conversion time), so all twelve loads of a lane are in flight
    // together under the MFMAs of the current tile.  (A first version loaded bytes inside the bounds branches: the compiler waited at
    // every branch, eight dependent round trips per tile, 850 us for the layer.)
    auto fetch_raw = [&](int tile, int q, unsigned short (&raw)[2][3], int& flags) {
        flags = 0;
        int tt = tile < p.n_tiles ? tile : 0;
        const int txi = tt % p.tiles_x; tt /= p.tiles_x;
        const int tyi = tt % p.tiles_y, b = tt / p.tiles_y;
        const int qq = q < PS * PS ? q : 0;
        const int ry = qq / PS, rx = qq - ry * PS;
        const int gy = tyi * TS - 1 + ry, gx = txi * TS - 1 + rx;
        if (q < PS * PS && tile < p.n_tiles && (unsigned)gy < (unsigned)Hr && (unsigned)gx < (unsigned)Wr) flags |= 1;
        const int xx = 2 * gx - p.left;                                  // even: the pixel pair is inside or outside together
        if ((unsigned)xx < (unsigned)p.new_w) flags |= 8;
        const int xc = min(max(xx, 0), p.W0 - 2);
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const int yy = 2 * gy + dy - p.top;
            if ((unsigned)yy < (unsigned)p.new_h) flags |= 2 << dy;
            const int yc = min(max(yy, 0), p.H0 - 1);
            const unsigned short* sp = (const unsigned short*)(p.img + (((size_t)b * p.H0 + yc) * p.W0 + xc) * 3);
            raw[dy][0] = sp[0]; raw[dy][1] = sp[1]; raw[dy][2] = sp[2];
        }
    };
    auto convert_raw = [&](const unsigned short (&raw)[2][3], int flags, half_t* v) {
#pragma unroll
        for (int c = 12; c < 16; ++c) v[c] = (half_t)0.f;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const bool ok = (flags & 8) && (flags & (2 << dy));
            const unsigned w0 = raw[dy][0] | ((unsigned)raw[dy][1] << 16), w1 = raw[dy][2];
            const unsigned char by[6] = {(unsigned char)w0, (unsigned char)(w0 >> 8), (unsigned char)(w0 >> 16), (unsigned char)(w0 >> 24),
                                         (unsigned char)w1, (unsigned char)(w1 >> 8)};
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const float u = ok ? (float)by[dx * 3 + (2 - ch)] : 114.f;                 // BGR -> RGB; pad colour outside the frame
                    // (half)(u * (1/255)) == (half)(u / 255) for every byte value (checked exhaustively; tests compare with the layout kernel's division)
                    v[(dy + 2 * dx) * 3 + ch] = (flags & 1) ? (half_t)(u * 0.00392156862745098f) : (half_t)0.f;   // conv padding outside the map
                }
        }
    };
    // RESIZE = true: any letterbox geometry, four bilinear taps per source pixel (the CLI's --device_preprocess on non-native frames)
    auto fetch = [&](int tile, int q, half_t* v) {
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = (half_t)0.f;
        if (q >= PS * PS || tile >= p.n_tiles) return;
        int tt = tile;
        const int txi = tt % p.tiles_x; tt /= p.tiles_x;
        const int tyi = tt % p.tiles_y, b = tt / p.tiles_y;
        const int ry = q / PS, rx = q - ry * PS;
        const int gy = tyi * TS - 1 + ry, gx = txi * TS - 1 + rx;      // reorg-map coordinates (conv padding 1 = outside -> zeros)
        if ((unsigned)gy >= (unsigned)Hr || (unsigned)gx >= (unsigned)Wr) return;
#pragma unroll
        for (int g = 0; g < 4; ++g) {                                   // cat order of ReOrg.forward: g = row parity + 2 * column parity
            float bgr[3];
            lb_pixel<true>(p, b, 2 * gy + (g & 1), 2 * gx + (g >> 1), sy, sx, bgr);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) v[g * 3 + ch] = (half_t)(bgr[2 - ch] / 255.0f);     // BGR -> RGB, /255
        }
    };
    auto store_patch = [&](char* patch, int q, const half_t* v) {
        if (q >= PS * PS) return;
        *(half8*)(patch + q * PIXB) = *(const half8*)v;
        *(half8*)(patch + q * PIXB + 16) = *(const half8*)(v + 8);
    };
    unsigned short ra[2][3], rb[2][3];
    int fa = 0, fb = 0;
    auto fetch_pair = [&](int tile, half_t* va_, half_t* vb_) {
        if (RESIZE) { fetch(tile, tid, va_); fetch(tile, tid + 256, vb_); }
        else { fetch_raw(tile, tid, ra, fa); fetch_raw(tile, tid + 256, rb, fb); }
    };
    auto finish_pair = [&](half_t* va_, half_t* vb_) {
        if (!RESIZE) { convert_raw(ra, fa, va_); convert_raw(rb, fb, vb_); }
    };

    const int t0 = blockIdx.x;
    half_t va[16], vb[16];
    fetch_pair(t0, va, vb);
    finish_pair(va, vb);
    store_patch(smem, tid, va); store_patch(smem, tid + 256, vb);
    __syncthreads();
    int cur = 0;
    for (int tile = t0; tile < p.n_tiles; tile += gridDim.x) {
        const int nxt = tile + gridDim.x;
        fetch_pair(nxt, va, vb);                                       // global loads in flight under the MFMAs
        const char* patch = smem + cur * PATCH_BYTES;
        // wave w owns pixel tiles 2w, 2w+1 (32 pixels = 2 image rows of 16) x both channel tiles
        floatx16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int kh = t / 3, kw = t - kh * 3;
            half8 xf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = (wave * 2 + j) * 2 + (l31 >> 4), x = l31 & 15;
                xf[j] = *(const half8*)(patch + ((row + kh) * PS + x + kw) * PIXB + hi32 * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[t][i], xf[j], acc[i][j], 0, 0, 0);
        }
        // ---- epilogue: bias + activation, 16-byte NHWC pieces straight from the registers ----
        int tt = tile;
        const int txi = tt % p.tiles_x; tt /= p.tiles_x;
        const int tyi = tt % p.tiles_y, b = tt / p.tiles_y;
        act_dispatch(p.act, [&](auto act_c) {
            constexpr int ACT = decltype(act_c)::value;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int gy = tyi * TS + (wave * 2 + j) * 2 + (l31 >> 4), gx = txi * TS + (l31 & 15);
                const bool okp = gy < Hr && gx < Wr;
                half_t* orow = p.out + ((size_t)(b * Hr + gy) * Wr + gx) * p.ldout + p.cout_off;
                char* scr = smem + 2 * PATCH_BYTES + wave * 4096;      // (LINES) this wave's 32 pixels x 128 bytes, slot = chunk ^ (pixel & 7)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        unsigned w[2][2];
#pragma unroll
                        for (int gg = 0; gg < 2; ++gg) {
                            const int g = gp * 2 + gg;
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = act_t<ACT>(acc[i][j][g * 4 + e] + bv[i][g][e]);
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
                        if (LINES) *(uint4v*)(scr + l31 * 128 + (((nn >> 3) ^ (l31 & 7)) << 4)) = pk;
                        else if (okp) *(uint4v*)(orow + nn) = pk;
                    }
                if (LINES) {      // wave-private: LDS instructions of one wave execute in order
                    typedef __attribute__((ext_vector_type(4))) unsigned uint4v;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int px = k * 8 + (lane >> 3), ch = lane & 7;
                        const uint4v v4 = *(const uint4v*)(scr + px * 128 + ((ch ^ (px & 7)) << 4));
                        const int py = tyi * TS + (wave * 2 + j) * 2 + (px >> 4), pxx = txi * TS + (px & 15);
                        if (py < Hr && pxx < Wr) *(uint4v*)(p.out + ((size_t)(b * Hr + py) * Wr + pxx) * p.ldout + p.cout_off + ch * 8) = v4;
                    }
                }
            }
        });
        // the next tile's patch goes into the other buffer; one barrier per tile (the buffer being overwritten was last read a tile ago)
        char* np_ = smem + (cur ^ 1) * PATCH_BYTES;
        finish_pair(va, vb);
        store_patch(np_, tid, va); store_patch(np_, tid + 256, vb);
        __syncthreads();
        cur ^= 1;
    }
#endif
}

}   // namespace

// op: the stem conv of the plan (3x3 / stride 1 / pad 1 on the ReOrg'd input, 64 output channels, weights packed with Cin padded to 16)
int y7t_stem_u8_launch(const void* frames_u8, int B, int H0, int W0, int H, int W, int new_h, int new_w, int top, int left, const _Float16* w, int K_pad,
                       const float* bias, _Float16* out, int ldout, int cout_off, int act, hipStream_t s) {
    if ((H & 31) || (W & 31) || ldout % 8 || cout_off % 8) { y7t_set_error("stem: image %dx%d / output slice not supported", H, W); return Y7T_E_ARG; }
    StemArgs a;
    a.img = (const uint8_t*)frames_u8; a.B = B; a.H0 = H0; a.W0 = W0; a.H = H; a.W = W; a.new_h = new_h; a.new_w = new_w; a.top = top; a.left = left;
    a.w = w; a.K_pad = K_pad; a.bias = bias; a.out = out; a.ldout = ldout; a.cout_off = cout_off; a.act = act;
    a.tiles_x = (W / 2 + TS - 1) / TS; a.tiles_y = (H / 2 + TS - 1) / TS; a.n_tiles = B * a.tiles_x * a.tiles_y;
    const bool resize = !(new_h == H0 && new_w == W0 && (left & 1) == 0 && (W0 & 1) == 0 && W0 >= 2);   // (odd geometry: the generic sampler)
    int grid = a.n_tiles < 2048 ? a.n_tiles : 2048;
    static Y7TOncePerDevice attr;      // (the attribute is per device: ADVICE r4)
    constexpr int LDSL = 2 * PATCH_BYTES + 4 * 4096;
    if (int e_ = y7t_once_per_device(attr, [&]() -> int {
        Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_stem_u8<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PATCH_BYTES));
        Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_stem_u8<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * PATCH_BYTES));
        Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_stem_u8<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDSL));
        return 0;
    })) return e_;
    static const int lines = y7t_exp_switch("Y7T_STEM_LINES", 1);      // full-line stores through LDS: 597 -> 546 us at 32 frames (profiles/r04_small_experiments.txt); 0 = straight from the registers
    if (resize) hipLaunchKernelGGL((k_stem_u8<true, false>), dim3(grid), dim3(256), 2 * PATCH_BYTES, s, a);
    else if (lines) hipLaunchKernelGGL((k_stem_u8<false, true>), dim3(grid), dim3(256), LDSL, s, a);
    else hipLaunchKernelGGL((k_stem_u8<false, false>), dim3(grid), dim3(256), 2 * PATCH_BYTES, s, a);
    Y7T_LAUNCH_CHECK();
    y7t_note_kernel("stem_u8<%s>", resize ? "letterbox-resize" : "direct");
    return 0;
}
