// y7t_post.hip -- the HBM-bound kernels around the convolutions:
//   * input layout kernels: (B,3,H,W) fp32 RGB in [0,1]  or  (B,H,W,3) uint8 BGR  ->  NHWC fp16, optionally fused with
//     ReOrg (space-to-depth; /root/reference/models/common.py:48-53) -- tracker_dataloader.py:83-88 semantics (/255, BGR->RGB)
//   * nearest x2 upsample (nn.Upsample(None, 2, 'nearest'), cfg/deploy/yolov7-w6.yaml:75,89,103) into a concat slice
//   * max-pool k x k (SPPCSPC's 5/9/13 as a 5-cascade, models/common.py:271-278; MP 2x2/s2, SP, common.py:30-45)
//   * Detect decode (models/yolo.py:39-57) fused with the candidate filter of non_max_suppression
//     (utils/general.py:607-665), rank sort by confidence, class-offset bitmask NMS with torchvision's greedy
//     semantics (general.py:676-682), top-300, scale_coords + clip + round (general.py:319-340, tracker/track.py:234-244)
#include "y7t_common.h"
#include <stdlib.h>
#include "y7t_det.h"
#include <string.h>

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;

// ------------------------------------------------------------------------------------------------ input layout
// out[b][y][x][c]: reorg: c = g*3 + ch, g = (row parity) + 2*(col parity)  (cat order of ReOrg.forward), padded to ldout
template <bool U8>
__global__ void __launch_bounds__(256) k_input_layout(const void* __restrict__ img, int B, int H, int W, int reorg, half_t* __restrict__ out,
                                                      int ldout) {
    const int Ho = reorg ? H / 2 : H, Wo = reorg ? W / 2 : W;
    const long long tot = (long long)B * Ho * Wo;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < tot; p += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(p / ((long long)Ho * Wo));
        const int rem = (int)(p - (long long)b * Ho * Wo);
        const int yo = rem / Wo, xo = rem - yo * Wo;
        half_t v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = (half_t)0.f;
        const int ng = reorg ? 4 : 1;
        for (int g = 0; g < ng; ++g) {
            const int y = reorg ? 2 * yo + (g & 1) : yo, x = reorg ? 2 * xo + (g >> 1) : xo;
            for (int ch = 0; ch < 3; ++ch) {
                float f;
                if (U8) f = (float)((const uint8_t*)img)[(((size_t)b * H + y) * W + x) * 3 + (2 - ch)] / 255.0f;  // BGR -> RGB, /255
                else f = ((const float*)img)[(((size_t)b * 3 + ch) * H + y) * W + x];
                v[g * 3 + ch] = (half_t)f;
            }
        }
        half_t* o = out + (size_t)p * ldout;
        for (int c = 0; c < ldout; c += 8) *(half8*)(o + c) = *(half8*)(v + c);
    }
}

// fast path of the above for uint8 frames with ReOrg and W % 8 == 0: a thread turns an 8 x 2 pixel block (2 x 24 contiguous bytes,
// read as 8-byte words) into 4 output pixels = one full 128-byte line (the generic kernel's byte loads ran at 1.1 TB/s)
__global__ void __launch_bounds__(256) k_input_layout_u8_reorg4(const uint8_t* __restrict__ img, int B, int H, int W, half_t* __restrict__ out) {
    const int Ho = H / 2, Wq = W / 8;                    // Wq: groups of 4 output pixels per row
    const long long tot = (long long)B * Ho * Wq;
    typedef __attribute__((ext_vector_type(2))) unsigned uint2v;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < tot; p += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(p / ((long long)Ho * Wq));
        const int rem = (int)(p - (long long)b * Ho * Wq);
        const int yo = rem / Wq, xq = rem - yo * Wq;
        union { uint2v w[3]; uint8_t u[24]; } r[2];
#pragma unroll
        for (int dy = 0; dy < 2; ++dy) {
            const uint2v* src = (const uint2v*)(img + (((size_t)b * H + 2 * yo + dy) * W + 8 * xq) * 3);
#pragma unroll
            for (int k = 0; k < 3; ++k) r[dy].w[k] = src[k];
        }
        half_t* o = out + ((size_t)(b * Ho + yo) * (W / 2) + 4 * xq) * 16;
#pragma unroll
        for (int px = 0; px < 4; ++px) {
            half_t v[16];
#pragma unroll
            for (int g = 0; g < 4; ++g)                  // g = row parity + 2 * column parity (cat order of ReOrg.forward)
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) v[g * 3 + ch] = (half_t)((float)r[g & 1].u[(2 * px + (g >> 1)) * 3 + (2 - ch)] / 255.0f);
#pragma unroll
            for (int c = 12; c < 16; ++c) v[c] = (half_t)0.f;
            *(half8*)(o + px * 16) = *(half8*)v;
            *(half8*)(o + px * 16 + 8) = *(half8*)(v + 8);
        }
    }
}

// letterbox (tracker_dataloader.py:100-130: resize INTER_LINEAR to `new_unpad`, pad 114 to the stride multiple) fused with the
// layout above: out pixel (y, x) of the H x W letterboxed image samples the H0 x W0 frame bilinearly (half-pixel centres,
// clamped taps, result rounded to uint8 like the resized image the reference feeds on) or is the pad colour.
__global__ void __launch_bounds__(256) k_letterbox_layout(const uint8_t* __restrict__ img, int B, int H0, int W0, int H, int W, int new_h, int new_w,
                                                          int top, int left, int reorg, half_t* __restrict__ out, int ldout) {
    const int Ho = reorg ? H / 2 : H, Wo = reorg ? W / 2 : W;
    const long long tot = (long long)B * Ho * Wo;
    const float sy = (float)H0 / (float)new_h, sx = (float)W0 / (float)new_w;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < tot; p += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(p / ((long long)Ho * Wo));
        const int rem = (int)(p - (long long)b * Ho * Wo);
        const int yo = rem / Wo, xo = rem - yo * Wo;
        half_t v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = (half_t)0.f;
        const int ng = reorg ? 4 : 1;
        for (int g = 0; g < ng; ++g) {
            const int y = reorg ? 2 * yo + (g & 1) : yo, x = reorg ? 2 * xo + (g >> 1) : xo;
            const int yy = y - top, xx = x - left;
            float px[3] = {114.f, 114.f, 114.f};   // BGR pad colour
            if ((unsigned)yy < (unsigned)new_h && (unsigned)xx < (unsigned)new_w) {
                if (new_h == H0 && new_w == W0) {
                    const uint8_t* s = img + (((size_t)b * H0 + yy) * W0 + xx) * 3;
                    px[0] = s[0]; px[1] = s[1]; px[2] = s[2];
                } else {
                    float fy = ((float)yy + 0.5f) * sy - 0.5f, fx = ((float)xx + 0.5f) * sx - 0.5f;
                    int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
                    const float wy = fy - (float)y0, wx = fx - (float)x0;
                    const int y1 = min(max(y0 + 1, 0), H0 - 1), x1 = min(max(x0 + 1, 0), W0 - 1);
                    y0 = min(max(y0, 0), H0 - 1); x0 = min(max(x0, 0), W0 - 1);
                    const uint8_t* r0 = img + ((size_t)b * H0 + y0) * W0 * 3;
                    const uint8_t* r1 = img + ((size_t)b * H0 + y1) * W0 * 3;
                    for (int c = 0; c < 3; ++c) {
                        const float t0 = (1.f - wx) * r0[x0 * 3 + c] + wx * r0[x1 * 3 + c];
                        const float t1 = (1.f - wx) * r1[x0 * 3 + c] + wx * r1[x1 * 3 + c];
                        px[c] = rintf((1.f - wy) * t0 + wy * t1);
                    }
                }
            }
            for (int ch = 0; ch < 3; ++ch) v[g * 3 + ch] = (half_t)(px[2 - ch] / 255.0f);   // BGR -> RGB, /255
        }
        half_t* o = out + (size_t)p * ldout;
        for (int c = 0; c < ldout; c += 8) *(half8*)(o + c) = *(half8*)(v + c);
    }
}

// ------------------------------------------------------------------------------------------------ upsample / pool
// out[b][y][x][coff + c] = in[b][y/2][x/2][cin_off + c]; 8 channels (16 B) per thread
__global__ void __launch_bounds__(256) k_upsample2x(const half_t* __restrict__ in, int ldin, int cin_off, int B, int H, int W, int C,
                                                    half_t* __restrict__ out, int ldout, int cout_off) {
    const int C8 = C / 8, Ho = 2 * H, Wo = 2 * W;
    const long long tot = (long long)B * Ho * Wo * C8;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < tot; t += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(t % C8);
        const long long p = t / C8;
        const int xo = (int)(p % Wo), yo = (int)((p / Wo) % Ho), b = (int)(p / ((long long)Wo * Ho));
        const half8 v = *(const half8*)(in + (((size_t)b * H + (yo >> 1)) * W + (xo >> 1)) * ldin + cin_off + c8 * 8);
        *(half8*)(out + (size_t)p * ldout + cout_off + c8 * 8) = v;
    }
}

// max-pool k x k, stride s, padding pd (-inf padding like nn.MaxPool2d)
__global__ void __launch_bounds__(256) k_maxpool(const half_t* __restrict__ in, int ldin, int cin_off, int B, int H, int W, int C, int k, int s,
                                                 int pd, half_t* __restrict__ out, int ldout, int cout_off, int Ho, int Wo) {
    const int C8 = C / 8;
    const long long tot = (long long)B * Ho * Wo * C8;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < tot; t += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(t % C8);
        const long long p = t / C8;
        const int xo = (int)(p % Wo), yo = (int)((p / Wo) % Ho), b = (int)(p / ((long long)Wo * Ho));
        half8 m;
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (half_t)(-65504.f);
        bool any = false;
        for (int dy = 0; dy < k; ++dy) {
            const int y = yo * s - pd + dy;
            if ((unsigned)y >= (unsigned)H) continue;
            for (int dx = 0; dx < k; ++dx) {
                const int x = xo * s - pd + dx;
                if ((unsigned)x >= (unsigned)W) continue;
                const half8 v = *(const half8*)(in + (((size_t)b * H + y) * W + x) * ldin + cin_off + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = (any && m[e] > v[e]) ? m[e] : v[e];
                any = true;
            }
        }
        *(half8*)(out + (size_t)p * ldout + cout_off + c8 * 8) = m;
    }
}

// The SPPCSPC pools (models/common.py:262-280: MaxPool2d(5 / 9 / 13, 1, k // 2) of one 20 x 20 tensor, run as the cascade 5 o 5 o 5): k x k / stride 1 on a map small
// enough for a workgroup to hold one image's 16-channel slab in LDS -- load it once (one full 32-byte piece per pixel), row maxima, column maxima (separable: 2 k instead
// of k^2 comparisons, no bounds branches around global loads).  The generic kernel above needs 31 us for 6.5 M elements (25 dependent, predicated 16-byte loads per
// thread); this one is bound by its launch.  Same -inf padding semantics: out-of-range taps are skipped.
template <int K>
__global__ void __launch_bounds__(256) k_maxpool_s1_lds(const half_t* __restrict__ in, int ldin, int cin_off, int H, int W, int C, half_t* __restrict__ out, int ldout,
                                                        int cout_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8* A = (half8*)smem;
    const int HW = H * W, n = HW * 2, C16 = C / 16;
    half8* R = A + n;
    const int b = blockIdx.x / C16, cg = blockIdx.x - b * C16;
    const size_t ibase = (size_t)b * HW * ldin + cin_off + cg * 16, obase = (size_t)b * HW * ldout + cout_off + cg * 16;
    for (int i = threadIdx.x; i < n; i += 256) A[i] = *(const half8*)(in + ibase + (size_t)(i >> 1) * ldin + (i & 1) * 8);
    __syncthreads();
    constexpr int P = K / 2;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int px = i >> 1, y = px / W, x = px - y * W;
        const int x0 = x - P < 0 ? 0 : x - P, x1 = x + P >= W ? W - 1 : x + P;
        half8 m = A[(y * W + x0) * 2 + (i & 1)];
        for (int xx = x0 + 1; xx <= x1; ++xx) {
            const half8 v = A[(y * W + xx) * 2 + (i & 1)];
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = m[e] > v[e] ? m[e] : v[e];
        }
        R[i] = m;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) {
        const int px = i >> 1, y = px / W, x = px - y * W;
        const int y0 = y - P < 0 ? 0 : y - P, y1 = y + P >= H ? H - 1 : y + P;
        half8 m = R[(y0 * W + x) * 2 + (i & 1)];
        for (int yy = y0 + 1; yy <= y1; ++yy) {
            const half8 v = R[(yy * W + x) * 2 + (i & 1)];
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = m[e] > v[e] ? m[e] : v[e];
        }
        *(half8*)(out + obase + (size_t)px * ldout + (i & 1) * 8) = m;
    }
}

// ... and the whole SPPCSPC cascade 5 o 5 o 5 (= the 5 / 9 / 13 pools of models/common.py:262-280) in ONE launch: the slab stays in LDS between the three pools, each
// result goes to its own channel slice (out + k * C channels further for pool k of the cascade: the three results are consecutive slices of the concat buffer).
__global__ void __launch_bounds__(256) k_spp3_lds(const half_t* __restrict__ in, int ldin, int cin_off, int H, int W, int C, half_t* __restrict__ out, int ldout, int cout_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half8* A = (half8*)smem;
    const int HW = H * W, n = HW * 2, C16 = C / 16;
    half8* R = A + n;
    const int b = blockIdx.x / C16, cg = blockIdx.x - b * C16;
    const size_t ibase = (size_t)b * HW * ldin + cin_off + cg * 16, obase = (size_t)b * HW * ldout + cout_off + cg * 16;
    for (int i = threadIdx.x; i < n; i += 256) A[i] = *(const half8*)(in + ibase + (size_t)(i >> 1) * ldin + (i & 1) * 8);
    __syncthreads();
    for (int pool = 0; pool < 3; ++pool) {
        for (int i = threadIdx.x; i < n; i += 256) {
            const int px = i >> 1, y = px / W, x = px - y * W;
            const int x0 = x - 2 < 0 ? 0 : x - 2, x1 = x + 2 >= W ? W - 1 : x + 2;
            half8 m = A[(y * W + x0) * 2 + (i & 1)];
            for (int xx = x0 + 1; xx <= x1; ++xx) {
                const half8 v = A[(y * W + xx) * 2 + (i & 1)];
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = m[e] > v[e] ? m[e] : v[e];
            }
            R[i] = m;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) {
            const int px = i >> 1, y = px / W, x = px - y * W;
            const int y0 = y - 2 < 0 ? 0 : y - 2, y1 = y + 2 >= H ? H - 1 : y + 2;
            half8 m = R[(y0 * W + x) * 2 + (i & 1)];
            for (int yy = y0 + 1; yy <= y1; ++yy) {
                const half8 v = R[(yy * W + x) * 2 + (i & 1)];
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = m[e] > v[e] ? m[e] : v[e];
            }
            A[i] = m;      // (every thread rewrites exactly the entries it read nothing else of: its own i; the row pass of the next pool starts behind the barrier)
            *(half8*)(out + obase + (size_t)pool * C + (size_t)px * ldout + (i & 1) * 8) = m;
        }
        __syncthreads();
    }
}

// the three cascaded 5 x 5 / 1 pools of an SPP block as one launch; 1 if it cannot run this shape (the caller launches them one by one)
int y7t_spp3_try(const half_t* in, int ldin, int cin_off, int B, int H, int W, int C, half_t* out, int ldout, int cout_off, hipStream_t s) {
    static const int on = y7t_exp_switch("Y7T_SPP3", 1);
    if (!on || C % 16 || ldin % 8 || cin_off % 8 || ldout % 8 || cout_off % 8 || H * W > 1024) return 1;
    hipLaunchKernelGGL(k_spp3_lds, dim3(B * (C / 16)), dim3(256), (size_t)H * W * 64, s, in, ldin, cin_off, H, W, C, out, ldout, cout_off);
    Y7T_LAUNCH_CHECK();
    y7t_note_kernel("spp3<5,5,5> lds");
    return 0;
}

int y7t_upsample_launch(const half_t* in, int ldin, int cin_off, int B, int H, int W, int C, half_t* out, int ldout, int cout_off, hipStream_t s) {
    if (C % 8 || ldin % 8 || cin_off % 8 || ldout % 8 || cout_off % 8) { y7t_set_error("upsample: channel alignment"); return Y7T_E_ARG; }
    const long long tot = (long long)B * 4 * H * W * (C / 8);
    int blocks = (int)((tot + 255) / 256); if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_upsample2x, dim3(blocks), dim3(256), 0, s, in, ldin, cin_off, B, H, W, C, out, ldout, cout_off);
    Y7T_LAUNCH_CHECK();
    y7t_note_kernel("upsample2x");
    return 0;
}

int y7t_maxpool_launch(const half_t* in, int ldin, int cin_off, int B, int H, int W, int C, int k, int st, int pd, half_t* out, int ldout,
                       int cout_off, hipStream_t s) {
    if (C % 8 || ldin % 8 || cin_off % 8 || ldout % 8 || cout_off % 8) { y7t_set_error("maxpool: channel alignment"); return Y7T_E_ARG; }
    const int Ho = (H + 2 * pd - k) / st + 1, Wo = (W + 2 * pd - k) / st + 1;
    static const int lds_pool = y7t_exp_switch("Y7T_POOL_LDS", 1);
    if (lds_pool && (k == 5 || k == 9 || k == 13) && st == 1 && pd == k / 2 && C % 16 == 0 && H * W <= 1024) {
        const dim3 grid(B * (C / 16)), blk(256);
        const size_t lds = (size_t)H * W * 64;
        if (k == 5) hipLaunchKernelGGL(k_maxpool_s1_lds<5>, grid, blk, lds, s, in, ldin, cin_off, H, W, C, out, ldout, cout_off);
        else if (k == 9) hipLaunchKernelGGL(k_maxpool_s1_lds<9>, grid, blk, lds, s, in, ldin, cin_off, H, W, C, out, ldout, cout_off);
        else hipLaunchKernelGGL(k_maxpool_s1_lds<13>, grid, blk, lds, s, in, ldin, cin_off, H, W, C, out, ldout, cout_off);
        Y7T_LAUNCH_CHECK();
        y7t_note_kernel("maxpool<%d,%d> lds", k, st);
        return 0;
    }
    const long long tot = (long long)B * Ho * Wo * (C / 8);
    int blocks = (int)((tot + 255) / 256); if (blocks > 8192) blocks = 8192; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_maxpool, dim3(blocks), dim3(256), 0, s, in, ldin, cin_off, B, H, W, C, k, st, pd, out, ldout, cout_off, Ho, Wo);
    Y7T_LAUNCH_CHECK();
    y7t_note_kernel("maxpool<%d,%d>", k, st);
    return 0;
}

extern "C" int y7t_input_layout(const void* img, int is_u8, int B, int H, int W, int reorg, void* out_f16, int ldout, y7t_stream stream) {
    Y7T_ARG_CHECK(img && out_f16 && B > 0 && H > 0 && W > 0);
    Y7T_ARG_CHECK(ldout == 8 || ldout == 16);
    Y7T_ARG_CHECK(reorg ? (ldout == 16 && H % 2 == 0 && W % 2 == 0) : 1);
    const long long tot = (long long)B * (reorg ? H / 2 : H) * (reorg ? W / 2 : W);
    int blocks = (int)((tot + 255) / 256); if (blocks > 8192) blocks = 8192;
    if (is_u8 && reorg && W % 8 == 0 && ldout == 16) {
        const long long tq = tot / 4;
        int bq = (int)((tq + 255) / 256); if (bq > 16384) bq = 16384;
        hipLaunchKernelGGL(k_input_layout_u8_reorg4, dim3(bq), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)img, B, H, W, (half_t*)out_f16);
    } else if (is_u8) hipLaunchKernelGGL(k_input_layout<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, B, H, W, reorg, (half_t*)out_f16, ldout);
    else hipLaunchKernelGGL(k_input_layout<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, img, B, H, W, reorg, (half_t*)out_f16, ldout);
    Y7T_LAUNCH_CHECK();
    return 0;
}

extern "C" int y7t_letterbox_layout_u8(const void* img, int B, int H0, int W0, int H, int W, int new_h, int new_w, int top, int left, int reorg,
                                       void* out_f16, int ldout, y7t_stream stream) {
    Y7T_ARG_CHECK(img && out_f16 && B > 0 && H0 > 0 && W0 > 0 && H > 0 && W > 0 && new_h > 0 && new_w > 0 && top >= 0 && left >= 0);
    Y7T_ARG_CHECK(top + new_h <= H && left + new_w <= W);
    Y7T_ARG_CHECK(ldout == 8 || ldout == 16);
    Y7T_ARG_CHECK(reorg ? (ldout == 16 && H % 2 == 0 && W % 2 == 0) : 1);
    const long long tot = (long long)B * (reorg ? H / 2 : H) * (reorg ? W / 2 : W);
    int blocks = (int)((tot + 255) / 256); if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_letterbox_layout, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)img, B, H0, W0, H, W, new_h, new_w, top, left,
                       reorg, (half_t*)out_f16, ldout);
    Y7T_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------ decode + filter
struct DecodeLevel {
    const float* p;   // head conv output, NHWC fp32: [B][ny][nx][na*no]
    int ny, nx, row0; // row0: index of this level's first row in the reference's (B, A, no) ordering
    float stride;
    float aw[3], ah[3];
};
struct DecodeArgs {
    DecodeLevel lv[4];
    int nl, na, no, B;
    float conf_thres;
    int cap;              // candidate capacity per image
    float* cbox;          // [B][cap][4] xyxy
    float* cscore;        // [B][cap]
    float* ccls;          // [B][cap]
    int* cidx;            // [B][cap] row index (tie-break / parity checks)
    int* count;           // [B] candidates found (may exceed cap -> overflow)
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void __launch_bounds__(256) k_decode_filter(const DecodeArgs a, int level) {
    const DecodeLevel L = a.lv[level];
    const int per_img = a.na * L.ny * L.nx;
    const long long tot = (long long)a.B * per_img;
    const int nc = a.no - 5;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < tot; t += (long long)gridDim.x * blockDim.x) {
        const int b = (int)(t / per_img), r = (int)(t - (long long)b * per_img);
        // thread order: x fastest, then anchor, then y -> neighbouring threads read neighbouring NHWC rows
        const int x = r % L.nx, an = (r / L.nx) % a.na, y = r / (L.nx * a.na);
        const float* p = L.p + ((((size_t)b * L.ny + y) * L.nx + x) * a.na + an) * a.no;
        const float obj = sigmoidf_(p[4]);
        if (!(obj > a.conf_thres)) continue;                  // xc = prediction[..., 4] > conf_thres
        float best = -1.f; int bj = 0;
        for (int c = 0; c < nc; ++c) {                        // x[:, 5:] *= x[:, 4:5]; conf, j = x[:, 5:].max(1)
            const float s = sigmoidf_(p[5 + c]) * obj;
            if (s > best) { best = s; bj = c; }
        }
        if (!(best > a.conf_thres)) continue;
        const float sx = sigmoidf_(p[0]), sy = sigmoidf_(p[1]), sw = sigmoidf_(p[2]), sh = sigmoidf_(p[3]);
        const float cx = (sx * 2.f - 0.5f + (float)x) * L.stride, cy = (sy * 2.f - 0.5f + (float)y) * L.stride;
        const float w = (sw * 2.f) * (sw * 2.f) * L.aw[an], h = (sh * 2.f) * (sh * 2.f) * L.ah[an];
        const int slot = atomicAdd(a.count + b, 1);
        if (slot >= a.cap) continue;
        float* bo = a.cbox + ((size_t)b * a.cap + slot) * 4;
        bo[0] = cx - w / 2; bo[1] = cy - h / 2; bo[2] = cx + w / 2; bo[3] = cy + h / 2;   // xywh2xyxy
        a.cscore[(size_t)b * a.cap + slot] = best;
        a.ccls[(size_t)b * a.cap + slot] = (float)bj;
        a.cidx[(size_t)b * a.cap + slot] = L.row0 + (an * L.ny + y) * L.nx + x;
    }
}

// ------------------------------------------------------------------------------------------------ rank sort
// rank by (score desc, row index asc); scatter class-offset boxes (x[:, :4] + cls*4096) into sorted order
__global__ void __launch_bounds__(256) k_rank_sort(const float* __restrict__ cbox, const float* __restrict__ cscore, const float* __restrict__ ccls,
                                                   const int* __restrict__ cidx, const int* __restrict__ count, int cap, int max_nms,
                                                   float* __restrict__ sbox, int* __restrict__ sorder, int* __restrict__ nsorted, int mcap) {
    // round 6: (score desc, row index asc) as ONE unsigned 64-bit key -- the score's bit pattern (scores are positive floats: their bit patterns order like their
    // values) over the complement of the row index -- so a comparison is one LDS read and one 64-bit compare instead of two reads and three compares (k_rank_sort at one
    // frame: 52 us of the batch-1 frame).  Keys are unique (row indices are), so `>` alone gives the rank; an empty slot has key 0 and outranks nothing.
    __shared__ unsigned long long sk[256];
    const int b = blockIdx.y;
    int n = count[b]; if (n > cap) n = cap;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x * 256 >= n) return;
    const float* sc = cscore + (size_t)b * cap;
    const int* ix = cidx + (size_t)b * cap;
    auto key_of = [&](int j) -> unsigned long long {
        const float v = sc[j];
        if (!(v > 0.f)) return 0ull;      // (not a candidate score: conf_thres > 0 keeps every real one positive; NaN sorts last)
        return ((unsigned long long)__builtin_bit_cast(unsigned, v) << 32) | (unsigned long long)(0xFFFFFFFFu - (unsigned)ix[j]);
    };
    const unsigned long long my_k = i < n ? key_of(i) : 0ull;
    int rank = 0;
    unsigned long long nxt = (int)threadIdx.x < n ? key_of(threadIdx.x) : 0ull;      // the next chunk's key: its two global loads are in flight while the current chunk is compared
    for (int base = 0; base < n; base += 256) {
        __syncthreads();
        sk[threadIdx.x] = nxt;
        __syncthreads();
        const int jn = base + 256 + threadIdx.x;
        nxt = jn < n ? key_of(jn) : 0ull;
        const int lim = (n - base) < 256 ? (n - base) : 256;
        if (lim == 256) {
#pragma unroll 8
            for (int t = 0; t < 256; ++t) rank += sk[t] > my_k;
        } else {
            for (int t = 0; t < lim; ++t) rank += sk[t] > my_k;
        }
    }
    if (i < n && rank < max_nms) {
        const float off = ccls[(size_t)b * cap + i] * 4096.f;
        const float* bo = cbox + ((size_t)b * cap + i) * 4;
        float* so = sbox + ((size_t)b * mcap + rank) * 4;
        so[0] = bo[0] + off; so[1] = bo[1] + off; so[2] = bo[2] + off; so[3] = bo[3] + off;
        sorder[(size_t)b * mcap + rank] = i;
    }
    if (i == 0) nsorted[b] = n < max_nms ? n : max_nms;
}

// ------------------------------------------------------------------------------------------------ greedy NMS, kept-list form
// torchvision.ops.nms semantics (general.py:676-682): walk the candidates in score order; keep one unless an EARLIER KEPT box of its class
// (class-offset boxes) overlaps it with IoU > thr; stop after max_det keeps.  One workgroup per image walks the sorted list in chunks of 256:
//   (a) every lane tests its candidate against the kept list so far (<= max_det boxes in LDS, broadcast reads);
//   (b) the four waves take turns: a wave first tests against what the earlier waves of this chunk kept, then resolves its own 64 candidates
//       with ballots (the first alive lane is kept, the later lanes test against it, repeat).
// Work = (candidates walked) x (boxes kept), and the walk ends at max_det keeps -- typically a few hundred candidates of a few thousand --
// where the previous form computed the full upper-triangular IoU bit-matrix (n^2 / 2 pairs, n * n / 8 bytes) before a serial scan.
// The IoU arithmetic is the same expression as before (raw coordinates, no +1), so keep lists stay bit-identical.
__device__ __forceinline__ bool nms_overlaps(const float* a, float aa, float x1, float y1, float x2, float y2, float ab, float thr) {
    const float xx1 = fmaxf(a[0], x1), yy1 = fmaxf(a[1], y1), xx2 = fminf(a[2], x2), yy2 = fminf(a[3], y2);
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    const float inter = w * h;
    // round 6: boxes that do not intersect (most pairs: other classes sit 4096 px apart) skip the IEEE division -- exact for thr >= 0 (the launcher checks): 0 / u > thr is
    // false for every u (NaN for u == 0 compares false too), and a NaN intersection compares false on both paths.  k_nms_keep is ONE workgroup per image: at batch 1
    // its ~170 us were 7 % of the frame the reference's Timer brackets.
    if (!(inter > 0.f)) return false;
    return inter / (aa + ab - inter) > thr;
}

__global__ void __launch_bounds__(256) k_nms_keep(const float* __restrict__ sbox, const int* __restrict__ nsorted, const int* __restrict__ sorder,
                                                  const float* __restrict__ cbox, const float* __restrict__ cscore, const float* __restrict__ ccls, int cap,
                                                  int mcap, float thr, int max_det, const float* __restrict__ lb /*[B][5] gain,padw,padh,H0,W0*/,
                                                  float* __restrict__ dets /*[B][max_det][6]*/, int* __restrict__ ndets, int* __restrict__ keep_idx /*[B][max_det]*/) {
    extern __shared__ float ksm[];                 // kept boxes [max_det][4] | areas [max_det] | their positions in the sorted list [max_det]
    __shared__ int s_nkeep;
    float* kbox = ksm;
    float* karea = ksm + 4 * max_det;
    int* kidx = (int*)(ksm + 5 * max_det);         // (round 6: the keep loop's global store per kept box moved out of the serial chain)
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = nsorted[b];
    const float* sb = sbox + (size_t)b * mcap * 4;
    __shared__ int s_turn;                         // the 64-candidate chunk whose wave may append to the kept list (kTurnDone: the list is full)
    constexpr int kTurnDone = 0x7fffffff;
    if (tid == 0) { s_nkeep = 0; s_turn = 0; }
    __syncthreads();
    // Round 6: the chunks are 64 candidates (one wave each, chunk c on wave c % 4) and the waves no longer take turns behind barriers.  The order of the greedy walk is
    // kept by a TURN counter: a wave appends only when s_turn is its chunk; while it waits it already tests its candidates against the kept boxes as the waves in front
    // of it publish them (s_nkeep, released after the boxes), so that when its turn comes only the last few kept boxes are left to test and the serial chain of the
    // launch is the keep loop alone -- 300 x (ballot, kept box through the scalar registers, one overlap test) -- instead of 32 barrier-separated turns each with its
    // share of pair tests (k_nms_keep at one frame: 131 us of a 2.5 ms frame).  Same pairs, same expression, same order of keeps: the keep list is bit-identical.
    // (A wave with the turn never waits for anybody, and the turn only moves forward: no cycle.)
    for (int c = wave; c * 64 < n; c += 4) {
        const int i = c * 64 + lane;
        const bool valid = i < n;
        float bx[4] = {0.f, 0.f, 0.f, 0.f};
        if (valid) { const float4 v = *(const float4*)(sb + 4 * (size_t)i); bx[0] = v.x; bx[1] = v.y; bx[2] = v.z; bx[3] = v.w; }
        const float area = (bx[2] - bx[0]) * (bx[3] - bx[1]);
        bool alive = valid;
        // a candidate against the kept boxes [k0, k1): the verdicts of the pairs are independent (the candidate dies if ANY earlier kept box overlaps it), so they are
        // taken four kept boxes at a time WITHOUT a branch between them (one LDS round trip per group, not per box), and the IEEE division only runs for a group in which
        // this lane intersects something
        auto against_kept = [&](int k0, int k1) {
            bool dead = false;
            int k = k0;
            for (; k + 4 <= k1; k += 4) {
                float in_[4], un_[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float* a = kbox + 4 * (k + q);
                    const float xx1 = fmaxf(a[0], bx[0]), yy1 = fmaxf(a[1], bx[1]), xx2 = fminf(a[2], bx[2]), yy2 = fminf(a[3], bx[3]);
                    const float w_ = fmaxf(0.f, xx2 - xx1), h_ = fmaxf(0.f, yy2 - yy1);
                    in_[q] = w_ * h_;
                    un_[q] = karea[k + q] + area - in_[q];
                }
                if ((in_[0] > 0.f) | (in_[1] > 0.f) | (in_[2] > 0.f) | (in_[3] > 0.f)) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) dead |= (in_[q] > 0.f) && (in_[q] / un_[q] > thr);      // (the same expression as nms_overlaps: inter / (aa + ab - inter) > thr)
                }
            }
            for (; k < k1; ++k) dead |= nms_overlaps(kbox + 4 * k, karea[k], bx[0], bx[1], bx[2], bx[3], area, thr);
            return dead;
        };
        // the wave's own pairs, ahead of its turn: bit j of `ov` = THIS lane's box, were it kept, suppresses candidate j of the wave (j > lane) -- the very test the
        // keep loop used to make per kept box (kept box first, candidate second: aa + ab - inter in that order).  Built once, over the lanes that are alive at that
        // moment (a lane that dies later only leaves bits nobody reads), while the wave would otherwise wait; the keep loop is then scalar bit arithmetic.
        unsigned long long ov = 0ull;
        bool have_ov = false;
        auto build_ov = [&]() {
            unsigned long long it = __ballot(alive);
            while (it) {
                const int j = __ffsll((long long)it) - 1;
                it &= it - 1ull;
                auto rl = [&](float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), j)); };
                const float jx1 = rl(bx[0]), jy1 = rl(bx[1]), jx2 = rl(bx[2]), jy2 = rl(bx[3]), ja = rl(area);
                if (alive && lane < j && nms_overlaps(bx, area, jx1, jy1, jx2, jy2, ja, thr)) ov |= 1ull << j;
            }
            have_ov = true;
        };
        int tested = 0, nk = 0, turn = 0;
        for (;;) {                                 // (uniform over the wave: turn / nk are workgroup-scope atomic loads of one word each)
            turn = __hip_atomic_load(&s_turn, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);      // the turn FIRST: whatever it shows, the kept boxes of every chunk in front of it are published
            nk = __hip_atomic_load(&s_nkeep, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (nk > tested) {
                if (alive && against_kept(tested, nk)) alive = false;
                tested = nk;
            } else if (turn != c && turn != kTurnDone) {
                if (!have_ov) build_ov();          // nothing new to test against: the wave's own pairs now
                else __builtin_amdgcn_s_sleep(2);
            }
            if (turn == c || turn == kTurnDone) break;
        }
        if (turn == kTurnDone) break;              // max_det boxes are kept: nothing behind them is looked at (general.py:684-685)
        {   // my turn: every chunk in front of this one is final and tested against.  The greedy walk over the wave's 64 candidates: the first alive lane is kept, the
            // candidates its box suppresses (its row of `ov`, fetched through the scalar registers) leave the alive set, repeat -- scalar bit arithmetic, ~10 instructions
            // per kept box where the loop used to be five lane broadcasts, an overlap test with its division and a ballot.  The kept lanes then write their boxes at once.
            if (!have_ov) build_ov();
            unsigned long long am = __ballot(alive), kept = 0ull;
            int nk1 = nk;
            const unsigned ov_lo = (unsigned)ov, ov_hi = (unsigned)(ov >> 32);
            while (am && nk1 < max_det) {
                const int t = __ffsll((long long)am) - 1;
                const unsigned long long row = (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)ov_lo, t) |
                                               ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)ov_hi, t) << 32);
                kept |= 1ull << t;
                ++nk1;
                am &= ~row;
                am &= ~(1ull << t);
            }
            if ((kept >> lane) & 1ull) {
                const int pos = nk + __popcll(kept & ((1ull << lane) - 1ull));
                kbox[4 * pos] = bx[0]; kbox[4 * pos + 1] = bx[1]; kbox[4 * pos + 2] = bx[2]; kbox[4 * pos + 3] = bx[3];
                karea[pos] = area;
                kidx[pos] = i;
            }
            nk = nk1;
            // publish: the boxes, then their count, then the turn (release: a wave that sees the count sees the boxes; one that sees the turn sees the count)
            if (lane == 0) {
                __hip_atomic_store(&s_nkeep, nk, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_store(&s_turn, (nk >= max_det || (c + 1) * 64 >= n) ? kTurnDone : c + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            if (nk >= max_det) break;
        }
    }
    __syncthreads();
    const int nkeep = s_nkeep;
    if (tid == 0) ndets[b] = nkeep;
    const int* so = sorder + (size_t)b * mcap;
    const float gain = lb[b * 5 + 0], padw = lb[b * 5 + 1], padh = lb[b * 5 + 2], H0 = lb[b * 5 + 3], W0 = lb[b * 5 + 4];
    for (int k = tid; k < nkeep; k += 256) {
        const int i = so[kidx[k]];
        const float* bo = cbox + ((size_t)b * cap + i) * 4;
        float x1 = (bo[0] - padw) / gain, y1 = (bo[1] - padh) / gain, x2 = (bo[2] - padw) / gain, y2 = (bo[3] - padh) / gain;
        x1 = fminf(fmaxf(x1, 0.f), W0); x2 = fminf(fmaxf(x2, 0.f), W0);
        y1 = fminf(fmaxf(y1, 0.f), H0); y2 = fminf(fmaxf(y2, 0.f), H0);
        float* o = dets + ((size_t)b * max_det + k) * 6;
        o[0] = rintf(x1); o[1] = rintf(y1); o[2] = rintf(x2); o[3] = rintf(y2);
        o[4] = cscore[(size_t)b * cap + i]; o[5] = ccls[(size_t)b * cap + i];
    }
    for (int k = tid; k < nkeep; k += 256) keep_idx[(size_t)b * max_det + k] = so[kidx[k]];   // candidate slots, for callers that want raw boxes
}

// ------------------------------------------------------------------------------------------------ host side
size_t y7t_post_ws_bytes(int B, int cap, int max_nms) {
    const size_t mcap = (size_t)(cap < max_nms ? cap : max_nms);    // NMS runs on the top max_nms candidates only (general.py:664-665)
    size_t o = 0;
    auto take = [&](size_t bytes) { o = (o + bytes + 255) & ~(size_t)255; };
    take((size_t)B * cap * 16); take((size_t)B * cap * 4); take((size_t)B * cap * 4); take((size_t)B * cap * 4);  // cbox cscore ccls cidx
    take((size_t)B * 4); take((size_t)B * 4);                                                                    // count nsorted
    take((size_t)B * mcap * 16); take((size_t)B * mcap * 4);                                                     // sbox sorder
    take((size_t)B * 5 * 4);                                                                                     // letterbox params
    return o + 256;
}

Y7TCandWs y7t_post_cand_ws(void* ws, int B, int cap) {
    char* base = (char*)ws;
    size_t o = 0;
    auto take = [&](size_t bytes) { char* p = base + o; o = (o + bytes + 255) & ~(size_t)255; return p; };
    Y7TCandWs c;
    c.cbox = (float*)take((size_t)B * cap * 16);
    c.cscore = (float*)take((size_t)B * cap * 4);
    c.ccls = (float*)take((size_t)B * cap * 4);
    c.cidx = (int*)take((size_t)B * cap * 4);
    c.count = (int*)take((size_t)B * 4);
    return c;
}

int y7t_post_run(const Y7TPostArgs& a, hipStream_t s) {
    const int B = a.B, cap = a.cap;
    if (a.max_det <= 0 || (size_t)a.max_det * 6 * sizeof(float) > 60u * 1024u) {      // k_nms_keep holds the kept boxes in LDS: 24 bytes each (box, area, position)
        y7t_set_error("postprocess: max_det = %d outside (0, %d] (the kept-list NMS keeps max_det boxes in LDS; the reference uses 300, general.py:619)",
                      a.max_det, (int)(60u * 1024u / (6 * sizeof(float))));
        return Y7T_E_ARG;
    }
    const size_t mcap = (size_t)(cap < a.max_nms ? cap : a.max_nms);
    char* base = (char*)a.ws;
    size_t o = 0;
    auto take = [&](size_t bytes) { char* p = base + o; o = (o + bytes + 255) & ~(size_t)255; return p; };
    float* cbox = (float*)take((size_t)B * cap * 16);
    float* cscore = (float*)take((size_t)B * cap * 4);
    float* ccls = (float*)take((size_t)B * cap * 4);
    int* cidx = (int*)take((size_t)B * cap * 4);
    int* count = (int*)take((size_t)B * 4);
    int* nsorted = (int*)take((size_t)B * 4);
    float* sbox = (float*)take((size_t)B * mcap * 16);
    int* sorder = (int*)take((size_t)B * mcap * 4);
    float* lb = (float*)take((size_t)B * 5 * 4);
    if (o + 256 > a.ws_bytes) { y7t_set_error("postprocess workspace too small (%zu < %zu)", a.ws_bytes, o + 256); return Y7T_E_ARG; }
    if (!a.predecoded) Y7T_HIP_CHECK(hipMemsetAsync(count, 0, sizeof(int) * B, s));
    Y7T_HIP_CHECK(hipMemcpyAsync(lb, a.letterbox_dev, sizeof(float) * 5 * B, hipMemcpyDeviceToDevice, s));
    DecodeArgs d;
    memset(&d, 0, sizeof(d));
    d.nl = a.nl; d.na = a.na; d.no = a.no; d.B = B; d.conf_thres = a.conf_thres; d.cap = cap;
    d.cbox = cbox; d.cscore = cscore; d.ccls = ccls; d.cidx = cidx; d.count = count;
    int row0 = 0;
    for (int l = 0; l < a.nl; ++l) {
        d.lv[l].p = a.head[l]; d.lv[l].ny = a.ny[l]; d.lv[l].nx = a.nx[l]; d.lv[l].stride = a.stride[l]; d.lv[l].row0 = row0;
        for (int k = 0; k < a.na; ++k) { d.lv[l].aw[k] = a.anchors[(l * a.na + k) * 2]; d.lv[l].ah[k] = a.anchors[(l * a.na + k) * 2 + 1]; }
        row0 += a.na * a.ny[l] * a.nx[l];
    }
    for (int l = 0; l < a.nl && !a.predecoded; ++l) {
        const long long tot = (long long)B * a.na * a.ny[l] * a.nx[l];
        int blocks = (int)((tot + 255) / 256); if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(k_decode_filter, dim3(blocks), dim3(256), 0, s, d, l);
        Y7T_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_rank_sort, dim3((cap + 255) / 256, B), dim3(256), 0, s, cbox, cscore, ccls, cidx, count, cap, a.max_nms, sbox, sorder, nsorted, (int)mcap);
    Y7T_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_nms_keep, dim3(B), dim3(256), (size_t)a.max_det * 6 * sizeof(float), s, sbox, nsorted, sorder, cbox, cscore, ccls, cap, (int)mcap,
                       a.iou_thres, a.max_det, lb, a.dets, a.ndets, a.keep_idx);
    Y7T_LAUNCH_CHECK();
    if (a.count_out) Y7T_HIP_CHECK(hipMemcpyAsync(a.count_out, count, sizeof(int) * B, hipMemcpyDeviceToDevice, s));
    return 0;
}
