// y7t_reid.hip -- the appearance branch of DeepSORT (BASELINE config 4) on the device: crops of the detections taken straight from the
// uint8 frame in HBM, resized and normalised, through OSNet x0_25 to one embedding per detection.
//
// Restates /root/reference/tracker/deepsort.py:19-41 (get_feature: crop ori_img[y1:y2, x1:x2]),
// tracker/reid_models/deepsort_reid.py:112-153 (Extractor: /255, cv2.resize to 64 x 128 INTER_LINEAR on the float image, ToTensor,
// Normalize(mean, std) in the frame's channel order) and tracker/reid_models/OSNet.py:28-438 (ConvLayer, Conv1x1, Conv1x1Linear,
// LightConv3x3 = 1x1 linear + depthwise 3x3 + BN + ReLU, ChannelGate, OSBlock, OSNet.forward in eval mode -> fc output).
//
// The network is tiny (x0_25: 16/64/96/128 channels, ~60 MMAC per crop) and latency-bound at ~80 crops per frame, so it runs as a
// data-driven list of plain fp32 NHWC kernels (one thread per output value, weights through L1) -- BatchNorm folded on the host.
// Nothing here is worth MFMA: the whole forward is a few hundred microseconds next to an 18 ms detector forward.
#include "y7t_common.h"
#include "y7t_conv_common.h"
#include "y7t_reid_fused.h"
#include <string.h>
#include <stdlib.h>
#include <vector>

static_assert(sizeof(y7t_reid_op) == 96, "y7t_reid_op layout must match tracker/reid.py OP_DTYPE");

enum { R_CONV = 0, R_DWCONV3 = 1, R_MAXPOOL3S2 = 2, R_AVGPOOL2 = 3, R_GATE_ACC = 4, R_ADD_RELU = 5, R_GAP = 6, R_FC = 7, R_L2NORM = 8,
       // fp16 NHWC activations, convolutions on the detector's MFMA kernels (y7t_conv_launch): the reference's DeepSORT embedding network
       // (reid_models/deepsort_reid.py Net: 1.1 GMAC per crop -- GEMM-shaped, unlike OSNet's depthwise stacks)
       R_H_PACK = 9, R_H_CONV = 10, R_H_MAXPOOL_RELU = 11, R_H_RELU = 12, R_H_ADD_RELU = 13, R_H_GAP_L2NORM = 14 };

struct y7t_reid {
    std::vector<y7t_reid_op> ops;
    std::vector<int64_t> bufs;      // float offsets of the activation buffers (laid out for max_n crops)
    float* arena; size_t arena_floats;
    const float* w;
    int max_n, in_h, in_w, feat_dim;
    const char* fused_blob = nullptr;   // set: frame crops of a 128 x 64 x0_25 network go through k_osnet_x025 (y7t_reid_fused.hip)
    float* splitk_ws = nullptr;         // R_H_CONV: this extractor's split-K slabs (small launches), Y7T_SPLITK_WS_BYTES
    ~y7t_reid() { if (splitk_ws) (void)hipFree(splitk_ws); }
};

// crop + resize + normalise: out[n][y][x][c], c in the frame's channel order (BGR), cv2.INTER_LINEAR geometry on the float image
__global__ void __launch_bounds__(256) k_reid_crop(const uint8_t* __restrict__ frames, long long frame_stride, const int* __restrict__ frame_idx, int n_frames, int H, int W,
                                                   const float* __restrict__ boxes, int N, int oh, int ow, float* __restrict__ out) {
    const long long tot = (long long)N * oh * ow;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < tot; t += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(t % ow), y = (int)((t / ow) % oh), n = (int)(t / ((long long)ow * oh));
        const float* b = boxes + 4 * (size_t)n;
        const uint8_t* frame = frames + (frame_idx ? (size_t)min(max(frame_idx[n], 0), n_frames - 1) * frame_stride : 0);
        int x1 = (int)b[0], y1 = (int)b[1], x2 = (int)b[2], y2 = (int)b[3];       // list(map(int, tlbr))
        x1 = min(max(x1, 0), W); x2 = min(max(x2, 0), W); y1 = min(max(y1, 0), H); y2 = min(max(y2, 0), H);
        const int cw = x2 - x1, ch = y2 - y1;
        float* o = out + (size_t)t * 3;
        const float mean[3] = {0.485f, 0.456f, 0.406f}, sd[3] = {0.229f, 0.224f, 0.225f};
        if (cw <= 0 || ch <= 0) { o[0] = o[1] = o[2] = 0.f; continue; }
        const float fy = ((float)y + 0.5f) * ((float)ch / (float)oh) - 0.5f, fx = ((float)x + 0.5f) * ((float)cw / (float)ow) - 0.5f;
        int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
        const float wy = fy - (float)y0, wx = fx - (float)x0;
        const int yb = min(max(y0 + 1, 0), ch - 1), xb = min(max(x0 + 1, 0), cw - 1);
        y0 = min(max(y0, 0), ch - 1); x0 = min(max(x0, 0), cw - 1);
        const uint8_t* r0 = frame + ((size_t)(y1 + y0) * W + x1) * 3;
        const uint8_t* r1 = frame + ((size_t)(y1 + yb) * W + x1) * 3;
        for (int c = 0; c < 3; ++c) {
            const float p00 = r0[x0 * 3 + c] / 255.0f, p01 = r0[xb * 3 + c] / 255.0f, p10 = r1[x0 * 3 + c] / 255.0f, p11 = r1[xb * 3 + c] / 255.0f;
            const float v = (1.f - wy) * ((1.f - wx) * p00 + wx * p01) + wy * ((1.f - wx) * p10 + wx * p11);
            o[c] = (v - mean[c]) / sd[c];
        }
    }
}

// dense conv k x k, stride s, padding p, + bias (folded BN) + optional ReLU; weights [co][kh][kw][ci]; NHWC fp32
__global__ void __launch_bounds__(256) k_reid_conv(const float* __restrict__ in, int N, int H, int W, int Ci, const float* __restrict__ w,
                                                   const float* __restrict__ bias, int k, int s, int p, int Ho, int Wo, int Co, int relu,
                                                   int kmajor, float* __restrict__ out) {
    const long long tot = (long long)N * Ho * Wo * Co;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < tot; t += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(t % Co);
        const long long px = t / Co;
        const int xo = (int)(px % Wo), yo = (int)((px / Wo) % Ho), n = (int)(px / ((long long)Wo * Ho));
        float acc = bias ? bias[co] : 0.f;
        const float* wc = kmajor ? w + co : w + (size_t)co * k * k * Ci;     // kmajor: (kh, kw, ci, co) -- a wave reads one line per ci
        const size_t wstep = kmajor ? (size_t)Co : 1;
        for (int kh = 0; kh < k; ++kh) {
            const int y = yo * s - p + kh;
            if ((unsigned)y >= (unsigned)H) continue;
            for (int kw = 0; kw < k; ++kw) {
                const int x = xo * s - p + kw;
                if ((unsigned)x >= (unsigned)W) continue;
                const float* ip = in + (((size_t)n * H + y) * W + x) * Ci;
                const float* wp = wc + (size_t)(kh * k + kw) * Ci * wstep;
                for (int ci = 0; ci < Ci; ++ci) acc += ip[ci] * wp[ci * wstep];
            }
        }
        out[t] = relu ? fmaxf(acc, 0.f) : acc;
    }
}

// depthwise 3x3, padding 1, + bias + ReLU; weights [c][3][3]
__global__ void __launch_bounds__(256) k_reid_dwconv3(const float* __restrict__ in, int N, int H, int W, int C, const float* __restrict__ w,
                                                      const float* __restrict__ bias, int relu, float* __restrict__ out) {
    const long long tot = (long long)N * H * W * C;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < tot; t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const long long px = t / C;
        const int x = (int)(px % W), y = (int)((px / W) % H), n = (int)(px / ((long long)W * H));
        float acc = bias[c];
        for (int kh = 0; kh < 3; ++kh) {
            const int yy = y - 1 + kh;
            if ((unsigned)yy >= (unsigned)H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int xx = x - 1 + kw;
                if ((unsigned)xx >= (unsigned)W) continue;
                acc += in[(((size_t)n * H + yy) * W + xx) * C + c] * w[c * 9 + kh * 3 + kw];
            }
        }
        out[t] = relu ? fmaxf(acc, 0.f) : acc;
    }
}

// mode 0: max 3x3 / stride 2 / pad 1 (-inf padding); mode 1: average 2x2 / stride 2
__global__ void __launch_bounds__(256) k_reid_pool(const float* __restrict__ in, int N, int H, int W, int C, int mode, int Ho, int Wo, float* __restrict__ out) {
    const long long tot = (long long)N * Ho * Wo * C;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < tot; t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const long long px = t / C;
        const int xo = (int)(px % Wo), yo = (int)((px / Wo) % Ho), n = (int)(px / ((long long)Wo * Ho));
        float r;
        if (mode == 0) {
            r = -3.0e38f;
            for (int kh = 0; kh < 3; ++kh) for (int kw = 0; kw < 3; ++kw) {
                const int y = yo * 2 - 1 + kh, x = xo * 2 - 1 + kw;
                if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) r = fmaxf(r, in[(((size_t)n * H + y) * W + x) * C + c]);
            }
        } else {
            r = 0.f;
            for (int kh = 0; kh < 2; ++kh) for (int kw = 0; kw < 2; ++kw) r += in[(((size_t)n * H + yo * 2 + kh) * W + xo * 2 + kw) * C + c];
            r *= 0.25f;
        }
        out[t] = r;
    }
}

// global average pool: out[n][c] = mean over H*W; one workgroup per crop
__global__ void __launch_bounds__(256) k_reid_gap(const float* __restrict__ in, int HW, int C, float* __restrict__ out) {
    __shared__ float part[256];
    const int n = blockIdx.x, c0 = blockIdx.y * 256;     // grid.y: 256-channel slices (one for C <= 256)
    const int Cc = C - c0 < 256 ? C - c0 : 256;
    const float* base = in + (size_t)n * HW * C + c0;
    const int groups = 256 / Cc > 0 ? 256 / Cc : 1;      // `groups` row-partitions per channel
    const int c = threadIdx.x % Cc, g = threadIdx.x / Cc;
    float acc = 0.f;
    if (g < groups) for (int i = g; i < HW; i += groups) acc += base[(size_t)i * C + c];
    part[threadIdx.x] = (g < groups) ? acc : 0.f;
    __syncthreads();
    if (threadIdx.x < Cc) {
        float s = 0.f;
        for (int k = 0; k < groups; ++k) s += part[k * Cc + threadIdx.x];
        out[(size_t)n * C + c0 + threadIdx.x] = s / (float)HW;
    }
}

// ChannelGate (OSNet.py:162-220) on the pooled vector: gate[n][c] = sigmoid(fc2(relu(fc1(pooled[n]))));  one workgroup per crop
__global__ void __launch_bounds__(128) k_reid_gate(const float* __restrict__ pooled, int C, int R, const float* __restrict__ w1, const float* __restrict__ b1,
                                                   const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ gate) {
    __shared__ float hid[64];
    const int n = blockIdx.x;
    const float* p = pooled + (size_t)n * C;
    if ((int)threadIdx.x < R) {
        float a = b1[threadIdx.x];
        for (int c = 0; c < C; ++c) a += w1[threadIdx.x * C + c] * p[c];
        hid[threadIdx.x] = fmaxf(a, 0.f);
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float a = b2[c];
        for (int r = 0; r < R; ++r) a += w2[c * R + r] * hid[r];
        gate[(size_t)n * C + c] = 1.0f / (1.0f + expf(-a));
    }
}

// acc = (first ? 0 : acc) + x * gate[n][c]
__global__ void __launch_bounds__(256) k_reid_scale_acc(const float* __restrict__ x, const float* __restrict__ gate, int HW, int C, long long tot, int first,
                                                        float* __restrict__ acc) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < tot; t += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const int n = (int)(t / ((long long)HW * C));
        const float v = x[t] * gate[(size_t)n * C + c];
        acc[t] = first ? v : acc[t] + v;
    }
}

__global__ void __launch_bounds__(256) k_reid_add_relu(const float* __restrict__ a, const float* __restrict__ b, long long tot, float* __restrict__ out) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < tot; t += (long long)gridDim.x * blockDim.x) out[t] = fmaxf(a[t] + b[t], 0.f);
}

// fc: out[n][o] = relu(bias[o] + sum_c w[o][c] * in[n][c])   (Linear + folded BatchNorm1d + ReLU, OSNet.py:367-386)
__global__ void __launch_bounds__(256) k_reid_fc(const float* __restrict__ in, int N, int C, const float* __restrict__ w, const float* __restrict__ bias, int O,
                                                 int relu, float* __restrict__ out) {
    const long long tot = (long long)N * O;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < tot; t += (long long)gridDim.x * blockDim.x) {
        const int o = (int)(t % O), n = (int)(t / O);
        float a = bias[o];
        for (int c = 0; c < C; ++c) a += w[(size_t)o * C + c] * in[(size_t)n * C + c];
        out[t] = relu ? fmaxf(a, 0.f) : a;
    }
}

// x / |x|_2 per crop (deepsort_reid.py:104: x.div(x.norm(p=2, dim=1, keepdim=True))); one workgroup per crop, fixed-shape tree sum
__global__ void __launch_bounds__(256) k_reid_l2norm(const float* __restrict__ in, int C, float* __restrict__ out) {
    __shared__ float part[256];
    const float* x = in + (size_t)blockIdx.x * C;
    float acc = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) acc += x[c] * x[c];
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    const float nrm = sqrtf(part[0]);
    for (int c = threadIdx.x; c < C; c += 256) out[(size_t)blockIdx.x * C + c] = x[c] / nrm;
}

// ---- fp16 NHWC helpers of the MFMA op list --------------------------------------------------------------------------------
// fp32 (N, H, W, 3) crops -> fp16 (N, H, W, 16), channels 3..15 zero (the conv kernels read 16-byte channel groups)
__global__ void __launch_bounds__(256) k_h_pack(const float* __restrict__ in, long long npix, half_t* __restrict__ out) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < npix; t += (long long)gridDim.x * blockDim.x) {
        half8 lo, hi;
#pragma unroll
        for (int e = 0; e < 8; ++e) { lo[e] = (half_t)0.f; hi[e] = (half_t)0.f; }
        lo[0] = (half_t)in[t * 3]; lo[1] = (half_t)in[t * 3 + 1]; lo[2] = (half_t)in[t * 3 + 2];
        *(half8*)(out + t * 16) = lo;
        *(half8*)(out + t * 16 + 8) = hi;
    }
}

// ReLU then MaxPool2d(3, 2, padding=1) (max and ReLU commute), C % 8 == 0
__global__ void __launch_bounds__(256) k_h_maxpool3s2_relu(const half_t* __restrict__ in, int N, int H, int W, int C, int Ho, int Wo, half_t* __restrict__ out) {
    const int C8 = C / 8;
    const long long tot = (long long)N * Ho * Wo * C8;
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < tot; t += (long long)gridDim.x * blockDim.x) {
        const int c8 = (int)(t % C8);
        const long long px = t / C8;
        const int xo = (int)(px % Wo), yo = (int)((px / Wo) % Ho), n = (int)(px / ((long long)Wo * Ho));
        half8 m;
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (half_t)0.f;
        for (int kh = 0; kh < 3; ++kh) {
            const int y = yo * 2 - 1 + kh;
            if ((unsigned)y >= (unsigned)H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int x = xo * 2 - 1 + kw;
                if ((unsigned)x >= (unsigned)W) continue;
                const half8 v = *(const half8*)(in + (((size_t)n * H + y) * W + x) * C + c8 * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
            }
        }
        *(half8*)(out + (size_t)px * C + c8 * 8) = m;
    }
}

// out = relu(a + b) (b == nullptr: relu(a)); fp32 add of the two fp16 values, one rounding
__global__ void __launch_bounds__(256) k_h_add_relu(const half_t* __restrict__ a, const half_t* __restrict__ b, long long n8, half_t* __restrict__ out) {
    for (long long t = blockIdx.x * (long long)blockDim.x + threadIdx.x; t < n8; t += (long long)gridDim.x * blockDim.x) {
        const half8 x = *(const half8*)(a + t * 8);
        half8 r;
        if (b) {
            const half8 y = *(const half8*)(b + t * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = (half_t)fmaxf((float)x[e] + (float)y[e], 0.f);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = x[e] > (half_t)0.f ? x[e] : (half_t)0.f;
        }
        *(half8*)(out + t * 8) = r;
    }
}

// AvgPool over the whole map + x / |x|_2 (deepsort_reid.py:96-105), one workgroup per crop: fp16 (HW, C) -> fp32 (C), C <= 1024
__global__ void __launch_bounds__(256) k_h_gap_l2norm(const half_t* __restrict__ in, int HW, int C, float* __restrict__ out) {
    __shared__ float mean[1024];
    __shared__ float part[256];
    const half_t* x = in + (size_t)blockIdx.x * HW * C;
    float ss = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float a = 0.f;
        for (int i = 0; i < HW; ++i) a += (float)x[(size_t)i * C + c];
        a /= (float)HW;
        mean[c] = a;
        ss += a * a;
    }
    part[threadIdx.x] = ss;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    const float nrm = sqrtf(part[0]);
    for (int c = threadIdx.x; c < C; c += 256) out[(size_t)blockIdx.x * C + c] = mean[c] / nrm;
}

static int blocks_for(long long tot) { long long b = (tot + 255) / 256; return (int)(b < 1 ? 1 : (b > 16384 ? 16384 : b)); }

extern "C" int y7t_reid_create(const y7t_reid_op* ops, int n_ops, const int64_t* buf_offsets, int n_bufs, void* arena, size_t arena_bytes, const void* weights_f32,
                               int max_crops, int in_h, int in_w, int feat_dim, y7t_reid** out) {
    Y7T_ARG_CHECK(ops && n_ops > 0 && buf_offsets && n_bufs > 0 && arena && weights_f32 && out && max_crops > 0 && in_h > 0 && in_w > 0 && feat_dim > 0);
    bool need_ws = false;
    for (int i = 0; i < n_ops; ++i) {
        Y7T_ARG_CHECK(ops[i].type >= R_CONV && ops[i].type <= R_H_GAP_L2NORM);
        Y7T_ARG_CHECK(ops[i].in_buf >= 0 && ops[i].in_buf < n_bufs && ops[i].out_buf >= 0 && ops[i].out_buf < n_bufs && ops[i].aux_buf < n_bufs);
        if (ops[i].type == R_GATE_ACC) Y7T_ARG_CHECK(ops[i].C <= 256 && ops[i].R <= 64 && ops[i].R >= 1);
        if (ops[i].type == R_H_CONV) Y7T_ARG_CHECK(ops[i].C % 8 == 0 && ops[i].Co % 64 == 0 && (ops[i].k == 1 || ops[i].k == 3) && ops[i].b_off >= 0);
        if (ops[i].type == R_H_MAXPOOL_RELU || ops[i].type == R_H_RELU || ops[i].type == R_H_ADD_RELU) Y7T_ARG_CHECK(ops[i].C % 8 == 0);
        if (ops[i].type == R_H_ADD_RELU) Y7T_ARG_CHECK(ops[i].aux_buf >= 0);
        if (ops[i].type == R_H_GAP_L2NORM) Y7T_ARG_CHECK(ops[i].C <= 1024);
        need_ws = need_ws || ops[i].type == R_H_CONV;
    }
    y7t_reid* r = new y7t_reid();
    if (need_ws && hipMalloc((void**)&r->splitk_ws, Y7T_SPLITK_WS_BYTES) != hipSuccess) {
        delete r;
        y7t_set_error("y7t_reid_create: cannot allocate the split-K workspace");
        return Y7T_E_HIP;
    }
    r->ops.assign(ops, ops + n_ops);
    r->bufs.assign(buf_offsets, buf_offsets + n_bufs);
    r->arena = (float*)arena; r->arena_floats = arena_bytes / 4; r->w = (const float*)weights_f32;
    r->max_n = max_crops; r->in_h = in_h; r->in_w = in_w; r->feat_dim = feat_dim;
    *out = r;
    return 0;
}

extern "C" int y7t_reid_destroy(y7t_reid* r) { delete r; return 0; }

extern "C" size_t y7t_reid_fused_blob_size(void) { return y7t_reid_fused_blob_bytes(); }

extern "C" int y7t_reid_set_fused(y7t_reid* r, const void* blob, size_t blob_bytes) {
    Y7T_ARG_CHECK(r);
    if (!blob) { r->fused_blob = nullptr; return 0; }
    if (r->in_h != 128 || r->in_w != 64 || r->feat_dim != 512) { y7t_set_error("reid: the fused kernel is OSNet x0_25 on 128 x 64 crops with 512 features"); return Y7T_E_ARG; }
    if (blob_bytes != y7t_reid_fused_blob_bytes()) {
        y7t_set_error("reid: fused parameter blob has %zu bytes, the kernel consumes %zu", blob_bytes, y7t_reid_fused_blob_bytes());
        return Y7T_E_ARG;
    }
    r->fused_blob = (const char*)blob;
    return 0;
}

static int reid_forward_impl(y7t_reid* r, const void* frames_u8, int n_frames, int H, int W, const float* boxes, const int* frame_idx, int N, const float* crops_f32,
                             float* feats, y7t_stream stream) {
    Y7T_ARG_CHECK(r && feats && N >= 0 && N <= r->max_n);
    Y7T_ARG_CHECK((frames_u8 && boxes && H > 0 && W > 0 && n_frames >= 1) || crops_f32);
    if (N == 0) return 0;
    hipStream_t s = (hipStream_t)stream;
    const long long fstride = (long long)H * W * 3;
    if (r->fused_blob && !crops_f32) {
        Y7TReidFusedArgs a;
        a.frames = (const uint8_t*)frames_u8; a.frame_stride = fstride; a.H = H; a.W = W; a.boxes = boxes; a.frame_idx = frame_idx; a.n_frames = n_frames; a.N = N;
        a.blob = r->fused_blob; a.feats = feats; a.prof = nullptr;
        return y7t_reid_fused_launch(a, s);
    }
    float* b0 = r->arena + r->bufs[0];
    if (crops_f32) Y7T_HIP_CHECK(hipMemcpyAsync(b0, crops_f32, sizeof(float) * (size_t)N * r->in_h * r->in_w * 3, hipMemcpyDeviceToDevice, s));
    else {
        hipLaunchKernelGGL(k_reid_crop, dim3(blocks_for((long long)N * r->in_h * r->in_w)), dim3(256), 0, s, (const uint8_t*)frames_u8, fstride, frame_idx, n_frames, H, W, boxes, N,
                           r->in_h, r->in_w, b0);
        Y7T_LAUNCH_CHECK();
    }
    for (const y7t_reid_op& op : r->ops) {
        const float* in = r->arena + r->bufs[op.in_buf];
        float* out = r->arena + r->bufs[op.out_buf];
        float* aux = op.aux_buf >= 0 ? r->arena + r->bufs[op.aux_buf] : nullptr;
        const float* w = r->w + op.w_off;
        const float* bias = op.b_off >= 0 ? r->w + op.b_off : nullptr;
        switch (op.type) {
        case R_CONV:
            hipLaunchKernelGGL(k_reid_conv, dim3(blocks_for((long long)N * op.Ho * op.Wo * op.Co)), dim3(256), 0, s, in, N, op.H, op.W, op.C, w, bias, op.k, op.s, op.p,
                               op.Ho, op.Wo, op.Co, op.relu, op.w_kmajor, out);
            break;
        case R_DWCONV3:
            hipLaunchKernelGGL(k_reid_dwconv3, dim3(blocks_for((long long)N * op.H * op.W * op.C)), dim3(256), 0, s, in, N, op.H, op.W, op.C, w, bias, op.relu, out);
            break;
        case R_MAXPOOL3S2: case R_AVGPOOL2:
            hipLaunchKernelGGL(k_reid_pool, dim3(blocks_for((long long)N * op.Ho * op.Wo * op.C)), dim3(256), 0, s, in, N, op.H, op.W, op.C, op.type == R_AVGPOOL2, op.Ho, op.Wo, out);
            break;
        case R_GATE_ACC: {   // aux: [N][C] pooled | [N][C] gates; out: accumulator
            float* pooled = aux;
            float* gate = aux + (size_t)r->max_n * op.C;
            hipLaunchKernelGGL(k_reid_gap, dim3(N), dim3(256), 0, s, in, op.H * op.W, op.C, pooled);
            hipLaunchKernelGGL(k_reid_gate, dim3(N), dim3(128), 0, s, (const float*)pooled, op.C, op.R, w, r->w + op.b_off, r->w + op.w2_off, r->w + op.b2_off, gate);
            const long long tot = (long long)N * op.H * op.W * op.C;
            hipLaunchKernelGGL(k_reid_scale_acc, dim3(blocks_for(tot)), dim3(256), 0, s, in, (const float*)gate, op.H * op.W, op.C, tot, op.relu /* first */, out);
            break;
        }
        case R_ADD_RELU: {
            const long long tot = (long long)N * op.H * op.W * op.C;
            hipLaunchKernelGGL(k_reid_add_relu, dim3(blocks_for(tot)), dim3(256), 0, s, in, (const float*)aux, tot, out);
            break;
        }
        case R_GAP:
            hipLaunchKernelGGL(k_reid_gap, dim3(N, (op.C + 255) / 256), dim3(256), 0, s, in, op.H * op.W, op.C, out);
            break;
        case R_L2NORM:
            hipLaunchKernelGGL(k_reid_l2norm, dim3(N), dim3(256), 0, s, in, op.C, out);
            break;
        case R_H_PACK: {
            const long long npix = (long long)N * op.H * op.W;
            hipLaunchKernelGGL(k_h_pack, dim3(blocks_for(npix)), dim3(256), 0, s, in, npix, (half_t*)out);
            break;
        }
        case R_H_CONV: {     // weights: fp16 [Co][K_pad] (k = (kh * 3 + kw) * C + ci) stored in the float blob at w_off; bias fp32 [Co] at b_off
            Y7TConvArgs a;
            memset(&a, 0, sizeof(a));
            a.in = (const _Float16*)in; a.ldin = op.C; a.cin_off = 0; a.B = N; a.H = op.H; a.W = op.W; a.Cin = op.C;
            a.w = (const _Float16*)w; a.bias = bias; a.out = out; a.ldout = op.Co; a.cout_off = 0; a.out_f32 = 0;
            a.Ho = op.Ho; a.Wo = op.Wo; a.Cout = op.Co; a.Cout_pad = op.Co; a.KH = op.k; a.KW = op.k; a.stride = op.s; a.pad = op.p;
            a.K = op.k * op.k * op.C; a.K_pad = (a.K + 63) / 64 * 64; a.M = N * op.Ho * op.Wo; a.act = Y7T_ACT_NONE; a.splitk_ws = r->splitk_ws;
            // plain (kh, kw, ci) weights: the generic kernel is the one tested on this layout for every shape here; Y7T_REID_PATCH=1 lets the
            // 64 / 128-channel stride-1 layers of large batches take the LDS-patch kernel (faster; enable by default once it is covered)
            { static int rp = -1; if (rp < 0) rp = y7t_exp_switch("Y7T_REID_PATCH", 0); a.no_patch = !rp; }
            if (int rc = y7t_conv_launch(a, s)) return rc;
            break;
        }
        case R_H_MAXPOOL_RELU:
            hipLaunchKernelGGL(k_h_maxpool3s2_relu, dim3(blocks_for((long long)N * op.Ho * op.Wo * (op.C / 8))), dim3(256), 0, s, (const half_t*)in, N, op.H, op.W, op.C,
                               op.Ho, op.Wo, (half_t*)out);
            break;
        case R_H_RELU: case R_H_ADD_RELU: {
            const long long n8 = (long long)N * op.H * op.W * (op.C / 8);
            hipLaunchKernelGGL(k_h_add_relu, dim3(blocks_for(n8)), dim3(256), 0, s, (const half_t*)in, op.type == R_H_ADD_RELU ? (const half_t*)aux : (const half_t*)nullptr, n8,
                               (half_t*)out);
            break;
        }
        case R_H_GAP_L2NORM:
            hipLaunchKernelGGL(k_h_gap_l2norm, dim3(N), dim3(256), 0, s, (const half_t*)in, op.H * op.W, op.C, out);
            break;
        case R_FC:
            hipLaunchKernelGGL(k_reid_fc, dim3(blocks_for((long long)N * op.Co)), dim3(256), 0, s, in, N, op.C, w, bias, op.Co, op.relu, out);
            break;
        }
        Y7T_LAUNCH_CHECK();
    }
    const y7t_reid_op& last = r->ops.back();
    Y7T_HIP_CHECK(hipMemcpyAsync(feats, r->arena + r->bufs[last.out_buf], sizeof(float) * (size_t)N * r->feat_dim, hipMemcpyDeviceToDevice, s));
    return 0;
}

extern "C" int y7t_reid_forward(y7t_reid* r, const void* frame_u8, int H, int W, const float* boxes, int N, const float* crops_f32, float* feats, y7t_stream stream) {
    return reid_forward_impl(r, frame_u8, 1, H, W, boxes, nullptr, N, crops_f32, feats, stream);
}

extern "C" int y7t_reid_forward_batch(y7t_reid* r, const void* frames_u8, int n_frames, int H, int W, const float* boxes, const int* frame_idx, int N, float* feats,
                                      y7t_stream stream) {
    Y7T_ARG_CHECK(frames_u8 && frame_idx);
    return reid_forward_impl(r, frames_u8, n_frames, H, W, boxes, frame_idx, N, nullptr, feats, stream);
}
