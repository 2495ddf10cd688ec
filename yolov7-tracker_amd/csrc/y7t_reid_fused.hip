// y7t_reid_fused.hip -- OSNet x0_25 for DeepSORT's appearance branch (BASELINE config 4) as ONE kernel: one workgroup per crop, the whole
// network between the uint8 frame in HBM and the 512-d embedding runs out of LDS.
//
// Restates /root/reference/tracker/deepsort.py:19-41 (get_feature: crop ori_img[y1:y2, x1:x2]), tracker/reid_models/deepsort_reid.py:112-153
// (Extractor: /255, resize to 64 x 128 (W x H) INTER_LINEAR, Normalize) and tracker/reid_models/OSNet.py:28-438,567-579 (osnet_x0_25 in eval
// mode: conv1 7x7/2 + maxpool, three stages of two OSBlocks with channels 16 -> 64 -> 96 -> 128 and 1x1 + avgpool transitions, conv5, global
// average pool, fc 512 + BatchNorm1d + ReLU).  BatchNorm is folded on the host (tracker/reid.py::pack_fused).
//
// Why one workgroup per crop: a crop's activations are tiny (512 pixels x 64 channels at the widest point = 64 KiB in fp16) but the network is
// ~70 thin layers, so a layer-per-launch executor is launch- and HBM-round-trip-bound (y7t_reid.hip: 3.4 ms for 80 crops).  Here every
// intermediate lives in the CU's 160 KiB LDS, HBM sees the crop's source pixels, ~0.5 MB of L2-resident weights and the 2 KiB result.
//   * 1x1 convolutions (and the 7x7 stem as an implicit GEMM over 4-channel pixels, K = 7 rows x 8 pixels x 4) run on v_mfma_f32_16x16x16_f16:
//     A = weights (pre-packed on the host in the MFMA lane order, straight global -> VGPR), B = 16 pixels x 16 channels read from LDS with
//     ds_read_b64, D = 16 output channels x 16 pixels in fp32; bias / ReLU / fp16 convert in registers, ds_write_b64 back to LDS.
//   * depthwise 3x3 + BN + ReLU: VALU, 8 channels per thread (ds_read_b128 x 9 from a zero-haloed image, fp32 accumulate).
//   * OSBlock tail: the four gated streams are never summed in memory -- conv3 (W3 . sum_s g_s * t_s) is accumulated as sum_s (W3 * g_s) . t_s in the
//     MFMA accumulators, which also take the downsample branch; the residual add + ReLU is the accumulators' epilogue.
// Storage fp16, accumulation fp32 (the detector's convention); the fp32 op-list executor in y7t_reid.hip stays as the exact path and the checker.
#include "y7t_common.h"
#include "y7t_conv_common.h"
#include "y7t_reid_fused.h"
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) float floatx4;
typedef __attribute__((ext_vector_type(2))) _Float16 half2v;

namespace {

constexpr int NT = 512, NW = 8;     // threads / waves per workgroup

template <int C> struct Pitch { static constexpr int v = (C == 16) ? 32 : C * 2 + 16; };   // bytes per pixel row: 4 * odd dwords -> conflict-free b64 fragments

__device__ __forceinline__ floatx4 mfma16(half4 a, half4 b, floatx4 c) { return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0); }

// byte offset of pixel px: plain [px][C] or inside a (H+2) x (W+2) zero-haloed image
template <int W, int PITCH, bool HALO>
__device__ __forceinline__ int pxaddr(int px) {
    if (HALO) return ((px / W + 1) * (W + 2) + (px % W) + 1) * PITCH;
    return px * PITCH;
}

__device__ __forceinline__ half4 to_half4(floatx4 v) {
    half4 h; h[0] = (_Float16)v[0]; h[1] = (_Float16)v[1]; h[2] = (_Float16)v[2]; h[3] = (_Float16)v[3];
    return h;
}
__device__ __forceinline__ floatx4 relu4(floatx4 v) { floatx4 r; for (int e = 0; e < 4; ++e) r[e] = fmaxf(v[e], 0.f); return r; }

// one MFMA task: D[16 couts][16 pixels of group pg] += sum_nk A[nk] . B(pixels, channels nk*16..)
template <int NK, int W, int PS>
__device__ __forceinline__ floatx4 gemm_task(floatx4 acc, const half4 (&a)[NK], const char* src, int pg, int lane) {
    const char* p = src + pxaddr<W, PS, false>(pg * 16 + (lane & 15)) + (lane >> 4) * 8;
#pragma unroll
    for (int nk = 0; nk < NK; ++nk) acc = mfma16(a[nk], *(const half4*)(p + nk * 32), acc);
    return acc;
}

template <int NK>
__device__ __forceinline__ void load_frags(half4 (&a)[NK], const char* wfr, int ng, int lane) {
#pragma unroll
    for (int nk = 0; nk < NK; ++nk) a[nk] = ((const half4*)wfr)[(ng * NK + nk) * 64 + lane];
}

// dst[px][cout] = act(bias + W . src[px][:]) for every pixel; dst plain or haloed
template <int NPX, int W, int CIN, int COUT, int PS, int PD, bool HALO_D, bool RELU, bool BIAS>
__device__ __forceinline__ void conv1x1(const char* src, char* dst, const char* wfr, const float* bias, int wave, int lane) {
    constexpr int NG = COUT / 16, NK = CIN / 16, NPG = NPX / 16;
    auto store = [&](floatx4 acc, int pg, int ng) {
        if (RELU) acc = relu4(acc);
        *(half4*)(dst + pxaddr<W, PD, HALO_D>(pg * 16 + (lane & 15)) + ng * 32 + (lane >> 4) * 8) = to_half4(acc);
    };
    if constexpr (NPG >= NW) {
#pragma unroll
        for (int ng = 0; ng < NG; ++ng) {
            half4 a[NK];
            load_frags<NK>(a, wfr, ng, lane);
            floatx4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (BIAS) b4 = *(const floatx4*)(bias + ng * 16 + (lane >> 4) * 4);
#pragma unroll
            for (int pg = wave; pg < NPG; pg += NW) store(gemm_task<NK, W, PS>(b4, a, src, pg, lane), pg, ng);
        }
    } else {
        for (int t = wave; t < NPG * NG; t += NW) {
            const int pg = t % NPG, ng = t / NPG;
            half4 a[NK];
            load_frags<NK>(a, wfr, ng, lane);
            floatx4 b4 = {0.f, 0.f, 0.f, 0.f};
            if (BIAS) b4 = *(const floatx4*)(bias + ng * 16 + (lane >> 4) * 4);
            store(gemm_task<NK, W, PS>(b4, a, src, pg, lane), pg, ng);
        }
    }
}

// the OSBlock output accumulators: conv3 of the gated streams + the downsample branch, kept in registers across the whole block
template <int NPX, int COUT> struct AccShape {
    static constexpr int NPG = NPX / 16, NG = COUT / 16;
    static constexpr bool BIG = NPG >= NW;
    static constexpr int N = BIG ? NG * (NPG / NW) : (NPG * NG + NW - 1) / NW;
};

template <int NPX, int COUT> using AccArr = floatx4[AccShape<NPX, COUT>::N];

template <int NPX, int W, int CIN, int COUT, int PS, bool GATED>
__device__ __forceinline__ void acc_add(AccArr<NPX, COUT>& acc, const char* src, const char* wfr, const float* gate, int wave, int lane) {
    using S = AccShape<NPX, COUT>;
    constexpr int NK = CIN / 16;
    auto frags = [&](half4 (&a)[NK], int ng) {
        load_frags<NK>(a, wfr, ng, lane);
        if (GATED) {
#pragma unroll
            for (int nk = 0; nk < NK; ++nk) {
                const floatx4 g = *(const floatx4*)(gate + nk * 16 + (lane >> 4) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) a[nk][e] = (_Float16)((float)a[nk][e] * g[e]);
            }
        }
    };
    if constexpr (S::BIG) {
#pragma unroll
        for (int ng = 0; ng < S::NG; ++ng) {
            half4 a[NK];
            frags(a, ng);
#pragma unroll
            for (int i = 0; i < S::NPG / NW; ++i) acc[ng * (S::NPG / NW) + i] = gemm_task<NK, W, PS>(acc[ng * (S::NPG / NW) + i], a, src, wave + i * NW, lane);
        }
    } else {
#pragma unroll
        for (int i = 0; i < S::N; ++i) {
            const int t = wave + i * NW;
            if (t < S::NPG * S::NG) {
                half4 a[NK];
                frags(a, t / S::NPG);
                acc[i] = gemm_task<NK, W, PS>(acc[i], a, src, t % S::NPG, lane);
            }
        }
    }
}

// X[px][cout] = relu(acc + bias (+ X[px][cout]))      (IDENT: the block's input IS the identity and is overwritten in place)
template <int NPX, int W, int COUT, int PX, bool IDENT>
__device__ __forceinline__ void acc_store(const AccArr<NPX, COUT>& acc, char* X, const float* bias, int wave, int lane) {
    using S = AccShape<NPX, COUT>;
    auto put = [&](floatx4 v, int pg, int ng) {
        char* p = X + (pg * 16 + (lane & 15)) * PX + ng * 32 + (lane >> 4) * 8;
        const floatx4 b4 = *(const floatx4*)(bias + ng * 16 + (lane >> 4) * 4);
        if (IDENT) { const half4 id = *(const half4*)p; for (int e = 0; e < 4; ++e) v[e] += (float)id[e]; }
        for (int e = 0; e < 4; ++e) v[e] += b4[e];
        *(half4*)p = to_half4(relu4(v));
    };
    if constexpr (S::BIG) {
#pragma unroll
        for (int ng = 0; ng < S::NG; ++ng)
#pragma unroll
            for (int i = 0; i < S::NPG / NW; ++i) put(acc[ng * (S::NPG / NW) + i], wave + i * NW, ng);
    } else {
#pragma unroll
        for (int i = 0; i < S::N; ++i) {
            const int t = wave + i * NW;
            if (t < S::NPG * S::NG) put(acc[i], t % S::NPG, t / S::NPG);
        }
    }
}

// LightConv3x3's second half: T[px][c] = relu(bias[c] + sum_taps w[c][tap] * U[px + tap][c]) from the zero-haloed U; optionally the per-channel sums of
// the output over the crop (ChannelGate's global average pool), one partial per wave in gap[NW][MID]
template <int H, int W, int MID, bool GAP>
__device__ __forceinline__ void dwconv3(const char* U, char* T, const float* dww, const float* dwb, float* gap, int tid) {
    constexpr int PM = Pitch<MID>::v, CG = MID / 8, TASKS = H * W * CG;
    const int cg = tid % CG;
    float w[9][8], b[8], gs[8];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const floatx4 w0 = *(const floatx4*)(dww + (cg * 9 + t) * 8), w1 = *(const floatx4*)(dww + (cg * 9 + t) * 8 + 4);
        for (int e = 0; e < 4; ++e) { w[t][e] = w0[e]; w[t][4 + e] = w1[e]; }
    }
    {
        const floatx4 b0 = *(const floatx4*)(dwb + cg * 8), b1 = *(const floatx4*)(dwb + cg * 8 + 4);
        for (int e = 0; e < 4; ++e) { b[e] = b0[e]; b[4 + e] = b1[e]; gs[e] = 0.f; gs[4 + e] = 0.f; }
    }
#pragma unroll
    for (int task = tid; task < (TASKS + NT - 1) / NT * NT; task += NT) {
        if (task < TASKS) {
            const int px = task / CG, y = px / W, x = px % W;
            const char* u = U + (y * (W + 2) + x) * PM + cg * 16;      // tap (0, 0) = haloed pixel (y, x)
            float acc[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = b[e];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const half8 v = *(const half8*)(u + (kh * (W + 2) + kw) * PM);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(w[kh * 3 + kw][e], (float)v[e], acc[e]);
                }
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) { acc[e] = fmaxf(acc[e], 0.f); o[e] = (_Float16)acc[e]; if (GAP) gs[e] += acc[e]; }
            *(half8*)(T + px * PM + cg * 16) = o;
        }
    }
    if (GAP) {      // per-wave partial sums in fixed slots (no float atomics: the result must not depend on the order waves arrive in)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = gs[e];
#pragma unroll
            for (int off = CG; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
            if ((tid & 63) < CG) gap[(tid >> 6) * MID + cg * 8 + e] = v;
        }
    }
}

// one OSBlock (OSNet.py:223-279).  Param stream: conv1 frags + bias | gate fc1 w, b (4 floats), fc2 w, b | conv3 frags + bias (b3 + downsample bias) |
// downsample frags | 10 x (light 1x1 frags, depthwise weights [MID/8][9][8], bias [MID])
// O_WH / O_WL: where the block's parameters are staged in LDS (header = conv1, gate, conv3, downsample; lights = the ten LightConv3x3), O_WH < 0: the
// header is read from global memory.  A phase that fetches its own weights pays an L2 round trip (~1.4 us) before it can start, and a block is ~25 phases
// of a few hundred cycles of work each; one cooperative copy per block replaces them.
template <int H, int W, int CIN, int COUT, int MID, int R, int O_XIN, int O_XOUT, int O_X1, int O_T, int O_U, int O_S, int O_WH, int O_WL>
__device__ __forceinline__ const char* osblock(char* lds, const char* wp, int tid) {
    constexpr int NPX = H * W, PIN = Pitch<CIN>::v, POUT = Pitch<COUT>::v, PM = Pitch<MID>::v;
    constexpr bool DOWN = CIN != COUT;
    constexpr int U_BYTES = (H + 2) * (W + 2) * PM;
    constexpr int HDR_BYTES = (MID / 16) * (CIN / 16) * 512 + MID * 4 + R * MID * 4 + 16 + R * MID * 4 + MID * 4 + (COUT / 16) * (MID / 16) * 512 + COUT * 4 +
                              (DOWN ? (COUT / 16) * (CIN / 16) * 512 : 0);
    constexpr int LIGHTS_BYTES = 10 * ((MID / 16) * (MID / 16) * 512 + MID * 9 * 4 + MID * 4);
    const int wave = tid >> 6, lane = tid & 63;
    {
        auto stage = [&](char* dst, const char* src, int bytes) {
            for (int i = tid; i < bytes / 16; i += 4 * NT) {       // four loads in flight per thread before the first store
                floatx4 v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) if (i + q * NT < bytes / 16) v[q] = ((const floatx4*)src)[i + q * NT];
#pragma unroll
                for (int q = 0; q < 4; ++q) if (i + q * NT < bytes / 16) ((floatx4*)dst)[i + q * NT] = v[q];
            }
        };
        stage(lds + O_WL, wp + HDR_BYTES, LIGHTS_BYTES);
        if (O_WH >= 0) stage(lds + (O_WH >= 0 ? O_WH : 0), wp, HDR_BYTES);
        for (int i = tid; i < U_BYTES / 16; i += NT) ((floatx4*)(lds + O_U))[i] = floatx4{0.f, 0.f, 0.f, 0.f};      // the haloed image's border stays zero
    }
    const char* hp = O_WH >= 0 ? (const char*)(lds + (O_WH >= 0 ? O_WH : 0)) : wp;
    const char* lp = lds + O_WL;
    wp += HDR_BYTES + LIGHTS_BYTES;
    if (O_WH >= 0) __syncthreads();
    const char* c1f = hp; hp += (MID / 16) * (CIN / 16) * 512;
    const float* c1b = (const float*)hp; hp += MID * 4;
    const float* g_w1 = (const float*)hp; hp += R * MID * 4;
    const float* g_b1 = (const float*)hp; hp += 16;
    const float* g_w2 = (const float*)hp; hp += R * MID * 4;
    const float* g_b2 = (const float*)hp; hp += MID * 4;
    const char* c3f = hp; hp += (COUT / 16) * (MID / 16) * 512;
    const float* c3b = (const float*)hp; hp += COUT * 4;
    const char* dnf = hp;
    float* S = (float*)(lds + O_S);          // S[NW][MID]: per-wave pooled sums of the stream being finished | gate[MID]

    // conv1 (1x1 + BN + ReLU) -> x1
    conv1x1<NPX, W, CIN, MID, PIN, PM, false, true, true>(lds + O_XIN, lds + O_X1, c1f, c1b, wave, lane);
    floatx4 acc[AccShape<NPX, COUT>::N];
#pragma unroll
    for (int i = 0; i < AccShape<NPX, COUT>::N; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (DOWN) acc_add<NPX, W, CIN, COUT, PIN, false>(acc, lds + O_XIN, dnf, nullptr, wave, lane);
    __syncthreads();

#pragma unroll
    for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int j = 0; j <= s; ++j) {
            const char* lf = lp; lp += (MID / 16) * (MID / 16) * 512;
            const float* dww = (const float*)lp; lp += MID * 9 * 4;
            const float* dwb = (const float*)lp; lp += MID * 4;
            conv1x1<NPX, W, MID, MID, PM, PM, true, false, false>(lds + (j == 0 ? O_X1 : O_T), lds + O_U, lf, nullptr, wave, lane);
            __syncthreads();
            if (j == s) dwconv3<H, W, MID, true>(lds + O_U, lds + O_T, dww, dwb, S, tid);
            else dwconv3<H, W, MID, false>(lds + O_U, lds + O_T, dww, dwb, nullptr, tid);
            __syncthreads();
        }
        // ChannelGate (OSNet.py:162-220): g = sigmoid(fc2(relu(fc1(mean)))); the hidden layer is 1-2 values, every thread computes it
        float* gate = S + NW * MID;
        if (tid < MID) {
            float a = g_b2[tid];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                float hsum = g_b1[r];
                for (int c = 0; c < MID; ++c) {
                    float pooled = 0.f;
#pragma unroll
                    for (int w = 0; w < NW; ++w) pooled += S[w * MID + c];
                    hsum = __builtin_fmaf(g_w1[r * MID + c], pooled * (1.0f / (float)NPX), hsum);
                }
                a = __builtin_fmaf(g_w2[r * MID + tid], fmaxf(hsum, 0.f), a);
            }
            gate[tid] = 1.0f / (1.0f + __expf(-a));
        }
        __syncthreads();
        acc_add<NPX, W, MID, COUT, PM, true>(acc, lds + O_T, c3f, gate, wave, lane);
    }
    acc_store<NPX, W, COUT, POUT, !DOWN>(acc, lds + O_XOUT, c3b, wave, lane);
    __syncthreads();
    return wp;
}

// Conv1x1 + BN + ReLU then AvgPool2d(2)  (OSNet.py:338-350): X -> Y (full resolution) -> OUT (half resolution)
template <int H, int W, int C, int O_X, int O_Y, int O_OUT>
__device__ __forceinline__ const char* transition(char* lds, const char* wp, int tid) {
    constexpr int P = Pitch<C>::v, NPX = H * W;
    const char* f = wp; wp += (C / 16) * (C / 16) * 512;
    const float* b = (const float*)wp; wp += C * 4;
    conv1x1<NPX, W, C, C, P, P, false, true, true>(lds + O_X, lds + O_Y, f, b, tid >> 6, tid & 63);
    __syncthreads();
    constexpr int CG = C / 8, TASKS = (NPX / 4) * CG;
    for (int task = tid; task < TASKS; task += NT) {
        const int cg = task % CG, px = task / CG, y = px / (W / 2), x = px % (W / 2);
        const char* s = lds + O_Y + ((2 * y) * W + 2 * x) * P + cg * 16;
        const half8 a = *(const half8*)s, bq = *(const half8*)(s + P), c = *(const half8*)(s + W * P), d = *(const half8*)(s + W * P + P);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (_Float16)(((float)a[e] + (float)bq[e] + (float)c[e] + (float)d[e]) * 0.25f);
        *(half8*)(lds + O_OUT + px * P + cg * 16) = o;
    }
    __syncthreads();
    return wp;
}

// LDS map (bytes).  conv1 phase: CR (the normalised crop, 4-channel pixels, 3-pixel zero halo: 134 x 72 x 8) | C1 (64 x 32 x 16ch)
constexpr int O_CR = 0, CR_W = 72, CR_H = 134, O_C1 = 77824;
// stage 2 (32 x 16): X0 16ch | x1 | T | U (34 x 18 x 32) | S | X 64ch (pitch 144);  transition: Y at 0, pooled output over X
constexpr int S2_X0 = 0, S2_X1 = 16384, S2_T = 32768, S2_U = 49152, S2_S = 68736, S2_X = 77824, S2_Y = 0, S2_WL = 151552;      // + the lights' weights (11.25 KiB)
// stage 3 (16 x 8): input 64ch at 77824 (pitch 144) | X 96ch (pitch 208) | x1 / T (pitch 80) | U (18 x 10 x 80) | S
constexpr int S3_XIN = 77824, S3_X = 96256, S3_X1 = 0, S3_T = 10240, S3_U = 20480, S3_S = 34880, S3_Y = 0, S3_WL = 36864, S3_WH = 122880;
// stage 4 (8 x 4): input 96ch (pitch 208) | X 128ch (pitch 272) | x1 / T | U (10 x 6 x 80) | S;  conv5 output, pooled vector
constexpr int S4_XIN = 40960, S4_X = 49152, S4_X1 = 0, S4_T = 2560, S4_U = 5120, S4_S = 9984, S4_Y5 = 61440, S4_V = 71680, S4_WL = 73728, S4_WH = 108544;
constexpr int LDS_BYTES = 163840;

__global__ void __launch_bounds__(NT) k_osnet_x025(const Y7TReidFusedArgs p) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, n = blockIdx.x;
    const char* wp = p.blob;
    int stamp = 0;
#define REID_PROF() do { if (p.prof && n == 0 && tid == 0) p.prof[stamp] = clock64(); ++stamp; } while (0)
    REID_PROF();

    // ---- crop + /255 + bilinear resize to 128 x 64 + Normalize -> CR (fp16, channel order of the frame, 4th channel 0) ----
    for (int i = tid; i < CR_H * CR_W * 8 / 16; i += NT) ((floatx4*)(lds + O_CR))[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    {
        const float* b = p.boxes + 4 * (size_t)n;
        const uint8_t* frame = p.frames + (p.frame_idx ? (size_t)min(max(p.frame_idx[n], 0), p.n_frames - 1) * p.frame_stride : 0);
        int x1 = (int)b[0], y1 = (int)b[1], x2 = (int)b[2], y2 = (int)b[3];       // list(map(int, tlbr)), clipped like a numpy slice
        x1 = min(max(x1, 0), p.W); x2 = min(max(x2, 0), p.W); y1 = min(max(y1, 0), p.H); y2 = min(max(y2, 0), p.H);
        const int cw = x2 - x1, ch = y2 - y1;
        const float mean[3] = {0.485f, 0.456f, 0.406f}, isd[3] = {1.0f / 0.229f, 1.0f / 0.224f, 1.0f / 0.225f};
        if (cw > 0 && ch > 0) {
            for (int t = tid; t < 128 * 64; t += NT) {
                const int x = t & 63, y = t >> 6;
                const float fy = ((float)y + 0.5f) * ((float)ch / 128.0f) - 0.5f, fx = ((float)x + 0.5f) * ((float)cw / 64.0f) - 0.5f;
                int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
                const float wy = fy - (float)y0, wx = fx - (float)x0;
                const int yb = min(max(y0 + 1, 0), ch - 1), xb = min(max(x0 + 1, 0), cw - 1);
                y0 = min(max(y0, 0), ch - 1); x0 = min(max(x0, 0), cw - 1);
                const uint8_t* r0 = frame + ((size_t)(y1 + y0) * p.W + x1) * 3;
                const uint8_t* r1 = frame + ((size_t)(y1 + yb) * p.W + x1) * 3;
                half4 o;
                o[3] = (_Float16)0.f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float k255 = 0.00392156862745098f;
                    const float p00 = r0[x0 * 3 + c] * k255, p01 = r0[xb * 3 + c] * k255, p10 = r1[x0 * 3 + c] * k255, p11 = r1[xb * 3 + c] * k255;
                    const float v = (1.f - wy) * ((1.f - wx) * p00 + wx * p01) + wy * ((1.f - wx) * p10 + wx * p11);
                    o[c] = (_Float16)((v - mean[c]) * isd[c]);
                }
                *(half4*)(lds + O_CR + ((y + 3) * CR_W + x + 3) * 8) = o;
            }
        }
    }
    __syncthreads();

    REID_PROF();
    // ---- conv1: 7x7 / stride 2 / pad 3, 3 -> 16, BN, ReLU as an implicit GEMM: K-step (kh, half) = 4 pixels x 4 channels of input row 2*yo - 3 + kh ----
    {
        half4 a[14];
#pragma unroll
        for (int ks = 0; ks < 14; ++ks) a[ks] = ((const half4*)wp)[ks * 64 + lane];
        const float* bias = (const float*)(wp + 14 * 512);
        wp += 14 * 512 + 64;
        const floatx4 b4 = *(const floatx4*)(bias + (lane >> 4) * 4);
        const int q = lane >> 4, j = lane & 15;
        for (int pg = wave; pg < 128; pg += NW) {
            const int yo = pg >> 1, xo = (pg & 1) * 16 + j;
            floatx4 acc = b4;
            const char* base = lds + O_CR + ((2 * yo) * CR_W + 2 * xo + q) * 8;
#pragma unroll
            for (int kh = 0; kh < 7; ++kh)
#pragma unroll
                for (int h = 0; h < 2; ++h) acc = mfma16(a[kh * 2 + h], *(const half4*)(base + (kh * CR_W + 4 * h) * 8), acc);
            *(half4*)(lds + O_C1 + (yo * 32 + xo) * 32 + q * 8) = to_half4(relu4(acc));
        }
    }
    __syncthreads();
    REID_PROF();
    // ---- maxpool 3x3 / stride 2 / pad 1 -> X0 (32 x 16 x 16ch); inputs are >= 0, so skipping the padding == -inf padding ----
    for (int task = tid; task < 32 * 16 * 2; task += NT) {
        const int cg = task & 1, px = task >> 1, yo = px >> 4, xo = px & 15;
        half8 m;
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = (_Float16)0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int y = 2 * yo - 1 + kh, x = 2 * xo - 1 + kw;
                if ((unsigned)y < 64u && (unsigned)x < 32u) {
                    const half8 v = *(const half8*)(lds + O_C1 + (y * 32 + x) * 32 + cg * 16);
#pragma unroll
                    for (int e = 0; e < 8; ++e) m[e] = v[e] > m[e] ? v[e] : m[e];
                }
            }
        *(half8*)(lds + S2_X0 + px * 32 + cg * 16) = m;
    }
    __syncthreads();

    REID_PROF();
    wp = osblock<32, 16, 16, 64, 16, 1, S2_X0, S2_X, S2_X1, S2_T, S2_U, S2_S, -1, S2_WL>(lds, wp, tid);
    REID_PROF();
    wp = osblock<32, 16, 64, 64, 16, 1, S2_X, S2_X, S2_X1, S2_T, S2_U, S2_S, -1, S2_WL>(lds, wp, tid);
    REID_PROF();
    wp = transition<32, 16, 64, S2_X, S2_Y, S3_XIN>(lds, wp, tid);
    REID_PROF();
    wp = osblock<16, 8, 64, 96, 32, 1, S3_XIN, S3_X, S3_X1, S3_T, S3_U, S3_S, S3_WH, S3_WL>(lds, wp, tid);
    REID_PROF();
    wp = osblock<16, 8, 96, 96, 32, 1, S3_X, S3_X, S3_X1, S3_T, S3_U, S3_S, S3_WH, S3_WL>(lds, wp, tid);
    REID_PROF();
    wp = transition<16, 8, 96, S3_X, S3_Y, S4_XIN>(lds, wp, tid);
    REID_PROF();
    wp = osblock<8, 4, 96, 128, 32, 2, S4_XIN, S4_X, S4_X1, S4_T, S4_U, S4_S, S4_WH, S4_WL>(lds, wp, tid);
    REID_PROF();
    wp = osblock<8, 4, 128, 128, 32, 2, S4_X, S4_X, S4_X1, S4_T, S4_U, S4_S, S4_WH, S4_WL>(lds, wp, tid);
    REID_PROF();

    // ---- conv5 (1x1 + BN + ReLU), global average pool, fc + BatchNorm1d + ReLU ----
    {
        constexpr int P = Pitch<128>::v;
        const char* f = wp; wp += 8 * 8 * 512;
        const float* b = (const float*)wp; wp += 128 * 4;
        conv1x1<32, 4, 128, 128, P, P, false, true, true>(lds + S4_X, lds + S4_Y5, f, b, wave, lane);
        __syncthreads();
        float* v = (float*)(lds + S4_V);
        if (tid < 128) {
            float s = 0.f;
            for (int px = 0; px < 32; ++px) s += (float)*(const _Float16*)(lds + S4_Y5 + px * P + tid * 2);
            v[tid] = s * (1.0f / 32.0f);
        }
        __syncthreads();
        const half2v* wt = (const half2v*)wp;                 // [64 channel pairs][512 outputs][2]
        const float* fb = (const float*)(wp + 64 * 512 * 4);
        float a = fb[tid];
#pragma unroll 8
        for (int c2 = 0; c2 < 64; ++c2) {
            const half2v w2 = wt[c2 * 512 + tid];
            a = __builtin_fmaf((float)w2[0], v[2 * c2], a);
            a = __builtin_fmaf((float)w2[1], v[2 * c2 + 1], a);
        }
        p.feats[(size_t)n * 512 + tid] = fmaxf(a, 0.f);
    }
    REID_PROF();
}

}  // namespace

size_t y7t_reid_fused_blob_bytes() {
    auto block = [](size_t cin, size_t cout, size_t mid, size_t R) {
        size_t b = (mid / 16) * (cin / 16) * 512 + mid * 4 + R * mid * 4 + 16 + R * mid * 4 + mid * 4 + (cout / 16) * (mid / 16) * 512 + cout * 4;
        if (cin != cout) b += (cout / 16) * (cin / 16) * 512;
        return b + 10 * ((mid / 16) * (mid / 16) * 512 + mid * 9 * 4 + mid * 4);
    };
    size_t t = 14 * 512 + 64;
    t += block(16, 64, 16, 1) + block(64, 64, 16, 1) + 4 * 4 * 512 + 64 * 4;
    t += block(64, 96, 32, 1) + block(96, 96, 32, 1) + 6 * 6 * 512 + 96 * 4;
    t += block(96, 128, 32, 2) + block(128, 128, 32, 2);
    t += 8 * 8 * 512 + 128 * 4 + 64 * 512 * 4 + 512 * 4;
    return t;
}

int y7t_reid_fused_launch(const Y7TReidFusedArgs& a, hipStream_t s) {
    static Y7TOncePerDevice attr;      // (the attribute is per device: ADVICE r4)
    if (int e_ = y7t_once_per_device(attr, [&]() -> int {
        Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_osnet_x025, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
        return 0;
    })) return e_;
    if (a.N <= 0) return 0;
    static int prof = -1;
    static long long* prof_dev = nullptr;
    if (prof < 0) { prof = y7t_exp_switch("Y7T_REID_PROF", 0); if (prof) Y7T_HIP_CHECK(hipMalloc((void**)&prof_dev, 32 * sizeof(long long))); }
    Y7TReidFusedArgs b = a;
    b.prof = prof ? prof_dev : nullptr;
    hipLaunchKernelGGL(k_osnet_x025, dim3(a.N), dim3(NT), LDS_BYTES, s, b);
    Y7T_LAUNCH_CHECK();
    if (prof) {      // diagnostics only: synchronous
        long long h[32];
        Y7T_HIP_CHECK(hipStreamSynchronize(s));
        Y7T_HIP_CHECK(hipMemcpy(h, prof_dev, sizeof(h), hipMemcpyDeviceToHost));
        static const char* names[] = {"crop", "conv1 7x7", "maxpool", "block 2.0", "block 2.1", "transition 2", "block 3.0", "block 3.1", "transition 3", "block 4.0", "block 4.1",
                                      "conv5+gap+fc"};
        fprintf(stderr, "osnet_x025 workgroup 0 (kcycles):");
        for (int i = 0; i < 12; ++i) fprintf(stderr, " %s %.1f,", names[i], (h[i + 1] - h[i]) / 1e3);
        fprintf(stderr, " total %.1f\n", (h[12] - h[0]) / 1e3);
    }
    y7t_note_kernel("osnet_x025_fused");
    return 0;
}
