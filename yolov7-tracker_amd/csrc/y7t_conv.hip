// y7t_conv.hip -- YOLOv7 Conv(+folded BN)+bias+activation as an im2col-free implicit GEMM on the CDNA4 matrix
// cores (v_mfma_f32_32x32x16_f16), NHWC fp16 activations, fp32 accumulate.
//
// Restates /root/reference/models/common.py:99-111 (Conv.fuseforward = act(conv2d(x, W') + b') with autopad :23-27)
// after utils/torch_utils.py:181-201 (fuse_conv_and_bn) has folded the BatchNorm into W', b'.
//
// GEMM view:  D[n][m] = sum_k Wp[n][k] * X[m][k]      n = output channel, m = output pixel (b, ho, wo),
//             k = (kh*KW + kw)*Cin + ci  -- the im2col matrix X is never built: each 16-byte K-chunk of a
//             pixel row is fetched straight from the NHWC tensor (8 consecutive channels of one tap), or from a
//             zero page when the tap falls into the padding / the row is past M / k is past K.
// Tiling:     256 threads = 4 waves (2 along n x 2 along m); block tile BN x BM, K-step 64 (128-byte rows);
//             both operands staged through LDS with `global_load_lds` (16 B per lane, lane-linear destination),
//             double-buffered: the loads of K-step t+1 are in flight while the MFMAs of step t run; one barrier
//             per K-step.  LDS rows are XOR-swizzled at 16-byte granularity (slot = chunk ^ (row & 7)) -- applied
//             to the per-lane SOURCE address on the way in and to the ds_read address on the way out.
//             The weight operand goes to MFMA's A side so that each lane ends up holding 4 consecutive output
//             channels of ONE pixel: the epilogue (bias + SiLU/LeakyReLU) packs them into one 8-byte NHWC store.
// Concat elimination: input and output are channel SLICES of wider NHWC buffers (ldin/cin_off, ldout/cout_off),
//             so producers write straight into their slot of a concat buffer and consumers read slices.
#include "y7t_common.h"
#include "y7t_det.h"

typedef _Float16 half_t;
typedef __attribute__((ext_vector_type(8))) _Float16 half8;
typedef __attribute__((ext_vector_type(4))) _Float16 half4;
typedef __attribute__((ext_vector_type(16))) float floatx16;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

__device__ __forceinline__ void load16_to_lds(const void* gptr, void* lds_wave_base) {
    // global -> LDS DMA, 16 bytes per lane; destination = wave-uniform base + lane*16
    __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)gptr, (LDS_AS void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float act_fn(float v, int act) {
    if (act == Y7T_ACT_SILU) return v / (1.0f + __expf(-v));
    if (act == Y7T_ACT_LEAKY) return v > 0.f ? v : 0.1f * v;
    return v;
}

template <int BM, int BN>
__global__ void __launch_bounds__(256) k_conv_igemm(const Y7TConvArgs p) {
    constexpr int BK = 64;                       // halfs per K-step (128-byte LDS rows)
    constexpr int WTN = BN / 2, WTM = BM / 2;    // wave tile
    constexpr int TN = WTN / 32, TM = WTM / 32;  // 32x32 MFMA tiles per wave
    constexpr int RM = BM / 32, RN = BN / 32;    // load rounds (32 rows per round over 4 waves)
    constexpr int STAGE = (BM + BN) * 128;       // bytes per LDS stage
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wm = wave & 1;
    // XCD-aware tile order: consecutive blocks of one XCD walk the m-tiles of one n-panel (weights stay in that L2)
    const int n_tiles_m = (p.M + BM - 1) / BM;
    const int bid = blockIdx.x;
    const int tile_m = bid % n_tiles_m, tile_n = bid / n_tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread load geometry ----
    const int lrow = wave * 8 + (lane >> 3);       // row inside a 32-row round
    const int gchunk = (lane & 7) ^ (lane >> 3);   // global 16-byte chunk this lane fetches (XOR swizzle)
    long long xbase[RM];                           // element offset of (b, hi0, wi0, cin_off) or < 0 when the row is invalid
    int xhi0[RM], xwi0[RM];
#pragma unroll
    for (int r = 0; r < RM; ++r) {
        const int m = m0 + r * 32 + lrow;
        if (m < p.M) {
            const int b = m / (p.Ho * p.Wo), rem = m - b * (p.Ho * p.Wo);
            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
            xhi0[r] = ho * p.stride - p.pad;
            xwi0[r] = wo * p.stride - p.pad;
            xbase[r] = (((long long)b * p.H + xhi0[r]) * p.W + xwi0[r]) * p.ldin + p.cin_off;
        } else {
            xhi0[r] = -(1 << 28); xwi0[r] = -(1 << 28); xbase[r] = 0;
        }
    }
    const half_t* wrow[RN];
#pragma unroll
    for (int r = 0; r < RN; ++r) wrow[r] = p.w + (size_t)(n0 + r * 32 + lrow) * p.K_pad + gchunk * 8;

    int k = gchunk * 8;          // this lane's k index for the current K-step
    int tap = k / p.Cin, ci = k - tap * p.Cin;
    const int nk = p.K_pad / BK;

    auto issue_loads = [&](int stage, int kt) {
        char* xs = smem + stage * STAGE;
        char* ws = xs + BM * 128;
        const int kh = (p.KW == 1) ? tap : (tap * 43) >> 7;   // tap / 3 for tap < 128
        const int kw = tap - kh * p.KW;
        const bool kvalid = (k < p.K);
        const long long tapoff = ((long long)kh * p.W + kw) * p.ldin + ci;
#pragma unroll
        for (int r = 0; r < RM; ++r) {
            const int hi = xhi0[r] + kh, wi = xwi0[r] + kw;
            const bool ok = kvalid && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            const half_t* src = ok ? p.in + xbase[r] + tapoff : p.zeros;
            load16_to_lds(src, xs + (r * 32 + wave * 8) * 128);
        }
#pragma unroll
        for (int r = 0; r < RN; ++r) load16_to_lds(wrow[r] + (size_t)kt * BK, ws + (r * 32 + wave * 8) * 128);
        // advance this lane's k by one K-step
        k += BK; ci += BK;
        while (ci >= p.Cin) { ci -= p.Cin; ++tap; }
    };

    floatx16 acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    issue_loads(0, 0);
    const int l31 = lane & 31, hi32 = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        __syncthreads();   // (vmcnt(0) + barrier): stage `cur` has landed for everyone, stage cur^1 is free
        if (kt + 1 < nk) issue_loads(cur ^ 1, kt + 1);
        const char* xs = smem + cur * STAGE;
        const char* ws = xs + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int q = ks * 2 + hi32;   // logical 16-byte chunk of this lane group
            half8 wf[TN], xf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i) {
                const int row = wn * WTN + i * 32 + l31;
                wf[i] = *(const half8*)(ws + row * 128 + ((q ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < TM; ++j) {
                const int row = wm * WTM + j * 32 + l31;
                xf[j] = *(const half8*)(xs + row * 128 + ((q ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: bias + activation, 4 consecutive channels per lane -> one NHWC store ----
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int m = m0 + wm * WTM + j * 32 + l31;
        if (m >= p.M) continue;
#pragma unroll
        for (int i = 0; i < TN; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0 + wn * WTN + i * 32 + 8 * g + 4 * hi32;
                if (n >= p.Cout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = act_fn(acc[i][j][g * 4 + e] + p.bias[n + e], p.act);
                const size_t o = (size_t)m * p.ldout + p.cout_off + n;
                if (p.out_f32) {
                    float* op = (float*)p.out + o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (n + e < p.Cout) op[e] = v[e];
                } else {
                    half_t* op = (half_t*)p.out + o;
                    if (n + 3 < p.Cout) {
                        half4 h = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
                        *(half4*)op = h;
                    } else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < p.Cout) op[e] = (half_t)v[e];
                    }
                }
            }
        }
    }
}

template <int BM, int BN>
static int launch_conv(const Y7TConvArgs& a, hipStream_t s) {
    constexpr unsigned lds = 2 * (BM + BN) * 128;
    static bool attr = false;
    if (!attr) {
        Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv_igemm<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = a.Cout_pad / BN;
    hipLaunchKernelGGL((k_conv_igemm<BM, BN>), dim3(tiles_m * tiles_n), dim3(256), lds, s, a);
    Y7T_LAUNCH_CHECK();
    return 0;
}

int y7t_conv_launch(const Y7TConvArgs& a, hipStream_t s) {
    if (a.Cin % 8 || a.ldin % 8 || a.cin_off % 8 || a.K_pad % 64 || a.Cout_pad % 64 || (!a.out_f32 && (a.ldout % 4 || a.cout_off % 4))) {
        y7t_set_error("conv: unsupported alignment (Cin=%d ldin=%d cin_off=%d K_pad=%d Cout_pad=%d ldout=%d cout_off=%d)", a.Cin, a.ldin,
                      a.cin_off, a.K_pad, a.Cout_pad, a.ldout, a.cout_off);
        return Y7T_E_ARG;
    }
    if (a.KW != 1 && a.KW != 3) { y7t_set_error("conv: kernel width %d unsupported (1 or 3)", a.KW); return Y7T_E_ARG; }
    if (a.Cout_pad % 128 == 0) return launch_conv<128, 128>(a, s);
    return launch_conv<128, 64>(a, s);
}
