// y7t_conv_patch.hip -- 3x3 / stride 1 / pad 1 Conv(+folded BN)+bias+activation with an LDS-resident input PATCH.
//
// Same math as k_conv_igemm (y7t_conv.hip; /root/reference/models/common.py:99-111 after utils/torch_utils.py:181-201), other
// data movement.  The generic kernel fetches every (pixel, tap) row of the implicit im2col matrix separately: 512 bytes of
// buffer->LDS DMA per MFMA, which at full matrix rate is the whole 64 B/clk/CU of the texture addresser.
// Here a workgroup owns a 2-D tile of 256 output pixels (16x16 or 32 wide x 8 high) x BN channels and keeps, per 32-channel
// chunk of the input, the (TH+2) x (TW+2) pixel patch in LDS ONCE; the nine taps are nine K-steps that read the SAME patch at
// shifted addresses.  Per K-step only the BN x 32 weight panel (and 1/9 of the next patch) comes in:
//        DMA bytes per MFMA:  512 -> 208 (BN=128) / 256 (BN=64);   ds_read_b128 per MFMA: 1 -> 0.75 (BN=128)
// and the K loop contains no address arithmetic at all:
//   * LDS rows (one pixel's 32 channels, 64 B) are PADDED to 80 B instead of XOR-swizzled, so every fragment address is
//     lane_base + compile-time immediate (tap shift, tile row, k-substep all fold into the ds_read offset field).  80-byte rows
//     are conflict-free for ds_read_b128 (slot = 5*pixel + chunk mod 16 is a bijection over the 16 lanes of a service group);
//     for the 16-wide tile the patch-row pitch is a multiple of 256 B so that the two image rows of one MFMA tile interleave.
//   * DMA sources are per-lane byte offsets computed once per workgroup (image border and the 16 pad bytes of each row =
//     out-of-range offset -> hardware zero fill); chunk and tap advance through the instruction's SCALAR offset.
// K order inside the kernel: 32-channel chunk outermost, tap innermost (weights are addressed in place in either packing,
// Y7TConvArgs::korder).  Two chunks (18 K-steps) are unrolled so that ring positions are compile-time.
#include "y7t_common.h"
#include "y7t_conv_common.h"
#include <stdlib.h>

namespace {

constexpr unsigned kOOB = 0xFF000000u;   // voffset of a zero-filled lane: out of range with or without the (< 16 MiB) scalar offset

template <int TW, int TH, int BN>
struct PatchCfg {
    static constexpr int PIXB = 80;                                       // bytes per patch pixel / weight row in LDS (64 data + 16 pad)
    static constexpr int RP = (TW == 16) ? 1536 : (TW + 2) * PIXB;        // patch row pitch
    static constexpr int PATCH_DMA = ((TH + 2) * RP + 1023) / 1024;       // wave-wide 1 KiB DMAs per patch
    static constexpr int NPX = (PATCH_DMA + 3) / 4;                       // patch DMAs per wave
    static constexpr int PATCH_BYTES = NPX * 4 * 1024;                    // every wave issues NPX DMAs; the tail ones zero-fill
    static constexpr int W_DMA = (BN * PIXB + 1023) / 1024;               // 10 (BN=128) / 5 (BN=64)
    static constexpr int NWX = (W_DMA + 3) / 4;                           // weight DMAs per wave per K-step
    static constexpr int W_BYTES = NWX * 4 * 1024;
    static constexpr int W_OFF = 0, P_OFF = 2 * W_BYTES;                  // LDS map: W ring (2 stages) | patch A | patch B
    static constexpr int LDS_LOOP = P_OFF + 2 * PATCH_BYTES;
    static constexpr int OROW = BN * 2 + 16;
    static constexpr int LDS_EPI = 256 * OROW;
    static constexpr int LDS = LDS_LOOP > LDS_EPI ? LDS_LOOP : LDS_EPI;
    static constexpr int WN = BN / 64, WM = 4 / WN;                       // waves along channels / pixels
    static constexpr int TM = 8 / WM;                                     // 32-pixel MFMA tiles per wave (4 or 2)
    static constexpr int RPT = (TW == 16) ? 2 : 1;                        // image rows per MFMA tile
};

template <int TW, int TH, int BN>
__global__ void __launch_bounds__(256, 2) k_conv3x3_patch(const Y7TConvArgs p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using C = PatchCfg<TW, TH, BN>;
    static_assert(TW * TH == 256, "256 output pixels per workgroup");
    constexpr int PIXB = C::PIXB, RP = C::RP, TM = C::TM, RPT = C::RPT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave id in an SGPR: DMA destinations are scalar
    const int wn = (C::WN == 2) ? (wave >> 1) : 0, wm = (C::WN == 2) ? (wave & 1) : wave;
    const int l31 = lane & 31, hi32 = lane >> 5;

    // ---- tile decode: channel tiles fastest (workgroups sharing a patch run next to each other), XCD-contiguous ranges ----
    int bid = blockIdx.x;
    if (p.xcd_swizzle) {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int n_tiles_n = p.Cout_pad / BN;
    const int tile_n = bid % n_tiles_n;
    int pt = bid / n_tiles_n;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int txi = pt % tiles_x; pt /= tiles_x;
    const int tyi = pt % tiles_y;
    const int b = pt / tiles_y;
    const int h0 = tyi * TH, w0 = txi * TW, n0 = tile_n * BN;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.in, 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.w_bytes, 0x00020000);

    // ---- per-lane DMA sources (computed once) ----
    unsigned xoff[C::NPX], woff[C::NWX];
#pragma unroll
    for (int i = 0; i < C::NPX; ++i) {
        const int I = wave + 4 * i;
        const int byte = I * 1024 + lane * 16;
        const int r = byte / RP, rb = byte - r * RP;
        const int x = rb / PIXB, cs = (rb - x * PIXB) >> 4;
        const int gy = h0 + r - 1, gx = w0 + x - 1;
        const bool ok = I < C::PATCH_DMA && r < TH + 2 && x < TW + 2 && cs < 4 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        xoff[i] = ok ? (unsigned)(((((b * p.H + gy) * p.W + gx) * p.ldin + p.cin_off) + cs * 8) * 2) : kOOB;
    }
#pragma unroll
    for (int i = 0; i < C::NWX; ++i) {
        const int I = wave + 4 * i;
        const int byte = I * 1024 + lane * 16;
        const int row = byte / PIXB, cs = (byte - row * PIXB) >> 4;
        const bool ok = I < C::W_DMA && row < BN && cs < 4;
        woff[i] = ok ? (unsigned)(((n0 + row) * p.K_pad + cs * 8) * 2) : kOOB;
    }

    // ---- fragment read bases ----
    const char* wlane = smem + C::W_OFF + (wn * 64 + l31) * PIXB + hi32 * 16;
    const char* plane = smem + C::P_OFF + (wm * TM * RPT + (TW == 16 ? (l31 >> 4) : 0)) * RP + (TW == 16 ? (l31 & 15) : l31) * PIXB + hi32 * 16;

    const int nc32 = p.Cin >> 5;
    // byte offset (in a weight row) of K-step (chunk c, tap kh,kw) = kh * s_kh + kw * s_kw + chunk part; the chunk part of an even
    // chunk c0 is `cbase`, of c0 + 1 it is cbase + 64 in both packings, and a chunk pair advances it by s_pair
    const int s_kh = p.korder ? (p.Cin >> 6) * 3 * 128 : 3 * p.Cin * 2;
    const int s_kw = p.korder ? 128 : p.Cin * 2;
    const int s_pair = p.korder ? 3 * 128 : 128;
    int cbase = 0;
    auto issue_w = [&](int stage, int coff, int kh, int kw) {
        const int so = kh * s_kh + kw * s_kw + coff;
#pragma unroll
        for (int i = 0; i < C::NWX; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (LDS_AS void*)(smem + C::W_OFF + stage * C::W_BYTES + (wave + 4 * i) * 1024), 16, woff[i], so, 0, 0);
    };
    auto issue_patch_piece = [&](int pb, int c, int i, bool real) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (LDS_AS void*)(smem + C::P_OFF + pb * C::PATCH_BYTES + (wave + 4 * i) * 1024), 16,
                                                 real ? xoff[i] : kOOB, c << 6, 0, 0);
    };

    floatx16 acc[2][TM];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // prologue: patch of chunk 0 and the weights of K-step 0
#pragma unroll
    for (int i = 0; i < C::NPX; ++i) issue_patch_piece(0, 0, i, true);
    issue_w(0, 0, 0, 0);

    static_assert(C::NPX <= 8, "one patch piece per tap, taps 0..7");
    static_assert(C::LDS_LOOP <= 81920, "two workgroups per CU");
    for (int c0 = 0; c0 < nc32; c0 += 2) {
#pragma unroll
        for (int u = 0; u < 18; ++u) {
            const int cc = u / 9, t = u % 9, kh = t / 3, kw = t % 3;   // compile-time after unrolling
            const int c = c0 + cc;
            const int stage = u & 1, pb = cc;
            // everything older than the (at most one) patch piece issued behind this step's weights has landed
            const bool piece_prev = (u > 0) ? (((u - 1) % 9) < C::NPX) : false;   // step u-1 issued a piece after W(u)
            if (piece_prev) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // next K-step's weights, then one piece of the next chunk's patch
            {
                const int un = u + 1, ccn = (un / 9) & 1, tn = un % 9;
                const int cn = (un == 18) ? c0 + 2 : c0 + ccn;
                if (cn < nc32) issue_w(stage ^ 1, (un == 18) ? cbase + s_pair : cbase + ccn * 64, tn / 3, tn % 3);
                if (t < C::NPX) issue_patch_piece(pb ^ 1, c + 1, t, c + 1 < nc32);
            }
            const char* ws = wlane + stage * C::W_BYTES;
            const char* ps = plane + pb * C::PATCH_BYTES + kh * RP + kw * PIXB;
            half8 wf[2][2], xf[2][TM];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int i = 0; i < 2; ++i) wf[ks][i] = *(const half8*)(ws + i * 32 * PIXB + ks * 32);
#pragma unroll
                for (int j = 0; j < TM; ++j) xf[ks][j] = *(const half8*)(ps + j * RPT * RP + ks * 32);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ks][i], xf[ks][j], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_s_setprio(0);
            }
        }
        cbase += s_pair;
    }

    // ---- epilogue: bias + activation, transpose through LDS, full-line NHWC stores ----
    constexpr int OROW = C::OROW;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TM; ++j) {
        const int pix = (wm * TM + j) * 32 + l31;   // tile-local pixel id: TW=16 -> row = pix >> 4, x = pix & 15; TW=32 -> row = pix >> 5
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int nl = wn * 64 + i * 32;
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                unsigned w[2][2];
#pragma unroll
                for (int gg = 0; gg < 2; ++gg) {
                    const int g = gp * 2 + gg;
                    const int n = n0 + nl + 8 * g + 4 * hi32;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = act_fn(acc[i][j][g * 4 + e] + p.bias[n + e], p.act);
                    typedef __attribute__((ext_vector_type(2))) _Float16 half2v;
                    half2v h0 = {(half_t)v[0], (half_t)v[1]}, h1 = {(half_t)v[2], (half_t)v[3]};
                    w[gg][0] = __builtin_bit_cast(unsigned, h0);
                    w[gg][1] = __builtin_bit_cast(unsigned, h1);
                }
                auto r0 = __builtin_amdgcn_permlane32_swap(w[0][0], w[1][0], false, false);
                auto r1 = __builtin_amdgcn_permlane32_swap(w[0][1], w[1][1], false, false);
                typedef __attribute__((ext_vector_type(4))) unsigned uint4v;
                uint4v pk = {r0[0], r1[0], r0[1], r1[1]};
                *(uint4v*)(smem + pix * OROW + (nl + 8 * (gp * 2 + hi32)) * 2) = pk;
            }
        }
    }
    __syncthreads();
    {
        typedef __attribute__((ext_vector_type(4))) unsigned uint4v;
        constexpr int CPP = BN / 8;
        half_t* outp = (half_t*)p.out;
#pragma unroll 4
        for (int cidx = tid; cidx < 256 * CPP; cidx += 256) {
            const int pix = cidx / CPP, ch = cidx - pix * CPP;
            const int r = pix / TW, x = pix - r * TW;
            const int gy = h0 + r, gx = w0 + x, n = n0 + ch * 8;
            if (gy < p.H && gx < p.W && n < p.Cout) {
                const uint4v v = *(const uint4v*)(smem + pix * OROW + ch * 16);
                *(uint4v*)(outp + ((size_t)(b * p.H + gy) * p.W + gx) * p.ldout + p.cout_off + n) = v;
            }
        }
    }
#endif
}

template <int TW, int TH, int BN>
int launch_patch(const Y7TConvArgs& a, hipStream_t s) {
    using C = PatchCfg<TW, TH, BN>;
    static bool attr = false;
    if (!attr) {
        Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)k_conv3x3_patch<TW, TH, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS));
        attr = true;
    }
    const int tiles = a.B * ((a.H + TH - 1) / TH) * ((a.W + TW - 1) / TW) * (a.Cout_pad / BN);
    hipLaunchKernelGGL((k_conv3x3_patch<TW, TH, BN>), dim3(tiles), dim3(256), C::LDS, s, a);
    Y7T_LAUNCH_CHECK();
    return 0;
}

}   // namespace

// 1 if the layer was launched on the patch kernel, 0 if it is not eligible (caller falls back to k_conv_igemm), < 0 on error
int y7t_conv_patch_try(const Y7TConvArgs& a, hipStream_t s) {
    static int mode = -1;
    if (mode < 0) { const char* e = getenv("Y7T_CONV_PATCH"); mode = e ? atoi(e) : 1; }
    if (!mode) return 0;
    if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.Cin % 64 || a.out_f32 || (a.Cout & 7) || (a.ldout & 7) || (a.cout_off & 7)) return 0;
    if (a.Ho != a.H || a.Wo != a.W || a.in_bytes > kOOB - (1u << 24) || a.ablate) return 0;
    // tile shape: the one that wastes fewer computed pixels; below 80 % useful pixels the linear-M kernel wins
    auto eff = [&](int tw, int th) {
        const int tx = (a.W + tw - 1) / tw, ty = (a.H + th - 1) / th;
        return (double)(a.W * a.H) / ((double)tx * tw * ty * th);
    };
    const double e16 = eff(16, 16), e32 = eff(32, 8);
    const bool use16 = e16 >= e32;
    if (!a.force_patch && (use16 ? e16 : e32) < 0.8) return 0;
    const bool wide = a.Cout_pad % 128 == 0;
    int rc;
    if (use16) rc = wide ? launch_patch<16, 16, 128>(a, s) : launch_patch<16, 16, 64>(a, s);
    else rc = wide ? launch_patch<32, 8, 128>(a, s) : launch_patch<32, 8, 64>(a, s);
    return rc ? rc : 1;
}
