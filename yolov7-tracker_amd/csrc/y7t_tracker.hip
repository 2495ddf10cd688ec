// y7t_tracker.hip -- gfx950 kernels + C ABI for the tracker half of the hot path
// (Kalman predict/update, IoU cost, linear assignment, fused ByteTrack/SORT frame step).
// The arithmetic lives in y7t_track_core.h / y7t_track_step.h (reference citations there).
//
// Roofline notes: all of these are HBM/latency bound, none is a
// contraction -> no MFMA.  Kalman predict moves 1152 B/track, IoU 32(N+M) B in + 8NM B out,
// the fused step keeps the LAP work arrays and (when it fits) the cost matrix in LDS.
#include "y7t_common.h"
#include "y7t_track_step.h"
#include "y7t_track_deepsort.h"
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <unordered_map>

static thread_local char g_err[512] = "";
void y7t_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* y7t_last_error(void) { return g_err; }
// diagnostics: name of the kernel the last launch helper of this thread chose (conv dispatch rules live on the host side of the
// library, so "which kernel runs this layer at this batch" is a property tests and bench.py can read back)
static thread_local char g_kernel[128] = "";
void y7t_note_kernel(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
    va_end(ap);
}
extern "C" const char* y7t_last_kernel(void) { return g_kernel; }
extern "C" int y7t_version(void) { return 100; }
extern "C" int y7t_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

// ---------------------------------------------------------------------------------------------
// per-op kernels
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_iou_cost(const double* __restrict__ a, int n, const double* __restrict__ b, int m,
                                                  double* __restrict__ cost) {
    // one thread per (i, j); j fastest -> coalesced 8-byte stores, box rows broadcast through L1
    const long long tot = (long long)n * m;
    for (long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x; k < tot; k += (long long)gridDim.x * blockDim.x) {
        const int i = (int)(k / m), j = (int)(k - (long long)i * m);
        double bi[4], bj[4];
        for (int c = 0; c < 4; ++c) { bi[c] = a[4 * (size_t)i + c]; bj[c] = b[4 * (size_t)j + c]; }
        cost[k] = y7t_iou_dist(bi, bj);
    }
}

__global__ void __launch_bounds__(64) k_kf_initiate(int kind, const double* __restrict__ z, double* mean, double* cov, int K,
                                                    int flags) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    double zz[4], m[8], P[64];
    for (int c = 0; c < 4; ++c) zz[c] = z[4 * (size_t)k + c];
    y7t_kf_initiate(kind, zz, flags & 1, m, P);
    for (int c = 0; c < 8; ++c) mean[8 * (size_t)k + c] = m[c];
    for (int c = 0; c < 64; ++c) cov[64 * (size_t)k + c] = P[c];
}

__global__ void __launch_bounds__(64) k_kf_predict(int kind, double* mean, double* cov, const uint8_t* __restrict__ mask, int N) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    double m[8], P[64];
    for (int c = 0; c < 8; ++c) m[c] = mean[8 * (size_t)k + c];
    for (int c = 0; c < 64; ++c) P[c] = cov[64 * (size_t)k + c];
    if (mask && mask[k]) m[7] = 0.0;
    y7t_kf_predict(kind, m, P);
    for (int c = 0; c < 8; ++c) mean[8 * (size_t)k + c] = m[c];
    for (int c = 0; c < 64; ++c) cov[64 * (size_t)k + c] = P[c];
}

__global__ void __launch_bounds__(64) k_kf_project(int kind, const double* __restrict__ mean, const double* __restrict__ cov,
                                                   const double* __restrict__ conf, double* pm, double* pc, int N) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    double m[8], P[64], a[4], S[16];
    for (int c = 0; c < 8; ++c) m[c] = mean[8 * (size_t)k + c];
    for (int c = 0; c < 64; ++c) P[c] = cov[64 * (size_t)k + c];
    y7t_kf_project(kind, m, P, conf ? conf[k] : 0.0, a, S);
    for (int c = 0; c < 4; ++c) pm[4 * (size_t)k + c] = a[c];
    for (int c = 0; c < 16; ++c) pc[16 * (size_t)k + c] = S[c];
}

__global__ void __launch_bounds__(64) k_kf_update(int kind, double* mean, double* cov, const double* __restrict__ z,
                                                  const int* __restrict__ idx, const double* __restrict__ conf, int K) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    const int t = idx ? idx[k] : k;
    double m[8], P[64], zz[4];
    for (int c = 0; c < 8; ++c) m[c] = mean[8 * (size_t)t + c];
    for (int c = 0; c < 64; ++c) P[c] = cov[64 * (size_t)t + c];
    for (int c = 0; c < 4; ++c) zz[c] = z[4 * (size_t)k + c];
    y7t_kf_update(kind, m, P, zz, conf ? conf[k] : 0.0);
    for (int c = 0; c < 8; ++c) mean[8 * (size_t)t + c] = m[c];
    for (int c = 0; c < 64; ++c) cov[64 * (size_t)t + c] = P[c];
}

__global__ void __launch_bounds__(64) k_kf_gating(int kind, const double* __restrict__ mean, const double* __restrict__ cov,
                                                  const double* __restrict__ z, int N, int M, int only_pos, double* out) {
    const long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    if (k >= (long long)N * M) return;
    const int i = (int)(k / M), j = (int)(k - (long long)i * M);
    double m[8], P[64], zz[4];
    for (int c = 0; c < 8; ++c) m[c] = mean[8 * (size_t)i + c];
    for (int c = 0; c < 64; ++c) P[c] = cov[64 * (size_t)i + c];
    for (int c = 0; c < 4; ++c) zz[c] = z[4 * (size_t)j + c];
    out[k] = y7t_kf_gating(kind, m, P, zz, only_pos);
}

// ---------------------------------------------------------------------------------------------
// single-workgroup programs: LAPJV and the fused tracker step.  Dynamic LDS layout:
//   [0, 512)  rv  (64 doubles)   [512, 768) ri (64 ints)   [1024, ...) fast scratch
// ---------------------------------------------------------------------------------------------
#define Y7T_LDS_HDR 1024
extern __shared__ __attribute__((aligned(16))) char y7t_smem[];

__device__ __forceinline__ Y7TExec make_exec(unsigned fast_bytes) {
    Y7TExec ex;
    ex.tid = threadIdx.x; ex.nt = blockDim.x;
    ex.rv = (double*)y7t_smem; ex.ri = (int*)(y7t_smem + 512);
    ex.fast = fast_bytes ? y7t_smem + Y7T_LDS_HDR : nullptr;
    ex.fast_bytes = fast_bytes;
    ex.arena = nullptr; ex.arena_bytes = 0;
    return ex;
}

__global__ void k_lapjv(const double* __restrict__ cost, int nr, int nc, double limit, int* x, int* y, double* opt, void* ws_g,
                        unsigned fast_bytes, int jv_extended) {
    const Y7TExec ex = make_exec(fast_bytes);
    Y7TLap L;
    L.nr = nr; L.nc = nc; L.ld = nc; L.n = nr + nc; L.half = limit / 2.0; L.prof = nullptr;
    const size_t ws = y7t_al(y7t_lap_ws_bytes(L.n)), cb = (size_t)nr * nc * sizeof(double);
    void* lapws = ws_g;
    size_t off = 0;
    if (ws <= fast_bytes) { lapws = ex.fast; off = ws; }
    const double* c = cost;
    if (ex.fast && off + cb <= fast_bytes) {  // stage the cost matrix in LDS
        double* cl = (double*)(ex.fast + off);
        for (int k = ex.tid; k < nr * nc; k += ex.nt) cl[k] = cost[k];
        c = cl;
    }
    __syncthreads();
    L.c = c;
    y7t_lap_bind(L, lapws, L.n);
    if (jv_extended) y7t_lap_solve(ex, L);
    else if (y7t_lap_solve_sap(ex, L)) y7t_lap_solve_literal(ex, L);      // ties: lapjv's own order decides
    for (int i = ex.tid; i < nr; i += ex.nt) x[i] = (L.x[i] >= nc) ? -1 : L.x[i];
    for (int j = ex.tid; j < nc; j += ex.nt) y[j] = (L.y[j] >= nr) ? -1 : L.y[j];
    if (opt && ex.tid == 0) {
        double s = 0.0;
        for (int i = 0; i < nr; ++i) if (L.x[i] < nc) s += cost[(size_t)i * nc + L.x[i]];
        *opt = s;
    }
}

__global__ void k_tracker_init(void* blob, Y7TTrkCfg cfg, unsigned long long idc) {
    Y7TExec ex;
    ex.tid = threadIdx.x; ex.nt = blockDim.x; ex.rv = nullptr; ex.ri = nullptr; ex.fast = nullptr; ex.fast_bytes = 0; ex.arena = nullptr; ex.arena_bytes = 0;
    y7t_tracker_init(ex, blob, cfg, idc);
}

template <int MAXT>      // (the thread bound of the launch: see k_tracker_step1)
__global__ void __launch_bounds__(MAXT) k_tracker_step(void* const* states, const float* const* dets, const int* n_dets, double* const* out_rows,
                                                        int* out_count, int out_cap, unsigned fast_bytes, const double* const* warps) {
    const int b = blockIdx.x;
    const Y7TExec ex = make_exec(fast_bytes);
    if (MAXT <= 512) y7t_tracker_step_body(ex, states[b], dets[b], n_dets[b], out_rows[b], out_cap, out_count + b, warps ? warps[b] : nullptr);
    else y7t_tracker_step(ex, states[b], dets[b], n_dets[b], out_rows[b], out_cap, out_count + b, warps ? warps[b] : nullptr);
}

// MAXT: the launch's thread bound.  Compiled for 1024 threads a lane has 128 registers and the frame step spills what it keeps live around its Kalman updates to SCRATCH
// memory; the <= 256-thread launches (scenes of up to ~384 objects: step_threads) run an instance compiled for 256 threads with the step's body inlined (round 6)
template <int MAXT>
__global__ void __launch_bounds__(MAXT) k_tracker_step1(void* state, const float* dets, int n, double* out_rows, int out_cap, int* out_count,
                                                         unsigned fast_bytes, const double* warp) {
    const Y7TExec ex = make_exec(fast_bytes);
    if (MAXT <= 512) y7t_tracker_step_body(ex, state, dets, n, out_rows, out_cap, out_count, warp);      // inlined: under this kernel's register budget
    else y7t_tracker_step(ex, state, dets, n, out_rows, out_cap, out_count, warp);
}

// n_frames consecutive frames of ONE tracker in one launch: the same frame step, frame after frame, by the same workgroup (the state stays hot in
// this CU's caches and nothing is launched between frames).  For pipelines that have a whole batch's detections before the tracker runs.
template <int MAXT>
__global__ void __launch_bounds__(MAXT) k_tracker_step_frames(void* state, const float* const* dets, const int* n_dets, double* const* out_rows, int* const* out_count, int out_cap,
                                                               int n_frames, unsigned fast_bytes, unsigned arena_bytes, const double* const* warps) {
    Y7TExec ex = make_exec(fast_bytes);
    if (arena_bytes) { ex.arena = y7t_smem + Y7T_LDS_HDR + fast_bytes; ex.arena_bytes = arena_bytes; }      // the index lists live in LDS for the whole launch
    y7t_arena_load(ex, state);
    for (int f = 0; f < n_frames; ++f) {
        if (MAXT <= 512) y7t_tracker_step_body(ex, state, dets[f], n_dets[f], out_rows[f], out_cap, out_count[f], warps ? warps[f] : nullptr);
        else y7t_tracker_step(ex, state, dets[f], n_dets[f], out_rows[f], out_cap, out_count[f], warps ? warps[f] : nullptr);
        y7t_sync(ex);
    }
    y7t_arena_store(ex, state);
}

// ---- DeepSORT (y7t_track_deepsort.h) ----
__global__ void k_feat_init(void* fblob, int cap_t, int cap_d, int dim, int budget) {
    Y7TExec ex;
    ex.tid = threadIdx.x; ex.nt = blockDim.x; ex.rv = nullptr; ex.ri = nullptr; ex.fast = nullptr; ex.fast_bytes = 0; ex.arena = nullptr; ex.arena_bytes = 0;
    y7t_feat_init(ex, fblob, cap_t, cap_d, dim, budget);
}

// detection features of the frame -> normalised rows (a wave per detection)
__global__ void __launch_bounds__(256) k_ds_normalize(void* fblob, const float* __restrict__ det_feats, int n) {
    const Y7TFeat f = y7t_feat_bind(fblob);
    Y7TExec ex;
    ex.tid = blockIdx.x * blockDim.x + threadIdx.x; ex.nt = gridDim.x * blockDim.x; ex.rv = nullptr; ex.ri = nullptr; ex.fast = nullptr; ex.fast_bytes = 0; ex.arena = nullptr; ex.arena_bytes = 0;
    if (n > f.h->cap_d) n = f.h->cap_d;      // (the step reports Y7T_ERR_CAP_D; nothing is written past the state)
    y7t_feat_normalize_dets(ex, f, det_feats, n);
}

// nearest-embedding distances, tiled: a workgroup takes (slot, 64 detections), the slot's stored rows 32 at a time, 32-deep k chunks of both operands in LDS,
// 2 x 4 products per thread.  Every product is the sequential FMA chain over k = 0 .. dim-1 of y7t_embed_slot (= numpy's sgemm), so the costs -- and
// the assignment that follows them -- are bit-identical; only the order in which DIFFERENT products advance changes.
// Round 6: the grid is (Y7T_EMBED_WORKERS, detection tiles) instead of (cap_tracks, detection tiles).  Every workgroup walks the state column once, ranks the live
// slots (Tracked / Lost) in slot order and takes those whose rank is its blockIdx.x modulo the worker count: a frame with 64 live tracks of a 512-slot pool launched
// 1024 workgroups of which ~900 found nothing to do -- beside the detector each of them queued for a CU slot first (the kernel ran 128 us per frame in the cfg4
// pipeline against 38 us alone, profiles/r06_batch_80.txt section 4).
#define Y7T_EMBED_WORKERS 128
__global__ void __launch_bounds__(256) k_embed_dist(void* blob, void* fblob, int n) {
    const Y7TTrkHdr* h = (const Y7TTrkHdr*)blob;
    const Y7TTrk s = y7t_trk_bind(blob, h->cfg.cap_t, h->cfg.cap_d);
    const int j0 = blockIdx.y * 64, cap_t = h->cfg.cap_t;
    if (n > h->cfg.cap_d) n = h->cfg.cap_d;
    const Y7TFeat f = y7t_feat_bind(fblob);
    const int dim = f.h->dim, tid = threadIdx.x;
    __shared__ float sA[32][33], sB[64][33], smin[16][64];
    __shared__ int s_cnt[4], s_own[64], s_nown;
    // ---- the live slots of rank = blockIdx.x (mod gridDim.x), in slot order ----
    if (tid == 0) s_nown = 0;
    int seen = 0;      // live slots below `base` (uniform)
    for (int base = 0; base < cap_t; base += 256) {
        const int sl = base + tid;
        bool live = false;
        if (sl < cap_t) { const int st = s.state[sl]; live = st == Y7T_TRACKED || st == Y7T_LOST; }
        const unsigned long long bal = __ballot(live);
        __syncthreads();                                   // (s_cnt of the previous round has been read; s_nown's initial store is visible)
        if ((tid & 63) == 0) s_cnt[tid >> 6] = __popcll(bal);
        __syncthreads();
        int before = seen;
        for (int w = 0; w < (tid >> 6); ++w) before += s_cnt[w];
        const int rank = before + __popcll(bal & ((1ull << (tid & 63)) - 1ull));
        if (live && rank % (int)gridDim.x == (int)blockIdx.x) { const int k = rank / (int)gridDim.x; if (k < 64) s_own[k] = sl; atomicMax(&s_nown, k + 1); }
        seen += s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
    }
    __syncthreads();
    const int n_own = s_nown < 64 ? s_nown : 64;      // (64 x the worker count live tracks: more than any pool this library is given; the launcher checks cap_tracks)
    for (int oi = 0; oi < n_own; ++oi) {
        const int slot = s_own[oi];
        const int nf = f.nfeat[slot];
        if (dim & 31) {      // feature dimensions that are not a multiple of the k chunk: the plain form, one workgroup per slot
            if (blockIdx.y == 0) {
                Y7TExec ex;
                ex.tid = threadIdx.x; ex.nt = blockDim.x; ex.rv = nullptr; ex.ri = nullptr; ex.fast = nullptr; ex.fast_bytes = 0; ex.arena = nullptr; ex.arena_bytes = 0;
                y7t_embed_slot(ex, f, slot, n, s.tsu[slot]);
            }
            continue;
        }
        if (nf <= 0 || j0 >= n) continue;
        const int ty = tid >> 4, tx = tid & 15, lrow = tid >> 3, kq = (tid & 7) * 4;
        const float* hist = f.ring + (size_t)slot * f.h->budget * dim;
        float best[4] = {3.0e38f, 3.0e38f, 3.0e38f, 3.0e38f};
        // (round 6: the NEXT k chunk's global loads are issued before the current chunk's products -- beside the detector a load takes several times what it takes alone,
        //  and the loop used to wait for every chunk with nothing else to do)
        auto fetch = [&](int h0, int k0, float4& va, float4& vb0, float4& vb1) {
            va = make_float4(0.f, 0.f, 0.f, 0.f); vb0 = va; vb1 = va;
            if (h0 + lrow < nf) va = *(const float4*)(hist + (size_t)(h0 + lrow) * dim + k0 + kq);
            if (j0 + lrow < n) vb0 = *(const float4*)(f.detn + (size_t)(j0 + lrow) * dim + k0 + kq);
            if (j0 + 32 + lrow < n) vb1 = *(const float4*)(f.detn + (size_t)(j0 + 32 + lrow) * dim + k0 + kq);
        };
        float4 va, vb0, vb1;
        fetch(0, 0, va, vb0, vb1);
        for (int h0 = 0; h0 < nf; h0 += 32) {
            float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            for (int k0 = 0; k0 < dim; k0 += 32) {
                __syncthreads();
                sA[lrow][kq] = va.x; sA[lrow][kq + 1] = va.y; sA[lrow][kq + 2] = va.z; sA[lrow][kq + 3] = va.w;
                sB[lrow][kq] = vb0.x; sB[lrow][kq + 1] = vb0.y; sB[lrow][kq + 2] = vb0.z; sB[lrow][kq + 3] = vb0.w;
                sB[32 + lrow][kq] = vb1.x; sB[32 + lrow][kq + 1] = vb1.y; sB[32 + lrow][kq + 2] = vb1.z; sB[32 + lrow][kq + 3] = vb1.w;
                __syncthreads();
                if (k0 + 32 < dim) fetch(h0, k0 + 32, va, vb0, vb1);
                else if (h0 + 32 < nf) fetch(h0 + 32, 0, va, vb0, vb1);
#pragma unroll 8
                for (int k = 0; k < 32; ++k) {
                    const float a0 = sA[2 * ty][k], a1 = sA[2 * ty + 1][k];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const float b = sB[tx + 16 * c][k];
                        acc[0][c] = __builtin_fmaf(a0, b, acc[0][c]);
                        acc[1][c] = __builtin_fmaf(a1, b, acc[1][c]);
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 2; ++r)
                if (h0 + 2 * ty + r < nf)
#pragma unroll
                    for (int c = 0; c < 4; ++c) { const float d = 1.0f - acc[r][c]; best[c] = d < best[c] ? d : best[c]; }
        }
        __syncthreads();                                   // (the previous slot's reduction has read smin)
#pragma unroll
        for (int c = 0; c < 4; ++c) smin[ty][tx + 16 * c] = best[c];
        __syncthreads();
        if (tid < 64 && j0 + tid < n) {
            float m = smin[0][tid];
            for (int q = 1; q < 16; ++q) m = smin[q][tid] < m ? smin[q][tid] : m;
            f.app[(size_t)slot * f.h->cap_d + j0 + tid] = m;
            if ((double)m <= 0.15) { const int a = s.tsu[slot]; atomicMin(f.cmin + j0 + tid, a); atomicMax(f.cmax + j0 + tid, a); }
        }
    }
}

// the appearance vectors the step queued -> the rings (a wave per vector, across the grid)
__global__ void __launch_bounds__(256) k_ds_store(void* fblob, const float* __restrict__ det_feats) {
    const Y7TFeat f = y7t_feat_bind(fblob);
    Y7TExec ex;
    ex.tid = blockIdx.x * blockDim.x + threadIdx.x; ex.nt = gridDim.x * blockDim.x; ex.rv = nullptr; ex.ri = nullptr; ex.fast = nullptr; ex.fast_bytes = 0; ex.arena = nullptr; ex.arena_bytes = 0;
    y7t_feat_store_pending(ex, f, det_feats);
}

template <int MAXT>
__global__ void __launch_bounds__(MAXT) k_tracker_step_deepsort(void* state, void* fblob, const float* dets, int n, const float* det_feats, double* out_rows,
                                                                 int out_cap, int* out_count, unsigned fast_bytes) {
    const Y7TExec ex = make_exec(fast_bytes);
    y7t_tracker_step_deepsort(ex, state, fblob, dets, n, det_feats, out_rows, out_cap, out_count);
}

__global__ void __launch_bounds__(64) k_kf_gmc(double* mean, double* cov, const double* __restrict__ warp, int N) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= N) return;
    double m[8], P[64], H[6];
    for (int c = 0; c < 6; ++c) H[c] = warp[c];
    for (int c = 0; c < 8; ++c) m[c] = mean[8 * (size_t)k + c];
    for (int c = 0; c < 64; ++c) P[c] = cov[64 * (size_t)k + c];
    y7t_kf_gmc(y7t_warp_load(H), m, P);
    for (int c = 0; c < 8; ++c) mean[8 * (size_t)k + c] = m[c];
    for (int c = 0; c < 64; ++c) cov[64 * (size_t)k + c] = P[c];
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
static inline hipStream_t S(y7t_stream s) { return (hipStream_t)s; }
static const unsigned kFastBytes = 128 * 1024;  // fast scratch per workgroup (of the CU's 160 KiB LDS)
// Fast scratch of a frame step.  Asking for less than the maximum on small frames (so that the step's workgroup can share a CU with the detector's
// workgroups instead of waiting for one to drain) was measured: no gain in the step's start latency, and the association's cost matrix / candidate
// lists falling back to the state blob made the in-pipeline chain 12 % slower (14.1 -> 15.8 ms per 32 frames).  Y7T_TRACKER_FAST_KB overrides.
static unsigned step_fast_bytes(int n_dets) {
    static int forced = -1;
    if (forced < 0) forced = y7t_exp_switch("Y7T_TRACKER_FAST_KB", 0) * 1024;
    if (forced > 0) return (unsigned)(forced < (int)kFastBytes ? forced : (int)kFastBytes);
    (void)n_dets;
    return kFastBytes;
}

// (Round 6 measured the step kernels asking for the CU's WHOLE 160 KiB so that no convolution workgroup shares their CU: cfg3 chain 26.95 / 26.91 ms with, 27.01 without;
// cfg4 35.02 / 35.01 vs 35.17; cfg2 11.21 / 11.27 vs 11.33 -- nothing, removed (profiles/r06_small_experiments.txt).  What slows the step beside a running detector is
// not a neighbour on its CU.)
template <class K>
static int ensure_lds(K kernel, unsigned bytes) {
    Y7T_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return 0;
}
// the attribute is per device and the callers may be threads of several trackers: done-bits per device, set after the call succeeded
template <class K>
static int ensure_lds_once(K kernel, unsigned bytes, std::atomic<unsigned long long>& done) {
    int dev = 0;
    Y7T_HIP_CHECK(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return 0;
    if (int e = ensure_lds(kernel, bytes)) return e;
    done.fetch_or(bit, std::memory_order_release);
    return 0;
}

extern "C" int y7t_iou_cost_f64(const double* a, int n, const double* b, int m, double* cost, y7t_stream stream) {
    Y7T_ARG_CHECK(n >= 0 && m >= 0);
    if (n == 0 || m == 0) return 0;
    Y7T_ARG_CHECK(a && b && cost);
    const long long tot = (long long)n * m;
    int blocks = (int)((tot + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_iou_cost, dim3(blocks), dim3(256), 0, S(stream), a, n, b, m, cost);
    Y7T_LAUNCH_CHECK();
    return 0;
}

static int kind_ok(int kind) { return kind == Y7T_KF_XYAH || kind == Y7T_KF_XYWH || kind == Y7T_KF_NSA; }

extern "C" int y7t_kf_initiate_f64(int kind, const double* z, double* mean, double* cov, int K, int flags, y7t_stream stream) {
    Y7T_ARG_CHECK(K >= 0);
    if (!kind_ok(kind)) { y7t_set_error("kalman kind %d is not implemented on the device (default/botsort/strongsort are)", kind); return Y7T_E_ARG; }
    if (K == 0) return 0;
    Y7T_ARG_CHECK(z && mean && cov);
    hipLaunchKernelGGL(k_kf_initiate, dim3((K + 63) / 64), dim3(64), 0, S(stream), kind, z, mean, cov, K, flags);
    Y7T_LAUNCH_CHECK();
    return 0;
}

extern "C" int y7t_kf_multi_predict_f64(int kind, double* mean, double* cov, const uint8_t* mask, int N, y7t_stream stream) {
    Y7T_ARG_CHECK(N >= 0);
    if (!kind_ok(kind)) { y7t_set_error("kalman kind %d is not implemented on the device", kind); return Y7T_E_ARG; }
    if (N == 0) return 0;
    Y7T_ARG_CHECK(mean && cov);
    hipLaunchKernelGGL(k_kf_predict, dim3((N + 63) / 64), dim3(64), 0, S(stream), kind, mean, cov, mask, N);
    Y7T_LAUNCH_CHECK();
    return 0;
}

extern "C" int y7t_kf_project_f64(int kind, const double* mean, const double* cov, const double* conf, double* pmean,
                                  double* pcov, int N, y7t_stream stream) {
    Y7T_ARG_CHECK(N >= 0);
    if (!kind_ok(kind)) { y7t_set_error("kalman kind %d is not implemented on the device", kind); return Y7T_E_ARG; }
    if (N == 0) return 0;
    Y7T_ARG_CHECK(mean && cov && pmean && pcov);
    hipLaunchKernelGGL(k_kf_project, dim3((N + 63) / 64), dim3(64), 0, S(stream), kind, mean, cov, conf, pmean, pcov, N);
    Y7T_LAUNCH_CHECK();
    return 0;
}

extern "C" int y7t_kf_update_batch_f64(int kind, double* mean, double* cov, const double* z, const int* track_idx,
                                       const double* conf, int K, y7t_stream stream) {
    Y7T_ARG_CHECK(K >= 0);
    if (!kind_ok(kind)) { y7t_set_error("kalman kind %d is not implemented on the device", kind); return Y7T_E_ARG; }
    if (K == 0) return 0;
    Y7T_ARG_CHECK(mean && cov && z);
    hipLaunchKernelGGL(k_kf_update, dim3((K + 63) / 64), dim3(64), 0, S(stream), kind, mean, cov, z, track_idx, conf, K);
    Y7T_LAUNCH_CHECK();
    return 0;
}

extern "C" int y7t_kf_gating_f64(int kind, const double* mean, const double* cov, const double* z, int N, int M,
                                 int only_position, double* out, y7t_stream stream) {
    Y7T_ARG_CHECK(N >= 0 && M >= 0);
    if (!kind_ok(kind)) { y7t_set_error("kalman kind %d is not implemented on the device", kind); return Y7T_E_ARG; }
    if (N == 0 || M == 0) return 0;
    Y7T_ARG_CHECK(mean && cov && z && out);
    const long long tot = (long long)N * M;
    hipLaunchKernelGGL(k_kf_gating, dim3((unsigned)((tot + 63) / 64)), dim3(64), 0, S(stream), kind, mean, cov, z, N, M,
                       only_position, out);
    Y7T_LAUNCH_CHECK();
    return 0;
}

extern "C" int y7t_lap_literal_calls(void) {
    int v = -1;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(y7t_g_literal_calls), sizeof(int)) != hipSuccess) return -1;
    return v;
}

extern "C" size_t y7t_lapjv_workspace_bytes(int n, int m) {
    if (n < 0 || m < 0) return 0;
    return y7t_al(y7t_lap_ws_bytes(n + m)) + 64;
}

extern "C" int y7t_lapjv_f64(const double* cost, int n, int m, double cost_limit, int* x, int* y, double* opt, void* workspace,
                             y7t_stream stream) {
    Y7T_ARG_CHECK(n >= 0 && m >= 0);
    if (n == 0 || m == 0) {
        // matching.py:31-32: an empty cost matrix leaves everything unmatched
        if (n) Y7T_HIP_CHECK(hipMemsetAsync(x, 0xff, sizeof(int) * (size_t)n, S(stream)));
        if (m) Y7T_HIP_CHECK(hipMemsetAsync(y, 0xff, sizeof(int) * (size_t)m, S(stream)));
        if (opt) Y7T_HIP_CHECK(hipMemsetAsync(opt, 0, sizeof(double), S(stream)));
        return 0;
    }
    Y7T_ARG_CHECK(cost && x && y && workspace);
    static std::atomic<unsigned long long> attr_done{0};
    if (int e = ensure_lds_once(k_lapjv, kFastBytes + Y7T_LDS_HDR, attr_done)) return e;
    const int nn = n + m;
    const int threads = nn <= 128 ? 64 : (nn <= 512 ? 256 : 1024);
    static int jv = -1;
    if (jv < 0) jv = y7t_exp_switch("Y7T_LAP_JV_EXTENDED", 0);
    hipLaunchKernelGGL(k_lapjv, dim3(1), dim3(threads), kFastBytes + Y7T_LDS_HDR, S(stream), cost, n, m, cost_limit, x, y, opt,
                       workspace, kFastBytes, jv);
    Y7T_LAUNCH_CHECK();
    return 0;
}

extern "C" int y7t_lapjv_f64_host(const double* cost_host, int n, int m, double cost_limit, int* x_host, int* y_host, double* opt_host, y7t_stream stream) {
    Y7T_ARG_CHECK(n >= 0 && m >= 0 && (n == 0 || x_host) && (m == 0 || y_host));
    if (n == 0 || m == 0) {
        for (int i = 0; i < n; ++i) x_host[i] = -1;
        for (int j = 0; j < m; ++j) y_host[j] = -1;
        if (opt_host) *opt_host = 0.0;
        return 0;
    }
    Y7T_ARG_CHECK(cost_host);
    static char* buf = nullptr;            // one staging allocation per process, grown on demand: cost | x | y | opt | solver workspace
    static size_t cap = 0;
    static std::mutex mu;                  // matching.linear_assignment routes every numpy call through here: one caller at a time owns the buffer
    std::lock_guard<std::mutex> lock(mu);  // (the call synchronises its stream before returning, so nothing of it is in flight when the lock drops)
    const size_t cb = y7t_al(sizeof(double) * (size_t)n * m), xb = y7t_al(sizeof(int) * (size_t)n), yb = y7t_al(sizeof(int) * (size_t)m), ob = y7t_al(sizeof(double));
    const size_t need = cb + xb + yb + ob + y7t_lapjv_workspace_bytes(n, m);
    if (need > cap) {
        if (buf) Y7T_HIP_CHECK(hipFree(buf));
        buf = nullptr; cap = 0;
        Y7T_HIP_CHECK(hipMalloc((void**)&buf, need));
        cap = need;
    }
    double* dc = (double*)buf; int* dx = (int*)(buf + cb); int* dy = (int*)(buf + cb + xb); double* dopt = (double*)(buf + cb + xb + yb);
    Y7T_HIP_CHECK(hipMemcpyAsync(dc, cost_host, sizeof(double) * (size_t)n * m, hipMemcpyHostToDevice, S(stream)));
    if (int e = y7t_lapjv_f64(dc, n, m, cost_limit, dx, dy, dopt, buf + cb + xb + yb + ob, stream)) return e;
    Y7T_HIP_CHECK(hipMemcpyAsync(x_host, dx, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, S(stream)));
    Y7T_HIP_CHECK(hipMemcpyAsync(y_host, dy, sizeof(int) * (size_t)m, hipMemcpyDeviceToHost, S(stream)));
    if (opt_host) Y7T_HIP_CHECK(hipMemcpyAsync(opt_host, dopt, sizeof(double), hipMemcpyDeviceToHost, S(stream)));
    Y7T_HIP_CHECK(hipStreamSynchronize(S(stream)));
    return 0;
}

// tracker kind of every initialised state blob (host side; the header itself lives in device memory): the plain frame step refuses a DeepSORT pool
// with detections -- it would create / update tracks without the appearance-ring bookkeeping, and a reused slot would keep its previous
// occupant's vectors.  Only the predict-only form (n < 0) is shared between the trackers.
static std::mutex g_kind_mu;
static std::unordered_map<const void*, int> g_state_kind;
static std::unordered_map<const void*, size_t> g_state_arena;      // bytes of LDS the pool's index lists take (y7t_arena_bytes of its capacities)
static void note_state_kind(const void* state, int kind, int cap_t = 0, int cap_d = 0) {
    std::lock_guard<std::mutex> l(g_kind_mu);
    g_state_kind[state] = kind;
    g_state_arena[state] = cap_t > 0 ? y7t_arena_bytes(cap_t, cap_d) : 0;
}
static size_t state_arena_bytes(const void* state) {
    std::lock_guard<std::mutex> l(g_kind_mu);
    auto it = g_state_arena.find(state);
    return it == g_state_arena.end() ? 0 : it->second;
}
static int state_kind(const void* state) {
    std::lock_guard<std::mutex> l(g_kind_mu);
    auto it = g_state_kind.find(state);
    return it == g_state_kind.end() ? -1 : it->second;
}

// the caller is about to free / reuse `state`: forget what the host side knows about it, so that a later blob at the same address is judged by its own
// y7t_tracker_init (the registry only ever grew, and a stale DEEPSORT entry made the plain step refuse a valid pool)
extern "C" int y7t_tracker_release(void* state) {
    std::lock_guard<std::mutex> l(g_kind_mu);
    g_state_kind.erase(state);
    g_state_arena.erase(state);
    return 0;
}

extern "C" size_t y7t_tracker_state_bytes(int cap_t, int cap_d) {
    if (cap_t <= 0 || cap_d <= 0) return 0;
    return y7t_trk_layout(cap_t, cap_d).total;
}

extern "C" int y7t_tracker_init(void* state, size_t state_bytes, int tracker_kind, int kalman_kind, int cap_t, int cap_d,
                                double conf_thresh, double iou_thresh, int max_time_lost, int flags, int* id_counter,
                                y7t_stream stream) {
    Y7T_ARG_CHECK(state && id_counter && cap_t > 0 && cap_d > 0);
    Y7T_ARG_CHECK(tracker_kind == Y7T_SORT || tracker_kind == Y7T_BYTETRACK || tracker_kind == Y7T_BOTSORT || tracker_kind == Y7T_DEEPSORT);
    if (tracker_kind == Y7T_DEEPSORT && kalman_kind == Y7T_KF_XYWH) {
        y7t_set_error("DeepSORT gates on xyah measurements (deepsort.py:59): kalman_format default / strongsort only");
        return Y7T_E_ARG;
    }
    if (!kind_ok(kalman_kind)) { y7t_set_error("kalman kind %d is not implemented on the device", kalman_kind); return Y7T_E_ARG; }
    Y7T_ARG_CHECK(state_bytes >= y7t_trk_layout(cap_t, cap_d).total);
    Y7TTrkCfg c;
    memset(&c, 0, sizeof(c));
    c.tracker = tracker_kind; c.kf = kalman_kind; c.cap_t = cap_t; c.cap_d = cap_d; c.max_time_lost = max_time_lost;
    c.f32_quirk = flags & 1;
    c.det_thresh = conf_thresh;
    c.low_thresh = (conf_thresh - 0.3 > 0.15) ? conf_thresh - 0.3 : 0.15;  // bytetrack.py:15
    c.iou_thresh = iou_thresh;
    hipLaunchKernelGGL(k_tracker_init, dim3(1), dim3(256), 0, S(stream), state, c, (unsigned long long)(uintptr_t)id_counter);
    Y7T_LAUNCH_CHECK();
    note_state_kind(state, tracker_kind, cap_t, cap_d);
    return 0;
}

static int step_threads(int threads, int n_hint = -1) {
    // measured on MI355X: four waves up to ~384 detections, sixteen beyond (500-object frames 971 us with four waves, 808 us with sixteen).  Round 1
    // chose ONE wave for ~100-object scenes from stand-alone timings (no cross-wave barriers in the LAP reductions); inside the pipeline, where the step
    // runs beside the detector's kernels, four waves hide the longer memory round trips: tracker chain 14.1 -> 12.1 ms per 32 frames on the same box
    // (bench.py --tracker_threads 0 / 256, profiles/r02_bench_variants.txt), and stand-alone they are no slower any more (179 vs 186 us).
    // Round 6: the <= 512-thread launches run instances with the step's body inlined under their own register budget (no scratch spills: k_tracker_step1<MAXT>); 80 objects
    // 169 -> 98 us with four waves (105 with eight), 500 objects 434 us with sixteen waves (the called copy, 128 registers a lane) -> 362 with four -> 261 with eight
    if (threads == 0) return (n_hint > 384) ? 512 : 256;
    if (threads != 64 && threads != 128 && threads != 256 && threads != 512 && threads != 1024) return -1;
    return threads;
}

extern "C" int y7t_kf_multi_gmc_f64(double* mean, double* cov, const double* warp, int N, y7t_stream stream) {
    Y7T_ARG_CHECK(N >= 0);
    if (N == 0) return 0;
    Y7T_ARG_CHECK(mean && cov && warp);
    hipLaunchKernelGGL(k_kf_gmc, dim3((N + 63) / 64), dim3(64), 0, S(stream), mean, cov, warp, N);
    Y7T_LAUNCH_CHECK();
    return 0;
}

extern "C" int y7t_tracker_step_batch(void* const* states, const float* const* dets, const int* n_dets, double* const* out_rows,
                                      int* out_count, int out_cap, int batch, int threads, const double* const* gmc_warps,
                                      y7t_stream stream) {
    Y7T_ARG_CHECK(batch >= 0 && out_cap >= 0);
    if (batch == 0) return 0;
    Y7T_ARG_CHECK(states && dets && n_dets && out_rows && out_count);
    const int nt = step_threads(threads);
    Y7T_ARG_CHECK(nt > 0);
    static std::atomic<unsigned long long> attr_done{0}, attr_done_m{0};
    if (nt <= 512) {
        if (int e = ensure_lds_once(k_tracker_step<512>, kFastBytes + Y7T_LDS_HDR, attr_done_m)) return e;
        hipLaunchKernelGGL(k_tracker_step<512>, dim3(batch), dim3(nt), kFastBytes + Y7T_LDS_HDR, S(stream), states, dets, n_dets, out_rows, out_count, out_cap, kFastBytes, gmc_warps);
    } else {
        if (int e = ensure_lds_once(k_tracker_step<1024>, kFastBytes + Y7T_LDS_HDR, attr_done)) return e;
        hipLaunchKernelGGL(k_tracker_step<1024>, dim3(batch), dim3(nt), kFastBytes + Y7T_LDS_HDR, S(stream), states, dets, n_dets, out_rows, out_count, out_cap, kFastBytes, gmc_warps);
    }
    Y7T_LAUNCH_CHECK();
    return 0;
}

extern "C" int y7t_tracker_step(void* state, const float* dets, int n, double* out_rows, int out_cap, int* out_count, int threads,
                                const double* gmc_warp, y7t_stream stream) {
    Y7T_ARG_CHECK(state && out_rows && out_count && out_cap >= 0);
    Y7T_ARG_CHECK(n <= 0 || dets);
    const int nt = step_threads(threads, n);
    Y7T_ARG_CHECK(nt > 0);
    if (n >= 0 && state_kind(state) == Y7T_DEEPSORT) {
        y7t_set_error("y7t_tracker_step: this pool was initialised as DeepSORT -- frames with detections go through y7t_tracker_step_deepsort "
                      "(appearance rings); only the predict-only step (n < 0) is shared");
        return Y7T_E_STATE;
    }
    static std::atomic<unsigned long long> attr_done{0}, attr_done_s{0}, attr_done_m{0};
    const unsigned fb = step_fast_bytes(n);
    if (nt <= 256) {
        if (int e = ensure_lds_once(k_tracker_step1<256>, kFastBytes + Y7T_LDS_HDR, attr_done_s)) return e;
        hipLaunchKernelGGL(k_tracker_step1<256>, dim3(1), dim3(nt), fb + Y7T_LDS_HDR, S(stream), state, dets, n, out_rows, out_cap, out_count, fb, gmc_warp);
    } else if (nt <= 512) {
        if (int e = ensure_lds_once(k_tracker_step1<512>, kFastBytes + Y7T_LDS_HDR, attr_done_m)) return e;
        hipLaunchKernelGGL(k_tracker_step1<512>, dim3(1), dim3(nt), fb + Y7T_LDS_HDR, S(stream), state, dets, n, out_rows, out_cap, out_count, fb, gmc_warp);
    } else {
        if (int e = ensure_lds_once(k_tracker_step1<1024>, kFastBytes + Y7T_LDS_HDR, attr_done)) return e;
        hipLaunchKernelGGL(k_tracker_step1<1024>, dim3(1), dim3(nt), fb + Y7T_LDS_HDR, S(stream), state, dets, n, out_rows, out_cap, out_count, fb, gmc_warp);
    }
    Y7T_LAUNCH_CHECK();
    return 0;
}

extern "C" int y7t_tracker_step_frames(void* state, const float* const* dets, const int* n_dets, double* const* out_rows, int* const* out_count, int out_cap,
                                       int n_frames, int threads, const double* const* gmc_warps, y7t_stream stream) {
    Y7T_ARG_CHECK(state && out_cap >= 0 && n_frames >= 0);
    if (n_frames == 0) return 0;
    Y7T_ARG_CHECK(dets && n_dets && out_rows && out_count);
    const int nt = step_threads(threads);
    Y7T_ARG_CHECK(nt > 0);
    if (state_kind(state) == Y7T_DEEPSORT) {
        y7t_set_error("y7t_tracker_step_frames: a DeepSORT pool steps through y7t_tracker_step_deepsort (appearance rings)");
        return Y7T_E_STATE;
    }
    // LDS of the launch: header | fast scratch (cost matrix, assignment work arrays) | the pool's index lists for the length of the launch (y7t_arena_*), when the
    // CU's 160 KiB hold them beside at least 64 KiB of fast scratch (the default capacities, 1024 tracks x 1024 detections: 68 KiB of lists, 91 KiB of scratch)
    static const unsigned kLdsMax = 160 * 1024;
    static const int use_arena = y7t_switch("Y7T_TRACKER_ARENA", 1);      // (thread-safe static initialisation)
    const size_t ab = use_arena ? state_arena_bytes(state) : 0;
    unsigned arena = 0, fast = kFastBytes;
    if (ab && ab + 64 * 1024 + Y7T_LDS_HDR <= kLdsMax) { arena = (unsigned)((ab + 15) & ~(size_t)15); fast = (kLdsMax - Y7T_LDS_HDR - arena) & ~15u; if (fast > kFastBytes) fast = kFastBytes; }
    static std::atomic<unsigned long long> attr_done{0}, attr_done_s{0}, attr_done_m{0};
    if (nt <= 256) {
        if (int e = ensure_lds_once(k_tracker_step_frames<256>, kLdsMax, attr_done_s)) return e;
        hipLaunchKernelGGL(k_tracker_step_frames<256>, dim3(1), dim3(nt), Y7T_LDS_HDR + fast + arena, S(stream), state, dets, n_dets, out_rows, out_count, out_cap, n_frames,
                           fast, arena, gmc_warps);
    } else if (nt <= 512) {
        if (int e = ensure_lds_once(k_tracker_step_frames<512>, kLdsMax, attr_done_m)) return e;
        hipLaunchKernelGGL(k_tracker_step_frames<512>, dim3(1), dim3(nt), Y7T_LDS_HDR + fast + arena, S(stream), state, dets, n_dets, out_rows, out_count, out_cap, n_frames,
                           fast, arena, gmc_warps);
    } else {
        if (int e = ensure_lds_once(k_tracker_step_frames<1024>, kLdsMax, attr_done)) return e;
        hipLaunchKernelGGL(k_tracker_step_frames<1024>, dim3(1), dim3(nt), Y7T_LDS_HDR + fast + arena, S(stream), state, dets, n_dets, out_rows, out_count, out_cap, n_frames,
                           fast, arena, gmc_warps);
    }
    Y7T_LAUNCH_CHECK();
    return 0;
}

extern "C" size_t y7t_deepsort_feature_bytes(int cap_t, int cap_d, int feat_dim, int budget) {
    if (cap_t <= 0 || cap_d <= 0 || feat_dim <= 0 || budget <= 0) return 0;
    return y7t_feat_layout(cap_t, cap_d, feat_dim, budget).total;
}

extern "C" int y7t_deepsort_init(void* feat_state, size_t bytes, int cap_t, int cap_d, int feat_dim, int budget, y7t_stream stream) {
    Y7T_ARG_CHECK(feat_state && cap_t > 0 && cap_d > 0 && feat_dim > 0 && budget > 0);
    Y7T_ARG_CHECK(bytes >= y7t_feat_layout(cap_t, cap_d, feat_dim, budget).total);
    hipLaunchKernelGGL(k_feat_init, dim3(1), dim3(256), 0, S(stream), feat_state, cap_t, cap_d, feat_dim, budget);
    Y7T_LAUNCH_CHECK();
    return 0;
}

extern "C" int y7t_tracker_step_deepsort(void* state, void* feat_state, int cap_tracks, const float* dets, int n, const float* det_feats,
                                         double* out_rows, int out_cap, int* out_count, int threads, y7t_stream stream) {
    Y7T_ARG_CHECK(state && feat_state && out_rows && out_count && out_cap >= 0 && cap_tracks > 0 && n >= 0);
    Y7T_ARG_CHECK(n == 0 || (dets && det_feats));
    Y7T_ARG_CHECK(cap_tracks <= 64 * Y7T_EMBED_WORKERS);      // (k_embed_dist: a workgroup owns at most 64 live slots)
    const int nt = step_threads(threads, n);
    Y7T_ARG_CHECK(nt > 0);
    static std::atomic<unsigned long long> attr_done{0}, attr_done_s{0}, attr_done_m{0};
    if (int e = nt <= 256 ? ensure_lds_once(k_tracker_step_deepsort<256>, kFastBytes + Y7T_LDS_HDR, attr_done_s)
              : nt <= 512 ? ensure_lds_once(k_tracker_step_deepsort<512>, kFastBytes + Y7T_LDS_HDR, attr_done_m)
                          : ensure_lds_once(k_tracker_step_deepsort<1024>, kFastBytes + Y7T_LDS_HDR, attr_done)) return e;
    if (n > 0) {
        hipLaunchKernelGGL(k_ds_normalize, dim3((n + 3) / 4), dim3(256), 0, S(stream), feat_state, det_feats, n);      // a wave per detection
        Y7T_LAUNCH_CHECK();
        hipLaunchKernelGGL(k_embed_dist, dim3(cap_tracks < Y7T_EMBED_WORKERS ? cap_tracks : Y7T_EMBED_WORKERS, (n + 63) / 64), dim3(256), 0, S(stream), state, feat_state, n);
        Y7T_LAUNCH_CHECK();
    }
    const unsigned fb = step_fast_bytes(n);
    if (nt <= 256) hipLaunchKernelGGL(k_tracker_step_deepsort<256>, dim3(1), dim3(nt), fb + Y7T_LDS_HDR, S(stream), state, feat_state, dets, n, det_feats, out_rows, out_cap, out_count, fb);
    else if (nt <= 512) hipLaunchKernelGGL(k_tracker_step_deepsort<512>, dim3(1), dim3(nt), fb + Y7T_LDS_HDR, S(stream), state, feat_state, dets, n, det_feats, out_rows, out_cap, out_count, fb);
    else hipLaunchKernelGGL(k_tracker_step_deepsort<1024>, dim3(1), dim3(nt), fb + Y7T_LDS_HDR, S(stream), state, feat_state, dets, n, det_feats, out_rows, out_cap, out_count, fb);
    Y7T_LAUNCH_CHECK();
    if (n > 0) {
        hipLaunchKernelGGL(k_ds_store, dim3((n + 3) / 4), dim3(256), 0, S(stream), feat_state, det_feats);
        Y7T_LAUNCH_CHECK();
    }
    return 0;
}

static const char* kFieldNames[] = {"mean", "cov", "box", "score", "cls", "tid", "start", "frame", "tsu", "state", "act", "len",
                                    "inrem", "f32m", "tracked", "lost", "hdr_frame_id", "hdr_n_tracked", "hdr_n_lost",
                                    "hdr_status", "hdr_n_out", "hdr_n_removed_total", "hdr_prof", "total"};

extern "C" const char* y7t_tracker_field_name(int i) {
    const int n = (int)(sizeof(kFieldNames) / sizeof(kFieldNames[0]));
    return (i >= 0 && i < n) ? kFieldNames[i] : nullptr;
}

extern "C" int y7t_tracker_layout(int cap_t, int cap_d, int64_t* offsets, int max_fields) {
    const int n = (int)(sizeof(kFieldNames) / sizeof(kFieldNames[0]));
    if (!offsets || max_fields < n || cap_t <= 0 || cap_d <= 0) return n;
    const Y7TTrkLayout L = y7t_trk_layout(cap_t, cap_d);
    const size_t v[] = {L.mean, L.cov, L.box, L.score, L.cls, L.tid, L.start, L.frame, L.tsu, L.state, L.act, L.len, L.inrem, L.f32m,
                        L.tracked, L.lost, offsetof(Y7TTrkHdr, frame_id), offsetof(Y7TTrkHdr, n_tracked), offsetof(Y7TTrkHdr, n_lost),
                        offsetof(Y7TTrkHdr, status), offsetof(Y7TTrkHdr, n_out), offsetof(Y7TTrkHdr, n_removed_total), offsetof(Y7TTrkHdr, prof), L.total};
    for (int i = 0; i < n; ++i) offsets[i] = (int64_t)v[i];
    return n;
}
