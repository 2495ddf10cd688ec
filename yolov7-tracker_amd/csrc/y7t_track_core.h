// y7t_track_core.h -- tracker arithmetic for the MI355X tracking hot path.
//
// One source, two compilations:
//   * hipcc --offload-arch=gfx950 : every function is a __device__ function executed by ONE
//     workgroup (nt = 64..1024 threads).  This is the product.
//   * g++ -DY7T_HOSTSIM           : the same text with nt = 1, used ONLY by the CPU test-suite
//     (tests/_hostsim) to exercise the control flow where no GPU exists.  It is never loaded by
//     the product package.
//
// What it restates (reference file:line, /root/reference/...):
//   * Kalman filters  tracker/kalman_filter.py:158-411 (xyah), :414-605 (xywh), :607-646 (NSA)
//   * IoU distance    tracker/matching.py:44-82  (-> cython_bbox.bbox_overlaps, "+1" convention)
//   * linear_assignment tracker/matching.py:30-41 (-> lap.lapjv(extend_cost=True, cost_limit=t))
//   * ByteTrack / SORT frame step  tracker/bytetrack.py:41-204, tracker/basetrack.py:368-537
//   * STrack glue     tracker/basetrack.py:74-339, list helpers :540-576
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define Y7T_FN __device__ __forceinline__
#define Y7T_HD __host__ __device__ __forceinline__
#define Y7T_NOINL __device__ __noinline__
#define Y7T_MFN __device__ __forceinline__      /* member functions */
#define Y7T_DEVICE 1
#else
#define Y7T_FN static inline
#define Y7T_HD static inline
#define Y7T_NOINL static
#define Y7T_MFN inline
#define Y7T_DEVICE 0
#endif

// full unrolling of the fixed-size loops of the Kalman arithmetic: their local arrays (K, W, S, L: 100 doubles in y7t_kf_update) are indexed by the loop counters, and a
// loop the optimiser leaves rolled keeps them in SCRATCH memory (1088 bytes per lane in the frame step: every element a round trip to the L2); unrolled they are registers
#if Y7T_DEVICE
#define Y7T_UNROLL _Pragma("unroll")
#else
#define Y7T_UNROLL
#endif

#define Y7T_LARGE 1000000.0
#define Y7T_SP (1.0 / 20)
#define Y7T_SV (1.0 / 160)

enum { Y7T_KF_XYAH = 0, Y7T_KF_NAIVE = 1, Y7T_KF_XYWH = 2, Y7T_KF_NSA = 3 };
enum { Y7T_NEW = 0, Y7T_TRACKED = 1, Y7T_LOST = 2, Y7T_REMOVED = 3 };
enum { Y7T_SORT = 0, Y7T_BYTETRACK = 1, Y7T_BOTSORT = 2, Y7T_DEEPSORT = 3 };

// ---------------------------------------------------------------------------------------------
// execution context: one workgroup; rv/ri are >=32-entry cross-wave scratch arrays (LDS on device)
// ---------------------------------------------------------------------------------------------
struct Y7TExec {
    int tid, nt;
    double* rv;
    int* ri;
    char* fast;         // workgroup-private fast scratch (LDS) or null
    size_t fast_bytes;
    char* arena;        // workgroup-private home of the step's index lists for the duration of a launch (LDS; y7t_track_step.h: y7t_arena_*) or null
    size_t arena_bytes;
};

Y7T_FN void y7t_sync(const Y7TExec&) {
#if Y7T_DEVICE
    __syncthreads();
#endif
}

#if Y7T_DEVICE
#define Y7T_ATOMIC_MAX(p, v) atomicMax((p), (v))
#define Y7T_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#define Y7T_FETCH_ADD(p, v) atomicAdd((p), (v))
#define Y7T_ATOMIC_MIN_I(p, v) atomicMin((p), (v))
#else
#define Y7T_ATOMIC_MIN_I(p, v) (*(p) = (*(p) < (v)) ? *(p) : (v))
#define Y7T_FETCH_ADD(p, v) y7t_fetch_add_host((p), (v))
#define Y7T_ATOMIC_MAX(p, v) (*(p) = (*(p) > (v)) ? *(p) : (v))
#define Y7T_ATOMIC_ADD(p, v) (*(p) += (v))
#endif

#if !Y7T_DEVICE
static inline int y7t_fetch_add_host(int* p, int v) { const int o = *p; *p += v; return o; }
#endif

Y7T_FN bool y7t_lex_less(double av, int ai, double bv, int bi) { return av < bv || (av == bv && ai < bi); }

#if Y7T_DEVICE
// ---- wave64 reductions on the DPP crossbar (no LDS round trips): row_shr 1/2/4/8 inside each 16-lane row, then
//      row_bcast:15 / row_bcast:31 carry the row totals up; lane 63 ends with the full result, broadcast by readlane ----
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int y7t_dpp_i(int identity, int v) {
    return __builtin_amdgcn_update_dpp(identity, v, CTRL, ROW_MASK, 0xf, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double y7t_dpp_d(double identity, double v) {
    const long long iv = __builtin_bit_cast(long long, v), id = __builtin_bit_cast(long long, identity);
    const int lo = __builtin_amdgcn_update_dpp((int)id, (int)iv, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(id >> 32), (int)(iv >> 32), CTRL, ROW_MASK, 0xf, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double y7t_wave_min_d(double v) {
    const double inf = HUGE_VAL;
    v = fmin(v, y7t_dpp_d<0x111, 0xf>(inf, v));
    v = fmin(v, y7t_dpp_d<0x112, 0xf>(inf, v));
    v = fmin(v, y7t_dpp_d<0x114, 0xf>(inf, v));
    v = fmin(v, y7t_dpp_d<0x118, 0xf>(inf, v));
    v = fmin(v, y7t_dpp_d<0x142, 0xa>(inf, v));
    v = fmin(v, y7t_dpp_d<0x143, 0xc>(inf, v));
    const long long r = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)r, 63), hi = __builtin_amdgcn_readlane((int)(r >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ int y7t_wave_min_i(int v) {
    const int big = 0x7fffffff;
    int t;
    t = y7t_dpp_i<0x111, 0xf>(big, v); v = t < v ? t : v;
    t = y7t_dpp_i<0x112, 0xf>(big, v); v = t < v ? t : v;
    t = y7t_dpp_i<0x114, 0xf>(big, v); v = t < v ? t : v;
    t = y7t_dpp_i<0x118, 0xf>(big, v); v = t < v ? t : v;
    t = y7t_dpp_i<0x142, 0xa>(big, v); v = t < v ? t : v;
    t = y7t_dpp_i<0x143, 0xc>(big, v); v = t < v ? t : v;
    return __builtin_amdgcn_readlane(v, 63);
}
#endif

// ---- one wave as a 64-lane vector machine (y7t_assoc_sparse_try, step 4a): a per-lane variable is `T a[Y7T_WVN]` -- ONE register per lane on the device; the host
//      build (one thread) keeps all 64 lanes' values and runs every lane statement as a loop, so the goldens walk the same text.  `wv_lane` must be in scope. ----
#if Y7T_DEVICE
__device__ __forceinline__ double y7t_readlane_d(double v, int s) {
    const long long r = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)r, s), hi = __builtin_amdgcn_readlane((int)(r >> 32), s);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
// a small trivially copyable struct from lane `s` (uniform) to every lane, word by word through the scalar registers
template <class T>
__device__ __forceinline__ T y7t_readlane_t(const T& v, int s) {
    static_assert(sizeof(T) % 4 == 0, "words");
    struct W { int w[sizeof(T) / 4]; };
    W a = __builtin_bit_cast(W, v), b;
#pragma unroll
    for (int k = 0; k < (int)(sizeof(T) / 4); ++k) b.w[k] = __builtin_amdgcn_readlane(a.w[k], s);
    return __builtin_bit_cast(T, b);
}
__device__ __forceinline__ int y7t_ctz64(unsigned long long m) { return __ffsll((long long)m) - 1; }
__device__ __forceinline__ int y7t_popc64(unsigned long long m) { return __popcll(m); }
#define Y7T_WVN 1
#define Y7T_WV_EACH(l) for (int l __attribute__((unused)) = wv_lane, l##_once = 1; l##_once; l##_once = 0)
#define Y7T_WV(a, l) (a)[0]
#define Y7T_WV_AT_I(a, s) __builtin_amdgcn_readlane((a)[0], (s))
#define Y7T_WV_AT_D(a, s) y7t_readlane_d((a)[0], (s))
#define Y7T_WV_GATHER_I(a, idx) __shfl((a)[0], (idx))
#define Y7T_WV_SET(a, s, val) do { if (wv_lane == (s)) (a)[0] = (val); } while (0)
#define Y7T_WV_MIN_D(out, l, expr) do { const int l = wv_lane; (void)l; (out) = y7t_wave_min_d(expr); } while (0)
#define Y7T_WV_BALLOT(out, l, pred) do { const int l = wv_lane; (void)l; (out) = __ballot(pred); } while (0)
// a wave striding over n items that live in MEMORY (the work arrays of a component with more rows or columns than a wave has lanes): lane l takes l, l + 64, ...;
// Y7T_WV_FENCE between two phases that hand values from lane to lane through that memory -- a wave's own LDS / vector-memory operations complete in issue order,
// so the fence only has to hold the compiler to the program order
#define Y7T_WV_STRIDE(p, n) for (int p = wv_lane; p < (n); p += 64)
#define Y7T_WV_FENCE() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#else
static inline int y7t_ctz64(unsigned long long m) { return __builtin_ctzll(m); }
static inline int y7t_popc64(unsigned long long m) { return __builtin_popcountll(m); }
#define Y7T_WVN 64
#define Y7T_WV_EACH(l) for (int l = 0; l < 64; ++l)
#define Y7T_WV(a, l) (a)[l]
#define Y7T_WV_AT_I(a, s) (a)[s]
#define Y7T_WV_AT_D(a, s) (a)[s]
#define Y7T_WV_GATHER_I(a, idx) (a)[idx]
#define Y7T_WV_SET(a, s, val) do { (a)[s] = (val); } while (0)
#define Y7T_WV_MIN_D(out, l, expr) do { (out) = HUGE_VAL; for (int l = 0; l < 64; ++l) { const double e_ = (expr); if (e_ < (out)) (out) = e_; } } while (0)
#define Y7T_WV_BALLOT(out, l, pred) do { (out) = 0ull; for (int l = 0; l < 64; ++l) if (pred) (out) |= 1ull << l; } while (0)
#define Y7T_WV_STRIDE(p, n) for (int p = 0; p < (n); ++p)
#define Y7T_WV_FENCE() do { } while (0)
#endif

// all-reduce: lexicographic minimum of (v, i) over the workgroup; every thread gets the result
Y7T_FN void y7t_argmin(const Y7TExec& ex, double& v, int& i) {
#if Y7T_DEVICE
    {
        const double m = y7t_wave_min_d(v);
        const int mi = y7t_wave_min_i(v == m ? i : 0x7fffffff);
        v = m; i = mi;
    }
    if (ex.nt > 64) {
        const int w = ex.tid >> 6, nw = ex.nt >> 6;
        __syncthreads();
        if ((ex.tid & 63) == 0) { ex.rv[w] = v; ex.ri[w] = i; }
        __syncthreads();
        v = ex.rv[0]; i = ex.ri[0];
        for (int k = 1; k < nw; ++k) {
            const double ov = ex.rv[k];
            const int oi = ex.ri[k];
            if (y7t_lex_less(ov, oi, v, i)) { v = ov; i = oi; }
        }
    }
#else
    (void)ex; (void)v; (void)i;
#endif
}

// all-reduce: the two lexicographically smallest (v, i) pairs
struct Y7TMin2 { double v1, v2; int i1, i2; };
Y7T_FN void y7t_min2_push(Y7TMin2& m, double v, int i) {
    if (y7t_lex_less(v, i, m.v1, m.i1)) { m.v2 = m.v1; m.i2 = m.i1; m.v1 = v; m.i1 = i; }
    else if (y7t_lex_less(v, i, m.v2, m.i2)) { m.v2 = v; m.i2 = i; }
}
Y7T_FN void y7t_min2(const Y7TExec& ex, Y7TMin2& m) {
#if Y7T_DEVICE
    {
        // best pair of the wave, then the best pair once the winner is removed from the lane that owned it
        const double b1 = y7t_wave_min_d(m.v1);
        const int j1 = y7t_wave_min_i(m.v1 == b1 ? m.i1 : 0x7fffffff);
        const bool own = (m.i1 == j1) && (m.v1 == b1);
        const double c = own ? m.v2 : m.v1;
        const int ci = own ? m.i2 : m.i1;
        const double b2 = y7t_wave_min_d(c);
        const int j2 = y7t_wave_min_i(c == b2 ? ci : 0x7fffffff);
        m.v1 = b1; m.i1 = j1; m.v2 = b2; m.i2 = j2;
    }
    if (ex.nt > 64) {
        const int w = ex.tid >> 6, nw = ex.nt >> 6;
        __syncthreads();
        if ((ex.tid & 63) == 0) { ex.rv[2 * w] = m.v1; ex.rv[2 * w + 1] = m.v2; ex.ri[2 * w] = m.i1; ex.ri[2 * w + 1] = m.i2; }
        __syncthreads();
        Y7TMin2 r = {HUGE_VAL, HUGE_VAL, 0x7fffffff, 0x7fffffff};
        for (int k = 0; k < 2 * nw; ++k) y7t_min2_push(r, ex.rv[k], ex.ri[k]);
        m = r;
    }
#else
    (void)ex; (void)m;
#endif
}

// all-reduce of two independent integer minima
Y7T_FN void y7t_imin2(const Y7TExec& ex, int& a, int& b) {
#if Y7T_DEVICE
    a = y7t_wave_min_i(a);
    b = y7t_wave_min_i(b);
    if (ex.nt > 64) {
        const int w = ex.tid >> 6, nw = ex.nt >> 6;
        __syncthreads();
        if ((ex.tid & 63) == 0) { ex.ri[2 * w] = a; ex.ri[2 * w + 1] = b; }
        __syncthreads();
        a = ex.ri[0]; b = ex.ri[1];
        for (int k = 1; k < nw; ++k) {
            a = ex.ri[2 * k] < a ? ex.ri[2 * k] : a;
            b = ex.ri[2 * k + 1] < b ? ex.ri[2 * k + 1] : b;
        }
    }
#else
    (void)ex; (void)a; (void)b;
#endif
}

// ordered stream compaction: out[cnt++] = i for every i in [0,n) with flag(i), ascending i.
// Returns the new count (uniform).  `out` may be appended to (cnt0).
template <class F>
Y7T_FN int y7t_compact(const Y7TExec& ex, int n, F flag, int* out, int cnt0) {
#if Y7T_DEVICE
    int cnt = cnt0;
    const int lane = ex.tid & 63, w = ex.tid >> 6, nw = (ex.nt + 63) >> 6;
    for (int base = 0; base < n; base += ex.nt) {
        const int i = base + ex.tid;
        const bool f = (i < n) && flag(i);
        const unsigned long long b = __ballot(f);
        const int rank = __popcll(b & ((1ull << lane) - 1ull));
        const int tot = __popcll(b);
        int off = 0, all = tot;
        if (nw > 1) {
            __syncthreads();
            if (lane == 0) ex.ri[w] = tot;
            __syncthreads();
            all = 0;
            for (int k = 0; k < nw; ++k) { if (k == w) off = all; all += ex.ri[k]; }
        }
        if (f) out[cnt + off + rank] = i;
        cnt += all;
    }
    __syncthreads();
    return cnt;
#else
    (void)ex;
    int cnt = cnt0;
    for (int i = 0; i < n; ++i) if (flag(i)) out[cnt++] = i;
    return cnt;
#endif
}

// exclusive prefix scans over the workgroup in thread order (thread t gets the reduction of the values of threads 0 .. t-1): minimum of doubles (HUGE_VAL for
// thread 0), sum of ints (+ the total, uniform).  One thread (the host build): the identity / the value itself.
Y7T_FN double y7t_block_excl_min_d(const Y7TExec& ex, double v) {
#if Y7T_DEVICE
    const int lane = ex.tid & 63, w = ex.tid >> 6, nw = (ex.nt + 63) >> 6;
    double inc = v;
    for (int off = 1; off < 64; off <<= 1) { const double t = __shfl_up(inc, off); if (lane >= off) inc = fmin(inc, t); }
    double exc = __shfl_up(inc, 1);
    if (lane == 0) exc = HUGE_VAL;
    if (nw > 1) {
        __syncthreads();
        if (lane == 63) ex.rv[w] = inc;
        __syncthreads();
        for (int k = 0; k < w; ++k) exc = fmin(exc, ex.rv[k]);
    }
    return exc;
#else
    (void)ex; (void)v;
    return HUGE_VAL;
#endif
}
Y7T_FN int y7t_block_excl_sum_i(const Y7TExec& ex, int v, int& total) {
#if Y7T_DEVICE
    const int lane = ex.tid & 63, w = ex.tid >> 6, nw = (ex.nt + 63) >> 6;
    int inc = v;
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
    int exc = inc - v;
    total = __shfl(inc, 63);
    if (nw > 1) {
        __syncthreads();
        if (lane == 63) ex.ri[w] = inc;
        __syncthreads();
        int before = 0, all = 0;
        for (int k = 0; k < nw; ++k) { if (k < w) before += ex.ri[k]; all += ex.ri[k]; }
        exc += before; total = all;
    }
    return exc;
#else
    (void)ex;
    total = v;
    return 0;
#endif
}
// one int from thread 0 to every thread (two barriers on the device)
Y7T_FN int y7t_bcast_i(const Y7TExec& ex, int v, int slot) {
#if Y7T_DEVICE
    if (ex.nt > 64) {
        __syncthreads();
        if (ex.tid == 0) ex.ri[slot] = v;
        __syncthreads();
        return ex.ri[slot];
    }
    return __builtin_amdgcn_readfirstlane(v);
#else
    (void)ex; (void)slot;
    return v;
#endif
}

// ---------------------------------------------------------------------------------------------
// Kalman filters (8-d state; float64).  kind: 0 xyah ('default'), 2 xywh ('botsort'), 3 NSA
// ('strongsort').  The 7-d 'naive' filter is not part of the ByteTrack/SORT hot path.
// ---------------------------------------------------------------------------------------------
Y7T_FN double y7t_sq(double a) { return a * a; }

// kalman_filter.py:190-221 / :436-467.  z = measurement (xyah or xywh).  f32_std reproduces the
// reference under numpy>=2: a float32 measurement keeps the std products in float32 (NEP 50).
Y7T_FN void y7t_kf_initiate(int kind, const double* z, int f32_std, double* mean, double* cov) {
    Y7T_UNROLL
    for (int i = 0; i < 4; ++i) { mean[i] = z[i]; mean[4 + i] = 0.0; }
    Y7T_UNROLL
    for (int i = 0; i < 64; ++i) cov[i] = 0.0;
    double std_[8];
    if (f32_std) {
        const float p2 = (float)(2 * Y7T_SP), v10 = (float)(10 * Y7T_SV);
        if (kind == Y7T_KF_XYWH) {
            const float w = (float)z[2], h = (float)z[3];
            const float s[8] = {p2 * w, p2 * h, p2 * w, p2 * h, v10 * w, v10 * h, v10 * w, v10 * h};
            // every entry is float32 there, so np.square runs in float32 too
            Y7T_UNROLL
            for (int i = 0; i < 8; ++i) { const float q = s[i] * s[i]; cov[i * 9] = (double)q; }
            return;
        }
        const float h = (float)z[3];
        const float a = p2 * h, b = v10 * h;
        std_[0] = a; std_[1] = a; std_[2] = 1e-2; std_[3] = a;
        std_[4] = b; std_[5] = b; std_[6] = 1e-5; std_[7] = b;
    } else if (kind == Y7T_KF_XYWH) {
        const double w = z[2], h = z[3];
        std_[0] = 2 * Y7T_SP * w; std_[1] = 2 * Y7T_SP * h; std_[2] = 2 * Y7T_SP * w; std_[3] = 2 * Y7T_SP * h;
        std_[4] = 10 * Y7T_SV * w; std_[5] = 10 * Y7T_SV * h; std_[6] = 10 * Y7T_SV * w; std_[7] = 10 * Y7T_SV * h;
    } else {
        const double h = z[3];
        std_[0] = 2 * Y7T_SP * h; std_[1] = 2 * Y7T_SP * h; std_[2] = 1e-2; std_[3] = 2 * Y7T_SP * h;
        std_[4] = 10 * Y7T_SV * h; std_[5] = 10 * Y7T_SV * h; std_[6] = 1e-5; std_[7] = 10 * Y7T_SV * h;
    }
    Y7T_UNROLL
    for (int i = 0; i < 8; ++i) cov[i * 9] = y7t_sq(std_[i]);
}

// kalman_filter.py:289-329 (multi_predict) / :534-571.  In place on one track.
// F = [[I4, I4], [0, I4]]:  x <- F x ;  P <- F P F^T + diag(q)
Y7T_FN void y7t_kf_predict(int kind, double* mean, double* cov) {
    double q[8];
    if (kind == Y7T_KF_XYWH) {
        const double w = mean[2], h = mean[3];
        q[0] = Y7T_SP * w; q[1] = Y7T_SP * h; q[2] = Y7T_SP * w; q[3] = Y7T_SP * h;
        q[4] = Y7T_SV * w; q[5] = Y7T_SV * h; q[6] = Y7T_SV * w; q[7] = Y7T_SV * h;
    } else {
        const double h = mean[3];
        q[0] = Y7T_SP * h; q[1] = Y7T_SP * h; q[2] = 1e-2; q[3] = Y7T_SP * h;
        q[4] = Y7T_SV * h; q[5] = Y7T_SV * h; q[6] = 1e-5; q[7] = Y7T_SV * h;
    }
    Y7T_UNROLL
    for (int i = 0; i < 4; ++i) mean[i] = mean[i] + mean[4 + i];
    // left = F P : rows 0..3 += rows 4..7
    Y7T_UNROLL
    for (int r = 0; r < 4; ++r)
        Y7T_UNROLL
        for (int c = 0; c < 8; ++c) cov[r * 8 + c] = cov[r * 8 + c] + cov[(r + 4) * 8 + c];
    // (F P) F^T : cols 0..3 += cols 4..7 ; + Q
    Y7T_UNROLL
    for (int r = 0; r < 8; ++r) {
        Y7T_UNROLL
        for (int c = 0; c < 4; ++c) cov[r * 8 + c] = cov[r * 8 + c] + cov[r * 8 + c + 4];
        cov[r * 9] = cov[r * 9] + y7t_sq(q[r]);
    }
}

// measurement-noise std, kalman_filter.py:277-282 / :522-527 / :618-626
Y7T_FN void y7t_kf_rstd(int kind, const double* mean, double conf, double* r) {
    if (kind == Y7T_KF_XYWH) {
        const double w = mean[2], h = mean[3];
        r[0] = Y7T_SP * w; r[1] = Y7T_SP * h; r[2] = Y7T_SP * w; r[3] = Y7T_SP * h;
    } else {
        const double h = mean[3];
        r[0] = Y7T_SP * h; r[1] = Y7T_SP * h; r[2] = 1e-1; r[3] = Y7T_SP * h;
        if (kind == Y7T_KF_NSA) for (int i = 0; i < 4; ++i) r[i] = (1 - conf) * r[i];
    }
}

// kalman_filter.py:260-287: projected mean (4) and S = H P H^T + R (4x4)
Y7T_FN void y7t_kf_project(int kind, const double* mean, const double* cov, double conf, double* pm, double* S) {
    double r[4];
    y7t_kf_rstd(kind, mean, conf, r);
    Y7T_UNROLL
    for (int a = 0; a < 4; ++a) {
        pm[a] = mean[a];
        Y7T_UNROLL
        for (int b = 0; b < 4; ++b) S[a * 4 + b] = cov[a * 8 + b];
        S[a * 5] = S[a * 5] + y7t_sq(r[a]);
    }
}

// lower Cholesky of a 4x4 SPD matrix (LAPACK dpotrf 'L' order of operations)
Y7T_FN void y7t_chol4(const double* S, double* L) {
    Y7T_UNROLL
    for (int j = 0; j < 4; ++j) {
        double s = S[j * 4 + j];
        Y7T_UNROLL
        for (int k = 0; k < j; ++k) s -= L[j * 4 + k] * L[j * 4 + k];
        const double d = sqrt(s);
        L[j * 4 + j] = d;
        Y7T_UNROLL
        for (int i = j + 1; i < 4; ++i) {
            double t = S[i * 4 + j];
            Y7T_UNROLL
            for (int k = 0; k < j; ++k) t -= L[i * 4 + k] * L[j * 4 + k];
            L[i * 4 + j] = t / d;
        }
    }
}

// solve (L L^T) x = b in place
Y7T_FN void y7t_chol4_solve(const double* L, double* b) {
    Y7T_UNROLL
    for (int i = 0; i < 4; ++i) {
        double t = b[i];
        Y7T_UNROLL
        for (int k = 0; k < i; ++k) t -= L[i * 4 + k] * b[k];
        b[i] = t / L[i * 4 + i];
    }
    Y7T_UNROLL
    for (int i = 3; i >= 0; --i) {
        double t = b[i];
        Y7T_UNROLL
        for (int k = i + 1; k < 4; ++k) t -= L[k * 4 + i] * b[k];
        b[i] = t / L[i * 4 + i];
    }
}

// kalman_filter.py:331-363.  K = P H^T S^-1 (Cholesky), x += K (z - Hx), P -= K (S K^T).  In place.
Y7T_FN void y7t_kf_update(int kind, double* mean, double* cov, const double* z, double conf) {
    double pm[4], S[16], L[16], K[32], W[32];
    y7t_kf_project(kind, mean, cov, conf, pm, S);
    y7t_chol4(S, L);
    Y7T_UNROLL
    for (int r = 0; r < 8; ++r) {
        double b[4] = {cov[r * 8 + 0], cov[r * 8 + 1], cov[r * 8 + 2], cov[r * 8 + 3]};
        y7t_chol4_solve(L, b);
        Y7T_UNROLL
        for (int a = 0; a < 4; ++a) K[r * 4 + a] = b[a];
    }
    double innov[4];
    Y7T_UNROLL
    for (int a = 0; a < 4; ++a) innov[a] = z[a] - pm[a];
    Y7T_UNROLL
    for (int r = 0; r < 8; ++r) {
        double s = 0.0;
        Y7T_UNROLL
        for (int a = 0; a < 4; ++a) s += innov[a] * K[r * 4 + a];
        mean[r] = mean[r] + s;
    }
    // numpy multi_dot((K, S, K^T)) with equal costs evaluates K (S K^T)
    Y7T_UNROLL
    for (int a = 0; a < 4; ++a)
        Y7T_UNROLL
        for (int c = 0; c < 8; ++c) {
            double s = 0.0;
            Y7T_UNROLL
            for (int b = 0; b < 4; ++b) s += S[a * 4 + b] * K[c * 4 + b];
            W[a * 8 + c] = s;
        }
    Y7T_UNROLL
    for (int r = 0; r < 8; ++r)
        Y7T_UNROLL
        for (int c = 0; c < 8; ++c) {
            double s = 0.0;
            Y7T_UNROLL
            for (int a = 0; a < 4; ++a) s += K[r * 4 + a] * W[a * 8 + c];
            cov[r * 8 + c] = cov[r * 8 + c] - s;
        }
}

// botsort.py:250-269 multi_gmc: camera-motion compensation of one track.  H = [R | t] (2x3, row-major);
// R8 = kron(I4, R):  mean <- R8 mean, mean[:2] += t,  cov <- R8 cov R8^T
struct Y7TWarp { double r00, r01, tx, r10, r11, ty; };
Y7T_FN Y7TWarp y7t_warp_load(const double* H) { return Y7TWarp{H[0], H[1], H[2], H[3], H[4], H[5]}; }
Y7T_FN void y7t_kf_gmc(const Y7TWarp& H, double* mean, double* cov) {
    const double r00 = H.r00, r01 = H.r01, tx = H.tx, r10 = H.r10, r11 = H.r11, ty = H.ty;
    Y7T_UNROLL
    for (int b = 0; b < 4; ++b) {
        const double a = mean[2 * b], c = mean[2 * b + 1];
        mean[2 * b] = r00 * a + r01 * c;
        mean[2 * b + 1] = r10 * a + r11 * c;
    }
    mean[0] = mean[0] + tx; mean[1] = mean[1] + ty;
    // cov <- R8 cov R8^T, a row pair (2b, 2b+1) at a time: the left multiply mixes the two rows, the right multiply then mixes column pairs INSIDE each row -- the same
    // products and sums on the same values as two in-place passes over the whole matrix, with every element loaded and stored once (round 5: the in-place passes, 128
    // dependent loads and stores through a generic pointer per track, were the part of the BoT-SORT step that three builds of the frame step mis-executed on the device --
    // right without camera-motion warps, right on the host; scripts/debug_botsort.py, profiles/r05_tracker_association.txt)
    Y7T_UNROLL
    for (int b = 0; b < 4; ++b) {
        double u[8], w[8];
        Y7T_UNROLL
        for (int c = 0; c < 8; ++c) {
            const double a = cov[(2 * b) * 8 + c], d = cov[(2 * b + 1) * 8 + c];
            u[c] = r00 * a + r01 * d;
            w[c] = r10 * a + r11 * d;
        }
        Y7T_UNROLL
        for (int q = 0; q < 4; ++q) {
            const double a = u[2 * q], d = u[2 * q + 1];
            cov[(2 * b) * 8 + 2 * q] = a * r00 + d * r01;
            cov[(2 * b) * 8 + 2 * q + 1] = a * r10 + d * r11;
            const double e = w[2 * q], f = w[2 * q + 1];
            cov[(2 * b + 1) * 8 + 2 * q] = e * r00 + f * r01;
            cov[(2 * b + 1) * 8 + 2 * q + 1] = e * r10 + f * r11;
        }
    }
}

// kalman_filter.py:365-411: squared Mahalanobis distance of one measurement (metric='maha')
Y7T_FN double y7t_kf_gating(int kind, const double* mean, const double* cov, const double* z, int only_position) {
    double pm[4], S[16], L[16] = {0};
    y7t_kf_project(kind, mean, cov, 0.0, pm, S);
    const int n = only_position ? 2 : 4;
    // Cholesky of the leading n x n block, forward substitution
    double acc = 0.0, yv[4];
    Y7T_UNROLL
    for (int j = 0; j < n; ++j) {
        double s = S[j * 4 + j];
        Y7T_UNROLL
        for (int k = 0; k < j; ++k) s -= L[j * 4 + k] * L[j * 4 + k];
        const double d = sqrt(s);
        L[j * 4 + j] = d;
        Y7T_UNROLL
        for (int i = j + 1; i < n; ++i) {
            double t = S[i * 4 + j];
            Y7T_UNROLL
            for (int k = 0; k < j; ++k) t -= L[i * 4 + k] * L[j * 4 + k];
            L[i * 4 + j] = t / d;
        }
    }
    Y7T_UNROLL
    for (int i = 0; i < n; ++i) {
        double t = z[i] - pm[i];
        Y7T_UNROLL
        for (int k = 0; k < i; ++k) t -= L[i * 4 + k] * yv[k];
        yv[i] = t / L[i * 4 + i];
        acc += yv[i] * yv[i];
    }
    return acc;
}

// ---------------------------------------------------------------------------------------------
// IoU distance with the "+1 pixel" convention (cython_bbox.bbox_overlaps; matching.py:56-82)
// ---------------------------------------------------------------------------------------------
Y7T_FN double y7t_iou_dist(const double* b, const double* q) {
    const double box_area = (q[2] - q[0] + 1) * (q[3] - q[1] + 1);
    double ov = 0.0;
    const double iw = fmin(b[2], q[2]) - fmax(b[0], q[0]) + 1;
    if (iw > 0) {
        const double ih = fmin(b[3], q[3]) - fmax(b[1], q[1]) + 1;
        if (ih > 0) {
            const double ua = (b[2] - b[0] + 1) * (b[3] - b[1] + 1) + box_area - iw * ih;
            ov = iw * ih / ua;
        }
    }
    return 1 - ov;
}

// a row's value out of the lane that loaded it (y7t_pairs): lane r of the wave, or the caller's own when r < 0 (host build, partial waves)
#if Y7T_DEVICE
__device__ __forceinline__ int y7t_row_at(int v, int r) { return r < 0 ? v : __builtin_amdgcn_readlane(v, r); }
__device__ __forceinline__ float y7t_row_at(float v, int r) { return r < 0 ? v : __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), r)); }
__device__ __forceinline__ double y7t_row_at(double v, int r) { return r < 0 ? v : y7t_readlane_d(v, r); }
#else
static inline int y7t_row_at(int v, int) { return v; }
static inline float y7t_row_at(float v, int) { return v; }
static inline double y7t_row_at(double v, int) { return v; }
#endif

// every (row, column) pair of an na x nb problem: body(i, j, rl, r, cj) with cj = colctx(j) and row i's context = y7t_row_at(rl.., r).  Device: a lane per column, a wave
// per row residue; what the body needs of a ROW is loaded 64 rows at a time, a lane each, and taken out of that lane's registers through the scalar registers when the
// row's turn comes (round 5: a row's box used to be four dependent global loads in front of every pair's arithmetic -- 244 pairs per lane at 500 x 500, each a memory
// round trip: 368 of the ByteTrack step's 2217 kcycles, profiles/r03_tracker_phases.txt).
// Round 6 -- GROUP REJECTION (geo.on): the columns are enumerated through `colperm` (position -> column: the caller has sorted them by x), so the 64 columns of a wave
// are neighbours in the image; geo.group() bounds them (a wave reduction per 64 columns), every lane tests the row IT has loaded against that bound, and only the rows
// of the ballot are walked at all -- 64 row tests for the price of one pair.  A skipped pair is one the body would have seen at "apart" (cost exactly 1), so the body
// sees every pair that can matter, under the SAME indices: only the order of enumeration changes, which no caller depends on (candidate appends are unordered already).
struct Y7TNoGeo {
    static constexpr bool on = false;
    struct G { int none; };
    template <class C> Y7T_MFN G group(const C&, bool) const { return G{0}; }
    template <class R> Y7T_MFN bool near(const R&, const G&) const { return true; }
    template <class C> Y7T_MFN double key(const C&) const { return 0.0; }
};
// position -> column in ascending order of the BIN of key[] (a counting sort over 64 equal bins between the smallest and the largest key; inside a bin the order is
// whatever the atomics make it -- the callers only want a wave's 64 consecutive positions to be neighbours in the image, and no result depends on the order of
// enumeration).  A full rank sort (nb broadcast reads per column) cost as much as the group rejection saved at 500 columns: measured, round 6, session r6e.
// key[] and perm[] are nb entries, hist[] 128 ints of workgroup scratch.  NaN and +inf keys go to the last bin, -inf to the first.  Device only.
#if Y7T_DEVICE
Y7T_FN void y7t_bin_perm(const Y7TExec& ex, int nb, const double* key, int* perm, int* hist) {
    double lo = HUGE_VAL, hi = -HUGE_VAL;
    for (int j = ex.tid; j < nb; j += ex.nt) { const double k = key[j]; if (fabs(k) < 1e300) { lo = k < lo ? k : lo; hi = k > hi ? k : hi; } }      // (finite keys only)
    lo = y7t_wave_min_d(lo); hi = -y7t_wave_min_d(-hi);
    if (ex.tid < 128) hist[ex.tid] = 0;
    if (ex.nt > 64) {
        const int w = ex.tid >> 6, nw = ex.nt >> 6;
        __syncthreads();
        if ((ex.tid & 63) == 0) { ex.rv[2 * w] = lo; ex.rv[2 * w + 1] = hi; }
        __syncthreads();
        for (int k = 0; k < nw; ++k) { lo = ex.rv[2 * k] < lo ? ex.rv[2 * k] : lo; hi = ex.rv[2 * k + 1] > hi ? ex.rv[2 * k + 1] : hi; }
    } else {
        for (int k = ex.tid + 64; k < 128; k += 64) hist[k] = 0;
    }
    const double scale = (hi > lo) ? 64.0 / (hi - lo) : 0.0;
    auto bin_of = [&](double k) -> int { const double t = (k - lo) * scale; return t >= 63.0 ? 63 : t >= 0.0 ? (int)t : (t == t ? 0 : 63); };      // (+inf and NaN: the last bin, -inf: the first)
    __syncthreads();
    for (int j = ex.tid; j < nb; j += ex.nt) atomicAdd(hist + bin_of(key[j]), 1);
    __syncthreads();
    if (ex.tid < 64) {                                     // exclusive prefix of the 64 bins -> the bins' cursors in hist[64 ..]
        const int c = hist[ex.tid];
        int inc = c;
        for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off); if (ex.tid >= off) inc += t; }
        hist[64 + ex.tid] = inc - c;
    }
    __syncthreads();
    for (int j = ex.tid; j < nb; j += ex.nt) perm[atomicAdd(hist + 64 + bin_of(key[j]), 1)] = j;
    __syncthreads();
}
#endif
template <class ColFn, class RowFn, class Body, class Geo = Y7TNoGeo>
Y7T_FN void y7t_pairs(const Y7TExec& ex, int na, int nb, ColFn colctx, RowFn rowctx, Body body, const int* colperm = nullptr, Geo geo = Geo()) {
#if Y7T_DEVICE
    if (ex.nt >= 64) {
        const int nw = ex.nt >> 6, wave = ex.tid >> 6, lane = ex.tid & 63;
        for (int jb = 0; jb < nb; jb += 64) {
            const bool jv = jb + lane < nb;
            const int jp = jv ? jb + lane : 0;
            const int j = colperm ? colperm[jp] : jp;
            const auto cj = colctx(j);
            const auto grp = geo.group(cj, jv);
            for (int ib = wave; ib < na; ib += nw * 64) {               // this wave's rows ib, ib + nw, ...: 64 of them at a time
                const int il = ib + nw * lane;
                const auto rl = rowctx(il < na ? il : 0);
                int nr = (na - ib + nw - 1) / nw;
                nr = nr < 64 ? nr : 64;
                if (Geo::on) {
                    unsigned long long m = __ballot(lane < nr && geo.near(rl, grp));
                    while (m) {
                        const int r = y7t_ctz64(m);
                        m &= m - 1ull;
                        if (jv) body(ib + nw * r, j, rl, r, cj);
                    }
                } else {
                    for (int r = 0; r < nr; ++r)
                        if (jv) body(ib + nw * r, j, rl, r, cj);
                }
            }
        }
        return;
    }
#endif
    (void)colperm; (void)geo;
    for (int j = ex.tid; j < nb; j += ex.nt) {
        const auto cj = colctx(j);
        for (int i = 0; i < na; ++i) body(i, j, rowctx(i), -1, cj);
    }
}

// IoU pairs: the boxes with float32 copies for a pre-test.  A row box entirely outside the column box grown by 2 px has iw <= -1 or ih <= -1 in y7t_iou_dist, i.e. IoU 0
// and cost exactly 1; the float32 copies are off by < 0.75 px for |coordinate| < 2^21, boxes beyond that (or NaN) get infinities that never reject.
struct Y7TBoxC { double v[4]; float g[4]; };      // a column: the box, and the box grown by 2 px
struct Y7TBoxR { double v[4]; float f[4]; };      // a row
Y7T_FN Y7TBoxC y7t_box_col(const double* q) {
    const bool ok = fabs(q[0]) < 2097152.0 && fabs(q[1]) < 2097152.0 && fabs(q[2]) < 2097152.0 && fabs(q[3]) < 2097152.0;
    const float inf = (float)HUGE_VAL;
    return Y7TBoxC{{q[0], q[1], q[2], q[3]}, {ok ? (float)(q[0] - 2.0) : -inf, ok ? (float)(q[1] - 2.0) : -inf, ok ? (float)(q[2] + 2.0) : inf, ok ? (float)(q[3] + 2.0) : inf}};
}
Y7T_FN Y7TBoxR y7t_box_row(const double* b) {
    const bool ok = fabs(b[0]) < 2097152.0 && fabs(b[1]) < 2097152.0 && fabs(b[2]) < 2097152.0 && fabs(b[3]) < 2097152.0;
    const float inf = (float)HUGE_VAL;
    return Y7TBoxR{{b[0], b[1], b[2], b[3]}, {ok ? (float)b[0] : -inf, ok ? (float)b[1] : -inf, ok ? (float)b[2] : inf, ok ? (float)b[3] : inf}};
}
// true: the pair cannot overlap (four float32 subtractions and a maximum, against the ~20 float64 operations of the exact test)
Y7T_FN bool y7t_box_apart(const Y7TBoxR& rl, int r, const Y7TBoxC& q) {
    const float f0 = y7t_row_at(rl.f[0], r), f1 = y7t_row_at(rl.f[1], r), f2 = y7t_row_at(rl.f[2], r), f3 = y7t_row_at(rl.f[3], r);
    return fmaxf(fmaxf(q.g[0] - f2, f0 - q.g[2]), fmaxf(q.g[1] - f3, f1 - q.g[3])) >= 0.0f;
}
Y7T_FN double y7t_box_iou_dist(const Y7TBoxR& rl, int r, const Y7TBoxC& q) {
    if (y7t_box_apart(rl, r, q)) return 1.0;
    const double b[4] = {y7t_row_at(rl.v[0], r), y7t_row_at(rl.v[1], r), y7t_row_at(rl.v[2], r), y7t_row_at(rl.v[3], r)};
    return y7t_iou_dist(b, q.v);
}
// group rejection for box pairs (y7t_pairs): the union of a wave's GROWN column boxes; a row box apart from the union is apart from every one of them (y7t_box_apart's
// own test against a larger box), i.e. every pair of that row and those columns costs exactly 1.  Infinities (coordinates beyond 2^21, NaN) make a bound that rejects nothing.
struct Y7TBoxGeo {
    static constexpr bool on = true;
    struct G { float x0, y0, x1, y1; };
    Y7T_MFN G group(const Y7TBoxC& c, bool valid) const {
        const float inf = (float)HUGE_VAL;
        float x0 = valid ? c.g[0] : inf, y0 = valid ? c.g[1] : inf, x1 = valid ? c.g[2] : -inf, y1 = valid ? c.g[3] : -inf;
#if Y7T_DEVICE
        for (int o = 32; o > 0; o >>= 1) {
            x0 = fminf(x0, __shfl_xor(x0, o)); y0 = fminf(y0, __shfl_xor(y0, o));
            x1 = fmaxf(x1, __shfl_xor(x1, o)); y1 = fmaxf(y1, __shfl_xor(y1, o));
        }
#endif
        return G{x0, y0, x1, y1};
    }
    Y7T_MFN double key(const Y7TBoxC& c) const { return c.v[0]; }      // the left edge
    Y7T_MFN bool near(const Y7TBoxR& r, const G& g) const { return !(fmaxf(fmaxf(g.x0 - r.f[2], r.f[0] - g.x1), fmaxf(g.y0 - r.f[3], r.f[1] - g.y1)) >= 0.0f); }
};


// ---------------------------------------------------------------------------------------------
// lap.lapjv(cost, extend_cost=True, cost_limit=limit): Jonker-Volgenant on the IMPLICIT
// (nr+nc)^2 extended matrix [[cost, limit/2], [limit/2, 0]] -- never materialised.
// Work arrays (length n = nr + nc) live in LDS on the device.
// Column reduction, reduction transfer and augmenting row reduction reproduce lap's sequential
// scans exactly (parallel reductions break ties towards the lowest index, which is what the
// sequential strict-< scans do).  The augmentation phase explores tied columns in ascending
// column order instead of lap's internal permutation order; both are optimal, and they return
// the same assignment whenever the optimal partial matching is unique.
// ---------------------------------------------------------------------------------------------
struct Y7TLap {
    const double* c;  // nr x nc, row stride ld
    int nr, nc, ld, n;
    double half;
    int *x, *y, *fr, *pred, *st, *cnt;  // n each
    double *v, *d;                      // n each
    long long* prof;                    // optional phase stamps (diagnostics)
};

Y7T_FN double y7t_lap_cost(const Y7TLap& L, int i, int j) {
    if (i < L.nr) return j < L.nc ? L.c[(size_t)i * L.ld + j] : L.half;
    return j < L.nc ? L.half : 0.0;
}

Y7T_HD size_t y7t_lap_ws_bytes(int n) { return (size_t)n * (6 * sizeof(int) + 2 * sizeof(double)); }

Y7T_FN void y7t_lap_bind(Y7TLap& L, void* ws, int n) {
    double* dp = (double*)ws;
    L.v = dp; L.d = dp + n;
    int* ip = (int*)(dp + 2 * (size_t)n);
    L.x = ip; L.y = ip + n; L.fr = ip + 2 * (size_t)n; L.pred = ip + 3 * (size_t)n; L.st = ip + 4 * (size_t)n;
    L.cnt = ip + 5 * (size_t)n;
}

// returns nothing; fills L.x[0..n), L.y[0..n) with the square solution of the extended problem
#if Y7T_DEVICE
#define Y7T_LPROF(i) do { if (L.prof && ex.tid == 0) L.prof[i] = clock64(); } while (0)
#else
#define Y7T_LPROF(i) do { } while (0)
#endif

// lap's _ccrrt_dense (column reduction + reduction transfer) and its two _carr_dense passes (augmenting row reduction) with every scan a workgroup-wide reduction
// whose ties go to the lowest index -- what the sequential strict-< scans do, so the state they leave (x, y, v, the free rows in fr[0 .. n_free) in lap's order) is
// lap's own.  -> n_free.  Shared by y7t_lap_solve and y7t_lap_solve_literal.
Y7T_FN int y7t_lap_reduce(const Y7TExec& ex, Y7TLap& L) {
    const int n = L.n, tid = ex.tid, nt = ex.nt;
    Y7T_LPROF(0);
    // ---- column reduction (lap: _ccrrt_dense) ----
    for (int j = tid; j < n; j += nt) {
        double best = Y7T_LARGE;
        int bi = 0;
        for (int i = 0; i < n; ++i) {
            const double c = y7t_lap_cost(L, i, j);
            if (c < best) { best = c; bi = i; }
        }
        L.v[j] = best; L.y[j] = bi;
        L.x[j] = -1; L.cnt[j] = 0;
    }
    y7t_sync(ex);
    for (int j = tid; j < n; j += nt) { Y7T_ATOMIC_MAX(&L.x[L.y[j]], j); Y7T_ATOMIC_ADD(&L.cnt[L.y[j]], 1); }
    y7t_sync(ex);
    for (int j = tid; j < n; j += nt) if (L.x[L.y[j]] != j) L.y[j] = -1;
    y7t_sync(ex);
    Y7T_LPROF(1);
    // free rows, ascending
    int n_free = y7t_compact(ex, n, [&](int i) { return L.x[i] < 0; }, L.fr, 0);
    // reduction transfer: rows assigned to a column that only they claimed
    for (int i = 0; i < n; ++i) {
        const int j = L.x[i];
        if (j < 0 || L.cnt[i] != 1) continue;  // uniform
        double m = Y7T_LARGE;
        int mi = 0;
        for (int j2 = tid; j2 < n; j2 += nt) {
            if (j2 == j) continue;
            const double c = y7t_lap_cost(L, i, j2) - L.v[j2];
            if (c < m) m = c;
        }
        y7t_argmin(ex, m, mi);
        y7t_sync(ex);
        if (tid == 0) L.v[j] -= m;
        y7t_sync(ex);
    }
    Y7T_LPROF(2);
    // ---- augmenting row reduction, two passes (lap: _carr_dense) ----
    for (int pass = 0; pass < 2 && n_free > 0; ++pass) {
        unsigned current = 0, rr_cnt = 0;
        int new_free = 0;
        while (current < (unsigned)n_free) {
            rr_cnt++;
            const int free_i = L.fr[current++];
            Y7TMin2 m = {HUGE_VAL, HUGE_VAL, 0x7fffffff, 0x7fffffff};
            for (int j = tid; j < n; j += nt) y7t_min2_push(m, y7t_lap_cost(L, free_i, j) - L.v[j], j);
            y7t_min2(ex, m);
            int j1 = m.i1, j2 = m.i2;
            double v1 = m.v1, v2 = m.v2;
            if (n < 2 || !(v2 < Y7T_LARGE)) { v2 = Y7T_LARGE; j2 = -1; }
            int i0 = L.y[j1];
            const double vj1 = L.v[j1];
            const double v1_new = vj1 - (v2 - v1);
            const bool lowers = v1_new < vj1;
            const bool budget = rr_cnt < current * (unsigned)n;
            if (budget && !lowers && i0 >= 0 && j2 >= 0) { j1 = j2; i0 = L.y[j2]; }
            y7t_sync(ex);
            if (budget) {
                if (i0 >= 0) {
                    if (lowers) { --current; if (tid == 0) L.fr[current] = i0; }
                    else { if (tid == 0) L.fr[new_free] = i0; ++new_free; }
                }
                if (lowers && tid == 0) L.v[j1] = v1_new;
            } else if (i0 >= 0) {
                if (tid == 0) L.fr[new_free] = i0;
                ++new_free;
            }
            if (tid == 0) { L.x[free_i] = j1; L.y[j1] = free_i; }
            y7t_sync(ex);
        }
        n_free = new_free;
    }
    Y7T_LPROF(3);
    if (L.prof && ex.tid == 0) L.prof[6] = n_free;
    return n_free;
}

Y7T_NOINL void y7t_lap_solve(const Y7TExec& ex, Y7TLap& L) {
    const int n = L.n, tid = ex.tid, nt = ex.nt;
    if (n <= 0) return;
    const int n_free = y7t_lap_reduce(ex, L);
    // ---- augmentation: shortest augmenting paths (lap: _ca_dense / _find_path_dense) ----
    for (int f = 0; f < n_free; ++f) {
        const int start = L.fr[f];
        for (int j = tid; j < n; j += nt) {
            L.st[j] = 0;
            L.pred[j] = start;
            L.d[j] = y7t_lap_cost(L, start, j) - L.v[j];
        }
        y7t_sync(ex);
        int final_j = -1;
        double mind = 0.0;
        while (final_j < 0) {
            // new frontier: all TODO columns at the minimum distance
            double mv = HUGE_VAL;
            int mj = 0x7fffffff;
            for (int j = tid; j < n; j += nt) {
                if (L.st[j] == 3) L.st[j] = 2;
                if (L.st[j] == 0 && y7t_lex_less(L.d[j], j, mv, mj)) { mv = L.d[j]; mj = j; }
            }
            y7t_argmin(ex, mv, mj);
            mind = mv;
            int fin = 0x7fffffff, nxt = 0x7fffffff;
            for (int j = tid; j < n; j += nt) {
                if (L.st[j] == 0 && L.d[j] == mind) {
                    L.st[j] = 1;
                    if (L.y[j] < 0) { if (j < fin) fin = j; }
                    else if (j < nxt) nxt = j;
                }
            }
            y7t_imin2(ex, fin, nxt);
            y7t_sync(ex);
            if (fin != 0x7fffffff) { final_j = fin; break; }
            // scan the frontier (it may grow while scanning)
            while (nxt != 0x7fffffff) {
                const int jc = nxt, i = L.y[jc];
                const double h = y7t_lap_cost(L, i, jc) - L.v[jc] - mind;
                fin = 0x7fffffff; nxt = 0x7fffffff;
                for (int j = tid; j < n; j += nt) {
                    int s = L.st[j];
                    if (j == jc) { L.st[j] = 3; continue; }
                    if (s == 0) {
                        const double cred = y7t_lap_cost(L, i, j) - L.v[j] - h;
                        if (cred < L.d[j]) {
                            L.d[j] = cred;
                            L.pred[j] = i;
                            if (cred == mind) {
                                if (L.y[j] < 0) { if (j < fin) fin = j; }
                                else { L.st[j] = 1; s = 1; }
                            }
                        }
                    }
                    if (s == 1 && j < nxt) nxt = j;
                }
                y7t_imin2(ex, fin, nxt);
                y7t_sync(ex);
                if (fin != 0x7fffffff) { final_j = fin; break; }
            }
        }
        // dual update of the columns that were READY before the last frontier
        for (int j = tid; j < n; j += nt) if (L.st[j] == 2) L.v[j] += L.d[j] - mind;
        y7t_sync(ex);
        if (tid == 0) {
            int i = -1, j = final_j;
            while (i != start) {
                i = L.pred[j];
                L.y[j] = i;
                const int t = j;
                j = L.x[i];
                L.x[i] = t;
            }
        }
        y7t_sync(ex);
    }
    Y7T_LPROF(4);
}

// ---------------------------------------------------------------------------------------------
// lapjv.cpp LITERALLY (lap 0.4.0: _ccrrt_dense, _carr_dense x 2, _ca_dense with its `cols` permutation, find / scan order and "last free column of
// the frontier" rule) on the implicit extended matrix.  Which optimum lapjv returns when there are several is a property of exactly these orders, so the
// re-solve of a problem with ties runs them as written.  Rare (about one frame in two thousand of the random-scene tests).
//
// Round 6: the whole workgroup runs it.  Through round 5 everything order-dependent ran on thread 0, one dependent global load per matrix element: 9 ms at
// 80 + 80 rows + columns, 35 ms at 160 + 160, 218 ms at 400 + 400 (profiles/r06_large_components.txt) -- a single tied pair in a crowded frame stalled the stream
// for a fifth of a second.  Now:
//   * the reduction phases are y7t_lap_reduce (workgroup-wide scans, ties to the lowest index = the sequential scans);
//   * in the augmentation, what lapjv does PER COLUMN (d[j] = c - v, the relaxation cred < d[j]) is done by all threads, each on a contiguous run of positions
//     of `cols`; what depends on the ORDER of the scan -- _find_dense's running minimum with its swaps, _scan_dense's swaps of the columns that reach the
//     minimum and its stop at the first free one -- is reduced to the positions where something happens ("events": d <= the minimum of everything before it,
//     found with an exclusive prefix minimum; cred == mind) and replayed by thread 0 in position order.  A swap writes position k and a position in front of it,
//     so when the sequential scan reaches a position it still holds the column it held when the scan began: the events found on the unswapped `cols` are the
//     sequential scan's events, and replaying them alone gives its permutation.  (The relaxations behind the position where _scan_dense stops are applied here and
//     not there: they touch d / pred of columns that are neither READY nor on the path, which nothing reads before the next free row re-initialises them.)
// One thread (the host build) runs the same text: its single run of positions is the whole scan.
// Work arrays as y7t_lap_solve (n = nr + nc each): st = `cols`; cnt = the event list.
// ---------------------------------------------------------------------------------------------
#ifndef Y7T_COUNT_LITERAL
#if Y7T_DEVICE
static __device__ int y7t_g_literal_calls = 0;      // how many assignments met a tie and were re-solved literally (y7t_lap_literal_calls())
#define Y7T_COUNT_LITERAL() do { if (ex.tid == 0) atomicAdd(&y7t_g_literal_calls, 1); } while (0)
#else
#define Y7T_COUNT_LITERAL() do { } while (0)      // (the CPU test build counts how often ties are met)
#endif
#endif
Y7T_NOINL void y7t_lap_solve_literal(const Y7TExec& ex, Y7TLap& L) {
    const int n = L.n, tid = ex.tid, nt = ex.nt;
    if (n <= 0) return;
    Y7T_COUNT_LITERAL();
    const int n_free = y7t_lap_reduce(ex, L);
    int* x = L.x; int* y = L.y; int* fr = L.fr; int* pred = L.pred; int* cols = L.st; int* evl = L.cnt;
    double* v = L.v; double* d = L.d;
    for (int f = 0; f < n_free; ++f) {                         // _ca_dense / _find_path_dense
        const int start = fr[f];
        int lo = 0, hi = 0, n_ready = 0, final_j = -1;         // (uniform: every thread keeps its own copy)
        for (int j = tid; j < n; j += nt) { cols[j] = j; pred[j] = start; d[j] = y7t_lap_cost(L, start, j) - v[j]; }
        y7t_sync(ex);
        while (final_j == -1) {
            if (lo == hi) {
                n_ready = lo;
                {                                              // _find_dense: hi = lo + 1; mind = d[cols[lo]]; for k in (lo, n): if d[cols[k]] <= mind: (if <: hi = lo, mind = it); swap cols[k] <-> cols[hi++]
                    const int cnt = n - lo, C = (cnt + nt - 1) / nt, k0 = lo + tid * C, k1 = (k0 + C < n) ? k0 + C : n;
                    double cmin = HUGE_VAL;
                    for (int k = k0; k < k1; ++k) { const double t = d[cols[k]]; cmin = t < cmin ? t : cmin; }
                    double run = y7t_block_excl_min_d(ex, cmin);      // the minimum over the positions in front of this thread's run
                    int ne = 0;
                    for (int k = k0; k < k1; ++k) { const double t = d[cols[k]]; if (k > lo && t <= run) ++ne; run = t < run ? t : run; }
                    int total;
                    int off = y7t_block_excl_sum_i(ex, ne, total);
                    run = y7t_block_excl_min_d(ex, cmin);      // (recomputed rather than kept: one register pair less across the barriers above)
                    for (int k = k0; k < k1; ++k) { const double t = d[cols[k]]; if (k > lo && t <= run) evl[off++] = k; run = t < run ? t : run; }
                    y7t_sync(ex);
                    int h = lo + 1;
                    if (tid == 0) {
                        double mind = d[cols[lo]];
                        for (int e = 0; e < total; ++e) {
                            const int k = evl[e], j = cols[k];
                            if (d[j] < mind) { h = lo; mind = d[j]; }
                            cols[k] = cols[h];
                            cols[h++] = j;
                        }
                    }
                    hi = y7t_bcast_i(ex, h, 30);
                }
                {                                              // for k in [lo, hi): if y[cols[k]] < 0: final_j = cols[k]   (the LAST free column of the frontier)
                    int best = 0x7fffffff, dummy = 0x7fffffff;
                    for (int k = lo + tid; k < hi; k += nt) if (y[cols[k]] < 0 && -k < best) best = -k;
                    y7t_imin2(ex, best, dummy);
                    if (best != 0x7fffffff) final_j = cols[-best];
                    y7t_sync(ex);
                }
            }
            if (final_j == -1) {                               // _scan_dense (on its own copies of lo / hi: it returns without writing them back when it ends the search)
                int lo2 = lo, hi2 = hi;
                while (lo2 != hi2 && final_j == -1) {
                    const int jc = cols[lo2++];
                    const int i = y[jc];
                    const double mind = d[jc];
                    const double h = y7t_lap_cost(L, i, jc) - v[jc] - mind;
                    const int cnt = n - hi2, C = (cnt + nt - 1) / nt, k0 = hi2 + tid * C, k1 = (k0 + C < n) ? k0 + C : n;
                    int ne = 0;
                    unsigned long long hitm = 0ull;            // (runs of up to 64 positions per thread: n <= 64 * nt)
                    for (int k = k0; k < k1; ++k) {
                        const int j = cols[k];
                        const double cred = y7t_lap_cost(L, i, j) - v[j] - h;
                        if (cred < d[j]) {
                            d[j] = cred;
                            pred[j] = i;
                            if (cred == mind) { ++ne; if (k - k0 < 64) hitm |= 1ull << (k - k0); }
                        }
                    }
                    int total;
                    int off = y7t_block_excl_sum_i(ex, ne, total);
                    if (C <= 64) { for (int k = k0; k < k1; ++k) if ((hitm >> (k - k0)) & 1ull) evl[off++] = k; }
                    else if (ne) { for (int k = k0; k < k1; ++k) { const int j = cols[k]; if (pred[j] == i && d[j] == mind) evl[off++] = k; } }      // (a run longer than 64: a tiny workgroup on a large matrix -- re-derive the hits: relaxed by this row to exactly mind)
                    y7t_sync(ex);
                    int fj = -1, h2 = hi2;
                    if (tid == 0) {
                        for (int e = 0; e < total; ++e) {
                            const int k = evl[e], j = cols[k];
                            if (y[j] < 0) { fj = j; break; }
                            cols[k] = cols[h2];
                            cols[h2++] = j;
                        }
                    }
#if Y7T_DEVICE
                    if (nt > 64) {                             // thread 0's verdict to everybody (one pair of barriers for both words)
                        __syncthreads();
                        if (tid == 0) { ex.ri[30] = fj; ex.ri[31] = h2; }
                        __syncthreads();
                        fj = ex.ri[30]; h2 = ex.ri[31];
                    } else { fj = __builtin_amdgcn_readfirstlane(fj); h2 = __builtin_amdgcn_readfirstlane(h2); }
#endif
                    hi2 = h2;
                    final_j = fj;
                }
                if (final_j == -1) { lo = lo2; hi = hi2; }
            }
        }
        {
            const double mind = d[cols[lo]];
            y7t_sync(ex);
            for (int k = tid; k < n_ready; k += nt) { const int j = cols[k]; v[j] += d[j] - mind; }
        }
        if (tid == 0) {
            int i = -1, j = final_j;
            while (i != start) {
                i = pred[j];
                y[j] = i;
                const int t = j; j = x[i]; x[i] = t;
            }
        }
        y7t_sync(ex);
    }
}

// ---------------------------------------------------------------------------------------------
// The same optimisation problem without the dummy rows/columns.  lap's extended matrix prices every unmatched row and
// column at limit/2, so its objective is  const + sum over matched pairs of (c_ij - limit):  a rectangular assignment
// where a row may also take a "null" column of cost 0 that never fills up.  Shortest augmenting paths (the augmentation
// phase of Jonker-Volgenant, started from the empty assignment with zero duals) over nr rows x (nc + 1) columns solve it
// exactly with O(nr) Dijkstra searches that usually end after one scan -- instead of the (nr+nc)^2 problem whose
// column/row-reduction phases are defeated by the ties among the dummies (measured on MI355X: ~10x fewer cycles).
// Same optimum as y7t_lap_solve; identical assignment whenever that optimum is unique.
// Work arrays: v, d (nc + 1), y, pred, st (nc + 1), x (nr) -- y7t_lap_bind(L, ws, nr + nc + 1) is large enough.
// ---------------------------------------------------------------------------------------------
Y7T_FN double y7t_sap_cost(const Y7TLap& L, int i, int j) { return j < L.nc ? L.c[(size_t)i * L.ld + j] - 2.0 * L.half : 0.0; }

// Returns true when the optimum may not be unique, because which optimum lapjv returns is then a property of ITS algorithm (column reduction /
// augmenting row reduction / scan order): the caller re-solves with y7t_lap_solve_literal.  Two signs are watched:
//   * a pair costs EXACTLY cost_limit (matching it and leaving both sides unmatched cost the same) -- always;
//   * equal distances inside the search (two columns at the same shortest distance, a second path of the same length) -- for problems of up to
//     Y7T_TIE_FULL_N rows + columns: the association of the unconfirmed tracks, whose boxes are still the integer boxes they were created from (a
//     Kalman-filtered box meets an integer detection at an exactly equal cost practically never).  The primal-dual search produces such equalities by
//     construction too (tight edges), so this over-reports (one association in ~900 at 80 objects, one in four at 500), and the literal re-solve is
//     serial: ~5 ms at 160 rows + columns -- fine for a 20-column problem, not for the frame's main association.  Beyond the limit, equal-cost
//     alternatives are resolved to the lowest column index.
// IoU costs of integer boxes are small rationals (1 - 368/920 == 0.6; two detections at the same IoU from a fresh track; 5/6 + 1/3 == 2/3 + 1/2):
// about one frame in 2000 of the random-scene tests has such a tie.
#ifndef Y7T_TIE_REASON
#define Y7T_TIE_REASON(k) do { } while (0)
#endif
#define Y7T_TIE_FULL_N 64      // problems up to this many rows + columns are also checked for equal-distance ties inside the search
#define Y7T_TIE_EPS 1e-11      // two costs / path lengths closer than this are treated as tied (IoU rationals differ by >= ~1e-8; rounding noise is ~1e-16)
Y7T_FN bool y7t_near(double a, double b) { const double d = a - b; return d <= Y7T_TIE_EPS && d >= -Y7T_TIE_EPS; }

template <bool small>      // small: also watch for equal distances inside the search (problems of up to Y7T_TIE_FULL_N rows + columns)
Y7T_NOINL bool y7t_lap_solve_sap_t(const Y7TExec& ex, Y7TLap& L) {
    const int nr = L.nr, nc = L.nc, ncol = nc + 1, tid = ex.tid, nt = ex.nt;
    Y7T_LPROF(0);
    int* tie = L.fr + nr;         // (fr holds nr row candidates; the work arrays are nr + nc long)
    if (tid == 0) *tie = 0;
    for (int j = tid; j < ncol; j += nt) { L.v[j] = 0.0; L.y[j] = -1; }
    for (int i = tid; i < nr; i += nt) L.x[i] = -1;
    y7t_sync(ex);
    // ---- census of the CANDIDATE edges (c <= cost_limit, i.e. reduced cost <= 0): IoU cost matrices are sparse -- a track
    // overlaps a handful of detections -- so most rows can be settled without a search:
    //   * a row without candidates takes the null column;
    //   * a row whose only candidate column has no other candidate row is matched to it (strictly negative reduced cost), with
    //     the column price v = c - cost_limit that makes the edge tight.
    // Both are forced in every optimal solution (the objective separates over connected components of the candidate graph) and
    // leave a feasible dual with complementary slackness, so the shortest-augmenting-path loop below simply skips those rows.
    const double lim = 2.0 * L.half, lim_hi = lim + Y7T_TIE_EPS;
    int* rowcnt = L.cnt;          // [nr]
    int* colcnt = L.cnt + nr;     // [nc]
    int* rowcand = L.fr;          // [nr]
    for (int j = tid; j < nc; j += nt) {
        int k = 0;
        for (int i = 0; i < nr; ++i) k += (L.c[(size_t)i * L.ld + j] <= lim);
        colcnt[j] = k;
    }
    for (int i = tid; i < nr; i += nt) {
        int k = 0, cand = -1;
        for (int j = 0; j < nc; ++j) {
            const double c = L.c[(size_t)i * L.ld + j];
            if (c <= lim_hi) {
                if (c <= lim) { ++k; cand = j; }
                if (c >= lim - Y7T_TIE_EPS) { *tie = 1; Y7T_TIE_REASON(0); }
            }
        }
        rowcnt[i] = k;
        rowcand[i] = cand;
    }
    y7t_sync(ex);
    if (*tie) return true;
    for (int i = tid; i < nr; i += nt) {
        if (rowcnt[i] == 0) L.x[i] = nc;
        else if (rowcnt[i] == 1 && colcnt[rowcand[i]] == 1) {
            const int j = rowcand[i];
            const double red = L.c[(size_t)i * L.ld + j] - lim;
            if (red < 0.0) { L.x[i] = j; L.y[j] = i; L.v[j] = red; }
        }
    }
    y7t_sync(ex);
    for (int start = 0; start < nr; ++start) {
        if (L.x[start] != -1) continue;   // settled by the census (uniform: every thread reads the same word)
        for (int j = tid; j < ncol; j += nt) {
            L.st[j] = 0;
            L.pred[j] = start;
            L.d[j] = y7t_sap_cost(L, start, j) - L.v[j];
        }
        y7t_sync(ex);
        int final_j = -1;
        double mind = 0.0;
        while (final_j < 0) {
            double mv = HUGE_VAL;
            int mj = 0x7fffffff;
            for (int j = tid; j < ncol; j += nt) {
                if (L.st[j] == 3) L.st[j] = 2;
                if (L.st[j] == 0 && y7t_lex_less(L.d[j], j, mv, mj)) { mv = L.d[j]; mj = j; }
            }
            y7t_argmin(ex, mv, mj);
            mind = mv;
            int fin = 0x7fffffff, nxt = 0x7fffffff;
            for (int j = tid; j < ncol; j += nt) {
                if (small && L.st[j] == 0 && j != mj && y7t_near(L.d[j], mind)) { *tie = 1; Y7T_TIE_REASON(1); }      // two columns at the same shortest distance: equal-cost alternatives
                if (L.st[j] == 0 && L.d[j] == mind) {
                    L.st[j] = 1;
                    if (j == nc || L.y[j] < 0) { if (j < fin) fin = j; }
                    else if (j < nxt) nxt = j;
                }
            }
            y7t_imin2(ex, fin, nxt);
            y7t_sync(ex);
            if (fin != 0x7fffffff) { final_j = fin; break; }
            while (nxt != 0x7fffffff) {
                const int jc = nxt, i = L.y[jc];
                const double h = y7t_sap_cost(L, i, jc) - L.v[jc] - mind;
                fin = 0x7fffffff; nxt = 0x7fffffff;
                for (int j = tid; j < ncol; j += nt) {
                    int sj = L.st[j];
                    if (j == jc) { L.st[j] = 3; continue; }
                    if (sj == 0) {
                        const double cred = y7t_sap_cost(L, i, j) - L.v[j] - h;
                        if (small && y7t_near(cred, L.d[j]) && cred <= L.d[nc] + Y7T_TIE_EPS) { *tie = 1; Y7T_TIE_REASON(2); }      // a second path of the same length to a column that can still matter
                        if (cred < L.d[j]) {
                            L.d[j] = cred;
                            L.pred[j] = i;
                            if (small && y7t_near(cred, mind)) { *tie = 1; Y7T_TIE_REASON(3); }       // a further column at the distance being settled
                            if (cred == mind) {
                                if (j == nc || L.y[j] < 0) { if (j < fin) fin = j; }
                                else { L.st[j] = 1; sj = 1; }
                            }
                        }
                    }
                    if (sj == 1 && j < nxt) nxt = j;
                }
                y7t_imin2(ex, fin, nxt);
                y7t_sync(ex);
                if (fin != 0x7fffffff) { final_j = fin; break; }
            }
        }
        for (int j = tid; j < ncol; j += nt) if (L.st[j] == 2) L.v[j] += L.d[j] - mind;
        y7t_sync(ex);
        if (tid == 0) {
            int i = -1, j = final_j;
            while (i != start) {
                i = L.pred[j];
                if (j != nc) L.y[j] = i;      // the null column never fills up
                const int t = j;
                j = L.x[i];
                L.x[i] = t;
            }
        }
        y7t_sync(ex);
    }
    // report like lap: x[i] = column or -1 (x == nc means "took the null column")
    for (int i = tid; i < nr; i += nt) if (L.x[i] >= nc) L.x[i] = nc + nr;   // >= nc reads as unmatched for the callers
    y7t_sync(ex);
    Y7T_LPROF(4);
    y7t_sync(ex);
    return *tie != 0;
}

Y7T_FN bool y7t_lap_solve_sap(const Y7TExec& ex, Y7TLap& L) {
#ifdef Y7T_ALWAYS_LITERAL      // experiments: every assignment by lapjv.cpp run literally
    return true;
#endif
    return (L.nr + L.nc <= Y7T_TIE_FULL_N) ? y7t_lap_solve_sap_t<true>(ex, L) : y7t_lap_solve_sap_t<false>(ex, L);
}

