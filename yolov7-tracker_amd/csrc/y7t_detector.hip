// y7t_detector.hip -- plan executor + C ABI of the detector half of the hot path.
// The launch list replaces the Python module walk of /root/reference/models/yolo.py:321-351 (forward_once): the graph
// is lowered once on the host, then every frame batch is 110-odd back-to-back kernel launches with no host logic,
// no allocation and no synchronisation in between (capturable in a hipGraph).
#include "y7t_common.h"
#include "y7t_det.h"
#include <string.h>
#include <vector>

static_assert(sizeof(y7t_op) == 136, "y7t_op layout must match detector/graph.py OP_DTYPE");

struct y7t_det {
    std::vector<y7t_op> ops;
    std::vector<int64_t> bufs;
    char* arena; size_t arena_bytes;
    const _Float16* w; const float* bias;
    _Float16* zeros;
    float* splitk_ws = nullptr;
    int* tile_ctr = nullptr;      // Y7T_TILE_CTR_INTS ints per op, zero between launches: the tile counters of the persistent conv kernels (Y7TConvArgs::tile_ctr)
    int max_batch;
    // single flight: the tile counters, the split-K slabs and the arena belong to ONE forward at a time.  Launches on one stream are ordered anyway; a
    // forward issued on ANOTHER stream than the previous one first waits for the event the previous one left (not while a stream is being captured into a hipGraph:
    // a captured list is replayed by its owner, who orders the replays).
    hipEvent_t done_ev = nullptr;
    hipStream_t last_stream = nullptr;
    bool has_last = false;
    // Detect levels (y7t_det_set_detect)
    int nl = 0, na = 0, no = 0;
    float stride[4] = {0, 0, 0, 0}, anchors[24] = {0};
};

struct Y7TFused { float conf_thres; int cap; Y7TCandWs ws; };

extern "C" int y7t_det_create(const y7t_op* ops, int n_ops, const int64_t* bufs, int n_bufs, void* arena, size_t arena_bytes,
                              const void* w, const void* bias, int max_batch, y7t_det** out) {
    Y7T_ARG_CHECK(ops && n_ops > 0 && bufs && n_bufs > 0 && arena && w && bias && out && max_batch > 0);
    for (int i = 0; i < n_ops; ++i) {
        Y7T_ARG_CHECK(ops[i].in_buf >= 0 && ops[i].in_buf < n_bufs && ops[i].out_buf >= 0 && ops[i].out_buf < n_bufs);
        Y7T_ARG_CHECK(ops[i].type >= Y7T_OP_CONV && ops[i].type <= Y7T_OP_MAXPOOL);
    }
    for (int i = 0; i < n_bufs; ++i) Y7T_ARG_CHECK(bufs[i] >= 0 && (size_t)bufs[i] < arena_bytes && bufs[i] % 256 == 0);
    y7t_det* d = new y7t_det();
    d->ops.assign(ops, ops + n_ops);
    d->bufs.assign(bufs, bufs + n_bufs);
    d->arena = (char*)arena; d->arena_bytes = arena_bytes;
    d->w = (const _Float16*)w; d->bias = (const float*)bias; d->max_batch = max_batch;
    d->zeros = nullptr;
    if (hipMalloc((void**)&d->zeros, 256) != hipSuccess || hipMemset(d->zeros, 0, 256) != hipSuccess) {
        delete d;
        y7t_set_error("y7t_det_create: cannot allocate the zero page");
        return Y7T_E_HIP;
    }
    if (hipMalloc((void**)&d->splitk_ws, Y7T_SPLITK_WS_BYTES) != hipSuccess) {     // this detector's split-K slabs (small-batch launches)
        (void)hipFree(d->zeros);
        delete d;
        y7t_set_error("y7t_det_create: cannot allocate the split-K workspace");
        return Y7T_E_HIP;
    }
    if (hipMalloc((void**)&d->tile_ctr, sizeof(int) * Y7T_TILE_CTR_INTS * (size_t)n_ops) != hipSuccess || hipMemset(d->tile_ctr, 0, sizeof(int) * Y7T_TILE_CTR_INTS * (size_t)n_ops) != hipSuccess) {
        (void)hipFree(d->zeros);
        (void)hipFree(d->splitk_ws);
        if (d->tile_ctr) (void)hipFree(d->tile_ctr);
        delete d;
        y7t_set_error("y7t_det_create: cannot allocate the tile counters");
        return Y7T_E_HIP;
    }
    // the memsets above ran on the legacy stream, which does not order against non-blocking streams: the counters are zero before any forward can start
    if (hipDeviceSynchronize() != hipSuccess || hipEventCreateWithFlags(&d->done_ev, hipEventDisableTiming) != hipSuccess) {
        (void)hipFree(d->zeros); (void)hipFree(d->splitk_ws); (void)hipFree(d->tile_ctr);
        delete d;
        y7t_set_error("y7t_det_create: cannot create the forward-done event");
        return Y7T_E_HIP;
    }
    *out = d;
    return 0;
}

extern "C" int y7t_det_destroy(y7t_det* d) {
    if (!d) return 0;
    if (d->tile_ctr) (void)hipFree(d->tile_ctr);
    if (d->done_ev) (void)hipEventDestroy(d->done_ev);
    if (d->zeros) (void)hipFree(d->zeros);
    if (d->splitk_ws) (void)hipFree(d->splitk_ws);
    delete d;
    return 0;
}

extern "C" int y7t_det_num_ops(const y7t_det* d) { return d ? (int)d->ops.size() : Y7T_E_ARG; }

extern "C" int y7t_det_forward(y7t_det* d, int B, y7t_stream stream) { return y7t_det_forward_ops(d, B, 0, -1, stream); }

static int forward_impl(y7t_det* d, int B, int first, int last, const Y7TFused* fused, hipStream_t s) {
    Y7T_ARG_CHECK(d && B > 0 && B <= d->max_batch);
    if (last < 0) last = (int)d->ops.size();
    Y7T_ARG_CHECK(first >= 0 && first <= last && last <= (int)d->ops.size());
    hipStreamCaptureStatus cap_st = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(s, &cap_st) == hipSuccess && cap_st != hipStreamCaptureStatusNone;
    if (!capturing && d->has_last && d->last_stream != s) Y7T_HIP_CHECK(hipStreamWaitEvent(s, d->done_ev, 0));      // single flight per detector
    struct Done {      // every exit of this function (also a failed launch) leaves the event behind what has been enqueued
        y7t_det* d; hipStream_t s; bool on;
        ~Done() { if (on && hipEventRecord(d->done_ev, s) == hipSuccess) { d->last_stream = s; d->has_last = true; } }
    } done{d, s, !capturing};
    for (int oi = first; oi < last; ++oi) {
        const y7t_op& op = d->ops[oi];
        const _Float16* in = (const _Float16*)(d->arena + d->bufs[op.in_buf]);
        void* outp = d->arena + d->bufs[op.out_buf];
        int rc = 0;
        if (op.type == Y7T_OP_CONV) {
            Y7TConvArgs a;
            memset(&a, 0, sizeof(a));
            a.in = in; a.ldin = op.in_ld; a.cin_off = op.in_coff; a.B = B; a.H = op.H; a.W = op.W; a.Cin = op.Cin;
            a.w = d->w + op.w_off; a.bias = d->bias + op.bias_off;
            a.out = outp; a.ldout = op.out_ld; a.cout_off = op.out_coff; a.out_f32 = op.out_f32;
            a.Ho = op.Ho; a.Wo = op.Wo; a.Cout = op.Cout; a.Cout_pad = op.Cout_pad;
            a.KH = op.KH; a.KW = op.KW; a.stride = op.stride; a.pad = op.pad; a.K = op.K; a.K_pad = op.K_pad;
            a.M = B * op.Ho * op.Wo; a.act = op.act; a.zeros = d->zeros; a.korder = op.korder; a.force_patch = 0;
            a.splitk_ws = d->splitk_ws;
            a.tile_ctr = d->tile_ctr + Y7T_TILE_CTR_INTS * oi;
            if (op.up_C > 0) {
                a.in2 = (const _Float16*)(d->arena + d->bufs[op.up_buf]);
                a.ldin2 = op.up_ld; a.cin2_off = op.up_coff; a.up_c0 = op.up_c0; a.up_C = op.up_C;
            }
            if (fused && op.detect_level >= 0) {
                const int l = op.detect_level;
                if (l >= d->nl || op.Cout != d->na * d->no) { y7t_set_error("fused Detect: level %d not described by y7t_det_set_detect", l); return Y7T_E_STATE; }
                a.epi = 1;
                Y7TDecode& q = a.dec;
                q.stride = d->stride[l]; q.ny = op.Ho; q.nx = op.Wo; q.na = d->na; q.no = d->no;
                q.row0 = 0;
                for (int k = 0; k < (int)d->ops.size(); ++k)    // rows of the finer levels come first (models/yolo.py:57 cat order)
                    if (d->ops[k].type == Y7T_OP_CONV && d->ops[k].detect_level >= 0 && d->ops[k].detect_level < l)
                        q.row0 += d->na * d->ops[k].Ho * d->ops[k].Wo;
                for (int k = 0; k < d->na; ++k) { q.aw[k] = d->anchors[(l * d->na + k) * 2]; q.ah[k] = d->anchors[(l * d->na + k) * 2 + 1]; }
                q.conf_thres = fused->conf_thres; q.cap = fused->cap;
                q.cbox = fused->ws.cbox; q.cscore = fused->ws.cscore; q.ccls = fused->ws.ccls; q.cidx = fused->ws.cidx; q.count = fused->ws.count;
            }
            // frames per launch: the kernels address a tensor through 32-bit byte offsets (2 GiB); a batch whose input or output tensor is larger goes out as several launches
            // over consecutive runs of frames (batch-major NHWC: a run of frames is a contiguous piece of every tensor of the op) -- the 640^2 / 320^2 layers of w6 @ 1280
            // above 40 frames, thousands of tiles each, so nothing is lost to the extra launch; the tile counters are zero again when a launch ends
            const long long f_in = (long long)op.H * op.W * op.in_ld * 2, f_out = (long long)op.Ho * op.Wo * op.out_ld * (op.out_f32 ? 4 : 2);
            const long long f_in2 = op.up_C > 0 ? (long long)(op.H / 2) * (op.W / 2) * op.up_ld * 2 : 0;
            const long long f_max = f_in > f_out ? f_in : f_out, lim = (1ll << 31) - 1;
            if (f_max > lim) { y7t_set_error("conv: one frame of op %d is larger than 2 GiB", oi); return Y7T_E_ARG; }
            const int n_runs = (int)(((long long)B * f_max + lim - 1) / lim), run = (B + n_runs - 1) / n_runs;
            if (n_runs > 1 && a.epi) { y7t_set_error("fused Detect: op %d would need %d launches per batch (candidate rows are indexed by the frame of the launch)", oi, n_runs); return Y7T_E_ARG; }
            for (int b0 = 0; b0 < B && rc == 0; b0 += run) {
                const int nb = B - b0 < run ? B - b0 : run;
                a.in = (const _Float16*)((const char*)in + b0 * f_in);
                a.out = (char*)outp + b0 * f_out;
                if (op.up_C > 0) a.in2 = (const _Float16*)(d->arena + d->bufs[op.up_buf] + b0 * f_in2);
                a.B = nb; a.M = nb * op.Ho * op.Wo;
                rc = y7t_conv_launch(a, s);
            }
        } else if (op.type == Y7T_OP_UPSAMPLE2X) {
            rc = y7t_upsample_launch(in, op.in_ld, op.in_coff, B, op.H, op.W, op.Cin, (_Float16*)outp, op.out_ld, op.out_coff, s);
        } else {
            // SPPCSPC (models/common.py:262-280): three 5 x 5 / 1 pools in cascade, each reading its predecessor's slice and writing the next slice of the same concat
            // buffer -> one launch that keeps the slab in LDS (when the whole chain is inside the requested op range)
            auto chained = [&](const y7t_op& a, const y7t_op& b2) {
                return b2.type == Y7T_OP_MAXPOOL && b2.KH == 5 && b2.stride == 1 && b2.pad == 2 && b2.H == a.H && b2.W == a.W && b2.Cin == a.Cin && b2.in_buf == a.out_buf &&
                       b2.in_ld == a.out_ld && b2.in_coff == a.out_coff && b2.out_buf == a.out_buf && b2.out_ld == a.out_ld && b2.out_coff == a.out_coff + a.Cin;
            };
            rc = 1;
            if (op.KH == 5 && op.stride == 1 && op.pad == 2 && oi + 2 < last && chained(op, d->ops[oi + 1]) && chained(d->ops[oi + 1], d->ops[oi + 2]) &&
                (op.in_buf != op.out_buf || op.in_coff + op.Cin <= op.out_coff || op.out_coff + 3 * op.Cin <= op.in_coff))
                rc = y7t_spp3_try(in, op.in_ld, op.in_coff, B, op.H, op.W, op.Cin, (_Float16*)outp, op.out_ld, op.out_coff, s);
            if (rc == 0) oi += 2;
            else if (rc == 1)
                rc = y7t_maxpool_launch(in, op.in_ld, op.in_coff, B, op.H, op.W, op.Cin, op.KH, op.stride, op.pad, (_Float16*)outp, op.out_ld,
                                        op.out_coff, s);
        }
        if (rc) return rc;
    }
    return 0;
}

extern "C" int y7t_det_forward_ops(y7t_det* d, int B, int first, int last, y7t_stream stream) {
    return forward_impl(d, B, first, last, nullptr, (hipStream_t)stream);
}

static bool stem_fusable(const y7t_det* d) {
    if (d->ops.empty()) return false;
    const y7t_op& op = d->ops[0];
    return op.type == Y7T_OP_CONV && op.in_buf == 0 && op.in_ld == 16 && op.in_coff == 0 && op.Cin == 16 && op.KH == 3 && op.KW == 3 && op.stride == 1 &&
           op.pad == 1 && op.Cout == 64 && op.Cout_pad == 64 && !op.out_f32 && op.korder == 0 && op.up_C == 0 && op.detect_level < 0 &&
           op.out_ld % 8 == 0 && op.out_coff % 8 == 0 && op.H % 16 == 0 && op.W % 16 == 0;
}

extern "C" int y7t_det_stem_fusable(const y7t_det* d) { return d && stem_fusable(d) ? 1 : 0; }

extern "C" int y7t_det_forward_stem_u8(y7t_det* d, const void* frames_u8, int B, int H0, int W0, int new_h, int new_w, int top, int left, y7t_stream stream) {
    Y7T_ARG_CHECK(d && frames_u8 && B > 0 && B <= d->max_batch && H0 > 0 && W0 > 0 && new_h > 0 && new_w > 0 && top >= 0 && left >= 0);
    if (!stem_fusable(d)) { y7t_set_error("the plan's first op is not a ReOrg + 3x3 -> 64 stem: use y7t_input_layout + y7t_det_forward"); return Y7T_E_STATE; }
    const y7t_op& op = d->ops[0];
    const int H = op.H * 2, W = op.W * 2;
    Y7T_ARG_CHECK(top + new_h <= H && left + new_w <= W);
    // (frames per launch as in forward_impl: the stem's output tensor passes 2 GiB at 41 frames of 1280 x 1280)
    const long long f_out = (long long)op.H * op.W * op.out_ld * 2, f_in = (long long)H0 * W0 * 3, lim = (1ll << 31) - 1;
    const long long f_max = f_out > f_in ? f_out : f_in;
    Y7T_ARG_CHECK(f_max <= lim);
    const int n_runs = (int)(((long long)B * f_max + lim - 1) / lim), run = (B + n_runs - 1) / n_runs;
    for (int b0 = 0; b0 < B; b0 += run) {
        const int nb = B - b0 < run ? B - b0 : run;
        if (int rc = y7t_stem_u8_launch((const char*)frames_u8 + b0 * f_in, nb, H0, W0, H, W, new_h, new_w, top, left, d->w + op.w_off, op.K_pad, d->bias + op.bias_off,
                                        (_Float16*)(d->arena + d->bufs[op.out_buf] + b0 * f_out), op.out_ld, op.out_coff, op.act, (hipStream_t)stream)) return rc;
    }
    return 0;
}

extern "C" int y7t_det_set_detect(y7t_det* d, int nl, int na, int no, const float* strides, const float* anchors) {
    Y7T_ARG_CHECK(d && nl >= 1 && nl <= 4 && na >= 1 && na <= 3 && no >= 6 && strides && anchors);
    d->nl = nl; d->na = na; d->no = no;
    for (int l = 0; l < nl; ++l) d->stride[l] = strides[l];
    for (int i = 0; i < nl * na * 2; ++i) d->anchors[i] = anchors[i];
    return 0;
}

extern "C" int y7t_det_forward_fused(y7t_det* d, int B, int first, int last, float conf_thres, int cap, int max_nms, void* ws, size_t ws_bytes,
                                     y7t_stream stream) {
    Y7T_ARG_CHECK(d && ws && cap >= 64 && max_nms >= 1 && B > 0 && B <= d->max_batch);
    Y7T_ARG_CHECK(d->nl > 0);
    Y7T_ARG_CHECK(ws_bytes >= y7t_post_ws_bytes(B, cap, max_nms));
    Y7TFused f;
    f.conf_thres = conf_thres; f.cap = cap; f.ws = y7t_post_cand_ws(ws, B, cap);
    if (first == 0) Y7T_HIP_CHECK(hipMemsetAsync(f.ws.count, 0, sizeof(int) * B, (hipStream_t)stream));
    return forward_impl(d, B, first, last, &f, (hipStream_t)stream);
}

extern "C" size_t y7t_det_postprocess_workspace_bytes(int B, int cap, int max_nms) {
    return (B > 0 && cap > 0 && max_nms > 0) ? y7t_post_ws_bytes(B, cap, max_nms) : 0;
}

extern "C" int y7t_det_postprocess(const float* const* head, const int* ny, const int* nx, const float* strides, const float* anchors, int nl,
                                   int na, int no, int B, float conf_thres, float iou_thres, int max_det, int max_nms, int cap,
                                   const float* letterbox, float* dets, int* ndets, int* keep_idx, int* cand_count, void* ws, size_t ws_bytes,
                                   y7t_stream stream) {
    Y7T_ARG_CHECK(letterbox && dets && ndets && keep_idx && ws);
    Y7T_ARG_CHECK(head == nullptr || (ny && nx && strides && anchors));
    Y7T_ARG_CHECK(nl >= 1 && nl <= 4 && na >= 1 && na <= 3 && no >= 6 && B >= 1 && cap >= 64 && max_det >= 1 && max_nms >= 1);
    Y7T_ARG_CHECK(iou_thres >= 0.0f && conf_thres >= 0.0f);      // (conf_thres >= 0: every candidate score is positive, which the rank sort's bit-pattern keys rely on)      // (the NMS skips the division for boxes that do not intersect: exact for a threshold that IoU = 0 does not exceed)
    Y7TPostArgs a;
    memset(&a, 0, sizeof(a));
    a.predecoded = head == nullptr;
    for (int l = 0; l < nl && head; ++l) { a.head[l] = head[l]; a.ny[l] = ny[l]; a.nx[l] = nx[l]; a.stride[l] = strides[l]; }
    for (int i = 0; i < nl * na * 2 && head; ++i) a.anchors[i] = anchors[i];
    a.nl = nl; a.na = na; a.no = no; a.B = B; a.conf_thres = conf_thres; a.iou_thres = iou_thres;
    a.max_det = max_det; a.max_nms = max_nms < cap ? max_nms : cap; a.cap = cap;
    a.letterbox_dev = letterbox; a.dets = dets; a.ndets = ndets; a.keep_idx = keep_idx; a.count_out = cand_count;
    a.ws = ws; a.ws_bytes = ws_bytes;
    return y7t_post_run(a, (hipStream_t)stream);
}

extern "C" int y7t_conv2d_nhwc_f16(const void* in, int in_ld, int in_coff, int B, int H, int W, int Cin, const void* w, const float* bias, void* out,
                                   int out_ld, int out_coff, int out_f32, int Cout, int Cout_pad, int KH, int KW, int stride, int pad, int act,
                                   const void* zeros16, y7t_stream stream) {
    Y7T_ARG_CHECK(in && w && bias && out && zeros16 && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && stride > 0);
    Y7TConvArgs a;
    memset(&a, 0, sizeof(a));
    a.in = (const _Float16*)in; a.ldin = in_ld; a.cin_off = in_coff; a.B = B; a.H = H; a.W = W; a.Cin = Cin;
    a.w = (const _Float16*)w; a.bias = bias; a.out = out; a.ldout = out_ld; a.cout_off = out_coff; a.out_f32 = out_f32;
    a.Ho = (H + 2 * pad - KH) / stride + 1; a.Wo = (W + 2 * pad - KW) / stride + 1;
    a.Cout = Cout; a.Cout_pad = Cout_pad; a.KH = KH; a.KW = KW; a.stride = stride; a.pad = pad;
    a.K = KH * KW * Cin; a.K_pad = (a.K + 63) / 64 * 64; a.M = B * a.Ho * a.Wo; a.act = act & 0xff; a.zeros = (const _Float16*)zeros16;
    a.korder = (act >> 19) & 1 ? 11 : (act >> 18) & 1 ? 10 : (act >> 17) & 1 ? 9 : (act >> 16) & 1 ? 8 : (act >> 15) & 1 ? 7 : (act >> 14) & 1 ? 6 : (act >> 13) & 1 ? 5 : (act >> 12) & 1 ? 4 : (act >> 11) & 1 ? 3 : (act >> 10) & 1 ? 2 : (act >> 8) & 1;   // `act` bit 8: (kh, chunk, kw) K order; bit 10: patch-kernel panels; bit 11: 1x1 panels; bit 12: stride-2 patch-kernel panels; bit 13: register-fragment order of the weights-stationary 64 -> 64 kernel
    a.force_patch = (act >> 9) & 1;   // bit 9: force the LDS-patch kernel for an eligible 3x3/s1 layer (tests)
    return y7t_conv_launch(a, (hipStream_t)stream);
}

// ---- streams with a compute-unit mask (include/y7t.h) ----
extern "C" int y7t_stream_create_cu_mask(const uint32_t* mask_words_host, int n_words, y7t_stream* out_stream_host) {
    Y7T_ARG_CHECK(mask_words_host && n_words > 0 && n_words <= 64 && out_stream_host);
    bool any = false;
    for (int i = 0; i < n_words; ++i) any = any || mask_words_host[i] != 0;
    if (!any) { y7t_set_error("y7t_stream_create_cu_mask: empty mask"); return Y7T_E_ARG; }
    hipStream_t s = nullptr;
    Y7T_HIP_CHECK(hipExtStreamCreateWithCUMask(&s, (uint32_t)n_words, mask_words_host));
    *out_stream_host = (y7t_stream)s;
    return 0;
}

extern "C" int y7t_stream_destroy(y7t_stream stream) {
    Y7T_ARG_CHECK(stream);
    Y7T_HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
    return 0;
}
