/*
 * y7t.h -- C ABI of liby7t.so, the MI355X (gfx950) implementation of the Yolov7-tracker hot path.
 *
 * The reference (JackWoo0831/Yolov7-tracker) is pure Python and has no FFI of its own; its
 * "operator API" for this path is the set of Python call sites cited on each function below
 * (paths relative to the reference repo).  This header is what a binding for those call sites
 * binds; INTEGRATION.md shows the ctypes stubs a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is caller-owned; pointers are DEVICE pointers unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous
 *     with respect to the host unless stated otherwise;
 *   - return value: 0 = OK, < 0 = error (message via y7t_last_error(), thread-local);
 *   - no torch types, no exceptions across the boundary, no hidden global state except the
 *     explicit id counter object that mirrors BaseTrack._count.
 */
#ifndef Y7T_H
#define Y7T_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* y7t_stream;

enum { Y7T_OK = 0, Y7T_E_ARG = -1, Y7T_E_HIP = -2, Y7T_E_CAPACITY = -3, Y7T_E_STATE = -4 };

/* Kalman filter kinds == KALMAN_DICT keys, tracker/basetrack.py:64-69 */
enum { Y7T_KALMAN_DEFAULT = 0, Y7T_KALMAN_NAIVE = 1, Y7T_KALMAN_BOTSORT = 2, Y7T_KALMAN_STRONGSORT = 3 };
/* tracker kinds == TRACKER_DICT keys implemented on the device, tracker/track.py:56-65 */
enum { Y7T_TRACKER_SORT = 0, Y7T_TRACKER_BYTETRACK = 1, Y7T_TRACKER_BOTSORT = 2, Y7T_TRACKER_DEEPSORT = 3 };

const char* y7t_last_error(void);
int y7t_version(void);
/* diagnostics: name of the kernel variant the last launch made by this thread selected, e.g. "patch<16,16,128>", "igemm<128,128,32,2> 1x1",
 * "patch_strip<42,128>" (the conv dispatch rules -- tile shape, LDS-patch vs generic, split-K -- live inside the library and depend on
 * the batch size; tests and bench.py read the launch list of a plan back through y7t_det_forward_ops(i, i+1) + this) */
const char* y7t_last_kernel(void);
/* number of HIP devices visible (0 when there is none); host-synchronous */
int y7t_device_count(void);
/* A HIP stream restricted to a set of compute units (bit i of mask_words_host = CU i; hipExtStreamCreateWithCUMask).  The reference runs the detector
 * and `tracker.update` one after the other on one stream (tracker/track.py:144-151); the pipelined form of that loop (bench.py, track.py --batch)
 * runs the tracker's single-workgroup frame steps beside the next batch's convolutions, and gives them compute units of their own so that they do
 * not queue behind / share issue slots with 256-thread convolution workgroups.  The stream is an ordinary hipStream_t for every other call here. */
int y7t_stream_create_cu_mask(const uint32_t* mask_words_host, int n_words, y7t_stream* out_stream_host);
int y7t_stream_destroy(y7t_stream stream);

/* ---------------------------------------------------------------- tracker math (float64) ---- */

/* matching.iou_distance -> ious -> cython_bbox.bbox_overlaps, tracker/matching.py:44-82.
 * cost[i*m + j] = 1 - IoU(a_i, b_j) with the "+1 pixel" convention. a: n x 4 tlbr, b: m x 4. */
int y7t_iou_cost_f64(const double* a_tlbr, int n, const double* b_tlbr, int m, double* cost, y7t_stream stream);

/* KalmanFilter.initiate, tracker/kalman_filter.py:190-221 (xyah), :436-467 (xywh).
 * z: K x 4 measurements; mean: K x 8; cov: K x 64.  flags bit0: keep the std products in
 * float32 (what the reference does under numpy >= 2 when handed a float32 measurement). */
int y7t_kf_initiate_f64(int kind, const double* z, double* mean, double* cov, int K, int flags, y7t_stream stream);

/* KalmanFilter.multi_predict, tracker/kalman_filter.py:289-329, fused with the "zero the last
 * state component of non-Tracked tracks" step of STrack.multi_predict, tracker/basetrack.py:253-271.
 * In place on mean (N x 8) / cov (N x 64); zero_last_mask: N bytes or NULL. */
int y7t_kf_multi_predict_f64(int kind, double* mean, double* cov, const uint8_t* zero_last_mask, int N,
                             y7t_stream stream);

/* KalmanFilter.project, tracker/kalman_filter.py:260-287 (conf: N confidences for the NSA filter or NULL).
 * pmean: N x 4, pcov: N x 16. */
int y7t_kf_project_f64(int kind, const double* mean, const double* cov, const double* conf, double* pmean,
                       double* pcov, int N, y7t_stream stream);

/* KalmanFilter.update, tracker/kalman_filter.py:331-363, batched over K (track, measurement) pairs:
 * pair k updates track track_idx[k] (or track k when track_idx is NULL) with z[k] (K x 4). In place. */
int y7t_kf_update_batch_f64(int kind, double* mean, double* cov, const double* z, const int* track_idx,
                            const double* conf, int K, y7t_stream stream);

/* KalmanFilter.gating_distance(metric='maha'), tracker/kalman_filter.py:365-411:
 * out[i*M + j] = squared Mahalanobis distance of measurement j to track i. */
int y7t_kf_gating_f64(int kind, const double* mean, const double* cov, const double* z, int N, int M,
                      int only_position, double* out, y7t_stream stream);

/* matching.linear_assignment -> lap.lapjv(cost, extend_cost=True, cost_limit=limit),
 * tracker/matching.py:30-41.  cost: n x m row-major.  x[n]: column of row i or -1; y[m]: row of
 * column j or -1; opt (may be NULL): sum of the kept costs.  One workgroup on the device;
 * workspace: y7t_lapjv_workspace_bytes(n, m) bytes of device memory. */
size_t y7t_lapjv_workspace_bytes(int n, int m);
int y7t_lapjv_f64(const double* cost, int n, int m, double cost_limit, int* x, int* y, double* opt, void* workspace,
                  y7t_stream stream);
/* The same with HOST pointers (SURVEY 8b "+ _host variant": matching.linear_assignment is called with numpy arrays): cost / x / y / opt live in host
 * memory; the library stages them through device buffers it owns (grown on demand), solves on the device and returns when x / y / opt are written.
 * No CPU solver behind it: without a GPU it fails like every other entry point.  Not re-entrant (one staging allocation per process). */
int y7t_lapjv_f64_host(const double* cost_host, int n, int m, double cost_limit, int* x_host, int* y_host, double* opt_host, y7t_stream stream);
/* Diagnostics: how many assignments of this process met a tie (a pair exactly at cost_limit, equal-cost alternatives in a small problem) and were
 * re-solved with lap's lapjv.cpp run literally, because the optimum lapjv returns is then a property of its own scan order (matching.py:30-44). */
int y7t_lap_literal_calls(void);

/* ---------------------------------------------------------------- device-resident tracker ---- */
/* BaseTrack._count (tracker/basetrack.py:22,43-46): one int in device memory, shared by every
 * tracker object of the process.  The caller allocates 4 bytes and zeroes them. */

/* size in bytes of the state blob of one tracker (tracked_stracks + lost_stracks pool etc.) */
size_t y7t_tracker_state_bytes(int cap_tracks, int cap_dets);

/* BaseTracker.__init__ / ByteTrack.__init__, tracker/basetrack.py:346-367, tracker/bytetrack.py:9-17.
 * conf_thresh = opts.conf_thresh, iou_thresh = opts.iou_thresh, max_time_lost = int(frame_rate/30*track_buffer).
 * flags bit0: numpy>=2 float32 flow of freshly initiated tracks (see y7t_kf_initiate_f64). */
int y7t_tracker_init(void* state, size_t state_bytes, int tracker_kind, int kalman_kind, int cap_tracks, int cap_dets,
                     double conf_thresh, double iou_thresh, int max_time_lost, int flags, int* id_counter,
                     y7t_stream stream);

/* the end of a tracker object's life (the reference: garbage collection of the BaseTracker instance, tracker/track.py:132 creates one per sequence): call before
 * freeing or re-purposing `state`; the library forgets the host-side notes it keeps per initialised blob (tracker kind, LDS arena size) */
int y7t_tracker_release(void* state);

/* ByteTrack.update / BaseTracker.update (tracker/bytetrack.py:41-204, tracker/basetrack.py:368-487) as ONE
 * kernel launch, one workgroup per tracker.  `batch` trackers step together (independent sequences):
 *   states[b]   state blob of tracker b
 *   dets[b]     n x 6 float32 rows [x1,y1,x2,y2,conf,cls] (what post_process_v7 hands over, track.py:234-244)
 *   n_dets[b]   row count, read ON THE DEVICE (so it can come straight from NMS); < 0 = update_without_detection
 *   out_rows[b] out_cap x 8 float64 rows (track_id, x, y, w, h, cls, score, slot) of the returned tracks
 *   out_count[b] number of returned tracks
 * states/dets/n_dets/out_rows/out_count (and gmc_warps, may be NULL) are DEVICE arrays of `batch` entries. threads: 0 = default. */
int y7t_tracker_step_batch(void* const* states, const float* const* dets, const int* n_dets, double* const* out_rows,
                           int* out_count, int out_cap, int batch, int threads, const double* const* gmc_warps,
                           y7t_stream stream);

/* n_frames CONSECUTIVE frames of one tracker in one launch (the per-frame loop of tracker/track.py:138-179 for a caller that already holds a batch's
 * detections, e.g. `--batch N`): frame f = dets[f] / n_dets[f] (read on the device) -> out_rows[f], *out_count[f]; all five are DEVICE arrays of
 * n_frames entries (gmc_warps may be NULL).  Exactly the frames y7t_tracker_step would produce one by one. */
int y7t_tracker_step_frames(void* state, const float* const* dets, const int* n_dets, double* const* out_rows, int* const* out_count, int out_cap,
                            int n_frames, int threads, const double* const* gmc_warps, y7t_stream stream);

/* convenience for batch == 1 with host-known n (n < 0: update_without_detection, basetrack.py:489-537) */
int y7t_tracker_step(void* state, const float* dets, int n, double* out_rows, int out_cap, int* out_count, int threads,
                     const double* gmc_warp, y7t_stream stream);

/* BoT-SORT (tracker/botsort.py:313-493, tracker kind Y7T_TRACKER_BOTSORT, Kalman kind botsort): `gmc_warp` / `gmc_warps[b]` is the
 * 2x3 camera-motion matrix of the frame (row-major, 6 doubles in DEVICE memory; NULL = no compensation) that the reference's
 * GMC.apply returns (botsort.py:13-248 -- OpenCV ORB/RANSAC estimation, out of scope); the step applies multi_gmc
 * (botsort.py:250-269) to the predicted pool and the unconfirmed tracks.  Same op on its own: */
int y7t_kf_multi_gmc_f64(double* mean, double* cov, const double* warp, int N, y7t_stream stream);

/* DeepSORT (tracker/deepsort.py:79-227, tracker kind Y7T_TRACKER_DEEPSORT): appearance + motion.  Next to the pool blob the tracker owns
 * a FEATURE STATE (y7t_deepsort_feature_bytes, y7t_deepsort_init): per slot the last `budget` appearance vectors of the track
 * (STrack.features, basetrack.py:97-103,324-332; budget = store_features_budget = 100) plus per-frame scratch.
 * y7t_tracker_step_deepsort = DeepSORT.update for one frame:
 *   det_feats   n x feat_dim float32, row j = what DeepSORT.get_feature (deepsort.py:19-41 -> the ReID network) returns for detection row j
 *               (only rows with conf > conf_thresh are read)
 *   launches: normalise the features, nearest_embedding_distance (tracker/matching.py:105-127) for every live slot (one workgroup per
 *   slot), then ONE workgroup for matching_cascade with gate_cost_matrix (matching.py:216-277, deepsort.py:43-66), the two IoU
 *   associations and the list bookkeeping.  update_without_detection: y7t_tracker_step(state, NULL, -1, ...) as for every tracker. */
size_t y7t_deepsort_feature_bytes(int cap_tracks, int cap_dets, int feat_dim, int budget);
int y7t_deepsort_init(void* feat_state, size_t bytes, int cap_tracks, int cap_dets, int feat_dim, int budget, y7t_stream stream);
int y7t_tracker_step_deepsort(void* state, void* feat_state, int cap_tracks, const float* dets, int n, const float* det_feats, double* out_rows,
                              int out_cap, int* out_count, int threads, y7t_stream stream);

/* byte offsets of the arrays inside a state blob, for host-side views (tracked_stracks, lost_stracks, ...).
 * names/offsets: see y7t_tracker_field_name(i); returns the number of fields. */
int y7t_tracker_layout(int cap_tracks, int cap_dets, int64_t* offsets, int max_fields);
const char* y7t_tracker_field_name(int i);


/* ---------------------------------------------------------------- detector (YOLOv7 forward) ---- */
/* One op of a lowered detector graph.  The host (Python) lowers the reference's model graph
 * (models/yolo.py:321-351 forward_once over the yaml layer list, models/yolo.py:443-520 parse_model) into a
 * static launch list over a pre-planned NHWC fp16 activation arena: concat is eliminated (producers write
 * channel slices), BN is folded (utils/torch_utils.py:181-201), weights are packed [Cout_pad][K_pad] fp16. */
enum { Y7T_OP_CONV = 0, Y7T_OP_UPSAMPLE2X = 1, Y7T_OP_MAXPOOL = 2 };
typedef struct y7t_op {
    int32_t type;
    int32_t in_buf, in_ld, in_coff;     /* arena buffer id, channels of that buffer, first channel of the slice */
    int32_t H, W, Cin;                  /* input spatial size and slice channels */
    int32_t out_buf, out_ld, out_coff, out_f32;
    int32_t Ho, Wo, Cout, Cout_pad;
    int32_t KH, KW, stride, pad;        /* conv / pool window */
    int32_t K, K_pad;
    int32_t act;                        /* 0 none, 1 SiLU, 2 LeakyReLU(0.1) */
    int32_t korder;                     /* weight packing: 0 k = (kh*KW+kw)*Cin + ci; 1 (kh, 64-ch chunk, kw); 2 LDS-patch panels; 3 1x1 panels;
                                           4 stride-2 LDS-patch panels (detector/weights.py::panel_pack_s2); 5 the 64 -> 64 filter bank as MFMA A-fragments
                                           (weights-stationary kernel, pack_ws); 7 1x1 layers with Cout % 256 == 0: per (channel tile, 64-deep K-tile) the swizzled LDS image of the 256 x 64 panel
                                           (csrc/y7t_conv_p8.hip, weights.py::panel_pack_p8); 8 the 64 -> 128 3x3 / stride 2 filter bank as MFMA A-fragments
                                           (csrc/y7t_conv_ws_s2.hip, pack_ws_s2); 9 / 10 as 2 / 3 with 64-row panels although Cout_pad % 128 == 0 (small maps);
                                           11 as 8 FOLLOWED by the 128 -> 128 bank of the twin 1x1 convolution that consumes the layer (pack_ws_s2_tail) and its 128 biases
                                           behind the layer's own: one launch computes both, Cout / out_* describe the 1x1 layer's output, the tensor between them is never written */
    int32_t detect_level;               /* -1: ordinary layer.  l >= 0: the 1x1 conv of Detect level l (models/yolo.py:46); in a fused forward
                                           (y7t_det_forward_fused) its epilogue decodes + filters instead of writing the head tensor */
    int64_t w_off;                      /* element offset into the fp16 weight blob */
    int64_t bias_off;                   /* element offset into the fp32 bias blob */
    /* upsample-on-read (1x1 / stride 1 convs; cfg/deploy/yolov7-w6.yaml:75,89,103: nn.Upsample feeds only a Concat whose consumers are
     * 1x1 convs): channels [up_c0, up_c0 + up_C) of the input slice are not stored at this resolution -- pixel (y, x) reads
     * buffer up_buf (H/2 x W/2, up_ld channels, slice starting at up_coff) at (y >> 1, x >> 1).  up_C == 0: off. */
    int32_t up_buf, up_ld, up_coff, up_c0, up_C, pad0;
} y7t_op;                               /* sizeof == 136 */

typedef struct y7t_det y7t_det;

/* Build an executable plan.  ops/buf_offsets are HOST arrays (copied); arena/weights/bias are DEVICE memory owned by
 * the caller.  buf_offsets[i]: byte offset of arena buffer i, laid out for max_batch images. */
int y7t_det_create(const y7t_op* ops_host, int n_ops, const int64_t* buf_offsets_host, int n_bufs, void* arena, size_t arena_bytes,
                   const void* weights_f16, const void* bias_f32, int max_batch, y7t_det** out);
int y7t_det_destroy(y7t_det* det);

/* models/yolo.py:345 `x = m(x)` for every layer: runs the whole launch list for B images whose input layout
 * (y7t_input_layout) is already in arena buffer 0.  Asynchronous on `stream`.
 * One forward of a detector at a time: the arena, the split-K slabs and the tile counters of the persistent kernels belong to the plan.  Launches on one
 * stream are ordered; a forward (or a piece of one, y7t_det_forward_ops / _fused) issued on another stream than the previous one waits for it on the device.
 * While `stream` is being captured into a hipGraph nothing is recorded or waited for: whoever replays the graph orders the replays.
 * The kernels address a tensor through 32-bit byte offsets: a conv whose input or output tensor of B images is larger than 2 GiB goes out as several launches over
 * runs of consecutive images (w6 @ 1280 x 1280: the 640^2 / 320^2 layers from 41 images on); results do not depend on where the runs are cut. */
int y7t_det_forward(y7t_det* det, int B, y7t_stream stream);
/* the same launch list in pieces: ops [first, last) of the plan (last < 0: to the end), so that a caller can record an event between
 * two parts of Model.forward_once (models/yolo.py:321-351) -- bench.py starts the previous batch's decode+NMS on another stream once
 * the memory-bound high-resolution layers of the next forward are through.  y7t_det_num_ops: length of the list. */
int y7t_det_num_ops(const y7t_det* det);
int y7t_det_forward_ops(y7t_det* det, int B, int first, int last, y7t_stream stream);

/* Fused Detect: describe the Detect levels once (strides / anchors in pixels, models/yolo.py:23-62), then y7t_det_forward_fused runs
 * ops [first, last) like y7t_det_forward_ops, except that the Detect 1x1 convs do not write their (B, ny, nx, na*no) fp32 tensors:
 * their epilogue applies sigmoid + decode (yolo.py:49-56) and the candidate filter of non_max_suppression (utils/general.py:629-662:
 * obj > conf_thres, best class, conf > conf_thres) and appends the survivors to the candidate arrays at the start of `workspace`
 * (layout of y7t_det_postprocess; `cap`, `max_nms` as there).  The candidate counters are zeroed when the range contains op 0.
 * Afterwards call y7t_det_postprocess with head == NULL on the same workspace: it skips its own decode pass.  The (B, 102000, 5+nc)
 * tensor of the reference and the four raw head tensors are never written. */
int y7t_det_set_detect(y7t_det* det, int nl, int na, int no, const float* strides_host, const float* anchors_host);
int y7t_det_forward_fused(y7t_det* det, int B, int first, int last, float conf_thres, int cap, int max_nms, void* workspace,
                          size_t workspace_bytes, y7t_stream stream);

/* TrackerLoader.__getitem__ tail (tracker/tracker_dataloader.py:83-88) + ReOrg (models/common.py:48-53):
 * img: (B,3,H,W) float32 RGB in [0,1] (is_u8 = 0) or (B,H,W,3) uint8 BGR (is_u8 = 1: BGR->RGB and /255 fused);
 * out: NHWC fp16 with ldout channels (16 with reorg: 12 + 4 zero; 8 without: 3 + 5 zero). */
int y7t_input_layout(const void* img, int is_u8, int B, int H, int W, int reorg, void* out_f16, int ldout, y7t_stream stream);

/* The whole front of the forward as one kernel, for plans that start with ReOrg + Conv 3x3 -> 64 (YOLOv7-w6, cfg/deploy/yolov7-w6.yaml:16-17;
 * y7t_det_stem_fusable): raw (B, H0, W0, 3) uint8 BGR frames -> TrackerLoader._letterbox (resize INTER_LINEAR to new_w x new_h at (top, left)
 * of the H x W network input, pad 114; new == source: no resampling) -> BGR->RGB, /255 -> ReOrg -> stem Conv + bias + activation, i.e. what
 * y7t_letterbox_layout_u8 / y7t_input_layout followed by op 0 compute, without the fp16 layout tensor in between.  Continue with
 * y7t_det_forward_ops / y7t_det_forward_fused from op 1. */
int y7t_det_stem_fusable(const y7t_det* det);
int y7t_det_forward_stem_u8(y7t_det* det, const void* frames_u8, int B, int H0, int W0, int new_h, int new_w, int top, int left, y7t_stream stream);

/* TrackerLoader._letterbox (tracker/tracker_dataloader.py:100-130) fused with the layout above, for raw (B,H0,W0,3) uint8 BGR
 * frames: bilinear resize (cv2.INTER_LINEAR geometry: half-pixel centres; float arithmetic, rounded to uint8) to new_w x new_h,
 * placed at (top, left) of the H x W letterboxed image, pad colour 114.  The host computes new_h/new_w/top/left exactly like
 * the reference (auto=True: pad only to the stride multiple). */
int y7t_letterbox_layout_u8(const void* img, int B, int H0, int W0, int H, int W, int new_h, int new_w, int top, int left, int reorg,
                            void* out_f16, int ldout, y7t_stream stream);

/* Detect.forward decode (models/yolo.py:39-57) + non_max_suppression (utils/general.py:607-695, multi_label=False,
 * agnostic=False) + scale_coords/clip/round (general.py:319-340, tracker/track.py:234-244), all on the device.
 * head[l]: NHWC fp32 output of the l-th Detect 1x1 conv, [B][ny][nx][na*no].  anchors: [nl][na][2] pixels (host).
 * letterbox: DEVICE [B][5] = gain, pad_w, pad_h, H0, W0.  dets: [B][max_det][6]; ndets: [B];
 * cand_count (may be NULL): [B] candidates above conf_thres (> cap == overflow, caller must check).
 * cap: candidate capacity per image (the number of anchors can never overflow it); the NMS itself runs on the top
 * min(cap, max_nms) candidates like the reference.  workspace: y7t_det_postprocess_workspace_bytes(B, cap, max_nms).
 * head == NULL: the candidates are already in the workspace (y7t_det_forward_fused); ny/nx/strides/anchors are then unused. */
size_t y7t_det_postprocess_workspace_bytes(int B, int cap, int max_nms);
int y7t_det_postprocess(const float* const* head_host_array_of_dev_ptrs, const int* ny, const int* nx, const float* strides,
                        const float* anchors, int nl, int na, int no, int B, float conf_thres, float iou_thres, int max_det,
                        int max_nms, int cap, const float* letterbox, float* dets, int* ndets, int* keep_idx, int* cand_count,
                        void* workspace, size_t workspace_bytes, y7t_stream stream);

/* ---------------------------------------------------------------- ReID embedding (DeepSORT's appearance branch) ---- */
/* DeepSORT.get_feature -> Extractor (tracker/deepsort.py:19-41, tracker/reid_models/deepsort_reid.py:112-153) with OSNet
 * (tracker/reid_models/OSNet.py:282-438) as the network: crops of the frame -> /255, bilinear resize to in_h x in_w (cv2.INTER_LINEAR
 * geometry on the float image), Normalize(mean, std) in the frame's channel order -> the network in eval mode -> (N, feat_dim) float32.
 * Like the detector, the network is a host-lowered op list (BatchNorm folded, fp32 weights in one blob) over a caller-owned arena. */
enum { Y7T_REID_CONV = 0, Y7T_REID_DWCONV3 = 1, Y7T_REID_MAXPOOL3S2 = 2, Y7T_REID_AVGPOOL2 = 3, Y7T_REID_GATE_ACC = 4, Y7T_REID_ADD_RELU = 5,
       Y7T_REID_GAP = 6, Y7T_REID_FC = 7, Y7T_REID_L2NORM = 8 /* x / |x| per crop: Net.forward with reid=True, reid_models/deepsort_reid.py:104 */,
       /* fp16 NHWC buffers (sized in floats like the others: halves / 2), convolutions on the detector's MFMA kernels -- the op list of the
        * reference's DeepSORT embedding network (reid_models/deepsort_reid.py:14-110), tracker/reid.py::lower_deepsort_net_f16:
        * H_PACK fp32 (H, W, 3) -> fp16 (H, W, 16); H_CONV k in {1, 3}, C % 8 == 0, Co % 64 == 0, weights fp16 [Co][round_up(k*k*C, 64)] with
        * k index (kh * k + kw) * C + ci stored at float offset w_off of the blob, fp32 bias [Co] at b_off, no activation; H_MAXPOOL_RELU
        * relu then 3x3 / 2 / pad 1; H_RELU; H_ADD_RELU relu(in + aux); H_GAP_L2NORM mean over the map then x / |x| -> fp32 (C) */
       Y7T_REID_H_PACK = 9, Y7T_REID_H_CONV = 10, Y7T_REID_H_MAXPOOL_RELU = 11, Y7T_REID_H_RELU = 12, Y7T_REID_H_ADD_RELU = 13, Y7T_REID_H_GAP_L2NORM = 14 };
typedef struct y7t_reid_op {
    int32_t type;
    int32_t in_buf, out_buf, aux_buf;   /* arena buffers; aux: second addend (ADD_RELU), pooled + gate scratch (GATE_ACC), -1 otherwise */
    int32_t H, W, C;                    /* input map and channels */
    int32_t Ho, Wo, Co;                 /* output map and channels (CONV, pools, FC) */
    int32_t k, s, p;                    /* CONV window */
    int32_t relu;                       /* CONV / DWCONV3 / FC: ReLU after the bias; GATE_ACC: 1 = first branch (overwrite the accumulator) */
    int32_t R, w_kmajor;                /* GATE_ACC: hidden width of the gate MLP (C / 16); CONV: 1 = weights stored (kh, kw, ci, co) instead of (co, kh, kw, ci) */
    int64_t w_off, b_off;               /* float offsets into the weight blob (b_off < 0: no bias); GATE_ACC: fc1 weight / bias */
    int64_t w2_off, b2_off;             /* GATE_ACC: fc2 weight / bias */
} y7t_reid_op;                          /* sizeof == 96 */
typedef struct y7t_reid y7t_reid;
int y7t_reid_create(const y7t_reid_op* ops_host, int n_ops, const int64_t* buf_offsets_host /* in floats */, int n_bufs, void* arena, size_t arena_bytes,
                    const void* weights_f32, int max_crops, int in_h, int in_w, int feat_dim, y7t_reid** out);
int y7t_reid_destroy(y7t_reid* reid);
/* frame_u8: (H, W, 3) uint8 frame in DEVICE memory, boxes: N x 4 float32 tlbr (device) -- or crops_f32 != NULL: N x in_h x in_w x 3 float32
 * crops that are already resized and normalised (device; frame_u8 / boxes ignored).  feats: N x feat_dim float32 (device). */
int y7t_reid_forward(y7t_reid* reid, const void* frame_u8, int H, int W, const float* boxes, int N, const float* crops_f32, float* feats,
                     y7t_stream stream);
/* Crops taken from several frames in one pass (a batch of frames and all their detections -- the throughput mode of BASELINE config 4):
 * frames_u8 = n_frames x (H, W, 3) uint8 contiguous (device), frame_idx[i] = frame of box i (device int32; clamped to [0, n_frames)). */
int y7t_reid_forward_batch(y7t_reid* reid, const void* frames_u8, int n_frames, int H, int W, const float* boxes, const int* frame_idx, int N,
                           float* feats, y7t_stream stream);
/* The MFMA path of config 4 ("ReID conv as MFMA kernel"): OSNet x0_25 on 128 x 64 crops as ONE kernel, one workgroup per crop, every
 * intermediate in LDS, fp16 storage / fp32 accumulate (tracker/reid_models/OSNet.py:223-279,282-438,567-579 with BatchNorm folded).  `blob`
 * (device memory, y7t_reid_fused_blob_size() bytes) holds the parameters in the kernel's consumption order with the 1x1 weights already in
 * MFMA fragment order (host: tracker/reid.py::pack_fused).  Once set, y7t_reid_forward / _batch on frame crops run the fused kernel; the
 * op list above stays the fp32 path (crops_f32 input, other widths / crop sizes).  blob == NULL switches back. */
size_t y7t_reid_fused_blob_size(void);
int y7t_reid_set_fused(y7t_reid* reid, const void* blob, size_t blob_bytes);

/* single fused Conv+bias+act launch (layer-level parity tests, rocprof attribution); same fields as y7t_op but
 * with raw device pointers.  Small launches split K and sum fp32 slabs from ONE process-wide workspace: call it from one stream at
 * a time (a y7t_det owns its own workspace and has no such limit). */
int y7t_conv2d_nhwc_f16(const void* in, int in_ld, int in_coff, int B, int H, int W, int Cin, const void* w_packed, const float* bias,
                        void* out, int out_ld, int out_coff, int out_f32, int Cout, int Cout_pad, int KH, int KW, int stride, int pad,
                        int act, const void* zeros16, y7t_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* Y7T_H */
