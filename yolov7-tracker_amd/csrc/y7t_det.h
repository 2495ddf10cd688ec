// y7t_det.h -- internal types shared by the detector translation units (conv, pooling, post-process, plan executor)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

enum { Y7T_ACT_NONE = 0, Y7T_ACT_SILU = 1, Y7T_ACT_LEAKY = 2 };

// Detect decode fused into the 1x1 conv's epilogue (y7t_det_forward_fused): level parameters + where the candidates go
struct Y7TDecode {
    float stride, aw[3], ah[3];   // models/yolo.py:52-53: xy = (s*2 - 0.5 + grid) * stride; wh = (s*2)^2 * anchor_grid
    int ny, nx, row0, na, no;     // row0: first row of this level in the reference's (B, A, no) ordering
    float conf_thres;
    int cap;                      // candidate capacity per image
    float *cbox, *cscore, *ccls;  // [B][cap][4] xyxy, [B][cap], [B][cap]
    int *cidx, *count;            // [B][cap] row index, [B] candidates found
};

struct Y7TConvArgs {
    const _Float16* in;  // NHWC fp16 buffer holding the input slice
    int ldin, cin_off;   // channels of that buffer, first channel of the slice
    int B, H, W, Cin;    // input batch / spatial size / channels of the slice (multiple of 8)
    const _Float16* w;   // packed weights [Cout_pad][K_pad], k = (kh*KW + kw)*Cin + ci
    const float* bias;   // [Cout_pad]
    void* out;           // NHWC buffer (fp16, or fp32 when out_f32)
    int ldout, cout_off, out_f32;
    int Ho, Wo, Cout, Cout_pad;
    int KH, KW, stride, pad;
    int K, K_pad;        // K = KH*KW*Cin, K_pad = round_up(K, 64)
    int M;               // B*Ho*Wo
    int act;
    const _Float16* zeros;  // unused (kept for ABI stability of y7t_conv2d_nhwc_f16)
    unsigned in_bytes, w_bytes;
    int xcd_swizzle, tile_order;
    int splitk, ksteps, allow_splitk;   // split-K: workgroups per output tile, K-steps per split
    float* partial;                     // fp32 slabs [splitk][M][Cout_pad]
    float* splitk_ws;                   // caller's split-K workspace of Y7T_SPLITK_WS_BYTES (a detector owns one, so detectors on different streams
                                        // never share slabs), or null: one process-wide workspace (the single-layer entry point; one stream at a time)
    int korder;   // 1: weights packed in (kh, 64-channel chunk, kw) K order (3x3, Cin % 64 == 0); 2: LDS-patch panels (9: the same with 64-row panels); 3: 1x1 panels; 4: stride-2 LDS-patch panels; 5 / 8: weights-stationary fragments; 7: p8 panels
    int panel64;       // 64-row weight panels although Cout_pad % 128 == 0 (korder 9: patch kernel, korder 10: 1x1 panels; small maps, where 128-row panels would leave most CUs without a workgroup)
    int force_patch;   // tests: run an eligible 3x3/s1 layer on k_conv3x3_patch whatever its tile efficiency
    int no_patch;      // host-side: keep this launch on the generic implicit-GEMM kernel
    int ablate;   // debug: bit0 skip DMA loads, bit1 skip MFMAs, bit2 skip the whole compute phase  // extents for the buffer descriptors (filled by y7t_conv_launch)
    int epi;      // 1: Detect decode + candidate filter instead of the output store (1x1 convs, dec below)
    Y7TDecode dec;
    // upsample-on-read (1x1 convs): channels [up_c0, up_c0 + up_C) of the input come from `in2` (H/2 x W/2, ldin2 channels, slice at cin2_off)
    const _Float16* in2;
    int ldin2, cin2_off, up_c0, up_C;
    unsigned in2_bytes;
    // dynamic tile scheduling of the persistent kernels (y7t_conv_ws.hip, y7t_conv_ws128.hip): Y7T_TILE_CTR_INTS ints, zero between launches -- [0 .. 3] chunks handed
    // out (one counter per output-channel tile of the launch), [Y7T_TILE_CTR_DONE] workgroups that have left (the last one resets them all).  One set per op of a
    // detector's plan (y7t_detector.hip); null = static partition (the single-layer entry point).
    int* tile_ctr;
};
#define Y7T_TILE_CTR_DONE 4
#define Y7T_TILE_CTR_INTS 8

#define Y7T_SPLITK_WS_BYTES (128ull << 20)
int y7t_conv_launch(const Y7TConvArgs& a, hipStream_t s);

// fused uint8 frame -> (letterbox) -> layout -> stem conv (y7t_stem.hip)
int y7t_stem_u8_launch(const void* frames_u8, int B, int H0, int W0, int H, int W, int new_h, int new_w, int top, int left, const _Float16* w, int K_pad,
                       const float* bias, _Float16* out, int ldout, int cout_off, int act, hipStream_t s);

int y7t_upsample_launch(const _Float16* in, int ldin, int cin_off, int B, int H, int W, int C, _Float16* out, int ldout, int cout_off, hipStream_t s);
int y7t_spp3_try(const _Float16* in, int ldin, int cin_off, int B, int H, int W, int C, _Float16* out, int ldout, int cout_off, hipStream_t s);   // 0 done, 1 not applicable
int y7t_maxpool_launch(const _Float16* in, int ldin, int cin_off, int B, int H, int W, int C, int k, int st, int pd, _Float16* out, int ldout,
                       int cout_off, hipStream_t s);

// decode + NMS chain (y7t_post.hip)
struct Y7TPostArgs {
    const float* head[4];   // per level NHWC fp32 [B][ny][nx][na*no]
    int predecoded;         // 1: the candidate arrays of `ws` are already filled (fused Detect epilogue): skip the decode pass
    int ny[4], nx[4];
    float stride[4];
    float anchors[24];      // [nl][na][2] in pixels
    int nl, na, no, B;
    float conf_thres, iou_thres;
    int max_det, max_nms, cap;
    const float* letterbox_dev;  // [B][5] gain, padw, padh, H0, W0 (device)
    float* dets;            // [B][max_det][6]
    int* ndets;             // [B]
    int* keep_idx;          // [B][max_det] candidate slot of each kept detection
    int* count_out;         // [B] number of candidates found (> cap means overflow) or null
    void* ws; size_t ws_bytes;
};
size_t y7t_post_ws_bytes(int B, int cap, int max_nms);
// the candidate arrays at the start of a postprocess workspace (shared with the fused Detect epilogue)
struct Y7TCandWs { float *cbox, *cscore, *ccls; int *cidx, *count; };
Y7TCandWs y7t_post_cand_ws(void* ws, int B, int cap);
int y7t_post_run(const Y7TPostArgs& a, hipStream_t s);
