// y7t_reid_fused.h -- launch interface of the one-workgroup-per-crop OSNet x0_25 kernel (y7t_reid_fused.hip)
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

struct Y7TReidFusedArgs {
    const uint8_t* frames;      // uint8 (H, W, 3) frames, `frame_stride` bytes apart
    long long frame_stride;
    int H, W;
    const float* boxes;         // (N, 4) tlbr in frame pixels
    const int* frame_idx;       // (N) frame of every box, or null: all boxes on frame 0
    int n_frames;               // frame indices are clamped to [0, n_frames)
    int N;
    const char* blob;           // parameters in the kernel's consumption order (tracker/reid.py::pack_fused)
    float* feats;               // (N, 512)
    long long* prof;            // diagnostics (Y7T_REID_PROF=1): shader-clock stamps of workgroup 0's phases, or null
};

size_t y7t_reid_fused_blob_bytes();
int y7t_reid_fused_launch(const Y7TReidFusedArgs& a, hipStream_t s);
