"""ctypes binding of liby7t.so (the C ABI declared in include/y7t.h).

The library is the product: if it is missing or cannot be loaded this module raises -- there is
no CPU fallback anywhere in the package (the CPU oracle lives in /oracle and is test-only).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("Y7T_LIB") or os.path.join(_HERE, "lib", "liby7t.so")   # Y7T_LIB: A/B runs against another build of the library (scripts/gpu_round.sh exp_noslp)

# Run-time switches (README "switches").  The product keeps ten, read wherever they apply; every other Y7T_* variable is an EXPERIMENT switch -- lowering thresholds,
# tile variants, timing ablations -- and is honoured only when the measuring build is loaded (Y7T_LIB=.../liby7t_ablate.so, the same sources with -DY7T_ABLATE_BUILD:
# csrc/y7t_common.h).  With the product library they answer their measured default whatever the environment says.
PRODUCT_SWITCHES = ("Y7T_LIB", "Y7T_CONV_WS", "Y7T_CONV_WS_S2", "Y7T_CONV_WS128", "Y7T_CONV_WS_DYN", "Y7T_CONV_P8", "Y7T_CONV_PATCH_S2", "Y7T_UPSAMPLE_ON_READ",
                    "Y7T_STEM_FUSED", "Y7T_TRACKER_ARENA")


def ablate_build():
    """True when the loaded library is the measuring build (liby7t_ablate.so)"""
    return "ablate" in os.path.basename(os.environ.get("Y7T_LIB") or LIB_PATH)


def switch(name, default):
    """value of a Y7T_* switch as the lowering should see it (a string, like os.environ.get)"""
    if name in PRODUCT_SWITCHES or ablate_build():
        return os.environ.get(name, default)
    return default


c_void_p, c_int, c_double, c_size_t, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_double, ctypes.c_size_t, ctypes.c_float

# name -> (restype, argtypes); kept in sync with include/y7t.h (tests/test_abi.py checks the header)
SIGNATURES = {
    "y7t_last_error": (ctypes.c_char_p, []),
    "y7t_version": (c_int, []),
    "y7t_last_kernel": (ctypes.c_char_p, []),
    "y7t_device_count": (c_int, []),
    "y7t_iou_cost_f64": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "y7t_kf_initiate_f64": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "y7t_kf_multi_predict_f64": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "y7t_kf_project_f64": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "y7t_kf_update_batch_f64": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "y7t_kf_gating_f64": (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "y7t_lap_literal_calls": (c_int, []),
    "y7t_lapjv_workspace_bytes": (c_size_t, [c_int, c_int]),
    "y7t_lapjv_f64": (c_int, [c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "y7t_lapjv_f64_host": (c_int, [c_void_p, c_int, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p]),
    "y7t_tracker_state_bytes": (c_size_t, [c_int, c_int]),
    "y7t_tracker_release": (c_int, [c_void_p]),
    "y7t_tracker_init": (c_int, [c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_double, c_double, c_int, c_int, c_void_p,
                                 c_void_p]),
    "y7t_tracker_step_frames": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "y7t_tracker_step_batch": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "y7t_tracker_step": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    "y7t_deepsort_feature_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "y7t_deepsort_init": (c_int, [c_void_p, c_size_t, c_int, c_int, c_int, c_int, c_void_p]),
    "y7t_tracker_step_deepsort": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p]),
    "y7t_kf_multi_gmc_f64": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "y7t_tracker_layout": (c_int, [c_int, c_int, c_void_p, c_int]),
    "y7t_tracker_field_name": (ctypes.c_char_p, [c_int]),
    "y7t_det_create": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_int, c_void_p]),
    "y7t_det_destroy": (c_int, [c_void_p]),
    "y7t_det_forward": (c_int, [c_void_p, c_int, c_void_p]),
    "y7t_det_num_ops": (c_int, [c_void_p]),
    "y7t_det_forward_ops": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p]),
    "y7t_det_set_detect": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "y7t_det_forward_fused": (c_int, [c_void_p, c_int, c_int, c_int, c_float, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "y7t_det_stem_fusable": (c_int, [c_void_p]),
    "y7t_det_forward_stem_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "y7t_input_layout": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "y7t_letterbox_layout_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_int, c_void_p]),
    "y7t_det_postprocess_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "y7t_det_postprocess": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_float, c_int,
                                    c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "y7t_reid_create": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_size_t, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "y7t_reid_destroy": (c_int, [c_void_p]),
    "y7t_reid_forward": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "y7t_reid_forward_batch": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "y7t_reid_fused_blob_size": (ctypes.c_size_t, []),
    "y7t_reid_set_fused": (c_int, [c_void_p, c_void_p, ctypes.c_size_t]),
    "y7t_stream_create_cu_mask": (c_int, [c_void_p, c_int, c_void_p]),
    "y7t_stream_destroy": (c_int, [c_void_p]),
    "y7t_conv2d_nhwc_f16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                    c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
}

_lib = None


class Y7TError(RuntimeError):
    pass


def load():
    """Load liby7t.so; raises Y7TError when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise Y7TError("liby7t.so is missing (%s): build it with `python -m yolov7_tracker_amd.build` -- "
                       "this package has no CPU fallback" % LIB_PATH)
    # torch's bundled HIP runtime must be the one in the process (the library works on torch-owned device memory and
    # streams): loading liby7t.so first would bind /opt/rocm's libamdhip64 and leave torch without a visible device
    import torch  # noqa: F401
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise Y7TError("cannot load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(L, name)
        except AttributeError:
            raise Y7TError("liby7t.so does not export %s (stale build?)" % name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise Y7TError("liby7t call failed (%d): %s" % (rc, load().y7t_last_error().decode()))


def require_gpu():
    """The hot path runs on the device only."""
    import torch
    if not torch.cuda.is_available() or load().y7t_device_count() < 1:
        raise Y7TError("no MI355X / HIP device visible: the yolov7_tracker_amd hot path has no CPU fallback")


def stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """device/host pointer of a torch tensor (or None)."""
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def cu_masked_streams(reserved, total=None):
    """-> (stream_rest, stream_reserved): two torch streams (torch.cuda.ExternalStream over hipExtStreamCreateWithCUMask) whose kernels run on
    disjoint sets of compute units: `reserved` CUs (mask bits 0 .. reserved-1; consecutive bits land on different XCDs) for the second, all others
    for the first.  For pipelines that keep single-workgroup latency-bound kernels (the tracker frame steps) off the CUs the convolutions fill."""
    import torch
    require_gpu()
    L = load()
    total = int(total or torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count)
    reserved = int(reserved)
    if not 0 < reserved < total:
        raise Y7TError("cu_masked_streams: reserved=%d of %d compute units" % (reserved, total))
    words = (total + 31) // 32
    out = []
    for bits in (range(reserved, total), range(0, reserved)):
        m = (ctypes.c_uint32 * words)()
        for b in bits:
            m[b // 32] |= 1 << (b % 32)
        h = ctypes.c_void_p()
        check(L.y7t_stream_create_cu_mask(m, words, ctypes.byref(h)))
        out.append(torch.cuda.ExternalStream(h.value))
    return out[0], out[1]
