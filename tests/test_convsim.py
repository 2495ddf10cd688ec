"""CPU: the implicit-GEMM convolution kernel of csrc/y7t_conv.hip, compiled for the host FROM ITS REAL SOURCE and run work-item by work-item
(tests/_convsim: OS threads per work-item, pthread barriers, models of the gfx950 builtins -- buffer->LDS DMA with the hardware range check,
v_mfma_f32_32x32x16_f16, v_permlane32_swap), against a plain convolution.  What this pins without a GPU: load geometry, LDS swizzle, MFMA fragment
mapping, K orders, ragged K, slices, split-K, the epilogue's transposition, the tile order -- and it is where a new kernel (the stride-2 patch kernel, the
weights-stationary 64 -> 64 kernel) is developed before GPU minutes are spent on it.  The `-m gpu` layer tests remain the check of the real thing."""
import numpy as np
import pytest
import torch

from tests import _convsim as cs

pytestmark = pytest.mark.skipif(not __import__("os").path.exists(cs._CLANG), reason="needs the ROCm clang++ (host compile of the kernel source)")


def pack_w(W, cin_pad, cout_pad, korder=0):
    cout, cin, k, _ = W.shape
    K = k * k * cin_pad
    Kp = (K + 63) // 64 * 64
    Wt = np.zeros((cout, k, k, cin_pad), np.float32)
    Wt[..., :cin] = W.transpose(0, 2, 3, 1)
    if korder == 1:      # (kh, 64-channel chunk, kw) K order
        Wt = Wt.reshape(cout, k, k, cin_pad // 64, 64).transpose(0, 1, 3, 2, 4)
    blk = np.zeros((cout_pad, Kp), np.float16)
    blk[:cout, :K] = Wt.reshape(cout, -1).astype(np.float16)
    if korder in (2, 9):      # the patch kernel's panel order (9: 64-row panels although Cout_pad % 128 == 0)
        from yolov7_tracker_amd.detector import weights
        blk = weights.panel_pack(blk, cin_pad, narrow=korder == 9)
    if korder in (3, 10):      # the 1x1 panel order (10: 64-row panels although Cout_pad % 128 == 0)
        from yolov7_tracker_amd.detector import weights
        blk = weights.panel_pack_linear(blk, narrow=korder == 10)
    if korder == 5:      # the weights-stationary kernel's register-fragment order
        from yolov7_tracker_amd.detector import weights
        blk = weights.pack_ws(blk)
    if korder == 4:      # the stride-2 patch kernel's panel order
        from yolov7_tracker_amd.detector import weights
        blk = weights.panel_pack_s2(blk, cin_pad)
    if korder == 6:      # the 128-channel weights-stationary kernel's register-fragment order
        from yolov7_tracker_amd.detector import weights
        blk = weights.pack_ws128(blk)
    if korder == 7:      # the 256 x 64 panels of the ping-pong 1x1 kernel
        from yolov7_tracker_amd.detector import weights
        blk = weights.panel_pack_p8(blk)
    if korder == 8:      # register-fragment order of the stride-2 weights-stationary kernel
        from yolov7_tracker_amd.detector import weights
        blk = weights.pack_ws_s2(blk)
    return blk


def run_case(L, B, H, W, Cin, Cout, k, s, act, tile, in_ld=None, in_coff=0, out_ld=None, out_coff=0, out_f32=0, korder=0, splitk=0, seed=0, force_patch=0):
    in_ld, out_ld = in_ld or Cin, out_ld or Cout
    rng = np.random.default_rng(seed)
    x = rng.normal(0, 1, (B, H, W, in_ld)).astype(np.float16)
    Wt = (rng.normal(0, 1, (Cout, Cin, k, k)) / np.sqrt(Cin * k * k)).astype(np.float32)
    bias = rng.normal(0, 0.5, Cout).astype(np.float32)
    cp = (Cout + 255) // 256 * 256 if korder == 7 else (Cout + 63) // 64 * 64 if korder not in (4, 6, 8) else (Cout + 127) // 128 * 128
    wp = pack_w(Wt, Cin, cp, korder)
    bp = np.zeros(cp, np.float32)
    bp[:Cout] = bias
    pad = k // 2
    Ho, Wo = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    out = np.full((B, Ho, Wo, out_ld), 7.0, np.float32 if out_f32 else np.float16)
    rc = L.cs_conv(x.ctypes.data, in_ld, in_coff, B, H, W, Cin, wp.ctypes.data, bp.ctypes.data, out.ctypes.data, out_ld, out_coff, out_f32, Cout, cp, k, k, s, pad,
                   act, korder, tile, splitk, force_patch)
    assert rc == 0, L.cs_last_error().decode()
    xt = torch.from_numpy(x[..., in_coff:in_coff + Cin].astype(np.float32)).permute(0, 3, 1, 2)
    ref = torch.nn.functional.conv2d(xt, torch.from_numpy(Wt.astype(np.float16).astype(np.float32)), torch.from_numpy(bias), s, pad)
    ref = ref * torch.sigmoid(ref) if act == 1 else torch.where(ref > 0, ref, 0.1 * ref) if act == 2 else ref
    ref = ref.permute(0, 2, 3, 1).numpy()
    got = out[..., out_coff:out_coff + Cout].astype(np.float32)
    np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3)       # fp16 output rounding (2^-11) + the summation order
    other = np.ones(out_ld, bool)
    other[out_coff:out_coff + Cout] = False
    assert np.all(out[..., other] == 7.0)                            # nothing outside the slice is written
    return L.cs_last_kernel().decode()


# B, H, W, Cin, Cout, k, s, act, tile, extras -- shapes that exercise every branch of the generic kernel on a few workgroups
CASES = [
    (1, 12, 12, 64, 128, 3, 1, 1, 128128064, {}),
    (1, 10, 14, 64, 64, 3, 2, 1, 128064064, {}),                                       # stride 2, 64-channel tile, ragged image edge
    (2, 9, 9, 128, 128, 1, 1, 1, 128128032, {}),                                       # 1x1 fast path, 32-deep stages
    (1, 16, 16, 16, 64, 3, 1, 1, 256064032, {}),                                       # stem-like: Cin = 16, K = 144 (ragged K)
    (1, 8, 8, 96, 192, 1, 1, 2, 0, {"in_ld": 256, "in_coff": 64, "out_ld": 384, "out_coff": 192}),      # slices of wider buffers, LeakyReLU, dispatch rules
    (1, 8, 8, 256, 45, 1, 1, 0, 0, {"out_ld": 45, "out_f32": 1}),                      # Detect head (plain): fp32 out, Cout not a multiple of 4
    (1, 11, 9, 128, 128, 3, 1, 1, 128128064, {"korder": 1}),                          # (kh, chunk, kw) K order
    (1, 13, 11, 192, 64, 3, 2, 1, 128064064, {"korder": 1, "in_ld": 256, "in_coff": 64}),
    (1, 17, 17, 64, 256, 3, 1, 1, 256256064, {}),                                      # the 256 x 256 tile
    (1, 10, 10, 128, 128, 3, 1, 1, 128128064, {"splitk": 1}),                          # split-K + k_splitk_reduce
    (1, 12, 12, 64, 128, 3, 1, 1, 128128364, {}),                                      # three-stage ring
    (2, 9, 9, 128, 128, 1, 1, 1, 0, {"korder": 3}),                                   # 1x1 panels through the dispatch rules: 128-row panels
    (2, 9, 9, 128, 256, 1, 1, 2, 0, {"korder": 10, "out_ld": 320, "out_coff": 64}),      # ... 64-row panels on a 256-channel layer (korder 10)
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%dx%dx%d_%d-%d_k%ds%d_t%d" % (c[0], c[1], c[2], c[3], c[4], c[5], c[6], c[8]))
def test_shipped_kernel_source_on_the_host(case):
    B, H, W, Cin, Cout, k, s, act, tile, kw = case
    name = run_case(cs.lib(), B, H, W, Cin, Cout, k, s, act, tile, **kw)
    assert name.startswith("igemm<") and (tile == 0 or ("splitK" in name) == bool(kw.get("splitk")))      # (the dispatch rules split K of small problems themselves)


@pytest.mark.parametrize("korder,Cout,lat_first,force", [(0, 128, True, 0), (3, 192, False, 0), (3, 256, True, 0)], ids=["rows-128", "panels-64", "panels-128"])
def test_upsample_on_read_loader_on_the_host(korder, Cout, lat_first, force):
    """the DUAL instance of the 1x1 fast path (default launch list: the three convs behind Concat[lateral, Upsample(x)], /root/reference/cfg/deploy/yolov7-w6.yaml:75,89,103):
    K-steps whose channels lie in the upsampled range DMA pixel (y >> 1, x >> 1) of the half-resolution tensor -- against a conv over the materialised concat"""
    L = cs.lib()
    rng = np.random.default_rng(5)
    B, H, W, C_lat, C_up = 2, 10, 14, 64, 128
    Cin = C_lat + C_up
    up_c0 = C_lat if lat_first else 0                                    # nn.Upsample output after or before the lateral tensor in the concat
    lat = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float16)            # the concat buffer: the upsampled channel range is never written (garbage) ...
    lat[..., up_c0:up_c0 + C_up] = 77.0                                  # ... and must never be read
    half = rng.normal(0, 1, (B, H // 2, W // 2, 192)).astype(np.float16)  # a wider half-resolution buffer, slice at channel 32
    Wt = (rng.normal(0, 1, (Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    bias = rng.normal(0, 0.5, Cout).astype(np.float32)
    cp = (Cout + 63) // 64 * 64
    wp = pack_w(Wt, Cin, cp, 0)
    if korder == 3:
        from yolov7_tracker_amd.detector import weights
        wp = weights.panel_pack_linear(wp)
    bp = np.zeros(cp, np.float32)
    bp[:Cout] = bias
    out = np.full((B, H, W, Cout), 7.0, np.float16)
    rc = L.cs_conv_dual(lat.ctypes.data, Cin, 0, half.ctypes.data, 192, 32, up_c0, C_up, B, H, W, Cin, wp.ctypes.data, bp.ctypes.data, out.ctypes.data, Cout, 0, Cout, cp, 1,
                        korder, force)
    assert rc == 0, L.cs_last_error().decode()
    name = L.cs_last_kernel().decode()
    assert "upsample-on-read" in name, name
    x = lat.astype(np.float32)
    x[..., up_c0:up_c0 + C_up] = np.repeat(np.repeat(half[..., 32:32 + C_up].astype(np.float32), 2, axis=1), 2, axis=2)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(Wt.astype(np.float16).astype(np.float32)), torch.from_numpy(bias))
    ref = (ref * torch.sigmoid(ref)).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(out.astype(np.float32), ref, rtol=2e-3, atol=2e-3)


# the 256 x 256 x 64 ping-pong 1x1 kernel (csrc/y7t_conv_p8.hip, korder 7): B, H, W, Cin, Cout, act, extras.  Every case runs twice: with the buffer->LDS DMAs landing at
# issue (the earliest they can: a half-tile re-staged while somebody still reads it gives wrong results) and landing only when the issuing lane's own `vmcnt` wait forces
# them (the latest: a fragment read that no wait + barrier covers sees the 0xAB fill) -- the two ends of what the hardware can do with the kernel's schedule.
P8_CASES = [
    (1, 16, 20, 128, 256, 1, {}),                                                       # two K-tiles; 320 pixels = one full and one ragged pixel tile
    (1, 16, 16, 64, 512, 2, {"in_ld": 128, "in_coff": 64, "out_ld": 768, "out_coff": 256}),   # ONE K-tile (prologue + zero-fill tail only), two channel tiles, slices, LeakyReLU
    (1, 15, 20, 192, 256, 0, {}),                                                       # three K-tiles (odd: the loop ends on buffer 0), 300 pixels, no activation
    (2, 12, 16, 320, 256, 1, {"out_ld": 512, "out_coff": 0}),                           # five K-tiles, 384 pixels over two images
    # round 6, the persistent form (k_conv1x1_p8p: the host model has 3 compute units, so >= 6 tiles with one channel tile / >= 4 with two and an even number of K-tiles take it):
    # three workgroups x two tiles, two K-tiles per tile (the shortest loop the cross-tile prefetch allows); two channel tiles, three tiles per workgroup, a ragged last tile, four K-tiles
    (2, 24, 32, 128, 256, 1, {}),
    (1, 35, 40, 256, 512, 2, {"out_ld": 640, "out_coff": 128}),
]


@pytest.mark.parametrize("deferred", [0, 1], ids=["dma-at-issue", "dma-at-wait"])
@pytest.mark.parametrize("case", P8_CASES, ids=lambda c: "%dx%dx%d_%d-%d" % c[:5])
def test_pingpong_1x1_kernel_source_on_the_host(case, deferred):
    B, H, W, Cin, Cout, act, kw = case
    L = cs.lib()
    L.cs_set_dma_deferred(deferred)
    try:
        name = run_case(L, B, H, W, Cin, Cout, 1, 1, act, 0, korder=7, seed=B * 1000 + H + W + Cin, **kw)
    finally:
        L.cs_set_dma_deferred(0)
    assert name == "p8<256,256,64> 1x1", name


@pytest.mark.parametrize("deferred", [0, 1], ids=["dma-at-issue", "dma-at-wait"])
@pytest.mark.parametrize("lat_first", [True, False])
def test_pingpong_1x1_upsample_on_read_on_the_host(lat_first, deferred):
    """the DUAL instance: K-tiles of the upsampled channel range DMA pixel (y >> 1, x >> 1) of the half-resolution tensor (cfg/deploy/yolov7-w6.yaml:75,89,103)"""
    from yolov7_tracker_amd.detector import weights
    L = cs.lib()
    rng = np.random.default_rng(11)
    B, H, W, C_lat, C_up, Cout = 2, 10, 14, 64, 128, 256
    Cin = C_lat + C_up
    up_c0 = C_lat if lat_first else 0
    lat = rng.normal(0, 1, (B, H, W, Cin)).astype(np.float16)            # the concat buffer: the upsampled channel range is never written (garbage) ...
    half = rng.normal(0, 1, (B, H // 2, W // 2, 192)).astype(np.float16)  # ... it lives at half resolution, as channels [32, 32 + C_up) of a wider buffer
    Wt = (rng.normal(0, 1, (Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    bias = rng.normal(0, 0.5, Cout).astype(np.float32)
    wp = pack_w(Wt, Cin, Cout, 7)
    out = np.full((B, H, W, Cout), 7.0, np.float16)
    L.cs_set_dma_deferred(deferred)
    try:
        rc = L.cs_conv_dual(lat.ctypes.data, Cin, 0, half.ctypes.data, 192, 32, up_c0, C_up, B, H, W, Cin, wp.ctypes.data, bias.ctypes.data, out.ctypes.data, Cout, 0, Cout, Cout, 1, 7, 0)
    finally:
        L.cs_set_dma_deferred(0)
    assert rc == 0, L.cs_last_error().decode()
    assert L.cs_last_kernel().decode() == "p8<256,256,64> 1x1 upsample-on-read"
    x = lat.astype(np.float32)
    x[..., up_c0:up_c0 + C_up] = np.repeat(np.repeat(half[..., 32:32 + C_up].astype(np.float32), 2, axis=1), 2, axis=2)
    ref = torch.nn.functional.conv2d(torch.from_numpy(x).permute(0, 3, 1, 2), torch.from_numpy(Wt.astype(np.float16).astype(np.float32)), torch.from_numpy(bias))
    ref = (ref * torch.sigmoid(ref)).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(out.astype(np.float32), ref, rtol=2e-3, atol=2e-3)


# the stride-2 weights-stationary kernel for the 64 -> 128 down-sampling layer (csrc/y7t_conv_ws_s2.hip, korder 8): B, H, W, act, extras.  2 x 32 output tiles; the host
# model's three workgroups walk ranges of them through the three-buffer patch ring; both DMA landing times, as for the ping-pong kernel
WS_S2_CASES = [
    (1, 4, 64, 1, {}),                                                                  # one tile
    (1, 12, 64, 1, {}),                                                                 # three tiles, one per workgroup: top / middle / bottom rows
    (2, 8, 128, 2, {"in_ld": 128, "in_coff": 64, "out_ld": 256, "out_coff": 128}),      # 2 images x 2 x 2 tiles = 8 tiles on 3 workgroups (the ring wraps), slices, LeakyReLU
    (1, 20, 192, 0, {}),                                                                # 5 x 3 = 15 tiles: left / interior / right columns, no activation
]


@pytest.mark.parametrize("deferred", [0, 1], ids=["dma-at-issue", "dma-at-wait"])
@pytest.mark.parametrize("case", WS_S2_CASES, ids=lambda c: "%dx%dx%d_act%d" % c[:4])
def test_stride2_weights_stationary_kernel_source_on_the_host(case, deferred):
    B, H, W, act, kw = case
    L = cs.lib()
    L.cs_set_dma_deferred(deferred)
    try:
        name = run_case(L, B, H, W, 64, 128, 3, 2, act, 0, korder=8, seed=B * 1000 + H + W, **kw)
    finally:
        L.cs_set_dma_deferred(0)
    assert name == "ws_s2<2,32>", name


@pytest.mark.parametrize("deferred", [0, 1], ids=["dma-at-issue", "dma-at-wait"])
@pytest.mark.parametrize("case", WS_S2_CASES, ids=lambda c: "%dx%dx%d_act%d" % c[:4])
def test_stride2_weights_stationary_kernel_with_the_twin_1x1_behind_it_on_the_host(case, deferred):
    """korder 11 (csrc/y7t_conv_ws_s2.hip, FUSE): 3x3 / stride 2 64 -> 128, activation, fp16 -- in the LDS gather, never in memory -- then the 128 -> 128 1x1 convolution
    (the two 1x1 layers that open the next ELAN block as one) + activation.  Reference: the two layers one after the other with the tensor between them rounded to fp16."""
    from yolov7_tracker_amd.detector import weights
    B, H, W, act, kw = case
    in_ld, in_coff = kw.get("in_ld", 64), kw.get("in_coff", 0)
    out_ld, out_coff = kw.get("out_ld", 128), kw.get("out_coff", 0)
    rng = np.random.default_rng(B * 1000 + H + W + 7)
    x = rng.normal(0, 1, (B, H, W, in_ld)).astype(np.float16)
    W1 = (rng.normal(0, 1, (128, 64, 3, 3)) / np.sqrt(576)).astype(np.float32)
    W2 = (rng.normal(0, 1, (128, 128, 1, 1)) / np.sqrt(128)).astype(np.float32)
    b1, b2 = rng.normal(0, 0.5, 128).astype(np.float32), rng.normal(0, 0.5, 128).astype(np.float32)
    blk1 = W1.transpose(0, 2, 3, 1).reshape(128, 576).astype(np.float16)
    blk2 = W2.reshape(128, 128).astype(np.float16)
    wp = np.concatenate([weights.pack_ws_s2(blk1).ravel(), weights.pack_ws_s2_tail(blk2).ravel()])
    bp = np.concatenate([b1, b2])
    Ho, Wo = H // 2, W // 2
    out = np.full((B, Ho, Wo, out_ld), 7.0, np.float16)
    L = cs.lib()
    L.cs_set_dma_deferred(deferred)
    try:
        rc = L.cs_conv(x.ctypes.data, in_ld, in_coff, B, H, W, 64, wp.ctypes.data, bp.ctypes.data, out.ctypes.data, out_ld, out_coff, 0, 128, 128, 3, 3, 2, 1, act, 11, 0, 0, 0)
    finally:
        L.cs_set_dma_deferred(0)
    assert rc == 0, L.cs_last_error().decode()
    assert L.cs_last_kernel().decode() == "ws_s2<2,32> + 1x1"
    f = (lambda t: torch.nn.functional.silu(t)) if act == 1 else (lambda t: torch.nn.functional.leaky_relu(t, 0.1)) if act == 2 else (lambda t: t)
    xt = torch.from_numpy(x[..., in_coff:in_coff + 64].astype(np.float32)).permute(0, 3, 1, 2)
    mid = f(torch.nn.functional.conv2d(xt, torch.from_numpy(W1.astype(np.float16).astype(np.float32)), torch.from_numpy(b1), 2, 1)).half().float()
    ref = f(torch.nn.functional.conv2d(mid, torch.from_numpy(W2.astype(np.float16).astype(np.float32)), torch.from_numpy(b2))).permute(0, 2, 3, 1).numpy()
    got = out[..., out_coff:out_coff + 128].astype(np.float32)
    np.testing.assert_allclose(got, ref, rtol=4e-3, atol=4e-3)       # fp16 output rounding + the 1-ulp differences of the fp16 tensor between the layers (summation order)
    other = np.ones(out_ld, bool)
    other[out_coff:out_coff + 128] = False
    assert np.all(out[..., other] == 7.0)


def test_stride2_weights_stationary_fragment_reads_are_bank_conflict_free_and_packing():
    """144-byte pixels: the 16 lanes of a ds_read_b128 service group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32) are 16 columns of ONE patch row -> 16 different
    16-byte bank quads for every tap plane and k-substep; and korder 8 is the documented permutation"""
    from yolov7_tracker_amd.detector import weights
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for hi in (0, 1):
        for g in groups:
            for plane in (0, 33 * 144, 144):
                for ks in range(4):
                    assert len({((l * 144 + hi * 16 + plane + ks * 32) // 16) % 16 for l in g}) == 16
    blk = np.random.default_rng(4).permutation(128 * 576).astype(np.float64).reshape(128, 576)
    out = weights.pack_ws_s2(blk).ravel()
    assert np.array_equal(np.sort(out), np.sort(blk.ravel()))
    blk2 = np.arange(128 * 128, dtype=np.float64).reshape(128, 128)          # the fused twin 1x1 bank: fragment (ks, q), lane l -> W2[q*32 + l%32][ks*16 + 8*(l//32) .. +7]
    t = weights.pack_ws_s2_tail(blk2).reshape(8, 4, 64, 8)
    for ks, q, l in ((0, 0, 0), (3, 2, 45), (7, 3, 63)):
        assert np.array_equal(t[ks, q, l], blk2[q * 32 + l % 32, ks * 16 + 8 * (l // 32):ks * 16 + 8 * (l // 32) + 8])
    for tap, ks, q, lane in ((0, 0, 0, 0), (8, 3, 3, 63), (4, 2, 1, 37), (7, 1, 2, 5)):
        f = (tap * 4 + ks) * 4 + q
        assert np.array_equal(out[(f * 64 + lane) * 8:(f * 64 + lane) * 8 + 8], blk[q * 32 + lane % 32, tap * 64 + ks * 16 + 8 * (lane // 32):][:8])


def test_pingpong_panel_packing_is_the_documented_lds_image():
    from yolov7_tracker_amd.detector import weights
    blk = np.random.default_rng(2).permutation(512 * 192).astype(np.float64).reshape(512, 192)
    out = weights.panel_pack_p8(blk).ravel()
    assert np.array_equal(np.sort(out), np.sort(blk.ravel()))
    for tile, kt, h, r, s_ in ((0, 0, 0, 0, 0), (1, 2, 1, 127, 7), (0, 1, 1, 37, 3), (1, 0, 0, 70, 5)):
        o = ((((tile * 3 + kt) * 2 + h) * 128 + r) * 8 + s_) * 8
        ch, oct_ = tile * 256 + (r // 32) * 64 + h * 32 + r % 32, s_ ^ ((r >> 1) & 7)
        assert np.array_equal(out[o:o + 8], blk[ch, kt * 64 + oct_ * 8:][:8])


# the LDS-patch kernels (csrc/y7t_conv_patch.hip; 3x3 / stride 1): B, H, W, Cin, Cout, act, korder, extras -- forced onto the patch kernel like the GPU layer tests
PATCH_CASES = [
    (1, 24, 32, 64, 128, 1, 0, {}),                                                   # 16x16 tiles, 128-channel panels
    (1, 16, 40, 128, 64, 2, 1, {"in_ld": 192, "in_coff": 64}),                        # 64-channel panels (multi-tile workgroups when forced), input slice
    (1, 37, 53, 64, 128, 1, 2, {"out_ld": 192, "out_coff": 64}),                      # ragged edges, panel-packed weights, output slice
    (2, 40, 40, 64, 128, 1, 2, {}),                                                   # strip tiling of the 40-wide map
    (3, 20, 20, 128, 64, 1, 1, {}),                                                   # strip tiling, 20-wide, 64 channels
    (2, 20, 20, 64, 128, 1, 9, {}),                                                   # korder 9: 64-row panels on a 128-channel layer, strip tiling
    (1, 24, 32, 64, 256, 2, 9, {"out_ld": 320, "out_coff": 64}),                      # ... 16x16 tiles, four panels, output slice
]


@pytest.mark.parametrize("case", PATCH_CASES, ids=lambda c: "%dx%dx%d_%d-%d_o%d" % (c[0], c[1], c[2], c[3], c[4], c[6]))
def test_patch_kernel_source_on_the_host(case):
    B, H, W, Cin, Cout, act, korder, kw = case
    name = run_case(cs.lib(), B, H, W, Cin, Cout, 3, 1, act, 0, korder=korder, force_patch=1, **kw)
    assert name.startswith("patch"), name


def test_patch_kernel_burst_reads_schedule_on_the_host():
    """Y7T_CONV_ABLATE=2048 in the MEASURING build (-DY7T_ABLATE_BUILD: the product library neither has these instances nor reads the variable; read once per
    process, hence the subprocess): the instances of the patch kernels that read a step's fragments as one burst behind the
    barrier (the form of rounds 1-3a, kept for A/B runs; the default spreads them over the step's MFMAs) -- every PATCH_CASES shape, same reference"""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tests import test_convsim as t, _convsim as cs\n"
            "for B, H, W, Cin, Cout, act, korder, kw in t.PATCH_CASES:\n"
            "    name = t.run_case(cs.lib(('-DY7T_ABLATE_BUILD',)), B, H, W, Cin, Cout, 3, 1, act, 0, korder=korder, force_patch=1, **kw)\n"
            "    assert name.endswith('burst-reads') or name.startswith('patch_mt'), name\n"
            "    print(name)\n" % root)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, Y7T_CONV_ABLATE="2048"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("burst-reads") >= 4, r.stdout


# the stride-2 LDS-patch kernel (csrc/y7t_conv_patch_s2.hip; korder 4): B, H, W, Cin, Cout, act, extras
S2_CASES = [
    (1, 16, 32, 64, 128, 1, {}),                                                      # one full 8 x 16 output tile, 128-channel panels (6-slot weight ring)
    (1, 16, 32, 64, 256, 1, {}),                                                      # 256-channel panels (3-slot ring, four waves along the channels)
    (2, 23, 45, 128, 128, 2, {"in_ld": 192, "in_coff": 64, "out_ld": 192, "out_coff": 64}),   # odd sizes: ragged tiles on both edges, slices, LeakyReLU
    (1, 20, 20, 64, 384, 1, {}),                                                      # Cout_pad = 384: 128-channel panels, three channel tiles
    (1, 18, 34, 192, 248, 0, {"out_ld": 256}),                                        # Cout not a multiple of the panel (padded rows), six chunk pairs, no activation
]


@pytest.mark.parametrize("case", S2_CASES, ids=lambda c: "%dx%dx%d_%d-%d" % (c[0], c[1], c[2], c[3], c[4]))
def test_stride2_patch_kernel_source_on_the_host(case):
    """csrc/y7t_conv_patch_s2.hip (parity-split patch columns, 16-channel chunks, panel-packed weights): its load geometry, plane addressing, ring
    positions and epilogue against a plain stride-2 convolution"""
    B, H, W, Cin, Cout, act, kw = case
    name = run_case(cs.lib(), B, H, W, Cin, Cout, 3, 2, act, 0, korder=4, **kw)
    cp = (Cout + 127) // 128 * 128
    assert name == "patch_s2<%d>" % (256 if cp % 256 == 0 else 128), name


def test_stride2_patch_kernel_rejects_what_it_cannot_run():
    """korder 4 weights are readable by the stride-2 patch kernel only: a stride-1 layer carrying them is an error, not a silent fallback"""
    L = cs.lib()
    x = np.zeros((1, 8, 8, 64), np.float16)
    w = np.zeros((128, 576), np.float16)
    b = np.zeros(128, np.float32)
    out = np.zeros((1, 8, 8, 128), np.float16)
    rc = L.cs_conv(x.ctypes.data, 64, 0, 1, 8, 8, 64, w.ctypes.data, b.ctypes.data, out.ctypes.data, 128, 0, 0, 128, 128, 3, 3, 1, 1, 1, 4, 0, 0, 0)
    assert rc != 0 and b"korder 4" in L.cs_last_error()


def test_stride2_patch_kernel_fragment_reads_are_bank_conflict_free():
    """Static model of the fragment addresses of csrc/y7t_conv_patch_s2.hip (the simulator does not model banks): ds_read_b128 is served in four groups of
    16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 (MI355X_MICROARCH.md, LDS) -- and a group is conflict-free when its 16 lanes
    touch 16 distinct 16-byte units mod 256 bytes.  Patch reads: 48-byte pixels, columns of the tile's second row rotated by 14; weight reads: 32-byte rows,
    half h of row r in slot h ^ ((r >> 3) & 1).  (Without the rotation the patch reads collide 2-way on two units -- checked here too, so that the
    model itself is known to see conflicts.)"""
    g1 = list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28))
    g2 = list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))
    groups = [g1, g2, [l + 32 for l in g1], [l + 32 for l in g2]]
    PIXB, RP = 48, 33 * 48

    def patch_addr(lane, rot):
        l31, hi32 = lane & 31, lane >> 5
        xl = (((l31 & 15) + rot) & 15) if l31 & 16 else (l31 & 15)
        return ((l31 >> 4) * 2) * RP + xl * PIXB + hi32 * 16

    def weight_addr(lane):
        l31, hi32 = lane & 31, lane >> 5
        return l31 * 32 + ((hi32 ^ ((l31 >> 3) & 1)) << 4)

    def conflict_free(addr):
        return all(len({(addr(l) // 16) % 16 for l in g}) == 16 for g in groups)

    assert conflict_free(lambda l: patch_addr(l, 14))
    assert not conflict_free(lambda l: patch_addr(l, 0))
    assert conflict_free(weight_addr)
    assert not conflict_free(lambda l: (l & 31) * 32 + (l >> 5) * 16)          # unswizzled 32-byte rows: 2-way
    # every immediate the kernel adds (tap row kh * RP, plane offsets 0 / 17 * 48 / 48, tile j * 4 * RP, buffers, ring slots) is a multiple of 16 bytes and
    # shifts all lanes alike, so it cannot create a conflict
    assert RP % 16 == 0 and (17 * PIXB) % 16 == 0 and PIXB % 16 == 0


# B, H, W, act, slices -- the weights-stationary 64 -> 64 kernel (csrc/y7t_conv_ws.hip; the fake device has 3 compute units, so a workgroup walks
# several tiles: the three-buffer patch ring wraps, tile ranges end unevenly, the last workgroups run out of tiles early)
WS_CASES = [
    (1, 16, 16, 1, {}),                                                                # one tile, one workgroup
    (1, 32, 48, 1, {}),                                                                # 6 tiles on 3 workgroups
    (2, 48, 32, 2, {"in_ld": 128, "in_coff": 64, "out_ld": 256, "out_coff": 128}),     # 12 tiles, two images, slices of concat buffers, LeakyReLU
    (1, 64, 80, 0, {"out_ld": 64}),                                                    # 20 tiles: 7 per workgroup (the ring wraps twice), the last one gets 6
    (5, 16, 16, 1, {}),                                                                # 5 tiles over 3 workgroups: 2, 2, 1
]


@pytest.mark.parametrize("case", WS_CASES, ids=lambda c: "%dx%dx%d_act%d" % c[:4])
def test_weights_stationary_kernel_source_on_the_host(case):
    B, H, W, act, kw = case
    name = run_case(cs.lib(), B, H, W, 64, 64, 3, 1, act, 0, korder=5, seed=B * 1000 + H + W, **kw)
    assert name == "ws64<16,16>", name


# the same kernel on the tile counter (DYN, round 5): the fake device runs its 3 workgroups one after the other, which is the most lopsided schedule there is --
# workgroup 0 takes its two static chunks and then EVERY chunk of the counter, workgroups 1 and 2 find it exhausted after their static chunks.  Cases: one chunk for one
# workgroup; an odd tile count (the last chunk holds one tile); fewer chunks than two per workgroup (chunk 1 dead from the start); many chunks (the ring of ids wraps);
# a map whose tile rows are odd (a chunk straddles two tile rows / two images)
WS_DYN_CASES = [WS_CASES[2], WS_CASES[4]] + [      # (two of the static form's cases: slices of concat buffers over two images; 5 one-tile images on 3 workgroups)
    (1, 16, 32, 1, {}),                                                                # 2 tiles = 1 chunk, one workgroup
    (3, 48, 48, 1, {}),                                                                # 27 tiles: 14 chunks, tile rows of 3 (chunks straddle rows and images), last chunk of one tile
    (1, 96, 64, 2, {"in_ld": 128, "in_coff": 64}),                                     # 24 tiles: 12 chunks, 6 static, 6 from the counter
]


@pytest.mark.parametrize("case", WS_DYN_CASES, ids=lambda c: "%dx%dx%d_act%d" % c[:4])
def test_weights_stationary_kernel_on_the_tile_counter_on_the_host(case):
    B, H, W, act, kw = case
    L = cs.lib()
    L.cs_set_dyn(1)
    try:
        name = run_case(L, B, H, W, 64, 64, 3, 1, act, 0, korder=5, seed=B * 1000 + H + W, **kw)
        assert name == "ws64<16,16> dyn", name
        assert all(L.cs_tile_counter(i) == 0 for i in range(8))          # the last workgroup to leave handed the counter back at zero
        name = run_case(L, B, H, W, 64, 64, 3, 1, act, 0, korder=5, seed=B * 1000 + H + W + 1, **kw)      # ... so the next launch starts from it
        assert name == "ws64<16,16> dyn" and all(L.cs_tile_counter(i) == 0 for i in range(8))
    finally:
        L.cs_set_dyn(0)


# B, H, W, Cout, act, slices -- the 128-channel sibling (csrc/y7t_conv_ws128.hip, korder 6): 4 x 16 tiles, four waves on the same 64 pixels with 32 output channels
# each, two channel tiles for Cout = 256 (the fake device's three workgroups split 2 + 1 over them)
WS128_CASES = [
    (1, 4, 16, 128, 1, {}),                                                            # one tile, one workgroup
    (1, 12, 32, 128, 1, {}),                                                           # 6 tiles on 3 workgroups: top / bottom / side borders
    (2, 16, 48, 128, 2, {"in_ld": 192, "in_coff": 64, "out_ld": 256, "out_coff": 128}),   # 24 tiles, two images, slices of wider buffers, LeakyReLU
    (1, 28, 32, 256, 1, {}),                                                           # 14 tiles x 2 channel tiles: workgroups 0 and 2 share channel tile 0, workgroup 1 walks all 14
    (3, 4, 16, 128, 0, {"out_ld": 128}),                                               # 3 tiles over 3 workgroups, no activation
]


@pytest.mark.parametrize("case,dyn", [(c, d) for i, c in enumerate(WS128_CASES) for d in (0, 1) if d or i in (1, 3)],      # every case on the tile counter; two of them also statically partitioned
                         ids=lambda v: ("%dx%dx%d_%d_act%d" % v[:5]) if isinstance(v, tuple) else ("tile_counter" if v else "static"))
def test_weights_stationary_128_kernel_source_on_the_host(case, dyn):
    """static partition (the single-layer entry point) and on the tile counter (a detector's plan): with Cout = 256 the two channel tiles draw from two counters"""
    B, H, W, Cout, act, kw = case
    L = cs.lib()
    L.cs_set_dyn(dyn)
    try:
        for rep in range(1 + dyn):      # (tile counter: the second launch starts from the counters the first one handed back)
            name = run_case(L, B, H, W, 128, Cout, 3, 1, act, 0, korder=6, seed=B * 1000 + H + W + rep, **kw)
            assert name == ("ws128<4,16> dyn" if dyn else "ws128<4,16>"), name
            assert all(L.cs_tile_counter(i) == 0 for i in range(8))
    finally:
        L.cs_set_dyn(0)


# the stride-2 form of the same kernel (S2: 2 x 16 output tiles, 5 x 33 patch with parity-split columns): B, H, W (input), Cout, act, slices
WS128_S2_CASES = [
    (1, 4, 32, 128, 1, {}),                                                            # one output tile (2 x 16): top / left padding only
    (1, 12, 64, 128, 1, {}),                                                           # 3 x 2 tiles on 3 workgroups: interior rows and columns
    (2, 8, 96, 256, 2, {"in_ld": 192, "in_coff": 64, "out_ld": 384, "out_coff": 128}),    # two images, two channel tiles, slices of wider buffers, LeakyReLU
    (3, 4, 32, 128, 0, {"out_ld": 128}),                                               # three one-tile images: chunks straddle images, no activation
]


@pytest.mark.parametrize("case,dyn", [(c, d) for i, c in enumerate(WS128_S2_CASES) for d in (0, 1) if d or i in (1, 2)],
                         ids=lambda v: ("%dx%dx%d_%d_act%d" % v[:5]) if isinstance(v, tuple) else ("tile_counter" if v else "static"))
def test_weights_stationary_128_stride2_kernel_source_on_the_host(case, dyn):
    B, H, W, Cout, act, kw = case
    L = cs.lib()
    L.cs_set_dyn(dyn)
    try:
        for rep in range(1 + dyn):
            name = run_case(L, B, H, W, 128, Cout, 3, 2, act, 0, korder=6, seed=B * 1000 + H + W + rep, **kw)
            assert name == ("ws128_s2<2,16> dyn" if dyn else "ws128_s2<2,16>"), name
            assert all(L.cs_tile_counter(i) == 0 for i in range(8))
    finally:
        L.cs_set_dyn(0)


def test_weights_stationary_128_fragment_reads_are_bank_conflict_free():
    """272-byte pixels (17 sixteen-byte slots), 5 KiB rows: the 16 addresses of a ds_read_b128 service group fall on 16 different 16-byte bank quads"""
    PIXB, RP = 272, 5120
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for hi in (0, 1):
        for g in groups:
            for kw in range(3):
                for ks in range(8):
                    addr = [((l >> 4) * RP + (l & 15) * PIXB + hi * 16 + kw * PIXB + ks * 32) for l in g]
                    assert len({(a // 16) % 16 for a in addr}) == 16
    assert RP % 256 == 0 and RP >= 18 * PIXB
    # stride-2 form: 9216-byte rows, an output row reads patch rows 2 r + kh, tap kw the even plane, the odd plane (behind 17 even pixels) or the even plane one column on
    RP2, O_OFF = 9216, 17 * PIXB
    for hi in (0, 1):
        for g in groups:
            for kwoff in (0, O_OFF, PIXB):
                for ks in range(8):
                    addr = [((l >> 4) * 2 * RP2 + (l & 15) * PIXB + hi * 16 + kwoff + ks * 32) for l in g]
                    assert len({(a // 16) % 16 for a in addr}) == 16
    assert RP2 % 256 == 0 and RP2 >= 33 * PIXB


def test_weights_stationary_kernel_rejects_what_it_cannot_run():
    """korder 5 weights are readable by that kernel only: a ragged map (its store count must be exact), other channel counts, a stride -> argument error"""
    L = cs.lib()
    x, w, b, out = np.zeros((1, 20, 16, 64), np.float16), np.zeros((64, 576), np.float16), np.zeros(64, np.float32), np.zeros((1, 20, 16, 64), np.float16)
    rc = L.cs_conv(x.ctypes.data, 64, 0, 1, 20, 16, 64, w.ctypes.data, b.ctypes.data, out.ctypes.data, 64, 0, 0, 64, 64, 3, 3, 1, 1, 1, 5, 0, 0, 0)
    assert rc != 0 and b"whole 16 x 16 tiles" in L.cs_last_error()


def test_weights_stationary_fragment_reads_are_bank_conflict_free():
    """ds_read_b128 is served in four groups of 16 lanes ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, + 32); with 144-byte pixels and a row pitch that is a
    multiple of 256 bytes the 16 addresses of a group fall on 16 different 16-byte bank quads (MI355X_MICROARCH.md, LDS)"""
    PIXB, RP = 144, 2816
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for hi in (0, 1):
        for g in groups:
            for kw in range(3):
                for ks in range(4):
                    addr = [((l >> 4) * RP + (l & 15) * PIXB + hi * 16 + kw * PIXB + ks * 32) for l in g]
                    assert len({(a // 16) % 16 for a in addr}) == 16
