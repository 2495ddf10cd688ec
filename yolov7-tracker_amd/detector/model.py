"""Detector seam of the reference: `attempt_load(weights)` -> callable whose output feeds
`non_max_suppression`, `scale_coords`, `check_img_size` -- /root/reference/models/experimental.py:83-106,
models/yolo.py:319-351, utils/general.py:123-128,319-340,607-695 -- running on the MI355X through liby7t.so.

The forward pass is a static launch list over an NHWC fp16 arena (detector/graph.py); the Detect decode is fused with
the candidate filter of NMS, so the reference's (B, 102000, 5+nc) tensor is never materialised: `model(img)` returns a
`HeadOutput` handle that `non_max_suppression` consumes on the device."""
import ctypes
import math
import os

import numpy as np
import torch

from .. import _lib
from . import arch, graph, weights

MAX_DET, MAX_NMS = 300, 30000   # utils/general.py:619-620


def make_divisible(x, divisor):
    return math.ceil(x / divisor) * divisor


def check_img_size(img_size, s=32):
    """utils/general.py:123-128"""
    new_size = make_divisible(img_size, int(s))
    if new_size != img_size:
        print('WARNING: --img-size %g must be multiple of max stride %g, updating to %g' % (img_size, s, new_size))
    return new_size


class HeadOutput:
    """what `model(img)[0]` stands for: the Detect outputs of one forward, still on the device.  Either the four raw head tensors
    (in the arena, or a staged private copy) or -- after a fused forward -- the candidate list the Detect epilogues wrote
    (post-processing set `pset`); the reference's (B, A, 5+nc) tensor is only built on request (`decoded()` / indexing / numpy)."""

    def __init__(self, det, B, img_shape, staged=None, fused=None, pset=0):
        self.det, self.B, self.img_shape = det, B, img_shape
        self.staged = staged        # (tensors, ctypes pointer array) of a private copy of the head buffers, or None = the arena
        self.fused = fused          # conf_thres the fused Detect epilogues filtered with, or None
        self.pset = pset
        self._epoch = det._epoch

    @property
    def shape(self):
        p = self.det.plan
        return (self.B, sum(p.det["na"] * h["ny"] * h["nx"] for h in p.heads), p.det["no"])

    def raw(self):
        """list of (B, na, ny, nx, no) float32 tensors == the second element the reference's Detect returns.  After a fused forward
        the head tensors do not exist yet: the four Detect 1x1 convs are re-run in plain mode on the activations still in the arena."""
        p, out = self.det.plan, []
        if self.fused is not None and self.staged is None:
            self.det._materialise_heads(self)
        for l, h in enumerate(p.heads):
            t = self.det.head_tensor(l, self.B).reshape(self.B, h["ny"], h["nx"], p.det["na"], p.det["no"])
            out.append(t.permute(0, 3, 1, 2, 4).contiguous())
        return out

    def decoded(self):
        """(B, A, no) float32 == the reference's `model(img)[0]` (models/yolo.py:39-57); convenience, not on the hot path"""
        p, z = self.det.plan, []
        anchors = torch.tensor(self.det.spec["anchors"], dtype=torch.float32, device="cuda").view(len(p.heads), -1, 2)
        for l, (h, x) in enumerate(zip(p.heads, self.raw())):
            ny, nx = h["ny"], h["nx"]
            yv, xv = torch.meshgrid(torch.arange(ny, device="cuda"), torch.arange(nx, device="cuda"), indexing="ij")
            grid = torch.stack((xv, yv), 2).view(1, 1, ny, nx, 2).float()
            y = x.sigmoid()
            y[..., 0:2] = (y[..., 0:2] * 2. - 0.5 + grid) * h["stride"]
            y[..., 2:4] = (y[..., 2:4] * 2) ** 2 * anchors[l].view(1, -1, 1, 1, 2)
            z.append(y.view(self.B, -1, p.det["no"]))
        return torch.cat(z, 1)

    # -- lazy tensor view: code written against the reference's `pred = model(img)[0]` tensor keeps working ------------------------
    def __getitem__(self, i):   # (`model(img)` is the tuple `(HeadOutput,)`, so `model(img)[0]` is this handle; indexing IT indexes the tensor)
        return self.decoded()[i]

    def tensor(self):
        return self.decoded()

    def cpu(self):
        return self.decoded().cpu()

    def float(self):
        return self.decoded()

    def numpy(self):
        return self.decoded().cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def __len__(self):
        return self.B

    @property
    def device(self):
        return torch.device("cuda")

    @property
    def dtype(self):
        return torch.float32


class Plan_PostSet:
    """workspace + outputs of one decode/NMS chain (y7t_det_postprocess)"""


class Detector:
    """spec: arch dict; state_dict: reference-style names (see detector/weights.py)."""

    def __init__(self, spec, state_dict=None, img_size=(1280, 1280), max_batch=1, max_cand=None, seed=0, bn_bias_mean=0.0, calib_image=None):
        _lib.require_gpu()
        self._L = _lib.load()
        self.spec = spec
        self.nodes, self.layer_out = graph.parse(spec)
        self.max_batch, self._max_cand_arg = int(max_batch), max_cand
        self.max_cand = 0
        self.names = [str(i) for i in range(spec["nc"])]
        self._sd, self._seed = state_dict, seed
        # seeded random weights only (state_dict None): see weights.random_state_dict / calibrate_bn
        self._bn_bias_mean, self._calib_image = float(bn_bias_mean), calib_image
        self._plans = {}
        self._epoch = 0             # forwards run so far (a HeadOutput is only valid against the arena of its own forward)
        self.plan = None
        self._handle = None
        strides = None
        self.stride = None
        self._select(tuple(img_size))
        self.stride = torch.tensor([float(h["stride"]) for h in self.plan.heads])

    # -- plan / arena per input size ---------------------------------------------------------------
    def _select(self, hw):
        if self.plan is not None and (self.plan.H, self.plan.W) == hw:
            return
        if hw not in self._plans:
            s = int(max(8 * 2 ** (len(self.spec["anchors"]) - 1), 32))
            if hw[0] % s or hw[1] % s:
                raise ValueError("input %dx%d is not a multiple of the max stride %d" % (hw[0], hw[1], s))
            nodes, _ = graph.parse(self.spec)
            plan = graph.lower(nodes, hw[0], hw[1], self.max_batch)
            # the kernels address a tensor through 32-bit byte offsets (2 GiB).  A conv whose input or output tensor of `max_batch` frames is larger goes out as
            # several launches over runs of frames (csrc/y7t_detector.hip::forward_impl; w6 @ 1280: the 640^2 / 320^2 layers above 40 frames); the pools, the
            # materialised upsample and the input layout kernel are single launches
            one = max(e * isz for e, isz in plan.buf_elems)
            if one >= 1 << 31:
                raise ValueError("%dx%d: the largest activation tensor of ONE frame would be %.1f GiB (limit 2 GiB)" % (hw[0], hw[1], one / 2 ** 30))
            single = [max(int(o["H"]) * int(o["W"]) * int(o["in_ld"]), int(o["Ho"]) * int(o["Wo"]) * int(o["out_ld"])) * 2 for o in plan.ops if int(o["type"]) != 0]
            single.append(plan.buf_elems[0][0] * plan.buf_elems[0][1])
            if max(single) * self.max_batch >= 1 << 31:
                raise ValueError("max_batch=%d at %dx%d: a tensor of a single-launch op (pool / upsample / input layout) would be %.1f GiB (limit 2 GiB); "
                                 "split the batch (max_batch <= %d)" % (self.max_batch, hw[0], hw[1], max(single) * self.max_batch / 2 ** 30,
                                                                         ((1 << 31) - 1) // max(single)))
            if self._sd is None:
                self._sd = weights.calibrate_bn(nodes, weights.random_state_dict(plan.wlayout, self._seed, bn_bias_mean=self._bn_bias_mean),
                                                seed=self._seed, image=self._calib_image)
            wb, bb = weights.pack(plan.wlayout, self._sd, plan.w_elems, plan.b_elems)
            plan.w_dev = torch.from_numpy(wb.view(np.int16)).cuda()
            plan.b_dev = torch.from_numpy(bb).cuda()
            plan.arena = torch.zeros(plan.arena_bytes, dtype=torch.uint8, device="cuda")
            h = ctypes.c_void_p()
            _lib.check(self._L.y7t_det_create(plan.ops.ctypes.data_as(ctypes.c_void_p), len(plan.ops),
                                              plan.buf_offsets.ctypes.data_as(ctypes.c_void_p), len(plan.buf_offsets),
                                              _lib.ptr(plan.arena), plan.arena_bytes, _lib.ptr(plan.w_dev), _lib.ptr(plan.b_dev),
                                              self.max_batch, ctypes.byref(h)))
            plan.handle = h
            n_anchors = sum(plan.det["na"] * hd["ny"] * hd["nx"] for hd in plan.heads)
            # candidate capacity: every anchor fits (no overflow, like the reference); NMS works on the top MAX_NMS of them
            plan.cap = int(self._max_cand_arg) if self._max_cand_arg else (n_anchors + 63) // 64 * 64
            B, cap = self.max_batch, plan.cap
            ws_bytes = int(self._L.y7t_det_postprocess_workspace_bytes(B, cap, MAX_NMS))
            plan.post = []          # two post-processing sets: decode+NMS of batch n can run beside the forward of batch n+1
            for _ in range(2):
                ps = Plan_PostSet()
                ps.ws = torch.zeros(ws_bytes, dtype=torch.uint8, device="cuda")
                ps.dets = torch.zeros((B, MAX_DET, 6), dtype=torch.float32, device="cuda")
                ps.ndets = torch.zeros(B, dtype=torch.int32, device="cuda")
                ps.keep = torch.zeros((B, MAX_DET), dtype=torch.int32, device="cuda")
                ps.cand = torch.zeros(B, dtype=torch.int32, device="cuda")
                plan.post.append(ps)
            plan.ws, plan.dets, plan.ndets, plan.keep, plan.cand = (getattr(plan.post[0], k) for k in ("ws", "dets", "ndets", "keep", "cand"))
            plan.lb = torch.zeros((B, 5), dtype=torch.float32, device="cuda")
            plan.head_ptrs = (ctypes.c_void_p * 4)(*[plan.arena.data_ptr() + int(plan.buf_offsets[hd["buf"]]) for hd in plan.heads] +
                                                   [None] * (4 - len(plan.heads)))
            plan.ny = (ctypes.c_int * 4)(*[hd["ny"] for hd in plan.heads] + [0] * (4 - len(plan.heads)))
            plan.nx = (ctypes.c_int * 4)(*[hd["nx"] for hd in plan.heads] + [0] * (4 - len(plan.heads)))
            plan.strides = (ctypes.c_float * 4)(*[float(hd["stride"]) for hd in plan.heads] + [0.0] * (4 - len(plan.heads)))
            flat = [float(v) for lvl in self.spec["anchors"] for v in lvl]
            plan.anchors = (ctypes.c_float * 24)(*(flat + [0.0] * (24 - len(flat))))
            _lib.check(self._L.y7t_det_set_detect(h, len(plan.heads), plan.det["na"], plan.det["no"], plan.strides, plan.anchors))
            plan.detect_ops = [i for i, op in enumerate(plan.ops) if int(op["type"]) == 0 and int(op["detect_level"]) >= 0]
            plan.stem_fused = bool(self._L.y7t_det_stem_fusable(h)) and _lib.switch("Y7T_STEM_FUSED", "1") != "0"
            plan.fusable = plan.det["na"] * plan.det["no"] <= 64 and all(int(plan.ops[i]["Cin"]) % 64 == 0 for i in plan.detect_ops)
            self._plans[hw] = plan
        self.plan = self._plans[hw]
        self.max_cand = self.plan.cap

    def head_tensor(self, level, B):
        p = self.plan
        h = p.heads[level]
        n = h["ny"] * h["nx"] * p.det["na"] * p.det["no"]
        off = int(p.buf_offsets[h["buf"]])
        return p.arena[off:off + 4 * n * B].view(torch.float32).view(B, h["ny"], h["nx"], p.det["na"] * p.det["no"])

    def buffer_view(self, buf, B, C):
        """NHWC fp16 view of an arena buffer (tests)"""
        p = self.plan
        elems = p.buf_elems[buf][0]
        off = int(p.buf_offsets[buf])
        return p.arena[off:off + 2 * elems * B].view(torch.float16).view(B, -1, C)

    # -- forward ---------------------------------------------------------------------------------------
    def _run_ops(self, B, first, last, fuse_decode, pset):
        p, s = self.plan, _lib.stream_ptr()
        if fuse_decode is None:
            _lib.check(self._L.y7t_det_forward_ops(p.handle, B, int(first), int(last), s))
        else:
            ps = p.post[pset]
            ps.last_B = B
            _lib.check(self._L.y7t_det_forward_fused(p.handle, B, int(first), int(last), float(fuse_decode), self.max_cand, MAX_NMS, _lib.ptr(ps.ws),
                                                     ps.ws.numel(), s))

    def forward(self, img, mid_hook=None, fuse_decode=None, pset=0):
        """img: (B,3,H,W) float32 RGB in [0,1] (the reference's input) or (B,H,W,3) uint8 BGR frames (fused
        BGR->RGB, /255).  -> HeadOutput.
        fuse_decode = conf_thres: the Detect 1x1 convs decode + filter in their epilogue straight into the candidate arrays of
        post-processing set `pset` (y7t_det_forward_fused); the head tensors are not written and `postprocess` skips its decode
        pass.  The threshold is fixed at forward time -- the tracking path always uses 0.01 (tracker/track.py:239)."""
        if img.dim() == 3:
            img = img[None]
        if img.device.type != "cuda":
            img = img.cuda(non_blocking=True)
        is_u8 = img.dtype == torch.uint8
        if is_u8:
            B, H, W = img.shape[0], img.shape[1], img.shape[2]
        else:
            img = img.float()
            B, H, W = img.shape[0], img.shape[2], img.shape[3]
        img = img.contiguous()
        if B > self.max_batch:
            raise ValueError("batch %d > max_batch %d" % (B, self.max_batch))
        self._select((H, W))
        p = self.plan
        if fuse_decode is not None and not p.fusable:
            fuse_decode = None          # e.g. nc = 80: 255 head channels do not fit one channel tile -> plain heads + decode pass
        first = self._front(img, is_u8, B, H, W, fuse_decode, pset)
        if mid_hook is None:
            self._run_ops(B, first, -1, fuse_decode, pset)
        else:   # (op index, callable): run the list up to that op, call the hook (e.g. record an event), run the rest
            k = max(first, min(int(mid_hook[0]), int(self._L.y7t_det_num_ops(p.handle))))
            self._run_ops(B, first, k, fuse_decode, pset)
            mid_hook[1]()
            self._run_ops(B, k, -1, fuse_decode, pset)
        self._img_keep = img
        self._epoch += 1
        return HeadOutput(self, B, (H, W), fused=fuse_decode, pset=pset)

    def _front(self, img, is_u8, B, H, W, fuse_decode, pset):
        """input side of the forward for frames of the plan's size: uint8 frames of a ReOrg + stem plan go through the fused
        frame -> stem kernel (no layout tensor), everything else through y7t_input_layout.  -> index of the first op still to run"""
        p, s = self.plan, _lib.stream_ptr()
        if is_u8 and p.stem_fused:
            if fuse_decode is not None:     # the fused Detect epilogues append to counters that op 0 would have zeroed
                _lib.check(self._L.y7t_det_forward_fused(p.handle, B, 0, 0, float(fuse_decode), self.max_cand, MAX_NMS, _lib.ptr(p.post[pset].ws),
                                                         p.post[pset].ws.numel(), s))
            _lib.check(self._L.y7t_det_forward_stem_u8(p.handle, _lib.ptr(img), B, H, W, H, W, 0, 0, s))
            return 1
        _lib.check(self._L.y7t_input_layout(_lib.ptr(img), int(is_u8), B, H, W, int(p.reorg), _lib.ptr(p.arena), p.in_ld, s))
        return 0

    def forward_part(self, img, first, last, fuse_decode=None, pset=0):
        """ops [first, last) of the current plan's launch list on the current stream (last < 0: to the end); with `img` (uint8
        NHWC or float32 NCHW device tensor of the plan's size) the input layout runs first.  For callers that capture the
        forward in pieces (bench.py --hipgraph 2); `forward` must have selected the plan before."""
        p, s = self.plan, _lib.stream_ptr()
        fd = fuse_decode if p.fusable else None
        if img is not None:
            is_u8 = img.dtype == torch.uint8
            B = img.shape[0]
            H, W = (img.shape[1], img.shape[2]) if is_u8 else (img.shape[2], img.shape[3])
            first = max(int(first), self._front(img, is_u8, B, H, W, fd, pset))
            self._part_B = B
        self._run_ops(self._part_B, first, last, fd, pset)

    def _materialise_heads(self, out):
        """re-run the Detect 1x1 convs of `out`'s forward in plain mode (fp32 head tensors into the arena)"""
        if out._epoch != self._epoch:
            raise _lib.Y7TError("the activations of this forward have been overwritten by a later one: the raw heads of a fused forward "
                                "can only be materialised before the next forward")
        p = self.plan
        for i in p.detect_ops:
            _lib.check(self._L.y7t_det_forward_ops(p.handle, out.B, i, i + 1, _lib.stream_ptr()))

    def plant_objectness_bias(self, frames, target=2000, level_offsets=None, level_quota=None):
        """No trained checkpoint ships with the reference, and a randomly initialised Detect head fires on ~half of the 102 000
        anchors.  SURVEY.md 8d: shift the Detect objectness biases so that ~`target` anchors of frames[0] exceed conf_thres = 0.01 (a
        typical VisDrone candidate load) and raise the class logits so the best class passes too.  Updates the state dict AND the
        device bias blob, so an oracle run on `self._sd` sees the same network.  level_offsets: extra objectness shift per Detect level
        (e.g. (0, 0, -3, -6): candidates mostly from the fine levels, the small-object regime of VisDrone).  level_quota: instead of ONE shift for all levels (a
        level whose logits are narrower than the others' then contributes nothing), every level gets its own shift so that it supplies that fraction of `target`
        (e.g. (0.55, 0.2, 0.15, 0.1): all four levels live, weighted towards the fine ones).  -> the objectness shift (of level 0 with quotas)."""
        out = self(frames[:1])[0]
        torch.cuda.synchronize()
        p = self.plan
        no, na = p.det["no"], p.det["na"]
        lo = [0.0] * len(p.heads) if level_offsets is None else [float(v) for v in level_offsets]
        per_level = [self.head_tensor(l, 1).view(-1, na, no)[..., 4].reshape(-1).float().cpu() + lo[l] for l in range(len(p.heads))]
        if level_quota is not None:
            assert len(level_quota) == len(p.heads) and abs(sum(level_quota) - 1.0) < 1e-6
            shifts = [float(np.log(0.01 / 0.99)) - torch.quantile(x, max(0.0, 1.0 - level_quota[l] * target / x.numel())).item() for l, x in enumerate(per_level)]
        else:
            logits = torch.cat(per_level)
            q = torch.quantile(logits, 1.0 - target / logits.numel()).item()
            shifts = [float(np.log(0.01 / 0.99)) - q] * len(p.heads)
        shift = shifts[0]
        self._sd = dict(self._sd)
        for pl in self._plans.values():
            for w in pl.wlayout:
                if w["kind"] == "conv":
                    continue
                delta = torch.zeros(na * no)
                if w["kind"] != "Detect":
                    raise NotImplementedError("plant_objectness_bias: plain Detect heads only (implicit layers are folded into the blob)")
                for a in range(na):
                    delta[a * no + 4] = shifts[w["level"]] + lo[w["level"]]
                    delta[a * no + 5:(a + 1) * no] = 4.0
                if pl is p:
                    self._sd[w["wkey"] + ".bias"] = self._sd[w["wkey"] + ".bias"].float() + delta
                pl.b_dev[w["b_off"]:w["b_off"] + na * no] += delta.cuda()
        return shift

    def launch_list(self, B=None):
        """Kernel variant of every op of the current plan at batch B (default: max_batch), in launch order: runs the ops one by one on
        whatever the arena holds and reads y7t_last_kernel() back.  The dispatch rules (LDS-patch / strip / multi-tile / 256-pixel
        tiles / split-K) depend on B, so this is how tests and bench.py state WHICH launch list they ran."""
        p, s = self.plan, _lib.stream_ptr()
        B = self.max_batch if B is None else int(B)
        names = []
        img = getattr(self, "_img_keep", None)
        u8_native = img is not None and img.dtype == torch.uint8 and tuple(img.shape[1:3]) == (p.H, p.W) and img.shape[0] >= B
        for i in range(int(self._L.y7t_det_num_ops(p.handle))):
            if i == 0 and p.stem_fused and u8_native:
                _lib.check(self._L.y7t_det_forward_stem_u8(p.handle, _lib.ptr(img), B, p.H, p.W, p.H, p.W, 0, 0, s))   # how uint8 frames enter
            else:
                _lib.check(self._L.y7t_det_forward_ops(p.handle, B, i, i + 1, s))
            names.append(self._L.y7t_last_kernel().decode())
        torch.cuda.synchronize()
        return names

    def launches_per_op(self, B=None):
        """kernel launches of every op of the current plan at batch B: 1, or the number of runs of frames a conv whose input or output tensor of B frames passes
        2 GiB goes out in (the rule of csrc/y7t_detector.hip::forward_impl, restated: the per-op table of the profiles attributes dispatches to ops with it)"""
        B = self.max_batch if B is None else int(B)
        lim, out = (1 << 31) - 1, []
        for i, o in enumerate(self.plan.ops):
            n = 1
            if int(o["type"]) == 0:
                f_in = int(o["H"]) * int(o["W"]) * int(o["in_ld"]) * 2
                f_out = int(o["Ho"]) * int(o["Wo"]) * int(o["out_ld"]) * (4 if int(o["out_f32"]) else 2)
                if i == 0 and self.plan.stem_fused:      # (the uint8 frame is the stem kernel's input)
                    f_in = self.plan.H * self.plan.W * 3
                runs = -(-B * max(f_in, f_out) // lim)
                n = -(-B // -(-B // runs))
            out.append(n)
        return out

    @staticmethod
    def letterbox_params(shape, new_shape, stride, auto=True, scaleup=True):
        """tracker_dataloader.py:100-130 -> (H, W of the letterboxed image, new_h, new_w, top, left)"""
        if isinstance(new_shape, int):
            new_shape = (new_shape, new_shape)
        r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
        if not scaleup:
            r = min(r, 1.0)
        new_w, new_h = int(round(shape[1] * r)), int(round(shape[0] * r))
        dw, dh = new_shape[1] - new_w, new_shape[0] - new_h
        if auto:
            dw, dh = np.mod(dw, stride), np.mod(dh, stride)
        dw /= 2
        dh /= 2
        top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
        left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
        return new_h + top + bottom, new_w + left + right, new_h, new_w, top, left

    def forward_frames(self, frames, img_size=1280, fuse_decode=None, pset=0):
        """raw (B,H0,W0,3) uint8 BGR frames (device or host) -> letterbox + layout on the device + forward.
        -> (HeadOutput, letterboxed (H, W)).  What TrackerLoader.__getitem__ + model(img) do in the reference."""
        if frames.dim() == 3:
            frames = frames[None]
        frames = frames.cuda(non_blocking=True).contiguous()
        B, H0, W0 = frames.shape[0], frames.shape[1], frames.shape[2]
        stride = int(self.stride.max())
        H, W, new_h, new_w, top, left = self.letterbox_params((H0, W0), img_size, stride)
        self._select((H, W))
        p = self.plan
        s = _lib.stream_ptr()
        if fuse_decode is not None and not p.fusable:
            fuse_decode = None
        first = 0
        if p.stem_fused:                # letterbox + layout + stem conv in one kernel
            if fuse_decode is not None:
                _lib.check(self._L.y7t_det_forward_fused(p.handle, B, 0, 0, float(fuse_decode), self.max_cand, MAX_NMS, _lib.ptr(p.post[pset].ws),
                                                         p.post[pset].ws.numel(), s))
            _lib.check(self._L.y7t_det_forward_stem_u8(p.handle, _lib.ptr(frames), B, H0, W0, new_h, new_w, top, left, s))
            first = 1
        else:
            _lib.check(self._L.y7t_letterbox_layout_u8(_lib.ptr(frames), B, H0, W0, H, W, new_h, new_w, top, left, int(p.reorg), _lib.ptr(p.arena),
                                                       p.in_ld, s))
        self._run_ops(B, first, -1, fuse_decode, pset)
        self._img_keep = frames
        self._epoch += 1
        return HeadOutput(self, B, (H, W), fused=fuse_decode, pset=pset), (H, W)

    def __call__(self, img, augment=False):
        return (self.forward(img),)

    def postprocess(self, out, conf_thres=0.01, iou_thres=0.45, ori_shapes=None):
        """decode + NMS + scale_coords + round on the device.  ori_shapes: list of (H0, W0) per image (None: no rescale).
        -> (dets (B, 300, 6) float32 device tensor, ndets (B,) int32 device tensor); asynchronous."""
        p, B = self.plan, out.B
        H, W = out.img_shape
        lb = np.zeros((B, 5), np.float32)
        for b in range(B):
            h0, w0 = (H, W) if ori_shapes is None else ori_shapes[b][:2]
            gain = min(H / h0, W / w0)
            lb[b] = (gain, (W - w0 * gain) / 2, (H - h0 * gain) / 2, h0, w0)
        key = lb.tobytes()
        if getattr(p, "_lb_key", None) != key:
            p.lb[:B].copy_(torch.from_numpy(lb))
            p._lb_key = key
        ps = p.post[out.pset]
        ps.last_B = B                    # the workspace arrays are laid out for THIS batch (y7t_post_cand_ws): candidate_arrays needs it
        if out.fused is not None:
            if abs(float(conf_thres) - float(out.fused)) > 1e-12:
                raise ValueError("this forward fused the Detect decode with conf_thres=%g; post-processing it with %g needs a plain forward"
                                 % (out.fused, conf_thres))
            head_ptrs = None            # the candidates are already in ps.ws
        else:
            head_ptrs = out.staged[1] if getattr(out, "staged", None) is not None else p.head_ptrs
        _lib.check(self._L.y7t_det_postprocess(head_ptrs, p.ny, p.nx, p.strides, p.anchors, len(p.heads), p.det["na"], p.det["no"], B,
                                               float(conf_thres), float(iou_thres), MAX_DET, MAX_NMS, self.max_cand, _lib.ptr(p.lb),
                                               _lib.ptr(ps.dets), _lib.ptr(ps.ndets), _lib.ptr(ps.keep), _lib.ptr(ps.cand), _lib.ptr(ps.ws),
                                               ps.ws.numel(), _lib.stream_ptr()))
        return ps.dets, ps.ndets

    def candidate_arrays(self, pset=0, B=None):
        """device views of post-processing set `pset`'s candidate arrays (the head of y7t_det_postprocess' workspace, each 256-byte aligned):
        -> (cbox (B, cap, 4) f32 xyxy, cscore (B, cap) f32, ccls (B, cap) f32, cidx (B, cap) i32 anchor row, count (B,) i32).  What the fused Detect
        epilogues / the decode pass wrote = `x` of utils/general.py:662 before the NMS; for parity checks and callers that want raw candidates."""
        p = self.plan
        # y7t_post_cand_ws lays the arrays out with the batch of the forward / post-process that filled them, not with max_batch
        B = B if B is not None else getattr(p.post[pset], "last_B", self.max_batch)
        if not 1 <= B <= self.max_batch:
            raise ValueError("candidate_arrays: batch %d outside [1, %d]" % (B, self.max_batch))
        cap, ws = self.max_cand, p.post[pset].ws
        rup = lambda n: (n + 255) // 256 * 256
        o1 = rup(B * cap * 16); o2 = o1 + rup(B * cap * 4); o3 = o2 + rup(B * cap * 4); o4 = o3 + rup(B * cap * 4)
        return (ws[:B * cap * 16].view(torch.float32).view(B, cap, 4), ws[o1:o1 + B * cap * 4].view(torch.float32).view(B, cap),
                ws[o2:o2 + B * cap * 4].view(torch.float32).view(B, cap), ws[o3:o3 + B * cap * 4].view(torch.int32).view(B, cap),
                ws[o4:o4 + B * 4].view(torch.int32))

    def stage_heads(self, out):
        """Copy the four raw head buffers of `out` (a few MB per frame, device to device, on the current stream) into a staging
        set owned by the plan and return a HeadOutput that reads from it.  With it decode+NMS of batch n can run on another
        stream while the forward of batch n+1 already rewrites the arena; the caller orders re-use of the staging set (the next
        stage_heads must come after the postprocess that read it -- bench.py does it with one event)."""
        p = self.plan
        if not hasattr(p, "_stage"):
            ts = [torch.empty_like(self.head_tensor(l, self.max_batch)) for l in range(len(p.heads))]
            p._stage = (ts, (ctypes.c_void_p * 4)(*[t.data_ptr() for t in ts] + [None] * (4 - len(ts))))
        for l, t in enumerate(p._stage[0]):
            t[:out.B].copy_(self.head_tensor(l, out.B), non_blocking=True)
        return HeadOutput(self, out.B, out.img_shape, staged=p._stage, pset=out.pset)

    def capture(self, img, conf_thres=0.01, iou_thres=0.45, ori_shapes=None):
        """Capture input layout + the whole conv launch list + decode/NMS for the (fixed) device buffer `img` into a
        hipGraph (torch.cuda.CUDAGraph drives the capture; every launch of liby7t.so goes to the capturing stream, and
        nothing in the list allocates or synchronises).  Returns (graph, dets, ndets): `graph.replay()` re-runs the chain
        on whatever `img` holds, with no per-launch host cost."""
        out = self.forward(img, fuse_decode=conf_thres)   # warm-up: plan selection, attribute setup, letterbox upload
        self.postprocess(out, conf_thres, iou_thres, ori_shapes)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = self.forward(img, fuse_decode=conf_thres)
            dets, nd = self.postprocess(out, conf_thres, iou_thres, ori_shapes)
        return g, dets, nd

    def check_overflow(self):
        c = max(int(ps.cand.max().item()) for ps in self.plan.post)
        if c > self.max_cand:
            raise _lib.Y7TError("%d NMS candidates exceed max_cand=%d" % (c, self.max_cand))

    def eval(self):
        return self

    def float(self):
        return self

    def fuse(self):
        return self

    def to(self, *a, **k):
        return self

    @property
    def gflop_per_frame(self):
        return 2 * self.plan.macs / 1e9


def non_max_suppression(prediction, conf_thres=0.25, iou_thres=0.45, classes=None, agnostic=False, multi_label=False, labels=()):
    """utils/general.py:607-695 for the arguments the tracking path uses (classes=None, agnostic=False, multi_label=False).
    prediction: the HeadOutput returned by Detector.  -> list of (n, 6) tensors [xyxy, conf, cls], score-descending."""
    if not isinstance(prediction, HeadOutput):
        raise TypeError("non_max_suppression expects the detector's HeadOutput (the decoded tensor is only materialised on request: "
                        "HeadOutput.decoded())")
    if classes is not None or agnostic or multi_label or labels:
        raise NotImplementedError("only the tracking path's NMS arguments are implemented")
    det = prediction.det
    dets, nd = det.postprocess(prediction, conf_thres, iou_thres, None)
    nd = nd.cpu().numpy()
    det.check_overflow()
    out = []
    for b in range(prediction.B):
        d = dets[b, :nd[b]].clone()
        d[:, :4] = det.plan_raw_boxes(b, int(nd[b]), prediction.pset)   # un-rounded boxes: rounding is the caller's (track.py:240)
        out.append(d)
    return out


def scale_coords(img1_shape, coords, img0_shape, ratio_pad=None):
    """utils/general.py:319-340 (in place, torch)"""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain, pad = ratio_pad[0][0], ratio_pad[1]
    coords[:, [0, 2]] -= pad[0]
    coords[:, [1, 3]] -= pad[1]
    coords[:, :4] /= gain
    coords[:, 0].clamp_(0, img0_shape[1])
    coords[:, 1].clamp_(0, img0_shape[0])
    coords[:, 2].clamp_(0, img0_shape[1])
    coords[:, 3].clamp_(0, img0_shape[0])
    return coords


def _plan_raw_boxes(self, b, n, pset=0):
    """xyxy of the kept detections before scale_coords/round (candidate boxes gathered by the kept slots)"""
    ps = self.plan.post[pset]
    cap = self.max_cand
    cbox = ps.ws[:self.max_batch * cap * 16].view(torch.float32).view(self.max_batch, cap, 4)
    idx = ps.keep[b, :n].long()
    return cbox[b, idx]


Detector.plan_raw_boxes = _plan_raw_boxes


def checkpoint_anchors(sd, spec):
    """Anchors the checkpoint's Detect module decodes with (models/yolo.py:54,100: `self.anchor_grid`), in pixels per level, or None.
    train.py's autoanchor (utils/autoanchor.py:40-60) and check_anchor_order rewrite these BUFFERS without touching model.yaml, so a
    trained checkpoint's buffers win over the yaml.  `model.N.anchor_grid` (nl,1,na,1,1,2) is in pixels; a checkpoint that only carries
    `model.N.anchors` (nl,na,2) holds them divided by the level's stride (yolo.py:229-231)."""
    grid = next((k for k in sd if k.endswith(".anchor_grid")), None)
    if grid is not None:
        a = sd[grid].detach().float().cpu().reshape(sd[grid].shape[0], -1)
        return [[float(v) for v in lvl] for lvl in a]
    anc = next((k for k in sd if k.endswith(".anchors") and sd[k].dim() == 3), None)
    if anc is not None:
        a = sd[anc].detach().float().cpu()
        probe = dict(spec, anchors=[[0.0] * (2 * a.shape[1])] * a.shape[0])
        strides = [h["stride"] for h in graph.lower(graph.parse(probe)[0], 256, 256, 1).heads]
        return [[float(v) * st for v in lvl.reshape(-1)] for lvl, st in zip(a, strides)]
    return None


def load_checkpoint(weights_path, cfg=None, nc=None):
    """Host half of attempt_load (models/experimental.py:83-106): -> (spec, state dict or None, seed).  `weights_path`: a .pt holding a
    state dict / {'model': state_dict} / a pickled reference Model (its classes must be importable, as in the reference's own
    `torch.load`; `ema` preferred like experimental.py:88-89), or 'random:<arch>[:seed]'."""
    if isinstance(weights_path, (list, tuple)):
        weights_path = weights_path[0]
    sd, seed = None, 0
    if str(weights_path).startswith("random:"):
        parts = str(weights_path).split(":")
        name = parts[1]
        seed = int(parts[2]) if len(parts) > 2 else 0
        return arch.ARCHS[name](nc if nc is not None else 80), None, seed
    if not os.path.isfile(weights_path):
        raise FileNotFoundError(weights_path)
    ck = torch.load(weights_path, map_location="cpu", weights_only=False)
    m = (ck.get("ema") or ck.get("model")) if isinstance(ck, dict) and ("model" in ck or "ema" in ck) else ck
    if hasattr(m, "state_dict"):
        sd = {k: v.float() for k, v in m.float().state_dict().items()}
        spec_yaml = getattr(m, "yaml", None)
    else:
        sd, spec_yaml = {k: v.float() for k, v in m.items()}, None
    if cfg is not None:
        spec = arch.load_yaml(cfg, nc) if os.path.isfile(str(cfg)) else arch.ARCHS[cfg](nc if nc is not None else 80)
    elif spec_yaml is not None:
        spec = {"nc": spec_yaml["nc"], "depth_multiple": spec_yaml.get("depth_multiple", 1.0), "width_multiple": spec_yaml.get("width_multiple", 1.0),
                "anchors": spec_yaml["anchors"], "layers": list(spec_yaml["backbone"]) + list(spec_yaml["head"])}
    else:
        raise ValueError("a state-dict checkpoint needs cfg=<yaml path or arch name>")
    spec = dict(spec)
    if isinstance(spec["anchors"], int):          # `anchors: 3` = "let autoanchor decide" (models/yolo.py:449): only the buffers know
        na = spec["anchors"]
        nl = next((sd[k].shape[0] for k in sd if k.endswith(".anchors") and sd[k].dim() == 3), None)
        if nl is None:
            raise ValueError("the model yaml gives only an anchor COUNT and the checkpoint has no Detect anchor buffers")
        spec["anchors"] = [[0.0] * (2 * na)] * nl
    a = checkpoint_anchors(sd, spec)
    if a is not None:
        spec["anchors"] = a
    elif any(v == 0 for lvl in spec["anchors"] for v in lvl):
        raise ValueError("no anchors: the yaml has a count only and the checkpoint carries none")
    return spec, sd, seed


def attempt_load(weights_path, map_location=None, cfg=None, nc=None, img_size=1280, max_batch=1):
    """models/experimental.py:83-106 seam.  `weights_path`: a .pt holding a state dict / {'model': state_dict} /
    (when the reference's classes are importable) a pickled reference Model; or 'random:<arch>[:seed]' for seeded
    random weights of a named architecture (no trained checkpoint ships with the reference).  Never touches the
    network (the reference's attempt_download would).  Decode anchors: the checkpoint's Detect buffers when present
    (`checkpoint_anchors`), else the yaml's."""
    spec, sd, seed = load_checkpoint(weights_path, cfg, nc)
    size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
    return Detector(spec, sd, img_size=size, max_batch=max_batch, seed=seed)
