"""GPU, the N > 1 code path of bench.py on a ONE-GPU box: two ranks (torch.distributed.run) sharing the device, the gloo backend for the
collective (RCCL refuses two ranks on one device).  Sequence-sharded mode: each rank tracks its own synthetic sequence with the DEVICE
ByteTrack; the gathered rows / id bases must equal a single-process run over the same two sequences with the reference's global id
counter (basetrack.py:22,43-46).  Frame-sharded single-stream mode: runs and produces tracks."""
import json
import os
import socket
import subprocess
import sys
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(mode, steps, warmup, batch, img, n_obj):
    env = dict(os.environ, Y7T_BENCH_SHARE_GPU="1", Y7T_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", str(steps), "--warmup", str(warmup), "--batch", str(batch),
           "--img", str(img), "--n_obj", str(n_obj), "--mode", mode, "--no_cpu_baseline", "--no_latency_mode"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    return json.loads(line)


def test_two_ranks_sequence_sharded_equals_single_process():
    import torch
    from yolov7_tracker_amd import synth
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack
    from yolov7_tracker_amd.tracker.bytetrack import ByteTrack
    steps, warmup, batch, img, n_obj = 2, 1, 4, 640, 40
    d = _run("sequences", steps, warmup, batch, img, n_obj)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["collective_backend"] == "gloo" and d["value"] > 0
    g = d["config"]["result_gather"]
    # single process, one global counter, sequence 0 then sequence 1 (the reference's serial loop over sequences, track.py:123)
    n_frames = (steps + warmup) * batch
    BaseTrack._count = 0
    rows, bases = [], []
    for seq in range(2):
        bases.append(BaseTrack._count)
        t = ByteTrack(types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format="default", img_size=1280, iou_thresh=0.5, max_tracks=512, max_dets=512))
        n = 0
        for det in synth.make_detections(n_frames, n_obj, img, seq_idx=seq, bounce=True):
            n += len(t.update(det, None))
        rows.append(n)
    assert g["rows_per_rank"] == rows, (g, rows)
    assert g["id_base_per_rank"] == bases, (g, bases)
    # the gather bench.py runs IS sharding.rebase_and_gather (the code the gloo test proves): 28-byte rows, one padded block per rank
    assert "sharding.rebase_and_gather" in g["via"] and g["bytes_per_row"] == 28 and g["payload_bytes_per_rank"] == 28 * max(rows), g
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(d, open(os.path.join(ROOT, "gpurun_out", "two_ranks_one_gpu_sequences.json"), "w"), indent=1)


def test_two_ranks_frame_sharded_single_stream_runs():
    d = _run("frames", 2, 1, 4, 640, 40)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["value"] > 0 and d["config"]["tracks_alive_last_frame"] > 10
    json.dump(d, open(os.path.join(ROOT, "gpurun_out", "two_ranks_one_gpu_frames.json"), "w"), indent=1)


def _bench_plain(gpus, extra_env=None):
    """`python bench.py --gpus N ...` with NO launcher: bench.py starts its own ranks"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "Y7T_BENCH_SHARE_GPU", "Y7T_BENCH_BACKEND")}
    env.update(extra_env or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "2", "--warmup", "1", "--batch", "4", "--img", "640", "--n_obj", "40",
           "--no_cpu_baseline", "--no_latency_mode"]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)


def test_gpus_n_with_too_few_devices_is_an_error_not_an_n_gpus_1_line():
    import torch
    n = torch.cuda.device_count()
    r = _bench_plain(n + 1)
    assert r.returncode != 0 and "visible on this node" in (r.stdout + r.stderr)
    assert not any(ln.startswith("{") for ln in r.stdout.splitlines())


def test_self_launched_two_ranks_share_the_gpu_through_the_same_code_path():
    """the self-launch itself on the one-GPU box (gloo, both ranks on the device): n_gpus = 2 in the line, ids re-based by the exclusive prefix"""
    r = _bench_plain(2, {"Y7T_BENCH_SHARE_GPU": "1", "Y7T_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    g = d["config"]["result_gather"]
    assert d["n_gpus"] == 2 and g["id_base_per_rank"][0] == 0 and g["id_base_per_rank"][1] > 0
    assert g["backend_reported"] == "gloo" and g["ranks_seen"] == 2
    _assert_gather_equals_single_process(g)      # (the same check the RCCL test makes where two devices are visible)


def _assert_gather_equals_single_process(g):
    """the two sequences of `_bench_plain(2)` tracked by ONE process with the reference's global BaseTrack._count: rows, id bases and the digest of the global ids"""
    from yolov7_tracker_amd import synth
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack
    from yolov7_tracker_amd.tracker.bytetrack import ByteTrack
    BaseTrack._count = 0
    rows, bases, digest, distinct = [], [], [], []
    for seq in range(2):
        bases.append(BaseTrack._count)
        t = ByteTrack(types.SimpleNamespace(conf_thresh=0.2, track_buffer=30, kalman_format="default", img_size=1280, iou_thresh=0.5, max_tracks=512, max_dets=512))
        ids = [[tr.track_id for tr in t.update(det, None)] for det in synth.make_detections(3 * 4, 40, 640, seq_idx=seq, bounce=True)]
        rows.append(sum(len(f) for f in ids))
        digest.append(sum((fi + 1) * i for fi, f in enumerate(ids) for i in f))
        distinct.append(len({i for f in ids for i in f}))
    assert g["rows_per_rank"] == rows and g["id_base_per_rank"] == bases and g["payload_bytes_per_rank"] == 28 * max(rows), (g, rows, bases)
    assert g["frame_x_id_digest_per_rank"] == digest and g["distinct_ids_per_rank"] == distinct, "the gathered ids are not those of the single-process run"


def test_rccl_two_ranks_on_two_devices():
    """needs >= 2 MI355X in the lease (the driver's 8-GPU tier; skipped on the 1-GPU boxes): bench.py --gpus 2 with no launcher -> two ranks, the
    `nccl` backend (= RCCL), result gather with the exclusive-prefix id bases, equal to a single-process run over the two sequences."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one device visible: RCCL wants a GPU per rank")
    r = _bench_plain(2)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["collective_backend"] == "nccl"
    g = d["config"]["result_gather"]
    from yolov7_tracker_amd import synth
    from yolov7_tracker_amd.tracker.basetrack import BaseTrack
    from yolov7_tracker_amd.tracker.bytetrack import ByteTrack
    assert g["backend_reported"] == "nccl" and g["ranks_seen"] == 2
    _assert_gather_equals_single_process(g)
