"""Step time of frames that leave the fast path of the association (profiles/r06_large_components.txt):
  * a connected component of the candidate graph with more than 64 rows or columns (tests/util.lattice_scene: ONE component of 160 / 320 rows; crowds of 250 / 400
    objects on a 640 / 480 px frame) -- solved by a wave with its state in the work arrays (y7t_assoc_sparse_try step 4a);
  * a TIE (integer boxes on a lattice: equal IoUs) -- the whole problem goes to lapjv.cpp run literally (y7t_lap_solve_literal), at 160 / 320 / 500 rows + columns.
Kernel time per frame with HIP events around the per-frame launches (detections resident), ids checked against the oracle where the scene has no tie.
    timeout 600 python scripts/time_large_components.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import util
from oracle import tracker_np
from test_tracker_gpu import make_opts
from yolov7_tracker_amd import synth, _lib
from yolov7_tracker_amd.tracker.basetrack import BaseTrack
from yolov7_tracker_amd.tracker.bytetrack import ByteTrack


def literal_calls():
    L = _lib.load()
    return int(L.y7t_lap_literal_calls()) if hasattr(L, "y7t_lap_literal_calls") else -1


def run(name, dets, check=True, cap=2048):
    ddev = [torch.from_numpy(np.ascontiguousarray(d)).cuda() for d in dets]
    BaseTrack._count = 0
    t = ByteTrack(make_opts(max_tracks=cap, max_dets=1024), frame_rate=30)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in dets]
    lit0 = literal_calls()
    torch.cuda.synchronize()
    for d, (e0, e1) in zip(ddev, ev):
        e0.record(); t._launch(d); e1.record()
    torch.cuda.synchronize()
    us = [e0.elapsed_time(e1) * 1e3 for e0, e1 in ev]
    lit = literal_calls() - lit0
    same = None
    if check:
        want = tracker_np.run("bytetrack", dets)
        BaseTrack._count = 0
        t2 = ByteTrack(make_opts(max_tracks=cap, max_dets=1024), frame_rate=30)
        same = all([tr.track_id for tr in t2.update(d, None)] == [r[0] for r in w] for d, w in zip(dets, want))
    print("%-58s frames %2d  dets/frame %3d  step us: first %6.0f  median %6.0f  max %6.0f  literal re-solves %d  ids == oracle: %s"
          % (name, len(dets), int(np.median([len(d) for d in dets])), us[0], float(np.median(us[1:])), max(us[1:]), lit, same), flush=True)


def integer_lattice(nx, ny, n_frames=6):
    """the lattice with INTEGER boxes of one size: IoUs tie, every association goes to the literal solver"""
    out = []
    for d in util.lattice_scene(n_frames=n_frames, nx=nx, ny=ny):
        d = d.copy()
        d[:, :2] = np.round(d[:, :2]); d[:, 2:4] = d[:, :2] + 100.0
        out.append(d)
    return out


if __name__ == "__main__":
    print("device: %s" % torch.cuda.get_device_name(0))
    run("reference point: 80 objects on 1280 px (no fallback)", synth.make_detections(12, 80, 1280, seq_idx=0))
    run("reference point: 500 objects on 1280 px (components <= 20 rows)", synth.make_detections(12, 500, 1280, seq_idx=0))
    run("lattice 40 x 4: ONE component of 160 rows", util.lattice_scene())
    run("lattice 40 x 4 + 60 extra columns", util.lattice_scene(extra_cols=60))
    run("lattice 80 x 4: ONE component of 320 rows", util.lattice_scene(nx=80))
    run("lattice 125 x 4: ONE component of 500 rows", util.lattice_scene(nx=125))
    for n_obj, size in ((250, 640), (400, 640), (400, 480)):
        run("crowd: %d objects on %d px (10 %% misses, 10 %% clutter)" % (n_obj, size), synth.make_detections(14, n_obj, size, seq_idx=300 + n_obj, miss=0.1, fp=0.1))
    for nx, ny in ((20, 4), (40, 4), (62, 4)):
        run("TIES: integer lattice %d x %d = %d rows + %d columns" % (nx, ny, nx * ny, nx * ny), integer_lattice(nx, ny), check=True)
