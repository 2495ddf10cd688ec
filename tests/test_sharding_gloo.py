"""CPU, world_size 2 over gloo: sequence-sharded tracking + id re-basing + result gather reproduce the single-process
run with the reference's global id counter.  (The per-rank tracker here is the CPU host-sim of the device programs --
test infrastructure; on the GPU box the same sharding code runs over RCCL with the device tracker, see bench.py.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N_SEQ, N_FRAMES, N_OBJ = 5, 25, 30


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _track(seq_idx, ids):
    from tests import _hostsim as hs
    from yolov7_tracker_amd import synth
    dets = synth.make_detections(N_FRAMES, N_OBJ, seq_idx=seq_idx)
    trk = hs.HostSimTracker("bytetrack", ids=ids, cap_t=256, cap_d=256)
    rows = []
    for f, d in enumerate(dets):
        for (tid, tlwh, cls, score) in trk.update(d):
            rows.append([f + 1, tid, tlwh[0], tlwh[1], tlwh[2], tlwh[3], cls, score])
    return torch.tensor(rows, dtype=torch.float64).reshape(-1, 8)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from yolov7_tracker_amd import sharding
    rows, nids = {}, {}
    for s in sharding.owned_sequences(N_SEQ, rank, world):
        ids = np.zeros(1, np.int32)          # local counter per sequence
        rows[s] = _track(s, ids)
        nids[s] = int(ids[0])
    res = sharding.rebase_and_gather(rows, nids, N_SEQ)
    if rank == 0:
        q.put(([r.numpy() for r in res], dict(sharding.last_gather_stats)))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_single_process_with_global_ids():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got, stats = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process, one GLOBAL counter across sequences in sorted order (reference semantics)
    ids = np.zeros(1, np.int32)
    n_rows = []
    for s in range(N_SEQ):
        want = _track(s, ids).numpy()
        n_rows.append(len(want))
        assert got[s].shape == (len(want), 7)                            # frame, id, x, y, w, h, cls
        np.testing.assert_array_equal(got[s][:, :2], want[:, :2])       # frame, id: bit-exact
        # boxes: rounded to the result file's two decimals in float64 BEFORE they are narrowed to the float32 they travel as ...
        np.testing.assert_array_equal(got[s][:, 2:6], (np.round(want[:, 2:6] * 100.0) / 100.0).astype(np.float32).astype(np.float64))
        np.testing.assert_array_equal(got[s][:, 6], want[:, 6])
        assert np.abs(got[s][:, 2:6] - want[:, 2:6]).max() <= 5e-3 + 1e-4    # ... so they are the float64 boxes to the half unit of that last digit, and `%.2f` of what arrives
        fmt = lambda a: ["%.2f,%.2f,%.2f,%.2f" % tuple(r) for r in a]       # prints the digits a single-process run prints from its float64 tlwh (track.py:266)
        a, b = fmt(got[s][:, 2:6]), fmt(want[:, 2:6])
        assert sum(x != y for x, y in zip(a, b)) <= len(a) // 200, sum(x != y for x, y in zip(a, b))      # (a float64 value within 1e-12 of a rounding boundary may print either way)
    # the wire format: 28 bytes per row, one gather padded to the fuller rank
    per_rank = [sum(n_rows[s] for s in range(N_SEQ) if s % 2 == r) for r in range(2)]
    assert stats["bytes_per_row"] == 28 and stats["rows_per_rank"] == per_rank and stats["payload_bytes_per_rank"] == 28 * max(per_rank)


def test_single_rank_path():
    from yolov7_tracker_amd import sharding
    rows = {0: torch.tensor([[1, 1, 0, 0, 1, 1, 0, .5]], dtype=torch.float64), 1: torch.tensor([[1, 1, 0, 0, 1, 1, 0, .5], [1, 2, 0, 0, 1, 1, 0, .5]], dtype=torch.float64)}
    res = sharding.rebase_and_gather(rows, {0: 3, 1: 2}, 2)
    assert res[0][0, 1] == 1 and res[1][0, 1] == 4 and res[1][1, 1] == 5 and res[1].shape == (2, 7) and sharding.last_gather_stats["payload_bytes_per_rank"] == 0
    w = sharding.pack_rows(torch.tensor([[7, 3, 10.25, -2.5, 33.125, 1e3, 9, .5]], dtype=torch.float64))
    assert w.dtype == torch.int32 and w.shape == (1, 7) and w.numel() * 4 == 28
    got = sharding.unpack_rows(w).tolist()
    assert got[0][:2] == [7, 3] and got[0][6] == 9 and got[0][2] == 10.25 and got[0][3] == -2.5 and got[0][5] == 1e3
    assert abs(got[0][4] - 33.12) < 1e-5 or abs(got[0][4] - 33.13) < 1e-5          # 33.125 travels as the two-decimal value it would be written as (track.py:266)


# ---- single-stream mode: frame-sharded detection, tracker on rank 0 ----
FS_BATCH, FS_STEPS, FS_MAXDET = 5, 6, 64


def _batch_dets(s):
    """the 'detector output' of batch s: the synthetic detections of its frames, padded to (B, max_det, 6) + counts"""
    from yolov7_tracker_amd import synth
    dets = synth.make_detections(FS_BATCH * FS_STEPS, N_OBJ, seq_idx=7)[s * FS_BATCH:(s + 1) * FS_BATCH]
    out = torch.zeros((FS_BATCH, FS_MAXDET, 6), dtype=torch.float32)
    nd = torch.zeros(FS_BATCH, dtype=torch.int32)
    for i, d in enumerate(dets):
        out[i, :len(d)] = torch.from_numpy(d)
        nd[i] = len(d)
    return out, nd


def _fs_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import _hostsim as hs
    from yolov7_tracker_amd import sharding
    relay = sharding.DetectionRelay(FS_BATCH, FS_MAXDET)
    if rank == 0:
        trk = hs.HostSimTracker("bytetrack", ids=np.zeros(1, np.int32), cap_t=256, cap_d=256)
        rows = []
        for s in range(FS_STEPS):
            owner = sharding.batch_owner(s, world)
            dets, nd = _batch_dets(s) if owner == 0 else relay.recv(owner)
            for i in range(FS_BATCH):
                for (tid, tlwh, cls, score) in trk.update(dets[i, :int(nd[i])].numpy().copy()):
                    rows.append([s * FS_BATCH + i + 1, tid, tlwh[0], tlwh[1], tlwh[2], tlwh[3]])
        q.put(np.asarray(rows))
    else:
        for s in range(FS_STEPS):
            if sharding.batch_owner(s, world) == rank:
                relay.send(*_batch_dets(s))
    dist.barrier()
    dist.destroy_process_group()


def test_frame_sharded_detection_feeds_one_tracker_in_order():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_fs_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from tests import _hostsim as hs
    trk = hs.HostSimTracker("bytetrack", ids=np.zeros(1, np.int32), cap_t=256, cap_d=256)
    want = []
    for s in range(FS_STEPS):
        dets, nd = _batch_dets(s)
        for i in range(FS_BATCH):
            for (tid, tlwh, cls, score) in trk.update(dets[i, :int(nd[i])].numpy().copy()):
                want.append([s * FS_BATCH + i + 1, tid, tlwh[0], tlwh[1], tlwh[2], tlwh[3]])
    np.testing.assert_array_equal(got, np.asarray(want))
