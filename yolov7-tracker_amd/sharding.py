"""Multi-GPU: the tracking path shards by SEQUENCE (a tracker is strictly sequential within a sequence, sequences are
independent), one process per GPU, no collective on the data path.  The only exchange is the result gather at the end
(RCCL over xGMI through torch.distributed backend 'nccl'; 'gloo' in the CPU tests), plus the id re-basing that makes
the ids equal the reference's single-process run, whose BaseTrack._count is global across sequences
(/root/reference/tracker/basetrack.py:22,43-46; new tracker per sequence at tracker/track.py:132):

    rank r owns sequences {s : s mod world == r} (sorted order, track.py:108), tracks each with a LOCAL id counter
    starting at 0, records n_ids[s];  all_gather(n_ids)  ->  exclusive prefix sum in sequence order  ->  id += offset;
    gather rows (frame, id, x, y, w, h, cls) to rank 0.   Payload: KBs to a few MB -> latency-bound, one hop per peer.
"""
import torch
import torch.distributed as dist


def owned_sequences(n_seqs, rank, world):
    return [s for s in range(n_seqs) if s % world == rank]


def rebase_and_gather(rows_by_seq, n_ids_by_seq, n_seqs, group=None, device="cpu"):
    """rows_by_seq: {seq index: float64 tensor (n, 8) [frame, id(local, 1-based), x, y, w, h, cls, score]} of THIS rank;
    n_ids_by_seq: {seq index: ids handed out in that sequence}.  Returns on rank 0 the list (per sequence) of rows with
    global ids; None elsewhere."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    counts = torch.zeros(n_seqs, dtype=torch.int64, device=device)
    for s, n in n_ids_by_seq.items():
        counts[s] = int(n)
    if world > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)      # every sequence has exactly one owner
    offsets = torch.cumsum(counts, 0) - counts                           # exclusive prefix, sequence order
    mine = owned_sequences(n_seqs, rank, world)
    sizes = torch.zeros(n_seqs, dtype=torch.int64, device=device)
    rebased = {}
    for s in mine:
        r = rows_by_seq[s].to(device=device, dtype=torch.float64).clone()
        if r.numel():
            r[:, 1] += float(offsets[s].item())
        rebased[s] = r
        sizes[s] = r.shape[0]
    if world == 1:
        return [rebased[s] for s in range(n_seqs)]
    dist.all_reduce(sizes, op=dist.ReduceOp.SUM, group=group)
    # one flat gather: every rank contributes its sequences' rows padded to the per-rank maximum
    per_rank = [int(sum(int(sizes[s].item()) for s in owned_sequences(n_seqs, r, world))) for r in range(world)]
    pad = max(per_rank + [1])
    flat = torch.zeros((pad, 8), dtype=torch.float64, device=device)
    o = 0
    for s in mine:
        n = rebased[s].shape[0]
        flat[o:o + n] = rebased[s]
        o += n
    out = [torch.zeros_like(flat) for _ in range(world)] if rank == 0 else None
    dist.gather(flat, out, dst=0, group=group)
    if rank != 0:
        return None
    res = [None] * n_seqs
    for r in range(world):
        o = 0
        for s in owned_sequences(n_seqs, r, world):
            n = int(sizes[s].item())
            res[s] = out[r][o:o + n].clone()
            o += n
    return res


# ---------------------------------------------------------------------------------------------------------------------
# Single-stream mode (second granularity): ONE sequence, the detector frame-sharded, the tracker on one rank.
# Batch s (consecutive frames) is detected by rank s % world; its (B, max_det, 6) detections + counts travel to rank 0 with one
# point-to-point message (RCCL send/recv over the direct xGMI link; ~7 KB per frame), where the tracker consumes the batches
# in order.  Strong scaling, bounded by the serial tracker (~0.2 ms/frame).
# ---------------------------------------------------------------------------------------------------------------------
def batch_owner(s, world):
    return s % world


class DetectionRelay:
    """one reusable message buffer per rank: rows [0, max_det) = detections, row max_det column 0 = the count"""

    def __init__(self, batch, max_det, device="cpu", group=None):
        self.max_det, self.group = max_det, group
        self.buf = torch.zeros((batch, max_det + 1, 6), dtype=torch.float32, device=device)

    def pack(self, dets, ndets):
        B = dets.shape[0]
        self.buf[:B, :self.max_det] = dets
        self.buf[:B, self.max_det, 0] = ndets.to(torch.float32)
        return self.buf

    def unpack(self):
        return self.buf[:, :self.max_det], self.buf[:, self.max_det, 0].to(torch.int32)

    def send(self, dets, ndets, dst=0):
        """stream-ordered on the current stream with the 'nccl' backend (enqueue behind the NMS that produced `dets`)"""
        dist.send(self.pack(dets, ndets), dst=dst, group=self.group)

    def recv(self, src):
        dist.recv(self.buf, src=src, group=self.group)
        return self.unpack()
