"""Multi-GPU: the tracking path shards by SEQUENCE (a tracker is strictly sequential within a sequence, sequences are
independent), one process per GPU, no collective on the data path.  The only exchange is the result gather at the end
(RCCL over xGMI through torch.distributed backend 'nccl'; 'gloo' in the CPU tests), plus the id re-basing that makes
the ids equal the reference's single-process run, whose BaseTrack._count is global across sequences
(/root/reference/tracker/basetrack.py:22,43-46; new tracker per sequence at tracker/track.py:132):

    rank r owns sequences {s : s mod world == r} (sorted order, track.py:108), tracks each with a LOCAL id counter
    starting at 0, records n_ids[s];  all_reduce(n_ids)  ->  exclusive prefix sum in sequence order  ->  id += offset;
    gather rows (frame, id, x, y, w, h, cls) to rank 0.   Payload: KBs to a few MB -> latency-bound, one hop per peer.
"""
import torch
import torch.distributed as dist


def owned_sequences(n_seqs, rank, world):
    return [s for s in range(n_seqs) if s % world == rank]


ROW_WORDS = 7            # a result row on the wire: frame:int32, id:int32, x, y, w, h:float32, cls:int32 = 28 bytes -- what the reference writes per row
                         # (`frame,id,x,y,w,h,1.0,-1,-1,-1` with %.2f, tracker/track.py:257-270) and nothing else
last_gather_stats = {}   # filled by rebase_and_gather: rows / payload bytes of the last call (tests, bench line)


def pack_rows(rows):
    """(n, >= 7) float rows [frame, id, x, y, w, h, cls, ...] -> (n, 7) int32 words (boxes as float32 bit patterns).  The boxes are rounded to the result file's two
    decimals (`%.2f`, tracker/track.py:257-270) in the precision they arrive in BEFORE they are narrowed to float32: a float32 holds a two-decimal pixel coordinate to
    6e-5, so `%.2f` of what arrives prints the digits a single-process run prints from its float64 tlwh -- the gathered files do not depend on the transit precision
    (ADVICE r4)."""
    out = torch.empty((rows.shape[0], ROW_WORDS), dtype=torch.int32, device=rows.device)
    if rows.shape[0]:
        out[:, 0:2] = rows[:, 0:2].to(torch.int32)
        out[:, 2:6] = (torch.round(rows[:, 2:6].to(torch.float64) * 100.0) / 100.0).to(torch.float32).contiguous().view(torch.int32)
        out[:, 6] = rows[:, 6].to(torch.int32)
    return out


def unpack_rows(words):
    """(n, 7) int32 words -> (n, 7) float64 rows [frame, id, x, y, w, h, cls]"""
    out = torch.empty((words.shape[0], ROW_WORDS), dtype=torch.float64, device=words.device)
    if words.shape[0]:
        out[:, 0:2] = words[:, 0:2].to(torch.float64)
        out[:, 2:6] = words[:, 2:6].contiguous().view(torch.float32).to(torch.float64)
        out[:, 6] = words[:, 6].to(torch.float64)
    return out


def rebase_and_gather(rows_by_seq, n_ids_by_seq, n_seqs, group=None, device="cpu"):
    """rows_by_seq: {seq index: float tensor (n, >= 7) [frame, id(local, 1-based), x, y, w, h, cls, ...]} of THIS rank;
    n_ids_by_seq: {seq index: ids handed out in that sequence}.  Returns on rank 0 the list (per sequence) of float64 rows
    (n, 7) [frame, id(global), x, y, w, h, cls] -- SEVEN columns: the score is not part of a result row (track.py:257-270) and does not travel; boxes rounded to the
    file format's two decimals (pack_rows), also when world == 1, so every world size writes the same bytes; None elsewhere.
    Two small collectives (id counts + row counts, one all_reduce each) and ONE gather of 28-byte rows."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    mine = owned_sequences(n_seqs, rank, world)
    meta = torch.zeros((2, n_seqs), dtype=torch.int64, device=device)      # row 0: ids handed out, row 1: result rows -- every sequence has exactly one owner
    for s, n in n_ids_by_seq.items():
        meta[0, s] = int(n)
    for s in mine:
        meta[1, s] = rows_by_seq[s].shape[0]
    if world > 1:
        dist.all_reduce(meta, op=dist.ReduceOp.SUM, group=group)
    counts, sizes = meta[0].cpu(), meta[1].cpu()
    offsets = torch.cumsum(counts, 0) - counts                              # exclusive prefix, sequence order: the reference's global BaseTrack._count
    packed = {}
    for s in mine:
        w = pack_rows(rows_by_seq[s].to(device))
        if w.shape[0]:
            w[:, 1] += int(offsets[s].item())
        packed[s] = w
    per_rank = [int(sum(int(sizes[s].item()) for s in owned_sequences(n_seqs, r, world))) for r in range(world)]
    last_gather_stats.clear()
    last_gather_stats.update({"rows_per_rank": per_rank, "bytes_per_row": 4 * ROW_WORDS, "world": world, "id_offset_per_seq": [int(v) for v in offsets.tolist()],
                              "payload_bytes_per_rank": 4 * ROW_WORDS * max(per_rank + [1]) if world > 1 else 0})
    if world == 1:
        return [unpack_rows(packed[s]) for s in range(n_seqs)]
    # one flat gather: every rank contributes its sequences' rows padded to the per-rank maximum
    pad = max(per_rank + [1])
    flat = torch.zeros((pad, ROW_WORDS), dtype=torch.int32, device=device)
    o = 0
    for s in mine:
        n = packed[s].shape[0]
        flat[o:o + n] = packed[s]
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
            res[s] = unpack_rows(out[r][o:o + n])
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
