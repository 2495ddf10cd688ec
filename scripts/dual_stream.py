"""experiment: ONE detector forward over 32 frames vs TWO concurrent forwards over 16 frames each on separate HIP streams
(staggered by half a forward), to see whether the memory-bound high-resolution layers of one half hide under the
power-/MFMA-bound layers of the other.   python scripts/dual_stream.py [B] [reps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from yolov7_tracker_amd.detector import arch, model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
frames = torch.randint(0, 256, (B, 1280, 1280, 3), dtype=torch.uint8, device="cuda")


def timed(fn, n):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


one = model.Detector(arch.ARCHS["yolov7-w6"](10), None, img_size=(1280, 1280), max_batch=B, seed=0)
t1 = timed(lambda: one(frames), reps)
print("single stream, %d frames per forward: %.2f ms  (%.0f frames/s detector only)" % (B, t1, B / t1 * 1e3), flush=True)
del one
torch.cuda.empty_cache()
h = B // 2
da = model.Detector(arch.ARCHS["yolov7-w6"](10), None, img_size=(1280, 1280), max_batch=h, seed=0)
db = model.Detector(arch.ARCHS["yolov7-w6"](10), None, img_size=(1280, 1280), max_batch=h, seed=0)
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def both():
    with torch.cuda.stream(sa):
        da(frames[:h])
    with torch.cuda.stream(sb):
        db(frames[h:])


t2 = timed(both, reps)
print("two streams, 2 x %d frames: %.2f ms  (%.0f frames/s detector only)" % (h, t2, B / t2 * 1e3), flush=True)
# staggered: stream b starts half a forward late (one half-batch forward is enqueued on a first)
with torch.cuda.stream(sa):
    da(frames[:h])
t3 = timed(both, reps)
print("two streams, staggered: %.2f ms  (%.0f frames/s detector only)" % (t3, B / t3 * 1e3), flush=True)

# staggered with events: only one stream at a time is inside the memory-bound high-resolution ops [0, k_mid) of its forward
k_mid = next((i for i, op in enumerate(da.plan.ops) if int(op["H"]) <= 160), 0)
n_iter = reps


def interleaved():
    evs = []
    prev_h = None
    for it in range(n_iter):
        for d, st, fr in ((da, sa, frames[:h]), (db, sb, frames[h:])):
            ev = torch.cuda.Event()
            with torch.cuda.stream(st):
                if prev_h is not None:
                    st.wait_event(prev_h)          # the other stream has left its high-resolution part
                d.forward(fr, mid_hook=(k_mid, lambda e=ev, s_=st: e.record(s_)))
            prev_h = ev


interleaved(); torch.cuda.synchronize()
t0 = time.perf_counter(); interleaved(); torch.cuda.synchronize()
t4 = (time.perf_counter() - t0) / n_iter * 1e3
print("two streams, high-res parts serialised by events: %.2f ms per %d frames  (%.0f frames/s detector only)" % (t4, B, B / t4 * 1e3), flush=True)
