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
