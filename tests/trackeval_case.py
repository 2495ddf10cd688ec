"""shared scenario for the evaluation-harness tests and tests/golden/make_golden.py: synthetic ground truth + the result files
a tracker would write (numpy ByteTrack oracle on the synthetic detections, plus hand-made disturbances: an id swap, a dropped
stretch, tracker boxes on a distractor object), on disk in the reference's folder layout."""
import os

import numpy as np

SEQS = {"synth-A": (60, 25, 640, 3), "synth-B": (45, 40, 640, 4)}      # name -> (frames, objects, size, seq_idx)


def build(root):
    """writes <root>/gt/<seq>.txt and <root>/trackers/bytetrack_oracle/<seq>.txt; -> dataset config dict"""
    from oracle import tracker_np
    from yolov7_tracker_amd import synth
    os.makedirs(os.path.join(root, "gt"), exist_ok=True)
    os.makedirs(os.path.join(root, "trackers", "bytetrack_oracle"), exist_ok=True)
    seq_info = {}
    for name, (nf, nobj, size, idx) in SEQS.items():
        gt = synth.make_ground_truth(nf, nobj, size, idx)
        dets = synth.make_detections(nf, nobj, size, idx)
        out = tracker_np.run("bytetrack", dets)
        with open(os.path.join(root, "gt", name + ".txt"), "w") as f:
            for t, rows in enumerate(gt):
                for r in rows:
                    gid = int(r[0])
                    cls = 8 if gid == 2 else (7 if gid == 5 else 1)      # objects 2 / 5 are a distractor / a static person
                    conf = 0 if gid == 7 else 1                           # object 7 is marked "do not evaluate"
                    f.write("%d,%d,%.2f,%.2f,%.2f,%.2f,%d,%d,1.0\n" % (t + 1, gid, r[1], r[2], r[3], r[4], conf, cls))
        with open(os.path.join(root, "trackers", "bytetrack_oracle", name + ".txt"), "w") as f:
            for t, rows in enumerate(out):
                for tid, tlwh, cls, score in rows:
                    if 20 <= t < 26 and tid % 5 == 0:
                        continue                                          # a dropped stretch -> FN, fragmentation
                    if t >= 30 and tid in (3, 4):
                        tid = 7 - tid                                     # ids 3 and 4 swap from frame 31 on -> ID switches
                    f.write("%d,%d,%.2f,%.2f,%.2f,%.2f,1.0,-1,-1,-1\n" % (t + 1, tid, tlwh[0], tlwh[1], tlwh[2], tlwh[3]))
        seq_info[name] = nf
    return {"GT_FOLDER": os.path.join(root, "gt"), "TRACKERS_FOLDER": os.path.join(root, "trackers"), "SKIP_SPLIT_FOL": True,
            "TRACKER_SUB_FOLDER": "", "SEQ_INFO": seq_info, "GT_LOC_FORMAT": "{gt_folder}/{seq}.txt", "PRINT_CONFIG": False}


def flatten(res):
    """{metric: {field: scalar | array}} -> {"metric.field": list | number}"""
    out = {}
    for m, fields in res.items():
        for k, v in fields.items():
            out["%s.%s" % (m, k)] = np.asarray(v, float).tolist()
    return out
