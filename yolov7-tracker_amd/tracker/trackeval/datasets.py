"""MOTChallenge-format 2-D box datasets for the evaluator: `MotChallenge2DBox` and `VisDrone2DBox`.

Restates the loading + pre-processing the reference's vendored TrackEval applies before the metrics run
(/root/reference/tracker/trackeval/datasets/mot_challenge_2d_box.py:17-48 config, :186-288 raw data, :290-433 pre-processing;
visdrone.py differs only in the class table and the distractor classes), for the configuration the reference uses
(`SEQ_INFO` given, plain text files, `track.py:196-227`): text rows `frame,id,x,y,w,h,conf,class,visibility`, similarity = IoU of
xywh boxes, tracker boxes that match a distractor ground-truth box at IoU >= 0.5 are dropped, ground truth is reduced to the
evaluated class with a non-zero `conf` flag, ids are relabelled 0..n-1."""
import os

import numpy as np
from scipy.optimize import linear_sum_assignment

EPS = np.finfo(float).eps


def box_iou_xywh(a, b):
    """IoU matrix of two sets of (x, y, w, h) boxes"""
    a, b = np.asarray(a, float).reshape(-1, 4), np.asarray(b, float).reshape(-1, 4)
    ax2, ay2, bx2, by2 = a[:, 0] + a[:, 2], a[:, 1] + a[:, 3], b[:, 0] + b[:, 2], b[:, 1] + b[:, 3]
    iw = np.clip(np.minimum(ax2[:, None], bx2[None, :]) - np.maximum(a[:, None, 0], b[None, :, 0]), 0, None)
    ih = np.clip(np.minimum(ay2[:, None], by2[None, :]) - np.maximum(a[:, None, 1], b[None, :, 1]), 0, None)
    inter = iw * ih
    union = (a[:, 2] * a[:, 3])[:, None] + (b[:, 2] * b[:, 3])[None, :] - inter
    out = np.zeros_like(inter)
    ok = union > EPS
    out[ok] = inter[ok] / union[ok]
    return out


class _Box2D:
    CLASS_IDS = {}
    VALID_CLASSES = []
    DISTRACTORS = []

    @classmethod
    def get_name(cls):
        return cls.__name__

    @classmethod
    def get_default_dataset_config(cls):
        return {
            'GT_FOLDER': './data/gt/mot_challenge/', 'TRACKERS_FOLDER': './data/trackers/mot_challenge/', 'OUTPUT_FOLDER': None,
            'TRACKERS_TO_EVAL': None, 'CLASSES_TO_EVAL': list(cls.VALID_CLASSES[:1]) if cls is MotChallenge2DBox else list(cls.VALID_CLASSES),
            'BENCHMARK': 'MOT17', 'SPLIT_TO_EVAL': 'train', 'INPUT_AS_ZIP': False, 'PRINT_CONFIG': True, 'DO_PREPROC': True,
            'TRACKER_SUB_FOLDER': 'data', 'OUTPUT_SUB_FOLDER': '', 'TRACKER_DISPLAY_NAMES': None, 'SEQMAP_FOLDER': None, 'SEQMAP_FILE': None,
            'SEQ_INFO': None, 'GT_LOC_FORMAT': '{gt_folder}/{seq}/gt/gt.txt', 'SKIP_SPLIT_FOL': False,
        }

    def __init__(self, config=None):
        cfg = self.get_default_dataset_config()
        for k, v in (config or {}).items():
            if k in cfg:
                cfg[k] = v
        self.config = cfg
        if cfg['INPUT_AS_ZIP']:
            raise NotImplementedError("zipped tracker input")
        if not cfg['SEQ_INFO']:
            raise NotImplementedError("sequence list: pass SEQ_INFO {name: number of frames} (what the reference's yaml files do)")
        self.benchmark = cfg['BENCHMARK']
        split = '' if cfg['SKIP_SPLIT_FOL'] else cfg['BENCHMARK'] + '-' + cfg['SPLIT_TO_EVAL']
        self.gt_fol = os.path.join(cfg['GT_FOLDER'], split)
        self.tracker_fol = os.path.join(cfg['TRACKERS_FOLDER'], split)
        self.output_fol = cfg['OUTPUT_FOLDER'] or self.tracker_fol
        self.do_preproc = cfg['DO_PREPROC']
        self.class_list = [c for c in cfg['CLASSES_TO_EVAL'] if c in self.VALID_CLASSES]
        if not self.class_list:
            raise ValueError("no valid class to evaluate among %r (valid: %r)" % (cfg['CLASSES_TO_EVAL'], self.VALID_CLASSES))
        self.seq_lengths = {k: int(v) for k, v in cfg['SEQ_INFO'].items()}
        self.seq_list = list(self.seq_lengths)
        self.tracker_list = cfg['TRACKERS_TO_EVAL'] or sorted(os.listdir(self.tracker_fol))
        for seq in self.seq_list:
            if not os.path.isfile(self._gt_file(seq)):
                raise FileNotFoundError("GT file not found: " + self._gt_file(seq))
        for trk in self.tracker_list:
            for seq in self.seq_list:
                if not os.path.isfile(self._tracker_file(trk, seq)):
                    raise FileNotFoundError("tracker file not found: " + self._tracker_file(trk, seq))

    def _gt_file(self, seq):
        return self.config['GT_LOC_FORMAT'].format(gt_folder=self.gt_fol, seq=seq)

    def _tracker_file(self, tracker, seq):
        return os.path.join(self.tracker_fol, tracker, self.config['TRACKER_SUB_FOLDER'], seq + '.txt')

    def get_eval_info(self):
        return self.tracker_list, self.seq_list, self.class_list

    def get_display_name(self, tracker):
        return tracker

    def get_output_fol(self, tracker):
        return os.path.join(self.output_fol, tracker, self.config['OUTPUT_SUB_FOLDER'])

    @staticmethod
    def _read(path):
        rows = []
        with open(path) as f:
            for line in f:
                line = line.strip()
                if line:
                    rows.append([float(v) for v in line.replace(' ', ',').split(',') if v != ''])
        width = max((len(r) for r in rows), default=0)
        return np.array([r + [-1.0] * (width - len(r)) for r in rows], float).reshape(len(rows), width)

    def get_raw_seq_data(self, tracker, seq):
        T = self.seq_lengths[seq]
        gt, tr = self._read(self._gt_file(seq)), self._read(self._tracker_file(tracker, seq))
        if len(tr) and (tr[:, 0].min() < 1 or tr[:, 0].max() > T):
            raise ValueError("tracker data of %s has frames outside 1..%d" % (seq, T))
        raw = {k: [None] * T for k in ('gt_ids', 'gt_dets', 'gt_classes', 'gt_zero_marked', 'tracker_ids', 'tracker_dets', 'tracker_classes',
                                       'tracker_confidences', 'similarity_scores')}
        for t in range(T):
            g = gt[gt[:, 0] == t + 1] if len(gt) else np.zeros((0, 9))
            k = tr[tr[:, 0] == t + 1] if len(tr) else np.zeros((0, 7))
            raw['gt_ids'][t] = g[:, 1].astype(int)
            raw['gt_dets'][t] = g[:, 2:6]
            raw['gt_classes'][t] = g[:, 7].astype(int) if g.shape[1] >= 8 else np.ones(len(g), int)
            raw['gt_zero_marked'][t] = g[:, 6].astype(int) if g.shape[1] >= 7 else np.ones(len(g), int)
            raw['tracker_ids'][t] = k[:, 1].astype(int)
            raw['tracker_dets'][t] = k[:, 2:6]
            raw['tracker_classes'][t] = np.ones(len(k), int)
            raw['tracker_confidences'][t] = k[:, 6] if k.shape[1] >= 7 else np.ones(len(k))
            raw['similarity_scores'][t] = box_iou_xywh(raw['gt_dets'][t], raw['tracker_dets'][t])
            for key in ('gt_ids', 'tracker_ids'):
                if len(np.unique(raw[key][t])) != len(raw[key][t]):
                    raise ValueError("%s: duplicate %s in frame %d" % (seq, key, t + 1))
        raw['num_timesteps'], raw['seq'] = T, seq
        return raw

    def get_preprocessed_seq_data(self, raw, cls):
        cls_id = self.CLASS_IDS[cls]
        distractors = [self.CLASS_IDS[c] for c in self.DISTRACTORS] + ([self.CLASS_IDS['non_mot_vehicle']] if self.benchmark == 'MOT20' and
                                                                      'non_mot_vehicle' in self.CLASS_IDS else [])
        preproc = self.do_preproc and self.benchmark != 'MOT15'
        T = raw['num_timesteps']
        data = {k: [None] * T for k in ('gt_ids', 'tracker_ids', 'gt_dets', 'tracker_dets', 'tracker_confidences', 'similarity_scores')}
        for t in range(T):
            sim = raw['similarity_scores'][t]
            drop = np.array([], int)
            if preproc and len(raw['gt_ids'][t]) and len(raw['tracker_ids'][t]):
                bad = np.setdiff1d(np.unique(raw['gt_classes'][t]), list(self.CLASS_IDS.values()))
                if len(bad):
                    raise ValueError("invalid gt classes %s in %s frame %d" % (bad, raw['seq'], t + 1))
                m = sim.copy()
                m[m < 0.5 - EPS] = 0
                rows, cols = linear_sum_assignment(-m)
                ok = m[rows, cols] > EPS
                rows, cols = rows[ok], cols[ok]
                drop = cols[np.isin(raw['gt_classes'][t][rows], distractors)]
            keep_t = np.ones(len(raw['tracker_ids'][t]), bool)
            keep_t[drop] = False
            keep_g = raw['gt_zero_marked'][t] != 0
            if preproc:
                keep_g &= raw['gt_classes'][t] == cls_id
            data['tracker_ids'][t] = raw['tracker_ids'][t][keep_t]
            data['tracker_dets'][t] = raw['tracker_dets'][t][keep_t]
            data['tracker_confidences'][t] = raw['tracker_confidences'][t][keep_t]
            data['gt_ids'][t] = raw['gt_ids'][t][keep_g]
            data['gt_dets'][t] = raw['gt_dets'][t][keep_g]
            data['similarity_scores'][t] = sim[keep_g][:, keep_t]
        for key in ('gt_ids', 'tracker_ids'):   # contiguous ids
            allv = np.concatenate(data[key]) if T else np.zeros(0, int)
            uniq = np.unique(allv)
            for t in range(T):
                data[key][t] = np.searchsorted(uniq, data[key][t]).astype(int)
            data['num_' + key] = len(uniq)
        data['num_gt_dets'] = int(sum(len(v) for v in data['gt_ids']))
        data['num_tracker_dets'] = int(sum(len(v) for v in data['tracker_ids']))
        data['num_timesteps'], data['seq'] = T, raw['seq']
        return data


class MotChallenge2DBox(_Box2D):
    CLASS_IDS = {'pedestrian': 1, 'person_on_vehicle': 2, 'car': 3, 'bicycle': 4, 'motorbike': 5, 'non_mot_vehicle': 6, 'static_person': 7,
                 'distractor': 8, 'occluder': 9, 'occluder_on_ground': 10, 'occluder_full': 11, 'reflection': 12, 'crowd': 13}
    VALID_CLASSES = ['pedestrian']
    DISTRACTORS = ['person_on_vehicle', 'static_person', 'distractor', 'reflection']


class VisDrone2DBox(_Box2D):
    CLASS_IDS = {'ignored': 0, 'pedestrian': 1, 'people': 2, 'bicycle': 3, 'car': 4, 'van': 5, 'truck': 6, 'tricycle': 7, 'awning-tricycle': 8,
                 'bus': 9, 'motor': 10, 'other': 11}
    VALID_CLASSES = ['pedestrian', 'people', 'bicycle', 'car', 'van', 'truck', 'tricycle', 'awning-tricycle', 'bus', 'motor']
    DISTRACTORS = ['ignored', 'other']
