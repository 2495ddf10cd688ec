"""`tracker/track.py` of the reference (/root/reference/tracker/track.py:53-386) with the same flags, config-file
format and result-file format, driving the MI355X hot path: detector forward + decode/NMS (liby7t.so) and the
device-resident SORT / ByteTrack trackers.

    python tracker/track.py --dataset visdrone --tracker bytetrack --model_path weights/best.pt
    python tracker/track.py --dataset synthetic --tracker bytetrack --model_path random:yolov7-w6 --synthetic_dets

Extra flags (defaults reproduce the reference's behaviour): --model_cfg (yaml / arch name for state-dict checkpoints),
--nc, --synthetic_dets, --synthetic_frames/--synthetic_objs/--synthetic_seqs, --results_root, --device_preprocess, --batch.
"""
import argparse
import os
from time import gmtime, strftime

import numpy as np
import torch
import yaml

from . import tracker_dataloader
from .basetrack import BaseTracker
from .bytetrack import ByteTrack
from .botsort import BoTSORT
from .deepsort import DeepSORT
from .timer import Timer
from ..detector import attempt_load, check_img_size, non_max_suppression, scale_coords

TRACKER_DICT = {'sort': BaseTracker, 'bytetrack': ByteTrack, 'botsort': BoTSORT, 'deepsort': DeepSORT}   # track.py:56-65; the other trackers are out of scope

timer = Timer()
seq_fps = []


def post_process_v7(out, img_size, ori_img_size, conf_thres=0.01, all_images=False):
    """track.py:234-244 (the reference keeps image 0 of the batch; all_images=True returns the list for every image)"""
    res = non_max_suppression(out, conf_thres=conf_thres)
    for o in res:
        o[:, :4] = scale_coords(img_size, o[:, :4], ori_img_size, ratio_pad=None).round()
    return res if all_images else res[0]


def save_results(results_root, folder_name, seq_name, results, data_type='mot17'):
    """track.py:247-273: `frame,id,x,y,w,h,1.0,-1,-1,-1` with %.2f"""
    assert len(results)
    os.makedirs(os.path.join(results_root, folder_name), exist_ok=True)
    with open(os.path.join(results_root, folder_name, seq_name + '.txt'), 'w') as f:
        for frame_id, target_ids, tlwhs, clses in results:
            for id, tlwh, cls in zip(target_ids, tlwhs, clses):
                if data_type == 'default':
                    f.write(f'{frame_id},{id},{tlwh[0]:.2f},{tlwh[1]:.2f},{tlwh[2]:.2f},{tlwh[3]:.2f},{int(cls)}\n')
                else:
                    f.write(f'{frame_id},{id},{tlwh[0]:.2f},{tlwh[1]:.2f},{tlwh[2]:.2f},{tlwh[3]:.2f},1.0,-1,-1,-1\n')
    return folder_name


def sequence_list(opts, cfgs):
    """track.py:93-109: the sequences to track -> (sorted names, folder that holds them or None).  'origin': the sub-folders of the dataset's sequence folder;
    'yolo': the parent-folder names of the image paths listed in ./<dataset>/test.txt.  IGNORE_SEQS / CERTAIN_SEQS of the dataset yaml apply to both."""
    DATASET_ROOT, CERTAIN_SEQS, IGNORE_SEQS = cfgs['DATASET_ROOT'], cfgs['CERTAIN_SEQS'], cfgs['IGNORE_SEQS']
    if opts.data_format == 'yolo':
        DATA_ROOT = None
        seqs = []
        with open(os.path.join(getattr(opts, 'yolo_root', './'), opts.dataset, 'test.txt'), 'r') as f:
            for line in f.readlines():
                elems = line.split('/')      # the sequence name is elems[-2]
                if len(elems) >= 2 and elems[-2] not in seqs:
                    seqs.append(elems[-2])
    elif opts.data_format == 'origin':
        DATA_ROOT = os.path.join(DATASET_ROOT, cfgs.get('SEQ_SUBDIR', 'VisDrone2019-MOT-test-dev/sequences'))
        seqs = os.listdir(DATA_ROOT)
    else:
        raise NotImplementedError
    seqs = sorted(seqs)
    seqs = [s for s in seqs if s not in IGNORE_SEQS]
    if None not in CERTAIN_SEQS:
        seqs = CERTAIN_SEQS
    return seqs, DATA_ROOT


def main(opts, cfgs):
    DATASET_ROOT, CERTAIN_SEQS, IGNORE_SEQS = cfgs['DATASET_ROOT'], cfgs['CERTAIN_SEQS'], cfgs['IGNORE_SEQS']
    if opts.tracker not in TRACKER_DICT:
        raise NotImplementedError("tracker %r: only %s run on the device path" % (opts.tracker, sorted(TRACKER_DICT)))
    if opts.tracker == 'botsort':
        opts.kalman_format = 'botsort'      # track.py:68-69
    img_size = opts.img_size[0] if isinstance(opts.img_size, (list, tuple)) else opts.img_size
    model = attempt_load(opts.model_path, cfg=opts.model_cfg, nc=opts.nc, img_size=img_size, max_batch=max(1, opts.batch))
    stride = int(model.stride.max())
    opts.img_size = check_img_size(img_size, s=stride)
    synthetic = opts.dataset == 'synthetic'
    if synthetic:
        seqs = ['synthetic-%03d' % i for i in range(opts.synthetic_seqs)]
    else:
        seqs, DATA_ROOT = sequence_list(opts, cfgs)
    print(f'Seqs will be evalueated, total{len(seqs)}:')
    print(seqs)
    folder_name = strftime("%Y-%d-%m %H:%M:%S", gmtime())[5:-3].replace('-', '_').replace(' ', '_').replace(':', '_')
    folder_name = opts.tracker + '_' + folder_name
    for si, seq in enumerate(seqs):
        print(f'--------------tracking seq {seq}--------------')
        if synthetic:
            loader = tracker_dataloader.SyntheticLoader(opts.synthetic_frames, opts.synthetic_objs, opts.img_size, si,
                                                        device_preprocess=opts.device_preprocess)
        else:
            # track.py:126: the sequence folder ('origin') or the path file every sequence is filtered out of ('yolo')
            path = os.path.join(DATA_ROOT, seq) if opts.data_format == 'origin' else os.path.join(getattr(opts, 'yolo_root', './'), opts.dataset, 'test.txt')
            loader = tracker_dataloader.TrackerLoader(path, opts.img_size, opts.data_format, seq,
                                                      pre_process_method='v7', model_stride=stride,
                                                      device_preprocess=opts.device_preprocess,
                                                      yolo_data_root=cfgs.get('YOLO_DATA_ROOT', DATASET_ROOT))
        data_loader = torch.utils.data.DataLoader(loader, batch_size=max(1, opts.batch))
        tracker = TRACKER_DICT[opts.tracker](opts, frame_rate=30, gamma=opts.gamma)
        results, frame_id, i = [], 0, -1
        for imgs, imgs0 in data_loader:
            # --batch N (extension, default 1 = the reference's frame-at-a-time loop): the detector + decode/NMS run once over N
            # consecutive frames (that is where the throughput of bench.py comes from), the tracker then takes them in order
            nb = imgs0.shape[0]
            timer.tic()
            detect = [not (i + 1 + k) % opts.detect_per_frame for k in range(nb)]
            outs = None
            if any(detect):
                # conf_thres 0.01 (track.py:239) is known here, so the Detect convs decode + filter in their epilogue (fused forward)
                if opts.device_preprocess:      # raw uint8 frames -> letterbox + layout on the GPU
                    head, lb_size = model.forward_frames(imgs0, img_size=opts.img_size, fuse_decode=0.01)
                else:
                    head, lb_size = model.forward(imgs.cuda(), fuse_decode=0.01), imgs.shape[2:]
                outs = post_process_v7(head, img_size=lb_size, ori_img_size=imgs0.shape[1:], all_images=True)
            for k in range(nb):
                if k:
                    timer.tic()
                i += 1
                img0 = imgs0[k]
                if detect[k]:
                    out = outs[k]
                    if opts.synthetic_dets and synthetic:
                        out = torch.from_numpy(loader.dets[i])      # the scene's detections stand in for a trained detector
                    current_tracks = tracker.update(out, img0)
                else:
                    current_tracks = tracker.update_without_detection(None, img0)
                cur_tlwh, cur_id, cur_cls = [], [], []
                for trk in current_tracks:
                    bbox = trk.tlwh
                    if bbox[2] * bbox[3] > opts.min_area:
                        cur_tlwh.append(bbox)
                        cur_id.append(trk.track_id)
                        cur_cls.append(trk.cls)
                results.append((frame_id + 1, cur_id, cur_tlwh, cur_cls))
                timer.toc()
                frame_id += 1
        seq_fps.append(i / timer.total_time)   # track.py:181 (sic: last index, not the frame count)
        timer.clear()
        save_results(opts.results_root, folder_name, seq, results)
    print(f'average fps: {np.mean(seq_fps)}')
    if opts.track_eval and (synthetic or cfgs.get('TRACK_EVAL')):
        evaluate_results(opts, cfgs, seqs, folder_name, synthetic)
    return os.path.join(opts.results_root, folder_name)


def evaluate_results(opts, cfgs, seqs, folder_name, synthetic):
    """track.py:196-227: HOTA / CLEAR / Identity of the result files just written, through the TrackEval-style harness
    (tracker/trackeval).  Real datasets take their `TRACK_EVAL` block from the dataset yaml like the reference; the synthetic
    dataset writes its ground truth (synth.make_ground_truth) next to the results first."""
    from . import trackeval
    eval_config = trackeval.Evaluator.get_default_eval_config()
    if synthetic:
        from .. import synth
        gt_folder = os.path.join(opts.results_root, folder_name + '_gt')
        os.makedirs(gt_folder, exist_ok=True)
        for si, seq in enumerate(seqs):
            synth.write_mot_gt(os.path.join(gt_folder, seq + '.txt'),
                               synth.make_ground_truth(opts.synthetic_frames, opts.synthetic_objs, opts.img_size, si))
        yaml_cfg = {'GT_FOLDER': gt_folder, 'TRACKERS_FOLDER': opts.results_root, 'TRACKERS_TO_EVAL': [folder_name], 'SKIP_SPLIT_FOL': True,
                    'TRACKER_SUB_FOLDER': '', 'SEQ_INFO': {seq: opts.synthetic_frames for seq in seqs}, 'GT_LOC_FORMAT': '{gt_folder}/{seq}.txt'}
        dataset_cls = trackeval.datasets.MotChallenge2DBox
    else:
        yaml_cfg = dict(cfgs['TRACK_EVAL'])
        yaml_cfg['SEQ_INFO'] = {k: v for k, v in yaml_cfg['SEQ_INFO'].items() if k in seqs}      # track.py:203-207
        assert len(yaml_cfg['SEQ_INFO']) == len(seqs)
        yaml_cfg.setdefault('TRACKERS_TO_EVAL', [folder_name])
        dataset_cls = trackeval.datasets.MotChallenge2DBox if opts.dataset in ['mot', 'uavdt'] else trackeval.datasets.VisDrone2DBox
    dataset_config = dataset_cls.get_default_dataset_config()
    dataset_config.update({k: v for k, v in yaml_cfg.items() if k in dataset_config})
    eval_config.update({k: v for k, v in yaml_cfg.items() if k in eval_config})
    metrics_config = {'METRICS': ['HOTA', 'CLEAR', 'Identity'], 'THRESHOLD': 0.5}
    metrics_list = [m(metrics_config) for m in (trackeval.metrics.HOTA, trackeval.metrics.CLEAR, trackeval.metrics.Identity)
                    if m.get_name() in metrics_config['METRICS']]
    return trackeval.Evaluator(eval_config).evaluate([dataset_cls(dataset_config)], metrics_list)


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument('--dataset', type=str, default='visdrone', help='visdrone, or mot')
    parser.add_argument('--data_format', type=str, default='origin', help='format of reading dataset')
    parser.add_argument('--det_output_format', type=str, default='yolo', help='data format of output of detector, yolo or other')
    parser.add_argument('--tracker', type=str, default='sort', help='sort, deepsort, etc')
    parser.add_argument('--model_path', type=str, default='./weights/best.pt', help='model path')
    parser.add_argument('--trace', type=bool, default=False, help='traced model of YOLO v7')
    parser.add_argument('--img_size', nargs='+', type=int, default=1280, help='[train, test] image sizes')
    parser.add_argument('--reid_model_path', type=str, default='./weights/ckpt.t7',
                        help='path for reid model path (an OSNet or a DeepSORT net_dict checkpoint; random[:osnet|:deepsort] = seeded random weights)')
    parser.add_argument('--dhn_path', type=str, default='./weights/DHN.pth', help='path of DHN path for DeepMOT')
    parser.add_argument('--conf_thresh', type=float, default=0.2, help='filter tracks')
    parser.add_argument('--nms_thresh', type=float, default=0.7, help='thresh for NMS')
    parser.add_argument('--iou_thresh', type=float, default=0.5, help='IOU thresh to filter tracks')
    parser.add_argument('--track_buffer', type=int, default=30, help='tracking buffer')
    parser.add_argument('--gamma', type=float, default=0.1, help='param to control fusing motion and apperance dist')
    parser.add_argument('--kalman_format', type=str, default='default', help='use what kind of Kalman, default, naive, strongsort or bot-sort like')
    parser.add_argument('--batch', type=int, default=1, help='(extension) frames per detector forward; the tracker still steps frame by frame')
    parser.add_argument('--device_preprocess', action='store_true', help='(extension) letterbox raw frames on the GPU instead of in the loader')
    parser.add_argument('--min_area', type=float, default=150, help='use to filter small bboxs')
    parser.add_argument('--save_images', action='store_true', help='save tracking results (image)')
    parser.add_argument('--save_videos', action='store_true', help='save tracking results (video)')
    parser.add_argument('--detect_per_frame', type=int, default=1, help='choose how many frames per detect')
    parser.add_argument('--track_eval', type=bool, default=True, help='Use TrackEval to evaluate')
    # additions (defaults keep the reference's behaviour)
    parser.add_argument('--model_cfg', type=str, default=None, help='model yaml / arch name for state-dict checkpoints')
    parser.add_argument('--nc', type=int, default=None)
    parser.add_argument('--synthetic_dets', action='store_true', help='--dataset synthetic: feed the scene detections to the tracker')
    parser.add_argument('--synthetic_frames', type=int, default=100)
    parser.add_argument('--synthetic_objs', type=int, default=80)
    parser.add_argument('--synthetic_seqs', type=int, default=1)
    parser.add_argument('--results_root', type=str, default='./tracker/results')
    parser.add_argument('--yolo_root', type=str, default='./', help="(extension) where ./<dataset>/test.txt of --data_format yolo lives (the reference: the working directory)")
    return parser


def cli(argv=None):
    opts = build_parser().parse_args(argv)
    if opts.dataset == 'synthetic':
        cfgs = {'DATASET_ROOT': '', 'CERTAIN_SEQS': [None], 'IGNORE_SEQS': [None], 'CATEGORY_DICT': {}, 'YAML_DICT': ''}
    else:
        path = f'./tracker/config_files/{opts.dataset}.yaml'
        if not os.path.isfile(path):
            path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'config_files', f'{opts.dataset}.yaml')
        with open(path, 'r') as f:
            cfgs = yaml.load(f, Loader=yaml.FullLoader)
    return main(opts, cfgs)


if __name__ == '__main__':
    cli()
