"""Evaluation driver of the hot path.

* `predict` -- one inference step: what the reference's eval loop does per batch
  (engine/core/functions/alignment_mi_function_term6_1.py:256-276): `model(kf, sup) -> (pred, kf_bb)` under
  `torch.no_grad()`, then `get_final_preds(pred, center, scale)` -- with the decode on device, so only [B,J,2]
  coordinates and [B,J,1] confidences leave the GPU instead of two full heatmap stacks.
* `EvalAccumulator` + `evaluate_loop` -- the loop around it (:222-328): per-batch PCK of both heatmap outputs (on
  device), `all_preds` / `all_bb` / `all_boxes` / `filenames_map` accumulation exactly as the reference lays them out.
* `write_posetrack_results` -- the PoseTrack JSON writer of
  datasets/zoo/posetrack/PoseTrack_Alignment.py:883-1017 (`evaluate` up to the poseval call): COCO->PoseTrack joint
  conversion (datasets/process/structure/keypoints_ord.py:14-73), the annorect structure
  (datasets/process/structure/data_format.py:12-49), empty frames filled in, one `{'annolist': [...]}` file per video
  named after the annotation file (posetrack_utils.py:13-56).

Out of scope: the vendored poseval AP/MOTA computation (`evaluate_simple.evaluate`: needs shapely and the PoseTrack
ground truth).  The files written here are its input.

The reference imports its joint-name lists from modules that are missing from the released tree
(`datasets.zoo.coco`, `datasets.zoo.posetrack.pose_topology`, keypoints_ord.py:10-11); COCO_JOINT / POSETRACK_JOINT
below are the standard COCO-17 and PoseTrack-15 orders of DCPose / Detect-and-Track, which the README defers to.
"""
import json
import os
import os.path as osp

import numpy as np
import torch

from .loss import get_final_preds

COCO_JOINT = ['nose', 'left_eye', 'right_eye', 'left_ear', 'right_ear', 'left_shoulder', 'right_shoulder',
              'left_elbow', 'right_elbow', 'left_wrist', 'right_wrist', 'left_hip', 'right_hip', 'left_knee',
              'right_knee', 'left_ankle', 'right_ankle']
POSETRACK_JOINT = ['right_ankle', 'right_knee', 'right_hip', 'left_hip', 'left_knee', 'left_ankle', 'right_wrist',
                   'right_elbow', 'right_shoulder', 'left_shoulder', 'left_elbow', 'left_wrist', 'neck', 'nose',
                   'head_top']


@torch.no_grad()
def predict(model, kf_x, sup_x, center, scale):
    """-> (preds [B,J,2] image coordinates, maxvals [B,J,1], final_hm [B,J,H/4,W/4]) as device tensors.
    `model` is a val/test-phase Alignment_V15 (2-tuple forward) or a train-phase one (3-tuple; MI terms ignored)."""
    out = model(kf_x, sup_x)
    final_hm = out[0]
    preds, maxvals = get_final_preds(final_hm, center, scale)
    return preds, maxvals, final_hm


# ---------------------------------------------------------------------------------------- accumulation (core fn :236-309)
class EvalAccumulator:
    """The arrays the reference's eval loop fills batch by batch (core fn :238-247, :283-309)."""

    def __init__(self, num_samples, num_joints=17):
        self.all_preds = np.zeros((num_samples, num_joints, 3), dtype=np.float64)
        self.all_bb = np.zeros((num_samples, num_joints, 3), dtype=np.float64)
        self.all_boxes = np.zeros((num_samples, 6))
        self.filenames_map = {}
        self.image_path = []
        self.idx = 0
        self._counter = 0
        self.acc_sum = [0.0, 0.0]         # AverageMeter.update(avg_acc, cnt) of final_hm / kf_bb_hm (:266-270)
        self.acc_cnt = [0, 0]

    def add_images(self, names):
        for nm in names:                                        # :283-289
            self.filenames_map.setdefault(nm, []).append(self._counter)
            self._counter += 1
        self.image_path.extend(names)

    def add(self, pred_coord, pred_maxvals, bb_coord, bb_maxvals, center, scale, score, names):
        """One batch: decoded coordinates / confidences of final_hm and kf_bb_hm (host arrays), box meta."""
        n = len(names)
        self.add_images(names)
        i = self.idx
        self.all_preds[i:i + n, :, :2] = pred_coord             # :296-302
        self.all_preds[i:i + n, :, 2:3] = pred_maxvals
        self.all_bb[i:i + n, :, :2] = bb_coord
        self.all_bb[i:i + n, :, 2:3] = bb_maxvals
        center, scale = np.asarray(center), np.asarray(scale)
        self.all_boxes[i:i + n, 0:2] = center[:, 0:2]           # :304-307
        self.all_boxes[i:i + n, 2:4] = scale[:, 0:2]
        self.all_boxes[i:i + n, 4] = np.prod(scale * 200, 1)
        self.all_boxes[i:i + n, 5] = np.asarray(score)
        self.idx += n

    def add_accuracy(self, k, avg_acc, cnt):
        self.acc_sum[k] += avg_acc * cnt
        self.acc_cnt[k] += cnt

    def accuracy(self, k=0):
        return self.acc_sum[k] / self.acc_cnt[k] if self.acc_cnt[k] else 0.0


@torch.no_grad()
def evaluate_loop(model, batches, num_samples, num_joints=17):
    """The reference's validation loop (core fn :256-309) on the HIP path.  `batches` yields
    (kf_x, sup_x, target_heatmaps | None, meta) with meta = {'image': [names], 'center': [B,2], 'scale': [B,2],
    'score': [B]}; tensors may live on the host (they are moved to the model's device).  -> EvalAccumulator."""
    from ._lib import lib
    dev = next(model.parameters()).device
    acc = EvalAccumulator(num_samples, num_joints)
    model.eval()
    for kf_x, sup_x, target, meta in batches:
        kf_x, sup_x = kf_x.to(dev), sup_x.to(dev)
        out = model(kf_x, sup_x)
        pred_hm, kf_bb_hm = out[0], out[1]
        center = np.asarray(meta['center'], dtype=np.float32)
        scale = np.asarray(meta['scale'], dtype=np.float32)
        pc, pm = get_final_preds(pred_hm, center, scale)
        bc, bm = get_final_preds(kf_bb_hm, center, scale)
        if target is not None:
            B, J, H, W = pred_hm.shape
            target = target.to(dev).float().contiguous()
            rows = torch.empty(2, J + 3, device=dev)
            iws = torch.empty(2 * B * J, dtype=torch.int64, device=dev)
            mws = torch.empty(2 * B * J, device=dev)
            s = torch.cuda.current_stream(dev).cuda_stream
            for k, hm in enumerate((pred_hm, kf_bb_hm)):
                lib().call('fami_pck_accuracy_f32', hm.data_ptr(), target.data_ptr(), rows[k].data_ptr(), iws.data_ptr(),
                           mws.data_ptr(), B, J, H, W, 0.5, s)
            r = rows.cpu().numpy()
            for k in range(2):
                acc.add_accuracy(k, float(r[k, J + 1]), int(r[k, J + 2]))
        acc.add(pc.cpu().numpy(), pm.cpu().numpy(), bc.cpu().numpy(), bm.cpu().numpy(), center, scale,
                np.asarray(meta['score']), list(meta['image']))
    return acc


# ---------------------------------------------------------------------------------------- PoseTrack JSON writer
def coco2posetrack_ord(preds, global_score=1):
    """keypoints_ord.py:14-73.  preds: 4x17 (x, y, score, score) in COCO order -> list of 15 PoseTrack points."""
    data = []
    src, dst = COCO_JOINT, POSETRACK_JOINT
    global_score = float(global_score)
    rsho, lsho, nose = src.index('right_shoulder'), src.index('left_shoulder'), src.index('nose')
    for k, name in enumerate(dst):
        if name in src:
            ind = src.index(name)
            local_score = (preds[2, ind] + preds[2, ind]) / 2.0
            data.append({'id': [k], 'x': [float(preds[0, ind])], 'y': [float(preds[1, ind])],
                         'score': [local_score * global_score]})
        elif name == 'neck':
            x = (preds[0, rsho] + preds[0, lsho]) / 2.0
            y = (preds[1, rsho] + preds[1, lsho]) / 2.0
            local_score = (preds[2, rsho] + preds[2, lsho]) / 2.0
            data.append({'id': [k], 'x': [float(x)], 'y': [float(y)], 'score': [local_score * global_score]})
        elif name == 'head_top':
            x_msho = (preds[0, rsho] + preds[0, lsho]) / 2.0
            y_msho = (preds[1, rsho] + preds[1, lsho]) / 2.0
            x_nose, y_nose = preds[0, nose], preds[1, nose]
            local_score = (preds[2, rsho] + preds[2, lsho]) / 2.0
            data.append({'id': [k], 'x': [float(x_nose - (x_msho - x_nose))], 'y': [float(y_nose - (y_msho - y_nose))],
                         'score': [local_score]})        # :72: the head-top score is NOT scaled by the box score
    return data


def convert_data_to_annorect_struct(poses, tracks, boxes, eval_tracking=False, tracking_threshold=0):
    """data_format.py:12-49."""
    annorect = []
    for j in range(len(poses)):
        score = boxes[j][0, 5]
        if eval_tracking and score > tracking_threshold:
            continue
        annorect.append({'annopoints': [{'point': coco2posetrack_ord(poses[j], global_score=score)}],
                         'score': [float(score)], 'track_id': [tracks[j]]})
    if len(poses) == 0:      # "MOTA requires each image to have at least one detection": a dummy prediction
        annorect.append({'annopoints': [{'point': [{'id': [0], 'x': [0], 'y': [0], 'score': [-100.0]}]}],
                         'score': [0], 'track_id': [0]})
    return annorect


def video2filenames(annot_dir):
    """posetrack_utils.py:13-56: {video dir -> annotation file name}, {video dir -> number of frames}."""
    files = [f for f in os.listdir(annot_dir) if osp.isfile(osp.join(annot_dir, f))]
    json_files = [f for f in files if '.json' in f]
    mat_files = [f for f in files if '.mat' in f]
    output, L = {}, {}
    if len(json_files) > 1:
        for fname in json_files:
            with open(osp.join(annot_dir, fname)) as fin:
                data = json.load(fin)
            if 'annolist' in data:
                temp = data['annolist'][0]['image'][0]['name']
                num_frames = len(data['annolist'])
            else:
                temp = data['images'][0]['file_name']
                num_frames = data['images'][0]['nframes']
            video = osp.dirname(temp)
            output[video], L[video] = fname, num_frames
    else:
        import scipy.io as sio
        for fname in mat_files:
            data = sio.loadmat(osp.join(annot_dir, fname), squeeze_me=True, struct_as_record=False)
            temp = data['annolist'][0].image.name
            num_frames = len(sio.loadmat(osp.join(annot_dir, fname))['annolist'][0])
            video = osp.dirname(temp)
            output[video], L[video] = fname.replace('.mat', '.json'), num_frames
    return output, L


def write_posetrack_results(preds, boxes, filenames_map, annot_dir, output_dir, is_posetrack18=False, phase='validate'):
    """PoseTrack_Alignment.evaluate up to the poseval call (:883-1017).  preds [N,17,3] (x, y, confidence in image
    coordinates), boxes [N,6] (center, scale, area, score), filenames_map {image path -> [row indices]}.
    Writes one JSON file per video under output_dir/{val,test}_set_json_results and returns {file path: video}."""
    output_dir = osp.join(output_dir, 'val_set_json_results' if phase == 'validate' else 'test_set_json_results')
    os.makedirs(output_dir, exist_ok=True)
    video_map, vid2frame_map, vid2name_map = {}, {}, {}
    all_preds, all_boxes = [], []
    cc = 0
    for key in filenames_map:
        parts = key.split('/')
        video_name = parts[-3] + '/' + parts[-2]
        img_sfx = parts[-3] + '/' + parts[-2] + '/' + parts[-1]
        frame_num = int(parts[-1].replace('.jpg', ''))
        video_map.setdefault(video_name, []).append(cc)
        vid2frame_map.setdefault(video_name, []).append(frame_num)
        vid2name_map.setdefault(video_name, []).append(img_sfx)
        pose_list, box_list = [], []
        for idx in filenames_map[key]:
            t = np.zeros((4, 17))
            t[0, :], t[1, :], t[2, :], t[3, :] = preds[idx, :, 0], preds[idx, :, 1], preds[idx, :, 2], preds[idx, :, 2]
            pose_list.append(t)
            b = np.zeros((1, 6))
            b[0, :] = boxes[idx, :]
            box_list.append(b)
        all_preds.append(pose_list)
        all_boxes.append(box_list)
        cc += 1

    out_filenames, L = video2filenames(annot_dir)
    out_data = {}
    for vid in video_map:
        cur_length = L['images/' + vid]
        kps_map, box_map, used = {}, {}, []
        for c, idx in enumerate(video_map[vid]):
            frame_num = vid2frame_map[vid][c]
            used.append(frame_num)
            kps_map[frame_num] = (vid2name_map[vid][c], all_preds[idx])
            box_map[frame_num] = all_boxes[idx]
        sid, fid = (1, cur_length + 1) if not is_posetrack18 else (0, cur_length)
        for frame_num in range(sid, fid):
            if frame_num not in used:            # frames without a detection still get an (empty) entry
                arr = vid2name_map[vid][0].split('/')
                img_sfx = arr[0] + '/' + arr[1] + '/' + str(frame_num).zfill(6 if is_posetrack18 else 8) + '.jpg'
                kps, tracks, bboxs = [], [], []
            else:
                img_sfx, kps = kps_map[frame_num]
                bboxs = box_map[frame_num]
                tracks = list(range(len(kps)))
            out_data.setdefault(vid, []).append({'image': {'name': img_sfx}, 'imgnum': [frame_num],
                                                 'annorect': convert_data_to_annorect_struct(kps, tracks, bboxs)})
    written = {}
    for vname, vdata in out_data.items():
        path = osp.join(output_dir, out_filenames[osp.join('images', vname)])
        with open(path, 'w') as f:
            json.dump({'annolist': vdata}, f)
        written[path] = vname
    return written
