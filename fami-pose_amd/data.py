"""Input pipeline on the device: what the reference's DataLoader workers do per clip with cv2 and
torchvision -- one affine crop shared by the key frame and its supporting frames, optional flip, ToTensor + Normalize,
joint transform, visibility rule (datasets/zoo/posetrack/PoseTrack_Alignment.py:380-440,
datasets/process/affine_transform.py:45-82, datasets/process/pose_process.py:12-26,
datasets/transforms/build.py:12-23) -- as host arithmetic on 2x3 matrices plus ONE kernel launch per clip
(`fami_warp_normalize_u8`).  The random draws of the augmentation (scale / rotation / flip) stay with the caller, as in
the reference; this module is deterministic given them.  HIP path only: there is no CPU fallback."""
import numpy as np
import torch

from ._lib import lib

MEAN = (0.485, 0.456, 0.406)      # datasets/transforms/build.py:13-14 (RGB)
STD = (0.229, 0.224, 0.225)
FLIP_PAIRS = ((3, 4), (5, 6), (7, 8), (9, 10), (11, 12), (13, 14), (15, 16))   # PoseTrack_Alignment.py:40


def _solve_affine(src, dst):
    """cv2.getAffineTransform: M with M @ [x, y, 1] = dst for three point pairs (float64)."""
    A = np.zeros((6, 6))
    b = np.zeros(6)
    for i in range(3):
        A[2 * i, 0:3] = (src[i, 0], src[i, 1], 1.0)
        A[2 * i + 1, 3:6] = (src[i, 0], src[i, 1], 1.0)
        b[2 * i], b[2 * i + 1] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(A, b).reshape(2, 3)


def dark_get_affine_transform(center, scale, rot, output_size, inv=0):
    """affine_transform.py:45-77 (shift 0): image -> crop transform anchored on (w-1)/2 pixel centres."""
    scale = np.asarray(scale, np.float64)
    scale_tmp = scale * 200.0
    src_w = scale_tmp[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rad = np.pi * rot / 180
    sn, cs = np.sin(rad), np.cos(rad)
    px, py = 0.0, (src_w - 1) * -0.5
    src_dir = np.array([px * cs - py * sn, px * sn + py * cs])
    dst_dir = np.array([0, (dst_w - 1) * -0.5], np.float32)
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0, :] = center
    src[1, :] = np.asarray(center) + src_dir
    dst[0, :] = [(dst_w - 1) * 0.5, (dst_h - 1) * 0.5]
    dst[1, :] = np.array([(dst_w - 1) * 0.5, (dst_h - 1) * 0.5]) + dst_dir
    for pts in (src, dst):
        d = pts[0] - pts[1]
        pts[2, :] = pts[1] + np.array([-d[1], d[0]], np.float32)
    src, dst = src.astype(np.float64), dst.astype(np.float64)
    return _solve_affine(dst, src) if inv else _solve_affine(src, dst)


def invert_affine(M):
    """The dst -> src map cv2.warpAffine derives from the forward matrix (float64, cv2's operation order)."""
    M = np.asarray(M, np.float64).copy().reshape(6)
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    M[0] = A11
    M[1] *= -D
    M[3] *= -D
    M[4] = A22
    b1 = -M[0] * M[2] - M[1] * M[5]
    b2 = -M[3] * M[2] - M[4] * M[5]
    M[2], M[5] = b1, b2
    return M.reshape(2, 3)


def fliplr_joints(joints, joints_vis, width, matched_parts=FLIP_PAIRS):
    """pose_process.py:12-26 on copies: mirror x, swap left/right joints, zero invisible joints."""
    joints, joints_vis = joints.copy(), joints_vis.copy()
    joints[:, 0] = width - joints[:, 0] - 1
    for a, b in matched_parts:
        joints[a, :], joints[b, :] = joints[b, :], joints[a, :].copy()
        joints_vis[a, :], joints_vis[b, :] = joints_vis[b, :], joints_vis[a, :].copy()
    return joints * joints_vis, joints_vis


def transform_joints(joints, joints_vis, trans, image_size):
    """PoseTrack_Alignment.py:434-443: visible joints through the crop transform (affine_transform.py:79-82); joints that
    leave [0, W] x [0, H] become invisible."""
    joints, joints_vis = joints.copy(), joints_vis.copy()
    for i in range(joints.shape[0]):
        if joints_vis[i, 0] > 0.0:
            joints[i, 0:2] = np.dot(trans, np.array([joints[i, 0], joints[i, 1], 1.0]).T)[:2]
    for i in range(joints.shape[0]):
        x, y = joints[i, 0], joints[i, 1]
        if x < 0 or y < 0 or x > image_size[0] or y > image_size[1]:
            joints_vis[i] = 0
    return joints, joints_vis


def crop_clip(frames_u8, center, scale, rot, image_size, flip=False, bgr=False, out_key=None, out_sup=None):
    """frames_u8: uint8 CUDA tensor [F, Hs, Ws, 3] (key frame first, then the supporting frames) of ONE clip.
    -> (key [3, H, W], sup [3*(F-1), H, W]) fp32 CUDA tensors: the reference's `input_x` and the channel-concatenated
    supporting frames, written into `out_key` / `out_sup` (e.g. slices of the batch tensors) when given; plus the 2x3
    transform for the joints.  center is the caller's (already flipped if flip, PoseTrack_Alignment.py:414)."""
    if not (torch.is_tensor(frames_u8) and frames_u8.is_cuda and frames_u8.dtype == torch.uint8):
        raise RuntimeError('crop_clip: uint8 CUDA frames expected (HIP path only, no CPU fallback)')
    F, Hs, Ws, C = frames_u8.shape
    assert C == 3 and frames_u8.is_contiguous()
    W, H = int(image_size[0]), int(image_size[1])
    trans = dark_get_affine_transform(center, scale, rot, image_size)
    mi = invert_affine(trans)
    dev = frames_u8.device
    key = out_key if out_key is not None else torch.empty(3, H, W, device=dev)
    sup = out_sup if out_sup is not None else torch.empty(3 * (F - 1), H, W, device=dev)
    assert key.is_contiguous() and sup.is_contiguous() and key.dtype == torch.float32 and sup.dtype == torch.float32
    st = torch.cuda.current_stream(dev).cuda_stream
    L = lib()
    common = (float(mi[0, 0]), float(mi[0, 1]), float(mi[0, 2]), float(mi[1, 0]), float(mi[1, 1]), float(mi[1, 2]),
              int(bool(flip)), int(bool(bgr)), *MEAN, *STD, st)
    L.call('fami_warp_normalize_u8', frames_u8.data_ptr(), key.data_ptr(), 1, Hs, Ws, Hs * Ws * 3, H, W, 3 * H * W,
           *common)
    if F > 1:
        L.call('fami_warp_normalize_u8', frames_u8[1:].data_ptr(), sup.data_ptr(), F - 1, Hs, Ws, Hs * Ws * 3, H, W,
               3 * H * W, *common)
    return key, sup, trans


def prepare_clip(frames_u8, joints, joints_vis, center, scale, rot, image_size, flip=False, bgr=False,
                 out_key=None, out_sup=None):
    """One training sample as the reference's `__getitem__` builds it (after its random draws): flip bookkeeping for the
    joints and the centre, crop of every frame on the device, joints through the same transform.
    joints, joints_vis: [J, 3] numpy (image coordinates).  -> key, sup (device), joints [J,3], joints_vis [J,3] (numpy);
    feed joints[:, :2] / joints_vis[:, 0] to Trainer(targets_from_joints=True) for the on-device Gaussian targets."""
    center = np.asarray(center, np.float64).copy()
    if flip:
        Ws = frames_u8.shape[2]
        joints, joints_vis = fliplr_joints(joints, joints_vis, Ws)
        center[0] = Ws - center[0] - 1
    key, sup, trans = crop_clip(frames_u8, center, scale, rot, image_size, flip, bgr, out_key, out_sup)
    joints, joints_vis = transform_joints(joints, joints_vis, trans, image_size)
    return key, sup, joints, joints_vis
