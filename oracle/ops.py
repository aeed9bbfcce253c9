"""Oracle (test infrastructure): op-level CPU restatements, torch fp32 / numpy.

Reference citations are relative to /root/reference.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------
# modulated deformable convolution
# --------------------------------------------------------------------------
def _bilinear_zero(img, y, x):
    """img [B,C,H,W]; y,x [B,G,K,Ho,Wo] float sample positions, one set per
    offset group G (C = G*cg).  Returns [B,G,cg,K,Ho,Wo].  A corner contributes
    only when it lies inside the map (torchvision deform_conv2d bilinear rule).
    """
    B, C, H, W = img.shape
    G = y.shape[1]
    cg = C // G
    y0 = torch.floor(y)
    x0 = torch.floor(x)
    ly, lx = y - y0, x - x0
    hy, hx = 1.0 - ly, 1.0 - lx
    y0, x0 = y0.long(), x0.long()
    flat = img.reshape(B, G, cg, H * W)
    out = 0
    for dy, dx, wgt in ((0, 0, hy * hx), (0, 1, hy * lx), (1, 0, ly * hx), (1, 1, ly * lx)):
        yy, xx = y0 + dy, x0 + dx
        ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W)
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1))          # [B,G,K,Ho,Wo]
        shp = idx.shape
        g = torch.gather(flat, 3, idx.reshape(B, G, 1, -1).expand(B, G, cg, -1))
        g = g.reshape(B, G, cg, *shp[2:])
        out = out + g * (wgt * ok.to(img.dtype)).unsqueeze(2)
    return out


DCN_IMPL = 'gather'   # 'gridsample' selects the (faster on CPU) F.grid_sample formulation of the same op


def deform_conv2d(inp, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1):
    """Modulated deformable conv, torchvision.ops.deform_conv2d semantics, as
    called at posetimation/zoo/Alignment/Alignment_V15.py:146,150,154,158
    (ctor :83,89,95,101: 48->48, 3x3, padding 3, dilation 3, weight groups 1).

    offset [B, 2*G*kh*kw, Ho, Wo] ordered (group, tap row-major, (dy,dx));
    mask   [B,   G*kh*kw, Ho, Wo] multiplicative (NO sigmoid: the reference
    feeds raw conv outputs, Alignment_V15.py:81-82); weight [Co,Ci,kh,kw].
    """
    if DCN_IMPL == 'gridsample':
        return deform_conv2d_gridsample(inp, offset, mask, weight, bias, stride, padding, dilation)
    B, C, H, W = inp.shape
    Co, Ci, kh, kw = weight.shape
    assert Ci == C, "weight groups == 1 only"
    K = kh * kw
    G = offset.shape[1] // (2 * K)
    Ho = (H + 2 * padding - dilation * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * padding - dilation * (kw - 1) - 1) // stride + 1
    off = offset.reshape(B, G, K, 2, Ho, Wo)
    ky = torch.arange(kh, dtype=inp.dtype).repeat_interleave(kw) * dilation
    kx = torch.arange(kw, dtype=inp.dtype).repeat(kh) * dilation
    oy = torch.arange(Ho, dtype=inp.dtype) * stride - padding
    ox = torch.arange(Wo, dtype=inp.dtype) * stride - padding
    py = oy.view(1, 1, 1, Ho, 1) + ky.view(1, 1, K, 1, 1) + off[:, :, :, 0]
    px = ox.view(1, 1, 1, 1, Wo) + kx.view(1, 1, K, 1, 1) + off[:, :, :, 1]
    samp = _bilinear_zero(inp, py, px)                                # [B,G,cg,K,Ho,Wo]
    if mask is not None:
        samp = samp * mask.reshape(B, G, 1, K, Ho, Wo)
    cols = samp.reshape(B, C * K, Ho * Wo)                            # (c, tap) major = weight.view(Co,-1)
    out = torch.matmul(weight.reshape(Co, C * K), cols).reshape(B, Co, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, Co, 1, 1)
    return out


def deform_conv2d_gridsample(inp, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1):
    """Independent formulation of the same op through F.grid_sample
    (align_corners=True, zeros padding, pixel units) used only to cross-check
    deform_conv2d above."""
    B, C, H, W = inp.shape
    Co, _, kh, kw = weight.shape
    K = kh * kw
    G = offset.shape[1] // (2 * K)
    cg = C // G
    Ho, Wo = offset.shape[2:]
    out = torch.zeros(B, Co, Ho, Wo, dtype=inp.dtype)
    ys = torch.arange(Ho, dtype=inp.dtype).view(1, Ho, 1) * stride - padding
    xs = torch.arange(Wo, dtype=inp.dtype).view(1, 1, Wo) * stride - padding
    for g in range(G):
        sub = inp[:, g * cg:(g + 1) * cg]
        for i in range(kh):
            for j in range(kw):
                t = g * K + i * kw + j
                py = ys + i * dilation + offset[:, 2 * t]
                px = xs + j * dilation + offset[:, 2 * t + 1]
                grid = torch.stack([2 * px / max(W - 1, 1) - 1, 2 * py / max(H - 1, 1) - 1], -1)
                s = F.grid_sample(sub, grid, mode='bilinear', padding_mode='zeros', align_corners=True)
                if mask is not None:
                    s = s * mask[:, t:t + 1]
                out = out + torch.einsum('oc,bchw->bohw', weight[:, g * cg:(g + 1) * cg, i, j], s)
    if bias is not None:
        out = out + bias.view(1, Co, 1, 1)
    return out


# --------------------------------------------------------------------------
# global translation (kornia.geometry.warp_affine with M = [[1,0,tx],[0,1,ty]])
# --------------------------------------------------------------------------
def warp_translate(src, t, align_corners=True):
    """out[b,c,y,x] = bilinear(src[b,c], y - ty, x - tx), zero padding, same
    size.  Call site Alignment_V15.py:133-135: M maps src->dst pixel coords, so
    the sampling position is M^-1 x_dst.  Pixel-exact (align_corners=True)
    semantics fixed by this build.  t [B,2] = (tx, ty).
    align_corners=False restates the kornia <= 0.4 default: warp_affine
    normalises M with normal_transform_pixel (x -> 2x/(W-1) - 1), inverts it,
    and feeds affine_grid / grid_sample with align_corners=False; for a pure
    translation that samples at x - tx*W/(W-1) (cross-checked against that
    very pipeline in warp_translate_legacy_gridsample)."""
    B, C, H, W = src.shape
    if not align_corners:
        t = t * torch.tensor([W / max(W - 1, 1), H / max(H - 1, 1)], dtype=t.dtype)
    ys = torch.arange(H, dtype=src.dtype).view(1, 1, 1, H, 1) - t[:, 1].view(B, 1, 1, 1, 1)
    xs = torch.arange(W, dtype=src.dtype).view(1, 1, 1, 1, W) - t[:, 0].view(B, 1, 1, 1, 1)
    py = ys.expand(B, 1, 1, H, W)
    px = xs.expand(B, 1, 1, H, W)
    return _bilinear_zero(src, py, px).reshape(B, C, H, W)


def warp_affine_like(src, M, dsize):
    """Signature-compatible stand-in for kornia.geometry.warp_affine restricted
    to pure translations (what Alignment_V15.py:133-135 builds)."""
    assert tuple(dsize) == tuple(src.shape[2:])
    return warp_translate(src, torch.stack([M[:, 0, 2], M[:, 1, 2]], 1))


def warp_translate_legacy_gridsample(src, t):
    """kornia <= 0.4 warp_affine(src, M, dsize) for M = [[1,0,tx],[0,1,ty]] spelled with torch only:
    normalize_homography with normal_transform_pixel, inverse, F.affine_grid + F.grid_sample, align_corners=False."""
    B, C, H, W = src.shape
    M = torch.eye(3, dtype=src.dtype).repeat(B, 1, 1)
    M[:, 0, 2], M[:, 1, 2] = t[:, 0], t[:, 1]
    Nm = torch.tensor([[2.0 / (W - 1), 0, -1], [0, 2.0 / (H - 1), -1], [0, 0, 1]], dtype=src.dtype)
    dst_norm_trans_src_norm = Nm @ M @ torch.linalg.inv(Nm)
    src_norm_trans_dst_norm = torch.linalg.inv(dst_norm_trans_src_norm)
    grid = F.affine_grid(src_norm_trans_dst_norm[:, :2, :], [B, C, H, W], align_corners=False)
    return F.grid_sample(src, grid, mode='bilinear', padding_mode='zeros', align_corners=False)


def warp_translate_gridsample(src, t):
    B, C, H, W = src.shape
    ys = torch.arange(H, dtype=src.dtype).view(1, H, 1) - t[:, 1].view(B, 1, 1)
    xs = torch.arange(W, dtype=src.dtype).view(1, 1, W) - t[:, 0].view(B, 1, 1)
    grid = torch.stack([(2 * xs / (W - 1) - 1).expand(B, H, W), (2 * ys / (H - 1) - 1).expand(B, H, W)], -1)
    return F.grid_sample(src, grid, mode='bilinear', padding_mode='zeros', align_corners=True)


# --------------------------------------------------------------------------
# MI surrogate (softmax / KL)
# --------------------------------------------------------------------------
MI_TEMPERATURE = 0.05


def softmax_kl_rows(a_rows, b_rows, temperature=MI_TEMPERATURE):
    """mean over ALL elements of t*(log t - a) with a = softmax(A/T) (probabilities,
    not log-probabilities -- reference behaviour, Alignment_V15.py:260,275) and
    t = softmax(Bt/T); rows = dim 0, softmax over dim 1.  Gradient flows only
    through b_rows (a is detached in the reference)."""
    a = torch.softmax(a_rows.detach() / temperature, dim=1)
    t = torch.softmax(b_rows / temperature, dim=1)
    ref_form = (torch.xlogy(t, t) - t * a).mean()          # value: bit-identical to F.kl_div(a, t, 'mean')
    # Gradient: wherever softmax(Bt/T) underflows to exactly 0 the reference's
    # autograd yields NaN (0 * log 0 in the xlogy/mul backward).  The build
    # defines the analytical limit there (t*(log t + 1 - a) -> 0); the two agree
    # whenever the reference is finite.  log_softmax keeps log t finite.
    logt = torch.log_softmax(b_rows / temperature, dim=1)
    safe = (torch.exp(logt) * (logt - a)).mean()
    return safe + (ref_form - safe).detach()


def feat_label_mi(feat, y, final_w, final_b, temperature=MI_TEMPERATURE):
    """Alignment_V15.feat_label_mi_estimation (Alignment_V15.py:250-263):
    A = hrnet.final_layer(Feat) (1x1 conv, detached), Bt = Y; rows [B*J, H*W]."""
    B = feat.shape[0]
    pred = F.conv2d(feat, final_w, final_b)
    J = pred.shape[1]
    return softmax_kl_rows(pred.reshape(B * J, -1), y.reshape(B * J, -1), temperature)


def feat_feat_mi(f1, f2, temperature=MI_TEMPERATURE):
    """Alignment_V15.feat_feat_mi_estimation (Alignment_V15.py:265-277)."""
    B, C = f1.shape[:2]
    return softmax_kl_rows(f1.reshape(B * C, -1), f2.reshape(B * C, -1), temperature)


# --------------------------------------------------------------------------
# heatmap loss / targets / decode
# --------------------------------------------------------------------------
def joint_mse(pred, gt, weight, use_target_weight=True, divided_num_joints=True):
    """JointMSELoss.forward (posetimation/loss/mse_loss.py:21-40):
    (1/J) sum_j mean_{b,p} (w_bj * (pred - gt))^2."""
    B, J = pred.shape[:2]
    p = pred.reshape(B, J, -1)
    g = gt.reshape(B, J, -1)
    if use_target_weight:
        w = weight.reshape(B, J, 1)
        p, g = p * w, g * w
    per_joint = ((p - g) ** 2).mean(dim=(0, 2))
    loss = per_joint.sum()
    return loss / J if divided_num_joints else loss


def generate_heatmaps(joints, joints_vis, sigma, image_size, heatmap_size, num_joints):
    """datasets/process/heatmaps_process.py:146-203.  joints [J,3], joints_vis
    [J,3], image_size/heatmap_size (w,h) -> target [J,Hh,Wh] f32, weight [J,1]."""
    image_size = np.asarray(image_size, dtype=np.float64)
    heatmap_size = np.asarray(heatmap_size)
    Wh, Hh = int(heatmap_size[0]), int(heatmap_size[1])
    weight = np.ones((num_joints, 1), np.float32)
    weight[:, 0] = joints_vis[:, 0]
    target = np.zeros((num_joints, Hh, Wh), np.float32)
    r = sigma * 3
    size = 2 * r + 1
    ax = np.arange(0, size, 1, np.float32)
    g = np.exp(-((ax[None, :] - size // 2) ** 2 + (ax[:, None] - size // 2) ** 2) / (2 * sigma ** 2))
    stride = image_size / heatmap_size
    for j in range(num_joints):
        mx = int(joints[j][0] / stride[0] + 0.5)
        my = int(joints[j][1] / stride[1] + 0.5)
        x0, y0 = int(mx - r), int(my - r)
        x1, y1 = int(mx + r + 1), int(my + r + 1)
        if x0 >= Wh or y0 >= Hh or x1 < 0 or y1 < 0:
            weight[j] = 0
            continue
        if weight[j] > 0.5:
            gx0, gx1 = max(0, -x0), min(x1, Wh) - x0
            gy0, gy1 = max(0, -y0), min(y1, Hh) - y0
            target[j][max(0, y0):min(y1, Hh), max(0, x0):min(x1, Wh)] = g[gy0:gy1, gx0:gx1]
    return target, weight


def get_max_preds(hm):
    """datasets/process/heatmaps_process.py:16-44: row-major flat argmax (first
    max on ties); coords zeroed where max <= 0.  hm ndarray [B,J,H,W]."""
    B, J, H, W = hm.shape
    flat = hm.reshape(B, J, -1)
    idx = np.argmax(flat, 2)
    maxvals = np.amax(flat, 2).reshape(B, J, 1)
    preds = np.zeros((B, J, 2), np.float32)
    preds[:, :, 0] = (idx % W).astype(np.float32)
    preds[:, :, 1] = np.floor(idx.astype(np.float32) / W)
    preds *= (maxvals > 0.0).astype(np.float32)
    return preds, maxvals


def argmax_indices(hm):
    """flat argmax indices [B,J] int64 (the bit-exact contract of north_star)."""
    B, J = hm.shape[:2]
    return np.argmax(hm.reshape(B, J, -1), 2)


def accuracy(output, target, thr=0.5):
    """engine/core/utils/evaluate.py:39-75 (hm_type='gaussian'): PCK on heatmap
    argmax, norm = (H,W)/10, targets with x<=1 or y<=1 ignored."""
    pred, _ = get_max_preds(output)
    tgt, _ = get_max_preds(target)
    B, J = pred.shape[:2]
    h, w = output.shape[2], output.shape[3]
    norm = np.ones((B, 2)) * np.array([h, w]) / 10
    dists = np.zeros((J, B))
    for n in range(B):
        for c in range(J):
            if tgt[n, c, 0] > 1 and tgt[n, c, 1] > 1:
                dists[c, n] = np.linalg.norm(pred[n, c].astype(np.float32) / norm[n] - tgt[n, c].astype(np.float32) / norm[n])
            else:
                dists[c, n] = -1
    acc = np.zeros(J + 1)
    avg, cnt = 0.0, 0
    for c in range(J):
        use = dists[c] != -1
        if use.sum() > 0:
            acc[c + 1] = (dists[c][use] < thr).sum() * 1.0 / use.sum()
        else:
            acc[c + 1] = -1
        if acc[c + 1] >= 0:
            avg += acc[c + 1]
            cnt += 1
    avg = avg / cnt if cnt != 0 else 0
    if cnt != 0:
        acc[0] = avg
    return acc, avg, cnt, pred


def cv2_get_affine_transform(src, dst):
    """cv2.getAffineTransform(src, dst): the 2x3 matrix M with M @ [x, y, 1] = dst for three point pairs.  cv2 is a
    third-party package absent from this image; this restates its documented definition (exact 6x6 linear solve in
    float64).  Used by the oracle AND injected as the reference's `cv2.getAffineTransform` when golden vectors are
    generated (oracle/ref_harness.py), so get_final_preds is pinned up to this one call."""
    src = np.asarray(src, np.float64)
    dst = np.asarray(dst, np.float64)
    A = np.zeros((6, 6))
    bvec = np.zeros(6)
    for i in range(3):
        A[2 * i, 0:3] = (src[i, 0], src[i, 1], 1.0)
        A[2 * i + 1, 3:6] = (src[i, 0], src[i, 1], 1.0)
        bvec[2 * i], bvec[2 * i + 1] = dst[i, 0], dst[i, 1]
    return np.linalg.solve(A, bvec).reshape(2, 3)


def get_affine_transform(center, scale, rot, output_size, inv=0):
    """datasets/process/affine_transform.py:13-43 (shift = 0)."""
    scale = np.asarray(scale, np.float64)
    scale_tmp = scale * 200.0
    src_w = scale_tmp[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rad = np.pi * rot / 180
    sn, cs = np.sin(rad), np.cos(rad)
    pt = (0.0, src_w * -0.5)
    src_dir = np.array([pt[0] * cs - pt[1] * sn, pt[0] * sn + pt[1] * cs])
    dst_dir = np.array([0, dst_w * -0.5], np.float32)
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0, :] = center
    src[1, :] = np.asarray(center) + src_dir
    dst[0, :] = [dst_w * 0.5, dst_h * 0.5]
    dst[1, :] = np.array([dst_w * 0.5, dst_h * 0.5]) + dst_dir

    def third(a, b):
        d = a - b
        return b + np.array([-d[1], d[0]], np.float32)
    src[2, :] = third(src[0], src[1])
    dst[2, :] = third(dst[0], dst[1])
    return cv2_get_affine_transform(dst, src) if inv else cv2_get_affine_transform(src, dst)


def get_final_preds(batch_heatmaps, center, scale):
    """datasets/process/heatmaps_process.py:47-73: argmax, quarter-pixel shift toward the higher neighbour,
    transform_preds (inverse affine, rot 0) to image coordinates."""
    coords, maxvals = get_max_preds(batch_heatmaps)
    H, W = batch_heatmaps.shape[2], batch_heatmaps.shape[3]
    for n in range(coords.shape[0]):
        for p in range(coords.shape[1]):
            hm = batch_heatmaps[n][p]
            px = int(math.floor(coords[n][p][0] + 0.5))
            py = int(math.floor(coords[n][p][1] + 0.5))
            if 1 < px < W - 1 and 1 < py < H - 1:
                diff = np.array([hm[py][px + 1] - hm[py][px - 1], hm[py + 1][px] - hm[py - 1][px]])
                coords[n][p] += np.sign(diff) * .25
    preds = coords.copy()
    for i in range(coords.shape[0]):
        trans = get_affine_transform(center[i], scale[i], 0, [W, H], inv=1)
        out = np.zeros(coords[i].shape)
        for p in range(coords.shape[1]):
            out[p, 0:2] = trans @ np.array([coords[i][p, 0], coords[i][p, 1], 1.0])
        preds[i] = out
    return preds, maxvals


def total_loss(final_hm, target, weight, mi_list, mse_weight=1.0, alpha=0.5, beta=0.1):
    """Loss assembly of engine/core/functions/alignment_mi_function_term6_1.py:108-148
    (local_warped_sup_hm_list is empty for Alignment_V15)."""
    loss = joint_mse(final_hm, target, weight) * mse_weight
    if len(mi_list) > 0:
        m1, m2, m3, m4, m5, m6 = mi_list
        loss = loss + alpha * (-beta * m1 + beta * m2 + m3 - m4 + m5 - m6)
    return loss


# ------------------------------------------------------------------ input pipeline
def dark_get_affine_transform(center, scale, rot, output_size, inv=0):
    """datasets/process/affine_transform.py:45-77 (shift = 0): the crop transform of the training pipeline
    (PoseTrack_Alignment.py:420), anchored on (w-1)/2 pixel centres."""
    scale = np.asarray(scale, np.float64)
    scale_tmp = scale * 200.0
    src_w = scale_tmp[0]
    dst_w, dst_h = output_size[0], output_size[1]
    rad = np.pi * rot / 180
    sn, cs = np.sin(rad), np.cos(rad)
    pt = (0.0, (src_w - 1) * -0.5)
    src_dir = np.array([pt[0] * cs - pt[1] * sn, pt[0] * sn + pt[1] * cs])
    dst_dir = np.array([0, (dst_w - 1) * -0.5], np.float32)
    src = np.zeros((3, 2), np.float32)
    dst = np.zeros((3, 2), np.float32)
    src[0, :] = center
    src[1, :] = np.asarray(center) + src_dir
    dst[0, :] = [(dst_w - 1) * 0.5, (dst_h - 1) * 0.5]
    dst[1, :] = np.array([(dst_w - 1) * 0.5, (dst_h - 1) * 0.5]) + dst_dir

    def third(a, b):
        d = a - b
        return b + np.array([-d[1], d[0]], np.float32)
    src[2, :] = third(src[0], src[1])
    dst[2, :] = third(dst[0], dst[1])
    return cv2_get_affine_transform(dst, src) if inv else cv2_get_affine_transform(src, dst)


def exec_affine_transform(pt, t):
    """affine_transform.py:79-82."""
    return np.dot(t, np.array([pt[0], pt[1], 1.0]).T)[:2]


def fliplr_joints(joints, joints_vis, width, matched_parts):
    """datasets/process/pose_process.py:12-26 (operates on copies)."""
    joints, joints_vis = joints.copy(), joints_vis.copy()
    joints[:, 0] = width - joints[:, 0] - 1
    for a, b in matched_parts:
        joints[a, :], joints[b, :] = joints[b, :], joints[a, :].copy()
        joints_vis[a, :], joints_vis[b, :] = joints_vis[b, :], joints_vis[a, :].copy()
    return joints * joints_vis, joints_vis


def cv2_invert_affine(M):
    """The inverse map cv2.warpAffine derives from the forward 2x3 matrix (float64, its own operation order)."""
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


def cv2_warp_affine_u8(src, M, dsize, flip=False):
    """cv2.warpAffine(src, M, dsize, flags=cv2.INTER_LINEAR) for 8-bit HWC images, borderMode=BORDER_CONSTANT(0)
    (PoseTrack_Alignment.py:421-427).  PARITY UNPINNED: cv2 is a third-party package absent from this image and the
    reference pins no version; this restates OpenCV's published generic (non-IPP) algorithm: inverse map in float64,
    source coordinates in fixed point with 10 fractional bits rounded to 1/32 pixel (AB_BITS 10, INTER_BITS 5), bilinear
    weights (32-fy)(32-fx)*32 ... summing to 2^15, result (sum + 2^14) >> 15, taps outside the image read 0.
    flip: the source is mirrored in x first (`data_numpy[:, ::-1, :]`, :409)."""
    src = np.asarray(src)
    if flip:
        src = src[:, ::-1, :]
    Hs, Ws = src.shape[:2]
    Wd, Hd = int(dsize[0]), int(dsize[1])
    Mi = cv2_invert_affine(M)
    x = np.arange(Wd, dtype=np.float64)
    y = np.arange(Hd, dtype=np.float64)
    rnd = lambda v: np.clip(np.rint(v), -2147483648.0, 2147483647.0).astype(np.int64)     # saturate_cast<int>
    adelta = rnd(Mi[0, 0] * x * 1024.0)
    bdelta = rnd(Mi[1, 0] * x * 1024.0)
    X0 = rnd((Mi[0, 1] * y + Mi[0, 2]) * 1024.0) + 16
    Y0 = rnd((Mi[1, 1] * y + Mi[1, 2]) * 1024.0) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx = np.clip(X >> 5, -32768, 32767)
    sy = np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    w = [(32 - fy) * (32 - fx) * 32, (32 - fy) * fx * 32, fy * (32 - fx) * 32, fy * fx * 32]
    out = np.zeros((Hd, Wd, src.shape[2]), np.int64)
    for k, (dy, dx) in enumerate(((0, 0), (0, 1), (1, 0), (1, 1))):
        yy, xx = sy + dy, sx + dx
        ok = (yy >= 0) & (yy < Hs) & (xx >= 0) & (xx < Ws)
        tap = src[np.clip(yy, 0, Hs - 1), np.clip(xx, 0, Ws - 1)].astype(np.int64) * ok[..., None]
        out += tap * w[k][..., None]
    return ((out + (1 << 14)) >> 15).astype(np.uint8)


def to_tensor_normalize(img_u8, mean, std):
    """transforms.ToTensor + transforms.Normalize (datasets/transforms/build.py:12-23): HWC uint8 -> CHW fp32,
    ((v / 255) - mean) / std evaluated in fp32 in this order."""
    t = torch.from_numpy(np.ascontiguousarray(img_u8.transpose(2, 0, 1))).float().div(255)
    m = torch.tensor(mean, dtype=torch.float32)[:, None, None]
    s = torch.tensor(std, dtype=torch.float32)[:, None, None]
    return t.sub_(m).div_(s)
