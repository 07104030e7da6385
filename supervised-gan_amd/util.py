"""Image and metric helpers of the drivers (util/util.py:15-28, :41-43, :86-128 of the reference): tensor -> uint8 image -> PNG,
the Rand F-score of a binary segmentation."""
import os

import numpy as np


def tensor2im(image_tensor, imtype=np.uint8):
    """[1, C, H, W] in [-1, 1] -> [H, W, 3] uint8; 1 channel is repeated, 2 channels get a zero blue plane (util.py:15-24)."""
    image_numpy = image_tensor[0].detach().cpu().float().numpy()
    image_numpy = (image_numpy + 1) / 2.0 * 255.0
    if image_numpy.shape[0] == 1:
        image_numpy = image_numpy.repeat(3, 0)
    elif image_numpy.shape[0] == 2:
        image_numpy = np.concatenate((image_numpy, np.zeros((1,) + image_numpy.shape[1:], dtype=image_numpy.dtype)), axis=0)
    return np.transpose(image_numpy, (1, 2, 0)).clip(0, 255).astype(imtype)


def save_image(image_numpy, image_path):
    from PIL import Image
    os.makedirs(os.path.dirname(os.path.abspath(image_path)), exist_ok=True)
    Image.fromarray(image_numpy).save(image_path)


def _label_false_regions(mask):
    """skimage.measure.label(mask, background=1) for a boolean 2-D image: the 8-connected components of the FALSE pixels numbered
    from 1, true pixels = 0 (the reference labels the membrane-free regions of the thresholded maps, util.py:101-102)."""
    from scipy import ndimage
    lab, _ = ndimage.label(~mask, structure=np.ones((3, 3), dtype=np.int32))
    return lab


def compute_Rand_F_scores(S, T, do_thin=False):
    """Rand F-score of prediction S against ground truth T per image (util/util.py:86-128): both thresholded at 0.5, the regions
    between the (true) boundary pixels labelled by 8-connectivity, then  F = 2 / (1/prec + 1/rec)  with
    prec = sum_ij p_ij^2 / sum_j b_j^2,  rec = sum_ij p_ij^2 / sum_i a_i^2  over the joint label distribution p without the
    ground-truth background row; prediction-background pixels count as singletons (the `aux / n` terms).
    S, T: [N, 1, H, W] (or a single [H, W] pair).  The contingency table is one bincount instead of the reference's pixel loop."""
    S, T = np.asarray(S), np.asarray(T)
    if S.ndim == 2:
        S, T = S.reshape((1, 1) + S.shape), T.reshape((1, 1) + T.shape)
    if do_thin:
        raise NotImplementedError("do_thin needs skimage.morphology.thin, which this image does not carry")
    scores = np.zeros(T.shape[0])
    for k in range(T.shape[0]):
        t_label = _label_false_regions(T[k].squeeze(axis=0) > 0.5)
        s_label = _label_false_regions(S[k].squeeze(axis=0) > 0.5)
        t_max, s_max = int(t_label.max()), int(s_label.max())
        p = np.bincount((t_label.astype(np.int64) * (s_max + 1) + s_label).ravel(), minlength=(t_max + 1) * (s_max + 1))
        p = p.reshape(t_max + 1, s_max + 1).astype(np.float64)
        n = p.sum()
        p_ = p[1:, :] / n
        p__ = p_[:, 1:]
        aux = p_[:, 0].sum()
        sumA2 = np.power(p_.sum(axis=1), 2).sum()
        sumB2 = np.power(p__.sum(axis=0), 2).sum() + aux / n
        sumAB2 = np.power(p__, 2).sum() + aux / n
        prec, rec = sumAB2 / sumB2, sumAB2 / sumA2
        scores[k] = 2 / (1 / prec + 1 / rec)
    return scores
