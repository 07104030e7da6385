"""CPU restatement of util.compute_Rand_F_scores (reference util/util.py:86-128) in plain Python loops -- test infrastructure only.

The reference calls skimage.measure.label(x, background=1) (skimage 2-D default: 8-connectivity, components numbered in raster
order of their first pixel, background pixels = 0); skimage is not in this image, so the labelling is restated as a flood fill and
the joint distribution as the reference's own double loop.  parity unpinned by the reference itself: it ships no vectors for this
metric; the hand-built cases of tests/test_host_logic.py pin the restatement."""
import numpy as np


def label_background1(x):
    """skimage.measure.label(x, background=1) for boolean x: false pixels form the labelled regions (8-connected)."""
    H, W = x.shape
    lab = np.zeros((H, W), dtype=np.int64)
    cur = 0
    for i in range(H):
        for j in range(W):
            if x[i, j] or lab[i, j]:
                continue
            cur += 1
            stack = [(i, j)]
            lab[i, j] = cur
            while stack:
                a, b = stack.pop()
                for da in (-1, 0, 1):
                    for db in (-1, 0, 1):
                        u, v = a + da, b + db
                        if 0 <= u < H and 0 <= v < W and not x[u, v] and not lab[u, v]:
                            lab[u, v] = cur
                            stack.append((u, v))
    return lab


def rand_f_score(s, t):
    """One image pair (2-D arrays)."""
    t = t > 0.5
    s = s > 0.5
    t_label, s_label = label_background1(t), label_background1(s)
    p = np.zeros([t_label.max() + 1, s_label.max() + 1])
    for i in range(t.shape[0]):
        for j in range(t.shape[1]):
            p[t_label[i, j], s_label[i, j]] += 1
    p_ = p[1:, :]
    n = p.sum()
    p_ = p_ / n
    p__ = p_[:, 1:]
    aux = p_[:, 0].sum()
    ai = np.sum(p_, axis=1)
    bj = np.sum(p__, axis=0)
    sumA2 = np.power(ai, 2).sum()
    sumB2 = np.power(bj, 2).sum() + aux / n
    sumAB2 = np.power(p__, 2).sum() + aux / n
    prec = sumAB2 / sumB2
    rec = sumAB2 / sumA2
    return 2 / (1 / prec + 1 / rec)
