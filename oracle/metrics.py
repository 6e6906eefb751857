"""Oracle (test infrastructure): per-class ROC-AUC as chexpert.py:130-146 computes it.

The reference calls sklearn `roc_curve` + `auc` (trapezoid over the ROC with ties collapsed) on raw
logits per class and averages with `np.nanmean` (chexpert.py:189).  The trapezoid area with tie
handling equals the Mann-Whitney statistic with mid-ranks, which is restated here in numpy and
pinned against sklearn 1.7.2 outputs in tests/golden/auroc.json.
"""
import numpy as np


def roc_auc(y_true, score):
    """AUROC of one class; NaN if only one class is present (sklearn: warning + nan)."""
    y_true = np.asarray(y_true, dtype=np.float64)
    score = np.asarray(score, dtype=np.float64)
    pos = y_true > 0.5
    n_pos, n_neg = int(pos.sum()), int((~pos).sum())
    if n_pos == 0 or n_neg == 0:
        return float("nan")
    order = np.argsort(score, kind="mergesort")
    s = score[order]
    ranks = np.empty(len(s), dtype=np.float64)
    i = 0
    while i < len(s):                       # mid-ranks over tie groups
        j = i
        while j + 1 < len(s) and s[j + 1] == s[i]:
            j += 1
        ranks[i:j + 1] = 0.5 * (i + j) + 1.0
        i = j + 1
    r = np.empty_like(ranks)
    r[order] = ranks
    return float((r[pos].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def per_class_auc(outputs, targets):
    """-> ({class: auc}, nanmean) for (N, n_classes) logits / labels."""
    outputs, targets = np.asarray(outputs), np.asarray(targets)
    aucs = {i: roc_auc(targets[:, i], outputs[:, i]) for i in range(outputs.shape[1])}
    vals = np.array(list(aucs.values()), dtype=np.float64)
    mean = float(np.nanmean(vals)) if np.any(~np.isnan(vals)) else float("nan")
    return aucs, mean
