"""Evaluation metrics of the reference (`compute_metrics`, /root/reference/chexpert.py:130-146).

The reference calls sklearn `roc_curve` / `auc` / `precision_recall_curve` per class on raw logits.
Here the ROC / PR curves are built directly (descending score sweep with tie groups collapsed, as
sklearn's `_binary_clf_curve` does) and the AUROC is the trapezoid area.  Host-side numpy: this runs once
per evaluation on (N, 5) arrays and is not on the GPU hot path.
"""
import numpy as np


def _binary_curve(y_true, score):
    y_true = np.asarray(y_true, dtype=np.float64) > 0.5
    score = np.asarray(score, dtype=np.float64)
    order = np.argsort(-score, kind="mergesort")
    y, s = y_true[order], score[order]
    distinct = np.where(np.diff(s))[0]
    idx = np.r_[distinct, y.size - 1]                 # last index of every tie group
    tps = np.cumsum(y)[idx].astype(np.float64)
    fps = (1 + idx) - tps
    return fps, tps, s[idx]


def roc_curve(y_true, score):
    fps, tps, thr = _binary_curve(y_true, score)
    fps, tps = np.r_[0.0, fps], np.r_[0.0, tps]
    if fps[-1] <= 0 or tps[-1] <= 0:
        nan = np.full(fps.shape, np.nan)
        return (nan if fps[-1] <= 0 else fps / fps[-1]), (nan if tps[-1] <= 0 else tps / tps[-1]), thr
    return fps / fps[-1], tps / tps[-1], thr


_trapz = getattr(np, "trapezoid", None) or np.trapz      # NumPy >= 2.0 / 1.x


def auc(x, y):
    return float(_trapz(y, x)) if not (np.any(np.isnan(x)) or np.any(np.isnan(y))) else float("nan")


def precision_recall_curve(y_true, score):
    fps, tps, thr = _binary_curve(y_true, score)
    precision = tps / np.maximum(tps + fps, 1e-300)
    recall = tps / tps[-1] if tps[-1] > 0 else np.ones_like(tps)
    sl = slice(None, None, -1)
    return np.r_[precision[sl], 1.0], np.r_[recall[sl], 0.0], thr[sl]


def compute_metrics(outputs, targets, losses):
    """Same dictionary layout as chexpert.py:130-146 (lists per class, json-serialisable)."""
    outputs, targets, losses = (np.asarray(t, dtype=np.float64) for t in (outputs, targets, losses))
    fpr, tpr, aucs, precision, recall = {}, {}, {}, {}, {}
    for i in range(outputs.shape[1]):
        f, t, _ = roc_curve(targets[:, i], outputs[:, i])
        aucs[i] = auc(f, t)
        p, r, _ = precision_recall_curve(targets[:, i], outputs[:, i])
        fpr[i], tpr[i], precision[i], recall[i] = f.tolist(), t.tolist(), p.tolist(), r.tolist()
    return {"fpr": fpr, "tpr": tpr, "aucs": aucs, "precision": precision, "recall": recall,
            "loss": dict(enumerate(losses.mean(0).tolist()))}


def mean_auc(metrics):
    v = np.array(list(metrics["aucs"].values()), dtype=np.float64)
    return float(np.nanmean(v)) if np.any(~np.isnan(v)) else float("nan")
