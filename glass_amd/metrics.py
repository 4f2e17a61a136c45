"""Evaluation metrics on host numpy arrays, as /root/reference/impl/metrics.py:5-27 (CPU, once per
evaluation — outside the accelerated path)."""
import numpy as np
from sklearn.metrics import f1_score, roc_auc_score


def binaryf1(pred, label):
    """micro-F1 of (logit > 0) against binary / multi-label targets."""
    return f1_score(label.reshape(pred.shape[0], -1), (pred > 0).astype(np.int64), average="micro")


def microf1(pred, label):
    """multi-class micro-F1 of argmax."""
    return f1_score(label, np.argmax(pred, axis=1), average="micro")


def auroc(pred, label):
    return roc_auc_score(label, pred)
