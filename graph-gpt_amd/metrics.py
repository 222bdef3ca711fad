"""Fine-tune metrics computed from `task_logits`: build-side counterparts of
  SingleLabelClassificationMetrics.update   reference src/utils/metrics_utils.py:38-56   (score = logit1 - logit0 -> AUROC / ACC)
  _eval_ogbl_ppa (OGB Hits@K)               reference src/utils/ogb_utils.py:83-90       (K = 100)
  _eval_ogbl_citation2 (OGB MRR)            reference src/utils/ogb_utils.py:93
  RegressionMetrics / _eval_pcqm4mv2 (MAE)  reference src/utils/metrics_utils.py:143-189, src/utils/ogb_utils.py:199-204
The reference delegates to `torchmetrics` / `ogb` (not installed here); these are plain NumPy statements of the
published definitions, pinned in tests against scikit-learn and closed-form cases."""
from __future__ import annotations

import numpy as np


def edge_score(task_logits: np.ndarray) -> np.ndarray:
    """y = logit[:,1] - logit[:,0] in fp32 (metrics_utils.py:45-48; also `auc_loss` input, modeling_finetune.py:203-206)."""
    lg = np.asarray(task_logits, np.float32)
    return lg[:, 1] - lg[:, 0]


def accuracy(task_logits: np.ndarray, labels: np.ndarray) -> float:
    return float((np.asarray(task_logits).argmax(-1) == np.asarray(labels)).mean())


def auroc(scores: np.ndarray, labels: np.ndarray) -> float:
    """Area under the ROC curve = Mann-Whitney U statistic with mid-ranks for ties."""
    s = np.asarray(scores, np.float64)
    y = np.asarray(labels).astype(bool)
    n_pos, n_neg = int(y.sum()), int((~y).sum())
    if n_pos == 0 or n_neg == 0:
        return float("nan")
    order = np.argsort(s, kind="mergesort")
    ranks = np.empty(len(s), np.float64)
    ss = s[order]
    i = 0
    while i < len(ss):
        j = i
        while j + 1 < len(ss) and ss[j + 1] == ss[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return float((ranks[y].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))


def hits_at_k(pos_scores: np.ndarray, neg_scores: np.ndarray, k: int = 100) -> float:
    """OGB link-prediction Hits@K (ogb.linkproppred.Evaluator._eval_hits): fraction of positive edges scored above
    the K-th best negative edge; 1.0 when there are fewer than K negatives."""
    pos, neg = np.asarray(pos_scores, np.float64), np.asarray(neg_scores, np.float64)
    if len(neg) < k:
        return 1.0
    kth = np.sort(neg)[-k]
    return float((pos > kth).mean())


def mrr(pos_scores: np.ndarray, neg_scores: np.ndarray) -> float:
    """OGB MRR (ogb.linkproppred.Evaluator._eval_mrr): pos [N], neg [N, n_neg]; mean of 1/rank of the positive among its
    own negatives with OGB's tie handling (average of the optimistic and pessimistic rank)."""
    pos = np.asarray(pos_scores, np.float64)[:, None]
    neg = np.asarray(neg_scores, np.float64)
    optimistic = (neg > pos).sum(1)
    pessimistic = (neg >= pos).sum(1)
    rank = 0.5 * (optimistic + pessimistic) + 1.0
    return float((1.0 / rank).mean())


def mae(pred: np.ndarray, target: np.ndarray) -> float:
    return float(np.abs(np.asarray(pred, np.float64).reshape(-1) - np.asarray(target, np.float64).reshape(-1)).mean())
