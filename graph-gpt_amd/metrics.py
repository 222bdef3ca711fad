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


# ----------------------------------------------------------------------------- accumulating metric objects of the fine-tune evaluation pass
class SingleLabelClassificationMetrics:
    """Counterpart of the reference class of the same name (src/utils/metrics_utils.py:17-80): per batch `update(logits,
    labels, idx)`; two classes: probability of class 1 feeds AUROC / accuracy, the edge score logit1 - logit0 (fp32) is what
    is kept for the OGB evaluators; more classes: arg-max accuracy.  torchmetrics is replaced by the NumPy statements above."""

    def __init__(self, device=None, num_labels: int = 2, **kwargs):
        self.device, self.num_labels = device, num_labels
        self.auroc = self.acc = None
        self.ls_prob, self.ls_pred, self.ls_labels, self.ls_idx = [], [], [], []

    def update(self, logits, labels, idx):
        import torch
        lg = logits.detach().float()
        if self.num_labels == 2:
            self.ls_prob.append(lg.softmax(dim=-1)[:, 1].cpu())
            y_pred = lg[:, 1] - lg[:, 0]
        else:
            y_pred = torch.argmax(lg, dim=-1)
            labels, idx = labels.reshape(y_pred.shape), idx.reshape(y_pred.shape)
        self.ls_pred.append(y_pred.cpu())
        self.ls_labels.append(labels.detach().cpu())
        self.ls_idx.append(idx.detach().cpu())

    def compute(self, gathered=None):
        """`gathered`: {"y_true", "prob" | "y_pred"} collected from ALL ranks (the reference's torchmetrics objects synchronise
        across ranks inside compute(); here the caller passes what it gathered).  None = this rank's own lists."""
        import torch
        y = torch.hstack(self.ls_labels).numpy() if gathered is None else np.asarray(gathered["y_true"])
        if self.num_labels == 2:
            prob = torch.hstack(self.ls_prob).numpy() if gathered is None else np.asarray(gathered["prob"])
            self.auroc = auroc(prob, y)
            self.acc = float(((prob > 0.5).astype(np.int64) == y).mean())
        else:
            self.auroc = -1
            pred = torch.hstack(self.ls_pred).numpy() if gathered is None else np.asarray(gathered["y_pred"])
            self.acc = float((pred == y).mean())

    def sync_dict(self):
        """what compute() needs from every rank (superset of to_dict for the two-class case: the class-1 probabilities)"""
        import torch
        d = {"y_true": torch.hstack(self.ls_labels), "y_pred": torch.hstack(self.ls_pred)}
        if self.num_labels == 2:
            d["prob"] = torch.hstack(self.ls_prob)
        return d

    def to_dict(self):
        import torch
        return {"y_true": torch.hstack(self.ls_labels), "y_pred": torch.hstack(self.ls_pred), "idx": torch.hstack(self.ls_idx)}

    def results_in_tuple(self):
        return self.auroc, self.acc

    def results_in_dict(self):
        return {"auroc": self.auroc, "acc": self.acc}


class RegressionMetrics:
    """Counterpart of the reference RegressionMetrics (src/utils/metrics_utils.py:143-189): mean absolute / squared error."""

    def __init__(self, device=None, num_labels: int = 1, **kwargs):
        self.device = device
        self.mae = self.mse = None
        self.ls_pred, self.ls_labels, self.ls_idx = [], [], []

    def update(self, logits, labels, idx):
        self.ls_pred.append(logits.detach().float().reshape(-1).cpu())
        self.ls_labels.append(labels.detach().float().reshape(-1).cpu())
        self.ls_idx.append(idx.detach().reshape(-1).cpu())

    def compute(self, gathered=None):
        import torch
        if gathered is None:
            p, y = torch.hstack(self.ls_pred).numpy(), torch.hstack(self.ls_labels).numpy()
        else:
            p, y = np.asarray(gathered["y_pred"]), np.asarray(gathered["y_true"])
        self.mae = mae(p, y)
        self.mse = float(((p.astype(np.float64) - y) ** 2).mean())

    def to_dict(self):
        import torch
        return {"y_true": torch.hstack(self.ls_labels), "y_pred": torch.hstack(self.ls_pred), "idx": torch.hstack(self.ls_idx)}

    def sync_dict(self):
        import torch
        return {"y_true": torch.hstack(self.ls_labels), "y_pred": torch.hstack(self.ls_pred)}

    def results_in_tuple(self):
        return self.mse, self.mae       # the reference's order (metrics_utils.py:184-185)

    def results_in_dict(self):
        return {"mae": self.mae, "mse": self.mse}


def get_metrics(metric_type: str, device=None, num_labels: int = 2):
    """reference `get_metrics` registry (metrics_utils.py:11-13) for the two problem types of the BASELINE configs."""
    if metric_type == "single_label_classification":
        return SingleLabelClassificationMetrics(device, num_labels=num_labels)
    if metric_type == "regression":
        return RegressionMetrics(device, num_labels=num_labels)
    raise NotImplementedError(f"metric_type={metric_type!r} (multi-label / sequence metrics are outside the hot-path scope)")


def evaluate_ogb(dataset_name: str, input_dict):
    """reference `evaluate_ogb` for the datasets of the BASELINE configs (src/utils/ogb_utils.py:83-90 ogbl-ppa Hits@100 over
    positive / negative edges split by label; :199-204 PCQM4Mv2 MAE).  None for a dataset this package has no evaluator for."""
    y_true, y_pred = np.asarray(input_dict["y_true"]), np.asarray(input_dict["y_pred"], np.float64)
    if dataset_name == "ogbl-ppa":
        return {"hits@100": hits_at_k(y_pred[y_true == 1], y_pred[y_true == 0], 100)}
    if dataset_name == "PCQM4Mv2":
        return {"mae": mae(y_pred, y_true)}
    return None
