"""A14: metrics from task_logits, pinned against scikit-learn and closed forms, on the reference's own fixture logits."""
import importlib

import numpy as np
from sklearn.metrics import mean_absolute_error, roc_auc_score

from _util import load_case

M = importlib.import_module("graph-gpt_amd.metrics")


def test_auroc_matches_sklearn_with_ties():
    rng = np.random.RandomState(0)
    y = rng.randint(0, 2, 500)
    s = np.round(rng.randn(500) + y, 1)  # rounding creates ties
    assert abs(M.auroc(s, y) - roc_auc_score(y, s)) < 1e-12


def test_edge_score_and_accuracy_on_reference_logits():
    z, spec, state, batch = load_case("ft_tiny_f4")
    lg = z["logits"]
    y = batch["task_labels"]
    sc = M.edge_score(lg)
    np.testing.assert_allclose(sc, lg[:, 1] - lg[:, 0])
    assert M.accuracy(lg, y) == float(((sc > 0).astype(int) == y).mean())


def test_hits_and_mrr_closed_form():
    neg = np.arange(1000, dtype=np.float64)          # K-th best negative (K=100) is 900
    pos = np.array([899.5, 900.0, 900.5, 2000.0])
    assert M.hits_at_k(pos, neg, 100) == 0.5
    assert M.hits_at_k(pos, neg[:50], 100) == 1.0
    p = np.array([5.0, 1.0])
    n = np.array([[1.0, 2.0, 3.0], [1.0, 2.0, 3.0]])  # ranks: 1 ; second ties with one negative -> (2+3)/2 + ... 
    # sample 0: no negative above -> rank 1 ; sample 1: 2 strictly above, 3 >= -> rank 0.5*(2+3)+1 = 3.5
    assert abs(M.mrr(p, n) - 0.5 * (1.0 + 1.0 / 3.5)) < 1e-12


def test_mae_on_reference_regression_logits():
    z, spec, state, batch = load_case("ft_tiny_reg")
    assert abs(M.mae(z["logits"], batch["task_labels"]) - mean_absolute_error(batch["task_labels"], z["logits"].reshape(-1))) < 1e-6
    # the reference's L1 task loss IS the MAE of the pooled logits (modeling_finetune.py:183-197)
    assert abs(M.mae(z["logits"], batch["task_labels"]) - float(z["loss"])) < 1e-5
