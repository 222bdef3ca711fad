

def test_check_batch_raises_like_the_reference_would():
    """GGET_CHECK_INPUTS=1 debug validation: out-of-vocabulary ids / labels and 2-D masks that are not right-padded."""
    import importlib
    import pytest
    import torch
    M = importlib.import_module("graph-gpt_amd.modeling")
    ids = torch.randint(0, 50, (2, 6, 3))
    mask = torch.tensor([[1, 1, 1, 1, 0, 0], [1, 1, 1, 1, 1, 1]])
    lab = torch.full((2, 6, 3), -100)
    lab[0, 1] = 7
    M.check_batch(ids, mask, lab, 50)
    with pytest.raises(IndexError):
        M.check_batch(ids + 48, mask, lab, 50)
    bad = lab.clone(); bad[1, 2, 0] = 50
    with pytest.raises(IndexError):
        M.check_batch(ids, mask, bad, 50)
    with pytest.raises(ValueError):
        M.check_batch(ids, torch.tensor([[1, 0, 1, 1, 0, 0], [1, 1, 1, 1, 1, 1]]), lab, 50)
    M.check_batch(ids, torch.ones(2, 6, 6, dtype=torch.int64), lab, 50)      # 3-D masks are not checked here


def test_elem_drop_keep_twin_rates_and_streams():
    """Python twin of the element dropouts (csrc/common.h:elem_drop_mul): drop rate ~ p, kept elements scaled by 1/(1-p),
    masks differ between streams / layers / seeds and repeat for the same coordinates."""
    import importlib
    import numpy as np
    M = importlib.import_module("graph-gpt_amd.modeling")
    a = M.elem_drop_keep(123, "mlp_act", 0, 512, 256, 0.2)
    assert a.shape == (512, 256) and abs((a == 0).mean() - 0.2) < 0.01
    assert np.allclose(a[a != 0], 1.0 / 0.8)
    assert np.array_equal(a, M.elem_drop_keep(123, "mlp_act", 0, 512, 256, 0.2))
    for other in (M.elem_drop_keep(124, "mlp_act", 0, 512, 256, 0.2), M.elem_drop_keep(123, "mlp_act", 1, 512, 256, 0.2),
                  M.elem_drop_keep(123, "mlp_out", 0, 512, 256, 0.2), M.elem_drop_keep(123, "embed", 0, 512, 256, 0.2),
                  M.elem_drop_keep(123, "head", 0, 512, 256, 0.2)):
        assert abs(((a == 0) & (other == 0)).mean() - 0.04) < 0.01        # independent masks overlap at p^2
    assert np.all(M.elem_drop_keep(5, "embed", 0, 8, 8, 0.0) == 1.0)
