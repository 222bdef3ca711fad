import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests must not silently pass on a CPU-only box
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """Every error the GPU parity tests measured, next to the tolerance it was held to (tests/_util.py:record_error) ->
    gpurun_out/parity_errors.json (copied to profiles/ per round)."""
    try:
        import json
        import math
        from _util import _ERRORS
    except Exception:
        return
    if not _ERRORS:
        return
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    clean = {c: {q: {k: (None if isinstance(v, float) and math.isnan(v) else v) for k, v in d.items()} for q, d in qs.items()}
             for c, qs in sorted(_ERRORS.items())}
    path = os.path.join(out, "parity_errors.json")
    old = {}
    if os.path.exists(path) and os.environ.get("GGET_PARITY_APPEND"):
        try:
            old = json.load(open(path))
        except Exception:
            old = {}
    old.update(clean)
    with open(path, "w") as fh:
        json.dump(old, fh, indent=1, sort_keys=True)
