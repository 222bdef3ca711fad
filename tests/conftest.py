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


@pytest.fixture(autouse=True)
def _reset_process_wide_kernel_menu(request):
    """gget_debug_set switches are process-wide (the data-parallel tests attach communicators, which selects the LDS-headroom launch menu
    - key 2 - for the rest of the process): every GPU test starts from the single-GPU defaults, whatever ran before it."""
    if "gpu" in request.keywords:
        try:
            import importlib
            L = importlib.import_module("graph-gpt_amd._lib")
            lib = L.load()
            for key, val in ((2, 1), (10, 0), (11, 0), (1, 0), (8, 0), (13, 1), (14, 1), (15, 0)):
                lib.gget_debug_set(key, val)
        except Exception:
            pass
    yield


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
