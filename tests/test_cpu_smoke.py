"""CPU-side checks that run without a GPU: the C-ABI library loads and exports every declared symbol."""
import os
import re

import numpy as np


def test_library_exports_every_declared_symbol():
    from svin_amd import estimator
    lib = estimator.load_library()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "svin_ba.h")).read()
    declared = sorted(set(re.findall(r"\b(svin_(?:ba|host)_[a-zA-Z0-9_]+)\s*\(", header)))
    assert len(declared) > 40
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert set(declared) == set(estimator.EXPORTS)


def test_library_exports_every_pose_graph_symbol():
    from svin_amd import estimator, posegraph
    lib = estimator.load_library()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "svin_pg.h")).read()
    declared = sorted(set(re.findall(r"\b(svin_pg_[a-zA-Z0-9_]+)\s*\(", header)))
    assert len(declared) >= 10
    missing = [n for n in declared if not hasattr(lib, n)]
    assert not missing, missing
    assert set(declared) == set(posegraph.PG_EXPORTS)


def test_pose_graph_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    from svin_amd import posegraph
    try:
        posegraph.PoseGraph(0)
    except RuntimeError as e:
        assert "no CPU fallback" in str(e) or "HIP" in str(e)
    else:
        raise AssertionError("creating a pose graph without a GPU must fail")


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        return
    from svin_amd import estimator
    try:
        estimator.Estimator(0)
    except RuntimeError as e:
        assert "no CPU fallback" in str(e) or "HIP" in str(e)
    else:
        raise AssertionError("creating an estimator without a GPU must fail")
