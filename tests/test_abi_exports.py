"""CPU-side checks of the C-ABI library: it loads without a GPU, exports every symbol the header
declares, and refuses to create a plan when no GPU is present (no CPU fallback)."""
import os
import re

import pytest


def test_library_exports_header_symbols():
    import mifwi
    lib = mifwi.load_library()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    header = open(os.path.join(root, "include", "mifwi.h")).read()
    declared = set(re.findall(r"\b(mifwi_[a-z0-9_]+)\s*\(", header))
    declared -= {"mifwi_plan", "mifwi_config", "mifwi_stats"}
    assert declared, "no declarations found"
    assert declared == set(mifwi.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mifwi_version() == 103


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import mifwi
    with pytest.raises(mifwi.MifwiError) as e:
        mifwi.Plan((20, 20), (10., 10.), (0., 0.), 4, 20, 1.0, 1, 3)
    assert e.value.code == -5
