"""CPU tests (-m "not gpu"): the C-ABI library builds for gfx950, loads without a GPU and exports every
symbol include/nsk.h declares; the product path fails loudly (no CPU fallback) when no MI355X is present."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "nsk.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(nsk_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    import nice_slam_cpp_amd as pkg
    pkg.build()
    L = pkg.nsk.lib()
    decl = _declared()
    assert len(decl) >= 30
    for s in decl:
        assert hasattr(L, s), "libnsk.so does not export %s" % s
    assert sorted(pkg.nsk.SYMBOLS) == decl
    assert L.nsk_version() >= 100


def test_param_counts_match_reference_layers():
    import nice_slam_cpp_amd as pkg
    L = pkg.nsk.lib()
    # coarse 6337, middle 15800, fine 20920, color 15899
    assert [L.nsk_decoder_param_count(i) for i in range(4)] == [6337, 15800, 20920, 15899]
    assert L.nsk_decoder_param_count(7) == 0


def test_no_cpu_fallback():
    import torch
    import nice_slam_cpp_amd as pkg
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.NskError):
        pkg.Context(0)


def test_product_does_not_touch_oracle():
    """the product package must not import, link or execute anything under oracle/"""
    pkg_dir = os.path.join(ROOT, "nice-slam-cpp_amd")
    for dp, _, fns in os.walk(pkg_dir):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cpp", ".hpp", "Makefile")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                # no import / include / link / dlopen of anything under oracle/ (comments may mention the word)
                for pat in (r"\bimport\s+oracle", r"\bfrom\s+oracle", r"oracle/[A-Za-z_]+\.(so|py|h|c)\b(?! make_layout)", r"libnso", r"\bnso_[a-z]"):
                    assert not re.search(pat, txt), (os.path.join(dp, fn), pat)
