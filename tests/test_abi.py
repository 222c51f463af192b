"""CPU tests of the drop-in boundary: the C-ABI library loads without a GPU and
exports every symbol include/ggc.h declares; the ctypes table matches the header."""
import ctypes
import re
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HEADER = (ROOT / "include" / "ggc.h").read_text()


def declared_functions():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(ggc_[a-z0-9_]+)\s*\(", body)))


def test_header_declares_the_whole_hot_path():
    names = declared_functions()
    for must in ("ggc_preprocess", "ggc_slic", "ggc_graph_count", "ggc_graph_fill", "ggc_resgcn_forward",
                 "ggc_gcn_aggregate", "ggc_refine_trimap", "ggc_seed_from_prior", "ggc_grabcut",
                 "ggc_clean_mask", "ggc_compose_outputs"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from gcn_grabcut import _native
    lib = _native.load_library()          # no compute call: works without a GPU
    for name in declared_functions():
        assert hasattr(lib, name), f"{name} declared in include/ggc.h but not exported"
    assert lib.ggc_version() == int(re.search(r"#define GGC_VERSION (\d+)", HEADER).group(1))


def test_ctypes_table_covers_the_header():
    from gcn_grabcut import _native
    assert sorted(_native.SIGNATURES) == declared_functions()


def test_argument_counts_match_header():
    from gcn_grabcut import _native
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    for name, argtypes in _native.SIGNATURES.items():
        m = re.search(r"\b" + name + r"\s*\(([^;]*?)\)\s*;", body, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else params.count(",") + 1
        assert n == len(argtypes), (name, n, len(argtypes))


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from gcn_grabcut import _native
    with pytest.raises(_native.GGCError):
        _native.Context(0)


def test_product_never_imports_the_oracle():
    pkg = ROOT / "gcn-grabcut_amd"
    for f in list(pkg.rglob("*.py")) + list(pkg.rglob("*.hip")) + list(pkg.rglob("*.h")):
        text = f.read_text()
        assert "oracle" not in text.replace("the oracle", "").replace("CPU oracle", "") or "import" not in text.split("oracle")[0][-40:], f
        assert "ggo_" not in text, f
        assert "libggc_oracle" not in text, f
