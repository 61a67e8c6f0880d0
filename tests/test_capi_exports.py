"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/tendrils_hip.h
declares, and fails loudly (no CPU fallback) when there is no GPU."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "tendrils_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(th_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    from tendrils_amd import _capi
    if not os.path.exists(_capi.LIB_PATH):
        g.build()
    return _capi.load()


def test_every_declared_symbol_is_exported_and_bound(lib):
    from tendrils_amd import _capi
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "library does not export %s" % s
    assert sorted(_capi.PROTOTYPES) == syms, "ctypes binding and header disagree"
    assert lib.th_abi_version() == 14


def test_one_rocm_runtime_in_the_process(lib):
    """torch bundles its own ROCm runtime; after _capi.load() the process must hold exactly one HIP runtime
    (the loader imports torch first and refuses two)."""
    from tendrils_amd import _capi
    assert len(_capi._mapped("libamdhip64")) == 1, _capi._mapped("libamdhip64")


def test_product_never_touches_the_oracle():
    """The oracle is test infrastructure: nothing under tendrils_amd/ or include/ may name it."""
    for base in ("tendrils_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cc", ".js", "Makefile")):
                    txt = open(os.path.join(dirpath, f), errors="replace").read()
                    assert "oracle" not in txt.lower(), "%s mentions the oracle" % os.path.join(dirpath, f)


def test_no_gpu_is_a_loud_error(lib):
    import ctypes as C
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import tendrils_amd as ta
    from tendrils_amd.tendrils import View
    t = ta.Tendrils(View(8, 8))
    t.resize()
    with pytest.raises(ta.TendrilsHipError):
        t.setup(8)
