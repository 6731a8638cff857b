"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/rayjoin_amd.h declares (no compute calls without a GPU), and fails loudly without one."""
import os
import re

import pytest

from rayjoin_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "rayjoin_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rj_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported_and_bound():
    L = _capi.load()
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), "librayjoin_amd.so does not export %s" % n
        assert n in _capi.SYMBOLS, "%s is not bound in rayjoin_amd/_capi.py" % n
    assert set(_capi.SYMBOLS) == set(names)
    assert b"gfx950" in L.rj_version()


def test_record_layouts():
    assert _capi.XSECT_DTYPE.itemsize == 48  # dev::Intersection<int64_t>, lsi.h:21-25


def test_no_gpu_is_a_loud_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_capi.RayJoinError):
        _capi.Handle(0)


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under rayjoin_amd/ (or the diagnostics in tools/)
    may reference it."""
    import itertools
    walks = itertools.chain(os.walk(os.path.join(ROOT, "rayjoin_amd")), os.walk(os.path.join(ROOT, "tools")))
    for dp, _, fns in walks:
        for fn in fns:
            if fn.endswith((".py", ".hip", ".h", ".cc", ".cpp", ".sh")) or fn == "Makefile":
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                for needle in ("rjoracle", "rj_oracle", "import oracle", "from oracle", "librj_oracle",
                               "liblsi_ref", "_ref/"):
                    assert needle not in txt, "%s references the oracle (%s)" % (os.path.join(dp, fn), needle)
                assert not re.search(r'#include\s*[<"][^>"]*oracle', txt), os.path.join(dp, fn)
