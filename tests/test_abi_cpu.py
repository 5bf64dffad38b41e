"""CPU: the C-ABI library builds, loads and exports every symbol include/rsdet.h declares
(no compute call without a GPU), and the product path refuses to run without it / on CPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rsdet.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rsdet_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    from rs_detection_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with `python -c 'import __graft_entry__ as g; g.build()'`"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "missing export %s" % s
    # the ctypes table covers the header exactly
    assert sorted(_lib.SIGNATURES) == syms
    _lib.load()
    lib.rsdet_abi_version.restype = ctypes.c_int
    assert lib.rsdet_abi_version() >= 1
    lib.rsdet_nms_rotated_ws_size.restype = ctypes.c_size_t
    assert lib.rsdet_nms_rotated_ws_size(0) == 0 and lib.rsdet_nms_rotated_ws_size(65) >= 65 * 2 * 8


def test_header_cites_reference_for_every_entry():
    text = open(os.path.join(ROOT, "include", "rsdet.h")).read()
    assert text.count("Replaces") >= 8 and "ops/box_iou_rotated.py" in text and "ops/nms_rotated.py" in text


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from rs_detection_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.RsdetError, match="no CPU fallback"):
        _lib.load()


def test_ops_reject_cpu_tensors():
    from rs_detection_amd import ops, _lib
    with pytest.raises(_lib.RsdetError):
        ops.box_iou_rotated(torch.zeros(2, 5), torch.zeros(3, 5))
    with pytest.raises(_lib.RsdetError):
        ops.delta2bbox_rotated(torch.zeros(2, 5), torch.zeros(2, 5))
    with pytest.raises(NotImplementedError):  # dcn_v1.py:588-589
        ops.deform_conv(torch.zeros(1, 4, 5, 5), torch.zeros(1, 18, 5, 5), torch.zeros(4, 4, 3, 3), 1, 1, 1, 2, 1)


def test_product_never_imports_oracle():
    """Grep guard: nothing under rs_detection_amd/ (nor bench's timed region) references oracle/."""
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "rs_detection_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h")):
                t = open(os.path.join(dp, f)).read()
                if re.search(r"^\s*(import|from)\s+oracle\b", t, flags=re.M) or "librsdet_oracle" in t:
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
