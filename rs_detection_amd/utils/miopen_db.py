"""Packaged MIOpen find / tuning records for the convolutions of the shipped configs (gfx950, the MIOpen of this image).

PyTorch-ROCm asks MIOpen for a convolution solver in "immediate" mode; with an empty user database MIOpen answers
from its heuristics, with the records below it answers with the solvers an exhaustive `miopenFind*` + tuning run
picked on an MI355X (S2ANet-R50-FPN step: fp32 63.3 -> 60.4 ms, bf16 44.6 -> 34.5 ms; Oriented R-CNN VAN-B3: usable
with MIOPEN_FIND_MODE=FAST, 347 -> 150 ms).  Producing those records takes ~10 minutes of
GPU time per model (`RSDET_CUDNN_BENCHMARK=1 MIOPEN_FIND_MODE=NORMAL python bench.py`), which no fresh box should
pay at start-up, so the two text files (`*.ufdb.txt` find results, `*.udb.txt` tuned kernel parameters; 360 KB) ship
with the package and are copied into a writable per-user directory that `MIOPEN_USER_DB_PATH` then points to.

The file names carry the GPU and the MIOpen build they were made with; any other MIOpen ignores them and behaves as
before.  Opt out with `RSDET_NO_MIOPEN_DB=1`; a `MIOPEN_USER_DB_PATH` set by the user always wins.
"""
import glob
import os
import shutil
import tempfile

_PKG_DB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "miopen_db")


def use_packaged_miopen_db():
    """Call before the first convolution of the process (MIOpen reads the variable when its handle is created).
    Returns the directory in use, or None when nothing was changed."""
    if os.environ.get("RSDET_NO_MIOPEN_DB", "0") == "1" or "MIOPEN_USER_DB_PATH" in os.environ:
        return None
    files = sorted(glob.glob(os.path.join(_PKG_DB, "*db.txt")))
    if not files:
        return None
    uid = os.getuid() if hasattr(os, "getuid") else 0
    dst = os.path.join(tempfile.gettempdir(), "rsdet_miopen_db_%d" % uid)
    try:
        os.makedirs(dst, exist_ok=True)
        for f in files:  # MIOpen appends to its user database: never hand it the packaged originals
            t = os.path.join(dst, os.path.basename(f))
            if not os.path.exists(t) or os.path.getsize(t) < os.path.getsize(f):
                tmp = "%s.%d.tmp" % (t, os.getpid())
                shutil.copyfile(f, tmp)
                os.replace(tmp, t)  # atomic: ranks of one node start together
    except OSError:
        return None
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    return dst
