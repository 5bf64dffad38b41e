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


def _private_dir():
    """A directory only this user can write: ``$XDG_CACHE_HOME|~/.cache/rsdet/miopen_db`` (mode 0700, owned by us, not
    a symlink); a fresh ``mkdtemp`` when the home directory is not writable.  A predictable name under /tmp would let
    another local user pre-create or symlink it and choose the solver records MIOpen loads (or where it appends)."""
    base = os.environ.get("XDG_CACHE_HOME") or os.path.join(os.path.expanduser("~"), ".cache")
    dst = os.path.join(base, "rsdet", "miopen_db")
    try:
        os.makedirs(dst, mode=0o700, exist_ok=True)
        st = os.lstat(dst)
        import stat
        if stat.S_ISLNK(st.st_mode) or not stat.S_ISDIR(st.st_mode):
            raise OSError("not a plain directory")
        if hasattr(os, "getuid") and st.st_uid != os.getuid():
            raise OSError("owned by another user")
        if st.st_mode & 0o022:
            os.chmod(dst, 0o700)
        return dst
    except OSError:
        try:
            return tempfile.mkdtemp(prefix="rsdet_miopen_db_")   # 0700, unpredictable name
        except OSError:
            return None


def _source_stamp(path):
    import hashlib
    with open(path, "rb") as fh:
        data = fh.read()
    return "%d:%s" % (len(data), hashlib.sha1(data).hexdigest())


def use_packaged_miopen_db():
    """Call before the first convolution of the process (MIOpen reads the variable when its handle is created).
    Returns the directory in use, or None when nothing was changed."""
    if os.environ.get("RSDET_NO_MIOPEN_DB", "0") == "1":
        return None
    user = os.environ.get("MIOPEN_USER_DB_PATH")
    if user is not None and user != os.environ.get("RSDET_MIOPEN_DB_IN_USE"):
        return None          # the user's own directory wins (a second call of ours just refreshes our copy)
    files = sorted(glob.glob(os.path.join(_PKG_DB, "*db.txt")))
    if not files:
        return None
    dst = _private_dir()
    if dst is None:
        return None
    try:
        for f in files:  # MIOpen appends to its user database: never hand it the packaged originals
            t = os.path.join(dst, os.path.basename(f))
            # the working copy grows (MIOpen appends), so its size says nothing about WHICH packaged file it started
            # from: a stamp beside it names the source (size + digest); another stamp = a stale copy, replaced
            stamp = _source_stamp(f)
            try:
                with open(t + ".src") as fh:
                    have = fh.read().strip()
            except OSError:
                have = None
            if not os.path.exists(t) or have != stamp:
                tmp = "%s.%d.tmp" % (t, os.getpid())
                shutil.copyfile(f, tmp)
                os.replace(tmp, t)  # atomic: ranks of one node start together
                with open(tmp, "w") as fh:
                    fh.write(stamp + "\n")
                os.replace(tmp, t + ".src")
    except OSError:
        return None
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    os.environ["RSDET_MIOPEN_DB_IN_USE"] = dst     # child processes (bench.py --gpus N, loader workers) inherit both
    return dst


def packaged_records_match():
    """True when the packaged record files were made by the MIOpen this process runs (same major.minor.patch -- the
    version is part of the file name MIOpen looks for) and will be used.  The fp32 channels_last step depends on them:
    on heuristics MIOpen picks a 7 x slower solver set for that layout (DESIGN R3.6), so callers that default to
    channels_last for fp32 fall back to NCHW when this is False."""
    import re
    if os.environ.get("RSDET_NO_MIOPEN_DB", "0") == "1":
        return False
    files = glob.glob(os.path.join(_PKG_DB, "*.ufdb.txt"))
    if not files:
        return False
    try:
        import torch
        v = int(torch.backends.cudnn.version() or 0)
    except Exception:
        return False
    want = "%d_%d_%d_" % (v // 1000000, (v // 1000) % 1000, v % 1000)
    user = os.environ.get("MIOPEN_USER_DB_PATH")
    mine = os.environ.get("RSDET_MIOPEN_DB_IN_USE")      # set by use_packaged_miopen_db(), here or in a parent process
    if user is None:                                     # not called yet in this process (or the variable was dropped
        mine = use_packaged_miopen_db()                  # since): do it now (idempotent)
        user = os.environ.get("MIOPEN_USER_DB_PATH")
    if user is None or mine is None or os.path.realpath(user) != os.path.realpath(mine):
        return False                                     # the copy failed, or the user points MIOpen elsewhere
    match = [f for f in files if re.search(r"\.HIP\." + re.escape(want), os.path.basename(f))]
    # ... and the directory MIOpen will read really holds them
    return bool(match) and all(os.path.exists(os.path.join(mine, os.path.basename(f))) for f in match)
