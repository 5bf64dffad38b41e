"""The packaged MIOpen records (rs_detection_amd/miopen_db) and the helper that points MIOPEN_USER_DB_PATH at a copy."""
import glob
import os

from rs_detection_amd.utils import miopen_db


def test_packaged_db_is_text_and_small():
    files = glob.glob(os.path.join(miopen_db._PKG_DB, "*db.txt"))
    assert files, "no packaged MIOpen records"
    for f in files:
        assert os.path.basename(f).startswith("gfx950")          # named after the GPU + MIOpen build they belong to
        assert os.path.getsize(f) < 2 << 20
        head = open(f).read(4096)
        assert "=" in head and "Conv" in head                    # "<problem>=<solver>:<...>" records


def test_helper_copies_and_respects_the_user(monkeypatch, tmp_path):
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    monkeypatch.delenv("RSDET_NO_MIOPEN_DB", raising=False)
    monkeypatch.delenv("RSDET_MIOPEN_DB_IN_USE", raising=False)
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path / "cache"))      # (never this user's real working copy)
    d = miopen_db.use_packaged_miopen_db()
    assert d and os.environ["MIOPEN_USER_DB_PATH"] == d
    pkg = sorted(os.path.basename(f) for f in glob.glob(os.path.join(miopen_db._PKG_DB, "*db.txt")))
    assert sorted(f for f in os.listdir(d) if not f.endswith(".src")) == pkg
    assert miopen_db.packaged_records_match() in (True, False)        # (True only with this image's MIOpen build)
    # a working copy that MIOpen appended to is kept; one that came from OTHER packaged records (stale stamp) is replaced
    t = os.path.join(d, pkg[0])
    with open(t, "a") as fh:
        fh.write("appended=by:MIOpen\n")
    grown = os.path.getsize(t)
    assert miopen_db.use_packaged_miopen_db() == d and os.path.getsize(t) == grown
    with open(t + ".src", "w") as fh:
        fh.write("0:stale\n")
    assert miopen_db.use_packaged_miopen_db() == d
    assert os.path.getsize(t) == os.path.getsize(os.path.join(miopen_db._PKG_DB, pkg[0]))
    # records that are not where MIOpen will look do not count as "in use"
    monkeypatch.setenv("RSDET_MIOPEN_DB_IN_USE", str(tmp_path / "elsewhere"))
    assert miopen_db.packaged_records_match() is False
    monkeypatch.delenv("RSDET_MIOPEN_DB_IN_USE")
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    # a path chosen by the user wins; the opt-out switch leaves the environment alone
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", "/somewhere/else")
    assert miopen_db.use_packaged_miopen_db() is None and os.environ["MIOPEN_USER_DB_PATH"] == "/somewhere/else"
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    monkeypatch.delenv("RSDET_MIOPEN_DB_IN_USE", raising=False)
    monkeypatch.setenv("RSDET_NO_MIOPEN_DB", "1")
    assert miopen_db.use_packaged_miopen_db() is None and "MIOPEN_USER_DB_PATH" not in os.environ
