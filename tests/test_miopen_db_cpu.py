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
    monkeypatch.setattr(miopen_db.tempfile, "gettempdir", lambda: str(tmp_path))
    d = miopen_db.use_packaged_miopen_db()
    assert d and os.environ["MIOPEN_USER_DB_PATH"] == d
    assert sorted(os.listdir(d)) == sorted(os.path.basename(f) for f in glob.glob(os.path.join(miopen_db._PKG_DB, "*db.txt")))
    # a path chosen by the user wins; the opt-out switch leaves the environment alone
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", "/somewhere/else")
    assert miopen_db.use_packaged_miopen_db() is None and os.environ["MIOPEN_USER_DB_PATH"] == "/somewhere/else"
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    monkeypatch.setenv("RSDET_NO_MIOPEN_DB", "1")
    assert miopen_db.use_packaged_miopen_db() is None and "MIOPEN_USER_DB_PATH" not in os.environ
