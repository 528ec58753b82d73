"""Checkpoint cache shared WITH the reference: ``s3prl/util/download.py:26-42,186-208`` stores a URL at
``~/.cache/s3prl/download/<sha256(url)>.<basename(url)>`` (``get_dir`` / ``set_dir`` move the directory) behind a
``<file>.lock`` FileLock.  ``urls_to_filepaths`` resolves a URL to that very file, so a checkpoint the reference already
downloaded is found by the MI355X hub entries and one fetched here is found by the reference.  When the file is absent a
download is attempted (``urllib``, temp file + atomic move, the reference's lock file); a machine without network gets a
``RuntimeError`` naming the cache path to drop the file at."""

import hashlib
import os
import shutil
import tempfile
import time
from pathlib import Path

_download_dir = Path.home() / ".cache" / "s3prl" / "download"

NEW_ENOUGH_SECS = 2.0
TIMEOUT_SECS = float(os.environ.get("S3PRL_AMD_DOWNLOAD_TIMEOUT", "30"))


def get_dir() -> Path:
    _download_dir.mkdir(exist_ok=True, parents=True)
    return _download_dir


def set_dir(d) -> None:
    global _download_dir
    _download_dir = Path(d)


def cache_path(url: str) -> Path:
    """The reference's naming rule (util/download.py:198-202)."""
    assert isinstance(url, str)
    return get_dir() / f"{hashlib.sha256(url.encode()).hexdigest()}.{Path(url).name}"


def _fetch(url: str, dst: Path) -> None:
    from urllib.request import Request, urlopen

    tmp = tempfile.NamedTemporaryFile(delete=False, dir=str(dst.parent))
    try:
        with urlopen(Request(url, headers={"User-Agent": "torch.hub"}), timeout=TIMEOUT_SECS) as u:
            shutil.copyfileobj(u, tmp, 1 << 20)
        tmp.close()
        shutil.move(tmp.name, str(dst))
    finally:
        tmp.close()
        if os.path.exists(tmp.name):
            os.remove(tmp.name)


def _lock(path: Path):
    try:
        from filelock import FileLock  # the reference's lock (same file name: the two cooperate)

        return FileLock(str(path) + ".lock")
    except ImportError:  # pragma: no cover
        import contextlib

        return contextlib.nullcontext()


def urls_to_filepaths(*urls, refresh: bool = False, download: bool = True):
    """Same contract as the reference's ``_urls_to_filepaths`` (util/download.py:186-208): one path per URL."""

    def one(url):
        path = cache_path(url)
        if download:
            with _lock(path):
                stale = refresh and path.is_file() and (time.time() - os.path.getmtime(path)) > NEW_ENOUGH_SECS
                if not path.is_file() or stale:
                    try:
                        _fetch(url, path)
                    except Exception as e:  # no network: keep a stale copy, else say where the file belongs
                        if not path.is_file():
                            raise RuntimeError(
                                f"cannot fetch {url} ({type(e).__name__}: {e}); this machine seems to have no network — "
                                f"place the checkpoint at {path} (the cache file the reference's s3prl.util.download "
                                "uses) or pass a local path to the *_local entry") from e
        return str(path.resolve())

    paths = [one(u) for u in urls]
    return paths if len(paths) > 1 else paths[0]


_urls_to_filepaths = urls_to_filepaths
