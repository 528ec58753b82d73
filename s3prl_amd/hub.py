"""Registry in the style of ``s3prl/hub.py`` (which star-imports every ``upstream/*/hubconf.py``, :1-37) plus
``register_into_s3prl()`` that installs our callables on an importable ``s3prl.hub`` so existing s3prl code
(``getattr(hub, name)(ckpt=...)``, downstream/runner.py:149-153; ``S3PRLUpstream(name)``, nn/upstream.py:113-117)
picks up the MI355X path without edits to the reference tree."""

from .upstream.hubert.hubconf import *  # noqa: F401,F403
from .upstream.wav2vec2.hubconf import *  # noqa: F401,F403
from .upstream.wavlm.hubconf import *  # noqa: F401,F403
from .upstream.unispeech_sat.hubconf import *  # noqa: F401,F403
from .upstream.baseline.hubconf import *  # noqa: F401,F403


def options(only_registered_ckpt: bool = False):
    """Like ``s3prl.hub.options`` (hub.py:40-54): the public callables of this module."""
    names = []
    for name, value in globals().items():
        if name.startswith("_") or not callable(value) or name in ("options", "register_into_s3prl"):
            continue
        if only_registered_ckpt and (name.endswith("_local") or name.endswith("_url") or name.endswith("_custom")):
            continue
        names.append(name)
    return sorted(names)


def register_into_s3prl(prefix: str = "", override: bool = True):
    """setattr our entries on ``s3prl.hub`` (``prefix="amd_"`` keeps the reference entries alongside)."""
    import importlib

    hub = importlib.import_module("s3prl.hub")
    installed = []
    for name in options():
        target = prefix + name
        if hasattr(hub, target) and not override:
            continue
        setattr(hub, target, globals()[name])
        installed.append(target)
    return installed
