"""Registry in the style of ``s3prl/hub.py`` (which star-imports every ``upstream/*/hubconf.py``, :1-37) plus
``register_into_s3prl()`` that installs our callables on an importable ``s3prl.hub`` so existing s3prl code
(``getattr(hub, name)(ckpt=...)``, downstream/runner.py:149-153; ``S3PRLUpstream(name)``, nn/upstream.py:113-117)
picks up the MI355X path without edits to the reference tree."""

from .upstream.hubert.hubconf import *  # noqa: F401,F403
from .upstream.wav2vec2.hubconf import *  # noqa: F401,F403
from .upstream.wavlm.hubconf import *  # noqa: F401,F403
from .upstream.unispeech_sat.hubconf import *  # noqa: F401,F403
from .upstream.distiller.hubconf import *  # noqa: F401,F403
from .upstream.data2vec.hubconf import *  # noqa: F401,F403
from .upstream.multires_hubert.hubconf import *  # noqa: F401,F403
from .upstream.hf_hubert.hubconf import *  # noqa: F401,F403
from .upstream.hf_wav2vec2.hubconf import *  # noqa: F401,F403
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


def register_into_s3prl(prefix: str = "", override: bool = False):
    """setattr our entries on ``s3prl.hub``.  By default existing reference entries are left alone (only names the
    reference does not define are added); ``prefix="amd_"`` installs every entry alongside the reference's;
    ``override=True`` replaces the reference entries of the same name — note that the URL-backed ones (``hubert``,
    ``wavlm_base_plus`` …) then need ``ckpt=`` because this build never downloads."""
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
