"""Factories for the released-checkpoint hub names.  Every reference hubconf registers its released models as thin
callables that set ``kwargs["ckpt"]`` to a URL and call the family's ``*_custom`` / ``*_url`` entry
(hubert/hubconf.py:85-156, wav2vec2/hubconf.py:84-270, wavlm/hubconf.py:45-79, ...).  The three parameter lists the
reference uses are reproduced exactly (a CPU test compares ``inspect.signature`` against the real ``s3prl.hub``):

* ``(refresh=False, legacy=False, **kwargs)`` — fairseq families; ``legacy=True`` selects the ORIGINAL fairseq file;
* ``(refresh=False, **kwds)``                — converted-only HuBERT variants;
* ``(refresh=False, *args, **kwargs)``       — WavLM / UniSpeech-SAT / data2vec / DistilHuBERT.
"""


def with_legacy(name, target, url, legacy_url, doc=""):
    def entry(refresh=False, legacy=False, **kwargs):
        kwargs["ckpt"] = legacy_url if legacy else url
        return target(refresh=refresh, legacy=legacy, **kwargs)

    entry.__name__ = entry.__qualname__ = name
    entry.__doc__ = doc or f"released checkpoint {url}"
    entry.url, entry.legacy_url = url, legacy_url
    return entry


def converted_only(name, target, url, doc="", kw="kwds"):
    """``kw``: the reference spells the catch-all ``**kwds`` in hubert/hubconf.py and ``**kwargs`` in multires_hubert/."""
    if kw == "kwds":
        def entry(refresh=False, **kwds):
            kwds["ckpt"] = url
            return target(refresh=refresh, **kwds)
    else:
        def entry(refresh=False, **kwargs):
            kwargs["ckpt"] = url
            return target(refresh=refresh, **kwargs)

    entry.__name__ = entry.__qualname__ = name
    entry.__doc__ = doc or f"released checkpoint {url}"
    entry.url = url
    return entry


def positional(name, target, url, doc=""):
    def entry(refresh=False, *args, **kwargs):
        kwargs["ckpt"] = url
        return target(refresh=refresh, *args, **kwargs)

    entry.__name__ = entry.__qualname__ = name
    entry.__doc__ = doc or f"released checkpoint {url}"
    entry.url = url
    return entry


def alias(name, get_target, doc=""):
    """``<family>(refresh=False, *args, **kwargs)``: the family's default released model (looked up late)."""

    def entry(refresh=False, *args, **kwargs):
        return get_target()(refresh=refresh, *args, **kwargs)

    entry.__name__ = entry.__qualname__ = name
    entry.__doc__ = doc
    return entry


def unsupported(name, reason, signature="legacy"):
    def raise_(*a, **k):
        raise NotImplementedError(f"{name}: {reason}")

    if signature == "legacy":
        def entry(refresh=False, legacy=False, **kwargs):
            raise_()
    else:
        def entry(refresh=False, *args, **kwargs):
            raise_()
    entry.__name__ = entry.__qualname__ = name
    entry.__doc__ = f"registered by the reference; not built here: {reason}"
    return entry
