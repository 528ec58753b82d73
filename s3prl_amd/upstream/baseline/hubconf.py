"""hub entries of the baseline upstream in the reference's naming (s3prl/upstream/baseline/hubconf.py:19-60);
only the kaldi-fbank configurations are implemented on the MI355X path."""

import os

from .expert import UpstreamExpert as _UpstreamExpert


def baseline_local(model_config, *args, **kwargs):
    assert os.path.isfile(model_config)
    return _UpstreamExpert(model_config, *args, **kwargs)


def baseline(*args, **kwargs):
    return fbank(*args, **kwargs)


def fbank(*args, **kwargs):
    kwargs["model_config"] = os.path.join(os.path.dirname(__file__), "fbank.yaml")
    return baseline_local(*args, **kwargs)


def fbank_no_cmvn(*args, **kwargs):
    kwargs["model_config"] = os.path.join(os.path.dirname(__file__), "fbank_no_cmvn.yaml")
    return baseline_local(*args, **kwargs)
