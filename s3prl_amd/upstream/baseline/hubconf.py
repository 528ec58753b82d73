"""hub entries of the baseline upstream in the reference's naming (s3prl/upstream/baseline/hubconf.py:19-60);
only the kaldi-fbank configurations are implemented on the MI355X path.  The hyper-parameters of the reference's
``fbank.yaml`` / ``fbank_no_cmvn.yaml`` (80 mel bins, 25 ms window, 10 ms shift, log; delta order 2 over 5 frames;
CMVN on / off) are stated here as a dict; ``baseline_local`` still takes a yaml file in the same schema."""

import os

from .expert import UpstreamExpert as _UpstreamExpert


def _fbank_config(use_cmvn: bool) -> dict:
    return {
        "kaldi": {"feat_type": "fbank",
                  "fbank": {"num_mel_bins": 80, "frame_length": 25.0, "frame_shift": 10.0, "use_log_fbank": True}},
        "delta": {"order": 2, "win_length": 5},
        "cmvn": {"use_cmvn": use_cmvn},
    }


def baseline_local(model_config, *args, **kwargs):
    assert isinstance(model_config, dict) or os.path.isfile(model_config)
    return _UpstreamExpert(model_config, *args, **kwargs)


def baseline(*args, **kwargs):
    return fbank(*args, **kwargs)


def fbank(*args, **kwargs):
    kwargs["model_config"] = _fbank_config(True)
    return baseline_local(*args, **kwargs)


def fbank_no_cmvn(*args, **kwargs):
    kwargs["model_config"] = _fbank_config(False)
    return baseline_local(*args, **kwargs)
