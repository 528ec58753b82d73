"""CPU-side checks of the boundary: the shared library loads and exports every symbol include/s3enc.h declares,
and fails loudly (no fallback) when there is no GPU."""

import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _built():
    from s3prl_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g

        g.build()
    return _lib


def test_library_exports_every_declared_symbol():
    _lib = _built()
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "s3enc.h")).read()
    declared = set(re.findall(r"\b(s3enc_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in s3enc.h but not exported by libs3enc.so"
    assert declared == set(_lib._PROTOS), "ctypes prototypes and header disagree"
    version = int(re.search(r"#define\s+S3ENC_VERSION\s+(\d+)", header).group(1))
    assert lib.s3enc_version() == _lib.ABI_VERSION == version


def test_graft_entry_build_passes_on_a_built_tree():
    """The driver's build check: make is a no-op on a built tree and the version cross-check (library, ctypes mirror,
    header) must agree — round 2 shipped a literal that went stale when the ABI version was raised."""
    _built()
    import __graft_entry__ as g

    g.build()


def test_struct_layout_matches_header(tmp_path):
    """The header is plain C: compile it with gcc and compare struct sizes / offsets with the ctypes mirror."""
    import subprocess

    _lib = _built()
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "s3enc.h"\n'
        'int main(void){printf("%zu %zu %zu %zu %zu\\n", sizeof(s3enc_config), sizeof(s3enc_tensor), '
        'sizeof(s3enc_profile_entry), offsetof(s3enc_config, compute_dtype), offsetof(s3enc_tensor, shape));'
        'printf("%zu %zu\\n", sizeof(s3enc_fbank_config), offsetof(s3enc_fbank_config, cmvn_eps));'
        'printf("%zu %zu %zu\\n", offsetof(s3enc_config, pred_heads), sizeof(s3enc_forward_opts), '
        'offsetof(s3enc_forward_opts, feat_w));'
        'printf("%zu %zu %zu\\n", offsetof(s3enc_config, mr_pairs), offsetof(s3enc_config, mr_layers), '
        'offsetof(s3enc_config, mr_plain));return 0;}\n'
    )
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    got = [int(v) for v in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got == [C.sizeof(_lib.S3Config), C.sizeof(_lib.S3Tensor), C.sizeof(_lib.S3ProfileEntry),
                   _lib.S3Config.compute_dtype.offset, _lib.S3Tensor.shape.offset,
                   C.sizeof(_lib.S3FbankConfig), _lib.S3FbankConfig.cmvn_eps.offset,
                   _lib.S3Config.pred_heads.offset, C.sizeof(_lib.S3ForwardOpts), _lib.S3ForwardOpts.feat_w.offset,
                   _lib.S3Config.mr_pairs.offset, _lib.S3Config.mr_layers.offset, _lib.S3Config.mr_plain.offset]


def test_no_gpu_means_loud_failure():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    _lib = _built()
    from s3prl_amd.encoder import HipEncoder
    from s3prl_amd.synth import named_config, synth_weights

    cfg = named_config("tiny_hubert")
    with pytest.raises(_lib.S3EncError, match="no CPU fallback"):
        HipEncoder(cfg, synth_weights(cfg, 0))
