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


def test_tuning_keys_are_validated_process_wide_and_per_handle():
    """s3enc_set_tuning / s3enc_set_handle_tuning share one key table: unknown keys and out-of-range values fail with a
    message, a null handle fails, and the process-wide defaults stay what they were."""
    _lib = _built()
    lib = _lib.load()
    assert lib.s3enc_set_tuning(b"gemm32_big", 1) == 0
    assert lib.s3enc_set_tuning(b"gemm32_big", 99) != 0 and b"gemm32_big must be 0..5" in lib.s3enc_last_error()
    assert lib.s3enc_set_tuning(b"no_such_key", 0) != 0 and b"unknown key" in lib.s3enc_last_error()
    assert lib.s3enc_set_tuning(None, 0) != 0
    assert lib.s3enc_set_handle_tuning(None, b"gemm32_big", 1) != 0 and b"null handle" in lib.s3enc_last_error()
    for key in (b"gemm_variant", b"gemm_x3_tile", b"gemm16_big", b"attn_lds_pad", b"gemm_lds_pad", b"x3_pack_cache", b"gelu32"):
        assert lib.s3enc_set_tuning(key, 0) == 0, key
    # restore the defaults other tests rely on
    assert lib.s3enc_set_tuning(b"gemm_variant", 3) == 0 and lib.s3enc_set_tuning(b"gemm_x3_tile", 1) == 0
    assert lib.s3enc_set_tuning(b"gemm16_big", 3) == 0 and lib.s3enc_set_tuning(b"gelu32", 1) == 0


def test_comm_entry_points_fail_cleanly_without_a_communicator():
    """The RCCL exchange behind the C ABI: librccl is dlopen'ed on first use (its version is readable without a GPU); null
    communicators / ids are rejected with a message instead of crashing."""
    _lib = _built()
    lib = _lib.load()
    v = C.c_int32(0)
    rc = lib.s3enc_comm_version(C.byref(v))
    assert (rc == 0 and v.value > 20000) or b"librccl" in lib.s3enc_last_error()
    assert lib.s3enc_comm_unique_id(None) != 0
    out = C.c_void_p()
    assert lib.s3enc_comm_init_rank(None, 1, 0, 0, C.byref(out)) != 0
    assert lib.s3enc_comm_allgather_states(None, None, 0, None, 0, 1, 4, None, None) != 0
    assert lib.s3enc_comm_destroy(None) == 0


def test_every_tuning_key_is_documented_in_the_header():
    """s3enc_set_tuning's keys (csrc/ops.hip) are part of the boundary: each one is described in include/s3enc.h."""
    import os
    import re

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ops = open(os.path.join(root, "s3prl_amd", "csrc", "ops.hip")).read()
    keys = re.findall(r'\{"([a-z0-9_]+)", &Tuning::', ops)
    assert len(keys) >= 8 and len(set(keys)) == len(keys)
    header = open(os.path.join(root, "include", "s3enc.h")).read()
    internal = {"gemm_lds_pad", "attn_lds_pad", "x3_pack_cache"}  # occupancy / micro-benchmark probes, named in kernels.h only
    missing = [k for k in keys if k not in internal and f'"{k}"' not in header]
    assert not missing, f"tuning keys without a description in include/s3enc.h: {missing}"
