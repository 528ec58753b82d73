"""bench.py's launcher contract on the CPU box: `python bench.py --gpus N` with no WORLD_SIZE must start N ranks itself
(re-exec under torch.distributed.run) and report the world size the process group really has — a run that silently
measures one GPU under `--gpus 8` would corrupt the driver's scaling table."""

import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", "--backend", "gloo", *extra],
                         capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_gpus_n_self_launches_n_ranks():
    line = _run(["--gpus", "2", "--batch", "3"])
    assert line["n_gpus"] == 2 and line["exchange_ok"] and line["utterances_per_rank"] == 3 and line["scaling"] == "weak"


def test_strong_scaling_splits_the_global_batch():
    line = _run(["--gpus", "3", "--scaling", "strong", "--global-batch", "8"])
    assert line["n_gpus"] == 3 and line["exchange_ok"] and line["utterances_per_rank"] == 3 and line["scaling"] == "strong"


def test_single_rank_needs_no_launcher():
    line = _run(["--gpus", "1"])
    assert line["n_gpus"] == 1 and line["exchange_ok"]


def test_tools_and_scripts_are_syntactically_valid():
    """The measurement tools only run on the GPU box; keep them at least importable / parseable here."""
    import glob
    import py_compile
    import subprocess

    for path in sorted(glob.glob(os.path.join(ROOT, "tools", "*.py"))) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")]:
        py_compile.compile(path, doraise=True)
    for path in sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh"))):
        subprocess.run(["bash", "-n", path], check=True)


def test_direct_exchange_algo_in_the_n_rank_dry_run():
    """--exchange-algo direct: the all-pairs send / receive form of the per-state exchange, three ranks over gloo."""
    line = _run(["--gpus", "3", "--batch", "2", "--exchange-algo", "direct"])
    assert line["n_gpus"] == 3 and line["exchange_ok"]


def test_stale_traffic_record_is_not_quoted(tmp_path):
    """roofline.traffic comes from committed PMC passes; a record measured on other kernels than this tree's (its csrc stamp
    differs, or it has none) must not be quoted as if it were current."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rec = {"model": "hubert_base", "dtype": "fp32", "batch": 32, "secs": 10.0, "gemm_bytes_per_launch": 123, "source": "x"}
    path = tmp_path / "traffic.json"
    path.write_text(json.dumps([rec]))
    got, note = bench.pmc_traffic(str(path), "hubert_base", "fp32", 32, 10.0)
    assert got is None and "stale" in note
    path.write_text(json.dumps([dict(rec, csrc_sha16=bench.csrc_sha16(), commit="abc1234")]))
    got, note = bench.pmc_traffic(str(path), "hubert_base", "fp32", 32, 10.0)
    assert got["gemm_bytes_per_launch"] == 123 and note is None
    assert bench.pmc_traffic(str(path), "hubert_large", "fp32", 32, 10.0) == (None, None)
    # round 5: the second identity — the gfx950 code objects of the built library.  A record whose source stamp is stale (host code
    # under csrc/ changed) is still current when the .hip_fatbin it was measured on is the one this tree builds; any other md5 is not
    md5 = bench.device_code_md5()
    assert md5 is not None and len(md5) == 32, "libs3enc.so is built by the CPU suite's first test (build())"
    path.write_text(json.dumps([dict(rec, csrc_sha16="0" * 16, device_code_md5=md5)]))
    got, note = bench.pmc_traffic(str(path), "hubert_base", "fp32", 32, 10.0)
    assert got is not None and note is None
    path.write_text(json.dumps([dict(rec, csrc_sha16="0" * 16, device_code_md5="f" * 32)]))
    got, note = bench.pmc_traffic(str(path), "hubert_base", "fp32", 32, 10.0)
    assert got is None and "stale" in note


def test_committed_traffic_records_of_the_headline_workload_are_current():
    """The default bench line quotes roofline.traffic from profiles/traffic.json: the committed record of the metric's workload must
    match this tree (by source stamp or by device code), else the driver's line says `traffic: null`."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rec, note = bench.pmc_traffic(os.path.join(ROOT, "profiles", "traffic.json"), "hubert_base", "fp32", 32, 10.0)
    assert rec is not None, note


def test_fp16_error_budget_tool_runs_on_a_tiny_fixture():
    """tools/fp16_error_budget.py (the float64 emulation behind DESIGN §5's choice of fp32 operands in the fp16x2 mode): every
    site's own contribution is below the all-sites total, and exact conv activations lower it on a GroupNorm extractor."""
    import re

    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fp16_error_budget.py"), "tiny_hubert_pl"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = dict(re.findall(r"\| (.+?) \| ([0-9.e+-]+) \|", out.stdout))
    total = float(rows["every site"])
    assert 1e-4 < total < 5e-3
    for site in ("conv", "feat", "ln_out", "attn_out", "fc1_out"):
        assert float(rows[f"only `{site}`"]) <= total * 1.05
    assert float(rows["every site but `conv`"]) < total


def test_scale_curve_summary_table(tmp_path):
    """tools/scale_curve.sh ends with a python summary over the bench lines it collected: run that part on fabricated lines."""
    import re

    script = open(os.path.join(ROOT, "tools", "scale_curve.sh")).read()
    body = re.search(r"<<'PY'\n(.*?)\nPY\n", script, re.S).group(1)
    for name, n, v in (("weak_fp32_n1", 1, 440000.0), ("weak_fp32_n2_ring", 2, 860000.0), ("weak_fp32_n2_direct", 2, 870000.0)):
        line = {"n_gpus": n, "value": v, "ms_per_step": 36.0,
                "comm": {"exposed_ms_per_step": 0.4, "bytes_received_per_gpu_per_step": 123} if n > 1 else None}
        (tmp_path / f"{name}.json").write_text(json.dumps(line) + "\n")
    out = subprocess.run([sys.executable, "-c", body, str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stderr
    assert "| weak_fp32_n2_direct | 2 | 870000 | 36.0 | 1.98 |" in out.stdout and "weak_fp32_n1" in out.stdout


def test_mx_second_term_emulation_on_a_tiny_fixture():
    """`tools/fp16_error_budget.py mx`: the MX-fp4 second weight term of the fp16x2 mode emulated in float64 (`mx4` = the format
    `pack_mx4_lo` writes and the kernel builds for A).  By itself the term costs ~1e-4 per GEMM on the hidden states — an order
    below one fp16 term's 5e-4 per weight — and under the mode's own activation roundings it stays inside the tolerance."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fp16_error_budget.py"), "mx", "tiny_hubert_pl"],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = [l for l in out.stdout.splitlines() if l.startswith("| `tiny_hubert_pl`")]
    assert len(rows) == 2
    alone = [float(c) for c in rows[0].strip().strip("|").split("|")[2:]]
    mode = [float(c) for c in rows[1].strip().strip("|").split("|")[2:]]
    assert alone[0] == 0 and all(1e-6 < e < 3e-4 for e in alone[1:]), alone       # conv1, q|k|v, fc1, fc2, three, all four
    assert all(e < 1e-3 for e in mode) and max(mode) < mode[0] + 1.5e-4, mode     # column 0: two fp16 terms


def test_mx4_quantiser_of_the_emulation_is_the_documented_format():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import fp16_error_budget as FB

    rng = np.random.default_rng(0)
    x = rng.standard_normal((7, 96)) * np.repeat(np.array([1e-3, 1.0, 300.0]), 32)
    q = FB.mx4(x)
    for b in range(3):
        blk, qb = x[:, 32 * b:32 * b + 32], q[:, 32 * b:32 * b + 32]
        s = np.exp2(np.ceil(np.log2(np.abs(blk).max(-1, keepdims=True) / 6.0)))
        codes = np.abs(qb) / s
        assert np.isin(np.round(codes * 2) / 2, FB.E2M1).all() and (codes <= 6.0).all()
        assert (np.sign(qb) * np.sign(blk) >= 0).all()
        grid = np.concatenate([-FB.E2M1[::-1], FB.E2M1])
        nearest = np.abs(blk[..., None] - grid * s[..., None]).min(-1)
        assert (np.abs(qb - blk) <= nearest + 1e-15).all()
    assert (FB.mx4(np.zeros((2, 32))) == 0).all()
