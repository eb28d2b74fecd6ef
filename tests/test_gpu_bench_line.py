"""The driver's line (`python bench.py`, N = 1) on a small grid: the contract fields are there, `value` is the CSR loop, and every committed PMC
constant is either tied to the binary that is running (`traffic_binary_matches: true`) or withheld -- never a number from another build.
GPU box only; the full-size line is what the driver runs itself."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_line_is_contract_complete_and_counters_are_tied_to_the_loaded_binary(pkg):
    env = dict(os.environ, MIK_BENCH_MIN_SECONDS="0.05")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--grid", "128", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", "--no-gmres", "--no-config5",
                          "--no-f-solvers", "--no-gmres-large"], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["metric"] == "cg_iters_per_sec" and line["unit"] == "iters/s" and line["n_gpus"] == 1 and line["dtype"] == "f64" and line["value_is_contract"]
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["kernel"].startswith("k_spmv_rowgather") and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert abs(line["value"] - 1e3 / line["ms_per_step"]) < 1e-6 * line["value"] and r["avg_launch_ms"] <= line["ms_per_step"]
    assert "extras" not in line and "idrs8_per_step" not in json.dumps(line)              # nothing outside the scope contract on the default line
    sha = hashlib.sha256(open(pkg._lib.LIB_PATH, "rb").read()).hexdigest()
    assert r["libmik_sha256"] == sha
    src = json.load(open(os.path.join(ROOT, r["traffic_source"])))
    same = src["_binary"]["libmik_sha256"] == sha
    assert r["traffic_binary_matches"] == same
    if same:
        assert r["traffic"] == src["k_spmv_rowgather"]["traffic_bytes_per_launch"] and "_Z16k_spmv_rowgather" in src["_binary"]["mangled"]["k_spmv_rowgather<double, true, true>"]
    else:
        assert r["traffic"] is None and "traffic_withheld" in r
    for kk in line["step_kernels"]:
        assert kk["traffic_binary_matches"] in (True, False, None) and (kk["traffic"] is None or kk["traffic_binary_matches"])
