"""First contact of `bench.py --gpus N` with a machine, with what the box has: two ranks on two devices where there are two,
otherwise both on device 0 (HIP IPC has no one-rank-per-device rule; RCCL has, and is then reported as "needs distinct devices").

  * the self-test children (iterativesolvers.jl_amd/selftest.py) prove the mailbox slots and the 4 MB landing buffers word by word;
  * the line carries `transport_selftest`, only transports that passed are entered;
  * with every transport between processes told to fail the line still comes -- through the in-process group -- contract-complete and
    bit-identical to the partition-aware oracle;
  * a launcher that cannot start ranks at all measures through the group itself;
  * a first transport that never returns inside the rank processes is rescued by the watchdog: a fresh process measures through the group."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _ndev():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", MIK_BENCH_MIN_SECONDS="0.05")
    if _ndev() < 2:
        env["MIK_FORCE_DEVICE"] = "0"
    env.update(kw)
    return env


def _bench(*args, **envkw):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--grid", "128", "--steps", "20", "--warmup", "3", "--no-cpu-baseline", *args],
                         capture_output=True, text=True, env=_env(**envkw), timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, out.stdout[-2000:]                # ONE line, whatever path produced it
    return json.loads(lines[0]), out.stderr


def _contract_complete(line):
    assert line["n_gpus"] == 2 and line["value_is_contract"] and "k_spmv_rowgather" in line["roofline"]["kernel"]
    assert abs(line["value"] - 1e3 / line["ms_per_step"]) < 1e-6 * line["value"]
    assert line["value_bytes_per_step_per_gpu"] / (line["ms_per_step"] * 1e-3) / 1e9 <= 8000.0
    assert 0.0 < line["roofline"]["frac"] <= 1.0 and line["roofline"]["avg_launch_ms"] <= line["ms_per_step"]
    assert line["parity_vs_oracle"]["bit_identical"], line["parity_vs_oracle"]
    assert line["contract_csr_loop"]["first_residuals_equal_the_default_layout_bit_for_bit"]
    assert line["wall"]["seconds_so_far"] < line["wall"]["limit_seconds"]


def test_selftest_children_prove_mailbox_slots_and_landing_buffers(tmp_path):
    devs = [0, 1] if _ndev() >= 2 else [0, 0]
    script = os.path.join(ROOT, "iterativesolvers.jl_amd", "selftest.py")
    procs = [subprocess.Popen([sys.executable, script, "--transport", "mailbox", "--rank", str(r), "--world", "2", "--device", str(devs[r]), "--dir", str(tmp_path / "m"),
                               "--timeout", "90"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=_env()) for r in range(2)]
    res = []
    for p in procs:
        so, se = p.communicate(timeout=150)
        res.append(json.loads([ln for ln in so.splitlines() if ln.startswith("{")][-1]))
    for r, rec in enumerate(res):
        assert rec["pass"], rec
        assert rec["mailbox"]["connected"] and rec["checks"]["mailbox_scalars"]["pass"] and rec["checks"]["mailbox_scalars"]["rounds"] == 64
        land = rec["checks"]["landing_4MB"]
        peer = 1 - r
        pn = land["per_neighbour"][f"{peer}->{r}"]
        assert land["pass"] and land["bytes_per_neighbour"] == 4 << 20 and pn["pass"] and pn["words_wrong"] == 0 and pn["words_checked"] == 6 * (4 << 20) // 8
        assert land["landing_buffer"]["connected"] and land["us_per_exchange_median"] > 0 and land["back_to_back_exchanges_timed"] == 20


def test_line_carries_the_selftest_and_only_enters_transports_that_passed():
    line, err = _bench()
    st = line["transport_selftest"]
    assert "bench.py: transport_selftest" in err                                   # on stderr before anything is timed
    assert st["mailbox"]["pass"] and "mailbox" in st["usable"]
    assert all(r["checks"]["landing_4MB"]["pass"] and r["checks"]["mailbox_scalars"]["pass"] for r in st["mailbox"]["ranks"])
    if _ndev() < 2:
        assert st["ranks_share_a_device"] and not st["rccl"]["pass"] and st["rccl"]["skipped"] and "distinct devices" in st["rccl"]["failure"]
        assert st["usable"] == ["mailbox"] and set(st["dropped_from_candidates"]) == {"rccl+mailbox", "rccl"}
    assert line["config"]["transport_chosen"] in st["usable"] and set(line["config"]["transports_measured"]) <= set(st["usable"])
    _contract_complete(line)
    assert set(line["parity_vs_oracle"]["transports"]) == {f"{t}/{lay}" for t in st["usable"] for lay in ("auto", "csr")}


def test_no_usable_transport_between_processes_still_gives_the_line_through_the_group():
    line, err = _bench(MIK_SELFTEST_FAIL="mailbox,rccl")
    st = line["transport_selftest"]
    assert st["usable"] == [] and not st["mailbox"]["pass"] and "simulated" in st["mailbox"]["failure"]
    assert line["config"]["transport_chosen"] == "group" and "in-process group" in line["config"]["transport"] and "in-process group" in err
    _contract_complete(line)
    assert set(line["parity_vs_oracle"]["transports"]) == {"group/auto", "group/csr"}


def test_launcher_that_cannot_start_ranks_measures_through_the_group_itself():
    line, err = _bench(MIK_SPAWN_FAIL="1")
    assert "produced no line" in err and line["config"]["transport_chosen"] == "group" and "bootstrap_failure" in line["config"]
    assert line["transport_selftest"]["reached"] is False
    _contract_complete(line)


def test_first_transport_that_never_returns_is_rescued_by_a_fresh_group_process():
    """the watchdog covers the first transport too: nothing measured yet -> the ranks leave, a fresh process measures through the in-process group"""
    line, err = _bench(MIK_BENCH_HANG_FIRST="1", MIK_BENCH_WATCHDOG_S="8")
    assert "did not return in time inside the rank processes" in err and line["config"]["transport_chosen"] == "group"
    _contract_complete(line)
