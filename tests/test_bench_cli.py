"""bench.py's launcher logic (no GPU work): `--gpus N` without torch.distributed.run must start N ranks itself, and must
fail loudly instead of silently running one rank when fewer than N devices are visible (`python3 bench.py
--gpus 8` printed n_gpus = 1)."""
import os
import subprocess
import sys

from conftest import ROOT


def test_gpus_2_without_two_devices_fails_loudly():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two devices are visible: the launcher would really start two ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0
    assert "2 ranks requested" in (p.stderr + p.stdout) and "device(s) visible" in (p.stderr + p.stdout)
    assert '"n_gpus"' not in p.stdout                      # no JSON line pretending to be a 2-GPU result


def test_world_size_mismatch_is_rejected():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2" in (p.stderr + p.stdout)
