"""Host logic of the transport self-test (iterativesolvers.jl_amd/selftest.py, bench_dist.transport_selftest): the ring plan and its landing
offsets, the payload words, the file rendezvous, and the verdict logic of the parent -- a transport whose child fails, cannot start or is
told to fail is dropped from the candidates, never entered.  No GPU: on this box the children fail loudly with "no HIP device"."""
import importlib.util
import os
import threading

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def st():
    spec = importlib.util.spec_from_file_location("mik_selftest", os.path.join(ROOT, "iterativesolvers.jl_amd", "selftest.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_ring_plan_is_consistent_between_sender_and_receiver(st, world):
    M = 5
    plans = [st.ring_plan(r, world, M) for r in range(world)]
    for r, (recv, send, dst) in enumerate(plans):
        assert [p for p, _, _ in recv] == [q for q in (r - 1, r + 1) if 0 <= q < world] == [p for p, _, _ in send]
        assert len(dst) == len(send)
        for (q, _off, cnt), d in zip(send, dst):
            theirs = plans[q][0]
            match = [sg for sg in theirs if sg[0] == r]
            assert len(match) == 1 and match[0][1] == d and match[0][2] == cnt        # lands exactly on the receiver's segment for this sender
    assert plans[0][0] == ([] if world == 1 else [(1, 0, M)])


def test_payload_words_name_sender_receiver_round_and_index(st):
    a, b = st.pattern(0, 1, 0, 8), st.pattern(1, 0, 0, 8)
    assert a.dtype == np.uint64 and not np.any(a == b)
    assert np.array_equal(a & np.uint64(0xFFFFFFFF), np.arange(8, dtype=np.uint64))
    assert len({int(st.pattern(s, r, t, 1)[0]) for s in range(8) for r in range(8) for t in range(6)}) == 8 * 8 * 6
    assert np.all(np.isfinite(st.pattern(7, 7, 5, 1 << 12).view(np.float64)))       # plain small doubles: nothing a copy could canonicalise


def test_file_rendezvous_gathers_in_rank_order_and_times_out(st, tmp_path):
    import time
    P, got = 3, [None] * 3

    def worker(r):
        m = st.Meet(str(tmp_path / "a"), r, P, time.monotonic() + 20)
        got[r] = m.gather("h", bytes([r]) * 4)
        m.barrier("b")
    ts = [threading.Thread(target=worker, args=(r,)) for r in range(P)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert all(g == [bytes([q]) * 4 for q in range(P)] for g in got)
    lone = st.Meet(str(tmp_path / "b"), 0, 2, time.monotonic() + 0.2)
    with pytest.raises(TimeoutError):
        lone.gather("h", b"x")


@pytest.fixture(scope="module")
def bench_dist(pkg):
    from importlib import import_module
    return import_module(pkg.__name__ + ".bench_dist")


def test_parent_verdicts_drop_failed_transports(dist, bench_dist):
    boot = dist.SelfComm()
    boot.rank, boot.size = 0, 1
    rep = bench_dist.transport_selftest(boot, 0, 1, 0, ["mailbox", "rccl"], timeout=60, simulate_failure=["mailbox"])
    assert rep["mailbox"]["pass"] is False and "simulated" in rep["mailbox"]["failure"]
    assert rep["rccl"]["pass"] is False and rep["rccl"]["skipped"] is True
    assert rep["usable"] == []


def test_child_without_a_device_fails_loudly_and_the_parent_reports_it(dist, bench_dist):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    boot = dist.SelfComm()
    boot.rank, boot.size = 0, 1
    rep = bench_dist.transport_selftest(boot, 0, 1, 0, ["mailbox"], timeout=120)
    assert rep["mailbox"]["pass"] is False and "no HIP device" in rep["mailbox"]["failure"] and rep["usable"] == []
