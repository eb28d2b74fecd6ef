"""Measurement harness of ``bench.py --gpus N`` (BASELINE.json configs[3]) -- NOT product code: the row-partitioned iterables, engines,
communicators and halo plans it drives live in ``dist.py``; this module holds what only the benchmark needs:

  transport_selftest   first contact with a machine: every transport proves itself in killable child processes (``selftest.py``)
  build_group_problem  all P slabs in one process, for the in-process group leg
  bench_main           the line: self-test -> transports that passed, default layout -> contract loop on the CSR arrays -> parity against the
                       partition-aware oracle; in-process group when no transport between processes is usable; watchdog; progress notes
"""
from __future__ import annotations

import ctypes as C
import json
import os
import sys
import time

import numpy as np

from .dist import (GroupCG, HipEngine, NativeComm, NativeDistCGIterable, DistCGIterable, SelfComm, TorchComm, build_rank_problem, build_self_halo_problem,
                   complete_plan)

# ==============================================================================================
# first contact with the machine: transport self-test (child processes) and the in-process group
# ==============================================================================================
def transport_selftest(boot, rank, world, device, wanted, *, timeout=None, simulate_failure=()):
    """Run iterativesolvers.jl_amd/selftest.py for every transport in `wanted` ("mailbox", "rccl") as a CHILD process of every rank,
    before anything is timed: sequence-numbered scalars through the mailbox slots, 4 MB payloads through the landing buffers,
    ncclAllGather of one double and 4 MB ncclSend / ncclRecv, every word checked, every check timed.  A child that does not return in
    `timeout` seconds is killed -- the parent never touches a transport whose self-test failed.  Collective over `boot` (which only
    carries the verdicts).  Returns {"mailbox": {...}, "rccl": {...}, "usable": [...]} -- identical on every rank."""
    import shutil
    import subprocess
    import tempfile
    timeout = float(os.environ.get("MIK_SELFTEST_TIMEOUT_S", "75")) if timeout is None else float(timeout)
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "selftest.py")
    base = boot.all_gather_objects(tempfile.mkdtemp(prefix="mik_selftest_") if rank == 0 else None)[0]
    devices = boot.all_gather_objects((os.uname().nodename, int(device)))
    shared = len(set(devices)) < len(devices)
    report = {"what": "child processes (one per rank and transport) before any timed leg: iterativesolvers.jl_amd/selftest.py", "world": world,
              "ranks_share_a_device": bool(shared), "timeout_seconds": timeout}
    for name in wanted:
        t0 = time.perf_counter()
        if name in simulate_failure:
            mine = {"pass": False, "failure": f"failure simulated by MIK_SELFTEST_FAIL={name} (development)"}
        elif name == "rccl" and shared and world > 1:
            mine = {"pass": False, "skipped": True, "failure": "RCCL needs distinct devices: two ranks of this run share one GPU (ncclCommInitRank refuses duplicate devices)"}
        elif name == "rccl" and world == 1:
            mine = {"pass": False, "skipped": True, "failure": "a world of one has nothing to exchange over RCCL"}
        else:
            cmd = [sys.executable, script, "--transport", name, "--rank", str(rank), "--world", str(world), "--device", str(device),
                   "--dir", os.path.join(base, name), "--timeout", str(max(10.0, timeout - 10.0))]
            env = dict(os.environ)
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):       # the child meets its peers through files, not through the launcher's store
                env.pop(k, None)
            try:
                proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True)
                try:
                    so, se = proc.communicate(timeout=timeout)
                    lines = [ln for ln in so.strip().splitlines() if ln.startswith("{")]
                    mine = json.loads(lines[-1]) if lines else {"pass": False, "failure": f"no result line (exit code {proc.returncode}): {se.strip()[-300:]}"}
                except subprocess.TimeoutExpired:
                    proc.kill()
                    proc.communicate()
                    mine = {"pass": False, "failure": f"the self-test did not return within {timeout:.0f} s and was killed"}
            except Exception as exc:       # noqa: BLE001 -- cannot spawn: the transport is not usable from this container
                mine = {"pass": False, "failure": f"could not start the self-test child: {type(exc).__name__}: {exc}"[:300]}
        mine["wall_seconds"] = time.perf_counter() - t0
        every = boot.all_gather_objects(mine)
        rec = {"pass": all(bool(e.get("pass")) for e in every), "ranks": every}
        if not rec["pass"]:
            rec["failure"] = next((f"rank {q}: {e.get('failure') or 'a check failed'}" for q, e in enumerate(every) if not e.get("pass")), None)
            rec["skipped"] = all(bool(e.get("skipped")) for e in every if not e.get("pass"))
        else:
            def med(check, key):
                vals = [e["checks"][check][key] for e in every if check in e.get("checks", {}) and key in e["checks"][check]]
                return float(np.median(vals)) if vals else None
            rec["summary"] = ({"mailbox_scalars_us": med("mailbox_scalars", "us_per_round_median"), "landing_4MB_us": med("landing_4MB", "us_per_exchange_median"),
                               "landing_4MB_gbs_received": med("landing_4MB", "gbs_received")} if name == "mailbox" else
                              {"rccl_allgather_us": med("rccl_allgather", "us_per_round_median"), "rccl_halo_4MB_us": med("rccl_halo_4MB", "us_per_exchange_median"),
                               "rccl_halo_4MB_gbs_received": med("rccl_halo_4MB", "gbs_received")})
        report[name] = rec
    boot.barrier()
    if rank == 0:
        shutil.rmtree(base, ignore_errors=True)
    ok = {n for n in wanted if report.get(n, {}).get("pass")}
    report["usable"] = [t for t, needs in (("mailbox", {"mailbox"}), ("rccl+mailbox", {"rccl", "mailbox"}), ("rccl", {"rccl"})) if needs <= ok]
    return report


class _OneOf:
    """rank `rank` of a world of `size` whose plans are completed by the caller (build_group_problem)"""

    def __init__(self, rank, size):
        self.rank, self.size = rank, size

    def all_gather_objects(self, obj):
        self.mine = obj
        return [np.zeros(0, np.int64)] * self.size          # completed later, once every rank's needs are known


def build_group_problem(pkg, N: int, nz_per_rank: int, P: int, devices, dtype=np.float64):
    """build_rank_problem for all P ranks in ONE process (the in-process group, include/mik.h "Transport 2"): rank p's slab is generated
    on devices[p].  Returns (list of per-rank tuples as build_rank_problem returns them)."""
    probs, needs = [], []
    for p in range(P):
        fake = _OneOf(p, P)
        probs.append(build_rank_problem(pkg, fake, N, nz_per_rank=nz_per_rank, dtype=dtype, device=devices[p]))
        needs.append(fake.mine)
    for p in range(P):
        complete_plan(probs[p][3], probs[p][6], needs)
    return probs

# ==============================================================================================
# bench entry (bench.py --gpus N: one rank per GPU, started by torch.distributed.run or by bench.py itself)
# ==============================================================================================
def bench_main(args):
    """BASELINE.json configs[3]: cg! on the z-slab partition of the 512 x 512 x 64 P Laplacian (P = 8: the 512^3 grid;
    64 planes and two 512^2-double halos per rank), exchanges over RCCL issued from inside libmik.so.  P = 1
    (--force-dist) runs the same code path on the 256^3 grid of configs[1]."""
    import math
    import torch
    import torch.distributed as dist
    import __graft_entry__ as graft
    pkg = graft.load_package()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    if "MIK_FORCE_DEVICE" in os.environ:          # development: several ranks on one GPU (if the backend allows it)
        local_rank = int(os.environ["MIK_FORCE_DEVICE"])
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} needs device {local_rank}, {torch.cuda.device_count()} device(s) visible")
    torch.cuda.set_device(local_rank)
    transport = os.environ.get("MIK_DIST_TRANSPORT", "native")   # "native": RCCL inside libmik.so; "torch": phases driven from Python
    t_bench0 = time.perf_counter()
    group_only, boot_failure = False, None

    def note(msg):
        """progress on stderr with the time since start (rank 0): where a first run on new hardware spends its time, or stops, is visible in the log"""
        if rank == 0:
            print(f"bench.py [{time.perf_counter() - t_bench0:7.1f} s] {msg}", file=sys.stderr, flush=True)
    if world > 1 or "RANK" in os.environ:
        # the process group only bootstraps (ncclUniqueId, barriers, max over ranks of the timings): gloo suffices for the
        # native transport; the legacy transport needs torch's own RCCL communicator
        try:
            import datetime
            if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost", "::1"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # one node: no hostname lookup (a container's hostname may not resolve, or resolve slowly)
            if os.environ.get("MIK_BOOT_FAIL") == "1":
                raise RuntimeError("bootstrap failure simulated by MIK_BOOT_FAIL=1 (development)")
            if transport == "torch" and os.environ.get("MIK_DIST_BACKEND", "nccl") == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(os.environ.get("MIK_DIST_BACKEND", "gloo") if transport == "torch" else "gloo", rank=rank, world_size=world,
                                        timeout=datetime.timedelta(seconds=float(os.environ.get("MIK_BOOT_TIMEOUT_S", "180"))))
            boot = TorchComm()
            assert dist.get_world_size() == world
            note("process group (gloo) up")
        except Exception as exc:       # noqa: BLE001
            # The ranks cannot even meet (rendezvous refused, store unreachable): rank 0 measures the partitioned system alone through the
            # in-process group (include/mik.h "Transport 2": one host thread, every rank's slab on its own device, peer copies) -- the
            # driver still gets a contract-complete line; the other ranks leave quietly.
            boot_failure = f"{type(exc).__name__}: {exc}"[:300]
            print(f"bench.py: rank {rank}: process-group bootstrap failed ({boot_failure})", file=sys.stderr)
            if rank != 0:
                sys.exit(0)
            group_only = True
            boot = SelfComm()
    else:
        boot = SelfComm()
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but {world} rank(s) are running")
    K, Wm = args.steps, args.warmup
    if args.n is not None:
        N, nz = args.n, max(1, args.n // world)              # --grid G: the G^3 cube cut into `world` slabs
    elif world == 1:
        N, nz = 256, 256                                     # configs[1] through the partitioned code path
    else:
        N, nz = 512, 64                                      # configs[3]: 512 x 512 x 64 P (P = 8: 512^3)
    if "MIK_DIST_NZ" in os.environ:                      # development: planes per rank (e.g. --grid 512 with 64 planes = one rank's slab of configs[3])
        nz = int(os.environ["MIK_DIST_NZ"])
    t_up = time.perf_counter()
    # The slab is generated on the HOST (numpy, ~1.5 s for 16.7 M rows) and uploaded like any SparseMatrixCSC.  MIK_DIST_HOST_BUILD=0 generates it with
    # PyTorch on the device instead -- measured in round 6 with 3 / 4 processes on one GPU: 1 s on a fresh box, then 30 ... 500 s on the same box in
    # later runs (inside torch's indexing / scan ops; libmik's own allocations and kernels stayed at their usual times), so it is not the default.
    on_host = os.environ.get("MIK_DIST_HOST_BUILD", "1") == "1"
    self_halo = world == 1 and os.environ.get("MIK_DIST_SELF_HALO", "0") == "1"      # z-periodic slab: the rank exchanges its halo with itself
    torch.cuda.synchronize()
    note("HIP runtime up on the rank's device")
    group_devices = [int(os.environ["MIK_FORCE_DEVICE"])] * world if "MIK_FORCE_DEVICE" in os.environ else list(range(world))
    group_probs = None
    if group_only:
        if max(group_devices) >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: the in-process group needs devices {group_devices}, {torch.cuda.device_count()} visible")
        group_probs = build_group_problem(pkg, N, nz, world, [None] * world if on_host else group_devices)
        ptr, local_idx, val, plan, b_loc, n, offsets = group_probs[0]
    elif self_halo:
        ptr, local_idx, val, plan, b_loc, n, offsets = build_self_halo_problem(pkg, N, nz, local_rank)
    else:
        ptr, local_idx, val, plan, b_loc, n, offsets = build_rank_problem(pkg, boot, N, nz_per_rank=nz, device=None if on_host else local_rank, note=note)
    nnz_loc = int(val.numel() if hasattr(val, "numel") else val.size)
    note(f"{world} rank(s) met, slabs of {N}x{N}x{nz} generated ({plan.n_loc} rows, {nnz_loc} entries, {plan.n_ghost} halo entries on rank 0)")
    ptr_keep = True
    state = {"k": 0, "it": None}

    def run_steps(count, batch, keep=None):
        done = 0
        while done < count:
            h = state["it"].iterate_many(state["k"], min(batch, count - done))
            assert h.size > 0
            if keep is not None:
                keep.extend(h.tolist())
            done += h.size
            state["k"] += h.size

    def sync_devices():
        if state.get("solo"):                       # the in-process group: rank 0 drives every device
            for dv in sorted(set(group_devices)):
                torch.cuda.synchronize(dv)
        else:
            torch.cuda.synchronize()

    def region(count, batch):
        if not state.get("solo"):
            boot.barrier()
        sync_devices()
        t0 = time.perf_counter()
        run_steps(count, batch)
        sync_devices()
        if not state.get("solo"):
            boot.barrier()
        return time.perf_counter() - t0

    def max_over_ranks(values):
        if world == 1 or state.get("solo"):
            return list(values)
        t = torch.tensor(list(values), dtype=torch.float64)
        if dist.get_backend() != "gloo":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.cpu().tolist()

    def timed(batch, count):
        """regions of exactly `count` steps until >= 0.25 s have been measured; every rank runs the same number"""
        first = max_over_ranks([region(count, batch)])[0]
        more = max(0, min(199, math.ceil(0.25 / max(first, 1e-6)) - 1))
        times = [first] + max_over_ranks([region(count, batch) for _ in range(more)])
        return times

    HBM_PEAK = 8000.0
    sqrt_eps = float(np.sqrt(np.finfo(np.float64).eps))
    big = dict(ptr=ptr, local_idx=local_idx, val=val, plan=plan, b_loc=b_loc)

    def bring_up(name, layout, prob, reltol, maxiter):
        """One transport on an engine of its own over `prob` (operator layout "auto" = mik_csr_create's choice, "csr" = the plain Int32 CSR
        arrays: the contract loop).  Collective: returns (engine, comm, iterable, None) or (None, None, None, first failure of any rank)."""
        e2 = c2 = i2 = None
        failure = None
        try:
            e2 = HipEngine(pkg, prob["ptr"], prob["local_idx"], prob["val"], prob["plan"], prob["b_loc"], abstol=0.0, reltol=reltol, maxiter=maxiter,
                           device=local_rank, layout=layout)
            c2 = NativeComm(pkg, e2.ctx, boot, force_rccl=force and name != "mailbox", transport=name)
            i2 = NativeDistCGIterable(pkg, e2, c2, maxiter=maxiter)
        except Exception as exc:       # noqa: BLE001
            failure = f"{type(exc).__name__}: {exc}"
        failures = [f for f in boot.all_gather_objects(failure) if f]
        if failures:
            for o in (e2, c2):
                try:
                    o and o.close()
                except Exception:      # noqa: BLE001
                    pass
            return None, None, None, failures[0][:300]
        return e2, c2, i2, None

    def tear_down(e2, c2):
        boot.barrier()
        for o in (e2, c2):
            try:
                o and o.close()
            except Exception:          # noqa: BLE001
                pass

    # The transports inside libmik.so (include/mik.h "Transport 1" / "Transport 3"), each on an engine of its own over the same slab:
    #   rccl          halo by ncclSend / ncclRecv on the side stream, the two scalars of a step by ncclAllGather
    #   rccl+mailbox  halo by RCCL, scalars as stores into peer-mapped mailboxes (no collective launch on the compute stream)
    #   mailbox       no RCCL at all: scalars by mailbox, halo pushed into the neighbours' IPC-mapped landing buffers
    # (2) every transport that came up runs the warm-up and the timed regions in the operator's default layout; their first residuals must
    #     agree bit for bit; the fastest of the largest agreeing group is `transport_chosen`.
    # (3) the CONTRACT loop: the chosen transport on the plain CSR arrays of the slab (mik_csr_set_layout(A_loc, 0), k_spmv_rowgather) --
    #     `value`, `ms_per_step` and `roofline` describe this loop, exactly as at N = 1.
    # (1) parity, last (a hang in it cannot cost the timed line): every transport solves a SMALL global system (64 x 64 x 8 P) to the default
    #     tolerance in both operator layouts and rank 0 compares history and solution with the partition-aware oracle (bench.py hands the
    #     checker in; this module never imports oracle/).
    # MIK_NATIVE_TRANSPORTS narrows / reorders the list.
    transports = {}
    chosen = None
    eng = it = ncomm = None
    import threading
    watchdog = {"timer": None}

    def emergency_line():
        """Something measured AFTER a good transport hangs (no device-to-device transfer of any kind could be tried before the driver's own
        multi-GPU run): every rank leaves, rank 0 first prints the line of what has been measured so far -- the complete line without
        parity_vs_oracle if the hang is in the parity leg, the default-layout line of the best transport if it is in a later transport or in the
        contract loop."""
        note = "a leg that ran after this measurement did not return in time; the process left with the line it had"
        if rank == 0 and state.get("line_ready"):
            line = make_line(note)
            if line.get("parity_vs_oracle") is None:
                line["parity_vs_oracle"] = {"reached": False, "note": "the parity leg runs last and did not finish"}
            print(json.dumps(line), flush=True)
        elif rank == 0 and chosen is not None:
            ms = transports[chosen]["ms_per_step"]
            print(json.dumps({
                "metric": "cg_iters_per_sec", "value": 1e3 / ms, "unit": "iters/s", "n_gpus": world, "world_size_checked": world, "steps": K, "warmup": Wm,
                "value_is_contract": False, "aggregate_slab_iters_per_sec": world * 1e3 / ms, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"cg! on the {N}x{N}x{nz * world} 3D 7-point Laplacian row-partitioned into {world} z-slab(s) of {N}x{N}x{nz} rows",
                           "n": int(n), "n_per_gpu": plan.n_loc, "host_sync_per_step": 1, "transport_chosen": chosen, "transports_measured": transports,
                           "operator_layout_of_the_timed_loop": "default (slice-constant); the CSR contract loop was not reached", "watchdog": note},
                "roofline": None}), flush=True)
        elif rank == 0 and state.get("solo"):
            print("bench.py: the in-process group leg itself did not return in time; no line", file=sys.stderr, flush=True)
            os._exit(3)
        elif rank == 0:
            # Nothing measured yet and the FIRST transport hangs inside this process (it passed its self-test in a child, so this should not happen;
            # a thread stuck in a library call cannot be cancelled).  Last resort: a fresh process measures through the in-process group -- the other
            # ranks leave now and free their devices -- and this one forwards its line.
            import subprocess
            print("bench.py: the first transport did not return in time inside the rank processes; a fresh process measures through the in-process group",
                  file=sys.stderr, flush=True)
            env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                                     "TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS")}
            env["MIK_SPAWN_FAIL"] = "1"
            time.sleep(3.0)                                     # (the other ranks are leaving)
            root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
            argv = [a for a in sys.argv[1:]]
            try:
                rc = subprocess.call([sys.executable, os.path.join(root, "bench.py")] + argv, env=env, timeout=float(os.environ.get("MIK_BENCH_GROUP_RESCUE_S", "300")))
            except Exception as exc:       # noqa: BLE001
                print(f"bench.py: the rescue process failed: {exc}", file=sys.stderr, flush=True)
                rc = 3
            os._exit(rc)
        os._exit(0 if (chosen is not None or rank != 0) else 3)

    def arm_watchdog():
        if watchdog["timer"] is not None:
            watchdog["timer"].cancel()
        watchdog["timer"] = None
        if world > 1 and not group_only:      # (also before anything has been measured: a first transport that hangs in here must not hang the run)
            watchdog["timer"] = threading.Timer(float(os.environ.get("MIK_BENCH_WATCHDOG_S", "150")), emergency_line)
            watchdog["timer"].daemon = True
            watchdog["timer"].start()

    parity = None
    contract = None
    names, force = [], False
    def run_parity():
        """every transport solves a SMALL global system (64 x 64 x 8 P) to the default tolerance in both operator layouts; rank 0 compares history
        and solution with the partition-aware oracle (bench.py hands the checker in; this module never imports oracle/)"""
        nonlocal parity
        check = getattr(args, "partition_oracle_fn", None)
        if check is not None and not self_halo and not getattr(args, "no_parity", False):
            Ns, nzs = 64, 8
            sp, sli, sv, splan, sb, sn, soff = build_rank_problem(pkg, boot, Ns, nz_per_rank=nzs, device=None if on_host else local_rank)
            small = dict(ptr=sp, local_idx=sli, val=sv, plan=splan, b_loc=sb)
            parity = {"workload": f"cg! to reltol = sqrt(eps) on the {Ns}x{Ns}x{nzs * world} Laplacian, {world} z-slab(s) of {nzs} planes, hashed rhs, x0 = 0",
                      "oracle": "oracle/mik_oracle.c cg, TREE mode with the same row partition (rank-ordered sums of the per-rank trees)", "transports": {}}
            for name in names:
                for layout in ("auto", "csr"):
                    key = f"{name}/{layout}"
                    note(f"parity: {key} on the {Ns}x{Ns}x{nzs * world} system")
                    e2, c2, i2, failure = bring_up(name, layout, small, sqrt_eps, 10 ** 6)
                    if failure:
                        parity["transports"][key] = {"came_up": False, "failure": failure}
                        continue
                    hist, k2, failure = [], 0, None
                    try:
                        while True:
                            h = i2.iterate_many(k2, 1 if k2 < 2 else 25)       # single steps, then batches: both host protocols
                            if h.size == 0:
                                break
                            hist.extend(h.tolist())
                            k2 += h.size
                        xs = e2.solution()
                    except Exception as exc:      # noqa: BLE001
                        failure, xs = f"{type(exc).__name__}: {exc}", None
                    shape = e2.ctx.cg_shape(np.float64)
                    gathered = boot.all_gather_objects((failure, [float(v).hex() for v in hist], xs))
                    tear_down(e2, c2)
                    if any(g[0] for g in gathered):
                        parity["transports"][key] = {"came_up": True, "failure": next(g[0] for g in gathered if g[0])[:300]}
                        continue
                    rec = {"came_up": True, "iters": len(hist), "ranks_agree": all(g[1] == gathered[0][1] for g in gathered)}
                    if rank == 0:
                        ref = check(Ns, nzs * world, soff, shape)
                        rec.update(oracle_iters=int(ref["iters"]), same_iters_isconverged=bool(len(hist) == ref["iters"] and ref["isconverged"]),
                                   history_bit_identical=bool(np.array_equal(np.asarray(hist), ref["resnorm"])),
                                   solution_bit_identical=bool(np.array_equal(np.concatenate([g[2] for g in gathered]), ref["x"])))
                        rec["bit_identical"] = bool(rec["ranks_agree"] and rec["history_bit_identical"] and rec["solution_bit_identical"] and rec["same_iters_isconverged"])
                    parity["transports"][key] = rec
            if rank == 0:
                ok = [k2 for k2, v in parity["transports"].items() if v.get("bit_identical")]
                parity["bit_identical"] = bool(ok) and all(v.get("bit_identical") for v in parity["transports"].values() if v.get("came_up"))
                parity["transports_bit_identical"] = ok


    selftest = None
    if transport == "native":
        # (order: the transport whose waits are all bounded first -- once it has been measured, a hang of a later one is survivable)
        default = "mailbox,rccl+mailbox,rccl" if (world > 1 or self_halo) else "rccl"
        names = [t for t in os.environ.get("MIK_NATIVE_TRANSPORTS", default).split(",") if t and t != "group"]
        force = self_halo or os.environ.get("MIK_DIST_FORCE_COLLECTIVES", "0") == "1"
        if group_only:
            selftest = {"reached": False, "failure": f"the ranks could not meet: {boot_failure}", "usable": []}
            names = []
        elif world > 1 and os.environ.get("MIK_SELFTEST", "1") != "0":
            # ---- (0) FIRST CONTACT: every transport proves itself in child processes before anything of this process touches it ------------
            wanted = [t for t in ("mailbox", "rccl") if any(t in nm.split("+") for nm in names)]
            selftest = transport_selftest(boot, rank, world, local_rank, wanted, simulate_failure=[f for f in os.environ.get("MIK_SELFTEST_FAIL", "").split(",") if f])
            selftest["candidates"] = list(names)
            selftest["dropped_from_candidates"] = [nm for nm in names if nm not in selftest["usable"]]
            names = [nm for nm in names if nm in selftest["usable"]]
            if rank == 0:      # on stderr at once: visible even if a later leg takes the process down
                brief = {k: ({"pass": v.get("pass"), **({"summary": v["summary"]} if "summary" in v else {"failure": v.get("failure")})} if isinstance(v, dict) and "pass" in v else v)
                         for k, v in selftest.items() if k != "what"}
                print("bench.py: transport_selftest " + json.dumps(brief), file=sys.stderr, flush=True)
        if world == 1:
            pkg.lib().mik_set_tuning(6, int(os.environ.get("MIK_KNOB6", "4")))     # a world of one still sends its scalars through the mailbox (development)

        # ---- (2) every transport in the operator's default layout -----------------------------------------------------------------
        alive = {}
        for name in names:
            arm_watchdog()                      # (only once a transport has been measured: then a hang of the next one is survivable)
            t_up = time.perf_counter()
            note(f"transport {name}: bring-up (default layout)")
            if os.environ.get("MIK_BENCH_HANG_FIRST") == "1" and not alive:        # development: what the watchdog does when the first transport never returns
                time.sleep(10 ** 6)
            e2, c2, i2, failure = bring_up(name, "auto", big, 0.0, 10 ** 9)
            rec = {"came_up": failure is None}
            if failure:
                rec["failure"] = failure
                transports[name] = rec
                continue
            rec["operator_build_and_upload_seconds"] = time.perf_counter() - t_up
            state.update(k=0, it=i2)
            first = []
            try:
                run_steps(max(Wm, 8), 1, keep=first)
                tms = timed(1, K)
                failure = None
            except Exception as exc:       # noqa: BLE001
                failure = f"{type(exc).__name__}: {exc}"
            failures = [f for f in boot.all_gather_objects(failure) if f]
            if failures:
                rec.update(came_up=False, failure=failures[0][:300])
                transports[name] = rec
                continue
            note(f"transport {name}: {float(np.median(tms)) / K * 1e3:.4f} ms per step over {len(tms)} timed region(s) of {K} steps")
            rec.update(operator_layout=e2.A.layout(), ms_per_step=float(np.median(tms)) / K * 1e3, iters_per_sec=K / float(np.median(tms)), timed_regions=len(tms),
                       first_residuals=[float(v).hex() for v in first[:8]], uses_rccl=c2.uses_rccl())
            transports[name] = rec
            alive[name] = (e2, c2, i2, tms, state["k"])
            # provisional choice (what the watchdog would print): the fastest of the LARGEST group of transports with identical bits
            groups = {}
            for nm in alive:
                groups.setdefault(tuple(transports[nm]["first_residuals"]), []).append(nm)
            best = max(groups.values(), key=lambda g2: (len(g2), "rccl" in g2))
            for nm in alive:
                transports[nm]["same_bits_as_the_majority"] = nm in best
            chosen = min(best, key=lambda nm: transports[nm]["ms_per_step"])
        for nm, (e2, c2, i2, tms, kk) in alive.items():
            if nm == chosen:
                eng, ncomm, it, chosen_times, chosen_k = e2, c2, i2, tms, kk
            else:
                e2.close()
                c2.close()
        if chosen is None:
            if watchdog["timer"] is not None:
                watchdog["timer"].cancel()
            if rank == 0:
                print(f"bench.py: no transport between processes is usable ({ {k: v.get('failure') for k, v in transports.items()} }); "
                      f"measuring through the in-process group (one host thread, every slab on its own device, peer copies)", file=sys.stderr)
            transport = "group"
    del ptr_keep
    group = None
    if transport == "group":
        # ---- last resort (include/mik.h "Transport 2"): rank 0 drives all `world` slabs itself; needs no IPC handle, no RCCL, no second process ----
        chosen = "group"
        if rank != 0:                                         # rank 0 goes on alone; leaving with status 0 is not a failure for the launcher
            if dist.is_initialized():
                dist.destroy_process_group()
            return
        state["solo"] = True
        note(f"in-process group: {world} slabs on devices {group_devices}")
        if max(group_devices) >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: the in-process group needs devices {group_devices}, {torch.cuda.device_count()} visible")
        del ptr, local_idx, val
        big = None
        if group_probs is None:
            group_probs = build_group_problem(pkg, N, nz, world, [None] * world if on_host else group_devices)
        plan = group_probs[0][3]

        def group_up(probs, layout, reltol, maxiter):
            engs = [HipEngine(pkg, q[0], q[1], q[2], q[3], q[4], abstol=0.0, reltol=reltol, maxiter=maxiter, device=group_devices[p2], layout=layout)
                    for p2, q in enumerate(probs)]
            return engs, GroupCG(pkg, engs, maxiter=maxiter)

        def group_down(engs, grp):
            grp.close()
            for e2 in engs:
                e2.close()
        t_up = time.perf_counter()
        engs, it = group_up(group_probs, "auto", 0.0, 10 ** 9)
        upload_seconds = time.perf_counter() - t_up
        eng = engs[0]
        uses_rccl = False
        state.update(k=0, it=it)
        first = []
        run_steps(max(Wm, 8), 1, keep=first)
        times = timed(1, K)
        transports["group"] = {"came_up": True, "operator_layout": eng.A.layout(), "ms_per_step": float(np.median(times)) / K * 1e3, "iters_per_sec": K / float(np.median(times)),
                               "timed_regions": len(times), "first_residuals": [float(v).hex() for v in first[:8]], "uses_rccl": False,
                               "devices": group_devices, "operator_build_and_upload_seconds": upload_seconds}
        group = {"engs": engs, "up": group_up, "down": group_down, "first": transports["group"]["first_residuals"]}
    if transport == "group":
        pass
    elif transport == "native":
        state.update(k=chosen_k, it=it)
        uses_rccl = ncomm.uses_rccl()
        upload_seconds = transports[chosen]["operator_build_and_upload_seconds"]
        times = chosen_times
    else:
        t_up = time.perf_counter()
        eng = HipEngine(pkg, ptr, local_idx, val, plan, b_loc, abstol=0.0, reltol=0.0, maxiter=10 ** 9, device=local_rank)
        upload_seconds = time.perf_counter() - t_up
        it = DistCGIterable(eng, boot, maxiter=10 ** 9)
        uses_rccl = world > 1
        state.update(k=0, it=it)
        run_steps(Wm, 1)
        times = timed(1, K)                  # one host-visible residual per step: the reference's protocol, as at N = 1
    kb = max(1, K // 25) * 25
    times_b = timed(25, kb)                  # one host wait per 25 steps
    dt = float(np.median(times))
    default_layout = eng.A.layout()
    default_kernel = eng.A.spmv_kernel()
    stored_bytes = eng.A.spmv_stored_bytes()
    alg_bytes = eng.A.spmv_algorithmic_bytes()          # SURVEY.md 8d on the rank's n_loc x n_ext block: nnz (s + 4) + (n_loc + 1) 4 + n_ext s + n_loc s
    u = pkg.HipVector.wrap(eng.u_ext.data_ptr(), plan.n_loc + plan.n_ghost, np.float64, eng.ctx, owner=eng.u_ext)
    d_b2b_ms = eng.A.time_spmv(u, eng.c, reps=20, fused_dot=True)
    default_first = transports[chosen]["first_residuals"] if transport == "native" else None

    # ---- (3) the contract loop: the chosen transport on the plain CSR arrays ---------------------------------------------------
    if transport == "group" and default_layout != "csr-rowblock" and not getattr(args, "no_csr", False):
        group["down"](group["engs"], it)
        e3s, i3 = group["up"](group_probs, "csr", 0.0, 10 ** 9)
        e3 = e3s[0]
        state.update(k=0, it=i3)
        first = []
        run_steps(max(Wm, 8), 1, keep=first)
        ms0, cnt0 = C.c_double(), C.c_int64()
        for q in e3s:
            pkg._lib.check(q.L.mik_cgd_profile(q.handle, 1, None, None), "mik_cgd_profile", q.ctx.handle)
        tms = timed(1, K)
        steps_timed = state["k"] - max(Wm, 8)
        per = []
        for q in e3s:
            pkg._lib.check(q.L.mik_cgd_profile(q.handle, 0, C.byref(ms0), C.byref(cnt0)), "mik_cgd_profile", q.ctx.handle)
            per.append((ms0.value / max(steps_timed, 1), int(cnt0.value)))
        tb3 = timed(25, kb)
        dt3 = float(np.median(tms))
        u3 = pkg.HipVector.wrap(e3.u_ext.data_ptr(), plan.n_loc + plan.n_ghost, np.float64, e3.ctx, owner=e3.u_ext)
        b2b = e3.A.time_spmv(u3, e3.c, reps=20, fused_dot=True)
        contract = {"came_up": True, "kernel": e3.A.spmv_kernel(), "operator_layout": e3.A.layout(), "iters_per_sec": K / dt3, "ms_per_step": dt3 / K * 1e3,
                    "spmv_in_loop_ms": per[0][0], "spmv_launches_timed": per[0][1], "steps_timed": int(steps_timed),
                    "spmv_in_loop_ms_per_rank": [q[0] for q in per], "spmv_back_to_back_ms": b2b,
                    "batched_25_steps_per_sync_iters_per_sec": kb / float(np.median(tb3)), "timed_regions": len(tms), "final_residual": i3.residual,
                    "first_residuals_equal_the_default_layout_bit_for_bit": bool([float(v).hex() for v in first[:8]] == group["first"])}
        group["down"](e3s, i3)
        # parity of the group against the partition-aware oracle on the small global system, both layouts
        check = getattr(args, "partition_oracle_fn", None)
        if check is not None and not getattr(args, "no_parity", False):
            Ns, nzs = 64, 8
            small = build_group_problem(pkg, Ns, nzs, world, [None] * world if on_host else group_devices)
            parity = {"workload": f"cg! to reltol = sqrt(eps) on the {Ns}x{Ns}x{nzs * world} Laplacian, {world} z-slab(s) of {nzs} planes, hashed rhs, x0 = 0",
                      "oracle": "oracle/mik_oracle.c cg, TREE mode with the same row partition (rank-ordered sums of the per-rank trees)", "transports": {}}
            for layout in ("auto", "csr"):
                es, g2 = group["up"](small, layout, sqrt_eps, 10 ** 6)
                hist, k2 = [], 0
                while True:
                    h = g2.iterate_many(k2, 1 if k2 < 2 else 25)
                    if h.size == 0:
                        break
                    hist.extend(h.tolist())
                    k2 += h.size
                xs = g2.solution()
                ref = check(Ns, nzs * world, small[0][6], es[0].ctx.cg_shape(np.float64))
                rec = {"came_up": True, "iters": len(hist), "ranks_agree": True, "oracle_iters": int(ref["iters"]),
                       "same_iters_isconverged": bool(len(hist) == ref["iters"] and ref["isconverged"]),
                       "history_bit_identical": bool(np.array_equal(np.asarray(hist), ref["resnorm"])), "solution_bit_identical": bool(np.array_equal(xs, ref["x"]))}
                rec["bit_identical"] = bool(rec["history_bit_identical"] and rec["solution_bit_identical"] and rec["same_iters_isconverged"])
                parity["transports"][f"group/{layout}"] = rec
                group["down"](es, g2)
            ok = [k2 for k2, v in parity["transports"].items() if v.get("bit_identical")]
            parity["bit_identical"] = bool(ok) and all(v.get("bit_identical") for v in parity["transports"].values())
            parity["transports_bit_identical"] = ok
    if transport == "native" and default_layout != "csr-rowblock" and not getattr(args, "no_csr", False):
        arm_watchdog()
        note(f"contract loop: transport {chosen} on the plain CSR arrays")
        e3, c3, i3, failure = bring_up(chosen, "csr", big, 0.0, 10 ** 9)
        if failure:
            contract = {"came_up": False, "failure": failure}
        else:
            state.update(k=0, it=i3)
            first = []
            run_steps(max(Wm, 8), 1, keep=first)
            ms0, cnt0 = C.c_double(), C.c_int64()
            pkg._lib.check(e3.L.mik_cgd_profile(e3.handle, 1, None, None), "mik_cgd_profile", e3.ctx.handle)     # HIP events around every SpMV launch of the loop
            tms = timed(1, K)
            steps_timed = state["k"] - max(Wm, 8)
            pkg._lib.check(e3.L.mik_cgd_profile(e3.handle, 0, C.byref(ms0), C.byref(cnt0)), "mik_cgd_profile", e3.ctx.handle)
            tb3 = timed(25, kb)
            dt3 = float(np.median(tms))
            spmv_ms = ms0.value / max(steps_timed, 1)                   # per STEP (a step whose halo is ordered by events launches its SpMV in two parts)
            u3 = pkg.HipVector.wrap(e3.u_ext.data_ptr(), plan.n_loc + plan.n_ghost, np.float64, e3.ctx, owner=e3.u_ext)
            b2b = e3.A.time_spmv(u3, e3.c, reps=20, fused_dot=True)
            per_rank = boot.all_gather_objects((spmv_ms, int(cnt0.value), [float(v).hex() for v in first[:8]]))
            contract = {"came_up": True, "kernel": e3.A.spmv_kernel(), "operator_layout": e3.A.layout(), "iters_per_sec": K / dt3, "ms_per_step": dt3 / K * 1e3,
                        "spmv_in_loop_ms": spmv_ms, "spmv_launches_timed": int(cnt0.value), "steps_timed": int(steps_timed),
                        "spmv_in_loop_ms_per_rank": [q[0] for q in per_rank], "spmv_back_to_back_ms": b2b,
                        "batched_25_steps_per_sync_iters_per_sec": kb / float(np.median(tb3)), "timed_regions": len(tms), "final_residual": i3.residual,
                        "first_residuals_equal_the_default_layout_bit_for_bit": bool(all(q[2] == default_first for q in per_rank))}
            tear_down(e3, c3)
        if watchdog["timer"] is not None:
            watchdog["timer"].cancel()
    elif watchdog["timer"] is not None:
        watchdog["timer"].cancel()

    def make_line(note=None):
        """the JSON line from whatever has been measured so far (the watchdog prints it too, with `note`)"""
        halo = int(plan.n_ghost)
        s8 = 8
        iter_alg = alg_bytes + 9 * plan.n_loc * s8               # B_cg = B_spmv + 9 n s, on this rank's slab
        iter_moved = stored_bytes + 8 * plan.n_loc * s8          # the default layout's SpMV + the two fused sweeps (x update rides on the u sweep)
        is_contract = bool(contract and contract.get("came_up"))
        v_ms = contract["ms_per_step"] if is_contract else dt / K * 1e3
        v_ips = 1e3 / v_ms
        pmc = getattr(args, "pmc_traffic", None) or (lambda k, with_source=False: (None, None, None) if with_source else None)
        if is_contract:
            c_ms = contract["spmv_in_loop_ms"]
            c_traffic, c_src, c_ok = pmc(contract["kernel"], with_source=True)
            roofline = {"bound": "hbm", "kernel": contract["kernel"] + "<double, fused dot>", "loop": "contract_csr_loop (rank 0's slab; every rank runs the same loop)",
                        "achieved": alg_bytes / (c_ms * 1e-3) / 1e9, "peak": HBM_PEAK, "unit": "GB/s", "frac": alg_bytes / (c_ms * 1e-3) / 1e9 / HBM_PEAK,
                        "traffic": c_traffic, "traffic_source": c_src, "traffic_binary_matches": c_ok,
                        "traffic_is": "committed constant from separate rocprofv3 --pmc passes of the single-GPU command (same kernel, same rows per GPU), not measured in this run",
                        "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": c_ms, "launches_timed": contract["spmv_launches_timed"],
                        "avg_launch_ms_per_rank": contract["spmv_in_loop_ms_per_rank"], "back_to_back_ms": contract["spmv_back_to_back_ms"],
                        "loop_ms_per_step": contract["ms_per_step"], "loop_iters_per_sec": contract["iters_per_sec"],
                        "loop_algorithmic_bytes_per_step_per_gpu": iter_alg, "loop_gbs_per_gpu": iter_alg / (contract["ms_per_step"] * 1e-3) / 1e9,
                        "loop_frac_per_gpu": iter_alg / (contract["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK, "target": 0.60,
                        "note": "SURVEY.md 8d on one rank's slab: algorithmic bytes of the Int32 CSR SpMV of its n_loc x n_ext block over the average HIP-event time of the "
                                "SpMV launch INSIDE the partitioned cg! loop (mik_cgd_profile; mik_csr_set_layout(A_loc, 0), k_spmv_rowgather).  loop_* = that loop: "
                                "every GPU moves loop_algorithmic_bytes_per_step_per_gpu per step of the ONE global system, so bytes / ms_per_step <= 8 TB/s per GPU."}
        else:
            moved = stored_bytes / (d_b2b_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": f"{default_kernel}<double, fused dot> (operator layout {default_layout}; rank 0, back-to-back on the live u)",
                        "loop": "default layout (the CSR contract loop did not run)", "achieved": moved, "peak": HBM_PEAK, "unit": "GB/s", "frac": moved / HBM_PEAK,
                        "traffic": pmc(default_kernel), "bytes_moved_per_launch": stored_bytes, "avg_launch_ms": d_b2b_ms,
                        "note": "bytes this layout moves per launch over the HIP-event time; NOT the CSR-algorithmic figure"}
        out = {
            "metric": "cg_iters_per_sec", "value": v_ips, "unit": "iters/s", "n_gpus": world, "world_size_checked": world, "steps": K, "warmup": Wm,
            "ms_per_step": v_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "value_is_contract": is_contract,
            "value_semantics": "cg! iterations per second of the ONE global system (K / dt, max over ranks; the absolute number the north star quotes at 1 / 2 / 4 / 8 "
                               "GPUs), on the plain CSR arrays of every rank's slab.  Weak scaling: 16.7 M rows per GPU at every N, so the ideal is value(N) = value(1); "
                               "the whole-job aggregate in slab-iterations (N * K / dt) is `aggregate_slab_iters_per_sec`, in row updates `aggregate_row_updates_per_sec`",
            "value_bytes_per_step_per_gpu": iter_alg if is_contract else iter_moved,
            "value_gbs_per_gpu": (iter_alg if is_contract else iter_moved) / (v_ms * 1e-3) / 1e9,
            "aggregate_slab_iters_per_sec": world * v_ips, "aggregate_row_updates_per_sec": v_ips * n,
            "default_layout_iters_per_sec": K / dt, "default_layout_ms_per_step": dt / K * 1e3,
            "config": {"workload": f"cg! on the {N}x{N}x{nz * world} 3D 7-point Laplacian row-partitioned into {world} z-slab(s) of {N}x{N}x{nz} rows"
                                   + (" (BASELINE.json configs[3]: the 512^3 grid on 8 GPUs)" if (N, nz, world) == (512, 64, 8) else
                                      " (BASELINE.json configs[3] layout, weak-scaled: 16.7 M rows per GPU)" if world > 1 else
                                      " (BASELINE.json configs[1] through the row-partitioned code path)")
                                   + (" -- z-PERIODIC variant: the slab exchanges its 2 N^2 halo entries with itself over RCCL (MIK_DIST_SELF_HALO)" if self_halo else ""),
                       "n": int(n), "n_per_gpu": plan.n_loc, "nnz_per_gpu": nnz_loc, "halo_doubles_received_per_rank": halo,
                       "host_sync_per_step": 1, "reltol_in_timed_loop": 0.0,
                       "operator_layout_of_the_timed_loop": "csr (mik_csr_set_layout(A_loc, 0): Int32 rowptr / col / val, k_spmv_rowgather)" if is_contract else default_layout,
                       "timed_regions": contract["timed_regions"] if is_contract else len(times),
                       "transport": ({"rccl": "RCCL inside libmik.so (mik_cgd_iterate_many: ncclSend/ncclRecv halo on a side stream underneath the sweep over u "
                                              "+ 2 ncclAllGather of one double per rank per step)",
                                      "rccl+mailbox": "halo by ncclSend/ncclRecv on a side stream; the two scalars of a step as stores into peer-mapped "
                                                      "mailboxes, summed inside the finalising kernels (no collective launch on the compute stream)",
                                      "mailbox": "peer-mapped mailbox (no RCCL): scalars as stores into IPC-mapped slots, halo pushed into the neighbours' IPC-mapped landing buffers and copied into the ghost tail by the receiver"}[chosen]
                                     if transport == "native" and (uses_rccl or world > 1 or self_halo) else
                                     "none (world of one)" if transport == "native" else
                                     "in-process group (include/mik.h Transport 2): rank 0's host thread drives every slab on its own device, halos and the two scalars "
                                     "of a step as event-ordered peer copies -- the last resort when no transport between processes passed its self-test"
                                     if transport == "group" else "torch.distributed driven from Python (legacy)"),
                       "transport_chosen": chosen, "transports_measured": transports,
                       "halo_overlap": bool(getattr(eng, "overlap", False)),
                       "machine_of_rank_0": {k: v for k, v in eng.ctx.info().items() if k in ("arch", "compute_units", "xcds", "lds_bytes_per_cu", "hbm_bytes", "xcd_maps")},
                       "operator_build_and_upload_seconds": upload_seconds, "final_residual": contract["final_residual"] if is_contract else it.residual,
                       "default_layout": {"operator_layout": default_layout, "kernel": default_kernel, "iters_per_sec": K / dt, "ms_per_step": dt / K * 1e3,
                                          "bytes_per_step_per_gpu": iter_moved, "gbs_per_gpu": iter_moved / (dt / K) / 1e9, "spmv_back_to_back_ms": d_b2b_ms,
                                          "batched_25_steps_per_sync_iters_per_sec": kb / float(np.median(times_b)), "timed_regions": len(times),
                                          "timed_seconds_total": float(sum(times)), "final_residual": it.residual,
                                          "note": "the same partitioned iteration with every slab in the layout mik_csr_create picks for this constant-coefficient operator "
                                                  "(one mask byte per row instead of the CSR arrays); same residuals bit for bit; NOT the contract figure"}},
            "contract_csr_loop": contract,
            "parity_vs_oracle": parity,
            "roofline": roofline,
            "transport_selftest": selftest,
            "wall": {"seconds_so_far": time.perf_counter() - t_bench0,
                     "expected_seconds_of_the_whole_command_at_8_gpus": "60 - 90: launcher + imports ~8 s, slab generation ~2 s, self-test children ~3 s per transport "
                                                   "(RCCL bootstrap of 8 ranks: up to ~15 s), ~2 s per measured transport, contract loop ~2 s, parity (3 transports x 2 "
                                                   "layouts, small system) ~6 s, CPU baseline on rank 0 ~14 s.  Measured with every rank on ONE GPU (one transport): "
                                                   "2 / 3 / 4 ranks = 21 / 13 / 15 s incl. launcher (profiles/r06_first_contact_wall.json); hard limits: self-test child "
                                                   "75 s, mailbox waits 10 s, watchdog 150 s per leg",
                     "limit_seconds": 600},
        }
        if boot_failure:
            out["config"]["bootstrap_failure"] = boot_failure
        if note:
            out["config"]["watchdog"] = note
        return out

    state["line_ready"] = True
    if transport == "native":
        # ---- (1) parity, LAST: whatever happens in it, the timed line exists (the watchdog prints it with parity_vs_oracle = "not reached") ----------
        arm_watchdog()
        run_parity()
        if watchdog["timer"] is not None:
            watchdog["timer"].cancel()
    note("all GPU legs done" + ("; CPU baseline on rank 0" if not getattr(args, "no_cpu_baseline", False) else ""))
    if rank == 0:
        out = make_line()
        fn = getattr(args, "cpu_baseline_fn", None)
        if fn is not None and not getattr(args, "no_cpu_baseline", False):
            # the reference-shaped CPU restatement on this box's host cores, in the same run (rank 0 only; the other ranks wait at the
            # teardown): one rank's share is a 16.7 M-row system, i.e. the 256^3 workload of the single-GPU line
            cb = fn(256, max(40, int(getattr(args, "cpu_iters", 120))))       # never fewer than 40 iterations (VERDICT r5 #1c)
            cb.pop("_history", None)
            cb["sample"] += f"; one rank's share of the {world}-rank system has the same 16.7 M rows (the CPU would need {world} x as long per iteration of the global system)"
            out["cpu_baseline"] = cb
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()
