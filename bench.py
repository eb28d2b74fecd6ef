#!/usr/bin/env python
"""bench.py -- CG iterations/s + achieved SpMV HBM GB/s on the 3D 7-point Laplacian (fp64).

    python bench.py --gpus N --steps K --warmup W

A "step" is one cg! iteration (iterate(::CGIterable), src/cg.jl:43-66) on synthetic input already resident in HBM.

N = 1: BASELINE.json configs[1] -- cg! on the 256^3 Laplacian through the fused single-GPU iterable.
N > 1: BASELINE.json configs[3] -- the same iteration row-partitioned into z-slabs, one process per GPU, halo exchange
       and scalar sums issued from INSIDE libmik.so (mik_cgd_iterate_many; RCCL or peer-mapped mailboxes over xGMI).  N = 8 is the
       512^3 grid (64 planes and two 512^2-double halos per rank); N = 2, 4 are its weak-scaled pieces 512 x 512 x 64N
       (16.7 M rows per GPU throughout, like the single-GPU 256^3).  If the process was not started by
       torch.distributed.run (WORLD_SIZE unset) bench.py launches the N ranks itself.
Rank 0 prints ONE JSON line.  At every N:
  `value` / `ms_per_step`  the CONTRACT loop: the operator (every rank's slab) on its plain Int32 CSR arrays (k_spmv_rowgather), one
                           host-visible residual per step like the reference's loop; iterations/s of the one (global) system;
                           `value_bytes_per_step[_per_gpu]` = SURVEY.md 8d's B_cg, so bytes / ms <= 8 TB/s is checkable from the top level
  `roofline`               SURVEY.md 8d bytes of the CSR SpMV over the in-loop HIP-event time of that kernel, + the loop it was timed in
  `default_layout_*`       the same iteration in the layout mik_csr_create picks by itself (one mask byte per row for this
                           constant-coefficient operator; bit-identical results) -- reported, NOT the contract figure
N = 1 adds: parity_full_history (the 613-step solve against the committed CPU histories), gmres_hbm_bound (gmres!(30) at 256^3),
f_solvers (PCG / Chebyshev / MINRES / BiCGStab(2)), gmres_config3 (configs[2]), config5 (configs[4] stand-ins), cpu_baseline (+ OpenMP).
N > 1 adds: parity_vs_oracle (every transport on a small global system against the partition-aware oracle), transports_measured.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC (before the HIP runtime initialises): what hipIpcGetMemHandle / RCCL need between processes on this driver
import __graft_entry__ as graft  # noqa: E402

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy ceiling 6290 GB/s
COPY_CEILING_GBS = 6290.0
MIN_TIMED_SECONDS = float(os.environ.get("MIK_BENCH_MIN_SECONDS", "0.25"))   # the K-step timed region is repeated until this much has been measured


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(N: int, iters: int):
    """The oracle's reference-shaped CG (serial CSC column-scatter SpMV with Int64 indices, serial fused loops, one
    thread) timed on this box's host cores on a bounded sample of the same workload."""
    orc = graft.load_oracle()
    A = orc.laplace(N, 3)
    b = orc.hashed_rhs(A.n)
    orc.cg(A, b, maxiter=2, mode="seq")                      # page-in / warm caches
    t0 = time.perf_counter()
    _, h = orc.cg(A, b, maxiter=iters, mode="seq")
    dt = time.perf_counter() - t0
    return {"value": h["iters"] / dt, "unit": "iters/s", "cores": 1, "kind": "port",
            "sample": f"{h['iters']} cg! iterations on the same {N}^3 operator and rhs, oracle/mik_oracle.c mode SEQ "
                      f"(host shows {os.cpu_count()} cores, {orc.effective_cpus()} usable under the cgroup quota; the reference's "
                      f"SpMV and broadcasts are single-threaded)",
            "seconds": dt, "cpu_model": cpu_model(), "final_residual": float(h["resnorm"][-1]) if h["iters"] else None,
            "_history": h["resnorm"]}


def cpu_baseline_omp(N: int, iters: int):
    """Best-effort multi-threaded host baseline (`cpu_ref_omp`): same algorithm and stopping rule,
    row-parallel CSR SpMV + OpenMP reductions on all host cores (oracle/mik_oracle_omp.c)."""
    orc = graft.load_oracle()
    A = orc.laplace(N, 3)
    b = orc.hashed_rhs(A.n)
    orc.omp_cg(A, b, maxiter=3)
    t0 = time.perf_counter()
    _, it, _, threads = orc.omp_cg(A, b, maxiter=iters)
    dt = time.perf_counter() - t0
    return {"value": it / dt, "unit": "iters/s", "cores": threads, "kind": "port",
            "sample": f"{it} cg! iterations on the same {N}^3 operator, OpenMP row-parallel CSR restatement "
                      f"(not the reference's serial loop; summation order differs)", "seconds": dt}


_LIB_SHA = {}


def loaded_library_sha256():
    """sha256 of the libmik.so this process has loaded (the file the ctypes binding opened)"""
    import hashlib
    pkg = graft.load_package()
    path = pkg._lib.LIB_PATH
    if path not in _LIB_SHA:
        _LIB_SHA[path] = hashlib.sha256(open(path, "rb").read()).hexdigest()
    return _LIB_SHA[path]


def traffic_file_matches(t: dict):
    """Does a committed *_traffic.json describe the binary that is running?  scripts/prof_collect.py records the sha256 of the libmik.so its
    counters were collected on; files written before round 6 carry none (-> False: their figures are withheld)."""
    sha = (t.get("_binary") or {}).get("libmik_sha256")
    return bool(sha) and sha == loaded_library_sha256()


def pmc_traffic(kernel_key: str, with_source: bool = False):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 PMC passes (profiles/*bench*_traffic.json, written by
    scripts/prof_r06.sh + prof_collect.py from separate --pmc runs of this same command: FETCH_SIZE x 2 per the gfx950 note in
    MI355X_MICROARCH.md + WRITE_SIZE).  A COMMITTED CONSTANT, not a measurement of this run (counters need rocprofv3 around the
    process) -- therefore tied to the binary: the newest file is used, and only if it was collected on the very libmik.so that is loaded
    now (`traffic_binary_matches`); otherwise the figure is WITHHELD (None).  with_source: (bytes | None, file, matches)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*bench*_traffic.json")), reverse=True):     # newest round first
        try:
            t = json.load(open(f))
            v = t.get(kernel_key, {}).get("traffic_bytes_per_launch")
        except Exception:
            continue
        if v:
            ok = traffic_file_matches(t)
            return ((v if ok else None), os.path.relpath(f, ROOT), ok) if with_source else (v if ok else None)
    return (None, None, None) if with_source else None


def traffic_fields(kernel_key: str):
    v, src, ok = pmc_traffic(kernel_key, with_source=True)
    out = {"traffic": v, "traffic_source": src, "traffic_binary_matches": ok}
    if src and not ok:
        out["traffic_withheld"] = "the counters in traffic_source were collected on another build of libmik.so (sha256 differs): re-run scripts/prof_r06.sh"
    return out


def history_parity(N: int, gpu_hist: np.ndarray, gpu_iters: int, gpu_mvps: int, gpu_converged: bool):
    """The full residual history of cg! to its default tolerance against the committed CPU histories of the same
    solve (tests/golden/cg_lap<N>.json, generated by tests/golden/make_golden.py from the oracle): one accumulator
    (seq), pairwise (pair), the host OpenBLAS with 1 and 8 threads (blas, blas8 -- what LinearAlgebra.dot / norm of
    the reference execute) and the device's documented tree (tree: must be bit-identical)."""
    path = os.path.join(ROOT, "tests", "golden", f"cg_lap{N}.json")
    if not os.path.exists(path):
        return None
    g = json.load(open(path))
    out = {"golden": os.path.relpath(path, ROOT), "gpu_iters": int(gpu_iters), "gpu_mvps": int(gpu_mvps), "gpu_isconverged": bool(gpu_converged),
           "blas_library": g.get("blas_library")}
    ref = {}
    for key in ("seq", "pair", "blas", "blas8", "tree"):
        if key not in g:
            continue
        r = np.array([float.fromhex(s) for s in g[key]["resnorm"]])
        ref[key] = r
        m = min(r.size, gpu_hist.size)
        dev = np.abs(gpu_hist[:m] - r[:m]) / r[:m]
        out[key] = {"iters": g[key]["iters"], "same_iters_mvps_isconverged": bool(g[key]["iters"] == gpu_iters and g[key]["mvps"] == gpu_mvps and
                                                                                 g[key]["isconverged"] == gpu_converged),
                    "gpu_vs_cpu_history_max_rel_dev": float(dev.max()), "at_iteration": int(dev.argmax()) + 1, "steps_compared": int(m),
                    "bit_identical": bool(m == r.size == gpu_hist.size and np.array_equal(gpu_hist, r))}

    # the same solve with dot / norm by OTHER hosts' OpenBLAS kernels (tests/golden/cg_lap<N>_blas_<host>.json, make_golden.py --blas-only): the
    # committed `blas` above was generated on the build container's Xeon (SkylakeX kernels); the GPU box's EPYC runs Zen kernels, a Julia
    # installation whatever its OpenBLAS build picks.  Every host must be within the north star's 1e-12; `spread` = how far the hosts are apart.
    import glob
    hosts = {}
    if "blas" in ref:
        hosts[str((g.get("blas_library") or {}).get("core", "golden")).lower()] = {"blas": ref["blas"], "blas8": ref.get("blas8"), "file": os.path.relpath(path, ROOT),
                                                                                  "cpu_model": "build container (Intel Xeon)", "core": (g.get("blas_library") or {}).get("core")}
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", f"cg_lap{N}_blas_*.json"))):
        try:
            h = json.load(open(f))
            hosts[h["host_tag"]] = {"blas": np.array([float.fromhex(s) for s in h["blas"]["resnorm"]]),
                                    "blas8": np.array([float.fromhex(s) for s in h["blas8"]["resnorm"]]) if "blas8" in h else None,
                                    "file": os.path.relpath(f, ROOT), "cpu_model": h.get("cpu_model"), "core": (h.get("blas_library") or {}).get("core"),
                                    "forced_coretype": h.get("forced_coretype"), "iters": h["blas"]["iters"], "isconverged": h["blas"]["isconverged"]}
        except Exception:      # noqa: BLE001
            continue
    if hosts:
        bh = {}
        for tag, h in hosts.items():
            rec = {"file": h["file"], "cpu_model": h["cpu_model"], "openblas_core": h["core"], **({"forced_coretype": h["forced_coretype"]} if h.get("forced_coretype") else {})}
            for key in ("blas", "blas8"):
                r = h.get(key)
                if r is None:
                    continue
                m = min(r.size, gpu_hist.size)
                rec[f"gpu_vs_{key}_history_max_rel_dev"] = float(np.max(np.abs(gpu_hist[:m] - r[:m]) / r[:m]))
                rec[f"{key}_same_iteration_count"] = bool(r.size == gpu_hist.size)
            bh[tag] = rec
        tags = list(hosts)
        spread = 0.0
        for i in range(len(tags)):
            for j in range(i + 1, len(tags)):
                a, b2 = hosts[tags[i]]["blas"], hosts[tags[j]]["blas"]
                m = min(a.size, b2.size)
                spread = max(spread, float(np.max(np.abs(a[:m] - b2[:m]) / b2[:m])))
        out["blas_hosts"] = bh
        out["blas_hosts_spread_single_thread"] = spread if len(tags) > 1 else None
        worst = max(v.get("gpu_vs_blas_history_max_rel_dev", 1.0) for v in bh.values())
        out["blas_hosts_worst_single_thread"] = worst
        out["blas_hosts_all_within_1e-12"] = bool(worst <= 1e-12)
        # SURVEY.md 8d's parity statement: "<= max(1e-12, 3 x floor)", the floor being what two equally valid CPU summation orders differ by --
        # here two OpenBLAS kernels (AVX-512 vs AVX2) running the reference's own dot / norm calls
        out["blas_hosts_all_within_max_of_1e-12_and_3x_their_own_spread"] = bool(worst <= max(1e-12, 3.0 * spread)) if len(tags) > 1 else None

    def floor(a, b):
        m = min(ref[a].size, ref[b].size)
        return float(np.max(np.abs(ref[a][:m] - ref[b][:m]) / ref[b][:m]))
    out["cpu_vs_cpu_floors"] = {f"{a}_vs_{b}": floor(a, b) for a, b in (("seq", "pair"), ("blas", "pair"), ("blas", "blas8"), ("seq", "blas"))
                                if a in ref and b in ref}
    return out


def timed_regions(step_fn, K: int, sync):
    """Time EXACTLY K steps between synchronisations, repeated until MIN_TIMED_SECONDS have been measured; returns the
    list of region times.  (A single 20-step region at 256^3 lasts 8 ms -- too short for a stable rate.)"""
    times = []
    while not times or (sum(times) < MIN_TIMED_SECONDS and len(times) < 200):
        sync()
        t0 = time.perf_counter()
        for _ in range(K):
            step_fn()
        sync()
        times.append(time.perf_counter() - t0)
    return times


def gmres_config3():
    """BASELINE.json configs[2]: gmres!(restart = 30) on advection_dominated(N = 50, beta = 1000), fp64 (benchmark/advection_diffusion.jl:3-30).
    Wall time per inner iteration (src/gmres.jl:57-106: one expand! + one orthogonalize_and_normalize!) with the loop of gmres!
    inside the library (mik_gmres_iterate_many), for the three orth_meth values of src/orthogonalize.jl:4-7; the CPU oracle
    (mode SEQ, one thread) beside it; the MGS history against the committed golden (tests/golden/gmres_advdiff50_r30.json)."""
    import torch
    pkg = graft.load_package()
    N, restart = 50, 30
    # inputs from the PRODUCT-side fixture (rhs through libm's exp / sin, bit-equal to the oracle's generator and to the golden run:
    # tests/test_host_logic.py::test_advection_dominated_fixture_equals_the_oracle_generator); the oracle is only the CPU leg below
    n, colptr, rowval, nzval, b = pkg.fixtures.advection_dominated(N, 1000.0)
    A = pkg.HipCSR(n, n, colptr, rowval, nzval, index_base=1)
    db = pkg.HipVector.from_numpy(b)
    out = {"workload": f"gmres!(restart={restart}) on advection_dominated(N={N}, beta=1000), fp64, n = {n}, nnz = {A.nnz} (BASELINE.json configs[2])",
           "operator_layout": A.layout(), "spmv_kernel": A.spmv_kernel(), "us_per_inner_iteration": {}}
    spmv_bytes = A.spmv_algorithmic_bytes()
    for name, M in (("mgs", pkg.ModifiedGramSchmidt()), ("cgs", pkg.ClassicalGramSchmidt()), ("dgks", pkg.DGKS())):
        best, hist = None, None
        for _rep in range(3):                                 # the first full-length call of a process is not representative
            it = pkg.gmres_iterable_(pkg.zerox(A, db), A, db, restart=restart, orth_meth=M, initially_zero=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hist = it.iterate_many(0, 4096)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        # algorithmic bytes of SURVEY.md 8d summed over the columns k = 1..restart of a cycle, averaged per inner iteration
        per = [spmv_bytes + ((3 * k + 2) if name == "mgs" else (2 * k + 3)) * n * 8 for k in range(1, restart + 1)]
        out["us_per_inner_iteration"][name] = {"us": best / max(hist.size, 1) * 1e6, "inner_iterations": int(hist.size), "mv_products": int(it.mv_products),
                                              "algorithmic_bytes_avg": float(np.mean(per)),
                                              "note": "launch- and hand-off-bound at this size (V = 31 MB lives in L2 / Infinity Cache): no roofline fraction is quoted"}
        if name == "mgs":
            path = os.path.join(ROOT, "tests", "golden", "gmres_advdiff50_r30.json")
            if os.path.exists(path):
                g = json.load(open(path))
                tree = np.array([float.fromhex(v) for v in g["tree"]["resnorm"]])
                blas = np.array([float.fromhex(v) for v in g["blas"]["resnorm"]])
                m = min(tree.size, hist.size)
                out["parity_mgs"] = {"golden": os.path.relpath(path, ROOT), "bit_identical_to_tree": bool(tree.size == hist.size and np.array_equal(tree, hist)),
                                     "iters": int(hist.size), "golden_iters": int(g["tree"]["iters"]),
                                     "first_cycle_max_rel_dev_vs_blas": float(np.max(np.abs(hist[:restart] - blas[:restart]) / blas[:restart])),
                                     "whole_history_max_rel_dev_vs_blas": float(np.max(np.abs(hist[:m] - blas[:m]) / blas[:m]))}
    orc = graft.load_oracle()                             # CPU leg only
    Ao = orc.CSC(n, colptr, rowval, nzval, 1)
    t0 = time.perf_counter()
    _, ho = orc.gmres(Ao, b, restart=restart)
    dt = time.perf_counter() - t0
    out["cpu_oracle_us_per_inner_iteration"] = {"us": dt / max(ho["iters"], 1) * 1e6, "iters": int(ho["iters"]), "cores": 1, "kind": "port",
                                                "what": "oracle/mik_oracle.c mode SEQ, MGS, the same solve"}
    return out


def gmres_hbm_bound(A, b, n: int, restart: int = 30, inner: int = 60, reps: int = 3, methods=("mgs", "cgs")):
    """gmres!(restart = 30) at an HBM-BOUND size: the 256^3 Laplacian of configs[1] on its plain CSR arrays, fp64 -- the Krylov
    basis V is 31 x 134 MB = 4.2 GB, nothing of it survives in a cache from one sweep to the next.  `inner` inner iterations
    (src/gmres.jl:57-106: expand! + orthogonalize_and_normalize!, two restart cycles with their solve / update / init!), loop inside
    the library (mik_gmres_iterate_many), ModifiedGramSchmidt (one launch per column with w kept on the chip between the passes: pass i = w -= h_i v_i fused with the next
    projection v_{i+1} . w, src/orthogonalize.jl:69-76) and ClassicalGramSchmidt (k_multidot + k_gemv_n, :43-45).
    Bytes per inner iteration: SURVEY.md 8d -- B_spmv + (3k + 2) n s (MGS) / (2k + 3) n s (CGS) for Arnoldi column k, plus per
    restart (k + 2) n s (update) and B_spmv + 3 n s (init!) -- summed over the call and divided by its inner iterations; `moved` = what the
    launches of this implementation stream (MGS, resident-w form: (2k + 2 + (2k + 1) g) n s with g the share of w that does not fit on the chip --
    v_i is read by the pass that projects on it and again by the pass that subtracts it; the multi-launch chain (4k + 3) n s; CGS (2k + 6) n s)."""
    import torch
    pkg = graft.load_package()
    s8 = 8
    spmv_bytes = A.spmv_algorithmic_bytes()
    out = {"workload": f"gmres!(restart={restart}) on the 256^3 7-point Laplacian, fp64, hashed rhs, x0 = 0, {inner} inner iterations (reltol = 0), "
                       f"operator on its plain CSR arrays; V = {(restart + 1) * n * s8 / 1e9:.2f} GB",
           "operator_layout": A.layout(), "spmv_kernel": A.spmv_kernel(), "spmv_algorithmic_bytes": spmv_bytes}
    for name, M in (("mgs", pkg.ModifiedGramSchmidt()), ("cgs", pkg.ClassicalGramSchmidt())):
        if name not in methods:
            continue
        best, hist, it = None, None, None
        for _rep in range(reps):
            it = pkg.gmres_iterable_(pkg.zerox(A, b), A, b, restart=restart, orth_meth=M, initially_zero=True, reltol=0.0, maxiter=inner)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hist = it.iterate_many(0, inner)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        iters = int(hist.size)
        ks = [(j % restart) + 1 for j in range(iters)]                     # Arnoldi column of every inner iteration
        cycles = (iters + restart - 1) // restart
        alg = sum(spmv_bytes + ((3 * k + 2) if name == "mgs" else (2 * k + 3)) * n * s8 for k in ks)
        # what the launches stream.  Modified Gram-Schmidt in the resident-w form (csrc/mik_mgs_res.h; n > 2048 segments): first sweep w + v_1 (2 n),
        # k - 1 middle passes v_i + v_{i+1} + the non-resident share g of w read and written (2 + 2 g) n, last pass (1 + 2 g) n, final scale (1 + g) n
        # -> (2 k + 2 + (2 k + 1) g) n per column, instead of the chain's (4 k + 3) n
        form = None
        if name == "mgs":
            import ctypes as C
            sl, sg, th, rr, rl = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
            L = pkg.lib()
            if L.mik_dev_gmres_form(it.handle, C.byref(sl), C.byref(sg), None, None) == 0 and L.mik_dev_mgs_resident_shape(C.byref(th), C.byref(rr), C.byref(rl)) == 0 \
                    and sl.value == 1 and sg.value > 8:
                spr = th.value // 256
                rounds = -(-sg.value // spr)
                g_share = max(0.0, 1.0 - (rr.value + rl.value) / rounds)
                form = {"form": "one launch per Arnoldi column, w resident in registers / LDS (k_mgs_resident)", "segments_per_workgroup": sg.value,
                        "rounds_per_workgroup": rounds, "rounds_in_registers": rr.value, "rounds_in_lds": rl.value, "resident_share_of_w": 1.0 - g_share}
                moved = sum(spmv_bytes + (2 * k + 2 + (2 * k + 1) * g_share) * n * s8 for k in ks)
        if form is None:
            moved = sum(spmv_bytes + ((4 * k + 3) if name == "mgs" else (2 * k + 6)) * n * s8 for k in ks)
        per_restart = [(min(restart, iters - c * restart) + 2) * n * s8 + spmv_bytes + 3 * n * s8 for c in range(cycles)]
        alg += sum(per_restart)
        moved += sum(per_restart)
        us = best / max(iters, 1) * 1e6
        rec = {"us_per_inner_iteration": us, "inner_iterations": iters, "calls_timed": reps, "mv_products": int(it.mv_products), "restart_cycles": cycles,
               "algorithmic_bytes_per_inner_iteration": alg / max(iters, 1), "achieved": alg / best / 1e9, "unit": "GB/s", "peak": HBM_PEAK_GBS,
               "frac": alg / best / 1e9 / HBM_PEAK_GBS, "frac_of_copy_ceiling_6290": alg / best / 1e9 / COPY_CEILING_GBS,
               "bytes_moved_per_inner_iteration": moved / max(iters, 1), "moved_gbs": moved / best / 1e9, "moved_frac": moved / best / 1e9 / HBM_PEAK_GBS,
               "final_residual": float(hist[-1]) if iters else None}
        if form is not None:
            rec["orthogonalisation"] = form
        tr = gmres_large_traffic(name)
        if tr is not None:
            rec["traffic_per_inner_iteration"] = tr["bytes"]
            rec["traffic_source"] = tr["source"]
            rec["traffic_binary_matches"] = tr["matches"]
            rec["traffic_is"] = "committed constant: sum over the kernels of one profiled call of 2 x FETCH_SIZE + WRITE_SIZE (separate rocprofv3 --pmc passes), divided by its inner iterations"
        out[name] = rec
        del it
    return out


def fixtures_shadow(n: int, j: int):
    """shadow vector j of the IDR(s) leg: the hashed right-hand side generator shifted into (0, 1), a different stream per column (the reference draws
    rand!, src/idrs.jl:136)"""
    pkg = graft.load_package()
    v = pkg.fixtures.hashed_rhs(n) + 0.5
    return np.roll(v, 7919 * (j + 1))


def f_solvers(A, b, n: int, iters: int = 40, extras: bool = False):
    """SURVEY.md 8f rows on the driver's line: per-iteration wall time of PCG with a Jacobi Pl (src/cg.jl:72-100), Chebyshev
    (src/chebyshev.jl:29-57), MINRES (src/minres.jl:95-159) and BiCGStab(2) (src/bicgstabl.jl:79-134; per OUTER iteration = 4 SpMV) on the 256^3
    operator, fp64, one host-visible residual per iteration, in the operator's default layout and on its plain CSR arrays.  `bytes_moved` = what the
    launches of one iteration stream (the layout's SpMV bytes x SpMVs + the words per row of its fused sweeps); `frac` = that over 8 TB/s."""
    import gc
    import torch
    pkg = graft.load_package()
    vec = 8 * n
    out = {}
    for layout in ("auto", "csr"):
        A.set_layout(layout)
        spmv_moved = A.spmv_stored_bytes()
        rec = {"operator_layout": A.layout(), "spmv_kernel": A.spmv_kernel()}

        def timed(it, start, mv, words, count):
            i = start
            for _ in range(4):
                _, i = it.iterate(i)
            gc.collect()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(count):
                _, i = it.iterate(i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / count
            moved = mv * spmv_moved + words * vec
            return {"us_per_iteration": dt * 1e6, "spmv_per_iteration": mv, "vector_words_per_row_moved": words, "bytes_moved": moved,
                    "gbs": moved / dt / 1e9, "frac": moved / dt / 1e9 / HBM_PEAK_GBS}
        d = pkg.HipVector.from_numpy(np.full(n, 6.0))
        rec["pcg_jacobi"] = timed(pkg.cg_iterator_(pkg.zerox(A, b), A, b, pkg.JacobiPrec(d), reltol=0.0, initially_zero=True, maxiter=10 ** 9), 0, 1, 10, iters)
        rec["chebyshev"] = timed(pkg.chebyshev_iterable_(pkg.zerox(A, b), A, b, 4.5e-4, 12.0, reltol=0.0, initially_zero=True, maxiter=10 ** 9), 0, 1, 9, iters)
        mit = pkg.minres_iterable_(pkg.zerox(A, b), A, b, reltol=0.0, initially_zero=True, maxiter=10 ** 9)
        ep = mit.proj_shape() == pkg.default_context().spmv_dot_shape()        # the Lanczos step rides on the SpMV launch
        rec["minres"] = timed(mit, 1, 1, 12 if ep else 15, iters)
        bit = pkg.bicgstabl_iterator_(pkg.zerox(A, b), A, b, 2, reltol=0.0, max_mv_products=10 ** 9, initial_zero=True)
        epb = bit.dot_shape() == pkg.default_context().spmv_dot_shape()        # sigma / rho leave the SpMV launches
        ll = 2
        words = sum((1 if epb and j else (0 if j == 0 else 2)) + 3 * (j + 1) + (1 if epb else 2) + 3 * (j + 1) + 3 for j in range(ll)) + (ll + 1) + (3 * ll + 4) + 1
        rec["bicgstab2_per_outer_iteration"] = timed(bit, 0, 2 * ll, words, max(iters // 3, 10))
        if extras:
            # --extras only (outside SURVEY section 8, unjudged): IDR(8) (src/idrs.jl:164-272), average over whole cycles of s + 1 = 9 steps, each one
            # SpMV.  Words per row of a cycle: step k (0-based, cnt = s - k): direction sweep 2 cnt + 2, bi-orthogonalisation 2 + 7 k - 1 (k > 0),
            # batched dot cnt + 1, update 6; the polynomial step: 2 + 5
            ss = 8
            cyc = sum(2 * (ss - k) + 2 + ((2 + 7 * k - 1) if k else 0) + (ss - k) + 1 + 6 for k in range(ss)) + 7
            Pm = pkg.HipMatrix(n, ss, np.float64)
            for j in range(ss):
                Pm.col(j).copy_from_host(fixtures_shadow(n, j))
            iit = pkg.extras.idrs_iterable_(None, pkg.zerox(A, b), A, b, ss, None, 0.0, 0.0, 10 ** 9, P=Pm)
            rec["idrs8_per_step"] = timed(iit, (1, 1), 1, cyc / (ss + 1), 4 * (ss + 1))
            del iit, Pm
        out["default_layout" if layout == "auto" else "csr_arrays"] = rec
        del d, mit, bit
    A.set_layout("auto")
    return out


def adjoint_solvers(A, csc, b, n: int, iters: int = 20):
    """LSQR, LSMR, QMR (src/lsqr.jl, src/lsmr.jl, src/qmr.jl) per iteration on the 256^3 operator, default layout: two SpMV per iteration, with A and
    with adjoint(A) -- a second upload of the SAME CSC arrays as CSR (no transpose formed) -- and their fused sweeps; several host-visible norms per
    iteration like the reference's loops.  Time of (iters + 3) iterations minus that of 3 (set-up cancelled)."""
    import gc
    import torch
    pkg = graft.load_package()
    t0 = time.perf_counter()
    A.adj = pkg.HipCSR(n, n, *csc, index_base=1, is_csc=False)
    A.adj.adj = A
    rec = {"adjoint_upload_seconds": time.perf_counter() - t0, "operator_layout": A.layout(), "adjoint_layout": A.adj.layout()}
    spmv = A.spmv_stored_bytes() + A.adj.spmv_stored_bytes()

    def run(name, fn, words):
        fn(3)
        gc.collect()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fn(iters + 3)
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t1
        t1 = time.perf_counter()
        fn(3)
        torch.cuda.synchronize()
        dt = (t_all - (time.perf_counter() - t1)) / iters
        moved = spmv + words * 8 * n
        rec[name] = {"us_per_iteration": dt * 1e6, "vector_words_per_row_moved": words, "bytes_moved": moved, "frac": moved / dt / 1e9 / HBM_PEAK_GBS}
    run("lsqr", lambda k: pkg.extras.lsqr(A, b, maxiter=k, atol=0.0, btol=0.0, conlim=0.0), 15)
    run("lsmr", lambda k: pkg.extras.lsmr(A, b, maxiter=k, atol=0.0, btol=0.0, conlim=0.0), 17)
    run("qmr", lambda k: pkg.extras.qmr(A, b, maxiter=k, reltol=0.0), 21)
    A.adj.adj = None
    A.adj = None
    return rec


def gmres_large_traffic(name: str):
    """HBM-side bytes per inner iteration of the HBM-bound GMRES leg from the committed PMC passes (profiles/*gmres_large_<name>_traffic.json)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*gmres_large_{name}_traffic.json")), reverse=True):
        try:
            t = json.load(open(f))
            ok = traffic_file_matches(t)
            return {"bytes": t["traffic_bytes_per_inner_iteration"] if ok else None, "source": os.path.relpath(f, ROOT), "matches": ok}
        except Exception:
            continue
    return None


def c5_traffic(kind: str):
    """PMC traffic per SpMV launch of a configs[4] stand-in from the committed rocprofv3 passes (profiles/*c5_<kind>_traffic.json,
    written by scripts/prof_r03.sh + prof_collect.py from separate --pmc runs of scripts/config5_bench.py on that one matrix)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*c5_{kind}_traffic.json")), reverse=True):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        cand = [v for k2, v in t.items() if k2.startswith("k_spmv") and v.get("traffic_bytes_per_launch")]
        if cand:
            ok = traffic_file_matches(t)
            return {"traffic": max(c["traffic_bytes_per_launch"] for c in cand) if ok else None, "traffic_source": os.path.relpath(f, ROOT), "traffic_binary_matches": ok}
    return {"traffic": None, "traffic_source": None, "traffic_binary_matches": None}


def config5(kinds):
    """BASELINE.json configs[4]: gmres!(restart = 50), fp32, on an unstructured CSR matrix (benchmark/matrixmarket.jl:5-22 names
    s3dkq4m2; no SuiteSparse file exists offline).  Stand-ins: `fe_shell` (s3dkq4m2's shape: 6 unknowns per node, 9-node
    neighbourhoods, n = 90,774), `fe_hex` (3 unknowns per node, 27-node neighbourhoods, 62 M entries: larger than the Infinity
    Cache), and SURVEY.md 8d's synthetic irregular matrix (n = 10^6, rows of 5-15 / 50-200 / 5,000-20,000 entries) with `banded`
    and with uniformly `random` columns.  SpMV: back-to-back HIP-event time, achieved = the CSR-algorithmic bytes of SURVEY.md
    8d over that time.  GMRES(50): wall time per inner iteration (loop inside the library) with MGS and CGS."""
    import torch
    pkg = graft.load_package()
    out = {}
    for kind in kinds:
        t0 = time.perf_counter()
        if kind == "fe_shell":
            n, rowptr, colidx, val = pkg.fixtures.fe_matrix((123, 123), 6, np.float32)
        elif kind == "fe_hex":
            n, rowptr, colidx, val = pkg.fixtures.fe_matrix((64, 64, 64), 3, np.float32)
        else:
            n, rowptr, colidx, val = pkg.fixtures.irregular_matrix(1_000_000, np.float32, bandwidth=0 if kind == "random" else 2000)
        tgen = time.perf_counter() - t0
        lens = np.diff(rowptr)
        t0 = time.perf_counter()
        A = pkg.HipCSR(n, n, rowptr, colidx, val, index_base=0, is_csc=False)
        tup = time.perf_counter() - t0
        del rowptr, colidx, val
        b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n, dtype=np.float32))
        y = pkg.HipVector(n, np.float32)
        A.time_spmv(b, y, reps=3)
        ms = A.time_spmv(b, y, reps=30)
        ab, sb = A.spmv_algorithmic_bytes(), A.spmv_stored_bytes()
        rec = {"n": int(n), "nnz": int(A.nnz), "row_length_min_median_max": [int(lens.min()), int(np.median(lens)), int(lens.max())],
               "rows_longer_than_256": int((lens > 256).sum()), "operator_layout": A.layout(), "kernel": A.spmv_kernel() + "<float>",
               "spmv_us": ms * 1e3, "algorithmic_bytes_per_launch": ab, "bytes_stored_per_launch": sb,
               "achieved": ab / (ms * 1e-3) / 1e9, "unit": "GB/s", "peak": HBM_PEAK_GBS, "frac": ab / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
               **c5_traffic(kind), "generate_seconds": tgen, "upload_seconds": tup}
        if ab < 200e6:
            rec["note"] = "operator smaller than the 256 MB Infinity Cache: back-to-back launches are served on-die, `frac` is not an HBM fraction"
        gm = {}
        for oname, M in (("mgs", pkg.ModifiedGramSchmidt()), ("cgs", pkg.ClassicalGramSchmidt())):
            best = None
            for _rep in range(2):
                it = pkg.gmres_iterable_(pkg.zerox(A, b), A, b, restart=50, orth_meth=M, initially_zero=True, reltol=0.0, maxiter=150)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                h = it.iterate_many(0, 150)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / max(h.size, 1)
                best = dt if best is None else min(best, dt)
            gm[oname] = best * 1e6
        rec["gmres50_us_per_inner_iteration"] = gm
        rec["gmres50_minus_spmv_us"] = {k2: v2 - ms * 1e3 for k2, v2 in gm.items()}
        out[kind] = rec
        del A, b, y
    return out


def stencil27(grid: int, steps: int):
    """cg! on the constant-coefficient 27-point box stencil, grid^3, fp64 (fixtures.box_stencil_matrix) -- the structured fast path
    beyond the reference's own 7-point fixture: up to 32 offsets per slice, one 32-bit presence mask per row, two
    rows per lane (k_spmv_sdiaw2), chosen automatically.  Reports the loop in that layout and on the plain CSR arrays of the same
    operator, and that both give the same residual history bit for bit."""
    import torch
    pkg = graft.load_package()
    t0 = time.perf_counter()
    n, rowptr, colidx, val = pkg.fixtures.box_stencil_matrix(grid, 3, np.float64)
    tgen = time.perf_counter() - t0
    t0 = time.perf_counter()
    A = pkg.HipCSR(n, n, rowptr, colidx, val, index_base=0, is_csc=False)
    tup = time.perf_counter() - t0
    del rowptr, colidx, val
    b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n))

    def loop(Aop):
        it = pkg.cg_iterator_(pkg.zerox(Aop, b), Aop, b, reltol=0.0, initially_zero=True, maxiter=10 ** 9)
        hist = [it.iterate(k)[0] for k in range(10)]
        it.profile(1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(10, 10 + steps):
            it.iterate(k)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / steps
        ms, cnt = it.profile(0)
        return dt, ms / max(cnt, 1), hist

    dt, spmv_ms, h1 = loop(A)
    sb, ab = A.spmv_stored_bytes(), A.spmv_algorithmic_bytes()
    out = {"workload": f"cg! on the constant-coefficient 27-point box stencil, {grid}^3, fp64", "n": int(n), "nnz": int(A.nnz), "operator_layout": A.layout(),
           "kernel": A.spmv_kernel() + "<double, fused dot>", "iters_per_sec": 1 / dt, "ms_per_step": dt * 1e3, "spmv_in_loop_ms": spmv_ms,
           "bytes_moved_per_launch": sb, "achieved_moved": sb / (spmv_ms * 1e-3) / 1e9, "frac_moved": sb / (spmv_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "algorithmic_bytes_per_launch": ab, "generate_seconds": tgen, "upload_seconds": tup,
           "note": "not HBM-bound: ~540 instructions and 20 vector-memory instructions per 128 rows, x re-read nine times from L2; PMC analysis in profiles/r03_s27_pmc_summary.txt, DESIGN.md section 5"}
    A.set_layout("csr")
    dt2, ms2, h2 = loop(A)
    out["csr_layout"] = {"kernel": A.spmv_kernel() + "<double, fused dot>", "iters_per_sec": 1 / dt2, "spmv_in_loop_ms": ms2,
                         "achieved": ab / (ms2 * 1e-3) / 1e9, "frac": ab / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS}
    out["same_history_in_both_layouts"] = bool(h1 == h2)
    return out


def run_single(args):
    import torch
    pkg = graft.load_package()
    N, K, Wm = args.n, args.steps, args.warmup
    ctx = pkg.default_context()
    n, colptr, rowval, nzval = pkg.fixtures.laplace_matrix(N, 3)
    t_up = time.perf_counter()
    A = pkg.HipCSR(n, n, colptr, rowval, nzval, index_base=1)
    upload_seconds = time.perf_counter() - t_up
    csc = (colptr, rowval, nzval) if (args.extras and not args.no_f_solvers and N >= 128) else None     # --extras: kept for the adjoint operator of the LSQR / LSMR / QMR leg
    del colptr, rowval, nzval
    nnz = A.nnz
    b = pkg.HipVector.from_numpy(pkg.fixtures.hashed_rhs(n))
    sync = torch.cuda.synchronize

    # ---- (1) the full solve to the default tolerance: parity of the WHOLE residual history -------------------
    parity = None
    if not args.no_parity:
        t0 = time.perf_counter()
        xs, ch = pkg.cg(A, b, log=True)
        sync()
        solve_seconds = time.perf_counter() - t0
        parity = history_parity(N, np.asarray(ch["resnorm"]), ch.iters, ch.mvps, ch.isconverged)
        if parity is not None:
            parity["solve_seconds"] = solve_seconds
        del xs

    # ---- (2) the timed loop: reltol = 0 so that every step does identical work ------------------------------
    def timed_loop(Aop):
        it = pkg.cg_iterator_(pkg.zerox(Aop, b), Aop, b, reltol=0.0, initially_zero=True, maxiter=10 ** 9)
        state = {"k": 0}

        def step():
            nxt = it.iterate(state["k"])          # one host-visible residual per step, like the reference loop
            assert nxt is not None
            state["k"] += 1
        for _ in range(Wm):
            step()
        it.profile(1)                             # HIP events on the ctx stream around every SpMV launch of the loop
        times = timed_regions(step, K, sync)
        prof = it.profile(0)

        def batch():
            r = it.iterate_many(state["k"], 25)
            assert r.size == 25
            state["k"] += 25
        tb = timed_regions(batch, max(1, K // 25), sync)
        return it, times, prof, tb, max(1, K // 25) * 25

    alg_bytes = A.spmv_algorithmic_bytes()         # nnz*(s+4) + (n+1)*4 + 2*n*s
    layout = A.layout()
    it, times, (spmv_ms_total, spmv_launches), tb, kb = timed_loop(A)
    dt = float(np.median(times))
    spmv_ms = spmv_ms_total / max(spmv_launches, 1)
    stored_bytes = A.spmv_stored_bytes()
    kern = A.spmv_kernel()
    u = pkg.HipVector.wrap(it.u.ptr, n, np.float64, ctx, owner=it.u)
    scratch = pkg.HipVector(n)
    A.time_spmv(u, scratch, reps=3, fused_dot=True)
    b2b_ms = A.time_spmv(u, scratch, reps=20, fused_dot=True)
    residual = it.residual
    # every streaming launch of the step bracketed by HIP events (a separate short loop: 6 events per step perturb the rate)
    it.profile(2)
    k0 = 10 ** 6
    for j in range(60):
        assert it.iterate(k0 + j) is not None
    sync()
    pk = it.profile_kernels()
    it.profile(0)
    fx = it.fused_x()       # x .+= alpha u rides on the next u = r + beta u sweep: 5 n + 3 n words per step instead of 3 n + 6 n
    vec_bytes = {"xpby": (5 if fx else 3) * n * 8, "update": (3 if fx else 6) * n * 8}
    step_kernels = [{"kernel": kern + "<double, fused dot>", "what": "c = A u + partial dot(u, c)  (src/cg.jl:54-55)",
                     "avg_launch_ms": pk["spmv"][0] / max(pk["spmv"][1], 1), "bytes_moved": stored_bytes},
                    {"kernel": "k_map<OpXpbyX>" if fx else "k_map<OpXpby>",
                     "what": ("x += alpha u of the previous step (src/cg.jl:58), then u = r + beta u  (src/cg.jl:51)" if fx else "u = r + beta u  (src/cg.jl:51)"),
                     "avg_launch_ms": pk["xpby"][0] / max(pk["xpby"][1], 1), "bytes_moved": vec_bytes["xpby"]},
                    {"kernel": "k_map<OpCgUpdateR>" if fx else "k_map<OpCgUpdate>",
                     "what": ("r -= alpha c; |r|^2  (src/cg.jl:59-62)" if fx else "x += alpha u; r -= alpha c; |r|^2  (src/cg.jl:58-62)"),
                     "avg_launch_ms": pk["update"][0] / max(pk["update"][1], 1), "bytes_moved": vec_bytes["update"]}]
    for kk in step_kernels:
        kk["gbs"] = kk["bytes_moved"] / (kk["avg_launch_ms"] * 1e-3) / 1e9
        kk["frac_of_8000"] = kk["gbs"] / HBM_PEAK_GBS
        kk.update(traffic_fields(kk["kernel"].split("<")[0] if kk["kernel"].startswith("k_spmv") else kk["kernel"]))
        if kk["gbs"] > COPY_CEILING_GBS:
            kk["note"] = "above the 6,290 GB/s HBM copy ceiling: part of this sweep is served by the Infinity Cache (the previous launch left it there); frac_of_8000 is then not an HBM fraction"
    del it

    # ---- (3) the CONTRACT loop: the same operator on its plain CSR arrays (the north star's "CSR SpMV inside cg!") --------
    # mik_csr_set_layout(A, 0): Int32 CSR, k_spmv_rowgather; bytes moved = the CSR-algorithmic bytes of SURVEY.md 8d.
    csr = None
    if layout != "csr-rowblock" and not args.no_csr:
        A.set_layout("csr")
        try:
            it2, times2, (ms2, n2), tb2, kb2 = timed_loop(A)
            c_ms = ms2 / max(n2, 1)
            dt2 = float(np.median(times2))
            u2 = pkg.HipVector.wrap(it2.u.ptr, n, np.float64, ctx, owner=it2.u)
            A.time_spmv(u2, scratch, reps=3, fused_dot=True)
            c_b2b = A.time_spmv(u2, scratch, reps=20, fused_dot=True)
            csr = {"kernel": A.spmv_kernel(), "iters_per_sec": K / dt2, "ms_per_step": dt2 / K * 1e3, "spmv_in_loop_ms": c_ms, "launches_timed": int(n2),
                   "spmv_back_to_back_ms": c_b2b, "batched_25_steps_per_sync_iters_per_sec": kb2 / float(np.median(tb2)),
                   "timed_regions": len(times2), "final_residual": it2.residual}
            del it2, u2
        finally:
            A.set_layout("auto")
    elif layout == "csr-rowblock":
        csr = {"kernel": kern, "iters_per_sec": K / dt, "ms_per_step": dt / K * 1e3, "spmv_in_loop_ms": spmv_ms, "launches_timed": int(spmv_launches),
               "spmv_back_to_back_ms": b2b_ms, "batched_25_steps_per_sync_iters_per_sec": kb / float(np.median(tb)), "timed_regions": len(times),
               "final_residual": residual}

    moved_gbs = stored_bytes / (spmv_ms * 1e-3) / 1e9
    iter_moved = stored_bytes + (8 if fx else 9) * n * 8          # bytes ONE step of the loop behind `value` moves (its SpMV + the two sweeps)
    iter_alg = alg_bytes + 9 * n * 8                               # B_cg = B_spmv + 9 n s
    d_tf = traffic_fields(kern)
    default_spmv = {"kernel": kern + "<double, fused dot>", "operator_layout": layout, "bytes_moved_per_launch": stored_bytes, "avg_launch_ms": spmv_ms,
                    "launches_timed": int(spmv_launches), "back_to_back_ms": b2b_ms, "achieved_moved": moved_gbs, "frac_moved": moved_gbs / HBM_PEAK_GBS,
                    "frac_of_copy_ceiling_6290": moved_gbs / COPY_CEILING_GBS, **d_tf,
                    "note": "the SpMV of the loop behind `default_layout_iters_per_sec`: this constant-coefficient operator keeps ONE mask byte per row instead of 57 B of values "
                            "and indices, so the launch moves 17 B per row instead of 73; achieved_moved / frac_moved = the bytes it actually streams "
                            "(`traffic`: committed PMC constant) over its HIP-event time inside the loop.  Not the north star's CSR figure: that is `roofline`."}
    if csr is not None:
        c_ms = csr["spmv_in_loop_ms"]
        c_tf = traffic_fields(csr["kernel"])
        roofline = {"bound": "hbm", "kernel": csr["kernel"] + "<double, fused dot>", "loop": "contract_csr_loop" if layout != "csr-rowblock" else "the timed loop",
                    "achieved": alg_bytes / (c_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / (c_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    **c_tf, "libmik_sha256": loaded_library_sha256(),
                    "traffic_is": "committed constant from separate rocprofv3 --pmc passes of this command on the binary named by libmik_sha256 (traffic_binary_matches), not measured in this run",
                    "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": c_ms, "launches_timed": csr["launches_timed"],
                    "back_to_back_ms": csr["spmv_back_to_back_ms"], "frac_of_copy_ceiling_6290": alg_bytes / (c_ms * 1e-3) / 1e9 / COPY_CEILING_GBS,
                    # the loop this kernel was timed in -- so that bytes/step / ms_per_step <= peak and avg_launch_ms <= ms_per_step can be checked from here alone
                    "loop_ms_per_step": csr["ms_per_step"], "loop_iters_per_sec": csr["iters_per_sec"],
                    "loop_algorithmic_bytes_per_step": iter_alg, "loop_gbs": iter_alg / (csr["ms_per_step"] * 1e-3) / 1e9,
                    "loop_frac": iter_alg / (csr["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "target": 0.60,
                    "note": "SURVEY.md 8d: algorithmic bytes of the Int32 CSR SpMV (nnz*(s+4) + (n+1)*4 + 2*n*s = 1,740,111,876 B at 256^3 fp64) over the "
                            "average HIP-event duration of the SpMV launch INSIDE the cg! loop, on the plain CSR arrays of the operator "
                            "(mik_csr_set_layout(A, 0); k_spmv_rowgather moves exactly those bytes: `traffic`).  loop_* = that CSR loop itself (the contract "
                            "rate: cg! iterations/s moving B_spmv + 9 n s per step) = `value`.  `default_layout_iters_per_sec` is the SAME iteration, "
                            "bit-identical results, in the operator's default layout, which moves config.default_layout.bytes_per_step instead (`default_layout_spmv`)."}
    else:
        roofline = {"bound": "hbm", "kernel": kern + "<double, fused dot>", "loop": "the timed loop (contract CSR loop skipped: --no-csr)",
                    "achieved": moved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": moved_gbs / HBM_PEAK_GBS, **d_tf,
                    "bytes_moved_per_launch": stored_bytes, "avg_launch_ms": spmv_ms,
                    "loop_ms_per_step": dt / K * 1e3, "loop_iters_per_sec": K / dt,
                    "note": "bytes this layout actually moves per launch over the in-loop HIP-event time; NOT the CSR-algorithmic figure"}
    # `value` = the CONTRACT loop (BASELINE.json configs[1]: "HIP CSR SpMV + fused dot/axpy"): cg! on the operator's plain Int32 CSR arrays, which
    # moves SURVEY.md 8d's B_cg = B_spmv + 9 n s per step -- so value_bytes_per_step / ms_per_step <= 8 TB/s can be checked from the top level alone.
    # The same iteration in the layout mik_csr_create picks by itself for this constant-coefficient operator (one mask byte per row,
    # bit-identical results) is reported next to it as `default_layout_iters_per_sec`.
    value_is_csr = csr is not None
    v_ips, v_ms = (csr["iters_per_sec"], csr["ms_per_step"]) if value_is_csr else (K / dt, dt / K * 1e3)
    v_regions = csr["timed_regions"] if value_is_csr else len(times)
    out = {
        "metric": "cg_iters_per_sec", "value": v_ips, "unit": "iters/s", "n_gpus": 1, "steps": K, "warmup": Wm,
        "ms_per_step": v_ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "value_is_contract": bool(value_is_csr),
        "value_bytes_per_step": iter_alg if value_is_csr else iter_moved,
        "value_gbs": (iter_alg if value_is_csr else iter_moved) / (v_ms * 1e-3) / 1e9,
        "default_layout_iters_per_sec": K / dt, "default_layout_ms_per_step": dt / K * 1e3,
        "config": {"workload": f"cg! on {N}^3 3D 7-point Laplacian (test/laplace_matrix.jl), fp64, hashed rhs, x0 = 0 "
                               f"(BASELINE.json configs[1])", "n": n, "nnz": nnz, "reltol_in_timed_loop": 0.0, "host_sync_per_step": 1,
                   "operator_layout_of_the_timed_loop": "csr (mik_csr_set_layout(A, 0): Int32 rowptr / col / val, k_spmv_rowgather)" if value_is_csr else layout,
                   "timed_regions": v_regions,
                   "operator_upload_seconds": upload_seconds, "final_residual": csr["final_residual"] if value_is_csr else residual,
                   "machine": {k: v for k, v in ctx.info().items() if k in ("arch", "compute_units", "xcds", "lds_bytes_per_cu", "l2_bytes", "hbm_bytes", "xcd_maps",
                                                                             "resident_workgroup_cap", "gs_single_launch_max_segments", "mgs_resident_max_segments")},
                   "default_layout": {"operator_layout": layout, "iters_per_sec": K / dt, "ms_per_step": dt / K * 1e3,
                                      "bytes_per_step": iter_moved, "gbs": iter_moved / (dt / K) / 1e9,
                                      "frac_of_8000": iter_moved / (dt / K) / 1e9 / HBM_PEAK_GBS, "spmv_avg_launch_ms": spmv_ms,
                                      "timed_regions": len(times), "timed_seconds_total": float(sum(times)),
                                      "region_seconds_min_max": [float(min(times)), float(max(times))], "final_residual": residual,
                                      "note": "what cg! does on this input when the layout is left to mik_csr_create: the SpMV streams one mask byte per row "
                                              "instead of the CSR arrays (constant-coefficient operator), same residual history bit for bit; NOT the contract figure"}},
        "contract_csr_loop": csr,
        "roofline": roofline,
        "default_layout_spmv": default_spmv,
        "longest_kernel_of_the_step": (lambda kk: {"kernel": kk["kernel"], "avg_launch_ms": kk["avg_launch_ms"], "bytes_moved": kk["bytes_moved"],
                                                   "frac_of_8000": kk["frac_of_8000"], "traffic": kk["traffic"], "traffic_source": kk["traffic_source"],
                                                   "traffic_binary_matches": kk["traffic_binary_matches"],
                                                   **({"note": kk["note"]} if "note" in kk else {})})(max(step_kernels, key=lambda q: q["avg_launch_ms"])),
        "step_kernels": step_kernels,
        "cg_iteration_moved_bytes": iter_moved, "cg_iteration_moved_gbs": iter_moved / (dt / K) / 1e9,
        "cg_iteration_moved_frac_of_8000": iter_moved / (dt / K) / 1e9 / HBM_PEAK_GBS,
        "cg_iteration_algorithmic_bytes": iter_alg,
        "contract_cg_iteration_gbs": (iter_alg / (csr["ms_per_step"] * 1e-3) / 1e9) if csr else None,
        "batched_25_steps_per_sync_iters_per_sec": kb / float(np.median(tb)),
        "parity_full_history": parity,
    }
    if not args.no_gmres_large and N >= 128:
        A.set_layout("csr")
        try:
            out["gmres_hbm_bound"] = gmres_hbm_bound(A, b, n)
        finally:
            A.set_layout("auto")
        # the same call in the layout mik_csr_create picks (mask bytes instead of the CSR arrays): reported beside it, not a roofline figure
        dl = gmres_hbm_bound(A, b, n, reps=2)
        out["gmres_hbm_bound"]["default_layout"] = {"operator_layout": dl["operator_layout"], "spmv_kernel": dl["spmv_kernel"],
                                                    **{m: {"us_per_inner_iteration": dl[m]["us_per_inner_iteration"], "final_residual": dl[m]["final_residual"]} for m in ("mgs", "cgs")},
                                                    "same_residual_as_csr": bool(all(dl[m]["final_residual"] == out["gmres_hbm_bound"][m]["final_residual"] for m in ("mgs", "cgs")))}
    if not args.no_f_solvers and N >= 128:
        try:
            out["f_solvers"] = f_solvers(A, b, n, extras=args.extras)
        finally:
            A.set_layout("auto")
        if args.extras:                # outside SURVEY section 8: never on the default line
            try:
                out["extras"] = {"note": "solvers outside the scope contract (SURVEY section 2: OUT OF SCOPE); reported only with --extras",
                                 "adjoint_solvers": adjoint_solvers(A, csc, b, n)}
            except Exception as e:     # noqa: BLE001
                out["extras"] = {"error": repr(e)[:300]}
            csc = None
    del A, b, scratch, u
    if not args.no_gmres:
        out["gmres_config3"] = gmres_config3()
    if not args.no_config5:
        out["config5"] = config5([k for k in args.config5_kinds.split(",") if k])
    if args.stencil27 > 0:
        out["stencil27"] = stencil27(args.stencil27, 60)
    if not args.no_cpu_baseline:
        cb = cpu_baseline(N, args.cpu_iters)
        cpu_hist = cb.pop("_history")
        if parity is not None and cpu_hist.size:
            # the oracle run of THIS process (SEQ order) against the first steps of the GPU solve
            path = os.path.join(ROOT, "tests", "golden", f"cg_lap{N}.json")
            g_tree = np.array([float.fromhex(s) for s in json.load(open(path))["tree"]["resnorm"]])
            if parity.get("tree", {}).get("bit_identical"):
                m = min(cpu_hist.size, g_tree.size)
                cb["gpu_vs_cpu_history_max_rel_dev"] = float(np.max(np.abs(g_tree[:m] - cpu_hist[:m]) / cpu_hist[:m]))
                cb["history_steps_compared"] = int(m)
                cb["gpu_vs_cpu_history_order"] = "SEQ (one accumulator left to right over 16.7 M terms: this run of the oracle); its own distance to the BLAS order is cpu_seq_vs_blas_floor"
            # the order the reference itself executes (LinearAlgebra.dot / norm -> OpenBLAS), over the WHOLE solve: committed golden history
            for key, name in (("blas", "gpu_vs_blas_order_history_max_rel_dev"), ("blas8", "gpu_vs_blas_8_threads_history_max_rel_dev"), ("pair", "gpu_vs_pairwise_history_max_rel_dev")):
                if key in parity:
                    cb[name] = parity[key]["gpu_vs_cpu_history_max_rel_dev"]
                    cb[name.replace("max_rel_dev", "steps_compared")] = parity[key]["steps_compared"]
            if "blas" in parity:
                cb["same_iters_mvps_isconverged_as_blas_order"] = parity["blas"]["same_iters_mvps_isconverged"]
                cb["parity_target"] = 1e-12
                cb["cpu_seq_vs_blas_floor"] = parity.get("cpu_vs_cpu_floors", {}).get("seq_vs_blas")
        out["cpu_baseline"] = cb
        try:
            out["cpu_baseline_omp"] = cpu_baseline_omp(N, args.cpu_iters)
        except Exception as e:                      # OpenMP runtime missing on the box: the serial baseline above stands
            out["cpu_baseline_omp"] = {"error": str(e)[:200]}
    print(json.dumps(out))


def partition_oracle(N: int, NZ: int, offsets, shape):
    """The checker of the N > 1 line's `parity_vs_oracle` (rank 0 only): the oracle's cg! in TREE mode with the run's row partition
    (rank-ordered sums of the per-rank trees) on the N x N x NZ Laplacian with the hashed rhs, to the default tolerance."""
    from importlib import import_module
    orc = graft.load_oracle()
    pkg = graft.load_package()
    d = import_module(pkg.__name__ + ".dist")
    n, ptr, idx, val = d._laplace_rows(pkg, N, NZ, 0, N * N * NZ, np.float64)
    A = orc.CSC(n, ptr, idx, val, 0)                 # symmetric: the CSR arrays are a valid CSC
    b = pkg.fixtures.hashed_rhs(n)
    orc.set_partition(np.asarray(offsets, np.int64))
    try:
        x, h = orc.cg(A, b, mode="tree", shape=shape)
    finally:
        orc.set_partition(None)
    return {"iters": int(h["iters"]), "isconverged": bool(h["isconverged"]), "resnorm": np.asarray(h["resnorm"]), "x": x}


def run_partitioned(args):
    from importlib import import_module
    pkg = graft.load_package()
    args.pmc_traffic = pmc_traffic
    args.cpu_baseline_fn = cpu_baseline
    args.partition_oracle_fn = partition_oracle
    return import_module(pkg.__name__ + ".bench_dist").bench_main(args)


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks under torch.distributed.run ourselves."""
    import socket
    import torch
    have = torch.cuda.device_count()
    if have < args.gpus and "MIK_FORCE_DEVICE" not in os.environ:      # (development: MIK_FORCE_DEVICE puts every rank on one GPU -- mailbox transport only)
        sys.exit(f"bench.py: {args.gpus} ranks requested, {have} device(s) visible -- one rank per GPU is required "
                 f"(RCCL refuses two ranks on one device); run with --gpus <= {max(have, 1)}")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    got_line, rc = False, None
    try:
        if os.environ.get("MIK_SPAWN_FAIL") == "1":
            raise OSError("spawn failure simulated by MIK_SPAWN_FAIL=1 (development)")
        proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
        for line in proc.stdout:
            got_line = got_line or (line.startswith("{") and '"metric"' in line)
            sys.stdout.write(line)
            sys.stdout.flush()
        rc = proc.wait()
    except OSError as e:
        print(f"bench.py: could not start the ranks ({e})", file=sys.stderr)
    if got_line:
        raise SystemExit(0 if rc == 0 else rc)
    # No line: the ranks could not be started, or died before anything was measured.  This process measures the partitioned system
    # itself through the in-process group (one host thread, every rank's slab on its own device): the line is still contract-complete.
    print(f"bench.py: the {args.gpus}-rank launch produced no line (exit code {rc}); measuring through the in-process group", file=sys.stderr)
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE=str(args.gpus), MIK_BOOT_FAIL="1")
    run_partitioned(args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid", dest="n", type=int, default=None, help="grid points per dimension (default: 256 at N = 1, 512 x 512 x 64 N at N > 1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the full solve compared with tests/golden/")
    ap.add_argument("--no-csr", action="store_true", help="skip the contract loop on the plain CSR arrays (roofline then describes the default layout)")
    ap.add_argument("--no-gmres", action="store_true", help="skip the configs[2] sub-benchmark (gmres_config3)")
    ap.add_argument("--no-gmres-large", action="store_true", help="skip the HBM-bound GMRES leg (gmres_hbm_bound: gmres!(restart=30) on the 256^3 operator)")
    ap.add_argument("--no-f-solvers", action="store_true", help="skip the SURVEY 8f solvers leg (f_solvers: PCG / Chebyshev / MINRES / BiCGStab(2) per iteration at 256^3)")
    ap.add_argument("--no-config5", action="store_true", help="skip the configs[4] stand-ins (config5)")
    ap.add_argument("--config5-kinds", default="fe_shell,fe_hex,banded,random")
    ap.add_argument("--stencil27", type=int, default=0, help="grid of the 27-point box-stencil sub-benchmark (off by default: outside every BASELINE.json config; frozen, VERDICT r3 #8)")
    ap.add_argument("--extras", action="store_true", help="also time the solvers OUTSIDE the scope contract (IDR(8) in f_solvers, LSQR / LSMR / QMR): unjudged, off by default")
    ap.add_argument("--cpu-iters", type=int, default=100, help="iterations of the CPU baseline's bounded sample (about 12 s of one EPYC core at 256^3)")
    ap.add_argument("--force-dist", action="store_true", help="run the row-partitioned code path even with one rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "0"))
    if args.gpus > 1 and world == 0:
        return launch_ranks(args)
    if world > 1 or args.gpus > 1 or args.force_dist:
        if world not in (0, args.gpus):
            sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
        return run_partitioned(args)
    if args.n is None:
        args.n = 256
    run_single(args)


if __name__ == "__main__":
    main()
