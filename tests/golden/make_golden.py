"""Regenerates the golden residual histories under tests/golden/ from the CPU oracle.

Julia (hence the reference itself) cannot run in this environment, so these vectors are produced
by oracle/mik_oracle.c -- the restatement pinned against the reference's own known-answer
material in tests/test_oracle_pinning.py.  They pin (a) the oracle against regressions and
(b) the HIP path at sizes where running the oracle inside the GPU test would be too slow.

    python tests/golden/make_golden.py          # rewrites the .json files (takes a few minutes)

`tree` histories depend on the device reduction shape (W, L) = mik_reduce_shape(); it is stored
in each file and the tests skip the bit-exact comparison if the library's shape has changed.

Summation orders per file: `seq` (one accumulator), `pair` (pairwise), `tree` (the device's fixed tree) and
`blas` / `blas8`: dot, norm and gemv evaluated by the host's OpenBLAS (the library SciPy bundles; recorded under
"blas_library") with 1 resp. 8 BLAS threads -- the library family the reference executes for LinearAlgebra.dot /
norm / mul!.  OpenBLAS picks its kernel by CPU (DYNAMIC_ARCH) and splits long vectors across threads, so the
`blas*` histories are those of THIS machine ("core" in the file).  `--blas-only TAG` regenerates only them on another host (or another
forced core type) into cg_lap<N>_blas_<TAG>.json: committed for the GPU box's EPYC 9575F (OpenBLAS 0.3.28 picks its SkylakeX / AVX-512 kernels there: the history equals the build container's bit
for bit) and for the forced Haswell core (AVX2 kernels; OPENBLAS_CORETYPE=Zen selects the same kernels in this build).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import orc  # noqa: E402

W, L = 2, 2          # mik_reduce_shape(MIK_F64) at the time of generation
LD = 1               # mik_spmv_dot_shape() -> (1, LD): the dot(u, c) fused into the CG SpMV


def hexlist(a):
    return [float(v).hex() for v in a]


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=0)
    print("wrote", name)


def blas_modes():
    """(key, oracle mode, BLAS threads): binding happens right before the run that uses it."""
    return (("blas", "blas", 1), ("blas8", "blas", 8))


def cg_case(N, maxiter=None):
    A = orc.laplace(N, 3)
    b = orc.hashed_rhs(A.n)
    out = dict(case=f"cg(laplace_matrix(Float64,{N},3), hashed_rhs) reltol=sqrt(eps) abstol=0", N=N, W=W, L=L, Ld=LD,
               maxiter=maxiter)
    runs = [("seq", "seq", (1, 1, 1, 1), 0), ("pair", "pair", (1, 1, 1, 1), 0), ("tree", "tree", (1, LD, W, L), 0)]
    runs += [(key, mode, (1, 1, 1, 1), thr) for key, mode, thr in blas_modes()]
    for key, mode, shape, thr in runs:
        if thr:
            out["blas_library"] = {k: v for k, v in orc.bind_blas(thr).items() if k != "threads"}
            out.setdefault("blas_threads", {})[key] = thr
        x, h = orc.cg(A, b, maxiter=maxiter, mode=mode, shape=shape)
        print(f"  cg {N}^3 {key}: {h['iters']} iterations", flush=True)
        out[key] = dict(iters=h["iters"], mvps=h["mvps"], isconverged=h["isconverged"], res0=float(h["res0"]).hex(),
                         tol=float(h["tol"]).hex(), resnorm=hexlist(h["resnorm"]),
                         x_checksum=float(np.sum(x)).hex(), x_norm=float(np.linalg.norm(x)).hex())
    return out


def gmres_case(N, restart):
    A, b = orc.advdiff(N, 1000.0)
    out = dict(case=f"gmres(advection_dominated(N={N}, beta=1000), restart={restart})", N=N, restart=restart, W=W, L=L,
               b_from="oracle/orc_advdiff_csc (glibc exp/sin)")
    runs = [("seq", "seq", (1, 1), 0), ("pair", "pair", (1, 1), 0), ("tree", "tree", (W, L), 0)]
    runs += [(key, mode, (1, 1), thr) for key, mode, thr in blas_modes()]
    for key, mode, shape, thr in runs:
        if thr:
            out["blas_library"] = {k: v for k, v in orc.bind_blas(thr).items() if k != "threads"}
            out.setdefault("blas_threads", {})[key] = thr
        x, h = orc.gmres(A, b, restart=restart, mode=mode, shape=shape)
        print(f"  gmres N={N} {key}: {h['iters']} iterations", flush=True)
        out[key] = dict(iters=h["iters"], mvps=h["mvps"], isconverged=h["isconverged"], res0=float(h["res0"]).hex(),
                         tol=float(h["tol"]).hex(), resnorm=hexlist(h["resnorm"]),
                         x_checksum=float(np.sum(x)).hex(), x_norm=float(np.linalg.norm(x)).hex())
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def blas_only(tag, sizes):
    """--blas-only TAG [N ...]: the `blas` / `blas8` histories of THIS host's OpenBLAS kernels (or of the core forced with OPENBLAS_CORETYPE, which
    must be set before the library loads: one process per core type) -> cg_lap<N>_blas_<TAG>.json next to the full goldens.  bench.py compares the
    device history with every such file (`parity_full_history.blas_hosts`) and reports the spread between the hosts: the band a Julia run
    (its own OpenBLAS build, an unknown core) lands in."""
    for N in sizes:
        A = orc.laplace(N, 3)
        b = orc.hashed_rhs(A.n)
        out = dict(case=f"cg(laplace_matrix(Float64,{N},3), hashed_rhs) reltol=sqrt(eps) abstol=0, dot / norm by the host OpenBLAS", N=N, host_tag=tag,
                   cpu_model=cpu_model(), forced_coretype=os.environ.get("OPENBLAS_CORETYPE"))
        for key, mode, thr in blas_modes():
            out["blas_library"] = {k: v for k, v in orc.bind_blas(thr).items() if k != "threads"}
            out.setdefault("blas_threads", {})[key] = thr
            x, h = orc.cg(A, b, mode=mode, shape=(1, 1, 1, 1))
            print(f"  cg {N}^3 {key} [{tag}: {out['blas_library']['core']}]: {h['iters']} iterations", flush=True)
            out[key] = dict(iters=h["iters"], mvps=h["mvps"], isconverged=h["isconverged"], res0=float(h["res0"]).hex(), tol=float(h["tol"]).hex(),
                            resnorm=hexlist(h["resnorm"]), x_checksum=float(np.sum(x)).hex(), x_norm=float(np.linalg.norm(x)).hex())
        dump(f"cg_lap{N}_blas_{tag}.json", out)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--blas-only":
        blas_only(sys.argv[2], [int(v) for v in sys.argv[3:]] or [64, 256])
        sys.exit(0)
    dump("cg_lap32.json", cg_case(32))
    dump("cg_lap64.json", cg_case(64))
    dump("gmres_advdiff50_r30.json", gmres_case(50, 30))
    dump("cg_lap256.json", cg_case(256))       # the full 613-iteration history of configs[1] (about ten minutes)
