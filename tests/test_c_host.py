"""The drop-in boundary used from plain C (tests/c_host/cg_host.c): include/mik.h must compile as C99, libmik.so must
link without Python / PyTorch in the process, fail cleanly without a device, and on a GPU reproduce the oracle's
cg! history bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def build(tmp_path, pkg):
    exe = str(tmp_path / "cg_host")
    libdir = os.path.dirname(pkg._lib.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c_host", "cg_host.c"), "-o", exe, "-L", libdir, "-l:libmik.so", f"-Wl,-rpath,{libdir}"]
    subprocess.check_call(cmd)
    return exe


def test_header_is_c99_and_host_fails_cleanly_without_a_device(pkg, tmp_path):
    import torch
    exe = build(tmp_path, pkg)
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    p = subprocess.run([exe, "6"], capture_output=True, text=True)
    assert p.returncode == 3 and "no HIP device" in p.stderr and p.stdout == ""


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["cg", "gmres", "cgop"])
def test_c_host_reproduces_the_oracle_history(pkg, orc, ctx, tmp_path, solver):
    exe = build(tmp_path, pkg)
    N = 12
    p = subprocess.run([exe, str(N)] + ([solver] if solver != "cg" else []), capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.split("\n")
    hist = np.array([float.fromhex(s) for s in lines if s.startswith("0x")])
    tail = next(s for s in lines if s.startswith("iters")).split()
    mach = next(s for s in lines if s.startswith("machine")).split()           # mik_ctx_info from plain C: the same answer as the ctypes binding
    info = ctx.info()
    assert mach[1] == info["arch"] and int(mach[3]) == info["compute_units"] and int(mach[5]) == info["xcds"] and int(mach[7]) == 64
    A = orc.laplace(N, 3)
    b = orc.hashed_rhs(A.n)
    if solver == "cg":
        xo, ho = orc.cg(A, b, mode="tree", shape=ctx.cg_shape(np.float64))
    elif solver == "cgop":                      # callback operator: dot(u, c) is reduced with the vector tree shape
        W, L = ctx.reduce_shape(np.float64)
        xo, ho = orc.cg(A, b, mode="tree", shape=(W, L, W, L))
        # every step of the batch enqueues one callback, also the no-op steps after the device-side stopping test fired
        assert int(next(l for l in lines if l.startswith("callback_calls")).split()[1]) >= ho["iters"]
    else:
        xo, ho = orc.gmres(A, b, restart=10, mode="tree", shape=ctx.reduce_shape(np.float64))
    assert int(tail[1]) == ho["iters"] and int(tail[3]) == int(ho["isconverged"]) and int(tail[7]) == ho["mvps"]
    assert np.array_equal(hist, ho["resnorm"])
    s = 0.0
    for v in xo:
        s += float(v)
    assert float.fromhex(next(l for l in lines if l.startswith("sum_x")).split()[1]) == s
