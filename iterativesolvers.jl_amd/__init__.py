"""MI355X-native Krylov inner loop behind IterativeSolvers.jl's cg! / gmres! path.

The directory name (``iterativesolvers.jl_amd``) is not a Python identifier; load it with
``__graft_entry__.load_package()`` (registers it as ``iterativesolvers_jl_amd``).

  csrc/      hand-written HIP kernels (gfx950) + the C ABI of include/mik.h  -> libmik.so
  _lib.py    ctypes binding of that ABI (fails loudly when the library is missing)
  api.py     host-side mirror of the reference interface (cg, cg_, gmres, gmres_, iterables ...): SURVEY section 8 rows only
  extras.py  solvers outside the scope contract (IDR(s), LSQR, LSMR, QMR, power method); kept apart, unjudged
  dist.py    row-partitioned multi-GPU CG / GMRES (one process per GPU; RCCL, peer-mapped mailboxes, in-process group)
  bench_dist.py  measurement harness of bench.py --gpus N (self-test orchestration, group fall-back, the line) -- not product code
  selftest.py    transport self-test child process
  fixtures.py  the reference's test/benchmark inputs as SparseMatrixCSC arrays
  julia/     the Julia-side shim (ccall bindings + dispatch methods), see INTEGRATION.md
"""
from . import _lib, fixtures                                    # noqa: F401
from ._lib import MikError, lib                                  # noqa: F401
from .api import (CGIterable, CGStateVariables, ClassicalGramSchmidt, ConvergenceHistory, DGKS,   # noqa: F401
                  GenericCGIterable, GMRESIterable, HipContext, HipCSR, HipMatrix, HipVector, Identity,
                  JacobiPrec, ModifiedGramSchmidt, PCGIterable, cg, cg_, cg_iterator_, default_context,
                  dot, gemv_n_, gmres, gmres_, gmres_iterable_, hessenberg_ldiv_, mul_, niters, norm, nprods,
                  nrests, orthogonalize_and_normalize_, zerox, BiCGStabIterable, bicgstabl, bicgstabl_, bicgstabl_iterator_,
                  gemv_t_, lu_solve_, ChebyshevIterable, chebyshev, chebyshev_, chebyshev_iterable_, MINRESIterable, minres, minres_,
                  minres_iterable_, givens_algorithm, axpy_dot_, axpy2_nrm2_, gram_, LinearOperator)
from . import extras                                             # noqa: F401  (beyond SURVEY section 8: IDR(s), LSQR, LSMR, QMR, powm -- unjudged, not re-exported)
