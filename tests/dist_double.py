"""CPU test double of dist.HipEngine: the same phase interface (mik_cgd_phase semantics, include/mik.h)
implemented with numpy + the oracle's tree reductions.  TEST INFRASTRUCTURE ONLY -- it lets the
partitioning, halo plans and the torch.distributed orchestration of dist.py run on gloo without a GPU."""
import contextlib

import numpy as np
import scipy.sparse as sp
import torch


class NumpyEngine:
    def __init__(self, orc, ptr, local_idx, val, plan, b_loc, x_loc=None, *, abstol, reltol, maxiter, shape=(1, 1, 2, 2)):
        self.orc, self.plan, self.shape = orc, plan, shape
        n_loc, n_ext = plan.n_loc, plan.n_loc + plan.n_ghost
        self.A = sp.csr_matrix((val, local_idx, ptr), shape=(n_loc, n_ext))
        self._u = np.zeros(max(n_ext, 1))
        self._send = np.zeros(max(plan.n_send, 1))
        self._dot = np.zeros(plan.nranks)
        self._rr = np.zeros(plan.nranks)
        self.u_ext, self.send_buf = torch.from_numpy(self._u), torch.from_numpy(self._send)      # shared memory
        self.dot_all, self.rr_all = torch.from_numpy(self._dot), torch.from_numpy(self._rr)
        self.b = np.array(b_loc, np.float64)
        self.initially_zero = x_loc is None
        self.x = np.zeros(n_loc) if x_loc is None else np.array(x_loc, np.float64)
        self.r, self.c = np.zeros(n_loc), np.zeros(n_loc)
        self.abstol, self.reltol, self.maxiter = abstol, reltol, maxiter
        self.res, self.prev, self.tol, self.beta, self.alpha = 1.0, 1.0, 0.0, 0.0, 0.0
        self.done, self.hist = False, []

    def ghost_view(self):
        return self.u_ext[self.plan.n_loc:self.plan.n_loc + self.plan.n_ghost]

    def dot_slot(self):
        return self.dot_all[self.plan.rank:self.plan.rank + 1]

    def rr_slot(self):
        return self.rr_all[self.plan.rank:self.plan.rank + 1]

    def stream_ctx(self):
        return contextlib.nullcontext()

    def _spmv(self):
        # scipy's CSR matvec adds each row's products in storage order with separate mul and add
        return self.A @ self._u[:self.plan.n_loc + self.plan.n_ghost]

    @staticmethod
    def _ranksum(a):
        s = a[0]
        for v in a[1:]:
            s = s + v
        return s

    def phase(self, ph, iteration=0):
        n, p = self.plan.n_loc, self.plan
        Wd, Ld, W, L = self.shape
        if ph == 10:
            if not self.initially_zero:
                self._u[:n] = self.x
                self._send[:p.n_send] = self._u[p.send_idx]
        elif ph == 11:
            if self.initially_zero:
                self.r[:] = self.b
            else:
                self.c[:] = self._spmv()
                self.r[:] = self.b - self.c
            self._rr[p.rank] = self.orc.dot(self.r, self.r, "tree", W, L)
            self._u[:] = 0
        elif ph == 12:
            self.res = float(np.sqrt(self._ranksum(self._rr)))
            self.prev = 1.0
            self.tol = max(self.reltol * self.res, self.abstol)
            self.beta = (self.res * self.res) / 1.0
            self.done = 0 >= self.maxiter or self.res <= self.tol
            self.hist = []
        elif self.done:
            return
        elif ph == 0:
            self._u[:n] = self.r + self.beta * self._u[:n]
            self._send[:p.n_send] = self._u[p.send_idx]
        elif ph == 1:
            self.c[:] = self._spmv()
            self._dot[p.rank] = self.orc.dot(self._u[:n].copy(), self.c, "tree", Wd, Ld)
        elif ph == 2:
            self.alpha = (self.res * self.res) / self._ranksum(self._dot)
            self.x += self.alpha * self._u[:n]
            self.r -= self.alpha * self.c
            self._rr[p.rank] = self.orc.dot(self.r, self.r, "tree", W, L)
        elif ph == 3:
            prev = self.res
            self.res = float(np.sqrt(self._ranksum(self._rr)))
            self.prev = prev
            self.beta = (self.res * self.res) / (prev * prev)
            self.hist.append(self.res)
            if iteration + 1 >= self.maxiter or self.res <= self.tol:
                self.done = True

    def wait(self, cap=1024):
        h, self.hist = np.array(self.hist, np.float64), []
        return self.res, self.tol, self.done, h

    def solution(self):
        return self.x.copy()


class NumpyGmresRank:
    """CPU test double of one rank of ``mik_gmres_create_partitioned`` (include/mik.h): the iterate of
    src/gmres.jl:57-106 on this rank's rows in numpy, coupling to the other ranks ONLY through
    ``dist.PartitionLinks`` (halo before every SpMV, rank-ordered sums of the projections / norms) -- so the
    partition plans, the communicator and the callback semantics run on gloo without a GPU.  fp64, MGS / CGS."""

    def __init__(self, orc, links, ptr, local_idx, val, plan, b_loc, *, restart, reltol, maxiter, method="mgs", shape=(2, 2)):
        self.orc, self.links, self.plan, self.shape, self.method = orc, links, plan, shape, method
        self.ptr, self.idx, self.val = np.asarray(ptr), np.asarray(local_idx), np.asarray(val, np.float64)
        n = plan.n_loc
        self.n, self.m, self.maxiter = n, restart, maxiter
        self.b = np.array(b_loc, np.float64)
        self.x = np.zeros(n)
        self.V = np.zeros((n, restart + 1), order="F")
        self.H = np.zeros((restart + 1, restart), order="F")
        self.nullvec = np.ones(restart + 1)
        self.mv_products = 1                                             # src/gmres.jl:122 (initially_zero)
        self.V[:, 0] = self.b                                            # init!: x = 0            :241
        beta = np.sqrt(self._sum(self.V[:, 0], self.V[:, 0]))            # :252
        self.V[:, 0] *= 1.0 / beta                                       # :253
        self.current, self.accumulator, self.res_beta, self.g_beta = beta, 1.0, beta, beta
        self.tol = max(reltol * beta, 0.0)
        self.k = 1

    # -- the two coupling points --------------------------------------------------------------
    def _spmv(self, v):
        p, xe = self.plan, self.links.x_ext.numpy()
        xe[:self.n] = v
        if p.n_send:
            self.links.send_buf.numpy()[:p.n_send] = xe[p.send_idx]
        self.links.halo()
        y = np.zeros(self.n)
        lens = np.diff(self.ptr)
        for j in range(int(lens.max()) if self.n else 0):                # per-row sums in storage (column) order
            rows = np.nonzero(lens > j)[0]
            e = self.ptr[rows] + j
            y[rows] = y[rows] + self.val[e] * xe[self.idx[e]]
        return y

    def _sum(self, a, b, count=1):
        W, L = self.shape
        vals = np.array([self.orc.dot(np.ascontiguousarray(a), np.ascontiguousarray(b), "tree", W, L)])
        self.links.reduce(vals)
        return vals[0]

    # -- src/gmres.jl:57-106 ----------------------------------------------------------------------
    def _is_done(self, iteration):
        return iteration >= self.maxiter or self.current <= self.tol

    def iterate(self, iteration):
        if self._is_done(iteration):
            return None
        k, V, H = self.k, self.V, self.H
        w = self._spmv(V[:, k - 1])                                      # expand!                  :285-288
        self.mv_products += 1
        h = np.zeros(k)
        if self.method == "mgs":                                         # src/orthogonalize.jl:69-76
            for i in range(k):
                h[i] = self._sum(V[:, i], w)
                w = w - h[i] * V[:, i]
        else:                                                            # ClassicalGramSchmidt     :43-45
            W, L = self.shape
            part = np.array([self.orc.dot(np.ascontiguousarray(V[:, i]), w, "tree", W, L) for i in range(k)])
            self.links.reduce(part)
            h[:] = part
            w = self.orc.gemv_n(V[:, :k], h, w, alpha=-1.0)
        nrm = np.sqrt(self._sum(w, w))
        V[:, k] = w * (1.0 / nrm)
        H[:k, k - 1], H[k, k - 1] = h, nrm
        if H[k, k - 1] == 0.0:                                           # update_residual!         :224-233
            self.current = 0.0
        else:
            d = 0.0
            for i in range(k):
                d = d + self.nullvec[i] * H[i, k - 1]
            self.nullvec[k] = -(d / H[k, k - 1])
            self.accumulator = self.accumulator + self.nullvec[k] * self.nullvec[k]
            self.current = self.res_beta / np.sqrt(self.accumulator)
        k += 1
        if k == self.m + 1 or self._is_done(iteration + 1):              # :82
            rhs = np.zeros(k)
            rhs[0] = self.g_beta
            _, y = self.orc.hessenberg_ldiv(H[:k, :k - 1], rhs)         # solve_least_squares!     :262-271
            self.x = self.orc.gemv_n(V[:, :k - 1], y[:k - 1], self.x, alpha=1.0)   # update_solution! :273-276
            k = 1
            if not self._is_done(iteration):                             # :93-101
                r = self.b - self._spmv(self.x)
                beta = np.sqrt(self._sum(r, r))
                V[:, 0] = r * (1.0 / beta)
                self.g_beta, self.accumulator, self.res_beta = beta, 1.0, beta
                self.mv_products += 1
        self.k = k
        return self.current, iteration + 1
