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
