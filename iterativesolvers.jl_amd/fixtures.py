"""Synthetic inputs of the hot path (numpy, host side): the reference's own fixtures, vectorised.

These build the *inputs* handed to the C ABI (SparseMatrixCSC arrays + right-hand sides) for
bench.py and the tests; they contain no solver arithmetic.  The oracle has its own independent
generators in oracle/mik_oracle.c; tests cross-check the two.
"""
from __future__ import annotations

import math

import numpy as np


def laplace_matrix(N: int, dims: int = 3, dtype=np.float64, index_base: int = 1, rows=None):
    """``laplace_matrix(T, N, dims)`` -- test/laplace_matrix.jl:1-12, as SparseMatrixCSC fields
    ``(n, colptr, rowval, nzval)`` with ``index_base``-based Int64 indices.

    The matrix is symmetric, so columns ``rows = (r0, r1)`` of the CSC are also rows r0..r1-1 of
    the CSR; passing ``rows`` returns only that slab's ``(ptr, idx, val)`` (idx global), which is
    how the row-partitioned multi-GPU case builds its local block without forming 512^3 on one host.
    """
    n = N ** dims
    r0, r1 = (0, n) if rows is None else rows
    strides = [N ** d for d in range(dims)]
    CH = 1 << 21                                          # rows per chunk: bounds the temporaries to ~0.4 GB whatever the slab size
    ptr = np.empty(r1 - r0 + 1, np.int64)
    ptr[0] = 0
    idx_parts, val_parts = [], []
    for c0 in range(r0, r1, CH):
        j = np.arange(c0, min(c0 + CH, r1), dtype=np.int64)
        cand = []
        for d in reversed(range(dims)):                   # rows above the diagonal: j - N^2, j - N, j - 1
            c = (j // strides[d]) % N
            cand.append((j - strides[d], c > 0, -1.0))
        cand.append((j, np.ones(j.shape, bool), 2.0 * dims))
        for d in range(dims):                             # j + 1, j + N, j + N^2
            c = (j // strides[d]) % N
            cand.append((j + strides[d], c < N - 1, -1.0))
        idx = np.stack([c[0] for c in cand], axis=1)
        mask = np.stack([c[1] for c in cand], axis=1)
        val = np.broadcast_to(np.asarray([c[2] for c in cand], dtype=dtype), idx.shape)
        np.cumsum(mask.sum(axis=1), out=ptr[c0 - r0 + 1:c0 - r0 + 1 + j.size])
        ptr[c0 - r0 + 1:c0 - r0 + 1 + j.size] += ptr[c0 - r0]
        idx_parts.append(idx[mask] + index_base)
        val_parts.append(np.ascontiguousarray(val[mask]))
    return n, ptr + index_base, np.concatenate(idx_parts) if idx_parts else np.empty(0, np.int64), \
        np.concatenate(val_parts) if val_parts else np.empty(0, dtype)


def advection_dominated(N: int = 50, beta: float = 1000.0, index_base: int = 1):
    """``advection_dominated(; N, β)`` -- benchmark/advection_diffusion.jl:3-30 -> CSC fields and b.

    A = laplace_matrix(Float64, N, 3) ./ -h^2 + kron(I, spdiagm(-1 => -β/2h, 1 => β/2h)), h = 1/(N+1).
    """
    n, colptr, rowval, _ = laplace_matrix(N, 3, np.float64, 0)
    h = 1.0 / (N + 1)
    mh2 = -(h * h)
    lap_diag, lap_off = 6.0 / mh2, -1.0 / mh2
    dx_sub, dx_sup = (-beta) / (2.0 * h), beta / (2.0 * h)
    cols = np.repeat(np.arange(n, dtype=np.int64), np.diff(colptr))
    diff = rowval - cols                                   # row - col
    nzval = np.full(rowval.shape, lap_off)
    nzval[diff == 0] = lap_diag
    nzval[diff == -1] = lap_off + dx_sup                    # A[i, i+1]
    nzval[diff == 1] = lap_off + dx_sub                     # A[i, i-1]
    # rhs (benchmark/advection_diffusion.jl:1,27): f(x, y, z) = exp(x y z) sin(pi x) sin(pi y) sin(pi z) on the interior grid, x fastest.
    # exp / sin are the C library's (math.exp / math.sin call libm, as Julia's and the oracle's do to within the same correctly-rounded
    # results on this platform); numpy's SIMD exp differs from libm by an ulp on some arguments, which would move every residual
    # of the committed golden history.  The products are formed left to right like the reference's expression.
    xs = np.arange(1, N + 1, dtype=np.float64) / np.float64(N + 1)
    sn = np.array([math.sin(math.pi * float(v)) for v in xs])
    X, Y, Z = xs[None, None, :], xs[None, :, None], xs[:, None, None]    # x fastest
    arg = np.ascontiguousarray(np.broadcast_to((X * Y) * Z, (N, N, N))).reshape(-1)
    ex = np.fromiter(map(math.exp, arg.tolist()), np.float64, count=arg.size).reshape(N, N, N)
    b = ((ex * sn[None, None, :]) * sn[None, :, None]) * sn[:, None, None]
    return n, colptr + index_base, rowval + index_base, nzval, np.ascontiguousarray(b.reshape(-1))


def read_matrix_market(path: str, dtype=np.float64, index_base: int = 1):
    """MatrixMarket ``coordinate`` reader -- the on-disk format of BASELINE.json configs[4]
    (benchmark/matrixmarket.jl:5-10 downloads ``s3dkq4m2.mtx`` and calls ``MatrixMarket.mmread``).
    Supports real / integer / pattern fields and general / symmetric / skew-symmetric symmetry; returns
    SparseMatrixCSC fields ``(n_rows, n_cols, colptr, rowval, nzval)`` (duplicates summed, rows ascending
    within a column) ready for ``HipCSR``."""
    with open(path, "rb") as f:
        header = f.readline().decode().strip().lower().split()
        if len(header) < 5 or header[0] != "%%matrixmarket" or header[1] != "matrix" or header[2] != "coordinate":
            raise ValueError(f"{path}: only '%%MatrixMarket matrix coordinate ...' files are supported")
        field, symmetry = header[3], header[4]
        if field not in ("real", "integer", "pattern", "double"):
            raise ValueError(f"{path}: field '{field}' is out of scope (complex element types are not on this path)")
        line = f.readline()
        while line.startswith(b"%") or not line.strip():
            line = f.readline()
        n_rows, n_cols, nnz = (int(t) for t in line.split())
        data = np.loadtxt(f, dtype=np.float64, ndmin=2) if nnz else np.zeros((0, 3))
    if data.shape[0] != nnz:
        raise ValueError(f"{path}: expected {nnz} entries, found {data.shape[0]}")
    i = data[:, 0].astype(np.int64) - 1
    j = data[:, 1].astype(np.int64) - 1
    v = np.ones(nnz) if field == "pattern" else data[:, 2]
    if symmetry in ("symmetric", "skew-symmetric"):
        off = i != j
        sign = -1.0 if symmetry == "skew-symmetric" else 1.0
        i, j, v = np.concatenate([i, j[off]]), np.concatenate([j, i[off]]), np.concatenate([v, sign * v[off]])
    elif symmetry != "general":
        raise ValueError(f"{path}: symmetry '{symmetry}' is not supported")
    order = np.lexsort((i, j))                                   # column-major, rows ascending
    i, j, v = i[order], j[order], v[order]
    if i.size:
        first = np.ones(i.size, bool)
        first[1:] = (i[1:] != i[:-1]) | (j[1:] != j[:-1])
        group = np.cumsum(first) - 1
        v = np.bincount(group, weights=v)                        # duplicates are summed
        i, j = i[first], j[first]
    colptr = np.zeros(n_cols + 1, np.int64)
    np.cumsum(np.bincount(j, minlength=n_cols), out=colptr[1:])
    return n_rows, n_cols, colptr + index_base, i + index_base, np.ascontiguousarray(v.astype(dtype))


def _hash32(x: np.ndarray, mult: int) -> np.ndarray:
    """(x * mult) mod 2^32 followed by an xorshift -- integer arithmetic only, exact in any language."""
    h = (x.astype(np.uint64) * np.uint64(mult)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    h ^= h >> np.uint64(13)
    return h


def irregular_matrix(n: int = 1_000_000, dtype=np.float32, long_rows: bool = True, bandwidth: int = 0):
    """Synthetic irregular CSR operator standing in for BASELINE.json configs[4] (the SuiteSparse files
    of benchmark/matrixmarket.jl:5 / matrixcollection.jl:6 cannot be downloaded here) -- SURVEY.md
    section 8d: row-length classes {90 %: 5-15, 9.9 %: 50-200, 0.1 %: 5,000-20,000 (capped at n/2)}
    chosen by an integer hash of the row index, column indices by a second hash, off-diagonal values in
    [-1, 1) by a third, diagonal = 1 + sum |off-diagonal| (strictly diagonally dominant => GMRES
    converges).  Returns 0-based CSR fields (n, rowptr, colidx, val) with columns ascending in a row;
    duplicate off-diagonal columns are merged by keeping the first.

    ``bandwidth`` = 0: columns uniformly random over the whole matrix (no locality at all: every gather of x misses
    L1 -- the worst case).  ``bandwidth`` = w > 0: "banded-irregular" -- the same row lengths and values, but a row's
    columns are drawn from [row - w_r, row + w_r] with w_r = max(w, 4 x row length): the locality a bandwidth-reducing
    ordering (RCM) gives finite-element / shell matrices such as the s3dkq4m2 of benchmark/matrixmarket.jl:5."""
    rows = np.arange(n, dtype=np.int64)
    cls = _hash32(rows, 2654435761) % np.uint64(1000)
    sel = _hash32(rows, 40503)
    length = np.where(cls < 900, 5 + sel % np.uint64(11),
                      np.where((cls < 999) | (not long_rows), 50 + sel % np.uint64(151),
                               5000 + sel % np.uint64(15001))).astype(np.int64)
    length = np.minimum(length, max(1, n // 2))
    start = np.zeros(n + 1, np.int64)
    np.cumsum(length, out=start[1:])
    total = int(start[-1])
    r = np.repeat(rows, length)
    k = np.arange(total, dtype=np.int64) - start[r]                   # position inside the row
    hcol = _hash32(r * 131071 + k, 2246822519)
    if bandwidth > 0:
        w = np.maximum(np.int64(bandwidth), 4 * length)[r]             # half-width of this row's band
        c = (r + (hcol % (2 * w + 1).astype(np.uint64)).astype(np.int64) - w) % n
    else:
        c = (hcol % np.uint64(n)).astype(np.int64)
    c = np.where(c == r, (c + 1) % n, c)                               # keep the diagonal separate
    v = (_hash32(r * 8191 + k + 7, 3266489917).astype(np.float64) / 2147483648.0 - 1.0)
    # sort by (row, col), first occurrence first.  One 64-bit key instead of a three-key lexsort (the same order: the keys are unique;
    # rows and columns < 2^24, positions < 2^15 -- asserted): a third of the time for 34 M entries
    assert n < (1 << 24) and int(length.max()) < (1 << 15)
    order = np.argsort((r.astype(np.uint64) << np.uint64(39)) | (c.astype(np.uint64) << np.uint64(15)) | k.astype(np.uint64), kind="stable")
    r, c, v = r[order], c[order], v[order]
    keep = np.ones(total, bool)
    keep[1:] = (r[1:] != r[:-1]) | (c[1:] != c[:-1])
    r, c, v = r[keep], c[keep], v[keep]
    diag = 1.0 + np.bincount(r, weights=np.abs(v), minlength=n)
    r_all = np.concatenate([r, rows])
    c_all = np.concatenate([c, rows])
    v_all = np.concatenate([v, diag])
    order = np.argsort((r_all.astype(np.uint64) << np.uint64(24)) | c_all.astype(np.uint64), kind="stable")      # (row, col): unique keys
    r_all, c_all, v_all = r_all[order], c_all[order], v_all[order]
    rowptr = np.zeros(n + 1, np.int64)
    np.cumsum(np.bincount(r_all, minlength=n), out=rowptr[1:])
    return n, rowptr, c_all, np.ascontiguousarray(v_all.astype(dtype))


def fe_matrix(grid, dof: int = 3, dtype=np.float32, renumber: bool = True):
    """Finite-element-shaped SPD operator: what the SuiteSparse inputs of benchmark/matrixmarket.jl:5 (``s3dkq4m2``: a
    cylindrical shell, 4-node quadrilaterals with 6 unknowns per node, n = 90,449, ~53 entries per row) and
    benchmark/matrixcollection.jl:6-8 look like, which cannot be downloaded here.  ``grid`` = nodes per dimension of a
    structured mesh (2 entries: quadrilaterals, 9-node neighbourhoods; 3 entries: hexahedra, 27-node neighbourhoods), ``dof``
    unknowns per node, unknown = node * dof + k (node-major, the usual FE numbering); every unknown of a node couples with every
    unknown of every node of the neighbourhood (dense dof x dof blocks) => interior rows carry 9 * dof resp. 27 * dof entries
    in clusters of ``dof`` consecutive columns.  ``fe_matrix((123, 123), 6)``: n = 90,774, 4.8 M entries, rows of 24 / 36 / 54;
    ``fe_matrix((64, 64, 64), 3)``: n = 786,432, 62 M entries, rows of 24 ... 81.

    Off-diagonal values in [-1, 1) from an integer hash of the unordered pair (symmetric), diagonal = 1 + sum |off-diagonal|
    (symmetric + strictly diagonally dominant + positive diagonal => SPD, so cg! applies as it does to s3dkq4m2).
    ``renumber`` (default): the nodes are numbered tile by tile (tiles of 4 nodes per dimension) and in hashed order inside a
    tile -- the numbering a mesh generator followed by a bandwidth-reducing ordering leaves: neighbours stay close, but
    (column - row) takes thousands of distinct values, as in a real unstructured mesh (the lexicographic numbering would
    qualify for the library's stencil-only 8-bit column codes).
    Returns 0-based CSR fields ``(n, rowptr, colidx, val)``, columns ascending in a row; being symmetric they are the CSC
    fields as well."""
    grid = tuple(int(g) for g in grid)
    d = len(grid)
    assert d in (2, 3) and dof >= 1
    nn = int(np.prod(grid))
    node = np.arange(nn, dtype=np.int64)
    coord, stride, s = [], [], 1
    for g in grid:                                          # first dimension fastest
        coord.append((node // s) % g)
        stride.append(s)
        s *= g
    offs = np.stack(np.meshgrid(*[np.arange(-1, 2)] * d, indexing="ij"), axis=-1).reshape(-1, d)[:, ::-1]
    lin = (offs * np.asarray(stride)).sum(axis=1)
    offs = offs[np.argsort(lin, kind="stable")]             # neighbours in ascending node order
    nb = np.empty((nn, offs.shape[0]), np.int64)
    ok = np.ones((nn, offs.shape[0]), bool)
    for j, o in enumerate(offs):
        nb[:, j] = node + int((o * np.asarray(stride)).sum())
        for a in range(d):
            ok[:, j] &= (coord[a] + o[a] >= 0) & (coord[a] + o[a] < grid[a])
    if renumber:
        tile = np.zeros(nn, np.int64)
        tmul = 1
        for a in range(d):
            tile += (coord[a] // 4) * tmul
            tmul *= (grid[a] + 3) // 4
        order = np.lexsort((_hash32(node, 2654435761), tile))          # old node ids in new order
        newid = np.empty(nn, np.int64)
        newid[order] = node
        nb = np.where(ok, newid[np.clip(nb, 0, nn - 1)], np.int64(nn))[order]     # rows in new order, neighbours as new ids
        nb.sort(axis=1)                                                # ascending; the absent ones (= nn) last
        ok = nb < nn
    cnt = ok.sum(axis=1)                                    # neighbour nodes per node
    pair_b = nb[ok]                                         # node-major, ascending neighbour
    pair_start = np.zeros(nn + 1, np.int64)
    np.cumsum(cnt, out=pair_start[1:])
    n = nn * dof
    row_len = np.repeat(cnt * dof, dof)
    rowptr = np.zeros(n + 1, np.int64)
    np.cumsum(row_len, out=rowptr[1:])
    E = int(rowptr[-1])
    colidx = np.empty(E, np.int64)
    val = np.empty(E, np.float64)
    CH = max(dof, (1 << 18) // dof * dof)                   # rows per chunk (bounds the temporaries)
    diag_pos = np.empty(n, np.int64)
    for r0 in range(0, n, CH):
        r1 = min(r0 + CH, n)
        rl = row_len[r0:r1]
        r = np.repeat(np.arange(r0, r1, dtype=np.int64), rl)
        q = np.arange(rowptr[r0], rowptr[r1], dtype=np.int64) - rowptr[r]
        c = pair_b[pair_start[r // dof] + q // dof] * dof + q % dof
        lo, hi = np.minimum(r, c), np.maximum(r, c)
        v = _hash32(lo * 1000003 + hi * 7 + 11, 3266489917).astype(np.float64) / 2147483648.0 - 1.0
        isd = r == c
        v[isd] = 0.0
        colidx[rowptr[r0]:rowptr[r1]] = c
        val[rowptr[r0]:rowptr[r1]] = v
        diag_pos[r0:r1] = np.flatnonzero(isd) + rowptr[r0]
    absum = np.add.reduceat(np.abs(val), rowptr[:-1])
    val[diag_pos] = 1.0 + absum
    return n, rowptr, colidx, np.ascontiguousarray(val.astype(dtype))


def box_stencil_matrix(N: int, dims: int = 3, dtype=np.float64):
    """Constant-coefficient (3^dims)-point box stencil on an N^dims grid, lexicographic numbering: 27-point in 3-D, 9-point in
    2-D -- every neighbour of the 3^dims box -1, the diagonal 3^dims - 1 everywhere (rows at the boundary are strictly
    dominant: SPD).  The operators the reference's `laplace_matrix` fixture (test/laplace_matrix.jl:1-12) generalises to when a
    discretisation couples the diagonal neighbours too.  0-based CSR fields ``(n, rowptr, colidx, val)``, symmetric."""
    n, rowptr, colidx, val = fe_matrix((N,) * dims, 1, np.float64, renumber=False)
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr))
    val = np.where(colidx == rows, float(3 ** dims - 1), -1.0).astype(dtype)
    return n, rowptr, colidx, np.ascontiguousarray(val)


def hashed_rhs(n: int, start: int = 0, stop=None, dtype=np.float64) -> np.ndarray:
    """b[i] = ((i * 2654435761) mod 2^32) / 2^32 - 0.5 for the 1-based i in (start, stop]
    (exact in any language)."""
    stop = n if stop is None else stop
    i = np.arange(start + 1, stop + 1, dtype=np.uint64)
    h = (i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    return (h.astype(np.float64) / 4294967296.0 - 0.5).astype(dtype)
