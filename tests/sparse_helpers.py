"""Sparse test inputs: random CSR patterns, CSR->CSC, BCSC block patterns (samples/xgemm_sparse/
spmm_kernel.c:219-375 layouts), VNNI-2 packing of per-block A, and a Matrix-Market reader for the
fixture patterns under tests/golden/."""
import numpy as np

from helpers import rand_values
from libxsmm_amd.capi import DT


def random_csr(rng, rows, cols, density):
    nnz_target = max(1, int(round(rows * cols * density)))
    flat = np.sort(rng.choice(rows * cols, size=nnz_target, replace=False))
    r, c = flat // cols, flat % cols
    rowptr = np.zeros(rows + 1, dtype=np.uint32)
    np.add.at(rowptr, r + 1, 1)
    rowptr = np.cumsum(rowptr).astype(np.uint32)
    return rowptr, c.astype(np.uint32)


def csr_to_csc(rowptr, colidx, vals, rows, cols):
    order = np.lexsort((np.repeat(np.arange(rows), np.diff(rowptr)), colidx))
    r_of = np.repeat(np.arange(rows), np.diff(rowptr))
    colptr = np.zeros(cols + 1, dtype=np.uint32)
    np.add.at(colptr, colidx + 1, 1)
    colptr = np.cumsum(colptr).astype(np.uint32)
    return colptr, r_of[order].astype(np.uint32), vals[order].copy()


def make_bcsc(rng, K, N, bk, bn, keep, dtype):
    """Block-sparse K x N matrix in BCSC: vals[blk][dn][dk] (k fastest), colptr over N/bn, rowidx = k-block."""
    nkb, nnb = K // bk, N // bn
    colptr = [0]
    rowidx = []
    for nb in range(nnb):
        kept = np.sort(rng.choice(nkb, size=max(1, int(round(nkb * keep))), replace=False)) if keep < 1.0 else np.arange(nkb)
        rowidx += list(kept)
        colptr.append(len(rowidx))
    vals = rand_values(rng, len(rowidx) * bn * bk, dtype)
    return np.array(colptr, dtype=np.uint32), np.array(rowidx, dtype=np.uint32), vals


def structured_2_of_8(K, N, bk, bn):
    """BASELINE config #4 pattern: 2 of every 8 K-blocks per N-block are non-zero."""
    nkb, nnb = K // bk, N // bn
    colptr, rowidx = [0], []
    for nb in range(nnb):
        for g in range(0, nkb, 8):
            grp = [g + (nb % 8), g + ((nb + 3) % 8)]
            rowidx += sorted(x for x in set(grp) if x < nkb)
        colptr.append(len(rowidx))
    return np.array(colptr, dtype=np.uint32), np.array(rowidx, dtype=np.uint32)


def pack_vnni2(A, mb, K, M):
    """[mb][K][M] (M fastest) -> [mb][K/2][M][2]."""
    a = A.reshape(mb, K // 2, 2, M)
    return np.ascontiguousarray(a.transpose(0, 1, 3, 2)).reshape(-1)


def pack_vnni4(A, mb, K, M):
    """[mb][K][M] (M fastest) -> [mb][K/4][M][4] (8-bit operands, [ref: samples/xgemm_sparse/spmm_kernel.c:254-262])."""
    a = A.reshape(mb, K // 4, 4, M)
    return np.ascontiguousarray(a.transpose(0, 1, 3, 2)).reshape(-1)


def read_mtx(path):
    """Matrix-Market coordinate file -> dense float64 array (fixtures: sparsity patterns + values)."""
    with open(path) as f:
        lines = [ln for ln in f if not ln.startswith("%")]
    rows, cols, nnz = (int(x) for x in lines[0].split()[:3])
    dense = np.zeros((rows, cols))
    for ln in lines[1:1 + nnz]:
        parts = ln.split()
        dense[int(parts[0]) - 1, int(parts[1]) - 1] = float(parts[2]) if len(parts) > 2 else 1.0
    return dense
