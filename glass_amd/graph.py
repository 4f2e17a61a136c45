"""Device-resident CSR form of the normalised adjacency and the K1 launch plans.

Replaces the reference's cached uncoalesced COO (`buildAdj`, /root/reference/impl/models.py:83-111,
cached at 154-156).  Built once per (graph, aggr) — index plumbing (sort, bincount) uses torch ops
on the device; the values come from the HIP kernel `glass_adj_values_f32`; the launch plans are
built by the library's host function `glass_spmm_plan_build` from the row pointer.

HBM layout (all on the device the graph lives on):
  rowptr int32[N+1], col int32[nnz], val fp32[nnz]            CSR of A   (row = destination)
  rowptr_t, col_t, val_t                                      CSR of A^T (for the backward)
  plan / plan_t int32[...]                                    opaque K1 schedules
Duplicate (row,col) entries are kept as separate entries: they add up in the product exactly as in
the reference's uncoalesced COO matmul.
"""
import ctypes

import numpy as np
import torch

from . import _lib

class CSROperand:
    """One CSR matrix + its K1 plan; `spmm` enqueues Y = M @ X on the current stream."""
    def __init__(self, rowptr, col, val, n_rows, n_cols):
        self.rowptr, self.col, self.val = rowptr, col, val
        self.n_rows, self.n_cols = int(n_rows), int(n_cols)
        lib = _lib.load()
        rp_host = rowptr.cpu().numpy()  # one-time sync: the plan is host arithmetic on the row pointer
        words = ctypes.c_int64(0)
        _lib.check(lib.glass_spmm_plan_build(rp_host.ctypes.data, self.n_rows, None, ctypes.byref(words)), "plan size")
        plan = np.zeros(words.value, dtype=np.int32)
        _lib.check(lib.glass_spmm_plan_build(rp_host.ctypes.data, self.n_rows, plan.ctypes.data, ctypes.byref(words)),
                   "plan build")
        self.header = plan[:_lib.PLAN_HEADER_WORDS].copy()  # read on the host by every launch
        self.plan = torch.from_numpy(plan).to(rowptr.device)
        self._ws = {}

    @property
    def nnz(self):
        return int(self.header[3])

    def workspace(self, H):
        """Partial-row scratch of products with rows cut into several chunks: one buffer per feature width and per BRANCH
        (set_workspace_branch: concurrent products on the same matrix — the parallel evaluation branches of
        evalstep.EvalGraph — must not share it)."""
        key = (H, _ws_branch)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = _lib.load().glass_spmm_ws_bytes(self.header.ctypes.data, H)
            if nbytes < 0:
                raise _lib.GlassHipError("bad plan header")
            ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device=self.rowptr.device)
            self._ws[key] = ws
        return ws

    def spmm(self, x, out=None):
        assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32 and x.is_cuda
        assert x.shape[0] == self.n_cols, f"X has {x.shape[0]} rows, matrix has {self.n_cols} columns"
        H = x.shape[1]
        if out is None:
            out = torch.empty((self.n_rows, H), dtype=torch.float32, device=x.device)
        assert out.stride(1) == 1 and out.shape == (self.n_rows, H)
        rc = _lib.load().glass_spmm_csr_f32(self.rowptr.data_ptr(), self.col.data_ptr(), self.val.data_ptr(),
                                            x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), self.n_rows, H,
                                            self.header.ctypes.data, self.plan.data_ptr(),
                                            self.workspace(H).data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "glass_spmm_csr_f32")
        return out


def _csr_from_sorted(row, n_rows):
    counts = torch.bincount(row, minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=row.device)
    rowptr[1:] = torch.cumsum(counts, 0)
    return rowptr.to(torch.int32)


_ws_branch = 0


def set_workspace_branch(b):
    """Which workspace set CSROperand.workspace hands out from now on (0: the default one)."""
    global _ws_branch
    _ws_branch = int(b)


class CSRAdj:
    """Normalised adjacency A (forward) and A^T (backward) for one aggregation scheme."""
    def __init__(self, edge_index, edge_weight, n_node, aggr):
        if aggr not in _lib.AGGR_MODES:
            raise NotImplementedError  # same error type as the reference (models.py:110-111)
        if not edge_index.is_cuda:
            raise _lib.GlassHipError("glass_amd runs on the GPU only (edge_index is on %s); the CPU path lives in "
                                     "oracle/ as a checker, not as a fallback" % edge_index.device)
        n = int(n_node)
        nnz = edge_index.shape[1]
        if n >= 2**31 - 1 or nnz >= 2**31 - 1:
            raise _lib.GlassHipError("int32 CSR: need n_node, nnz < 2^31")
        dev = edge_index.device
        row, col = edge_index[0].to(torch.int64), edge_index[1].to(torch.int64)
        perm = torch.argsort(row * n + col, stable=True)
        row, col = row[perm], col[perm]
        w = edge_weight.to(torch.float32)[perm].contiguous()
        rowptr = _csr_from_sorted(row, n)
        col32 = col.to(torch.int32).contiguous()
        self.deg = torch.empty(n, dtype=torch.float32, device=dev)
        val = torch.empty(nnz, dtype=torch.float32, device=dev)
        rc = _lib.load().glass_adj_values_f32(rowptr.data_ptr(), col32.data_ptr(), w.data_ptr(), n,
                                              _lib.AGGR_MODES[aggr], self.deg.data_ptr(), val.data_ptr(),
                                              torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "glass_adj_values_f32")
        # transpose: same entries keyed by (col,row); values are permuted, never recomputed
        perm_t = torch.argsort(col * n + row, stable=True)
        rowptr_t = _csr_from_sorted(col[perm_t], n)
        col_t = row[perm_t].to(torch.int32).contiguous()
        val_t = val[perm_t].contiguous()
        self.n_node, self.aggr = n, aggr
        self.fwd = CSROperand(rowptr, col32, val, n, n)
        self.bwd = CSROperand(rowptr_t, col_t, val_t, n, n)

    @property
    def nnz(self):
        return self.fwd.nnz

    def to_dense(self):
        """Dense [N,N] (tests on tiny graphs only)."""
        n = self.n_node
        rp = self.fwd.rowptr.to(torch.int64)
        rows = torch.repeat_interleave(torch.arange(n, device=rp.device), rp[1:] - rp[:-1])
        a = torch.zeros(n * n, dtype=torch.float32, device=rp.device)
        a.index_add_(0, rows * n + self.fwd.col.to(torch.int64), self.fwd.val)
        return a.reshape(n, n)


class Selection:
    """Transpose of the [N,V] one-hot selection matrix of an embedding lookup (x is static per
    dataset): CSR with V rows whose columns are the node ids using that table row, values 1.
    `dW = S^T @ dout` is the embedding backward (ATen embedding_dense_backward) run on K1."""
    def __init__(self, x_flat, n_rows_table):
        n = x_flat.shape[0]
        lo, hi = int(x_flat.min()), int(x_flat.max())  # one-time validation (nn.Embedding raises IndexError)
        if lo < 0 or hi >= n_rows_table:
            raise IndexError(f"index out of range in embedding: [{lo},{hi}] vs table of {n_rows_table} rows")
        perm = torch.argsort(x_flat, stable=True)
        rowptr = _csr_from_sorted(x_flat[perm], n_rows_table)
        self.op = CSROperand(rowptr, perm.to(torch.int32).contiguous(),
                             torch.ones(n, dtype=torch.float32, device=x_flat.device), n_rows_table, n)
        # the same plan with its reduce launch switched off (header word 6 = number of reduce rows): the consumer sums the
        # partial rows itself (glass_embed_norm_bwd_adam_f32)
        self._hdr_noreduce = self.op.header.copy()
        self._n_reduce = int(self.op.header[6])
        self._off_reduce = int(self.op.header[12])
        self._hdr_noreduce[6] = 0

    def product_without_reduce(self, x):
        """G = S^T @ x on K1 WITHOUT the final reduce launch: rows cut into several chunks are left as partial rows.
        Returns (G, partials pointer, device pointer of the plan's reduce list, number of reduce rows)."""
        op = self.op
        H = x.shape[1]
        G = torch.empty((op.n_rows, H), dtype=torch.float32, device=x.device)
        ws = op.workspace(H)
        rc = _lib.load().glass_spmm_csr_f32(op.rowptr.data_ptr(), op.col.data_ptr(), op.val.data_ptr(), x.data_ptr(),
                                            x.stride(0), G.data_ptr(), G.stride(0), op.n_rows, H,
                                            self._hdr_noreduce.ctypes.data, op.plan.data_ptr(), ws.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "glass_spmm_csr_f32")
        return G, ws.data_ptr(), op.plan.data_ptr() + 4 * self._off_reduce, self._n_reduce
