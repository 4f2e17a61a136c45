"""The evaluation loop's forward passes (reference impl/train.py:20-34: `test` runs the model on every validation / test
batch, every epoch) replayed from ONE hipGraph whose K parallel branches are K independent batches.

At the small BASELINE shapes a forward pass is a chain of ~20 short launches that leaves most of the chip idle (ppi_bp-shape:
214 workgroups of one wave per SIMD on 256 CUs, every launch bound by its dependent round trips).  Evaluation batches do not
depend on each other — no parameter changes between them — so K of them can run side by side: the graph forks K streams
after one shared prologue (the weight images are packed once), each branch labels its own batch (utils.MaxZOZ) and runs
the whole forward, and the branches join at the end.  Each branch owns its buffers (activations from the graph's pool, its
own K1 partial-row workspace); nothing is written by two branches except the packed weight images, which the shared
prologue writes before the fork.  K = 1 is the sequential form (one batch per replay)."""
import torch

from . import graph as ggraph
from . import stack, utils


# hipGraphExecs WITH PARALLEL BRANCHES are never destroyed.  On this ROCm (7.2, libamdhip64 bundled with torch 2.10) destroying
# one — torch.cuda.CUDAGraph's destructor: hipGraphExecDestroy — leaves dangling stream pointers behind in the runtime: a
# later replay of ANOTHER multi-branch exec then crashes on the host in
#     hip::Graph::UpdateStreams(hip::Stream*, std::vector<hip::Stream*> const&)  <-  hip::GraphExec::Run  <-  hipGraphLaunch
# (tools/gc_crash_probe.py reproduces it in three iterations of create / replay / drop; with the graphs kept alive it runs
# clean).  Single-stream graphs (K = 1, the training step) are not affected and are released normally.  A dropped EvalGraph
# therefore parks its exec here.  So that parked execs do not pin memory, ALL evaluation graphs of a device capture into ONE
# shared memory pool (torch.cuda.graph(pool=...)): the allocator hands the blocks a dropped graph's tensors occupied to the
# next capture — a parked graph is never replayed again, so nobody minds its buffers being overwritten.  Live graphs sharing
# the pool alias each other's INTERMEDIATES too (the documented property of shared pools): harmless here, because a replay
# recomputes everything from its own inputs (`pos`, allocated outside the pool), replays of different graphs never overlap
# (one stream of replays), and callers clone the outputs they keep (`__call__`).  `retired_bytes()` is what parked graphs
# would still cost without that sharing: 0.
_RETIRED = []
_POOLS = {}


def _pool(device):
    h = _POOLS.get(device)
    if h is None:
        h = _POOLS[device] = torch.cuda.graph_pool_handle()
    return h


def retired_bytes():
    return 0


# ... and every evaluation graph of a device forks onto the SAME side streams: the caching allocator keeps its free blocks
# per stream, so graphs (and their eager warm-ups) on ever new streams of torch's 32-stream pool could not reuse each other's
# memory — reserved memory grew by a model's activations per dropped graph until the stream pool wrapped around.
_STREAMS = {}


def _streams(device, k):
    have = _STREAMS.setdefault(device, [])
    while len(have) < k:
        have.append(torch.cuda.Stream(device=device))
    return have[:k]


class EvalGraph:
    def __init__(self, model, x, edge_index, edge_weight, batch_shape, k=8, id=0, est_bytes=0):
        self.model, self.x, self.ei, self.ew, self.k, self.id = model, x, edge_index, edge_weight, int(k), id
        self.est_bytes = int(est_bytes)  # the caller's estimate of the private pool (activations of k branches)
        dev = x.device
        self.pos = [torch.full(tuple(batch_shape), -1, dtype=torch.int64, device=dev) for _ in range(self.k)]
        self.out = [None] * self.k
        self.streams = _streams(dev, self.k)
        self.graph = None

    def _branch(self, i):
        ggraph.set_workspace_branch(i + 1)  # this branch's own K1 partial-row workspace
        try:
            z = utils.MaxZOZ(self.x, self.pos[i])
            return self.model(self.x, self.ei, self.ew, self.pos[i], z, id=self.id)
        finally:
            ggraph.set_workspace_branch(0)

    def _prologue(self):
        arena = getattr(self.model.conv, "_glass_arena", None)
        if arena is not None:
            arena.refresh_transposes()
            return True
        return False

    @torch.no_grad()
    def capture(self):
        assert not self.model.training, "EvalGraph: model.eval() first (dropout off, no parameter updates between batches)"
        cur = torch.cuda.current_stream()
        # eager warm-up, every branch on its own stream: adjacency / plans / per-branch workspaces exist before the capture
        for i, s in enumerate(self.streams):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                self._branch(i)
            cur.wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=_pool(self.x.device)):
            cap = torch.cuda.current_stream()
            shared = self._prologue()
            with stack.prologue_done(shared):
                for i, s in enumerate(self.streams):
                    s.wait_stream(cap)
                    with torch.cuda.stream(s):
                        self.out[i] = self._branch(i)
                for s in self.streams:
                    cap.wait_stream(s)
        self.graph = g
        return self

    def __del__(self):
        g = self.__dict__.get("graph")
        if g is not None and self.k > 1:
            _RETIRED.append(g)  # the exec only: its output tensors go back to the shared pool

    def __call__(self, batches):
        """batches: up to K padded node matrices of the captured shape -> their logits (views of the graph's output buffers:
        clone what must survive the next call)."""
        if self.graph is None:
            self.capture()
        n = len(batches)
        assert 0 < n <= self.k
        for i in range(n):
            self.pos[i].copy_(batches[i])
        self.graph.replay()
        return self.out[:n]
