"""One-shot peer exchange of the gradient arena fused with Adam (glass_peer_allreduce_adam_f32, csrc/peer.hip): the opt-in
alternative to dist.GradExchange's RCCL all-reduce for the SMALL bucket of the data-parallel step (reference
impl/train.py:10-16 + Adam at GLASSTest.py:213).

Every rank's gradient arena (and a 16-byte flag block) lives in an allocation of its own that the other ranks map through
hipIpc handles (one GPU per process on an xGMI node; in the functional test two processes share ONE GPU).  The step's last
launch then reads the peers' arenas directly, sums in rank order, and applies Adam — no collective call, capturable with the
rest of the step.  What stays with RCCL: an embedding-sized (big) bucket, which wants reduce-scatter bandwidth, not latency.

    peer.attach(arena)              after init_process_group (any backend: it only carries the 64-byte handles) and ParamArena
    FlatAdam(arena, ...).step()     -> the fused launch;  arena.all_reduce_mean() becomes a no-op
    arena._peer.check()             raises if a launch timed out waiting for a peer (sticky device status), e.g. once per epoch

RCCL remains the default until a multi-GPU run has measured this path (dist.predict_oneshot_us states the model)."""
import ctypes

import torch
import torch.distributed as td

from . import _lib
from . import dist as gdist

SPIN_LIMIT = 1 << 22   # polls (each ~0.2-0.5 us with the sleep between them): ~1-2 s before a launch gives up


class _PeerGroupStruct(ctypes.Structure):
    _fields_ = [("world", ctypes.c_int32), ("rank", ctypes.c_int32), ("grad", ctypes.c_void_p * 8), ("flags", ctypes.c_void_p * 8)]


class _RawDeviceBuffer:
    """A raw device allocation exposed to torch (zero-copy) through __cuda_array_interface__."""
    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n, ), "typestr": typestr, "data": (ptr, False), "version": 2}


def _alloc(nbytes):
    out = ctypes.c_void_p()
    _lib.check(_lib.load().glass_peer_alloc(nbytes, ctypes.byref(out)), "glass_peer_alloc")
    return out.value


class PeerExchange:
    def __init__(self, arena, spin_limit=SPIN_LIMIT):
        if not gdist.is_distributed():
            raise RuntimeError("peer.attach needs an initialised process group with more than one rank")
        self.world, self.rank = gdist.world_size(), gdist.rank()
        if self.world > 8:
            raise ValueError("the one-shot exchange serves up to 8 ranks (one xGMI node)")
        if arena.big_start < arena.flat.numel():
            raise ValueError("the one-shot exchange serves the small bucket only: this arena has an embedding-sized bucket "
                             "(reduce-scatter + sharded Adam through RCCL: dist.GradExchange)")
        lib = _lib.load()
        dev = arena.flat.device
        n = arena.flat.numel()
        self.n, self.spin_limit, self.arena = n, int(spin_limit), arena
        self._own_grad, self._own_flags = _alloc(max(n * 4, 16)), _alloc(64)
        # the gradient arena moves into the shared allocation: every .grad view follows (captured graphs are dropped)
        shared = torch.as_tensor(_RawDeviceBuffer(self._own_grad, n, "<f4"), device=dev)
        shared.copy_(arena.flat)
        arena.adopt_grad_storage(shared)
        self._keep = shared
        handles = []
        for p in (self._own_grad, self._own_flags):
            h = (ctypes.c_ubyte * 64)()
            _lib.check(lib.glass_peer_export(p, h), "glass_peer_export")
            handles.append(bytes(h))
        gathered = [None] * self.world
        td.all_gather_object(gathered, handles)
        self._grp = _PeerGroupStruct()
        self._grp.world, self._grp.rank = self.world, self.rank
        self._mapped = []
        for r in range(self.world):
            if r == self.rank:
                self._grp.grad[r], self._grp.flags[r] = self._own_grad, self._own_flags
                continue
            ptrs = []
            for hb in gathered[r]:
                out = ctypes.c_void_p()
                _lib.check(lib.glass_peer_import((ctypes.c_ubyte * 64).from_buffer_copy(hb), ctypes.byref(out)), "glass_peer_import")
                ptrs.append(out.value)
                self._mapped.append(out.value)
            self._grp.grad[r], self._grp.flags[r] = ptrs
        self.seq = torch.zeros(2, dtype=torch.int64, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        td.barrier()  # every rank has mapped every arena before the first launch polls a flag

    def step(self, opt, mean_out=None):
        """The fused exchange + Adam launch for FlatAdam / AdoptedAdam `opt` (whole arena)."""
        b1, b2, eps, wd = opt.hyper()
        a = self.arena
        rc = _lib.load().glass_peer_allreduce_adam_f32(ctypes.byref(self._grp), self.n, a.flat_param.data_ptr(), opt.exp_avg.data_ptr(),
                                                       opt.exp_avg_sq.data_ptr(), opt.lr_dev.data_ptr(), b1, b2, eps, wd,
                                                       opt.step_dev.data_ptr(), self.seq.data_ptr(), self.status.data_ptr(),
                                                       self.spin_limit, 0 if mean_out is None else mean_out.data_ptr(),
                                                       torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "glass_peer_allreduce_adam_f32")

    def check(self):
        """Host synchronisation point: raise if any launch gave up waiting for a peer."""
        if int(self.status.item()) != 0:
            raise RuntimeError("one-shot peer exchange: a rank's gradients did not arrive within the spin limit — parameters were "
                               "left untouched from that step on")

    def close(self):
        lib = _lib.load()
        torch.cuda.synchronize()
        for p in self._mapped:
            lib.glass_peer_close(p)
        self._mapped = []


def attach(arena, spin_limit=SPIN_LIMIT):
    """Switch `arena`'s small-bucket exchange to the one-shot peer form; returns the PeerExchange (also at arena._peer)."""
    arena._peer = PeerExchange(arena, spin_limit)
    return arena._peer
