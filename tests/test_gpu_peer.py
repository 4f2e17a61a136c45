"""The one-shot peer exchange fused with Adam (glass_peer_allreduce_adam_f32, glass_amd/peer.py).

No multi-GPU box is available to this repository, so the path is exercised the way the verdict asks: TWO PROCESSES sharing the ONE
GPU, each rank's gradient arena and flag block mapped into the other through hipIpc handles — the same code a rank per GPU of
an xGMI node would run.  Checked: (1) arena level, synthetic per-rank gradients: three eager launches + three replays of the
launch captured in a hipGraph give BIT-identical parameters and moments to the flat path of tests/test_grad_exchange_gloo.py
(gloo all-reduce of the gradients, mean, glass_adam_step_f32) on both ranks; (2) model level: TrainStep on the step program
with peer.attach against the same TrainStep with the gloo exchange; (3) a peer that never shows up: the launch gives up after
its spin limit, sets the sticky status, leaves parameters and moments untouched — and returns (no hang)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"

WORKER = r'''
import os, sys
ROOT = sys.argv[1]; rank = int(sys.argv[2]); port = sys.argv[3]; mode = sys.argv[4]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE="2")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import torch, torch.distributed as td
td.init_process_group("gloo", rank=rank, world_size=2)
from glass_amd import _lib, peer, dist as gdist, ops, losses, synth
from glass_amd.arena import ParamArena
from glass_amd.optim import FlatAdam
from helpers import build_glass
dev = torch.device("cuda:0")
torch.cuda.set_device(0)

def cpu_mean(t):
    c = t.detach().cpu().clone()
    td.all_reduce(c, op=td.ReduceOp.SUM)
    return (c / 2).to(dev)

if mode == "arena":
    torch.manual_seed(3)
    model = build_glass(64, 2, 7, 3, "mean", "sum", 0.9).to(dev)
    twin = build_glass(64, 2, 7, 3, "mean", "sum", 0.9).to(dev)
    twin.load_state_dict(model.state_dict())
    a, b = ParamArena(model), ParamArena(twin)
    oa, ob = FlatAdam(a, lr=1e-2), FlatAdam(b, lr=1e-2)
    px = peer.attach(a)
    n = a.flat.numel()
    def grad(step):
        return torch.randn(n, generator=torch.Generator().manual_seed(1000 * step + rank)).to(dev)
    mean_out = torch.empty(n, device=dev)
    for step in range(1, 4):
        g = grad(step)
        a.flat.copy_(g)
        px.step(oa, mean_out=mean_out) if step == 1 else oa.step()
        b.flat.copy_(cpu_mean(g))
        ob.step()
        torch.cuda.synchronize()
        if step == 1:
            assert torch.equal(mean_out, b.flat), "mean gradient differs from the gloo mean"
    px.check()
    assert torch.equal(a.flat_param, b.flat_param) and torch.equal(oa.exp_avg, ob.exp_avg) and torch.equal(oa.exp_avg_sq, ob.exp_avg_sq)
    # the launch inside a captured graph
    gph = torch.cuda.CUDAGraph()
    a.flat.copy_(grad(4))
    torch.cuda.synchronize(); td.barrier()
    with torch.cuda.graph(gph):
        oa.step()
    for step in range(4, 7):
        g = grad(step)
        a.flat.copy_(g)
        gph.replay()
        b.flat.copy_(cpu_mean(g))
        ob.step()
        torch.cuda.synchronize()
    px.check()
    assert int(oa.step_dev[0]) == 6 == int(ob.step_dev[0]) and int(px.seq[0]) == 6
    assert torch.equal(a.flat_param, b.flat_param) and torch.equal(oa.exp_avg_sq, ob.exp_avg_sq)
    both = [None, None]
    td.all_gather_object(both, a.flat_param.cpu())
    assert torch.equal(both[0], both[1]), "the ranks' replicas differ"
    td.barrier(); px.close()
    print("PEER_OK arena", rank)
else:
    from glass_amd.step import TrainStep
    w, ei, ew, x, pos, y = synth.make_workload("tiny", seed=0, n_batches=8)
    ei, ew, x, pos, y = (torch.from_numpy(t).to(dev) for t in (ei, ew, x, pos, y))
    B = w.batch
    out = []
    for use_peer in (True, False):
        torch.manual_seed(0); ops.rng_seed(77, dev)
        model = build_glass(64, w.layers, int(x.max()), w.n_class, w.aggr, w.pool, w.z_ratio, dropout=0.0).to(dev).train()
        arena = ParamArena(model)
        opt = FlatAdam(arena, lr=1e-2)
        px = peer.attach(arena) if use_peer else None
        if not use_peer:
            arena.all_reduce_mean = lambda: arena.flat.copy_(cpu_mean(arena.flat))   # the flat reference path: gloo mean of the arena
        step = TrainStep(model, opt, losses.CrossEntropy(), x, ei, ew, arena, use_graph=use_peer, warmup_iters=1)
        for k in range(4):
            b = 2 * k + rank   # this rank's batch of the step
            step(pos[b * B:(b + 1) * B], y[b * B:(b + 1) * B])
        torch.cuda.synchronize()
        if px is not None:
            px.check(); td.barrier(); px.close()
        out.append(arena.flat_param.clone())
    err = float((out[0] - out[1]).abs().max() / out[1].abs().max())
    assert err < 1e-6, err
    print("PEER_OK model", rank, err)
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("mode", ["arena", "model"])
def test_two_processes_one_gpu_peer_exchange(mode, tmp_path):
    script = tmp_path / "peer_worker.py"
    script.write_text(WORKER)
    port = str(_free_port())
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, str(r), port, mode], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True, env=env) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            p.kill()
            o, _ = p.communicate()
        outs.append(o)
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"PEER_OK {mode} {r}" in o, f"rank {r}:\n{o[-3000:]}"


def test_absent_peer_times_out_without_touching_the_replica():
    """world = 2 with a 'peer' whose flag block never advances (its arena and flags are this process's own buffers): the launch
    gives up after spin_limit polls, sets the sticky status, leaves parameters, moments and the step count as they were, and
    every later launch is a no-op too."""
    import ctypes
    from glass_amd import _lib, peer
    lib = _lib.load()
    dev = torch.device(DEV)
    n = 5000
    p0 = torch.randn(n, device=dev)
    p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    g_me, g_peer = torch.randn(n, device=dev), torch.randn(n, device=dev)
    f_me, f_peer = torch.zeros(8, dtype=torch.int64, device=dev), torch.zeros(8, dtype=torch.int64, device=dev)
    grp = peer._PeerGroupStruct()
    grp.world, grp.rank = 2, 0
    grp.grad[0], grp.grad[1] = g_me.data_ptr(), g_peer.data_ptr()
    grp.flags[0], grp.flags[1] = f_me.data_ptr(), f_peer.data_ptr()
    lr = torch.full((1,), 1e-2, device=dev)
    step_dev = torch.zeros(2, dtype=torch.int64, device=dev)
    seq = torch.zeros(2, dtype=torch.int64, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        rc = lib.glass_peer_allreduce_adam_f32(ctypes.byref(grp), n, p.data_ptr(), m.data_ptr(), v.data_ptr(), lr.data_ptr(), 0.9, 0.999,
                                               1e-8, 0.0, step_dev.data_ptr(), seq.data_ptr(), status.data_ptr(), 2000, 0, st)
        assert rc == 0, lib.glass_last_error_string()
        torch.cuda.synchronize()
    assert int(status[0]) != 0 and int(step_dev[0]) == 0 and int(f_me[0]) >= 1
    assert torch.equal(p, p0) and float(m.abs().max()) == 0.0 and float(v.abs().max()) == 0.0
    # with the peer present (its flags ahead of every sequence number) and a fresh status the same launch updates
    status.zero_()
    f_peer.fill_(1 << 40)
    rc = lib.glass_peer_allreduce_adam_f32(ctypes.byref(grp), n, p.data_ptr(), m.data_ptr(), v.data_ptr(), lr.data_ptr(), 0.9, 0.999, 1e-8,
                                           0.0, step_dev.data_ptr(), seq.data_ptr(), status.data_ptr(), 2000, 0, st)
    assert rc == 0
    torch.cuda.synchronize()
    assert int(status[0]) == 0 and int(step_dev[0]) == 1 and not torch.equal(p, p0)
    ref_p, ref_m, ref_v, ref_step = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(2, dtype=torch.int64, device=dev)
    gm = (g_me + g_peer) * 0.5
    lib.glass_adam_step_f32(ref_p.data_ptr(), gm.data_ptr(), ref_m.data_ptr(), ref_v.data_ptr(), n, lr.data_ptr(), 0.9, 0.999, 1e-8, 0.0,
                            ref_step.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(p, ref_p) and torch.equal(m, ref_m) and torch.equal(v, ref_v)
