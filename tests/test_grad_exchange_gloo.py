"""dist.GradExchange (VERDICT r01 task 7): the bucketed exchange — all-reduce of the small bucket,
reduce-scatter + sharded optimizer + all-gather for the embedding-sized bucket — gives, bit for bit, what the flat
path gives (one all-reduce of everything, every rank updating everything).  gloo, world_size 2, CPU tensors; the
optimizer is an elementwise Adam written with torch ops (the product's is glass_adam_step_f32 on the GPU)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as td
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD = 2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _adam_(p, g, m, v, step, lr=1e-2, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam's single-tensor update, elementwise, in place."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1**step, 1 - b2**step
    p.addcdiv_(m, (v.sqrt() / bc2**0.5).add_(eps), value=-lr / bc1)


def _worker(rank, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(WORLD))
    sys.path.insert(0, ROOT)
    td.init_process_group("gloo", rank=rank, world_size=WORLD)
    from glass_amd import dist as gdist
    small, big = 1003, 2 * 3001  # big: a multiple of the world size; small: deliberately odd
    total = small + big
    gen = torch.Generator().manual_seed(5)
    param0 = torch.randn(total, generator=gen)
    # flat path: one all-reduce of everything, every rank updates everything
    p_flat, m_flat, v_flat = param0.clone(), torch.zeros(total), torch.zeros(total)
    # bucketed path
    p_b, m_b, v_b = param0.clone(), torch.zeros(total), torch.zeros(total)
    g_b = torch.zeros(total)
    ex = gdist.GradExchange(g_b, p_b, small)
    assert ex.has_big and ex.shard_len == big // WORLD and ex.shard_lo == small + rank * ex.shard_len
    for step in range(1, 4):
        local = torch.randn(total, generator=torch.Generator().manual_seed(100 * step + rank))  # this rank's gradient
        g = local.clone()
        td.all_reduce(g, op=td.ReduceOp.SUM)
        g.div_(WORLD)
        _adam_(p_flat, g, m_flat, v_flat, step)
        g_b.copy_(local)
        ex.reduce_small()
        sg = ex.reduce_big()
        _adam_(p_b[:small], g_b[:small], m_b[:small], v_b[:small], step)
        ps, ms, vs = ex.shard_views(p_b, m_b, v_b)
        _adam_(ps, sg, ms, vs, step)
        ex.gather_params()
        assert torch.equal(g_b[:small], g[:small])                                   # small bucket: the same mean gradient
        assert torch.equal(sg, g[ex.shard_lo:ex.shard_lo + ex.shard_len])            # own shard of the big bucket too
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), flat=p_flat.numpy(), bucketed=p_b.numpy())
    assert ex.payload_bytes() == {"small_allreduce": small * 4, "big_reduce_scatter": big * 4, "big_all_gather": big * 4}
    # no big bucket: only the all-reduce runs
    ex0 = gdist.GradExchange(torch.ones(8) * (rank + 1), torch.zeros(8), 8)
    ex0.reduce_small()
    assert ex0.reduce_big() is None and not ex0.has_big and torch.equal(ex0.grad, torch.full((8, ), 1.5))
    td.barrier()
    td.destroy_process_group()


def test_bucketed_exchange_equals_flat_allreduce(tmp_path):
    mp.spawn(_worker, args=(_free_port(), str(tmp_path)), nprocs=WORLD, join=True)
    r = [np.load(tmp_path / f"rank{k}.npz") for k in range(WORLD)]
    for k in range(WORLD):
        assert np.array_equal(r[k]["flat"], r[k]["bucketed"])       # bit for bit, on every rank
    assert np.array_equal(r[0]["bucketed"], r[1]["bucketed"])       # and the ranks agree


def test_big_bucket_must_divide_by_world(tmp_path):
    """Host logic without a process group: world 1 accepts anything; the arena pads for the world it was built in."""
    sys.path.insert(0, ROOT)
    from glass_amd import dist as gdist
    assert gdist.world_size() == 1 and not gdist.is_distributed()
