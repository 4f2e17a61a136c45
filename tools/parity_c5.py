"""Builder-run parity record for BASELINE config 5's family near its own size: the step program
(fp32, MI355X) against the fp64 CPU oracle on the power-law graph generator of config 5, hidden 256 —
  (1) N = 250 000, 5 M edges, 2 layers (the full model on a quarter of the graph);
  (2) N = 1 000 000, 20 M edges (the full graph), 1 layer — only when the host has the memory for the fp64 tape.
Appends to gpurun_out/parity_r03.json (tests/helpers.record_parity); copy to profiles/.  Not a per-round test: ~3-6 min.
usage (GPU box): python tools/parity_c5.py [quarter|full|both|config5]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import psutil
import torch

from helpers import build_glass, flat_grads, record_parity, rel_inf
from oracle import glass_oracle as O

DEV = "cuda:0"


def run(tag, n_node, n_pairs, layers, need_gb, fp32_floor):
    from glass_amd import synth, stack, losses
    from glass_amd.arena import ParamArena
    avail = psutil.virtual_memory().available / 2**30
    if avail < need_gb:
        print(f"{tag}: skipped — {avail:.0f} GiB of host memory available, the fp64 tape needs ~{need_gb} GiB")
        record_parity(f"config5_family/{tag}", skipped=f"host memory {avail:.0f} GiB < {need_gb} GiB")
        return
    w = synth.WORKLOADS["powerlaw"]
    t0 = time.time()
    ei, ew = synth.make_graph(n_node, n_pairs, 0, w.powerlaw)
    x = synth.degree_feature(ei, n_node)
    pos, y = synth.make_subgraphs(n_node, w.batch, w.sub_size, w.n_class, 1, w.multilabel)
    ei, ew, x, pos, y = (torch.from_numpy(a) for a in (ei, ew, x, pos, y))
    print(f"{tag}: graph N={n_node} nnz={ei.shape[1]} V={int(x.max()) + 1} built in {time.time() - t0:.0f} s", flush=True)
    torch.manual_seed(0)
    model = build_glass(w.hidden, layers, int(x.max()), w.n_class, w.aggr, w.pool, w.z_ratio)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    loss_fn = losses.CrossEntropy()
    model.to(DEV).train()
    arena = ParamArena(model)
    assert stack.step_supported(model, loss_fn) and stack.covers_arena(model, arena)
    xg, eig, ewg, posg, yg = (t.to(DEV) for t in (x, ei, ew, pos, y))
    loss, logits = stack.loss_and_grads(model, loss_fn, xg, eig, ewg, posg, "pos", yg, overwrite=True)
    torch.cuda.synchronize()
    mine = {k: p.grad.cpu() for k, p in model.named_parameters()}
    keys = sorted(mine)
    loss, logits = loss.item(), logits.cpu()
    res = {}
    for dt in ([torch.float64, torch.float32] if fp32_floor else [torch.float64]):
        t0 = time.time()
        orc = O.OracleGLASS(w.hidden, layers, int(x.max()), w.n_class, aggr=w.aggr, pool=w.pool, z_ratio=w.z_ratio)
        orc.load_state_dict(sd)
        orc = orc.to(dt).train()
        po = orc(x, ei, ew.to(dt), pos, O.max_zero_one(x, pos))
        lo = loss_fn(po, y)
        lo.backward()
        res[dt] = (po.detach().double(), lo.item(), flat_grads({k: p.grad for k, p in orc.named_parameters()}, keys).double())
        del orc, po, lo
        print(f"{tag}: oracle {dt} in {time.time() - t0:.0f} s", flush=True)
    po, lo, g64 = res[torch.float64]
    rec = dict(n_node=float(n_node), nnz=float(ei.shape[1]), hidden=float(w.hidden), layers=float(layers),
               logits_rel_inf=rel_inf(logits, po), loss_rel=abs(loss - lo) / abs(lo), grad_rel_inf=rel_inf(flat_grads(mine, keys), g64))
    if fp32_floor:
        rec.update(oracle_fp32_vs_fp64_logits=rel_inf(res[torch.float32][0], po), oracle_fp32_vs_fp64_grad=rel_inf(res[torch.float32][2], g64))
    print(tag, {k: (f"{v:.2e}" if v < 1 else v) for k, v in rec.items()}, flush=True)
    record_parity(f"config5_family/{tag}", **rec)
    assert rec["logits_rel_inf"] < 1e-5 and rec["loss_rel"] < 1e-5 and rec["grad_rel_inf"] < max(1e-5, 2 * rec.get("oracle_fp32_vs_fp64_grad", 0.0))


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "both"
    torch.set_num_threads(os.cpu_count())
    if what in ("quarter", "both"):
        run("powerlaw_N250k_hidden256_L2_vs_fp64", 250_000, 2_500_000, 2, need_gb=56, fp32_floor=True)
    if what in ("full", "both"):
        run("powerlaw_N1M_hidden256_L1_vs_fp64", 1_000_000, 10_000_000, 1, need_gb=150, fp32_floor=False)
    if what in ("config5", ):   # BASELINE config 5 itself: the full graph, both layers (~300 GiB of fp64 tape on the host)
        run("powerlaw_N1M_hidden256_L2_vs_fp64_FULL_CONFIG5", 1_000_000, 10_000_000, 2, need_gb=400, fp32_floor=False)
