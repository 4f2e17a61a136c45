"""Self-supervised pre-training of node embeddings by link prediction on the MI355X path — the workload of
the reference's GNNEmb.py (EdgeGNN = EmbGConv(MyGCNConv) + 2-layer MLP, BCE on edge / non-edge pairs,
`work()` 108-163, search space `obj()` 169-192), with the same flags and log lines.

Optuna is not available here, so the hyper-parameter search is a seeded random search over the same space
(conv_layer 2..5, dropout 0..0.5 step 0.1, aggr sum/mean/gcn; hidden 64, lr 1e-3, batch 131072, jk off); the
best trial's embeddings go to `<path><name>_64.pt`, the file `GLASSTest.py --use_nodeid` loads.

    python GNNEmb.py --use_nodeid --device 0 --dataset density --name density --optruns 3
"""
import argparse
import os
import functools
import itertools
import random

import numpy as np
import torch
import torch.nn as nn
from torch.nn import BCEWithLogitsLoss
from torch.optim import Adam, lr_scheduler

import datasets
from impl import SubGDataset, config, metrics, models, train


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="")
    p.add_argument("--dataset", type=str, default="ppi_bp")
    p.add_argument("--use_deg", action="store_true")
    p.add_argument("--use_one", action="store_true")
    p.add_argument("--use_nodeid", action="store_true")
    p.add_argument("--repeat", type=int, default=1)
    p.add_argument("--test", action="store_true")
    p.add_argument("--abl", action="store_true")
    p.add_argument("--optruns", type=int, default=100)
    p.add_argument("--path", type=str, default="Emb/")
    p.add_argument("--name", type=str, default="opt")
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--use_seed", action="store_true")
    p.add_argument("--seed", type=int, default=0)
    p.add_argument("--max_epoch", type=int, default=100, help="(extension) cap on epochs per trial")
    return p.parse_args(argv)


class GraphedPairStep:
    """Forward + backward of one link batch, replayed from a hipGraph once a batch shape has been seen.

    The pre-training step is ~95 short launches (0.8 ms of kernel time at ppi_bp-shape, tools/ssl_step.py) that an eager
    loop enqueues in ~1.8 ms: host-bound.  Every kernel of libglass_hip only enqueues work on the caller's stream, so the
    per-op autograd path captures as it is (the way impl.train's TrainStep does for the GLASS step).  A shape's first batch
    runs eagerly (it builds the CSR / plans / scratch buffers), the second one is captured, later ones copy the batch into
    the graph's static buffers and replay.  The optimizer (and the reference's per-batch plateau scheduler, which needs the
    loss on the host) stay outside the graph.  GLASS_SSL_GRAPH=0 keeps the eager loop."""
    def __init__(self, model, loss_fn, x, edge_index, edge_attr, bce_mean=False):
        """bce_mean: loss_fn is BCEWithLogitsLoss()(pred.flatten(), target.flatten()) (the reference's, GNNEmb.py:129-130) —
        then a model the step program serves (glass_amd.ssl.PairProgram.supported: hidden 64, ParamArena attached) runs the
        whole forward + backward as that program: no autograd tape, no library GEMM, ~32 launches instead of ~97."""
        self.model, self.loss_fn = model, loss_fn
        self.x, self.ei, self.ea = x, edge_index, edge_attr
        self.seen, self.graphs = set(), {}
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.enabled = os.environ.get("GLASS_SSL_GRAPH", "1") != "0"
        from glass_amd import ssl
        self.program = ssl.program_for(model) if (bce_mean and os.environ.get("GLASS_SSL_PROGRAM", "1") != "0") else None
        self.key = (id(loss_fn), id(x), id(edge_index), id(edge_attr))  # what a cached step is valid for (train_epoch)

    def _fwd_bwd(self, pairs, target):
        if self.program is not None and self.model.training:
            # gradients are written into the arena (every .grad is a view of it), overwriting: no zero-fill
            return self.program.loss_and_grads(self.x, self.ei, self.ea, pairs, target, overwrite=self.program.covers_arena())
        emb = self.model.NodeEmb(self.x, self.ei, self.ea)
        loss = self.loss_fn(self.model.preds[0](self.model.Pool(emb, pairs, None)), target)
        loss.backward()
        return loss.detach()

    def _reset_grads(self):
        if self.program is None:
            for p in self.params:
                p.grad = None
        elif not self.program.covers_arena():
            self.model.conv._glass_arena.zero()

    def __call__(self, pairs, target):
        """Leaves the batch's gradients in .grad of every parameter and returns the loss (a 0-d device tensor)."""
        key = (tuple(pairs.shape), tuple(target.shape), target.dtype)
        ent = self.graphs.get(key)
        if ent is None and not (self.enabled and key in self.seen):
            self.seen.add(key)
            self._reset_grads()
            return self._fwd_bwd(pairs, target)
        if ent is None:
            s_pairs, s_target = pairs.clone(), target.clone()
            self._reset_grads()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            loss = torch.zeros((), device=pairs.device)  # caller-owned: outside the graph's private pool
            try:
                with torch.cuda.graph(g):
                    loss.copy_(self._fwd_bwd(s_pairs, s_target))
            except Exception as e:  # noqa: BLE001 — whatever the runtime refuses to capture: keep training eagerly
                print(f"GraphedPairStep: capture refused ({e!r}); continuing with the eager step", flush=True)
                self.enabled = False
                torch.cuda.synchronize()
                self._reset_grads()
                return self._fwd_bwd(pairs, target)
            ent = self.graphs[key] = (g, s_pairs, s_target, loss, [p.grad for p in self.params])
        g, s_pairs, s_target, loss, grads = ent
        s_pairs.copy_(pairs)
        s_target.copy_(target)
        if self.program is not None and not self.program.covers_arena():
            self.model.conv._glass_arena.zero()
        g.replay()
        if self.program is None:
            for p, gr in zip(self.params, grads):  # (an eager batch of another shape in between re-pointed .grad)
                p.grad = gr
        return loss


class Pretrain:
    def __init__(self, args):
        self.args = args
        g = datasets.load_dataset(args.dataset)
        if args.use_deg:
            g.setDegreeFeature()
        elif args.use_one:
            g.setOneFeature()
        elif args.use_nodeid:
            g.setNodeIdFeature()
        else:
            raise NotImplementedError
        self.max_deg = torch.max(g.x)
        g.to(config.device)
        x, ei, ea, pos, y = g.get_LPdataset()
        idx = torch.randperm(pos.shape[0], device=pos.device)
        cut = int(0.95 * idx.shape[0])
        self.trn = SubGDataset.GDataset(x, ei, ea, pos[idx[:cut]], y[idx[:cut]])
        self.val = SubGDataset.GDataset(x, ei, ea, pos[idx[cut:]], y[idx[cut:]])

    def build_model(self, hidden_dim, conv_layer, dropout, jk, aggr):
        conv = models.EmbGConv(hidden_dim, hidden_dim, hidden_dim, conv_layer, max_deg=self.max_deg,
                               activation=nn.ReLU(inplace=True), jk=jk, dropout=dropout,
                               conv=functools.partial(models.MyGCNConv, aggr=aggr), gn=True)
        head = models.MLP(hidden_dim * conv_layer if jk else hidden_dim, hidden_dim, 1, 2, dropout=dropout,
                          activation=nn.ReLU(inplace=True))
        return models.EdgeGNN(conv, nn.ModuleList([head]), nn.ModuleList([models.MeanPool()])).to(config.device)

    @staticmethod
    def make_optimizer(model, lr):
        """Adam as the reference builds it (GNNEmb.py:116).  Where the step program can serve the model (hidden 64, jk off) the
        parameters are flattened into one arena first and Adam is ONE launch over it (glass_amd.optim.FlatAdam — an
        Optimizer subclass, so the per-batch ReduceLROnPlateau works unchanged); otherwise torch's Adam."""
        from glass_amd import ssl
        from glass_amd.arena import ParamArena
        from glass_amd.optim import FlatAdam
        if os.environ.get("GLASS_SSL_PROGRAM", "1") != "0":
            try:
                arena = ParamArena(model)
            except Exception:  # noqa: BLE001 — any layout the arena does not take: plain Adam on the per-op path
                arena = None
            if arena is not None and ssl.program_for(model) is not None:
                return FlatAdam(arena, lr=lr)
        return Adam(model.parameters(), lr=lr)

    def node_embeddings(self, model):
        with torch.no_grad():
            return model.NodeEmb(self.trn.x, self.trn.edge_index, self.trn.edge_attr).detach().cpu()

    def train_epoch(self, model, optimizer, scheduler, loader, loss_fn, max_batches=10):
        """At most `max_batches` link-prediction batches (the whole graph is re-embedded for each); the plateau scheduler
        is stepped per batch, before the optimizer.  Returns the mean batch loss."""
        model.train()
        step = model.__dict__.get("_pair_step")
        key = (id(loss_fn), id(self.trn.x), id(self.trn.edge_index), id(self.trn.edge_attr))
        if step is None or step.key != key:  # (a cached step holds its first call's loss_fn and graph tensors: ADVICE r3)
            step = model.__dict__["_pair_step"] = GraphedPairStep(model, loss_fn, self.trn.x, self.trn.edge_index,
                                                                  self.trn.edge_attr, bce_mean=getattr(loss_fn, "bce_mean", False))
        seen = []
        for batch in itertools.islice(loader, max_batches):
            loss = step(batch[-2], batch[-1])
            scheduler.step(loss)
            seen.append(loss.item())
            optimizer.step()
        return np.average(seen)

    def work(self, hidden_dim, conv_layer, dropout, jk, lr, batch_size, aggr):
        """Score one hyper-parameter set: `repeat` runs of up to max_epoch epochs, validated every 5th epoch, stopped
        after 3 validations without improvement.  Returns (mean - std of the best validation F1, embeddings of the
        last run's best checkpoint)."""
        trn_loader = SubGDataset.GDataloader(self.trn, batch_size)
        val_loader = SubGDataset.GDataloader(self.val, self.val.y.shape[0], shuffle=False)

        def loss_fn(pred, target):
            return BCEWithLogitsLoss()(pred.flatten(), target.flatten())
        loss_fn.bce_mean = True  # (lets the pair step run as the fused program: glass_amd/ssl.py)

        scores, emb = [], None
        for _ in range(self.args.repeat):
            model = self.build_model(hidden_dim, conv_layer, dropout, jk, aggr)
            optimizer = self.make_optimizer(model, lr)
            emb = self.node_embeddings(model)
            scheduler = lr_scheduler.ReduceLROnPlateau(optimizer, factor=0.7, min_lr=5e-5, patience=50)
            best, stale = 0.0, 0
            for epoch in range(self.args.max_epoch):
                mean_loss = self.train_epoch(model, optimizer, scheduler, trn_loader, loss_fn)
                if epoch % 5:
                    print(f"iter {epoch} loss {mean_loss}", flush=True)
                    continue
                score, _ = train.test(model, val_loader, metrics.binaryf1, loss_fn)
                print(f"iter {epoch} loss {mean_loss} score {score}", flush=True)
                if score > best:
                    best, stale, emb = score, 0, self.node_embeddings(model)
                else:
                    stale += 1
                    if stale >= 3:
                        break
            scores.append(best)
        return np.average(scores) - np.std(scores), emb


def main(argv=None):
    args = parse_args(argv)
    config.set_device(args.device)
    if config.device.type != "cuda":
        raise SystemExit("this driver runs the MI355X HIP path only")
    if args.use_seed:
        random.seed(args.seed)
        np.random.seed(args.seed)
        torch.manual_seed(args.seed)
    run = Pretrain(args)
    print(args)
    rng = random.Random(args.seed)
    best_score, best_params = 0.0, None
    for trial in range(args.optruns):
        params = dict(conv_layer=rng.randint(2, 5), dropout=rng.choice([0.0, 0.1, 0.2, 0.3, 0.4, 0.5]),
                      aggr=rng.choice(["sum", "mean", "gcn"]))
        score, emb = run.work(64, params["conv_layer"], params["dropout"], False, 1e-3, 131072, params["aggr"])
        print(f"trial {trial} params {params} score {score}", flush=True)
        if score > best_score:
            out_dir = os.path.dirname(f"{args.path}{args.name}_64.pt")
            if out_dir:
                os.makedirs(out_dir, exist_ok=True)  # a fresh checkout has no Emb/ directory
            torch.save(emb, f"{args.path}{args.name}_64.pt")
            best_score, best_params = score, params
    print("best params ", best_params)
    print("best valf1 ", best_score)
    return best_score, best_params


if __name__ == "__main__":
    main()
