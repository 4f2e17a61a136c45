"""GLASS training driver on the MI355X path — same command line, YAML keys, control flow and log
lines as the reference's GLASSTest.py (flags 14-30, split 77-126, buildModel 129-175, test 178-269):

    python GLASSTest.py --use_one --use_seed --use_maxzeroone --repeat 1 --device 0 --dataset density

It builds what the reference builds — `Adam(gnn.parameters(), lr)`, `ReduceLROnPlateau`, the binary loss as a function
around `BCEWithLogitsLoss`, `CrossEntropyLoss()` — and nothing of glass_amd by name: `impl.train.train` itself moves such a
caller onto the captured step program (flat parameter arena, fused Adam, fused head + loss; glass_amd/optim.py `adopt`,
glass_amd/losses.py `fusable_mode`).  The reference's own GLASSTest.py run against this repo's `impl/` gets the same path.
Extension: `--dataset synthetic:<workload>` selects a seeded synthetic graph (glass_amd/synth.py).  GPU only:
`--device -1` stops with an error.
"""
import argparse
import functools
import random
import time

import numpy as np
import torch
import torch.nn as nn
import yaml
from torch.nn import BCEWithLogitsLoss, CrossEntropyLoss
from torch.optim import Adam, lr_scheduler

import datasets
from impl import SubGDataset, config, metrics, models, train, utils

SYNTHETIC_SETS = ("density", "component", "cut_ratio", "coreness")
POOLS = {"mean": models.MeanPool, "max": models.MaxPool, "sum": models.AddPool, "size": models.SizePool}


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="")
    p.add_argument("--dataset", type=str, default="ppi_bp")
    # node features: degree rank, all-ones, or pretrained node-id embeddings from ./Emb
    p.add_argument("--use_deg", action="store_true")
    p.add_argument("--use_one", action="store_true")
    p.add_argument("--use_nodeid", action="store_true")
    p.add_argument("--use_maxzeroone", action="store_true")
    p.add_argument("--repeat", type=int, default=1)
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--use_seed", action="store_true")
    p.add_argument("--max_epoch", type=int, default=300, help="(extension) cap on epochs per repeat")
    return p.parse_args(argv)


def set_seed(seed: int):
    print("seed ", seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


class Run:
    """State the reference keeps in module globals (baseG, datasets, loaders, task type)."""
    def __init__(self, args):
        self.args = args
        base = datasets.load_dataset(args.dataset)
        if base.y.unique().shape[0] == 2:  # binary / multi-label: BCE on flattened logits + micro-F1 of (logit > 0)
            def loss_fn(x, y):  # (the reference's function, GLASSTest.py:57-58; the training step recognises what it computes)
                return BCEWithLogitsLoss()(x.flatten(), y.flatten())

            self.loss_fn = loss_fn
            self.output_channels = base.y.shape[1] if base.y.ndim > 1 else 1
            self.score_fn = metrics.binaryf1
        else:  # multi-class
            self.loss_fn = CrossEntropyLoss()
            self.output_channels = base.y.unique().shape[0]
            self.score_fn = metrics.microf1
        self.max_deg = 0
        self.trn = self.val = self.tst = None

    def split(self):
        """Load, pick the node feature, move to the device, split (reference split(), 77-126)."""
        a = self.args
        g = datasets.load_dataset(a.dataset)
        g.y = g.y.to(torch.float) if g.y.unique().shape[0] == 2 else g.y.to(torch.int64)
        if a.use_deg:
            g.setDegreeFeature()
        elif a.use_one:
            g.setOneFeature()
        elif a.use_nodeid:
            g.setNodeIdFeature()
        else:
            raise NotImplementedError
        self.max_deg = torch.max(g.x)
        g.to(config.device)
        self.trn = SubGDataset.GDataset(*g.get_split("train"))
        self.val = SubGDataset.GDataset(*g.get_split("valid"))
        self.tst = SubGDataset.GDataset(*g.get_split("test"))

    def loaders(self, batch_size):
        """(train, valid, test) loaders.  With --use_maxzeroone every batch is labeled by utils.MaxZOZ and the training
        loader drops the last partial batch; without it plain loaders are used and nothing is dropped.  All shuffle."""
        if self.args.use_maxzeroone:
            make = functools.partial(SubGDataset.ZGDataloader, z_fn=utils.MaxZOZ, shuffle=True)
            # (under torch.distributed the TRAINING loader hands every rank its slice of each batch — subgraph-batch data
            # parallelism, glass_amd/dist.py; evaluation loaders are never sharded, so scores agree on every rank)
            from glass_amd import dist as gdist
            return (make(self.trn, batch_size, drop_last=True, shard=gdist.is_distributed()),
                    make(self.val, batch_size, drop_last=False), make(self.tst, batch_size, drop_last=False))
        from glass_amd import dist as gdist
        return (SubGDataset.GDataloader(self.trn, batch_size, shuffle=True, drop_last=False, shard=gdist.is_distributed()),
                SubGDataset.GDataloader(self.val, batch_size, shuffle=True, drop_last=False),
                SubGDataset.GDataloader(self.tst, batch_size, shuffle=True, drop_last=False))

    def build_model(self, hidden_dim, conv_layer, dropout, jk, pool, z_ratio, aggr):
        a = self.args
        if pool not in POOLS:
            raise NotImplementedError

        def build(h):
            conv = models.EmbZGConv(h, h, conv_layer, max_deg=self.max_deg,
                                    activation=nn.ELU(inplace=True), jk=jk, dropout=dropout,
                                    conv=functools.partial(models.GLASSConv, aggr=aggr, z_ratio=z_ratio, dropout=dropout),
                                    gn=True)
            if a.use_nodeid:
                print("load ", f"./Emb/{a.dataset}_{hidden_dim}.pt")
                emb = torch.load(f"./Emb/{a.dataset}_{hidden_dim}.pt", map_location=torch.device("cpu")).detach()
                if h > emb.shape[1]:  # (a width no kernel family serves runs zero-padded: glass_amd/widths.py)
                    emb = torch.nn.functional.pad(emb, (0, h - emb.shape[1]))
                conv.input_emb = nn.Embedding.from_pretrained(emb, freeze=False)
            mlp = nn.Linear(h * conv_layer if jk else h, self.output_channels)
            return models.GLASS(conv, nn.ModuleList([mlp]), nn.ModuleList([POOLS[pool]()]))

        # widths outside the kernel families (every shipped config is inside) run at the next family width, zero padded: exact
        from glass_amd import widths
        return widths.build_at_fused_width(hidden_dim, build).to(config.device)

    def evaluate(self, model, loader):
        """Score of the task metric on one split."""
        return train.test(model, loader, self.score_fn, loss_fn=self.loss_fn)[0]

    def fit_once(self, model, optimizer, scheduler, loaders, warmup_epochs, patience):
        """One training run.  Validation starts after `warmup_epochs`; the test score is the one measured at the best
        validation score (ties within 1e-5 keep the larger test score).  Returns (epochs, seconds, val, test)."""
        trn_loader, val_loader, tst_loader = loaders
        policy = Patience(patience)
        seconds = 0.0
        epoch = 0
        for epoch in range(self.args.max_epoch):
            start = time.time()
            loss = train.train(optimizer, model, trn_loader, self.loss_fn)
            seconds += time.time() - start
            scheduler.step(loss)
            if epoch >= warmup_epochs:
                val = self.evaluate(model, val_loader)
                verdict = policy.judge(val)
                if verdict == "better":
                    policy.test_at_best = self.evaluate(model, tst_loader)
                    print(f"iter {epoch} loss {loss:.4f} val {policy.best_val:.4f} tst {policy.test_at_best:.4f}", flush=True)
                elif verdict == "tie":
                    tst = self.evaluate(model, tst_loader)
                    policy.test_at_best = max(tst, policy.test_at_best)
                    print(f"iter {epoch} loss {loss:.4f} val {policy.best_val:.4f} tst {tst:.4f}", flush=True)
                elif epoch % 10 == 0:  # worse: a progress line every tenth epoch only
                    print(f"iter {epoch} loss {loss:.4f} val {val:.4f} tst {self.evaluate(model, tst_loader):.4f}", flush=True)
            if policy.saturated():
                policy.strikes += 1
            if policy.exhausted():
                break
        return epoch + 1, seconds, policy.best_val, policy.test_at_best

    def test(self, pool="size", aggr="mean", hidden_dim=64, conv_layer=8, dropout=0.3, jk=1, lr=1e-3, z_ratio=0.8,
             batch_size=None, resi=0.7):
        """Train `repeat` times with one hyper-parameter set (the YAML keys); prints the reference's log lines."""
        # evaluation cadence: both the warm-up and the patience are 100 test-set batches' worth of epochs (synthetic
        # sets: a fifth of that)
        batches_in_test = self.tst.y.shape[0] / batch_size
        if self.args.dataset in SYNTHETIC_SETS:
            batches_in_test /= 5
        horizon = 100 / batches_in_test
        results = []
        for repeat in range(self.args.repeat):
            set_seed((1 << repeat) - 1)
            print(f"repeat {repeat}")
            self.split()
            model = self.build_model(hidden_dim, conv_layer, dropout, jk, pool, z_ratio, aggr)
            optimizer = Adam(model.parameters(), lr=lr)  # (impl.train.train runs it as one fused launch inside the step's graph)
            scheduler = lr_scheduler.ReduceLROnPlateau(optimizer, factor=resi, min_lr=5e-5)
            epochs, seconds, val, tst = self.fit_once(model, optimizer, scheduler, self.loaders(batch_size), horizon, horizon)
            print(f"end: epoch {epochs}, train time {seconds:.2f} s, val {val:.3f}, tst {tst:.3f}", flush=True)
            results.append(tst)
        print(f"average {np.average(results):.3f} error {np.std(results) / np.sqrt(len(results)):.3f}")
        return results


class Patience:
    """Early-stop bookkeeping of the driver: an epoch is `better` (new best validation score), a `tie` (within 1e-5
    below the best) or `worse` (one strike).  A saturated validation score (>= 1 - 1e-5) also costs a strike per
    epoch.  More strikes than `limit` end the run; a better epoch clears them."""
    def __init__(self, limit):
        self.limit = limit
        self.best_val = 0
        self.test_at_best = 0
        self.strikes = 0

    def judge(self, val):
        if val > self.best_val:
            self.best_val, self.strikes = val, 0
            return "better"
        if val >= self.best_val - 1e-5:
            return "tie"
        self.strikes += 1
        return "worse"

    def saturated(self):
        return self.best_val >= 1 - 1e-5

    def exhausted(self):
        return self.strikes > self.limit


def main(argv=None):
    args = parse_args(argv)
    config.set_device(args.device)
    if config.device.type != "cuda":
        raise SystemExit("this driver runs the MI355X HIP path only; use the reference itself for --device -1")
    if args.use_seed:
        set_seed(0)
    run = Run(args)
    print(args)
    with open(f"config/{args.dataset.replace(':', '_')}.yml") as f:
        params = yaml.safe_load(f)
    print("params", params, flush=True)
    run.split()
    return run.test(**params)


if __name__ == "__main__":
    main()
