"""The link-prediction pre-training step as ONE explicit forward / backward program over the C ABI.

Reference: GNNEmb.py:108-163 (`work`: per batch of 131 072 edge / non-edge pairs — whole-graph node embeddings, mean over
each pair, 2-layer MLP, BCEWithLogitsLoss, backward, Adam) on EdgeGNN = EmbGConv(MyGCNConv layers) + MLP
(impl/models.py:361-509, 33-50).  Round 3 ran it on the per-op autograd path: ~97 launches per step, the Linear layers on
library GEMMs and ReLU / cat / dropout / bias / loss on framework kernels.  Here

  * the conv stack is `stack.StackProgram` in its unlabeled mode (`supported_unlabeled`): the same fused kernels as the
    GLASS step — gather inside layer 0's trans kernel, K1, GraphNorm statistics in exact accumulators, applies in the
    consumers' operand loads, data + weight gradient of a Linear in one launch, deferred weight-gradient reduction — with
    every Linear laid out as the second half of a pair (arena._pairs) and ReLU as an activation code;
  * the head is K9 (`pairhead.hip`): pair gather + Linear + dropout + ReLU + Linear + BCE in one launch, its backward in seven
    (weight partials, reduce, four for the node buckets, exact gather + the 64 x 64 product);
  * parameters live in a ParamArena, so Adam is one launch (optim.FlatAdam).

No library GEMM, no framework kernel, no float atomic (bitwise repeatable) inside the step; ~32 launches.  `PairProgram.supported`
is the gate (hidden 64, jk off, GraphNorm on, ReLU / ELU, the reference's MLP head); anything else keeps the per-op path.
"""
import torch
import torch.nn as nn

from . import _lib, ops, stack


def _head_layers(head):
    """(first Linear, dropout p, last Linear) of the reference's 2-layer MLP head (Linear, [Dropout], activation, Linear:
    impl/models.py:33-50 with num_layers = 2, gn = False, tail_activation = False) or None."""
    from .models import MLP
    if not isinstance(head, MLP):
        return None
    mods = list(head.seq.modlist)
    if len(mods) not in (3, 4) or not isinstance(mods[0], nn.Linear) or not isinstance(mods[-1], nn.Linear):
        return None
    p = 0.0
    mid = mods[1:-1]
    if len(mid) == 2:
        if not isinstance(mid[0], nn.Dropout):
            return None
        p = float(mid[0].p)
    if type(mid[-1]) is not nn.ReLU:
        return None
    return mods[0], p, mods[-1]


class PairProgram:
    def __init__(self, model):
        self.model = model
        self.prog = stack._program(model.conv)
        self.lin0, self.p_head, self.lin1 = _head_layers(model.preds[0])
        self._ws = None
        self.last = None  # tensors of the last pass, for tests: {"st", "hid", "logits"}

    @staticmethod
    def supported(model):
        """EdgeGNN(EmbGConv(MyGCNConv, hidden 64, jk off, gn on), [MLP(64, 64, 1, 2)], mean pair pool) with every parameter
        in one ParamArena."""
        from .models import EdgeGNN, EmbGConv
        if not (isinstance(model, EdgeGNN) and isinstance(model.conv, EmbGConv) and len(model.preds) == 1):
            return False
        if not stack.StackProgram.supported(model.conv):
            return False
        hl = _head_layers(model.preds[0])
        if hl is None:
            return False
        lin0, _p, lin1 = hl
        H = model.conv.input_emb.weight.shape[1]
        lib = _lib.load()
        arena = getattr(model.conv, "_glass_arena", None)
        ok = (lib.glass_pair_head_supported(H) and lin0.weight.shape == (H, H) and lin1.weight.shape == (1, H) and
              lin0.bias is not None and lin1.bias is not None and arena is not None and arena.attached())
        return bool(ok) and all(arena.offset_of(p) is not None for p in (lin0.weight, lin0.bias, lin1.weight, lin1.bias))

    def written_params(self):
        return self.prog.written_params(None) + [self.lin0.weight, self.lin0.bias, self.lin1.weight, self.lin1.bias]

    def covers_arena(self):
        arena = self.model.conv._glass_arena
        return {id(p) for p in arena.params} <= {id(p) for p in self.written_params()}

    def _scratch(self, n, P, dev):
        need = int(_lib.load().glass_pair_head_ws_bytes(n, P))
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            if self._ws is not None:
                ops._retired_scratch.append(self._ws)  # a captured graph may still launch with the old pointer
            self._ws = torch.empty(need + 16, dtype=torch.uint8, device=dev)
        return self._ws

    def loss_and_grads(self, x, edge_index, edge_weight, pairs, target, overwrite=True):
        """One training pass without the autograd tape: every parameter gradient lands in the arena (overwritten, or added
        when overwrite is False).  Returns the loss (0-d device tensor) — logits in self.last["logits"]."""
        model, lib = self.model, _lib.load()
        emb_mod = model.conv
        if x.dim() != 3 or x.shape[1] != 1 or x.shape[2] != 1:
            raise NotImplementedError("one integer feature per node (x of shape [N,1,1])")
        n = x.shape[0]
        x_flat = x.reshape(n)
        if x_flat.dtype != torch.int64:
            x_flat = x_flat.to(torch.int64)
        pairs = pairs.contiguous()
        if pairs.dtype != torch.int64:
            pairs = pairs.to(torch.int64)
        P = pairs.shape[0]
        if pairs.dim() != 2 or pairs.shape[1] != 2:
            raise ValueError("pairs must be [P, 2]")
        tgt = target.reshape(-1)
        if tgt.dtype != torch.float32 or not tgt.is_contiguous():
            tgt = tgt.contiguous().to(torch.float32)
        dev = x.device
        acc = 0 if overwrite else 1
        f32 = dict(dtype=torch.float32, device=dev)
        with torch.no_grad():
            emb, st = self.prog.forward(x_flat, None, edge_index, edge_weight, True, acc=acc)
            H = emb.shape[1]
            p = self.p_head if model.training else 0.0
            hid, logits, dlogit = torch.empty((P, H), **f32), torch.empty(P, **f32), torch.empty(P, **f32)
            loss, demb = torch.empty((), **f32), torch.empty((n, H), **f32)
            ws = self._scratch(n, P, dev)
            rng = ops.rng_tensor(dev).data_ptr() if p > 0 else 0
            W0, b0, w1, b1 = self.lin0.weight, self.lin0.bias, self.lin1.weight, self.lin1.bias
            stream = torch.cuda.current_stream().cuda_stream
            # (dropout stream 2: the embedding's is 1, layer l's GraphNorm dropouts are 16 (l + 1) [+ 1])
            _lib.check(lib.glass_pair_head_fwd_f32(emb.data_ptr(), emb.stride(0), n, pairs.data_ptr(), P, W0.data_ptr(), b0.data_ptr(),
                                                   w1.data_ptr(), b1.data_ptr(), tgt.data_ptr(), float(p), rng, 2, 0, hid.data_ptr(),
                                                   logits.data_ptr(), dlogit.data_ptr(), ws.data_ptr(), stream), "glass_pair_head_fwd_f32")
            _lib.check(lib.glass_pair_head_bwd_f32(emb.data_ptr(), emb.stride(0), n, pairs.data_ptr(), P, W0.data_ptr(), w1.data_ptr(),
                                                   hid.data_ptr(), dlogit.data_ptr(), float(p), W0.grad.data_ptr(), b0.grad.data_ptr(),
                                                   w1.grad.data_ptr(), b1.grad.data_ptr(), acc, loss.data_ptr(), demb.data_ptr(),
                                                   demb.stride(0), ws.data_ptr(), stream), "glass_pair_head_bwd_f32")
            self.last = {"st": st, "hid": hid, "logits": logits, "layers": list(st["layers"])}
            self.prog.backward(st, demb)
        return loss

    def predict(self, x, edge_index, edge_weight, pairs):
        """Evaluation forward: logits [P, 1] (no dropout in eval mode)."""
        model, lib = self.model, _lib.load()
        n = x.shape[0]
        x_flat = x.reshape(n).to(torch.int64)
        pairs = pairs.contiguous().to(torch.int64)
        P = pairs.shape[0]
        with torch.no_grad():
            emb, _ = self.prog.forward(x_flat, None, edge_index, edge_weight, False)
            logits = torch.empty(P, dtype=torch.float32, device=x.device)
            W0, b0, w1, b1 = self.lin0.weight, self.lin0.bias, self.lin1.weight, self.lin1.bias
            _lib.check(lib.glass_pair_head_fwd_f32(emb.data_ptr(), emb.stride(0), n, pairs.data_ptr(), P, W0.data_ptr(), b0.data_ptr(),
                                                   w1.data_ptr(), b1.data_ptr(), 0, 0.0, 0, 0, 0, 0, logits.data_ptr(), 0, 0,
                                                   torch.cuda.current_stream().cuda_stream), "glass_pair_head_fwd_f32")
        return logits.reshape(P, 1)


def program_for(model):
    """The model's PairProgram (cached on the model), or None when the per-op path has to serve it."""
    prog = model.__dict__.get("_glass_pair_prog")
    if prog is None and PairProgram.supported(model):
        prog = model.__dict__["_glass_pair_prog"] = PairProgram(model)
    return prog
