"""One training step of the hot loop (reference impl/train.py:10-16 + ZGDataloader), optionally
replayed from a captured hipGraph.

The step is a chain of 40-100 short dependent kernels on small graphs (ppi_bp-shape: 37 launches in the step-program
form, ~0.33 ms of GPU time), so eager launches are host-bound; every kernel in libglass_hip only enqueues work on the
caller's stream (no allocation, no sync, no memset / memcpy nodes), which makes the whole step capturable:
labels -> forward -> loss -> backward -> [all-reduce] -> Adam.  Dropout masks still change every replay
because the dropout (seed, step) pair lives in device memory and is advanced by a captured kernel.
With more than one rank the collectives stay outside the graph (captured forward/backward, eager RCCL exchange of
the gradient arena, eager fused Adam launches).  The exchange is bucketed (dist.GradExchange): the small bucket
(layer / head parameters) is all-reduced; an embedding-sized bucket (`--use_nodeid`) is reduce-scattered, Adam updates
this rank's shard, and the updated parameter shards are all-gathered.  With such a bucket the backward pass is captured
as TWO graphs cut where the small bucket becomes final, and its all-reduce runs on a second stream beside the rest of
the backward (emb_gn + embedding gradient).  (The big bucket's reduce-scatter is issued on the same process group, i.e.
on RCCL's one internal stream for that group: it queues behind the small all-reduce — only the overlap with the backward
tail is real.)"""
import os

import torch

from . import dist as gdist
from . import utils


CAPTURE_COLLECTIVE = os.environ.get("GLASS_CAPTURE_COLLECTIVE", "1") != "0"  # try the RCCL exchange + Adam inside the step's graph
# A captured collective is trusted only after ONE replay of the graph has reproduced an eager step (forward/backward, eager
# RCCL all-reduce, eager Adam) from the same state: exchanged gradient arena and updated parameters compared on every rank,
# the verdict agreed on by all ranks.  On a mismatch the process keeps running on the split form (graph + eager collective).
VERIFY_CAPTURED_COLLECTIVE = os.environ.get("GLASS_VERIFY_CAPTURED_COLLECTIVE", "1") != "0"
_VERIFY_TOL = 1e-5
# (Prefetching the label launch of step i + 1 on a side stream beside step i's graph — two label / batch buffer sets, the step
# captured once per set — was measured in round 4 at ppi_bp-shape: 0.2688-0.2694 ms/step with it against 0.2448 without;
# the cross-stream event in front of every replay costs ~25 us, three times what the hidden launch returns.  Not part of
# the product; DESIGN.md §7 keeps the record.)


class TrainStep:
    def __init__(self, model, optimizer, loss_fn, x, edge_index, edge_weight, bucket=None, use_graph=True,
                 warmup_iters=3, preserve_state=False):
        """preserve_state: the warm-up steps are run for their side effects only (CSR / plan / workspace / BLAS
        handle creation) and parameters, optimizer state and dropout stream are restored afterwards, so the
        training trajectory is exactly the one of a loop without warm-up (used by impl.train.train)."""
        self.model, self.opt, self.loss_fn = model, optimizer, loss_fn
        self.x, self.ei, self.ew = x, edge_index, edge_weight
        self.bucket = bucket if bucket is not None else gdist.bucket_for(model)
        self.use_graph = use_graph
        self.warmup_iters = warmup_iters
        self.preserve_state = preserve_state
        self.graphed = False
        self._pos = self._y = None
        self._labels = None  # stack.BatchLabels of the step program: label bytes + unique labeled rows of the current batch
        self._loss = torch.zeros((), device=x.device)
        # running sum of the step losses since reset_loss_sum(): kept by the step itself (the program's readout adds to it in
        # the launch that stores the loss; other forms add one captured kernel) so that an epoch loop needs no per-step launch
        self._loss_sum = torch.zeros((), dtype=torch.float32, device=x.device)
        self._hyper = None              # (betas, eps, weight_decay) baked into the captured optimizer launch
        self._one = torch.ones((), device=x.device)
        self._g_fb = self._g_tail = None
        self._split = False
        self._comm_stream = None
        self.time_collective = False   # bench.py: record HIP events around the exchange of the last steps
        self._coll_events = []
        self.collective_in_graph = False  # the RCCL exchange + Adam were captured with the step (one replay per step)
        self.capture_error = None
        self.capture_verified = None      # {"grad_rel_inf", "param_rel_inf", "ok"} of the replay-vs-eager check, or None
        self._force_verify_mismatch = False  # tests: exercise the opt-out path
        self._force_capture_failure = False  # tests: this rank's capture of the collective "fails" (the ranks must agree on it)
        self.exchange_enabled = True    # bench.py: False = skip the collectives (timing of the exposed share; ranks diverge)
        self._head_sig = None           # batch shape / target row size baked into a capture whose head launch labels the batch
        self._recapture = False         # the label form (in the head launch / eager in front of the step) changed since the capture

    # -- the step body, split at the collective ---------------------------------------------------
    def _fused_head(self):
        """Head + loss fusable: a bare nn.Linear head and one of glass_amd.losses' marker losses."""
        import torch.nn as nn
        from . import losses
        head = self.model.preds[0]
        return losses.fusable_mode(self.loss_fn) is not None and type(head) is nn.Linear and head.bias is not None

    def _program_step(self):
        from . import stack
        return (self._fused_head() and hasattr(self.bucket, "flat_param") and self.x.dim() == 3 and
                self.x.shape[1:] == (1, 1) and stack.step_supported(self.model, self.loss_fn))

    def _fwd_bwd(self, tail_hook=None, apply_opt=False):
        """Forward + backward of one batch.  apply_opt: the optimizer step may ride in the backward's last launch (step
        program, optim.FlatAdam, one rank); returns True when it did."""
        from . import stack
        if self._program_step():
            # whole step as one explicit program: no autograd tape, fused readout, labels straight from pos, and
            # — when the program writes every gradient of the arena — no zero-fill (glass_amd/stack.py)
            overwrite = stack.covers_arena(self.model, self.bucket)
            if not overwrite:
                self.bucket.zero()
            fused = self.opt if (apply_opt and overwrite and hasattr(self.opt, "fused_args")) else None
            prog = stack._program(self.model.conv)
            prog.loss_sum = self._loss_sum  # the readout adds this step's loss to the running sum in its own launch
            try:
                loss, _logits = stack.loss_and_grads(self.model, self.loss_fn, self.x, self.ei, self.ew, self._pos, "pos",
                                                     self._y, overwrite, tail_hook, labels=self._labels, fused_opt=fused)
            finally:
                prog.loss_sum = None
            self._loss = loss
            return stack.applied_optimizer(self.model)
        if tail_hook is not None:
            raise RuntimeError("tail_hook needs the step program")
        z = utils.MaxZOZ(self.x, self._pos)
        self.bucket.zero()
        if self._fused_head():
            from . import losses
            emb = self.model.NodeEmb(self.x, self.ei, self.ew, z)
            pooled = self.model.Pool(emb, self._pos, self.model.pools[0])
            loss, _logits = losses.head_loss(pooled, self.model.preds[0], self._y, losses.fusable_mode(self.loss_fn),
                                             direct=hasattr(self.bucket, "flat_param"))
        else:
            pred = self.model(self.x, self.ei, self.ew, self._pos, z, id=0)
            loss = self.loss_fn(pred, self._y)
        loss.backward(gradient=self._one)  # persistent seed gradient: no ones_like fill per step
        # Under capture `loss` lives at a fixed address of the graph's private pool, so keeping the
        # reference replaces a copy kernel; in eager mode it is simply the latest loss tensor.
        self._loss = loss.detach()
        self._loss_sum.add_(self._loss)

    def _warmup(self):
        """Real training steps on the first batch, on a side stream: builds the CSR / plans /
        workspaces / BLAS handles outside any capture.  Runs in eager mode too, so both modes follow
        the same trajectory."""
        snap = None
        if self.preserve_state and self.warmup_iters > 0:
            snap = self._snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup_iters):
                solo = not gdist.is_distributed()
                done = self._fwd_bwd(apply_opt=solo)
                self.bucket.all_reduce_mean()
                if not done:
                    self.opt.step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        if snap is not None:
            self._restore(snap)

    def _snapshot(self):
        """Everything a training step changes: parameters, optimizer state (incl. the fused Adam's device-side moments and
        step counters), the dropout stream."""
        import copy
        from . import ops
        return (copy.deepcopy(self.model.state_dict()), copy.deepcopy(self.opt.state_dict()),
                {k: getattr(self.opt, k).clone() for k in ("exp_avg", "exp_avg_sq", "step_dev", "step_dev_shard") if hasattr(self.opt, k)},
                ops.rng_state(self.x.device).clone(), self._loss_sum.clone())

    def _restore(self, snap):
        from . import ops
        self.model.load_state_dict(snap[0])          # in-place copies: arena aliasing is kept
        self.opt.load_state_dict(snap[1])
        for k, v in snap[2].items():
            getattr(self.opt, k).copy_(v)
        ops.rng_state(self.x.device).copy_(snap[3])
        self._loss_sum.copy_(snap[4])
        torch.cuda.synchronize()

    def _verify_collective_capture(self):
        """One replay of the freshly captured [forward + backward + all-reduce + Adam] graph against the same step run
        eagerly from the same state (same batch, same dropout words): the exchanged gradient arena and the updated
        parameters must agree on EVERY rank (rel-inf <= 1e-5: a captured collective may pick another channel count than
        the eager one, so bitwise equality is not demanded).  The ranks agree on the verdict through an eager all-reduce
        (MAX) — a rank must never decide alone, or the ranks would issue different collectives from then on.  State is
        restored afterwards: the trajectory is that of a run without this check.  Returns True when the capture is trusted."""
        import torch.distributed as td
        snap = self._snapshot()
        head = self._labels is not None and self._labels.in_head
        if head:
            self._labels.rewind()   # (labels in the head launch: the eager step and the replay must take the SAME batch of the cursor)
        self._fwd_bwd()
        self.bucket.all_reduce_mean()
        self.opt.step()
        torch.cuda.synchronize()
        g_eager, p_eager = self.bucket.flat.clone(), self.bucket.flat_param.clone()
        self._restore(snap)
        if head:
            self._labels.rewind()
        self._g_fb.replay()
        torch.cuda.synchronize()
        g_graph, p_graph = self.bucket.flat.clone(), self.bucket.flat_param.clone()
        self._restore(snap)

        def rel(a, b):
            d = float(b.abs().max())
            e = float((a - b).abs().max())
            return e / (d if d > 0 else 1.0) if (e == e and d == d) else float("inf")
        e_g, e_p = rel(g_graph, g_eager), rel(p_graph, p_eager)
        bad = 1.0 if (self._force_verify_mismatch or not (e_g <= _VERIFY_TOL and e_p <= _VERIFY_TOL)) else 0.0
        flag = torch.tensor([bad], device=self.x.device)
        td.all_reduce(flag, op=td.ReduceOp.MAX)
        ok = float(flag.item()) == 0.0
        self.capture_verified = {"grad_rel_inf": e_g, "param_rel_inf": e_p, "ok": ok, "tol": _VERIFY_TOL}
        return ok

    def _overlap_small_bucket(self):
        """The data-parallel step with an embedding-sized gradient bucket, on the step program: worth cutting the
        backward where the small bucket is final."""
        return (gdist.is_distributed() and hasattr(self.bucket, "sharded") and self.bucket.sharded() and
                self._program_step())

    def _capture(self):
        dist_on = gdist.is_distributed()
        self._g_fb = torch.cuda.CUDAGraph()
        # With a process group alive, its watchdog THREAD polls the completion events of earlier collectives with
        # hipEventQuery.  Under the default capture mode ("global") a call like that from ANY thread while this thread
        # captures is an error — the watchdog then aborts the process (seen once in ~6 runs of the 1-rank RCCL test).
        # "thread_local" confines the check to the capturing thread; nothing this thread enqueues is unsafe to capture.
        mode = "thread_local" if dist_on else "global"
        if dist_on:
            torch.cuda.synchronize()  # no collective of the warm-up still in flight when the capture starts
        if not dist_on:
            with torch.cuda.graph(self._g_fb):
                if not self._fwd_bwd(apply_opt=True):  # (with the step program Adam rides in the backward's last launch)
                    self.opt.step()
        elif not self._overlap_small_bucket():
            # First choice: the WHOLE data-parallel step as one replay — forward/backward, the RCCL all-reduce of the
            # gradient arena and Adam captured together (RCCL collectives are capturable; gloo's are host code).  A refused
            # capture raises inside the context manager, the partial capture is discarded and the split form below is taken:
            # the process stays usable either way.
            import torch.distributed as td
            if CAPTURE_COLLECTIVE and td.get_backend() == "nccl":
                try:
                    if self._force_capture_failure:
                        raise RuntimeError("forced capture failure (test hook)")
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode=mode):
                        self._fwd_bwd()
                        self.bucket.all_reduce_mean()
                        self.opt.step()
                    self._g_fb, self.collective_in_graph = g, True
                except Exception as e:  # noqa: BLE001 — whatever the runtime refuses: fall back
                    self.capture_error = repr(e)
                    torch.cuda.synchronize()
                    self._g_fb = torch.cuda.CUDAGraph()
                # The ranks agree on the capture's outcome BEFORE anything else is issued: a rank whose capture failed alone
                # (graph-pool OOM, a transient refusal) would otherwise start its per-step eager all-reduces while the others
                # enter the verification below — whose collectives would then pair with that rank's training all-reduces.
                # One eager all-reduce (MIN) of "captured ok", issued by EVERY rank whatever happened above.
                ok = torch.tensor([1.0 if self.collective_in_graph else 0.0], device=self.x.device)
                td.all_reduce(ok, op=td.ReduceOp.MIN)
                if float(ok.item()) == 0.0 and self.collective_in_graph:
                    self.capture_error = "another rank could not capture the collective: split form on every rank"
                    self.collective_in_graph = False
                    self._g_fb = torch.cuda.CUDAGraph()
                if self.collective_in_graph and VERIFY_CAPTURED_COLLECTIVE and hasattr(self.bucket, "flat_param"):
                    if not self._verify_collective_capture():
                        # automatic opt-out: same process, split form from here on (never a re-exec of a GPU-initialised process)
                        self.capture_error = (f"captured collective failed the replay-vs-eager check: {self.capture_verified}")
                        self.collective_in_graph = False
                        self._g_fb = torch.cuda.CUDAGraph()
            if not self.collective_in_graph:
                # the collective stays outside the graph; the optimizer is two launches, cheaper eager than a
                # second graph replay
                with torch.cuda.graph(self._g_fb, capture_error_mode=mode):
                    self._fwd_bwd()
        else:
            # two graphs sharing one memory pool, cut by the program's tail hook: [forward + backward down to the last
            # layer / head gradient] | [emb_gn + embedding gradient]
            self._g_tail = torch.cuda.CUDAGraph()
            self._comm_stream = torch.cuda.Stream()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._g_fb.capture_begin(capture_error_mode=mode)

                def cut():
                    self._g_fb.capture_end()
                    self._g_tail.capture_begin(pool=self._g_fb.pool(), capture_error_mode=mode)
                self._fwd_bwd(tail_hook=cut)
                self._g_tail.capture_end()
            torch.cuda.current_stream().wait_stream(side)
        self._split = dist_on and not self.collective_in_graph
        self.graphed = True

    def _exchange_and_update(self):
        """Eager part of the data-parallel step, after the (first) graph: collectives + Adam."""
        ev = None
        if self.time_collective:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
        if not self.exchange_enabled:
            if self._g_tail is not None:
                self._g_tail.replay()
        elif self._g_tail is not None:
            main, comm = torch.cuda.current_stream(), self._comm_stream
            ex = self.bucket.exchange
            comm.wait_stream(main)             # the small bucket is final here
            with torch.cuda.stream(comm):
                ex.reduce_small()              # beside the tail of the backward pass
            self._g_tail.replay()
            ex.reduce_big()                    # (same process group: runs after the small all-reduce inside RCCL)
            main.wait_stream(comm)
        else:
            self.bucket.all_reduce_mean()
        if ev:
            ev[1].record()
        self.opt.step()
        if ev:
            ev[2].record()
            self._coll_events = (self._coll_events + [ev])[-64:]

    def collective_share(self):
        """What the data-parallel exchange of this step looks like, for the bench line of every N > 1 run (so that "did
        RCCL see N ranks, and in which form" is answerable from the JSON): world size and backend as torch.distributed
        reports them, the RCCL version, whether the exchange was captured inside the step's graph (and what the capture
        attempt or its replay-vs-eager check said), the payload per step — and, in the split form, the mean device time of
        the exchange and of the optimizer part (Adam launches + parameter all-gather) over the recorded steps.  None on one
        rank without a process group."""
        import torch.distributed as td
        if not (td.is_available() and td.is_initialized()):
            return None
        backend = td.get_backend()
        out = {"world_size": td.get_world_size(), "backend": backend, "rccl_version": None,
               "in_graph": self.collective_in_graph, "collective_in_graph": self.collective_in_graph,
               "capture_error": self.capture_error, "capture_verified": self.capture_verified,
               "small_bucket_overlaps_backward_tail": self._g_tail is not None,
               "exchange_us": None, "adam_and_gather_us": None, "steps_timed": 0}
        if backend == "nccl":
            try:
                out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception as e:  # noqa: BLE001 — a version string must never fail a bench line
                out["rccl_version"] = f"unavailable ({type(e).__name__})"
        ex = getattr(self.bucket, "exchange", None)
        if ex is not None:
            out["payload_bytes"] = ex.payload_bytes()
            out["world"] = ex.world
        else:
            out["payload_bytes"] = {"small_allreduce": self.bucket.flat.numel() * self.bucket.flat.element_size(),
                                    "big_reduce_scatter": 0, "big_all_gather": 0}
            out["world"] = td.get_world_size()
        # a number to hold the measurement against (dist.predict_collective_us: stated model, assumed constants)
        out["predicted"] = gdist.predict_collective_us(out["payload_bytes"], out["world"])
        out["predicted_us"] = out["predicted"]["total_us"]
        if self._coll_events:
            torch.cuda.synchronize()
            red = [a.elapsed_time(b) * 1e3 for a, b, _ in self._coll_events]
            upd = [b.elapsed_time(c) * 1e3 for _, b, c in self._coll_events]
            out.update(exchange_us=sum(red) / len(red), adam_and_gather_us=sum(upd) / len(upd), steps_timed=len(red))
        return out

    def reset_loss_sum(self):
        self._loss_sum.zero_()

    def loss_sum(self):
        """Device scalar: sum of the losses of the steps since reset_loss_sum() (added in step order, fp32)."""
        return self._loss_sum

    def _alloc_batch(self, shape, pos, y):
        """The step's fixed batch buffers (first use) — and, on the step program, the label state of the batch."""
        self._pos = torch.full(shape, -1, dtype=pos.dtype, device=pos.device)
        self._y = torch.zeros((shape[0], ) + tuple(y.shape[1:]), dtype=y.dtype, device=y.device)
        if self._program_step() and pos.is_cuda and pos.dtype == torch.int64:
            from . import stack
            self._labels = stack.BatchLabels(self.x.shape[0], self._pos.numel(), pos.device)

    def begin_epoch(self, pos_all, y_all, idx_batches, wrap=False):
        """The epoch's batches up front: pos_all / y_all the DATA SET's node and target matrices, idx_batches int64
        [n_steps, B] the rows of every batch in step order (ZGDataloader's permutation cut into batches; a data-parallel
        rank passes its own slices).  When the step runs as a captured program, its HEAD launch then labels the batch
        itself — prologue || labels as one launch (glass_step_head_f32), the batch named by a device-resident cursor that
        the launch advances — and next_step() is one graph replay with no launch in front of it.  Returns False when this
        step cannot take that form (no step program, eager mode, unsuitable tensors): the caller keeps calling
        step(pos_all, y_all, index) per batch."""
        if not (self.use_graph and pos_all.is_cuda and pos_all.dim() == 2 and pos_all.dtype == torch.int64 and
                pos_all.is_contiguous() and y_all.is_cuda and y_all.is_contiguous() and y_all.shape[0] == pos_all.shape[0] and
                idx_batches.is_cuda and idx_batches.dim() == 2 and idx_batches.dtype == torch.int64 and
                idx_batches.is_contiguous() and idx_batches.shape[0] > 0 and
                (y_all.element_size() * (y_all.numel() // max(y_all.shape[0], 1))) % 4 == 0):
            return False
        shape = (idx_batches.shape[1], ) + tuple(pos_all.shape[1:])
        first = self._pos is None
        if first:
            if not self._program_step():
                return False      # (nothing allocated: the caller's step(pos, y, index) does its own first-use work)
            self._alloc_batch(shape, pos_all, y_all)
        if self._labels is None or shape != tuple(self._pos.shape) or y_all.dtype != self._y.dtype:
            return False
        was_head = self._labels.in_head
        self._labels.set_epoch(pos_all, y_all, idx_batches, self._pos, self._y, wrap=wrap)
        self._labels.in_head = True
        sig = self._labels.head_signature()
        hyper = self.opt.hyper() if hasattr(self.opt, "hyper") else None
        if first or not self.graphed or hyper != self._hyper or sig != self._head_sig or not was_head or self._recapture:
            if first:
                self._warmup()   # (eager steps through the same head launch: they consume the first batches of the cursor)
            self._hyper, self._head_sig, self._recapture = hyper, sig, False
            self._capture()
            self._labels.set_epoch(pos_all, y_all, idx_batches, self._pos, self._y, wrap=wrap)  # cursor back to batch 0
        return True

    def next_step(self):
        """One step on the cursor's batch (after begin_epoch returned True): a graph replay."""
        if hasattr(self.opt, "sync_lr"):
            self.opt.sync_lr()
        self._g_fb.replay()
        if self._split:
            self._exchange_and_update()
        return self._loss

    def __call__(self, pos, y, index=None):
        """One step on the batch (pos, y).  index (int64 device vector): pos / y are the DATA SET's whole node and target
        matrices and the batch is their rows `index` — the selection ZGDataloader does with `pos[perm], y[perm]`
        (impl/SubGDataset.py:69-72) happens inside the step's label launch instead of two index kernels before it."""
        shape = tuple(pos.shape) if index is None else (index.numel(), ) + tuple(pos.shape[1:])
        first = self._pos is None
        if first:
            self._alloc_batch(shape, pos, y)
        if shape != tuple(self._pos.shape):
            raise ValueError("TrainStep needs a fixed batch shape (drop_last=True): "
                             f"{shape} vs {tuple(self._pos.shape)}")
        if self._labels is not None and self._labels.in_head:
            # this call labels its batch eagerly: a graph captured with the labels in its head launch has to go
            self._labels.in_head = False
            self._recapture = self.graphed
        self._load_batch(pos, y, index)
        hyper = self.opt.hyper() if hasattr(self.opt, "hyper") else None
        if first or (self.graphed and (hyper != self._hyper or self._recapture)):
            self._recapture = False
            # (betas / eps / weight_decay are launch arguments: a changed param_groups entry means a new capture; the learning
            # rate lives in device memory and needs none)
            if first:
                self._warmup()
            self._hyper = hyper
            if self.use_graph:
                self._capture()
        if hasattr(self.opt, "sync_lr"):
            self.opt.sync_lr()  # a scheduler may have changed the learning rate since the capture
        if self.graphed:
            self._g_fb.replay()
            if self._split:
                self._exchange_and_update()
        elif gdist.is_distributed():
            self._fwd_bwd()
            self._exchange_and_update()
        elif not self._fwd_bwd(apply_opt=True):
            self.opt.step()
        return self._loss

    def _load_batch(self, pos, y, index=None):
        """The batch into the step's fixed buffers: one launch for both tensors when they are plain device tensors — with
        the step program the same launch maintains the label bytes and lists the unique labeled rows (glass_batch_labels),
        and selects the batch's rows itself when `index` is given (glass_batch_labels_gather)."""
        if index is not None:
            index = index.contiguous()  # (a data-parallel rank's slice perm[rank::world] is strided)
            if (self._labels is not None and pos.is_contiguous() and y.is_cuda and y.is_contiguous() and index.is_cuda and
                    index.dtype == torch.int64 and index.is_contiguous() and y.dtype == self._y.dtype and pos.dim() == 2 and
                    y.shape[0] == pos.shape[0] and (y.element_size() * (y.numel() // max(y.shape[0], 1))) % 4 == 0):
                self._labels.load_gather(pos, y, index, self._pos, self._y)
                return
            pos, y = pos[index], y[index]
        if self._labels is not None:
            pos_c = pos if pos.is_contiguous() else pos.contiguous()
            if (y.is_cuda and y.is_contiguous() and y.dtype == self._y.dtype and y.shape == self._y.shape and y.numel() and
                    (y.element_size() * y.numel()) % 4 == 0):
                self._labels.load(pos_c, self._pos, y, self._y)
            else:
                self._labels.load(pos_c, self._pos)
                self._y.copy_(y)
            return
        if (pos.is_cuda and y.is_cuda and pos.is_contiguous() and y.is_contiguous() and pos.dtype == self._pos.dtype and
                y.dtype == self._y.dtype and y.shape == self._y.shape and pos.numel() and y.numel() and
                (pos.element_size() * pos.numel()) % 4 == 0 and (y.element_size() * y.numel()) % 4 == 0):
            from . import _lib
            rc = _lib.load().glass_copy_pair(self._pos.data_ptr(), pos.data_ptr(), pos.numel() * pos.element_size(),
                                             self._y.data_ptr(), y.data_ptr(), y.numel() * y.element_size(),
                                             torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "glass_copy_pair")
            return
        self._pos.copy_(pos)
        self._y.copy_(y)

    def last_loss(self):
        return float(self._loss.item())
