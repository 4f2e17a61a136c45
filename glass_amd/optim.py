"""Adam over a ParamArena in one kernel launch (glass_adam_step_f32), with torch.optim.Adam's
update rule (GLASSTest.py:213 uses Adam(lr) with defaults).

Two front ends over the same device state (moments, step counters, learning rate in device memory so a captured step
follows a scheduler):

  FlatAdam      an Optimizer of its own — enough of the interface for `lr_scheduler.ReduceLROnPlateau`
                (GLASSTest.py:214-216, 225): the learning rate lives in `param_groups[0]['lr']`.
  AdoptedAdam   the engine UNDER a plain `torch.optim.Adam(model.parameters(), lr)` — what the reference driver builds
                (GLASSTest.py:213).  `adopt()` takes such an optimizer over: its `param_groups` stay the one place
                hyper-parameters are read from (a scheduler holding the torch optimizer keeps working), its per-parameter
                state (`exp_avg`, `exp_avg_sq`) becomes views into the flat moment buffers, `step` is republished once per
                epoch — `optimizer.state_dict()` / `load_state_dict()` / a later eager `optimizer.step()` all see and
                continue the same state.  impl.train.train does this for the caller; nothing repo-specific is needed.
"""
import torch

from . import _lib


class _FlatAdamCore:
    """Device state + launches.  Subclasses provide `param_groups` (a list whose first dict holds lr / betas / eps /
    weight_decay) and `arena`."""
    def _init_device_state(self, arena, lr):
        self.arena = arena
        arena.shard_optimizer = True  # this optimizer updates an embedding-sized bucket shard-wise (step() below)
        dev = arena.flat.device
        self.exp_avg = torch.zeros_like(arena.flat)
        self.exp_avg_sq = torch.zeros_like(arena.flat)
        self.step_dev = torch.zeros(2, dtype=torch.int64, device=dev)  # (steps completed, kernel ticket)
        self.step_dev_shard = torch.zeros(2, dtype=torch.int64, device=dev)  # the same for the sharded big bucket
        self.lr_dev = torch.full((1, ), float(lr), dtype=torch.float32, device=dev)
        self._lr_host = float(lr)

    def sync_lr(self):
        """Mirror a scheduler's change of param_groups[0]['lr'] into device memory (call outside a
        graph replay; cheap host compare otherwise)."""
        lr = float(self.param_groups[0]["lr"])
        if lr != self._lr_host:
            self.lr_dev.fill_(lr)
            self._lr_host = lr

    def hyper(self):
        """(beta1, beta2, eps, weight_decay): the values a captured step has baked in (TrainStep re-captures when they change)."""
        g = self.param_groups[0]
        return (float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]))

    def fusable(self):
        """The update can ride in the step program's last launch (glass_embed_norm_bwd_adam_f32): one unsharded arena."""
        return not self.arena.sharded() and self.arena.attached() and getattr(self.arena, "_peer", None) is None

    def fused_args(self):
        """(param, grad, exp_avg, exp_avg_sq, n, lr_dev, beta1, beta2, eps, weight_decay, step_dev) of the whole arena."""
        a = self.arena
        return (a.flat_param.data_ptr(), a.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                a.flat_param.numel(), self.lr_dev.data_ptr(), *self.hyper(), self.step_dev.data_ptr())

    def zero_grad(self, set_to_none=False):
        self.arena.zero()

    @torch.no_grad()
    def step(self, closure=None):
        b1, b2, eps, wd = self.hyper()
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        a = self.arena
        peer = getattr(a, "_peer", None)
        if peer is not None:  # one-shot peer exchange + Adam as ONE launch (glass_amd/peer.py); all_reduce_mean() was a no-op
            peer.step(self)
            return

        def launch(param, grad, m, v, counter):
            if param.numel() == 0:
                return
            rc = _lib.load().glass_adam_step_f32(param.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), param.numel(),
                                                 self.lr_dev.data_ptr(), b1, b2, eps, wd, counter.data_ptr(),
                                                 torch.cuda.current_stream().cuda_stream)
            _lib.check(rc, "glass_adam_step_f32")

        if not a.sharded():
            launch(a.flat_param, a.flat, self.exp_avg, self.exp_avg_sq, self.step_dev)
            return
        # data-parallel run with an embedding-sized bucket (dist.GradExchange): the small bucket holds the all-reduced
        # mean gradient; of the big bucket this rank owns one shard (mean gradient in ex.shard_grad) — update both,
        # then every rank receives every updated parameter shard.  Each region counts its own steps.
        ex = a.exchange
        s = a.big_start
        launch(a.flat_param[:s], a.flat[:s], self.exp_avg[:s], self.exp_avg_sq[:s], self.step_dev)
        p_sh, m_sh, v_sh = ex.shard_views(a.flat_param, self.exp_avg, self.exp_avg_sq)
        launch(p_sh, ex.shard_grad, m_sh, v_sh, self.step_dev_shard)
        ex.gather_params()


class FlatAdam(_FlatAdamCore, torch.optim.Optimizer):
    def __init__(self, arena, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        torch.optim.Optimizer.__init__(self, arena.params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._init_device_state(arena, lr)


class AdoptedAdam(_FlatAdamCore):
    """The flat Adam engine under a caller's torch.optim.Adam (see the module docstring and `adopt`)."""
    def __init__(self, torch_opt, arena):
        self.torch_opt = torch_opt
        self._init_device_state(arena, torch_opt.param_groups[0]["lr"])
        self._published = None  # step count last written into torch_opt.state
        self._views = {}        # id(param) -> (exp_avg view, exp_avg_sq view)
        for p in arena.params:
            o = arena.offset_of(p)
            self._views[id(p)] = (self.exp_avg[o:o + p.numel()].view_as(p), self.exp_avg_sq[o:o + p.numel()].view_as(p))
        self.import_state()

    @property
    def param_groups(self):
        return self.torch_opt.param_groups  # the caller's optimizer is the one place hyper-parameters live

    # TrainStep snapshots optimizer.state_dict() around its warm-up: everything of this engine that a step changes is device
    # state the snapshot copies by name (exp_avg, exp_avg_sq, step_dev, step_dev_shard); nothing else to save
    def state_dict(self):
        return {}

    def load_state_dict(self, sd):
        return None

    def _state_is_ours(self):
        st = self.torch_opt.state
        for p in self.arena.params:
            s = st.get(p)
            if not s or "exp_avg" not in s:
                return False
            m, v = self._views[id(p)]
            if s["exp_avg"].data_ptr() != m.data_ptr() or s["exp_avg_sq"].data_ptr() != v.data_ptr():
                return False
        return True

    @torch.no_grad()
    def import_state(self):
        """torch_opt.state -> the flat buffers (state from earlier eager steps or from load_state_dict), then the state
        entries are replaced by views of the flat buffers.  Cheap no-op when they already are."""
        st = self.torch_opt.state
        if self._state_is_ours():
            # an eager optimizer.step() in between advanced the per-parameter step tensors (the moments it updated ARE ours)
            n = int(float(st[self.arena.params[0]]["step"]))
            if n != self._published:
                self.step_dev[0] = n
                self.step_dev_shard[0] = n
                self._published = n
            return
        steps = set()
        for p in self.arena.params:
            m, v = self._views[id(p)]
            s = st.get(p)
            if s and "exp_avg" in s:
                m.copy_(s["exp_avg"])
                v.copy_(s["exp_avg_sq"])
                steps.add(int(float(s["step"])))
            else:
                m.zero_()
                v.zero_()
                steps.add(0)
        if len(steps) != 1:
            raise ValueError(f"torch.optim.Adam state with different step counts per parameter ({sorted(steps)}): cannot be "
                             "run as one flat update")
        n = steps.pop()
        self.step_dev.zero_()
        self.step_dev_shard.zero_()
        self.step_dev[0] = n
        self.step_dev_shard[0] = n
        for p in self.arena.params:
            m, v = self._views[id(p)]
            st[p] = {"step": torch.tensor(float(n), dtype=torch.float32), "exp_avg": m, "exp_avg_sq": v}
        self._published = n

    def publish(self):
        """The device step counter -> torch_opt.state[p]['step'] (one device read: call where the loop syncs anyway)."""
        n = int(self.step_dev[0].item())
        if n != self._published:
            for p in self.arena.params:
                self.torch_opt.state[p]["step"].fill_(float(n))
            self._published = n
            self.torch_opt._opt_called = True  # (what lr_scheduler's step-order check looks at: steps did happen)


def adoptable(optimizer, model):
    """None when `optimizer` is a plain torch.optim.Adam this engine reproduces exactly, else the reason (a string)."""
    if type(optimizer) is not torch.optim.Adam:
        return f"optimizer is {type(optimizer).__name__}, not torch.optim.Adam"
    if len(optimizer.param_groups) != 1:
        return "more than one parameter group"
    g = optimizer.param_groups[0]
    # fused=True: its eager step() wants state['step'] on the parameter's device; import_state installs a CPU scalar
    for key in ("amsgrad", "maximize", "capturable", "differentiable", "fused"):
        if g.get(key, False):
            return f"{key}=True"
    if isinstance(g["lr"], torch.Tensor):
        return "tensor learning rate"
    if g.get("decoupled_weight_decay", False):
        return "decoupled weight decay"
    mine = [p for p in model.parameters() if p.requires_grad]
    if {id(p) for p in g["params"]} != {id(p) for p in mine}:
        return "the optimizer's parameters are not exactly the model's trainable parameters"
    if any(p.dtype != torch.float32 or not p.is_cuda for p in mine):
        return "parameters must be fp32 on the GPU"
    return None


def adopt(optimizer, model):
    """The AdoptedAdam engine for (optimizer, model), cached on the model — built on first use: the model's parameters move
    into a ParamArena (Parameter objects keep their identity, so the optimizer's references stay valid) unless it has one.
    Raises ValueError when the optimizer is not adoptable (ask `adoptable` first)."""
    why = adoptable(optimizer, model)
    if why is not None:
        raise ValueError(why)
    from .arena import ParamArena
    arena = model.__dict__.get("_glass_grad_bucket")
    if not isinstance(arena, ParamArena) or {id(p) for p in arena.params} != {id(p) for p in optimizer.param_groups[0]["params"]}:
        arena = ParamArena(model)
    elif not arena.attached():
        arena.reattach()
    from .utils import RuntimeCache
    slot = model.__dict__.setdefault("_glass_adopted_adam", RuntimeCache())  # (deepcopy / pickle of the model: an empty slot)
    eng = slot.get("engine")
    if eng is None or eng.torch_opt is not optimizer or eng.arena is not arena:
        eng = slot["engine"] = AdoptedAdam(optimizer, arena)
    else:
        eng.import_state()
    return eng
