"""Adam over a ParamArena in one kernel launch (glass_adam_step_f32), with torch.optim.Adam's
update rule (GLASSTest.py:213 uses Adam(lr) with defaults) and enough of the Optimizer interface
for `lr_scheduler.ReduceLROnPlateau` (GLASSTest.py:214-216, 225): the learning rate lives in
`param_groups[0]['lr']`; it is mirrored into device memory so a captured step follows it."""
import torch

from . import _lib


class FlatAdam(torch.optim.Optimizer):
    def __init__(self, arena, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.arena = arena
        arena.shard_optimizer = True  # this optimizer updates an embedding-sized bucket shard-wise (step() below)
        super().__init__(arena.params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
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

    def fusable(self):
        """The update can ride in the step program's last launch (glass_embed_norm_bwd_adam_f32): one unsharded arena."""
        return not self.arena.sharded() and self.arena.attached()

    def fused_args(self):
        """(param, grad, exp_avg, exp_avg_sq, n, lr_dev, beta1, beta2, eps, weight_decay, step_dev) of the whole arena."""
        g, a = self.param_groups[0], self.arena
        return (a.flat_param.data_ptr(), a.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                a.flat_param.numel(), self.lr_dev.data_ptr(), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
                float(g["weight_decay"]), self.step_dev.data_ptr())

    def zero_grad(self, set_to_none=False):
        self.arena.zero()

    @torch.no_grad()
    def step(self, closure=None):
        from .ops import join_side_streams
        join_side_streams()  # weight gradients accumulate on a side stream
        g = self.param_groups[0]
        if not torch.cuda.is_current_stream_capturing():
            self.sync_lr()
        a = self.arena

        def launch(param, grad, m, v, counter):
            if param.numel() == 0:
                return
            rc = _lib.load().glass_adam_step_f32(param.data_ptr(), grad.data_ptr(), m.data_ptr(), v.data_ptr(), param.numel(),
                                                 self.lr_dev.data_ptr(), g["betas"][0], g["betas"][1], g["eps"],
                                                 g["weight_decay"], counter.data_ptr(),
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
