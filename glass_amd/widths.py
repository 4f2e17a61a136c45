"""Hidden widths outside the hand-written kernel families, served by EXACT zero padding.

The fused kernels exist for hidden <= 32 (any width), 64, 128, 256 and 512 (`glass_dense_caps(H).family`); every
`config/*.yml` of the reference and every BASELINE config is inside.  Any other width H <= 512 is run at the next family
width Hp: the model is built at Hp with every parameter zero outside its logical block.  That is exact, not approximate —
a padded column is 0 after the embedding lookup and stays 0 through both Linear layers of a GLASSConv (zero weight rows /
columns, zero bias), the label mix, the neighbour aggregation, GraphNorm (x - alpha * mean = 0 and beta = 0: the output is 0
whatever rstd is), ELU / ReLU, dropout, every pooling and the head; the padded parameters receive exactly zero gradients
(their inputs or their output gradients are zero), so Adam (no weight decay) leaves them at zero.  The logical blocks take
the values an unpadded construction draws (same generator consumption), `state_dict()` / `load_state_dict()` speak the
logical shapes (reference checkpoints load unchanged), and the reference's own classes (`impl.models`) are untouched: the
padding lives where the drivers build the model (glass_amd.factory.build_glass, GLASSTest.build_model).

Where a width dimension sits in a parameter is read off the shapes: a dimension of the logical tensor that differs from the
physical one is nb = size / H blocks of H (1: a hidden vector; 2: the comb pair's input [g || x_]; num_layers: the JK
concatenation in the last GraphNorm and the head), and block b maps to physical columns b * Hp .. b * Hp + H - 1.
"""
import torch


def fused_width(hidden):
    """The width the fused step program runs `hidden` at (hidden itself when a kernel family serves it or none can): the
    library's own answer, glass_dense_caps(hidden).serve_width."""
    hidden = int(hidden)
    from . import _lib
    w = int(_lib.dense_caps(hidden).serve_width)
    return w if w > 0 else hidden


def _dim_index(log_size, phys_size, H, Hp, device):
    if log_size == phys_size:
        return torch.arange(log_size, device=device)
    nb = log_size // H
    if nb * H != log_size or nb * Hp != phys_size:
        raise ValueError(f"cannot map a dimension of {log_size} onto {phys_size} (logical width {H}, physical {Hp})")
    i = torch.arange(log_size, device=device)
    return (i // H) * Hp + i % H


def _index(log_shape, phys_shape, H, Hp, device):
    if len(log_shape) != len(phys_shape):
        raise ValueError(f"rank mismatch {tuple(log_shape)} vs {tuple(phys_shape)}")
    idx = [_dim_index(a, b, H, Hp, device) for a, b in zip(log_shape, phys_shape)]
    nd = len(idx)
    return tuple(ix.reshape([-1 if d == k else 1 for d in range(nd)]) for k, ix in enumerate(idx))


def pad_tensor(t, phys_shape, H, Hp):
    """The logical tensor t placed into zeros of phys_shape."""
    out = torch.zeros(tuple(phys_shape), dtype=t.dtype, device=t.device)
    if t.dim() == 0:
        return t.clone()
    out[_index(t.shape, phys_shape, H, Hp, t.device)] = t
    return out


def unpad_tensor(t, log_shape, H, Hp):
    """The logical part of the physical tensor t (a new tensor)."""
    if t.dim() == 0:
        return t.clone()
    return t[_index(log_shape, t.shape, H, Hp, t.device)].clone()


def _install_state_dict_hooks(model, H, Hp, log_shapes):
    def save_hook(module, state_dict, prefix, local_metadata):
        for name, shape in log_shapes.items():
            key = prefix + name
            if key in state_dict and tuple(state_dict[key].shape) != tuple(shape):
                state_dict[key] = unpad_tensor(state_dict[key], shape, H, Hp)
        return state_dict

    def load_hook(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        phys = {n: tuple(p.shape) for n, p in model.named_parameters()}
        for name, shape in log_shapes.items():
            key = prefix + name
            if key in state_dict and tuple(state_dict[key].shape) == tuple(shape) and phys.get(name, shape) != tuple(shape):
                state_dict[key] = pad_tensor(state_dict[key], phys[name], H, Hp)

    model._register_state_dict_hook(save_hook)
    model._register_load_state_dict_pre_hook(load_hook)


def build_at_fused_width(hidden, build):
    """build(h) constructs the model at hidden width h.  Returns build(hidden) when a kernel family serves `hidden`; otherwise
    the model built at fused_width(hidden) whose logical blocks hold the parameters build(hidden) draws and whose padding is
    zero (module docstring).  The global CPU generator ends where the unpadded construction leaves it."""
    H, Hp = int(hidden), fused_width(hidden)
    if Hp == H:
        return build(H)
    logical = build(H)
    state = torch.get_rng_state()
    physical = build(Hp)
    torch.set_rng_state(state)
    log_params = dict(logical.named_parameters())
    with torch.no_grad():
        for name, p in physical.named_parameters():
            p.copy_(pad_tensor(log_params[name].detach(), p.shape, H, Hp))
    physical._glass_logical_width = (H, Hp)
    _install_state_dict_hooks(physical, H, Hp, {n: tuple(p.shape) for n, p in log_params.items()})
    return physical


def logical_named_grads(model):
    """{name: gradient in the logical shape} of a model built by build_at_fused_width (plain gradients otherwise), and the
    largest |gradient| that fell on padding (exactly 0 by construction)."""
    lw = getattr(model, "_glass_logical_width", None)
    out, pad_max = {}, 0.0
    if lw is None:
        return {n: p.grad for n, p in model.named_parameters()}, 0.0
    H, Hp = lw
    sd_shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    for n, p in model.named_parameters():
        g = p.grad
        if g is None:
            out[n] = None
            continue
        lg = unpad_tensor(g, sd_shapes[n], H, Hp)
        out[n] = lg
        if tuple(g.shape) != sd_shapes[n]:
            rest = g.clone()
            rest[_index(sd_shapes[n], g.shape, H, Hp, g.device)] = 0
            pad_max = max(pad_max, float(rest.abs().max()))
    return out, pad_max
