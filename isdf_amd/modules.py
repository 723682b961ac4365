"""nn.Module mirror of the reference's `SDFMap` / `PostionalEncoding`
(`isdf/modules/fc_map.py:63-111`, `isdf/modules/embedding.py:24-111`).

`trainer.sdf_map` must stay a real `nn.Module` with live `.parameters()` and the
reference's state_dict keys, because 16 inference / visualisation / evaluation
call sites, `copy.deepcopy`, `torch.save` and `load_state_dict` use it
(SURVEY 3.4, 8b).  Here the 14 parameters are VIEWS of the engine's flat fp32
buffer (so the kernels see one contiguous allocation and RCCL one message), and
`forward` runs the fused HIP inference kernel instead of eager torch ops.
"""
import copy

import numpy as np
import torch
import torch.nn as nn

from .engine import Engine, NetConfig


class PositionalEncodingHIP:
    """Carries the PE hyper-parameters (attribute names of the reference class;
    like there, `transform` is a plain attribute, not a buffer -- SURVEY 5)."""

    def __init__(self, min_deg=0, max_deg=5, scale=0.05937489, transform=None):
        self.min_deg, self.max_deg = min_deg, max_deg
        self.n_freqs = max_deg - min_deg + 1
        self.scale = scale
        self.transform = transform
        self.embedding_size = 2 * 21 * self.n_freqs + 3


def _fc_block(in_f, out_f):
    return nn.Sequential(nn.Linear(in_f, out_f), nn.Softplus(beta=100))


class SDFMapHIP(nn.Module):
    def __init__(self, positional_encoding, hidden_size=256, hidden_layers_block=1, scale_output=1.0,
                 device="cuda", fwd_operand="fp16x2", engine_factory=None, bwd_operand=None, spill_operand=None):
        super().__init__()
        object.__setattr__(self, "_engine_factory", engine_factory)
        self.scale_output = scale_output
        self.positional_encoding = positional_encoding
        E = positional_encoding.embedding_size
        # same construction + init order as the reference, so a seeded run draws
        # the same initial weights (fc_map.py:77-92: Linear default init, then
        # xavier_normal_ on every Linear weight via self.apply)
        self.in_layer = _fc_block(E, hidden_size)
        self.mid1 = nn.Sequential(*[_fc_block(hidden_size, hidden_size) for _ in range(hidden_layers_block)])
        self.cat_layer = _fc_block(hidden_size + E, hidden_size)
        self.mid2 = nn.Sequential(*[_fc_block(hidden_size, hidden_size) for _ in range(hidden_layers_block)])
        self.out_alpha = nn.Linear(hidden_size, 1)

        def init_weights(m):
            if isinstance(m, nn.Linear):
                nn.init.xavier_normal_(m.weight)
        self.apply(init_weights)

        T = positional_encoding.transform
        if torch.is_tensor(T):
            T = T.detach().cpu().numpy()
        net = NetConfig(hidden=hidden_size, blocks=hidden_layers_block, n_freqs=positional_encoding.n_freqs,
                        scale_input=positional_encoding.scale, scale_output=scale_output,
                        transform=None if T is None else np.asarray(T, np.float32), fwd_operand=fwd_operand,
                        bwd_operand=bwd_operand, spill_operand=spill_operand)
        make = Engine if engine_factory is None else engine_factory          # engine_factory: host-logic tests only
        object.__setattr__(self, "engine", make(net, device))   # not a submodule / not in state_dict
        self._bind()

    def _bind(self):
        """Re-point every parameter at its view of the engine's flat buffer."""
        eng = self.engine
        init = {k: p.detach().clone() for k, p in self.named_parameters()}
        eng.load_params(init)
        for k, p in self.named_parameters():
            p.data = eng.param_view(k)
        self._views_ok()

    def _views_ok(self):
        base = self.engine.params.data_ptr()
        for k, p in self.named_parameters():
            off, _ = self.engine.slices[k]
            if p.data_ptr() != base + 4 * off:
                raise RuntimeError("parameter %s no longer aliases the flat buffer" % k)

    def refresh(self):
        """Call after parameters were modified outside the engine (load_state_dict, manual edits)."""
        self._views_ok()
        self.engine.pack()

    def load_state_dict(self, state_dict, strict=True):
        out = super().load_state_dict(state_dict, strict)   # copies in place: views stay views
        self.refresh()
        return out

    def __deepcopy__(self, memo):
        """`copy.deepcopy(self.sdf_map)` (trainer.py:576): a frozen snapshot with its own buffers."""
        pe = copy.copy(self.positional_encoding)
        with torch.random.fork_rng(devices=[]):    # the throw-away initial weights must not advance the caller's generator
            new = SDFMapHIP(pe, self.engine.net.hidden, self.engine.net.blocks, self.scale_output,   # (upstream's deepcopy
                            device=self.engine.device, fwd_operand=self.engine.net.fwd_operand,      # draws nothing)
                            bwd_operand=self.engine.net.bwd_operand, spill_operand=getattr(self.engine.net, "spill_operand", None),
                            engine_factory=self._engine_factory)
        new.engine.params.copy_(self.engine.params)
        new.engine.pack()
        new.train(self.training)
        return new

    def forward(self, x, noise_std=None, pe_mask=None, sdf1=None):
        """`SDFMap.forward` (fc_map.py:94-111) on the fused HIP inference kernel.  Weight gradients never flow
        through this call (training is Engine.train_step); the INPUT gradient does: when `x.requires_grad`
        the kernel also returns d sdf / d x and `fc_map.gradient(x, sdf)` / `autograd.grad` (fc_map.py:12-22;
        render.render_normals, render.py:39-47, and the slice/vis call sites use it) get it back first-order."""
        if pe_mask is not None:
            raise NotImplementedError("pe_mask is unused by the reference's training/eval paths")
        noise = None
        if noise_std is not None:     # drawn whenever noise_std is not None, even 0
            noise = torch.randn(x.shape[:-1], device=x.device) * noise_std
        if torch.is_grad_enabled() and x.requires_grad:
            return _SdfWithInputGrad.apply(x, self, noise)
        with torch.no_grad():
            return self.engine.sdf_eval(x, noise=noise)

    @torch.no_grad()
    def forward_with_grad(self, x, noise_std=None):
        """(sdf, d sdf / d x): `fc_map.gradient(x, sdf_map(x))` (fc_map.py:12-22) fused."""
        noise = None
        if noise_std is not None:
            noise = torch.randn(x.shape[:-1], device=x.device) * noise_std
        return self.engine.sdf_eval(x, noise=noise, want_grad=True)


class _SdfWithInputGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, noise):
        sdf, grad = module.engine.sdf_eval(x.detach(), noise=noise, want_grad=True)
        ctx.save_for_backward(grad)
        return sdf

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return g.unsqueeze(-1) * grad, None, None


def chunks(pc, chunk_size, fc_sdf_map, to_cpu=False):
    """`fc_map.chunks` (fc_map.py:25-48): batched inference over a large point set."""
    out = []
    for s in range(0, pc.shape[0], chunk_size):
        a = fc_sdf_map(pc[s:s + chunk_size, :]).squeeze(dim=-1)
        out.append(a.cpu() if to_cpu else a)
    return torch.cat(out, dim=-1)
