"""Autograd-aware front ends of the fusion-classifier HIP kernels.

Every arithmetic step of the fusion model's forward AND backward runs in libmer_hip.so (mer_gemm32 on
the exact-fp32 MFMA + the small kernels of csrc/fusion.hip); torch.autograd only records the graph, so
`loss.backward()` / `torch.optim.Adam(model.parameters())` in main-release.py keep working unchanged.
"""
import torch

from . import _lib
from .ops import _p, gemm32, stream


def _c(t):
    t = t.contiguous()
    if t.dtype != torch.float32:
        t = t.float()
    if not t.is_cuda:
        raise _lib.MerError("mertools_amd fusion ops need CUDA/HIP tensors (there is no CPU path)")
    return t


def relu_bwd(dy, y):
    dz = torch.empty_like(dy)
    _lib.check(_lib.lib().mer_relu_bwd(_p(dy), _p(y), _p(dz), dy.numel(), stream()), "mer_relu_bwd")
    return dz


def colsum(x):
    M, N = x.shape
    out = torch.empty((N,), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().mer_colsum(_p(x), M, N, x.stride(0), _p(out), 0, stream()), "mer_colsum")
    return out


class LinearFn(torch.autograd.Function):
    """y = [relu](x W^T + b); x [B,in], W [out,in] (nn.Linear layout)."""

    @staticmethod
    def forward(ctx, x, w, b, relu):
        x, w = _c(x), _c(w)
        y = gemm32(x, w, _c(b) if b is not None else None, "relu" if relu else None)
        ctx.relu, ctx.has_bias = relu, b is not None
        ctx.save_for_backward(x, w, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        dy = _c(dy)
        dz = relu_bwd(dy, y) if ctx.relu else dy
        dx = gemm32(dz, w, trans_w=True) if ctx.needs_input_grad[0] else None          # dz [B,out] @ W [out,in]
        dw = gemm32(dz, x, trans_a=True, trans_w=True) if ctx.needs_input_grad[1] else None  # dz^T [out,B] @ x [B,in]
        db = colsum(dz) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dw, db, None


class LSTMLastFn(torch.autograd.Function):
    """h_T of a single-layer unidirectional batch_first nn.LSTM started from zeros (LSTMEncoder: encoder.py:63-72 uses
    final_states[0] only).  x [B,T,D]; w_ih [4H,D], w_hh [4H,H], b_ih, b_hh [4H] (nn.LSTM parameter layouts)."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh):
        x, w_ih, w_hh = _c(x), _c(w_ih), _c(w_hh)
        B, T, D = x.shape
        H = w_hh.shape[1]
        x2 = x.reshape(B * T, D)
        gx = gemm32(x2, w_ih, _c(b_ih + b_hh))                       # [B*T, 4H]
        gates = torch.empty((B, T, 4 * H), dtype=torch.float32, device=x.device)
        cs = torch.empty((B, T, H), dtype=torch.float32, device=x.device)
        hs = torch.empty((B, T, H), dtype=torch.float32, device=x.device)
        _lib.check(_lib.lib().mer_lstm_fwd(_p(gx), _p(w_hh.t().contiguous()), B, T, H, _p(gates), _p(cs), _p(hs), stream()), "mer_lstm_fwd")
        ctx.save_for_backward(x2, w_ih, w_hh, gates, cs, hs)
        ctx.dims = (B, T, D, H)
        return hs[:, T - 1].contiguous()

    @staticmethod
    def backward(ctx, dh):
        x2, w_ih, w_hh, gates, cs, hs = ctx.saved_tensors
        B, T, D, H = ctx.dims
        dA = torch.empty((B, T, 4 * H), dtype=torch.float32, device=dh.device)
        _lib.check(_lib.lib().mer_lstm_bwd(_p(_c(dh)), _p(gates), _p(cs), _p(w_hh), B, T, H, _p(dA), stream()), "mer_lstm_bwd")
        dA2 = dA.reshape(B * T, 4 * H)
        dx = gemm32(dA2, w_ih, trans_w=True).reshape(B, T, D) if ctx.needs_input_grad[0] else None
        dw_ih = gemm32(dA2, x2, trans_a=True, trans_w=True)            # dA^T [4H, B*T] @ X [B*T, D]
        hprev = torch.zeros_like(hs)
        hprev[:, 1:] = hs[:, :-1]
        dw_hh = gemm32(dA2, hprev.reshape(B * T, H), trans_a=True, trans_w=True)
        db = colsum(dA2)
        return dx, dw_ih, dw_hh, db, db.clone()


def lstm_last(x, rnn):
    """Final hidden state of `rnn` (an nn.LSTM container holding the parameters) on the HIP kernels."""
    if rnn.num_layers != 1 or rnn.bidirectional or not rnn.batch_first:
        raise _lib.MerError("only the reference's configuration is built: one unidirectional batch_first LSTM layer")
    return LSTMLastFn.apply(x, rnn.weight_ih_l0, rnn.weight_hh_l0, rnn.bias_ih_l0, rnn.bias_hh_l0)


class FuseFn(torch.autograd.Function):
    """fused[b,:] = sum_e h[b, e*H:(e+1)*H] * att[b,e]  (attention.py:45-50; weights are not softmaxed)."""

    @staticmethod
    def forward(ctx, h, att):
        h, att = _c(h), _c(att)
        B, E = att.shape
        H = h.shape[1] // E
        out = torch.empty((B, H), dtype=torch.float32, device=h.device)
        _lib.check(_lib.lib().mer_fuse_fwd(_p(h), _p(att), _p(out), B, H, E, stream()), "mer_fuse_fwd")
        ctx.save_for_backward(h, att)
        return out

    @staticmethod
    def backward(ctx, dout):
        h, att = ctx.saved_tensors
        dout = _c(dout)
        B, E = att.shape
        H = h.shape[1] // E
        dh, datt = torch.empty_like(h), torch.empty_like(att)
        _lib.check(_lib.lib().mer_fuse_bwd(_p(dout), _p(h), _p(att), _p(dh), _p(datt), B, H, E, stream()), "mer_fuse_bwd")
        return dh, datt


class DropoutFn(torch.autograd.Function):
    """Inverted dropout; the keep-mask comes from torch's RNG (so torch.manual_seed governs it), the arithmetic is HIP."""

    @staticmethod
    def forward(ctx, x, p):
        x = _c(x)
        keep = (torch.rand(x.shape, device=x.device) >= p).to(torch.uint8)
        scale = 1.0 / (1.0 - p)
        out = torch.empty_like(x)
        _lib.check(_lib.lib().mer_dropout(_p(x), _p(keep), scale, _p(out), x.numel(), stream()), "mer_dropout")
        ctx.save_for_backward(keep)
        ctx.scale = scale
        return out

    @staticmethod
    def backward(ctx, dy):
        (keep,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(dy)
        _lib.check(_lib.lib().mer_dropout(_p(dy), _p(keep), ctx.scale, _p(dx), dy.numel(), stream()), "mer_dropout")
        return dx, None


def dropout(x, p, training):
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x, p)


def linear(x, lin, relu=False):
    if lin.out_features == 0:   # MER2024: output_dim2 = 0 -> nn.Linear(hidden, 0), an empty [B, 0] head (no kernel to launch)
        return x.new_zeros((x.shape[0], 0))
    return LinearFn.apply(x, lin.weight, lin.bias, relu)


class CELossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        pred = _c(pred)
        target = target.long().contiguous()
        B, Cn = pred.shape
        probs = torch.empty_like(pred)
        rows = torch.empty((B,), dtype=torch.float32, device=pred.device)
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        _lib.check(_lib.lib().mer_ce_loss(_p(pred), _p(target), B, Cn, _p(probs), _p(rows), _p(loss), stream()), "mer_ce_loss")
        ctx.save_for_backward(probs, target)
        return loss

    @staticmethod
    def backward(ctx, g):
        probs, target = ctx.saved_tensors
        B, Cn = probs.shape
        d = torch.empty_like(probs)
        _lib.check(_lib.lib().mer_ce_loss_bwd(_p(probs), _p(target), _p(_c(g)), _p(d), B, Cn, stream()), "mer_ce_loss_bwd")
        return d, None


class MSELossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        shape = pred.shape
        pred, target = _c(pred).reshape(-1), _c(target).reshape(-1)
        B = pred.numel()
        rows = torch.empty((B,), dtype=torch.float32, device=pred.device)
        loss = torch.empty((), dtype=torch.float32, device=pred.device)
        _lib.check(_lib.lib().mer_mse_loss(_p(pred), _p(target), B, _p(rows), _p(loss), stream()), "mer_mse_loss")
        ctx.save_for_backward(pred, target)
        ctx.shape = shape
        return loss

    @staticmethod
    def backward(ctx, g):
        pred, target = ctx.saved_tensors
        d = torch.empty_like(pred)
        _lib.check(_lib.lib().mer_mse_loss_bwd(_p(pred), _p(target), _p(_c(g)), _p(d), pred.numel(), stream()), "mer_mse_loss_bwd")
        return d.reshape(ctx.shape), None


class HipAdam(torch.optim.Optimizer):
    """torch.optim.Adam semantics (main-release.py:205) with the update in one HIP kernel per tensor;
    clip_value > 0 folds clip_grad_value_ (main-release.py:64-65) into the same kernel."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip_value=-1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, clip_value=clip_value))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad.contiguous()
                _lib.check(_lib.lib().mer_adam_step(_p(p.data), _p(g), _p(st["exp_avg"]), _p(st["exp_avg_sq"]), p.numel(),
                                                    group["lr"], b1, b2, group["eps"], group["weight_decay"], st["step"],
                                                    float(group["clip_value"]), stream()), "mer_adam_step")
