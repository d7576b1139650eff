"""CPU restatement of the fusion classifier + losses + Adam step — TEST INFRASTRUCTURE ONLY.

Follows MERBench/toolkit/models/attention.py:36-57, modules/encoder.py:30-41, utils/loss.py:5-28 and the
optimiser call of main-release.py:205 (torch.optim.Adam, L2 weight decay), written functionally on a plain
state_dict with torch CPU ops.  Pinned against tests/golden/fusion_*.npz and losses.npz, which were produced by
running the reference's own modules (tests/golden/gen_golden.py).
"""
import torch
import torch.nn.functional as F


def mlp_encoder(sd, prefix, x):
    for i in (1, 2, 3):
        x = F.relu(F.linear(x, sd[f"{prefix}.linear_{i}.weight"], sd[f"{prefix}.linear_{i}.bias"]))
    return x


def attention_forward(sd, batch, prefix=""):
    """(features, emos_out, vals_out) of `Attention.forward` in eval mode / dropout 0."""
    a = mlp_encoder(sd, prefix + "audio_encoder", batch["audios"])
    t = mlp_encoder(sd, prefix + "text_encoder", batch["texts"])
    v = mlp_encoder(sd, prefix + "video_encoder", batch["videos"])
    m1 = torch.cat([a, t, v], dim=1)
    att = F.linear(mlp_encoder(sd, prefix + "attention_mlp", m1), sd[prefix + "fc_att.weight"], sd[prefix + "fc_att.bias"])
    fused = torch.matmul(torch.stack([a, t, v], dim=2), att.unsqueeze(2)).squeeze(2)
    return (fused, F.linear(fused, sd[prefix + "fc_out_1.weight"], sd[prefix + "fc_out_1.bias"]),
            F.linear(fused, sd[prefix + "fc_out_2.weight"], sd[prefix + "fc_out_2.bias"]))


def ce_loss(pred, target):
    return F.nll_loss(F.log_softmax(pred, 1), target.long(), reduction="sum") / len(pred)


def mse_loss(pred, target):
    return F.mse_loss(pred.view(-1, 1), target.view(-1, 1), reduction="sum") / len(pred)


def adam_step(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, wd=0.0):
    """torch.optim.Adam single-tensor update (amsgrad off); returns new (p, m, v)."""
    g = g + wd * p
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    denom = v.sqrt() / (1 - b2 ** step) ** 0.5 + eps
    return p - (lr / (1 - b1 ** step)) * (m / denom), m, v


def train_steps(sd, xs, emos, vals, steps, lr=1e-3, wd=1e-5):
    """Runs `steps` full-batch Adam steps with autograd on CPU; returns (losses, final state_dict, first-step grads)."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ms = {k: torch.zeros_like(v) for k, v in sd.items()}
    vs = {k: torch.zeros_like(v) for k, v in sd.items()}
    losses, grads0 = [], None
    for s in range(steps):
        batch = {k: x[s] for k, x in xs.items()}
        f, e, v = attention_forward(params, batch)
        loss = ce_loss(e, emos[s]) + mse_loss(v, vals[s])
        grads = torch.autograd.grad(loss, list(params.values()))
        if s == 0:
            grads0 = {k: g.clone() for k, g in zip(params, grads)}
        with torch.no_grad():
            for (k, p), g in zip(list(params.items()), grads):
                np_, ms[k], vs[k] = adam_step(p.detach(), g, ms[k], vs[k], s + 1, lr=lr, wd=wd)
                params[k] = np_.requires_grad_(True)
        losses.append(loss.item())
    return losses, {k: v.detach() for k, v in params.items()}, grads0
