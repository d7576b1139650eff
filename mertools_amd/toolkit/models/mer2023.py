"""MER2023's single-file fusion models — mirror of MER2023/main-release.py:226-290 (`MLP`, `Attention`):
[Linear, ReLU, Dropout] x layers on cat(a,t,v) resp. per-modality + attention; 3-tuple return (no interloss)."""
import torch
import torch.nn as nn

from ...fusion_ops import FuseFn, dropout, linear


class _Seq(nn.Module):
    """nn.Sequential(Linear, ReLU, Dropout, ...) with the reference's parameter names (module.0, module.3, ...)."""

    def __init__(self, input_dim, layers, p):
        super().__init__()
        mods = []
        for width in layers:
            mods += [nn.Linear(input_dim, width), nn.ReLU(), nn.Dropout(p)]
            input_dim = width
        self.seq = nn.Sequential(*mods)
        self.p = p

    def forward(self, x):
        for m in self.seq:
            if isinstance(m, nn.Linear):
                x = linear(x, m, relu=True)
            elif isinstance(m, nn.Dropout):
                x = dropout(x, self.p, self.training)
        return x


def _widths(layers):
    return [int(x) for x in layers.split(',')]


class MLP(nn.Module):
    def __init__(self, input_dim, output_dim1, output_dim2=1, layers='256,128', dropout=0.3):  # noqa: A002
        super().__init__()
        w = _widths(layers)
        self.module = _Seq(input_dim, w, dropout).seq
        self.p = dropout
        self.fc_out_1 = nn.Linear(w[-1], output_dim1)
        self.fc_out_2 = nn.Linear(w[-1], output_dim2)

    def forward(self, inputs):
        x = inputs
        for m in self.module:
            if isinstance(m, nn.Linear):
                x = linear(x, m, relu=True)
            elif isinstance(m, nn.Dropout):
                x = dropout(x, self.p, self.training)
        return x, linear(x, self.fc_out_1), linear(x, self.fc_out_2)


class Attention(nn.Module):
    def __init__(self, audio_dim, text_dim, video_dim, output_dim1, output_dim2=1, layers='256,128', dropout=0.3):  # noqa: A002
        super().__init__()
        w = _widths(layers)
        self.p = dropout
        self.audio_mlp = _Seq(audio_dim, w, dropout).seq
        self.text_mlp = _Seq(text_dim, w, dropout).seq
        self.video_mlp = _Seq(video_dim, w, dropout).seq
        self.attention_mlp = _Seq(w[-1] * 3, w, dropout).seq
        self.fc_att = nn.Linear(w[-1], 3)
        self.fc_out_1 = nn.Linear(w[-1], output_dim1)
        self.fc_out_2 = nn.Linear(w[-1], output_dim2)

    def _mlp(self, seq, x):
        for m in seq:
            if isinstance(m, nn.Linear):
                x = linear(x, m, relu=True)
            elif isinstance(m, nn.Dropout):
                x = dropout(x, self.p, self.training)
        return x

    def forward(self, audio_feat, text_feat, video_feat):
        a, t, v = self._mlp(self.audio_mlp, audio_feat), self._mlp(self.text_mlp, text_feat), self._mlp(self.video_mlp, video_feat)
        cat = torch.cat([a, t, v], dim=1)
        att = linear(self._mlp(self.attention_mlp, cat), self.fc_att)
        fused = FuseFn.apply(cat, att)
        return fused, linear(fused, self.fc_out_1), linear(fused, self.fc_out_2)
