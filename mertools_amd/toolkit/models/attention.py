"""Attention fusion — mirror of MERBench/toolkit/models/attention.py:8-57 (same ctor args, attribute names,
state_dict keys and 4-tuple return) with every Linear / fuse step on the HIP kernels."""
import torch
import torch.nn as nn

from ...fusion_ops import FuseFn, linear
from .modules.encoder import LSTMEncoder, MLPEncoder


class Attention(nn.Module):
    def __init__(self, args):
        super().__init__()
        text_dim, audio_dim, video_dim = args.text_dim, args.audio_dim, args.video_dim
        output_dim1, output_dim2 = args.output_dim1, args.output_dim2
        dropout, hidden_dim = args.dropout, args.hidden_dim
        self.grad_clip = args.grad_clip
        enc = MLPEncoder if args.feat_type in ['utt'] else LSTMEncoder
        self.audio_encoder = enc(audio_dim, hidden_dim, dropout)
        self.text_encoder = enc(text_dim, hidden_dim, dropout)
        self.video_encoder = enc(video_dim, hidden_dim, dropout)
        self.attention_mlp = MLPEncoder(hidden_dim * 3, hidden_dim, dropout)
        self.fc_att = nn.Linear(hidden_dim, 3)
        self.fc_out_1 = nn.Linear(hidden_dim, output_dim1)
        self.fc_out_2 = nn.Linear(hidden_dim, output_dim2)

    def forward(self, batch):
        audio_hidden = self.audio_encoder(batch['audios'])
        text_hidden = self.text_encoder(batch['texts'])
        video_hidden = self.video_encoder(batch['videos'])
        multi_hidden1 = torch.cat([audio_hidden, text_hidden, video_hidden], dim=1)  # [B, 3H] (memory move only)
        attention = linear(self.attention_mlp(multi_hidden1), self.fc_att)           # [B, 3], no softmax
        features = FuseFn.apply(multi_hidden1, attention)                            # == matmul([B,H,3],[B,3,1]).squeeze(2)
        emos_out = linear(features, self.fc_out_1)
        vals_out = linear(features, self.fc_out_2)
        interloss = torch.zeros((), dtype=torch.int64, device=features.device)       # reference: torch.tensor(0).cuda()
        return features, emos_out, vals_out, interloss
