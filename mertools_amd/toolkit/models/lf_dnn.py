"""LF_DNN (late-fusion MLP) — mirror of MER2024/toolkit/models/lf_dnn.py:12-65 on the HIP kernels."""
import torch
import torch.nn as nn

from ...fusion_ops import dropout, linear
from .modules.encoder import LSTMEncoder, MLPEncoder


class LF_DNN(nn.Module):
    def __init__(self, args):
        super().__init__()
        hidden_dim, p = args.hidden_dim, args.dropout
        self.grad_clip = args.grad_clip
        enc = MLPEncoder if args.feat_type in ['utt'] else LSTMEncoder
        self.audio_encoder = enc(args.audio_dim, hidden_dim, p)
        self.text_encoder = enc(args.text_dim, hidden_dim, p)
        self.video_encoder = enc(args.video_dim, hidden_dim, p)
        self.post_fusion_dropout = nn.Dropout(p=p)
        self.post_fusion_layer_1 = nn.Linear(hidden_dim * 3, hidden_dim)
        self.post_fusion_layer_2 = nn.Linear(hidden_dim, hidden_dim)
        self.fc_out_1 = nn.Linear(hidden_dim, args.output_dim1)
        self.fc_out_2 = nn.Linear(hidden_dim, args.output_dim2)

    def forward(self, batch):
        audio_h = self.audio_encoder(batch['audios'])
        video_h = self.video_encoder(batch['videos'])
        text_h = self.text_encoder(batch['texts'])
        fusion_h = torch.cat([audio_h, video_h, text_h], dim=-1)  # note the a,v,t order of the reference
        x = dropout(fusion_h, self.post_fusion_dropout.p, self.training)
        x = linear(x, self.post_fusion_layer_1, relu=True)
        features = linear(x, self.post_fusion_layer_2, relu=True)
        emos_out = linear(features, self.fc_out_1)
        vals_out = linear(features, self.fc_out_2)
        interloss = torch.zeros((), dtype=torch.int64, device=features.device)
        return features, emos_out, vals_out, interloss
