"""Attention_TOPN — attention fusion over the top-n features of every modality (up to 18 streams), the multi-feature variant of
`Attention`: mirror of MER2024/toolkit/models/attention_topn.py:7-89 (same ctor args — the list of feature widths arrives as
`args.audio_dim` —, the reference's attribute names `encoder0 .. encoder17` and therefore its state_dict keys, the 4-tuple
return) with every Linear / fuse step on the HIP kernels.  The reference spells the 18 encoders out one by one because a python
list of modules neither moves to the GPU nor registers parameters; `setattr` on the module does both and keeps the names."""
import torch
import torch.nn as nn

from ...fusion_ops import FuseFn, linear
from .modules.encoder import MLPEncoder

MAX_STREAMS = 3 * 6


class Attention_TOPN(nn.Module):
    def __init__(self, args):
        super().__init__()
        feat_dims = list(args.audio_dim)           # Data_Feat_TOPN.get_featdim() returns the width list three times
        assert 1 <= len(feat_dims) <= MAX_STREAMS
        hidden_dim, p = args.hidden_dim, args.dropout
        self.grad_clip = args.grad_clip
        self.feat_dims = feat_dims
        for i, dim in enumerate(feat_dims):
            setattr(self, f'encoder{i}', MLPEncoder(dim, hidden_dim, p))
        self.attention_mlp = MLPEncoder(hidden_dim * len(feat_dims), hidden_dim, p)
        self.fc_att = nn.Linear(hidden_dim, len(feat_dims))
        self.fc_out_1 = nn.Linear(hidden_dim, args.output_dim1)
        self.fc_out_2 = nn.Linear(hidden_dim, args.output_dim2)

    def forward(self, batch):
        hiddens = [getattr(self, f'encoder{i}')(batch[f'feat{i}']) for i in range(len(self.feat_dims))]
        multi_hidden1 = torch.cat(hiddens, dim=1)                                    # [B, n * H]
        attention = linear(self.attention_mlp(multi_hidden1), self.fc_att)           # [B, n], no softmax
        features = FuseFn.apply(multi_hidden1, attention)                            # == matmul([B,H,n],[B,n,1]).squeeze(2)
        emos_out = linear(features, self.fc_out_1)
        vals_out = linear(features, self.fc_out_2)
        interloss = torch.zeros((), dtype=torch.int64, device=features.device)
        return features, emos_out, vals_out, interloss
