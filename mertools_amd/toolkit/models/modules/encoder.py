"""MLPEncoder — mirror of MERBench/toolkit/models/modules/encoder.py:9-41 on the HIP fusion kernels.
Parameters live in nn.Linear containers so state_dict keys (linear_{1,2,3}.{weight,bias}) and default init
(nn.Linear's kaiming-uniform) are the reference's; forward = relu(L3(relu(L2(relu(L1(dropout(x)))))))."""
import torch.nn as nn

from ....fusion_ops import dropout, linear


class MLPEncoder(nn.Module):
    def __init__(self, in_size, hidden_size, dropout):  # noqa: A002 (reference argument name)
        super().__init__()
        self.drop = nn.Dropout(p=dropout)  # holds p; the masking itself runs in mer_dropout
        self.linear_1 = nn.Linear(in_size, hidden_size)
        self.linear_2 = nn.Linear(hidden_size, hidden_size)
        self.linear_3 = nn.Linear(hidden_size, hidden_size)

    def forward(self, x):
        dropped = dropout(x, self.drop.p, self.training)
        y_1 = linear(dropped, self.linear_1, relu=True)
        y_2 = linear(y_1, self.linear_2, relu=True)
        return linear(y_2, self.linear_3, relu=True)


class LSTMEncoder(nn.Module):
    """frm_align / frm_unalign feature types (encoder.py:45-72) are outside this round's hot-path scope."""

    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("LSTMEncoder (frame-level fusion) is not built; use feat_type='utt'")
