"""MLPEncoder — mirror of MERBench/toolkit/models/modules/encoder.py:9-41 on the HIP fusion kernels.
Parameters live in nn.Linear containers so state_dict keys (linear_{1,2,3}.{weight,bias}) and default init
(nn.Linear's kaiming-uniform) are the reference's; forward = relu(L3(relu(L2(relu(L1(dropout(x)))))))."""
import torch.nn as nn

from ....fusion_ops import dropout, linear, lstm_last


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
    """Mirror of MERBench/toolkit/models/modules/encoder.py:45-72 (frm_align / frm_unalign feature types): single-layer
    nn.LSTM (the container keeps the reference's parameter names rnn.weight_ih_l0 ... and default init), final hidden state
    -> dropout -> linear_1.  The recurrence, its BPTT and the projections run in libmer_hip.so (mer_lstm_fwd / mer_lstm_bwd /
    mer_gemm32).  Features are padded in FRONT (read_data.func_mapping_feature), which is why the final state is the summary."""

    def __init__(self, in_size, hidden_size, dropout, num_layers=1, bidirectional=False):  # noqa: A002
        super().__init__()
        if num_layers != 1 or bidirectional:
            raise NotImplementedError("LSTMEncoder: only num_layers=1, bidirectional=False (the reference's call sites) is built")
        self.rnn = nn.LSTM(in_size, hidden_size, num_layers=1, dropout=0.0, bidirectional=False, batch_first=True)
        self.drop = nn.Dropout(p=dropout)
        self.linear_1 = nn.Linear(hidden_size, hidden_size)

    def forward(self, x):
        h = lstm_last(x, self.rnn)
        return linear(dropout(h, self.drop.p, self.training), self.linear_1, relu=False)
