"""get_models(args) — mirror of MERBench/toolkit/models/__init__.py:18-46: looks up MODEL_MAP[args.model],
builds cls(args) as `.model`, forwards `batch`.  Hot-path models ('attention', 'lf_dnn', MER2024's 'attention_topn') run on the HIP
kernels; the other MMSA baselines keep their registry keys but are out of scope."""
import torch

from .attention import Attention
from .attention_topn import Attention_TOPN
from .lf_dnn import LF_DNN

_OUT_OF_SCOPE = ['lmf', 'misa', 'mmim', 'tfn', 'mfn', 'graph_mfn', 'mfm', 'mctn', 'mult', 'ef_lstm']


class get_models(torch.nn.Module):
    MODEL_MAP = {'attention': Attention, 'lf_dnn': LF_DNN, 'attention_topn': Attention_TOPN}

    def __init__(self, args):
        super().__init__()
        if args.model in _OUT_OF_SCOPE:
            raise NotImplementedError(f"fusion model '{args.model}' is outside the MI355X hot-path scope; use 'attention', 'attention_topn' or 'lf_dnn'")
        self.model = self.MODEL_MAP[args.model](args)

    def forward(self, batch):
        return self.model(batch)
