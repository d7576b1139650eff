"""get_dataloaders(args) — mirror of MERBench/toolkit/dataloader/__init__.py:14-42 (MER2024/toolkit/dataloader/__init__.py for
the MER2024 entry) for the corpora on the hot path.  Other corpora keep the same pattern and are out of scope."""
from .mer2023 import MER2023
from .mer2024 import MER2024

DATALOADER_MAP = {'MER2023': MER2023, 'MER2024': MER2024}


class get_dataloaders:
    def __init__(self, args):
        if getattr(args, 'train_dataset', None) is not None:
            raise NotImplementedError('cross-corpus loaders (CROSSDIM/CROSSDIS) are outside the hot-path scope')
        if args.dataset not in DATALOADER_MAP:
            raise NotImplementedError(f"dataset '{args.dataset}' is outside the hot-path scope (supported: {list(DATALOADER_MAP)})")
        self.dataloader = DATALOADER_MAP[args.dataset](args)

    def get_loaders(self):
        return self.dataloader.get_loaders()

    def calculate_results(self, emo_probs=[], emo_labels=[], val_preds=[], val_labels=[]):
        return self.dataloader.calculate_results(emo_probs, emo_labels, val_preds, val_labels)
