"""BASELINE configs[3]: clip-sharded tri-modal extraction feeding an Attention-fusion training step.

Every optimiser step consumes one global minibatch of B clips.  Rank r extracts the audio / visual / text features of its
contiguous block of B/W clips on its own GPU (weights replicated, no data-path collective), then ONE fused RCCL
all-gather (`distributed.gather_fusion_batch`: [B/W, Da + Dt + Dv + 2] rows, labels riding along) gives every rank the
full minibatch in the original clip order, and every rank replays the identical captured fusion step
(`FusionGraphTrainer.train_step`) on it.  No gradient all-reduce is needed and the parameters stay bit-identical across
ranks — and identical to a single-GPU run, because the gathered row order is the single-GPU order.

Reference loops this replaces: the extractor mains (MERBench/feature_extraction/*/extract_*_huggingface.py, one clip per
forward, features through .npy files) followed by MERBench/main-release.py:193-253 (fusion training on the dumped files).
"""
import torch

from . import distributed


def minibatch_block(n, rank, world):
    """[lo, hi) of rank's contiguous block of an n-clip minibatch (blocks differ by at most one clip; rank order == clip order)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class ExtractAndFuse:
    """encoders: dict with 'audio', 'visual', 'text' objects exposing `extract_utterance` (HipHubertModel, HipCLIPModel,
    HipBertModel — or stand-ins in the CPU tests); trainer: FusionGraphTrainer (or anything with train_step(batch, emos, vals)).

    A minibatch is a dict (tensors on the host or already on the device):
        audio [B, L] fp32 (normalised), frames [B*F, 3, S, S] fp32, frames_per_clip (list of B), input_ids [B, T] int64,
        lengths (list of B), emos [B] int64, vals [B] fp32
    `step(minibatch)` takes the FULL minibatch description on every rank (cheap: host tensors / a shared sampler) and touches
    only this rank's block of it."""

    def __init__(self, encoders, trainer, device, rank=None, world=None, text_strip=(1, -1)):
        self.enc, self.trainer = encoders, trainer
        self.device = torch.device(device)
        if rank is None or world is None:
            rank, world = distributed.rank_world()
        self.rank, self.world = rank, world
        self.text_strip = text_strip
        self.cuda = self.device.type == "cuda"
        self.streams = {m: torch.cuda.Stream(device=self.device) for m in ("visual", "audio", "text")} if self.cuda else None

    def local_features(self, mb):
        """This rank's block of the minibatch through the three encoders (one HIP stream each) -> (audios, texts, videos) on the device."""
        B = len(mb["lengths"])
        lo, hi = minibatch_block(B, self.rank, self.world)
        fpc = list(mb["frames_per_clip"])
        f0, f1 = sum(fpc[:lo]), sum(fpc[:hi])
        dev = self.device
        xa = torch.as_tensor(mb["audio"])[lo:hi].to(dev, non_blocking=True)
        xv = torch.as_tensor(mb["frames"])[f0:f1].to(dev, non_blocking=True)
        xt = torch.as_tensor(mb["input_ids"])[lo:hi].to(dev, non_blocking=True)
        start, end = self.text_strip
        jobs = {"visual": lambda: self.enc["visual"].extract_utterance(xv, fpc[lo:hi]),
                "audio": lambda: self.enc["audio"].extract_utterance(xa),
                "text": lambda: self.enc["text"].extract_utterance(xt, list(mb["lengths"])[lo:hi], start, end)}
        out = {}
        if self.cuda:
            cur = torch.cuda.current_stream(dev)
            for m, st in self.streams.items():   # longest first; the three forwards overlap each other's kernel tails
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    out[m] = jobs[m]()
            for st in self.streams.values():
                cur.wait_stream(st)
        else:
            for m in jobs:
                out[m] = jobs[m]()
        return out["audio"], out["text"], out["visual"], (lo, hi)

    def step(self, mb):
        a, t, v, (lo, hi) = self.local_features(mb)
        B = len(mb["lengths"])
        emos = torch.as_tensor(mb["emos"])[lo:hi].to(self.device, non_blocking=True)
        vals = torch.as_tensor(mb["vals"])[lo:hi].to(self.device, non_blocking=True)
        counts = [minibatch_block(B, r, self.world)[1] - minibatch_block(B, r, self.world)[0] for r in range(self.world)]
        fa, ft, fv, fe, fvl = distributed.gather_fusion_batch(a, t, v, emos, vals, counts=counts)
        return self.trainer.train_step(dict(audios=fa, texts=ft, videos=fv), fe, fvl)
