"""Multi-GPU plumbing: one process per GPU under torchrun, torch.distributed over RCCL/xGMI.

  * extraction shards embarrassingly: rank r owns sorted(clips)[r::W]; weights are replicated; every rank
    writes its own .npy files; there is NO data-path collective (an optional barrier at the end).
  * fusion training on freshly extracted features needs the full minibatch on every rank so that all ranks take
    the identical optimiser step (no gradient all-reduce, bit-parity with single-GPU): the per-rank
    [B/W, Da+Dt+Dv] feature rows are packed into ONE buffer and exchanged with ONE all-gather per step.  The
    message is a few hundred KB, i.e. latency-bound, which is why it is a single fused collective.
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None, device=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*). Returns (rank, world)."""
    if not dist.is_initialized():
        world = int(os.environ.get("WORLD_SIZE", "1"))
        if world == 1:
            return 0, 1
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            local = int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(local)
            kw["device_id"] = device or torch.device(f"cuda:{local}")
        dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()


def rank_world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


def shard(items, rank=None, world=None):
    """Rank r's share of the work list: sorted(items)[r::W] — deterministic, disjoint, covers everything."""
    if rank is None:
        rank, world = rank_world()
    return sorted(items)[rank::world]


def my_share(items, rank=None, world=None):
    """The extraction drivers' work list for this process: the list itself (original order) when there is one process,
    else shard().  rank / world default to the torch.distributed values (0 / 1 when not initialised)."""
    if rank is None or world is None:
        rank, world = rank_world()
    return list(items) if world <= 1 else shard(items, rank, world)


def barrier():
    if dist.is_initialized():
        dist.barrier()


def all_gather_rows(local, max_rows=None, counts=None):
    """Concatenate per-rank row blocks [n_r, D] in rank order; ranks may hold different n_r (a ragged last minibatch).
    Rows are padded to a common height for the collective and the padding is ALWAYS trimmed before returning:
      counts   per-rank row counts when the caller already knows them (no extra collective);
      max_rows common padded height when known (saves nothing on its own: without `counts` the row counts are exchanged
               anyway, because untrimmed zero rows would reach the fusion step as fake label-0 samples)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    if counts is None:
        n = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
        got = [torch.zeros_like(n) for _ in range(world)]
        dist.all_gather(got, n)
        counts = [int(c.item()) for c in got]
    counts = [int(c) for c in counts]
    assert len(counts) == world and counts[dist.get_rank()] == local.shape[0], (counts, local.shape)
    height = max(counts) if max_rows is None else int(max_rows)
    assert height >= max(counts), f"max_rows={height} < largest rank block {max(counts)}"
    if all(c == height for c in counts):
        out = local.new_empty((world * height,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    buf = local.new_zeros((height,) + tuple(local.shape[1:]))
    buf[:local.shape[0]] = local
    out = local.new_empty((world * height,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, buf)
    return torch.cat([out[r * height: r * height + c] for r, c in enumerate(counts)], 0)


def gather_fusion_batch(audios, texts, videos, emos=None, vals=None, counts=None):
    """One fused all-gather of the minibatch every rank needs for the identical fusion step.
    Inputs are this rank's rows; returns the full-batch (audios, texts, videos[, emos, vals]) in rank order.
    counts: per-rank row counts when known (equal shards: [B/W] * W) — skips the count exchange."""
    da, dt, dv = audios.shape[1], texts.shape[1], videos.shape[1]
    cols = [audios.float(), texts.float(), videos.float()]
    if emos is not None:
        cols.append(emos.float()[:, None])  # class ids < 2^24 are exact in fp32
    if vals is not None:
        cols.append(vals.float()[:, None])
    full = all_gather_rows(torch.cat(cols, 1).contiguous(), counts=counts)
    out = [full[:, :da], full[:, da:da + dt], full[:, da + dt:da + dt + dv]]
    c = da + dt + dv
    if emos is not None:
        out.append(full[:, c].round().long())
        c += 1
    if vals is not None:
        out.append(full[:, c])
    return tuple(out)
