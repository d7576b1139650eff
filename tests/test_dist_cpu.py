"""The N>1 path on CPU: world_size-2 gloo processes exercise clip sharding and the fused feature all-gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mertools_amd import distributed as D
    r, w = D.init(backend="gloo")
    assert (r, w) == (rank, world)
    clips = [f"clip{i:03d}" for i in range(11)][::-1]
    mine = D.shard(clips)
    allc = [None] * world
    dist.all_gather_object(allc, mine)
    assert sorted(sum(allc, [])) == sorted(clips) and len(set(sum(allc, []))) == len(clips)
    assert mine == sorted(clips)[rank::world]
    # ragged per-rank minibatch (rank 0: 3 rows, rank 1: 2 rows)
    n = 3 - rank
    g = torch.Generator().manual_seed(100 + rank)
    a, t, v = torch.randn(n, 8, generator=g), torch.randn(n, 6, generator=g), torch.randn(n, 4, generator=g)
    emos, vals = torch.arange(n) + 10 * rank, torch.randn(n, generator=g)
    fa, ft, fv, fe, fvl = D.gather_fusion_batch(a, t, v, emos, vals)
    exp = []
    for rr in range(world):
        gg = torch.Generator().manual_seed(100 + rr)
        nn = 3 - rr
        exp.append((torch.randn(nn, 8, generator=gg), torch.randn(nn, 6, generator=gg), torch.randn(nn, 4, generator=gg),
                    torch.arange(nn) + 10 * rr, torch.randn(nn, generator=gg)))
    for got, idx in [(fa, 0), (ft, 1), (fv, 2), (fe, 3), (fvl, 4)]:
        assert torch.equal(got, torch.cat([e[idx] for e in exp], 0)), idx
    assert fe.dtype == torch.int64
    D.barrier()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")
    dist.destroy_process_group()


def test_shard_and_fused_allgather_gloo_world2(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    assert all(os.path.exists(tmp_path / f"ok{r}") for r in range(world))


def test_single_process_is_identity():
    from mertools_amd import distributed as D
    x = torch.randn(4, 5)
    assert D.all_gather_rows(x) is x
    assert D.shard(["b", "a", "c"], 0, 1) == ["a", "b", "c"]
    assert D.shard(list(range(10)), 1, 4) == [1, 5, 9]


# ---- BASELINE configs[3] end to end (mertools_amd.config4) with stand-in encoders and a stand-in fusion trainer ----
class _Enc:
    """Deterministic stand-in encoder: a fixed random projection of its input rows (torch CPU, test infrastructure)."""

    def __init__(self, din, dout, seed):
        self.w = torch.randn(din, dout, generator=torch.Generator().manual_seed(seed)) / din ** 0.5


class _A(_Enc):
    def extract_utterance(self, x):
        return x[:, :self.w.shape[0]].float() @ self.w


class _V(_Enc):
    def extract_utterance(self, x, frames_per_clip):
        f = x.reshape(x.shape[0], -1)[:, :self.w.shape[0]].float() @ self.w
        out, r = [], 0
        for n in frames_per_clip:
            out.append(f[r:r + n].mean(0))
            r += n
        return torch.stack(out)


class _T(_Enc):
    def extract_utterance(self, ids, lengths, start, end):
        return (ids.float() / 1000.0)[:, :self.w.shape[0]] @ self.w


class _Trainer:
    """Stand-in for FusionGraphTrainer: a torch-CPU fusion MLP + Adam with the same train_step(batch, emos, vals) interface."""

    def __init__(self):
        torch.manual_seed(0)
        self.net = torch.nn.Sequential(torch.nn.Linear(12 + 10 + 8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 7))
        self.opt = torch.optim.Adam(self.net.parameters(), lr=1e-2)

    def train_step(self, batch, emos, vals):
        self.opt.zero_grad()
        out = self.net(torch.cat([batch["audios"], batch["texts"], batch["videos"]], 1))
        loss = torch.nn.functional.cross_entropy(out[:, :6], emos) + torch.nn.functional.mse_loss(out[:, 6], vals)
        loss.backward()
        self.opt.step()
        return loss, out[:, :6], out[:, 6:]

    def flat(self):
        return torch.cat([p.detach().reshape(-1) for p in self.net.parameters()])


def _minibatches(n_steps, B):
    g = torch.Generator().manual_seed(42)
    for _ in range(n_steps):
        fpc = [int(x) for x in torch.randint(1, 4, (B,), generator=g)]
        yield dict(audio=torch.randn(B, 40, generator=g), frames=torch.randn(sum(fpc), 3, 4, 4, generator=g), frames_per_clip=fpc,
                   input_ids=torch.randint(0, 1000, (B, 12), generator=g), lengths=[12] * B,
                   emos=torch.randint(0, 6, (B,), generator=g), vals=torch.randn(B, generator=g))


def _config4_run(rank, world):
    from mertools_amd.config4 import ExtractAndFuse
    tr = _Trainer()
    pipe = ExtractAndFuse({"audio": _A(40, 12, 1), "visual": _V(48, 8, 2), "text": _T(12, 10, 3)}, tr, "cpu", rank, world)
    losses = [float(pipe.step(mb)[0]) for mb in _minibatches(6, 11)]   # 11 clips: ragged blocks (6 + 5)
    return tr.flat(), losses


def _config4_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from mertools_amd import distributed as D
    D.init(backend="gloo")
    flat, losses = _config4_run(rank, world)
    torch.save((flat, losses), os.path.join(tmp, f"rank{rank}.pt"))
    D.barrier()
    dist.destroy_process_group()


def test_config4_extract_allgather_fusion_world2_matches_single_process(tmp_path):
    """Shard -> extract -> fused all-gather -> fusion step on 2 gloo ranks: both ranks end with bit-identical fusion parameters,
    and they equal the single-process run on the same minibatches (the gathered row order is the single-process order)."""
    world = 2
    mp.spawn(_config4_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    (f0, l0), (f1, l1) = torch.load(tmp_path / "rank0.pt"), torch.load(tmp_path / "rank1.pt")
    assert torch.equal(f0, f1) and l0 == l1
    fs, ls = _config4_run(0, 1)
    assert torch.equal(f0, fs) and l0 == ls


def test_all_gather_rows_always_trims(tmp_path):
    from mertools_amd.config4 import minibatch_block
    assert [minibatch_block(11, r, 2) for r in range(2)] == [(0, 6), (6, 11)]
    assert [minibatch_block(8, r, 4) for r in range(4)] == [(0, 2), (2, 4), (4, 6), (6, 8)]
    assert [minibatch_block(3, r, 4) for r in range(4)] == [(0, 1), (1, 2), (2, 3), (3, 3)]
