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
