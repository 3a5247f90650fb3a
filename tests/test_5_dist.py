"""N>1 path on CPU: two gloo ranks, weights broadcast from rank 0, images sharded round-robin, metric all-reduce.
(The GPU path is identical with backend "nccl" = RCCL; bench.py --gpus N uses these same functions.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeModel:
    def __init__(self, seed):
        g = torch.Generator().manual_seed(seed)
        self.w = {"a.w": torch.randn(37, 16, generator=g).to(torch.bfloat16), "a.b": torch.randn(16, generator=g),
                  "b.w": torch.randn(5, 8, generator=g).to(torch.bfloat16), "tab": torch.randn(3, 3, generator=g)}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from psalm_amd.dist import broadcast_weights, reduce_metrics, shard_indices
    m = _FakeModel(seed=rank)                      # ranks start with DIFFERENT weights
    nbytes, _ = broadcast_weights(m, src=0, bucket_bytes=256)   # tiny buckets: exercises the multi-bucket path
    ref = _FakeModel(seed=0)
    same = all(torch.equal(m.w[k], ref.w[k]) for k in ref.w)
    mine = shard_indices(7, rank, world)
    v = reduce_metrics(torch.tensor([float(len(mine)), float(sum(mine)), 1.0]))
    q.put((rank, same, nbytes, mine, v.tolist()))
    dist.destroy_process_group()


@pytest.mark.slow
def test_two_rank_broadcast_shard_reduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    expect_bytes = sum(t.numel() * t.element_size() for t in _FakeModel(0).w.values())
    assert [r[1] for r in res] == [True, True]
    assert all(r[2] == expect_bytes for r in res)
    assert res[0][3] == [0, 2, 4, 6] and res[1][3] == [1, 3, 5]          # round-robin, disjoint, complete
    assert res[0][4] == res[1][4] == [7.0, 21.0, 2.0]
