"""N > 1 path on CPU: world_size-2 (and 3) `gloo` process groups exercising the utterance sharding + variable-length gather the
batched-serving path and bench.py use on the GPUs over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import load_pkg


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _fake_utterance(i, item):
    """Deterministic stand-in for a synthesis: length and content depend on the GLOBAL index only."""
    g = torch.Generator().manual_seed(1000 + i)
    n = 50 + 37 * (i % 5) + int(item)
    return torch.randn(n, generator=g)


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        bd = load_pkg("batch_dispatch")
        items = list(range(n_items))
        out = bd.run_sharded(items, _fake_utterance)
        ref = [_fake_utterance(i, it) for i, it in enumerate(items)]
        ok = len(out) == n_items and all(torch.equal(a, b) for a, b in zip(out, ref))
        # int32 token grids take the same path as fp32 audio
        toks = bd.run_sharded(items, lambda i, it: torch.arange(i + 1, dtype=torch.int32))
        ok = ok and all(torch.equal(t, torch.arange(i + 1, dtype=torch.int32)) for i, t in enumerate(toks))
        # int64 / bf16 payloads with an EMPTY rank (n_items < world): the empty rank must adopt the others' dtype (8- and 2-byte
        # elements: a float32 fallback would enter the collective with a different size)
        big = bd.run_sharded(items, lambda i, it: torch.arange(i + 3, dtype=torch.int64) << 40)
        ok = ok and all(t.dtype == torch.int64 and torch.equal(t, torch.arange(i + 3, dtype=torch.int64) << 40) for i, t in enumerate(big))
        half = bd.run_sharded(items, lambda i, it: torch.full((i + 2,), 0.5 * (i + 1), dtype=torch.bfloat16))
        ok = ok and all(t.dtype == torch.bfloat16 and torch.equal(t, torch.full((i + 2,), 0.5 * (i + 1), dtype=torch.bfloat16)) for i, t in enumerate(half))
        q.put((rank, ok, bd.shard_indices(n_items, rank, world)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n_items", [(2, 7), (2, 2), (3, 8), (2, 1), (3, 1)])
def test_sharded_generation_gathers_in_input_order(world, n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_items, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in results)
    covered = sorted(i for _, _, idx in results for i in idx)
    assert covered == list(range(n_items))  # every utterance on exactly one rank


def test_single_process_degenerates_to_a_loop():
    bd = load_pkg("batch_dispatch")
    out = bd.run_sharded([3, 1, 4], _fake_utterance)
    assert [t.numel() for t in out] == [50 + 3, 50 + 37 + 1, 50 + 74 + 4]
