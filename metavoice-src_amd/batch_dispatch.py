"""Utterance-level data parallelism for the batched-serving path: one process per GPU, independent utterances sharded
across ranks, ONE gather of the finished audio / token grids at the end.

The reference has no multi-GPU code at all (`ddp = False`, fam/llm/config/finetune_params.py:63-67; one global TTS,
serving.py:46-51); the semantics to preserve are those of its batched slow path: results come back in INPUT order
(`fam/llm/mixins/causal.py:424`).  Utterances never interact (own prompt, KV cache, RNG), so there is no collective during
generation; weights are replicated (2.5 GB).  The gather is `torch.distributed.all_gather_into_tensor` on the process group
(backend "nccl" = RCCL over xGMI on the GPUs, "gloo" in the CPU tests): lengths first, then one zero-padded payload --
~5 MB per rank for 12.8 s of 24 kHz fp32 audio, ~0.2 ms on a 153 GB/s xGMI link, which is why a plain ring all-gather is
enough.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist


def shard_indices(n_items: int, rank: int, world_size: int) -> List[int]:
    """Round-robin assignment: item i runs on rank i % world_size (keeps per-rank load even when items arrive sorted by
    length, cf. `_sort_for_batching`, causal.py:290-338)."""
    return list(range(rank, n_items, world_size))


_DTYPE_CODES = [torch.float32, torch.int32, torch.int64, torch.bfloat16, torch.float16, torch.int16, torch.uint8, torch.float64]


def gather_variable(local: Sequence[torch.Tensor], n_items: int, group: Optional[dist.ProcessGroup] = None,
                    device: Optional[torch.device] = None, dtype: Optional[torch.dtype] = None) -> List[torch.Tensor]:
    """All ranks contribute the 1-D results of THEIR shard (`shard_indices` order); every rank returns the full list in
    input order.  Two collectives: lengths (int64) with the payload dtype's code riding along, then the zero-padded payload.
    A rank whose shard is empty (n_items < world_size) has no tensor to read the dtype from: it takes the one the other ranks
    announce, so every rank enters the payload collective with the same dtype and size."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        assert len(local) == n_items
        return [t.clone() for t in local]
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = shard_indices(n_items, rank, world)
    assert len(local) == len(mine), f"rank {rank} produced {len(local)} results for {len(mine)} items"
    per_rank = (n_items + world - 1) // world
    if dtype is None and len(local):
        dtype = local[0].dtype
    dev = device if device is not None else (local[0].device if len(local) else torch.device("cpu"))
    lens = torch.zeros(per_rank + 1, dtype=torch.int64, device=dev)  # last entry: dtype code, -1 = "no opinion"
    lens[per_rank] = _DTYPE_CODES.index(dtype) if dtype is not None else -1
    for j, t in enumerate(local):
        assert t.dim() == 1 and t.dtype == dtype
        lens[j] = t.numel()
    all_meta = torch.empty(world * (per_rank + 1), dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(all_meta, lens, group=group)
    all_meta = all_meta.view(world, per_rank + 1)
    codes = sorted({int(c) for c in all_meta[:, per_rank].tolist() if int(c) >= 0})
    assert len(codes) <= 1, f"ranks disagree on the payload dtype: {[_DTYPE_CODES[c] for c in codes]}"
    dtype = _DTYPE_CODES[codes[0]] if codes else torch.float32
    all_lens = all_meta[:, :per_rank].reshape(-1)
    width = int(all_lens.max().item()) if n_items else 0
    payload = torch.zeros(per_rank, max(width, 1), dtype=dtype, device=dev)
    for j, t in enumerate(local):
        payload[j, : t.numel()] = t.to(dev)
    gathered = torch.empty(world * per_rank, max(width, 1), dtype=dtype, device=dev)
    dist.all_gather_into_tensor(gathered, payload, group=group)
    out: List[Optional[torch.Tensor]] = [None] * n_items
    for r in range(world):
        for j, i in enumerate(shard_indices(n_items, r, world)):
            out[i] = gathered[r * per_rank + j, : int(all_lens[r * per_rank + j])].clone()
    return out  # type: ignore[return-value]


def run_sharded(items: Sequence, fn: Callable[[int, object], torch.Tensor], group: Optional[dist.ProcessGroup] = None,
                device: Optional[torch.device] = None) -> List[torch.Tensor]:
    """`fn(index, item) -> 1-D tensor` on this rank's shard, then `gather_variable`.  `fn` gets the GLOBAL index so that
    per-utterance seeds do not depend on the number of ranks."""
    n = len(items)
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    local = [fn(i, items[i]) for i in shard_indices(n, rank, world)]
    return gather_variable(local, n, group=group, device=device)
