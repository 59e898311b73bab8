"""Data-parallel decode across the GPUs of one node.

Utterances are independent: each rank owns a shard of the batch and runs its own
decode loop; there is NO per-step collective.  The only exchange is one broadcast of the repacked
weights from the rank that loaded the checkpoint (RCCL over xGMI when the backend is "nccl"), and an
optional gather of the small int32 token buffers at the end.
"""

from __future__ import annotations

from typing import Iterable, List, Sequence

import torch


def shard_utterances(n: int, world: int, rank: int) -> List[int]:
    """utterance u -> rank u mod world (both CFG rows of an utterance stay on one GPU)."""
    return [u for u in range(n) if u % world == rank]


def weight_tensors(w) -> List[torch.Tensor]:
    """Every device tensor of a DeviceWeights, in the order of its flat arena."""
    return w.tensors()


def broadcast_tensors(tensors: Iterable[torch.Tensor], src: int = 0, group=None) -> int:
    """In-place broadcast, one collective per tensor; 16-bit payloads travel as raw bytes so every backend (RCCL,
    gloo) can carry them.  Returns the number of bytes moved.  (Generic helper; the weights go as ONE buffer.)"""
    import torch.distributed as dist

    n = 0
    for t in tensors:
        buf = t.view(torch.uint8) if t.dtype in (torch.bfloat16, torch.float16) else t
        dist.broadcast(buf, src=src, group=group)
        n += t.numel() * t.element_size()
    return n


def broadcast_weights(w, src: int = 0, group=None) -> int:
    """ONE broadcast of the model's flat arena (DeviceWeights.flat: every repacked tensor, 3.3 GB for Dia-1.6B) from the
    rank that loaded the checkpoint — `ncclBroadcast` over xGMI when the backend is "nccl".  The receiving ranks hold
    a DeviceWeights.empty_like_config() of the same config: same tensor order, same offsets."""
    import torch.distributed as dist

    if getattr(w, "compacted", False):
        raise ValueError("compacted (structured-pruned) weights have checkpoint-dependent shapes: "
                         "load the checkpoint on every rank instead of broadcasting")
    if w.flat is None:
        raise ValueError("DeviceWeights without a flat arena")
    dist.broadcast(w.flat, src=src, group=group)
    return int(w.flat.numel())


def gather_token_buffers(tokens: torch.Tensor, world: int, group=None) -> List[torch.Tensor]:
    """all_gather of the per-rank int32 [B_local, T, C] token buffers (equal B_local on every rank)."""
    import torch.distributed as dist

    outs = [torch.empty_like(tokens) for _ in range(world)]
    dist.all_gather(outs, tokens.contiguous(), group=group)
    return outs


def gather_utterances(local: Sequence[torch.Tensor], n_total: int, world: int, rank: int, group=None) -> List[torch.Tensor]:
    """Per-rank results (one int32 [T, C] token buffer per owned utterance, in shard order) -> the list of all
    `n_total` buffers in utterance order on every rank.  Shards may be ragged (5 utterances over 2 ranks = 3 + 2):
    every rank pads its stack to the largest shard before the all_gather."""
    import torch.distributed as dist

    per = (n_total + world - 1) // world
    mine = shard_utterances(n_total, world, rank)
    if len(local) != len(mine):
        raise ValueError(f"rank {rank} owns {len(mine)} utterances, got {len(local)} results")
    proto = None
    for t in local:
        proto = t
        break
    shape = torch.tensor(list(proto.shape) if proto is not None else [0, 0], dtype=torch.int64)
    shapes = [torch.zeros_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape, group=group)
    T, Cc = (int(v) for v in torch.stack(shapes).max(dim=0).values)
    dev = proto.device if proto is not None else torch.device("cpu")
    stack = torch.full((per, T, Cc), -1, dtype=torch.int32, device=dev)
    for i, t in enumerate(local):
        stack[i] = t
    outs = [torch.empty_like(stack) for _ in range(world)]
    dist.all_gather(outs, stack, group=group)
    res: List[torch.Tensor] = [None] * n_total          # type: ignore[list-item]
    for r in range(world):
        for i, u in enumerate(shard_utterances(n_total, world, r)):
            res[u] = outs[r][i]
    return res
