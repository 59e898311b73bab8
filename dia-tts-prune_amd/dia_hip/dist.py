"""Data-parallel decode across the GPUs of one node.

Utterances are independent: each rank owns a shard of the batch and runs its own
decode loop; there is NO per-step collective.  The only exchange is one broadcast of the repacked
weights from the rank that loaded the checkpoint (RCCL over xGMI when the backend is "nccl"), and an
optional gather of the small int32 token buffers at the end.
"""

from __future__ import annotations

from typing import Iterable, List, Optional, Sequence

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


def _arena_signature(w) -> List[int]:
    """what must agree on every rank before the flat arena may travel: its size in bytes and the layout switches that
    decide the tensor order inside it"""
    return [int(w.flat.numel()) if w.flat is not None else -1, 1 if getattr(w, "compacted", False) else 0,
            int(getattr(w, "weight_planes", 1))]


def broadcast_weights(w, src: int = 0, group=None) -> int:
    """ONE broadcast of the model's flat arena (DeviceWeights.flat: every repacked tensor, 3.3 GB for Dia-1.6B) from the
    rank that loaded the checkpoint — `ncclBroadcast` over xGMI when the backend is "nccl".  The receiving ranks hold
    a DeviceWeights.empty_like_config() of the same config: same tensor order, same offsets.

    Every rank first takes part in one MIN and one MAX all-reduce of (arena bytes, compacted, weight planes): a rank
    whose arena differs — a compacted (structured-pruned) model has checkpoint-dependent shapes, three weight planes
    triple the arena — makes EVERY rank raise, instead of one rank raising alone while the others wait in the broadcast."""
    import torch.distributed as dist

    if w.flat is None:
        raise ValueError("DeviceWeights without a flat arena")
    sig = torch.tensor(_arena_signature(w), dtype=torch.int64, device=w.flat.device)
    lo, hi = sig.clone(), sig.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=group)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=group)
    lo, hi = lo.tolist(), hi.tolist()
    if hi[1] != 0:
        raise ValueError("compacted (structured-pruned) weights have checkpoint-dependent shapes: "
                         "load the checkpoint on every rank instead of broadcasting")
    if lo != hi:
        raise ValueError(f"weight arenas differ across ranks (bytes, compacted, weight planes): min {lo}, max {hi}; "
                         "build the receivers with DeviceWeights.empty_like_config(cfg, device, weight_planes=...)")
    dist.broadcast(w.flat, src=src, group=group)
    return int(w.flat.numel())


def gather_token_buffers(tokens: torch.Tensor, world: int, group=None) -> List[torch.Tensor]:
    """all_gather of the per-rank int32 [B_local, T, C] token buffers (equal B_local on every rank)."""
    import torch.distributed as dist

    outs = [torch.empty_like(tokens) for _ in range(world)]
    dist.all_gather(outs, tokens.contiguous(), group=group)
    return outs


def gather_utterances(local: Sequence[torch.Tensor], n_total: int, world: int, rank: int, group=None,
                      device: Optional[torch.device] = None) -> List[torch.Tensor]:
    """Per-rank results (one int32 [T, C] token buffer per owned utterance, in shard order) -> the list of all
    `n_total` buffers in utterance order on every rank.  Shards may be ragged (5 utterances over 2 ranks = 3 + 2) and a
    rank may own nothing (n_total < world): every rank pads its stack to the largest shard and to the largest [T, C]
    before the all_gather; a returned buffer is cut back to the shape its owner sent.

    `device` is where the collective's tensors live — the local GPU under the "nccl" (RCCL) backend, which cannot carry
    CPU tensors; default: the device of the first local buffer, which a rank without utterances cannot know and must pass."""
    import torch.distributed as dist

    per = (n_total + world - 1) // world
    mine = shard_utterances(n_total, world, rank)
    if len(local) != len(mine):
        raise ValueError(f"rank {rank} owns {len(mine)} utterances, got {len(local)} results")
    for t in local:
        if t.dim() != 2:
            raise ValueError("token buffers are [T, C]")
    if device is None:
        if not local:
            raise ValueError("a rank that owns no utterance must say on which device the collective runs (device=...)")
        device = local[0].device
    shp = torch.zeros((per, 2), dtype=torch.int64, device=device)          # [T, C] of every owned buffer, 0 where none
    for i, t in enumerate(local):
        shp[i, 0], shp[i, 1] = t.shape[0], t.shape[1]
    shapes = [torch.zeros_like(shp) for _ in range(world)]
    dist.all_gather(shapes, shp, group=group)
    allshp = torch.stack(shapes).cpu()
    T, Cc = int(allshp[..., 0].max()), int(allshp[..., 1].max())
    stack = torch.full((per, T, Cc), -1, dtype=torch.int32, device=device)
    for i, t in enumerate(local):
        stack[i, : t.shape[0], : t.shape[1]] = t.to(device=device, dtype=torch.int32)
    outs = [torch.empty_like(stack) for _ in range(world)]
    dist.all_gather(outs, stack, group=group)
    res: List[torch.Tensor] = [None] * n_total          # type: ignore[list-item]
    for r in range(world):
        for i, u in enumerate(shard_utterances(n_total, world, r)):
            res[u] = outs[r][i, : int(allshp[r, i, 0]), : int(allshp[r, i, 1])]
    return res
