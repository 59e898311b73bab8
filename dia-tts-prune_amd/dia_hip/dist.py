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
    """Every device tensor of a DeviceWeights, in a deterministic order."""
    out: List[torch.Tensor] = [w.enc_emb]
    for L in w.enc_layers:
        out += [L["g_sa"], L["g_mlp"], L["qkv"].t, L["o"].t, L["wi"].t, L["wo"].t]
    out += [w.enc_norm, w.dec_emb]
    for L in w.dec_layers:
        out += [L["g_sa"], L["g_ca"], L["g_mlp"]] + [L[k].t for k in ("qkv", "o", "cq", "co", "ckv", "wi", "wo")]
        out += [L[k] for k in ("cmap_ca", "cmap_mlp", "cmap_next", "smap_qkv", "smap_cq", "smap_ckv", "hmap_self", "hmap_cross")
                if L[k] is not None]
    out += [w.dec_norm, w.logits.t, w.cos_t, w.sin_t]
    if w.cmap_first is not None:
        out.append(w.cmap_first)
    return out


def broadcast_tensors(tensors: Iterable[torch.Tensor], src: int = 0, group=None) -> int:
    """In-place broadcast; 16-bit payloads travel as raw bytes so every backend (RCCL, gloo) can carry them.
    Returns the number of bytes moved."""
    import torch.distributed as dist

    n = 0
    for t in tensors:
        buf = t.view(torch.uint8) if t.dtype in (torch.bfloat16, torch.float16) else t
        dist.broadcast(buf, src=src, group=group)
        n += t.numel() * t.element_size()
    return n


def broadcast_weights(w, src: int = 0, group=None) -> int:
    if getattr(w, "compacted", False):
        raise ValueError("compacted (structured-pruned) weights have checkpoint-dependent shapes: "
                         "load the checkpoint on every rank instead of broadcasting")
    return broadcast_tensors(weight_tensors(w), src=src, group=group)


def gather_token_buffers(tokens: torch.Tensor, world: int, group=None) -> List[torch.Tensor]:
    """all_gather of the per-rank int32 [B_local, T, C] token buffers (equal B_local on every rank)."""
    import torch.distributed as dist

    outs = [torch.empty_like(tokens) for _ in range(world)]
    dist.all_gather(outs, tokens.contiguous(), group=group)
    return outs
