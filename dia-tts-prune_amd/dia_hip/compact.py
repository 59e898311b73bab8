"""Structured-pruned checkpoint -> physically smaller tensors (BASELINE config 4).

``offline_prune.py --prune-mode structured --prune-dim 0`` (reference offline_prune.py:82-156 ->
dia/pruning_utils.py:64-151) leaves a same-shape checkpoint in which whole dim-0 slices of every
DenseGeneral kernel are zero: input features of q/k/v/wi_fused/wo/logits, and whole heads of
o_proj.  The checkpoint carries no mask (``prune.remove``, pruning_utils.py:145), so the structure is
recovered from the zeros.  Compaction then

* drops the zero rows of every matrix (K-compaction): the consumer's activations are emitted by their
  producer directly in the compacted order (``cmap``), so nothing is gathered at run time;
* propagates dead structure: heads zeroed in ``o_proj`` make the matching q columns (and a kv head
  whose every query head is gone) unnecessary, hidden units zeroed in ``wo`` make the matching gate/up
  columns of ``wi_fused`` unnecessary (N-compaction, ``strip_map`` / ``head_map``).

The encoder (prefill only) is compacted the same way (``plan_encoder_layer``: a dead head drops its q, k and v
columns); the cross K/V projections keep their full K because every decoder layer has its own keep set over the
one encoder output.

A zero row contributes exactly 0.0 to an fp32 dot product, so results equal the zero-streaming path up
to summation order.  q, k and v share one activation plane set, so their K-compaction uses the union
of the three keep sets (rows zero in only one of them are stored as zeros).
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch


def nonzero_rows(w2d: torch.Tensor) -> torch.Tensor:
    """bool [K]: rows of a [K, N] matrix that are not identically zero."""
    return (w2d != 0).any(dim=1)


def _cmap(keep: torch.Tensor) -> torch.Tensor:
    """bool [D] -> int32 [D]: compact position of every kept column, -1 for dropped ones."""
    idx = torch.cumsum(keep.to(torch.int32), dim=0) - 1
    return torch.where(keep, idx, torch.full_like(idx, -1)).to(torch.int32)


K_GRANULE = 256      # rows: 8 k-tiles of 32, the per-wave K split of the 8-wave GEMV kernels


def _granule(n: int) -> int:
    """full-size matrices (K >= 2048): whole per-wave k-tile groups; smaller test models: whole k-tiles"""
    return K_GRANULE if n >= 2048 and n % K_GRANULE == 0 else 32


def pad_keep(keep: torch.Tensor, granule: int = 0) -> torch.Tensor:
    """bool [D] -> bool [D] with the kept count rounded up to a multiple of `granule` by re-adding the
    first dropped rows.  Those rows are zero in the checkpoint, so they change nothing but the shape: the
    fast kernels want K in whole per-wave k-tile groups (a union of q/k/v keep sets is ~1792 of 2048)."""
    granule = granule or _granule(keep.numel())
    n = int(keep.sum())
    pad = (-n) % granule
    if pad == 0 or keep.numel() % granule != 0:
        return keep
    dropped = torch.nonzero(~keep).flatten()[:pad]
    out = keep.clone()
    out[dropped] = True
    return out


@dataclass
class LayerPlan:
    keep_qkv: torch.Tensor            # bool [D]  union of q/k/v input rows
    keep_cq: torch.Tensor             # bool [D]
    keep_wi: torch.Tensor             # bool [D]
    live_q_heads: torch.Tensor        # bool [q_heads]      (from self o_proj)
    live_kv_heads: torch.Tensor       # bool [kv_heads]
    live_c_heads: torch.Tensor        # bool [cq_heads]     (from cross o_proj)
    live_hidden: torch.Tensor         # bool [F]            (from wo)
    keep_ckv: torch.Tensor            # bool [E] union of cross k/v input rows (prefill)


def plan_decoder_layer(sd: Dict[str, torch.Tensor], prefix: str, q_heads: int, kv_heads: int, cq_heads: int) -> LayerPlan:
    g = lambda n: sd[prefix + n]
    D = g("self_attention.q_proj.weight").shape[0]
    rows = lambda n: nonzero_rows(g(n).reshape(g(n).shape[0], -1))
    keep_qkv = rows("self_attention.q_proj.weight") | rows("self_attention.k_proj.weight") | rows("self_attention.v_proj.weight")
    live_q = nonzero_rows(g("self_attention.o_proj.weight").reshape(q_heads, -1))
    grp = q_heads // kv_heads
    live_kv = live_q.reshape(kv_heads, grp).any(dim=1)
    live_c = nonzero_rows(g("cross_attention.o_proj.weight").reshape(cq_heads, -1))
    keep_ckv = rows("cross_attention.k_proj.weight") | rows("cross_attention.v_proj.weight")
    return LayerPlan(pad_keep(keep_qkv), pad_keep(rows("cross_attention.q_proj.weight")), pad_keep(rows("mlp.wi_fused.weight")),
                     live_q, live_kv, live_c, rows("mlp.wo.weight"), keep_ckv)


@dataclass
class EncLayerPlan:
    keep_qkv: torch.Tensor            # bool [E]  union of q/k/v input rows (padded)
    keep_wi: torch.Tensor             # bool [E]
    live_heads: torch.Tensor          # bool [heads]   (from o_proj: a dead head drops its q, k AND v columns — MHA)
    live_hidden: torch.Tensor         # bool [F]       (from wo)


def plan_encoder_layer(sd: Dict[str, torch.Tensor], prefix: str, heads: int) -> EncLayerPlan:
    g = lambda n: sd[prefix + n]
    rows = lambda n: nonzero_rows(g(n).reshape(g(n).shape[0], -1))
    keep_qkv = rows("self_attention.q_proj.weight") | rows("self_attention.k_proj.weight") | rows("self_attention.v_proj.weight")
    live = nonzero_rows(g("self_attention.o_proj.weight").reshape(heads, -1))
    return EncLayerPlan(pad_keep(keep_qkv), pad_keep(rows("mlp.wi_fused.weight")), live, rows("mlp.wo.weight"))


def enc_is_pruned(plan: EncLayerPlan) -> bool:
    return not bool(plan.keep_qkv.all() and plan.keep_wi.all() and plan.live_heads.all() and plan.live_hidden.all())


def is_pruned(plan: LayerPlan) -> bool:
    return not bool(plan.keep_qkv.all() and plan.keep_cq.all() and plan.keep_wi.all() and plan.live_q_heads.all()
                    and plan.live_c_heads.all() and plan.live_hidden.all())


def head_map(live: torch.Tensor) -> torch.Tensor:
    return _cmap(live)


def strips_of_heads(live_heads: torch.Tensor, head_offset_cols: int = 0) -> List[int]:
    """original 16-column strip indices covered by the live 128-wide heads, in order"""
    out: List[int] = []
    for h in torch.nonzero(live_heads).flatten().tolist():
        base = (head_offset_cols + h * 128) // 16
        out.extend(range(base, base + 8))
    return out


def pad_hidden_keep(live_hidden: torch.Tensor) -> torch.Tensor:
    """indices of the live hidden units (int64), their count padded up to a multiple of K_GRANULE (8 for toy sizes) with -1
    (zero gate/up columns in wi, zero rows in wo)"""
    idx = torch.nonzero(live_hidden).flatten()
    # full-size models: whole multiples of 1024 hidden units = 32 k-tiles of wo's K, which every split-K form of
    # the fast kernels divides (a 30 %-pruned 5734 -> 6144; 5888 would fall to the generic kernel); small
    # models: whole 16-column strips only
    granule = 4 * K_GRANULE if (live_hidden.numel() >= 4096 and live_hidden.numel() % (4 * K_GRANULE) == 0) else \
        (K_GRANULE if (live_hidden.numel() >= 2048 and live_hidden.numel() % K_GRANULE == 0) else 8)
    pad = (-idx.numel()) % granule
    if pad:
        idx = torch.cat([idx, torch.full((pad,), -1, dtype=idx.dtype, device=idx.device)])
    return idx


def pad_rows(w2d: torch.Tensor, granule: int = K_GRANULE) -> torch.Tensor:
    """zero rows appended up to a multiple of `granule`: the K of o_proj after dead heads are dropped is
    live_heads * 128 (11 heads -> 1408 rows = 44 k-tiles, which no fast kernel divides).  The extra rows are ZERO
    weights; the activation-plane columns they meet are not written by this layer's attention (planes_a is shared by
    self- and cross-attention of every layer, so they hold whatever a layer with more live heads left there): finite
    stale values times 0.0 add exactly nothing.  Attention outputs are convex combinations of finite V rows, so the
    planes never hold inf / nan; tests/test_gpu_kernels.py::test_padded_o_rows_ignore_stale_planes pins it."""
    pad = (-w2d.shape[0]) % granule
    if pad == 0:
        return w2d
    return torch.cat([w2d, torch.zeros(pad, w2d.shape[1], dtype=w2d.dtype, device=w2d.device)], dim=0)


def take_rows(w2d: torch.Tensor, keep: torch.Tensor) -> torch.Tensor:
    return w2d[keep]


def take_cols_idx(w2d: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """columns by index; index -1 -> a zero column"""
    safe = idx.clamp(min=0)
    out = w2d[:, safe]
    if (idx < 0).any():
        out = out.clone()
        out[:, idx < 0] = 0
    return out
