"""Structured pruning of a Dia checkpoint and detection of pruned structure for repacking.

``structured_prune_state_dict`` restates what the reference's ``offline_prune.py`` does in
``--prune-mode structured`` (offline_prune.py:82-156 -> dia/pruning_utils.py:64-151): for every
DenseGeneral kernel, rank the slices along ``dim`` by their L_n norm and zero the
``round(amount * n_slices)`` weakest (``torch.nn.utils.prune.ln_structured`` followed by
``prune.remove``) — the result is a same-shape checkpoint with zero slices and no mask.

``kept_slices`` recovers the structure from the zeros (the checkpoint carries no mask,
pruning_utils.py:145), which is what the loader uses to build physically smaller tensors.
"""

from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch

from .config import DiaConfig
from .weights import param_shapes


def prunable_names(cfg: DiaConfig):
    """DenseGeneral kernels = everything the reference's get_prunable_modules finds in DiaModel
    (pruning_utils.py:13-40): projections, MLP matrices and the logits head; embeddings and RMSNorm
    weights are not DenseGeneral/Linear/Conv1d and stay untouched."""
    return [k for k in param_shapes(cfg) if not (k.endswith("norm.weight") or "embedding" in k)]


def structured_prune_state_dict(cfg: DiaConfig, sd: Dict[str, torch.Tensor], amount: float, dim: int = 0,
                                n: int = 2) -> Tuple["OrderedDict[str, torch.Tensor]", Dict[str, np.ndarray]]:
    """Returns (pruned state_dict, {name: kept slice indices along `dim`})."""
    if not (0.0 < amount < 1.0):
        raise ValueError("--prune-amount must be between 0.0 and 1.0 (exclusive).")   # offline_prune.py:58-60
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict((k, v.clone()) for k, v in sd.items())
    kept: Dict[str, np.ndarray] = {}
    for name in prunable_names(cfg):
        w = out[name].float()
        if dim >= w.dim():                      # pruning_utils.py:84-87: module skipped
            continue
        size = w.shape[dim]
        n_prune = int(round(amount * size))
        n_keep = size - n_prune
        other = [a for a in range(w.dim()) if a != dim]
        norm = torch.norm(w, p=n, dim=other)
        keep_idx = torch.topk(norm, k=n_keep, largest=True).indices
        mask = torch.zeros(size, dtype=torch.bool, device=w.device)
        mask[keep_idx] = True
        shape = [1] * w.dim()
        shape[dim] = size
        out[name] = (w * mask.view(shape)).to(sd[name].dtype)
        kept[name] = torch.nonzero(mask).flatten().cpu().numpy().astype(np.int32)
    return out, kept


def unstructured_prune_state_dict(cfg: DiaConfig, sd: Dict[str, torch.Tensor], amount: float) -> "OrderedDict[str, torch.Tensor]":
    """``offline_prune.py --prune-mode unstructured`` (offline_prune.py:101-103 -> pruning_utils.py:42-62):
    GLOBAL L1 magnitude pruning over all DenseGeneral kernels — the ``round(amount * N)`` entries of smallest
    absolute value across the concatenation of every prunable kernel (``prune.global_unstructured`` with
    ``L1Unstructured``: top-k of -|w| over the flattened concatenation, in module order) are zeroed.  The result
    has no exploitable structure: it loads as dense tensors whose zeros stream like any other value."""
    if not (0.0 < amount < 1.0):
        raise ValueError("--prune-amount must be between 0.0 and 1.0 (exclusive).")   # offline_prune.py:58-60
    names = prunable_names(cfg)
    flat = torch.cat([sd[k].detach().float().reshape(-1) for k in names])
    n_prune = int(round(amount * flat.numel()))
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict((k, v.clone()) for k, v in sd.items())
    if n_prune == 0:
        return out
    idx = torch.topk(flat.abs(), k=n_prune, largest=False).indices
    mask = torch.ones_like(flat, dtype=torch.bool)
    mask[idx] = False
    off = 0
    for k in names:
        n = sd[k].numel()
        out[k] = (sd[k].float() * mask[off: off + n].reshape(sd[k].shape)).to(sd[k].dtype)
        off += n
    return out


def kept_slices(w: torch.Tensor, dim: int = 0) -> np.ndarray:
    """Indices along `dim` whose slice is not identically zero."""
    moved = w.movedim(dim, 0).reshape(w.shape[dim], -1)
    return torch.nonzero((moved != 0).any(dim=1)).flatten().cpu().numpy().astype(np.int32)


def sparsity(cfg: DiaConfig, sd: Dict[str, torch.Tensor]) -> float:
    """Global zero fraction over the prunable kernels (pruning_utils.py:153-179)."""
    tot = zero = 0
    for k in prunable_names(cfg):
        tot += sd[k].numel()
        zero += int((sd[k] == 0).sum().item())
    return zero / max(tot, 1)
