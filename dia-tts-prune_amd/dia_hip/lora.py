"""Offline LoRA merge: adapters are folded into the dense kernels before tiling, so the
decode path streams exactly the same bytes as without adapters.

The reference attaches adapters at run time through PEFT (cli.py:166-174, model.py:598-625), which is not
installed here and which, as far as its Linear-only LoRA layers go, cannot wrap this model's DenseGeneral
modules at all — parity for this file is therefore UNPINNED; it follows PEFT's published LoRA arithmetic
for a Linear layer: ``W[out, in] += (lora_alpha / r) * B[out, r] @ A[r, in]``.  A DenseGeneral kernel is
stored ``[in..., out...]`` (layers.py:47-51), i.e. the transpose of that W once both sides are flattened.

Adapter directory layout (PEFT's): ``adapter_config.json`` with ``r``, ``lora_alpha`` (+ optional
``rank_pattern`` / ``alpha_pattern``, ``fan_in_fan_out``) and ``adapter_model.safetensors`` or
``adapter_model.bin`` with keys ``<prefix>.<module path>.lora_A.weight`` / ``.lora_B.weight`` (an optional
adapter name segment such as ``.default`` is accepted).
"""

from __future__ import annotations

import json
import os
import re
from typing import Dict

import torch

_KEY = re.compile(r"^(?:base_model\.model\.)?(?P<mod>.+?)\.lora_(?P<ab>[AB])(?:\.[^.]+)?\.weight$")


def load_adapter(adapter_dir: str):
    cfg_path = os.path.join(adapter_dir, "adapter_config.json")
    if not os.path.isfile(cfg_path):
        raise FileNotFoundError(f"adapter_config.json not found in {adapter_dir}")
    with open(cfg_path) as f:
        acfg = json.load(f)
    st = os.path.join(adapter_dir, "adapter_model.safetensors")
    bn = os.path.join(adapter_dir, "adapter_model.bin")
    if os.path.isfile(st):
        from safetensors.torch import load_file
        tensors = load_file(st)
    elif os.path.isfile(bn):
        tensors = torch.load(bn, map_location="cpu", weights_only=True)
    else:
        raise FileNotFoundError(f"adapter_model.safetensors / adapter_model.bin not found in {adapter_dir}")
    return acfg, tensors


def merge_lora_state_dict(sd: Dict[str, torch.Tensor], adapter_dir: str) -> Dict[str, torch.Tensor]:
    """Returns a new state dict with every adapted kernel replaced by ``kernel + scale * (B @ A)^T`` (fp32
    arithmetic, original dtype and shape kept).  Unknown modules raise: a silently ignored adapter would
    look like a working one."""
    acfg, tensors = load_adapter(adapter_dir)
    r_default, alpha_default = int(acfg.get("r", 8)), float(acfg.get("lora_alpha", 8))
    rank_pat, alpha_pat = acfg.get("rank_pattern") or {}, acfg.get("alpha_pattern") or {}
    pairs: Dict[str, Dict[str, torch.Tensor]] = {}
    for k, v in tensors.items():
        m = _KEY.match(k)
        if not m:
            continue
        pairs.setdefault(m.group("mod"), {})[m.group("ab")] = v
    if not pairs:
        raise RuntimeError(f"no lora_A / lora_B tensors found in {adapter_dir}")
    out = dict(sd)
    for mod, ab in pairs.items():
        if "A" not in ab or "B" not in ab:
            raise RuntimeError(f"adapter for {mod} lacks lora_{'B' if 'A' in ab else 'A'}")
        name = mod + ".weight"
        if name not in sd:
            raise RuntimeError(f"adapter targets {mod}, which is not a module of this model")
        A, B = ab["A"].float(), ab["B"].float()                      # [r, in], [out, r]
        r = A.shape[0]
        leaf = mod.rsplit(".", 1)[-1]
        alpha = float(next((v for p, v in alpha_pat.items() if mod.endswith(p) or leaf == p), alpha_default))
        r_cfg = int(next((v for p, v in rank_pat.items() if mod.endswith(p) or leaf == p), r_default))
        if r_cfg != r:
            raise RuntimeError(f"{mod}: adapter rank {r} does not match adapter_config ({r_cfg})")
        w = sd[name]
        delta = (B @ A) * (alpha / r)                                # [out, in]
        if acfg.get("fan_in_fan_out"):
            delta = delta.t()
        n_in, n_out = A.shape[1], B.shape[0]
        if n_in * n_out != w.numel():           # kernel is [in..., out...] (layers.py:47-51): both sides flatten
            raise RuntimeError(f"{mod}: adapter shapes A{tuple(A.shape)} B{tuple(B.shape)} do not fit kernel {tuple(w.shape)}")
        flat = w.float().reshape(n_in, n_out)
        out[name] = (flat + delta.t()).reshape(w.shape).to(w.dtype)
    return out
