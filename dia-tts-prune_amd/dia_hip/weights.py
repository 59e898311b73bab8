"""Checkpoint-side helpers: parameter inventory, synthetic weights, checkpoint IO.

Parameter names and shapes are those of the reference module tree
(``dia/layers.py``: Encoder 419-441, EncoderLayer 349-383, Decoder 587-628,
DecoderLayer 465-528, Attention 203-227, MlpBlock 76-90) so that a reference
``state_dict`` (``pytorch_model.bin`` written by ``offline_prune.py:153`` or
``model.safetensors`` written by the hub mixin, layers.py:769-784) loads here
unchanged.  DenseGeneral kernels are stored ``in_shapes + out_features``
(layers.py:47-51), i.e. contraction axes first.
"""

from __future__ import annotations

import json
import os
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch

from .config import DiaConfig


def param_shapes(cfg: DiaConfig) -> "OrderedDict[str, Tuple[int, ...]]":
    m, e, d = cfg.model, cfg.model.encoder, cfg.model.decoder
    s: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    s["encoder.embedding.weight"] = (m.src_vocab_size, e.n_embd)
    for i in range(e.n_layer):
        p = f"encoder.layers.{i}."
        s[p + "pre_sa_norm.weight"] = (e.n_embd,)
        s[p + "self_attention.q_proj.weight"] = (e.n_embd, e.n_head, e.head_dim)
        s[p + "self_attention.k_proj.weight"] = (e.n_embd, e.n_head, e.head_dim)
        s[p + "self_attention.v_proj.weight"] = (e.n_embd, e.n_head, e.head_dim)
        s[p + "self_attention.o_proj.weight"] = (e.n_head, e.head_dim, e.n_embd)
        s[p + "post_sa_norm.weight"] = (e.n_embd,)
        s[p + "mlp.wi_fused.weight"] = (e.n_embd, 2, e.n_hidden)
        s[p + "mlp.wo.weight"] = (e.n_hidden, e.n_embd)
    s["encoder.norm.weight"] = (e.n_embd,)
    for c in range(cfg.data.channels):
        s[f"decoder.embeddings.{c}.weight"] = (m.tgt_vocab_size, d.n_embd)
    for i in range(d.n_layer):
        p = f"decoder.layers.{i}."
        s[p + "pre_sa_norm.weight"] = (d.n_embd,)
        s[p + "pre_ca_norm.weight"] = (d.n_embd,)
        s[p + "pre_mlp_norm.weight"] = (d.n_embd,)
        s[p + "self_attention.q_proj.weight"] = (d.n_embd, d.gqa_query_heads, d.gqa_head_dim)
        s[p + "self_attention.k_proj.weight"] = (d.n_embd, d.kv_heads, d.gqa_head_dim)
        s[p + "self_attention.v_proj.weight"] = (d.n_embd, d.kv_heads, d.gqa_head_dim)
        s[p + "self_attention.o_proj.weight"] = (d.gqa_query_heads, d.gqa_head_dim, d.n_embd)
        s[p + "cross_attention.q_proj.weight"] = (d.n_embd, d.cross_query_heads, d.cross_head_dim)
        s[p + "cross_attention.k_proj.weight"] = (e.n_embd, d.cross_query_heads, d.cross_head_dim)
        s[p + "cross_attention.v_proj.weight"] = (e.n_embd, d.cross_query_heads, d.cross_head_dim)
        s[p + "cross_attention.o_proj.weight"] = (d.cross_query_heads, d.cross_head_dim, d.n_embd)
        s[p + "mlp.wi_fused.weight"] = (d.n_embd, 2, d.n_hidden)
        s[p + "mlp.wo.weight"] = (d.n_hidden, d.n_embd)
    s["decoder.norm.weight"] = (d.n_embd,)
    s["decoder.logits_dense.weight"] = (d.n_embd, cfg.data.channels, m.tgt_vocab_size)
    return s


def param_count(cfg: DiaConfig) -> int:
    n = 0
    for shp in param_shapes(cfg).values():
        k = 1
        for v in shp:
            k *= v
        n += k
    return n


# ----------------------------------------------------------------------------
# Synthetic weights (no Dia checkpoint exists offline).
#
# A counter-based generator in integer arithmetic only, so the same values come
# out on any device, in any order, in any chunking: element i of tensor `name`
# is the sum of four 16-bit fields of two 32-bit hashes (Irwin-Hall, n=4 — a
# bell-shaped approximation of a normal with support ±3.46 sigma), scaled to
# `std` and rounded to the nearest bf16-representable value.  RMSNorm weights
# are 1.0.
# ----------------------------------------------------------------------------

_M32 = 0xFFFFFFFF


def _mix32(x: torch.Tensor) -> torch.Tensor:
    x = x & _M32
    x = ((x ^ (x >> 16)) * 0x7FEB352D) & _M32
    x = ((x ^ (x >> 15)) * 0x846CA68B) & _M32
    return x ^ (x >> 16)


def _name_seed(name: str, seed: int) -> int:
    h = (seed * 0x9E3779B1 + 0x7F4A7C15) & _M32
    for ch in name.encode("utf-8"):
        h = ((h ^ ch) * 0x01000193) & _M32  # FNV-1a step
    return h


def synthetic_tensor(name: str, shape: Tuple[int, ...], seed: int, std: float,
                     device: torch.device | str = "cpu", chunk: int = 1 << 24) -> torch.Tensor:
    """fp32 tensor whose every value is bf16-representable."""
    n = 1
    for v in shape:
        n *= v
    out = torch.empty(n, dtype=torch.float32, device=device)
    base = _name_seed(name, seed)
    scale = torch.tensor(std / (65536.0 * (1.0 / 3.0) ** 0.5), dtype=torch.float32, device=device)
    for lo in range(0, n, chunk):
        hi = min(n, lo + chunk)
        i = torch.arange(lo, hi, dtype=torch.int64, device=device)
        a = _mix32(i * 2 + base)
        b = _mix32(i * 2 + 1 + (base ^ 0x5BD1E995))
        s = (a & 0xFFFF) + (a >> 16) + (b & 0xFFFF) + (b >> 16)
        z = (s - 131070).to(torch.float32) * scale
        out[lo:hi] = z.to(torch.bfloat16).to(torch.float32)
    return out.view(shape)


def synthetic_state_dict(cfg: DiaConfig, seed: int = 1234, std: float = 0.02,
                         device: torch.device | str = "cpu") -> "OrderedDict[str, torch.Tensor]":
    sd: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    for name, shape in param_shapes(cfg).items():
        if name.endswith("norm.weight"):
            sd[name] = torch.ones(shape, dtype=torch.float32, device=device)
        else:
            sd[name] = synthetic_tensor(name, shape, seed, std, device)
    return sd


# ----------------------------------------------------------------------------
# Checkpoint IO — both formats the reference reads
# ----------------------------------------------------------------------------

def load_state_dict_file(path: str) -> Dict[str, torch.Tensor]:
    """``.safetensors`` (hub-mixin directory format) or a pickled ``torch.save`` state_dict
    (reference model.py:169); keys containing ``lora_`` are dropped (model.py:172)."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file

        sd = load_file(path, device="cpu")
    else:
        sd = torch.load(path, map_location="cpu")
    return {k: v for k, v in sd.items() if "lora_" not in k}


def find_checkpoint_in_dir(model_dir: str) -> Tuple[str, str]:
    cfg = os.path.join(model_dir, "config.json")
    for cand in ("model.safetensors", "pytorch_model.bin", "dia-v0_1.pth"):
        p = os.path.join(model_dir, cand)
        if os.path.isfile(p):
            return cfg, p
    raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin in {model_dir}")


def read_hub_config(config_path: str) -> DiaConfig:
    """The hub mixin stores ``{"config": {...DiaConfig...}}`` (layers.py:778-783 coders);
    ``DiaConfig.save`` stores the bare config.  Accept both."""
    with open(config_path, "r", encoding="utf-8") as f:
        d = json.load(f)
    if "model" not in d and "config" in d:
        d = d["config"]
    return DiaConfig.model_validate(d)


def check_state_dict(cfg: DiaConfig, sd: Dict[str, torch.Tensor]) -> Tuple[list, list]:
    """(missing, unexpected) key lists, like ``load_state_dict(strict=False)`` (model.py:173)."""
    want = param_shapes(cfg)
    missing = [k for k in want if k not in sd]
    unexpected = [k for k in sd if k not in want]
    for k, shp in want.items():
        if k in sd and tuple(sd[k].shape) != tuple(shp):
            raise RuntimeError(f"size mismatch for {k}: checkpoint {tuple(sd[k].shape)} vs model {shp}")
    return missing, unexpected
