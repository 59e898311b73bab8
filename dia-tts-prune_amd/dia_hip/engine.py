"""Host side of the MI355X decode path: weight residency, prefill, and the decode loop.

Everything numerical happens in ``libdia_hip.so``; torch tensors are storage (``data_ptr()``),
streams and events.  The call sequence mirrors the reference's ``Dia._prepare_generation``
(dia/model.py:355-427) and the ``while`` loop of ``Dia.generate`` (model.py:748-807):

  prefill(b):  text ids -> encoder (12 layers) -> per-decoder-layer cross K/V     [model.py:382-397]
  decode:      one hipGraph replay per step; token state machine on the device     [model.py:748-807]

Rows of every activation buffer are ordered ``2*b + {0: uncond, 1: cond}`` (model.py:362).  The
uncond row never sees text, so the encoder runs on the packed non-pad tokens of
the cond row only — exact, SURVEY.md App. B3.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import binding as hb
from . import compact as cpt
from . import layout as lay
from .config import DiaConfig

HEAD_DIM = 128


def _ceil(v: int, m: int) -> int:
    return (v + m - 1) // m * m


@dataclass
class TiledW:
    t: torch.Tensor
    kt: int
    ns: int

    @property
    def nbytes(self) -> int:
        return self.t.numel() * 2


class DeviceWeights:
    """Checkpoint -> kernel layouts, resident in HBM (bf16 tiles; norms / embeddings fp32)."""

    def __init__(self, cfg: DiaConfig, sd: Dict[str, torch.Tensor], device: torch.device, compact: str = "auto",
                 weight_planes: int = 1, seg: str = "off"):
        """compact: "auto" = drop structure that a structured-pruned checkpoint zeroed (decoder only),
        "off" = keep every matrix at its checkpoint shape (zeros are streamed).
        weight_planes: 1 = every DenseGeneral kernel as ONE bf16 tile set (exact for bf16-representable checkpoints, the fast
        kernels); 3 = the hi / mid / lo bf16 planes of the fp32 weights (exact for any checkpoint, 3x the bytes, the generic
        kernel: the parity configuration of a genuine fp32 checkpoint).
        seg: "on" = a dense Dia-1.6B-shaped decoder on a GPU also carries the ring arenas of the persistent MLP segments
        (layout.seg_ring; + 2.2 GB: co, wi, wo and the following layer's qkv once more, per CU in consumption order), which
        batch 1-2 sessions then run instead of four launches per layer when the knob seg=1 is set.  EXPERIMENT, default "off":
        measured 43 us per segment against 31 us for the four launches it replaces."""
        if weight_planes not in (1, 3):
            raise ValueError("weight_planes must be 1 or 3")
        self.weight_planes = weight_planes
        m, e, d = cfg.model, cfg.model.encoder, cfg.model.decoder
        if d.gqa_head_dim != HEAD_DIM or d.cross_head_dim != HEAD_DIM or e.head_dim != HEAD_DIM:
            raise hb.DiaHipError("the HIP attention kernels are built for head_dim 128 (Dia-1.6B)")
        if d.n_embd % 32 or d.n_hidden % 32 or e.n_embd % 32 or e.n_hidden % 32:
            raise hb.DiaHipError("embedding / hidden sizes must be multiples of 32")
        self.cfg, self.device = cfg, device

        def dev(name):
            return sd[name].to(device=device, dtype=torch.float32)

        self.max_weight_rounding = 0.0                  # largest |w - bf16(w)| / max|w| over the DenseGeneral kernels

        def tile(w2d) -> TiledW:
            t, kt, ns = lay.tile_weight_planes(w2d) if weight_planes == 3 else lay.tile_weight(w2d)
            if w2d.numel() and weight_planes == 1:
                err = (w2d - w2d.to(torch.bfloat16).to(w2d.dtype)).abs().max()
                scale = w2d.abs().max()
                if float(scale) > 0.0:
                    self.max_weight_rounding = max(self.max_weight_rounding, float(err / scale))
            return TiledW(t, kt, ns)

        perm = lay.rope_pair_perm(HEAD_DIM).to(device)
        self.enc_emb = dev("encoder.embedding.weight").contiguous()
        self.enc_layers = []
        i32e = lambda t: t.to(device=device, dtype=torch.int32).contiguous()
        E, EH = e.n_embd, e.n_head
        eplans = []
        for i in range(e.n_layer):
            pl = cpt.plan_encoder_layer({k: v for k, v in sd.items() if k.startswith(f"encoder.layers.{i}.")},
                                        f"encoder.layers.{i}.", EH)
            eplans.append(pl if (compact != "off" and cpt.enc_is_pruned(pl)) else None)
        for i in range(e.n_layer):
            p = f"encoder.layers.{i}."
            P = eplans[i]
            wq, wk, wv = (dev(p + f"self_attention.{n}_proj.weight").reshape(E, -1) for n in "qkv")
            o = dev(p + "self_attention.o_proj.weight").reshape(-1, E)
            wi3 = dev(p + "mlp.wi_fused.weight")
            wo = dev(p + "mlp.wo.weight")
            EL = dict(g_sa=dev(p + "pre_sa_norm.weight").contiguous(), g_mlp=dev(p + "post_sa_norm.weight").contiguous(),
                      heads=EH, cmap_mlp=None, cmap_next=None)
            nxt = eplans[i + 1] if i + 1 < e.n_layer else None
            if nxt is not None:
                EL["cmap_next"] = i32e(cpt._cmap(nxt.keep_qkv))
            if P is not None:
                hc = P.live_heads.to(device).repeat_interleave(HEAD_DIM)            # live head columns
                kq = P.keep_qkv.to(device)
                wq, wk, wv = wq[kq][:, hc], wk[kq][:, hc], wv[kq][:, hc]
                o = cpt.pad_rows(o[hc])
                hid = cpt.pad_hidden_keep(P.live_hidden).to(device)
                wi3 = torch.stack([cpt.take_cols_idx(wi3[:, 0, :], hid), cpt.take_cols_idx(wi3[:, 1, :], hid)], dim=1)
                wi3 = wi3[P.keep_wi.to(device)]
                wo_c = wo[hid.clamp(min=0)].clone()
                wo_c[hid < 0] = 0
                wo = wo_c
                EL.update(heads=int(P.live_heads.sum()), cmap_mlp=i32e(cpt._cmap(P.keep_wi)))
            EL.update(qkv=tile(torch.cat([wq, wk, wv], dim=1)), o=tile(o), wi=tile(lay.interleave_gate_up(wi3)), wo=tile(wo))
            self.enc_layers.append(EL)
        self.enc_cmap_first = i32e(cpt._cmap(eplans[0].keep_qkv)) if eplans and eplans[0] is not None else None
        self.enc_compacted = any(pl is not None for pl in eplans)
        self.enc_norm = dev("encoder.norm.weight").contiguous()
        self.dec_emb = torch.stack([dev(f"decoder.embeddings.{c}.weight") for c in range(cfg.data.channels)]).contiguous()
        self.dec_layers = []
        D = d.n_embd
        QH, KVH, CH = d.gqa_query_heads, d.kv_heads, d.cross_query_heads
        plans = []
        for i in range(d.n_layer):
            pl = cpt.plan_decoder_layer({k: v for k, v in sd.items() if k.startswith(f"decoder.layers.{i}.")},
                                        f"decoder.layers.{i}.", QH, KVH, CH)
            plans.append(pl if (compact != "off" and cpt.is_pruned(pl)) else None)
        keep_logits = cpt.pad_keep(cpt.nonzero_rows(sd["decoder.logits_dense.weight"].reshape(D, -1)))
        logits_pruned = compact != "off" and not bool(keep_logits.all())
        self.compacted = any(p is not None for p in plans) or logits_pruned or self.enc_compacted
        i32 = lambda t: t.to(device=device, dtype=torch.int32).contiguous()
        ones_d = torch.ones(D, dtype=torch.bool)

        def next_keep(i):          # input order of what consumes layer i's wo output
            if i + 1 < d.n_layer:
                return plans[i + 1].keep_qkv if plans[i + 1] is not None else None
            return keep_logits if logits_pruned else None

        for i in range(d.n_layer):
            p = f"decoder.layers.{i}."
            P = plans[i]
            wq, wk, wv = (dev(p + f"self_attention.{n}_proj.weight").reshape(D, -1) for n in "qkv")
            qkv = torch.cat([wq, wk, wv], dim=1)
            o = dev(p + "self_attention.o_proj.weight").reshape(-1, D)
            cq = dev(p + "cross_attention.q_proj.weight").reshape(D, -1)
            co = dev(p + "cross_attention.o_proj.weight").reshape(-1, D)
            ck = dev(p + "cross_attention.k_proj.weight")[:, :, perm].reshape(e.n_embd, -1)
            cv = dev(p + "cross_attention.v_proj.weight").reshape(e.n_embd, -1)
            ckv = torch.cat([ck, cv], dim=1)
            wi3 = dev(p + "mlp.wi_fused.weight")
            wo = dev(p + "mlp.wo.weight")
            L = dict(g_sa=dev(p + "pre_sa_norm.weight").contiguous(), g_ca=dev(p + "pre_ca_norm.weight").contiguous(),
                     g_mlp=dev(p + "pre_mlp_norm.weight").contiguous(),
                     cmap_ca=None, cmap_mlp=None, cmap_next=None, smap_qkv=None, smap_cq=None, smap_ckv=None,
                     hmap_self=None, hmap_cross=None)
            nk = next_keep(i)
            if nk is not None:
                L["cmap_next"] = i32(cpt._cmap(nk))
            if P is not None:
                cols = lambda strips: (torch.tensor(strips, dtype=torch.long)[:, None] * 16 + torch.arange(16)[None, :]).reshape(-1).to(device)
                s_qkv = (cpt.strips_of_heads(P.live_q_heads, 0) + cpt.strips_of_heads(P.live_kv_heads, QH * 128)
                         + cpt.strips_of_heads(P.live_kv_heads, (QH + KVH) * 128))
                qkv = qkv[P.keep_qkv.to(device)][:, cols(s_qkv)]
                o = cpt.pad_rows(o[P.live_q_heads.to(device).repeat_interleave(128)])
                s_cq = cpt.strips_of_heads(P.live_c_heads, 0)
                cq = cq[P.keep_cq.to(device)][:, cols(s_cq)]
                co = cpt.pad_rows(co[P.live_c_heads.to(device).repeat_interleave(128)])
                s_ckv = cpt.strips_of_heads(P.live_c_heads, 0) + cpt.strips_of_heads(P.live_c_heads, CH * 128)
                ckv = ckv[:, cols(s_ckv)]
                hid = cpt.pad_hidden_keep(P.live_hidden).to(device)
                wi3 = torch.stack([cpt.take_cols_idx(wi3[:, 0, :], hid), cpt.take_cols_idx(wi3[:, 1, :], hid)], dim=1)
                wi3 = wi3[P.keep_wi.to(device)]
                wo_c = wo[hid.clamp(min=0)].clone()
                wo_c[hid < 0] = 0
                wo = wo_c
                L.update(cmap_ca=i32(cpt._cmap(P.keep_cq)), cmap_mlp=i32(cpt._cmap(P.keep_wi)),
                         smap_qkv=i32(torch.tensor(s_qkv)), smap_cq=i32(torch.tensor(s_cq)), smap_ckv=i32(torch.tensor(s_ckv)),
                         hmap_self=i32(cpt.head_map(P.live_q_heads)), hmap_cross=i32(cpt.head_map(P.live_c_heads)))
            L.update(qkv=tile(qkv), o=tile(o), cq=tile(cq), co=tile(co), ckv=tile(ckv),
                     wi=tile(lay.interleave_gate_up(wi3)), wo=tile(wo))
            # experiment (knob wo_diag=1): wo once more in the diagonal layout (4-column groups: 256 workgroups with the whole K each)
            L["wo_diag"] = None
            if (hb.get_tuning("wo_diag") == 1 and P is None and weight_planes == 1 and device.type == "cuda" and wo.shape[0] % 1024 == 0
                    and wo.shape[0] <= 8192 and wo.shape[1] % 8 == 0 and nk is None):
                L["wo_diag"] = lay.diag_tile_weight(wo)
            self.dec_layers.append(L)
        self.cmap_first = i32(cpt._cmap(plans[0].keep_qkv)) if plans[0] is not None else None
        self.seg_layers: List[torch.Tensor] = []
        nqkv = (QH + 2 * KVH) * HEAD_DIM
        if (seg == "on" and device.type == "cuda" and weight_planes == 1 and not self.compacted and D == 2048 and d.n_hidden == 8192
                and QH * HEAD_DIM == 2048 and CH * HEAD_DIM == 2048 and nqkv == 3072):
            for i in range(d.n_layer):
                p = f"decoder.layers.{i}."
                wi3 = dev(p + "mlp.wi_fused.weight")
                qn = None
                if i + 1 < d.n_layer:
                    pn = f"decoder.layers.{i + 1}."
                    qn = torch.cat([dev(pn + f"self_attention.{n_}_proj.weight").reshape(D, -1) for n_ in "qkv"], dim=1)
                self.seg_layers.append(lay.seg_ring(dev(p + "cross_attention.o_proj.weight").reshape(-1, D), wi3[:, 0, :], wi3[:, 1, :],
                                                    dev(p + "mlp.wo.weight"), qn))
                del wi3, qn
        self.dec_norm = dev("decoder.norm.weight").contiguous()
        lw = dev("decoder.logits_dense.weight").reshape(D, -1)
        self.logits = tile(lw[keep_logits.to(device)] if logits_pruned else lw)
        self.logits_cols = lw.shape[1]
        npos = max(cfg.data.audio_length, cfg.data.text_length) + 1
        cos, sin = lay.rope_tables(npos, HEAD_DIM, m.rope_min_timescale, m.rope_max_timescale)
        self.cos_t, self.sin_t = cos.to(device), sin.to(device)
        # strip map of the merged cross-K/V launch of a compacted decoder: layer l's compact strip s -> l * (CH * 16) + original strip
        self.smap_ckv_all = None
        if any(L["smap_ckv"] is not None for L in self.dec_layers):
            full = torch.arange(CH * 16, dtype=torch.int32, device=device)
            self.smap_ckv_all = torch.cat([(L["smap_ckv"] if L["smap_ckv"] is not None else full) + i * CH * 16
                                           for i, L in enumerate(self.dec_layers)]).to(torch.int32).contiguous()
        self.flat = None
        self.pack_flat()

    def ckv_all(self) -> Optional[TiledW]:
        """The cross-K/V tile sets of every decoder layer as ONE weight of sum(ns) strips, when they are one bf16 tile set each with
        the same K and sit back to back in the arena (pack_flat); None otherwise (three-plane weights: one launch per layer)."""
        ts = [L["ckv"] for L in self.dec_layers]
        if self.weight_planes != 1 or not ts or any(t.kt != ts[0].kt for t in ts):
            return None
        p0 = ts[0].t.data_ptr()
        for t in ts:
            if t.t.data_ptr() != p0:
                return None
            p0 += t.nbytes
        n = sum(t.t.numel() for t in ts)
        base = ts[0].t
        whole = torch.empty(0, dtype=base.dtype, device=base.device).set_(base.untyped_storage(), base.storage_offset(), (n,))
        return TiledW(whole, ts[0].kt, sum(t.ns for t in ts))

    def tensors(self) -> List[torch.Tensor]:
        """Every device tensor of the model, in a deterministic order (the layout of the flat arena)."""
        out: List[torch.Tensor] = [self.enc_emb]
        for L in self.enc_layers:
            out += [L["g_sa"], L["g_mlp"], L["qkv"].t, L["o"].t, L["wi"].t, L["wo"].t]
            out += [L[k] for k in ("cmap_mlp", "cmap_next") if L[k] is not None]
        out += [self.enc_norm, self.dec_emb]
        for L in self.dec_layers:
            out += [L["g_sa"], L["g_ca"], L["g_mlp"]] + [L[k].t for k in ("qkv", "o", "cq", "co", "wi", "wo")]
            out += [L[k] for k in ("cmap_ca", "cmap_mlp", "cmap_next", "smap_qkv", "smap_cq", "smap_ckv", "hmap_self", "hmap_cross", "wo_diag")
                    if L.get(k) is not None]
        # the cross-K/V projections of ALL layers back to back (whole KiB each, so the 256-byte slots leave no gaps): the prefill runs them
        # as ONE GEMM over their common input (ckv_all)
        out += [L["ckv"].t for L in self.dec_layers]
        if self.smap_ckv_all is not None:
            out.append(self.smap_ckv_all)
        out += [self.dec_norm, self.logits.t, self.cos_t, self.sin_t]
        out += [t for t in (self.cmap_first, self.enc_cmap_first) if t is not None]
        out += self.seg_layers
        return out

    def pack_flat(self):
        """Move every tensor into ONE contiguous byte arena (256-byte aligned slots, order of tensors()): the model is a
        single HBM allocation, and the multi-GPU weight distribution is a single broadcast of it (one
        ncclBroadcast of the repacked weights).  Tensor objects keep their identity; only their storage moves."""
        ts = self.tensors()
        offs, tot = [], 0
        for t in ts:
            offs.append(tot)
            tot += (t.numel() * t.element_size() + 255) // 256 * 256
        flat = torch.zeros(tot, dtype=torch.uint8, device=self.device)
        for t, o in zip(ts, offs):
            nb = t.numel() * t.element_size()
            v = flat[o: o + nb].view(t.dtype).view(t.shape)
            v.copy_(t)
            t.data = v
        self.flat = flat

    @classmethod
    def empty_like_config(cls, cfg: DiaConfig, device: torch.device, weight_planes: int = 1, seg: str = "off") -> "DeviceWeights":
        """Same tensors, zero-filled: the receive side of the multi-GPU weight broadcast (dense layout;
        a compacted, i.e. structured-pruned, model has checkpoint-dependent shapes: every rank then
        loads the checkpoint itself instead of receiving a broadcast).  `weight_planes` must be the sender's
        (dist.broadcast_weights checks it on every rank before the arena travels)."""
        from .weights import param_shapes
        sd = {k: torch.zeros(shp, dtype=torch.float32, device=device) for k, shp in param_shapes(cfg).items()}
        return cls(cfg, sd, device, compact="off", weight_planes=weight_planes, seg=seg)

    def prefill_weight_bytes(self) -> int:
        """bf16 bytes the prefill streams once per batch: the encoder and the cross K/V projections"""
        n = sum(L[k].nbytes for L in self.enc_layers for k in ("qkv", "o", "wi", "wo"))
        return n + sum(L["ckv"].nbytes for L in self.dec_layers)

    def decode_weight_bytes(self) -> int:
        """bf16 bytes one decode step streams (SURVEY.md §8d 'W'): every decoder matrix except the
        prefill-only cross K/V projections, plus the logits head."""
        n = self.logits.nbytes
        for L in self.dec_layers:
            n += sum(L[k].nbytes for k in ("qkv", "o", "cq", "co", "wi", "wo"))
        return n


@dataclass
class UtteranceResult:
    tokens: np.ndarray          # int32 [T, C] token buffer (DecoderOutput.generated_tokens)
    codes: np.ndarray           # rows [prefill_step : last_step+1]   (model.py:831)
    last_step: int
    preds: np.ndarray           # int32 [T, C] raw samples per step row (before the EOS state machine)
    text_len: int


class DecodeSession:
    """Buffers + engine for one batch of utterances."""

    def __init__(self, w: DeviceWeights, text_ids: Sequence[np.ndarray], *, kv_dtype: str = "bf16",
                 max_tokens: Optional[int] = None, cfg_scale: float = 3.0, temperature: float = 1.3,
                 top_p: float = 0.95, top_k: int = 35, seeds: Optional[Sequence[Optional[int]]] = None,
                 noise: Optional[torch.Tensor] = None, ignore_eos: bool = False,
                 teacher_tokens: Optional[Sequence[np.ndarray]] = None, stream: Optional[torch.cuda.Stream] = None,
                 s_cap: Optional[int] = None, audio_prompts: Optional[Sequence[Optional[np.ndarray]]] = None,
                 prompt_prefill: str = "auto", attention: str = "auto"):
        """audio_prompts: per utterance None or int codes [Tp, C] (reference model.py:311-353).  The prompt
        rows are replayed through the decode step before sampling starts (semantics: oracle.generate), or — with
        bf16 caches — prefilled as one packed MFMA batch; prompt_prefill="replay" forces the replay.
        attention="valu" keeps bf16 V caches row-major and runs the VALU attention kernel (comparison runs)."""
        if prompt_prefill not in ("auto", "replay") or attention not in ("auto", "valu"):
            raise ValueError("prompt_prefill must be 'auto' or 'replay', attention 'auto' or 'valu'")
        self.prompt_prefill = prompt_prefill
        cfg, dev = w.cfg, w.device
        self.w, self.cfg, self.dev = w, cfg, dev
        d, da = cfg.model.decoder, cfg.data
        self.B = B = len(text_ids)
        self.R = 2 * B
        self.rows_pad = _ceil(self.R, 16)
        self.T = da.audio_length
        self.C, self.V, self.D, self.F = da.channels, cfg.model.tgt_vocab_size, d.n_embd, d.n_hidden
        self.max_tokens = self.T if max_tokens is None else int(max_tokens)
        if not (2 <= self.max_tokens <= self.T):
            raise ValueError(f"max_tokens must be in [2, {self.T}]")
        # "bf16x2": every K / V value as hi + lo bf16 in two planes of the bf16 layouts (16 significand bits, the bytes of the fp32
        # caches): the MFMA attention kernel with logits inside the 1e-3 parity bound
        self.kv_code = {"f32": hb.KV_F32, "float32": hb.KV_F32, "bf16": hb.KV_BF16, "bfloat16": hb.KV_BF16, "bf16x2": hb.KV_BF16X2}[kv_dtype]
        if self.kv_code == hb.KV_BF16X2 and attention == "valu":
            raise ValueError("kv_dtype 'bf16x2' is served by the MFMA attention kernel only")
        # bf16 caches keep V blocked as [key/32][128][32] for the MFMA attention kernel
        self.v_blocked = int(self.kv_code in (hb.KV_BF16, hb.KV_BF16X2) and attention != "valu")
        kvt = torch.float32 if self.kv_code == hb.KV_F32 else torch.bfloat16
        self.kv_planes = 2 if self.kv_code == hb.KV_BF16X2 else 1
        self.stream = stream if stream is not None else torch.cuda.Stream(device=dev)
        self.lens = [int(len(t)) for t in text_ids]
        self.S = s_cap if s_cap is not None else max(32, _ceil(max(self.lens + [1]), 32))
        if self.S > da.text_length and s_cap is None:
            self.S = da.text_length
        self.text_ids = [np.asarray(t, dtype=np.int32) for t in text_ids]
        self.teacher = teacher_tokens is not None
        self.ignore_eos = ignore_eos
        md = max(da.delay_pattern)
        self.max_delay = md

        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=dev)
        mt = self.rows_pad // 16
        self.xkt, self.akt, self.hkt = self.D // 32, max(d.gqa_query_heads, d.cross_query_heads) * HEAD_DIM // 32, self.F // 32
        self.x = z(self.rows_pad, self.D)
        # activations between the kernels of a step travel as fp32 tiles in these buffers (the fragment order of one plane, 4-byte
        # values: 4 instead of the 6 bytes of three bf16 planes hi + mid + lo == fp32); every consumer splits the planes itself — the
        # M <= 4 GEMV while staging its image through LDS, the 16-row GEMM in registers — same arithmetic bit for bit.
        # Tuning knob act_f32=0 keeps the planes between the kernels.
        t_ = hb.get_tuning("act_f32")
        self.act_f32 = int(t_ != 0)
        # persistent MLP segments (csrc/seg.hip, experiment): batch 1-2 on a model that carries the ring arenas, knob seg=1
        self.seg = bool(self.R <= 4 and getattr(w, "seg_layers", None) and self.act_f32 and hb.get_tuning("seg") == 1)
        self.seg_ws = None
        if self.seg:
            self.seg_ws = torch.zeros(int(hb.lib().dia_seg_workspace_bytes()), dtype=torch.uint8, device=dev)
        self.planes_x = z(3, mt, self.xkt, 64, 8, dt=torch.bfloat16)
        self.planes_a = z(3, mt, self.akt, 64, 8, dt=torch.bfloat16)
        self.planes_h = z(3, mt, self.hkt, 64, 8, dt=torch.bfloat16)
        self.ssq = z(self.D // 8, self.rows_pad)          # strip sums of squares: D / 16 per row, D / 8 behind the diagonal-layout wo
        self.nqkv = (d.gqa_query_heads + 2 * d.kv_heads) * HEAD_DIM
        self.qkv = z(self.rows_pad, self.nqkv)
        self.qc = z(self.rows_pad, d.cross_query_heads * HEAD_DIM)
        self.ld_logits = w.logits.ns * 16
        self.logits = z(self.rows_pad, self.ld_logits)
        pl = (self.kv_planes,) if self.kv_planes > 1 else ()           # two-plane caches: [plane][...]
        self.k_self = [z(*pl, self.R, d.kv_heads, self.T, HEAD_DIM, dt=kvt) for _ in range(d.n_layer)]
        self.v_self = [z(*pl, self.R, d.kv_heads, self.T, HEAD_DIM, dt=kvt) for _ in range(d.n_layer)]
        # cross caches of all layers in ONE allocation each (per-layer views): the prefill's merged cross-K/V launch addresses a layer
        # as kc + layer * kv_layer_stride
        self.k_cross_all = z(d.n_layer, *pl, B, d.cross_query_heads, self.S, HEAD_DIM, dt=kvt)
        self.v_cross_all = z(d.n_layer, *pl, B, d.cross_query_heads, self.S, HEAD_DIM, dt=kvt)
        self.k_cross = [self.k_cross_all[i] for i in range(d.n_layer)]
        self.v_cross = [self.v_cross_all[i] for i in range(d.n_layer)]
        self.kv_plane_self = self.R * d.kv_heads * self.T * HEAD_DIM if self.kv_planes > 1 else 0
        self.kv_plane_cross = B * d.cross_query_heads * self.S * HEAD_DIM if self.kv_planes > 1 else 0
        self.text_len = torch.tensor(self.lens, dtype=torch.int32, device=dev)
        nsc = max(hb.lib().dia_attn_scratch_floats(self.R, d.kv_heads, self.T),
                  hb.lib().dia_attn_scratch_floats(B, d.cross_query_heads, self.S))
        self.attn_scratch = z(max(nsc, 1))
        self.attn_tickets = z(max(self.R * d.kv_heads, B * d.cross_query_heads), dt=torch.int32)
        # split-K slabs: up to 4 splits of wo (one or two m-tiles); with 17..32 rows every GEMM splits K in two
        ns_max = max([self.D // 16, w.logits.ns] + [DL[k].ns for DL in w.dec_layers for k in ("qkv", "o", "cq", "co", "wi", "wo")])
        n_scr = max((self.D // 16) * 4 * 512, ns_max * 2 * 512 if 16 < self.R <= 32 else 0)
        if 16 < self.R <= 128:   # 2..8 m-tiles: wo splits K four ways for every m-tile (k_gemm16 over gridDim.z)
            n_scr = max(n_scr, 2 * -(-self.R // 32) * (self.D // 16) * 4 * 256)       # (whole PAIRS of m-tiles: k_gemm2t hands two tiles over together)
        if 16 < self.R <= 32:    # k_gemm_blk32: column blocks x K ranges of >= 8 k-tiles, 512 floats per strip and range
            n_scr = max([n_scr, w.logits.ns * -(-w.logits.kt // 8) * 512] +
                        [DL[k].ns * -(-DL[k].kt // 8) * 512 for DL in w.dec_layers for k in ("qkv", "o", "cq", "co", "wi", "wo")])
        self.sk_scratch = z(n_scr)
        self.sk_tickets = z(max(ns_max, 8 * (self.D // 16)), dt=torch.int32)
        self.mlp_barrier = z(2, dt=torch.int32)          # dia_mlp_fused: arrivals, error flag

        # token buffer + state machine (state.py:178-208; model.py:736-741)
        from .tokens import delayed_prefill
        tok = np.full((B, self.T, self.C), -1, dtype=np.int32)
        self.first_steps = []
        for b in range(B):
            pr = None if audio_prompts is None else audio_prompts[b]
            if pr is not None:
                pr = np.asarray(pr)
                if pr.ndim == 3 and pr.shape[0] == 1:
                    pr = pr[0]
                if pr.ndim != 2 or pr.shape[1] != self.C:
                    raise ValueError(f"Unexpected audio_prompt shape: {pr.shape}. Expected [T, C] or [1, T, C].")   # model.py:316
            prefill, pstep = delayed_prefill(cfg, pr)
            if prefill.shape[0] > self.T:
                raise ValueError(f"audio prompt of {pstep - 1} frames does not fit audio_length {self.T}")
            tok[b, : prefill.shape[0]] = prefill
            self.first_steps.append(int(pstep))
        if self.teacher:
            for b in range(B):
                tt = np.asarray(teacher_tokens[b], dtype=np.int32)
                tok[b, : tt.shape[0]] = tt
        self.prefill_step = 1                     # every utterance starts at step 1; steps < first_step replay the prompt
        self.first_step = torch.tensor(self.first_steps, dtype=torch.int32, device=dev)
        self.tokens = torch.from_numpy(tok).to(dev)
        self.pred = torch.full((B, self.T, self.C), -1, dtype=torch.int32, device=dev)
        self.cur = torch.full((B,), 1, dtype=torch.int32, device=dev)
        fsm = np.zeros((B, 8), dtype=np.int32)
        fsm[:, 1] = -1
        fsm[:, 2] = md
        self.fsm = torch.from_numpy(fsm).to(dev)
        self.delay = torch.tensor(list(da.delay_pattern), dtype=torch.int32, device=dev)

        # Exp(1) variates for the multinomial draw: the reference's generator stream after
        # torch.manual_seed(seed) (model.py:679-683), one [C,V] draw per sampled step.  Drawing 3071 rows takes
        # 0.7 s per utterance on the host, so they are drawn in chunks as the decode advances (the stream of a
        # generator is the same whether it is consumed in one draw or in pieces) and uploaded ahead of the steps
        # that read them; a generation that ends early never draws the rest.
        self.noise_steps = self.max_tokens - 1
        self._noise_rows = 0                    # rows [0, _noise_rows) are on the device
        self._issued = 0                        # decode steps enqueued so far
        self._gens = None
        if temperature != 0.0:
            if noise is not None:
                nz = noise.to(torch.float32)
                if tuple(nz.shape) != (B, self.noise_steps, self.C, self.V):
                    raise ValueError(f"noise must be [B,{self.noise_steps},{self.C},{self.V}]")
                self.noise = nz.to(dev)
                self._noise_rows = self.noise_steps
            else:
                self.noise = torch.empty(B, self.noise_steps, self.C, self.V, dtype=torch.float32, device=dev)
                self._gens = []
                for b in range(B):
                    g = torch.Generator()
                    sd = None if seeds is None else seeds[b]
                    if sd is None:
                        g.seed()
                    else:
                        g.manual_seed(int(sd))
                    self._gens.append(g)
        else:
            self.noise = None
            self._noise_rows = self.noise_steps

        self.sample_params = dict(cfg_scale=float(cfg_scale), temperature=float(temperature), top_p=float(top_p),
                                  top_k=int(top_k or 0))
        self._engine = C.c_void_p()
        self._build_engine()
        self.prefilled = False

    # ------------------------------------------------------------------ engine descriptor
    def _embed_args(self) -> hb.EmbedArgs:
        e = hb.EmbedArgs()
        e.tokens, e.cur = hb.ptr(self.tokens), hb.ptr(self.cur)
        e.B, e.T, e.C, e.V, e.D = self.B, self.T, self.C, self.V, self.D
        e.emb, e.g, e.x = hb.ptr(self.w.dec_emb), hb.ptr(self.w.dec_layers[0]["g_sa"]), hb.ptr(self.x)
        e.P, e.p_plane_stride, e.p_ktiles = hb.ptr(self.planes_x), self.planes_x[0].numel(), self.xkt
        e.ssq_ld, e.ssq = self.rows_pad, hb.ptr(self.ssq)
        e.cmap = hb.ptr(self.w.cmap_first)
        e.act_f32 = self.act_f32
        return e

    def _sample_args(self) -> hb.SampleArgs:
        da = self.cfg.data
        s = hb.SampleArgs()
        s.logits, s.ld_logits, s.B, s.T, s.C, s.V = hb.ptr(self.logits), self.ld_logits, self.B, self.T, self.C, self.V
        s.max_tokens = self.max_tokens
        s.cfg_scale, s.temperature, s.top_p = (self.sample_params[k] for k in ("cfg_scale", "temperature", "top_p"))
        s.top_k = self.sample_params["top_k"]
        s.eos, s.pad, s.bos, s.max_delay = da.audio_eos_value, da.audio_pad_value, da.audio_bos_value, self.max_delay
        s.ignore_eos, s.teacher = int(self.ignore_eos), int(self.teacher)
        s.delay, s.noise, s.noise_steps = hb.ptr(self.delay), hb.ptr(self.noise), self.noise_steps
        s.tokens, s.pred, s.cur, s.fsm = hb.ptr(self.tokens), hb.ptr(self.pred), hb.ptr(self.cur), hb.ptr(self.fsm)
        s.first_step = hb.ptr(self.first_step) if any(f != 1 for f in self.first_steps) else None
        s.embed = self._embed_args()
        return s

    def _build_engine(self):
        d = self.cfg.model.decoder
        w = self.w
        n = d.n_layer
        self._layers = (hb.DecLayer * n)()
        for i, L in enumerate(w.dec_layers):
            dl = self._layers[i]
            for key, f in (("qkv", "qkv"), ("o", "o"), ("cq", "cq"), ("co", "co"), ("wi", "wi"), ("wo", "wo")):
                setattr(dl, "w_" + f, hb.ptr(L[key].t))
                setattr(dl, "kt_" + f, L[key].kt)
                setattr(dl, "ns_" + f, L[key].ns)
            dl.g_sa, dl.g_ca, dl.g_mlp = hb.ptr(L["g_sa"]), hb.ptr(L["g_ca"]), hb.ptr(L["g_mlp"])
            dl.k_self, dl.v_self = hb.ptr(self.k_self[i]), hb.ptr(self.v_self[i])
            dl.k_cross, dl.v_cross = hb.ptr(self.k_cross[i]), hb.ptr(self.v_cross[i])
            for f in ("cmap_ca", "cmap_mlp", "cmap_next", "smap_qkv", "smap_cq", "hmap_self", "hmap_cross"):
                setattr(dl, f, hb.ptr(L[f]))
            dl.w_wo_diag = hb.ptr(L.get("wo_diag"))
        ed = hb.EngineDesc()
        ed.n_layer, ed.D, ed.F = n, self.D, self.F
        ed.q_heads, ed.kv_heads, ed.cq_heads = d.gqa_query_heads, d.kv_heads, d.cross_query_heads
        ed.C, ed.V, ed.B, ed.T, ed.S = self.C, self.V, self.B, self.T, self.S
        ed.kv_dtype, ed.rows_pad, ed.ld_logits = self.kv_code, self.rows_pad, self.ld_logits
        ed.v_blocked = self.v_blocked
        ed.kv_plane_self, ed.kv_plane_cross = self.kv_plane_self, self.kv_plane_cross
        ed.eps = float(self.cfg.model.normalization_layer_epsilon)
        ed.layers = C.cast(self._layers, C.POINTER(hb.DecLayer))
        ed.w_logits, ed.kt_logits, ed.ns_logits = hb.ptr(w.logits.t), w.logits.kt, w.logits.ns
        ed.g_final = hb.ptr(w.dec_norm)
        ed.x, ed.planes_x, ed.planes_a, ed.planes_h = hb.ptr(self.x), hb.ptr(self.planes_x), hb.ptr(self.planes_a), hb.ptr(self.planes_h)
        ed.ssq, ed.qkv, ed.qc, ed.logits = hb.ptr(self.ssq), hb.ptr(self.qkv), hb.ptr(self.qc), hb.ptr(self.logits)
        ed.cos_t, ed.sin_t, ed.text_len = hb.ptr(w.cos_t), hb.ptr(w.sin_t), hb.ptr(self.text_len)
        ed.attn_scratch, ed.attn_tickets = hb.ptr(self.attn_scratch), hb.ptr(self.attn_tickets)
        ed.sk_scratch, ed.sk_tickets = hb.ptr(self.sk_scratch), hb.ptr(self.sk_tickets)
        ed.sk_scratch_floats = self.sk_scratch.numel()
        ed.mlp_barrier = hb.ptr(self.mlp_barrier)
        ed.act_f32 = self.act_f32
        ed.w_planes = w.weight_planes
        ed.sample = self._sample_args()
        if self.seg:
            self._seg_w = (C.c_void_p * n)(*[hb.ptr(t) for t in w.seg_layers])
            ed.seg_w = C.cast(self._seg_w, C.POINTER(C.c_void_p))
            ed.seg_ws = hb.ptr(self.seg_ws)
        self._desc = ed
        hb.check(hb.lib().dia_engine_create(C.byref(ed), C.c_void_p(self.stream.cuda_stream), C.byref(self._engine)),
                 "dia_engine_create")

    def close(self):
        """Tear the engine down: the stream is drained first (queued graph replays read the executable graph's own
        argument blocks), then dia_engine_destroy releases the graph, its side stream and events."""
        if self._engine:
            eng, self._engine = self._engine, C.c_void_p()
            self.stream.synchronize()
            hb.check(hb.lib().dia_engine_destroy(eng), "dia_engine_destroy")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ prefill
    def prefill(self, keep_encoder_out: bool = False):
        """Encoder + cross-K/V precompute for every utterance, then the first input embedding.
        Replaces model.py:382-397 (Encoder.forward layers.py:445-462; precompute_cross_attn_cache 632-669).

        All utterances run as ONE packed batch: utterance b occupies rows [off_b, off_b + L_b) of every
        activation buffer, off_b a multiple of 32 (whole m-tiles and key blocks), padding rows zero.  The dense layers
        (qkv, o, wi, wo) are single GEMMs over all rows — weights are read once per batch and the row count
        is what the MFMA-tiled kernel wants — and so are the bidirectional attention (dia_enc_attn) and the cross-K/V projection, which find a row's
        utterance and RoPE position through row_b / seg_off; only the embedding runs per utterance.  Only the non-pad tokens of the cond row are computed (exact)."""
        L = hb.lib()
        cfg, w, dev = self.cfg, self.w, self.dev
        e, d = cfg.model.encoder, cfg.model.decoder
        st = C.c_void_p(self.stream.cuda_stream)
        E, Fe = e.n_embd, e.n_hidden
        eps = float(cfg.model.normalization_layer_epsilon)
        offs, tot = [], 0
        for Lb in self.lens:
            offs.append(tot)
            tot += _ceil(Lb, 32)                     # whole 32-key blocks per utterance (blocked V planes)
        self.enc_out = [None] * self.B
        with torch.cuda.stream(self.stream):
            if tot > 0:
                Mp, mt = tot, tot // 16
                Lmax = _ceil(max(self.lens), 16)
                Hmax = max(EL["heads"] for EL in w.enc_layers)
                ekt = E // 32
                akt = max(max(1, Hmax * HEAD_DIM // 32), max(EL["o"].kt for EL in w.enc_layers))   # o rows may be zero-padded
                hkt = max(EL["wo"].kt for EL in w.enc_layers)                 # (compacted) hidden width in k-tiles
                # every buffer of the pass is a view of ONE zero-filled allocation (one memset instead of nine), every small integer
                # table one host array (one copy instead of 3 + B): the chain is ~80 launches, each of these was one more
                nq_max = 3 * Hmax * HEAD_DIM
                shapes = [((Mp, E), torch.float32), ((3, mt, ekt, 64, 8), torch.bfloat16), ((3, mt, akt, 64, 8), torch.bfloat16),
                          ((3, mt, hkt, 64, 8), torch.bfloat16), ((E // 16, Mp), torch.float32), ((Mp, nq_max), torch.float32),
                          ((3, Hmax, Mp, HEAD_DIM), torch.bfloat16), ((3, Hmax, Mp, HEAD_DIM), torch.bfloat16)]
                sizes = [_ceil(int(np.prod(sh)) * (4 if dt == torch.float32 else 2), 256) for sh, dt in shapes]
                ws = torch.zeros(sum(sizes), dtype=torch.uint8, device=dev)
                views, o_ = [], 0
                for (sh, dt), nb in zip(shapes, sizes):
                    n_el = int(np.prod(sh))
                    views.append(ws[o_: o_ + n_el * (4 if dt == torch.float32 else 2)].view(dt).view(*sh))
                    o_ += nb
                x, px, pa, ph, ssq, qkv, kp, vp = views          # (kp / vp: K / V planes of the attention, scratch)
                live = [b for b in range(self.B) if self.lens[b] > 0]
                id_off, n_ids = {}, 0
                for b in live:
                    id_off[b] = n_ids
                    n_ids += _ceil(self.lens[b], 4)                # (16-byte aligned runs)
                tab = np.zeros((Mp + 2 * _ceil(self.B, 4) + n_ids,), dtype=np.int32)
                tab[:Mp] = -1
                for b, Lb in enumerate(self.lens):
                    tab[offs[b]: offs[b] + Lb] = b
                o_so, o_sl = Mp, Mp + _ceil(self.B, 4)
                tab[o_so: o_so + self.B] = offs
                tab[o_sl: o_sl + self.B] = self.lens
                o_id = Mp + 2 * _ceil(self.B, 4)
                for b in live:
                    tab[o_id + id_off[b]: o_id + id_off[b] + self.lens[b]] = self.text_ids[b]
                # pinned + asynchronous: a pageable copy blocks the host until the stream has drained, and every launch behind it starts late
                self._pf_tab_host = torch.from_numpy(tab).pin_memory()        # (kept until the next prefill: the copy reads it in stream order)
                tab_d = self._pf_tab_host.to(dev, non_blocking=True)
                row_b, seg_off, seg_len = tab_d[:Mp], tab_d[o_so: o_so + self.B], tab_d[o_sl: o_sl + self.B]

                def rows(t, b, width):              # device pointer of row off_b of a [Mp, width] fp32 buffer
                    return t.data_ptr() + offs[b] * width * 4

                def planes_at(P, b, kt_):           # device pointer of m-tile off_b/16 of a plane set (bf16)
                    return P.data_ptr() + (offs[b] // 16) * kt_ * 512 * 2

                for b in live:
                    ids = tab_d[o_id + id_off[b]: o_id + id_off[b] + self.lens[b]]
                    hb.check(L.dia_embed_text(hb.ptr(ids), self.lens[b], hb.ptr(w.enc_emb), E, hb.ptr(w.enc_layers[0]["g_sa"]),
                                              rows(x, b, E), planes_at(px, b, ekt), px[0].numel(), ekt,
                                              ssq.data_ptr() + offs[b] * 4, Mp, hb.ptr(w.enc_cmap_first), st), "dia_embed_text")

                def gemm(A, a_kt, W: TiledW, epi, *, M=Mp, a_ptr=None, ssq_ptr=None, ssq_in=False, out=None, ldo=0, gnext=None,
                         P=None, p_kt=0, ssq_out=False, kv=None, strip_map=None, row_map=False, cmap=None, sk=0, kv_layers=None):
                    g = hb.GemmArgs()
                    g.A, g.a_plane_stride, g.a_ktiles, g.M = (a_ptr if a_ptr is not None else hb.ptr(A)), A[0].numel(), a_kt, M
                    g.W, g.KT, g.nstrips, g.epi = hb.ptr(W.t), W.kt, W.ns, epi
                    g.w_planes = w.weight_planes
                    sp = ssq_ptr if ssq_ptr is not None else hb.ptr(ssq)
                    if ssq_in:
                        g.ssq_in, g.ssq_in_n, g.inv_d, g.eps = sp, E // 16, 1.0 / E, eps
                    g.ssq_ld = Mp
                    g.out, g.ldo = hb.ptr(out), ldo
                    g.gnext = hb.ptr(gnext)
                    if P is not None:
                        g.P, g.p_plane_stride, g.p_ktiles = hb.ptr(P), P[0].numel(), p_kt
                    if ssq_out:
                        g.ssq_out = sp
                    g.strip_map = hb.ptr(strip_map)
                    g.cmap = hb.ptr(cmap)
                    g.kv_vblocked = self.v_blocked if kv is not None else 0
                    if kv is not None:
                        g.kc, g.vc, g.kv_dtype, g.kv_heads, g.kv_cap, g.kv_batch_index = kv
                        g.kv_plane_stride = self.kv_plane_cross
                        g.cos_t, g.sin_t = hb.ptr(w.cos_t), hb.ptr(w.sin_t)
                        if kv_layers is not None:
                            g.kv_layer_strips, g.kv_layer_stride = kv_layers
                    if row_map:
                        g.row_b, g.seg_off = hb.ptr(row_b), hb.ptr(seg_off)
                    if sk > 1:      # split-K over workgroups through the session's slab scratch (short prompts: the z-form needs K <= 2048 per workgroup)
                        g.sk, g.sk_scratch, g.sk_tickets, g.sk_scratch_floats = sk, hb.ptr(self.sk_scratch), hb.ptr(self.sk_tickets), self.sk_scratch.numel()
                    hb.check(L.dia_gemm(C.byref(g), st), "dia_gemm")

                for i, EL in enumerate(w.enc_layers):
                    Hl = EL["heads"]                 # live heads of this layer (all of them unless the checkpoint was pruned)
                    nq = 3 * Hl * HEAD_DIM
                    if Hl > 0:
                        gemm(px, ekt, EL["qkv"], hb.EPI_SCALE_STORE, ssq_in=True, out=qkv, ldo=nq)
                        ea_ = hb.EncAttnArgs()
                        ea_.qkv, ea_.ldq, ea_.q_off, ea_.k_off, ea_.v_off = hb.ptr(qkv), nq, 0, Hl * HEAD_DIM, 2 * Hl * HEAD_DIM
                        ea_.heads, ea_.rows = Hl, Mp
                        ea_.row_b, ea_.seg_off, ea_.seg_len = hb.ptr(row_b), hb.ptr(seg_off), hb.ptr(seg_len)
                        ea_.cos_t, ea_.sin_t, ea_.kp, ea_.vp = hb.ptr(w.cos_t), hb.ptr(w.sin_t), hb.ptr(kp), hb.ptr(vp)
                        ea_.P, ea_.p_plane_stride, ea_.p_ktiles = hb.ptr(pa), pa[0].numel(), akt
                        hb.check(L.dia_enc_attn(C.byref(ea_), st), "dia_enc_attn")
                        gemm(pa, akt, EL["o"], hb.EPI_RESID_EMIT, out=x, ldo=E, gnext=EL["g_mlp"], P=px, p_kt=ekt, ssq_out=True,
                             cmap=EL["cmap_mlp"])
                    else:
                        raise hb.DiaHipError("encoder layer with every attention head pruned is not supported")
                    gemm(px, ekt, EL["wi"], hb.EPI_SWIGLU_EMIT, ssq_in=True, P=ph, p_kt=hkt)
                    gnext = w.enc_layers[i + 1]["g_sa"] if i + 1 < len(w.enc_layers) else w.enc_norm
                    # 17..128 rows: wo (K = 4096) as two K halves per strip, so that it rides the z-form of the 16-row kernel instead
                    # of the generic one (whose 4-m-tile form spills)
                    wo_sk = 0
                    kt_wo, ns_wo = EL["wo"].kt, EL["wo"].ns
                    if (16 < Mp <= 128 and kt_wo % 16 == 0 and kt_wo // 2 <= 64 and w.weight_planes == 1 and EL["cmap_next"] is None
                            and self.sk_scratch.numel() >= mt * ns_wo * 2 * 256 and self.sk_tickets.numel() >= mt * ns_wo):
                        wo_sk = 2
                    gemm(ph, hkt, EL["wo"], hb.EPI_RESID_EMIT, out=x, ldo=E, gnext=gnext, P=px, p_kt=ekt, ssq_out=True,
                         cmap=EL["cmap_next"], sk=wo_sk)
                # px now holds planes(x * encoder.norm.weight); ssq the row sums of squares of x
                ckv_all = w.ckv_all() if hb.get_tuning("ckv_merge") != 0 else None
                if ckv_all is not None:             # ONE launch for the 18 layers: their input is the same (18 x 13.8 -> 1 x ~100 us at 98 rows)
                    gemm(px, ekt, ckv_all, hb.EPI_CROSSKV, ssq_in=True,
                         kv=(hb.ptr(self.k_cross_all), hb.ptr(self.v_cross_all), self.kv_code, d.cross_query_heads, self.S, 0),
                         strip_map=w.smap_ckv_all, row_map=True, kv_layers=(d.cross_query_heads * 16, self.k_cross[0].numel()))
                else:
                    for i, DL in enumerate(w.dec_layers):
                        gemm(px, ekt, DL["ckv"], hb.EPI_CROSSKV, ssq_in=True,
                             kv=(hb.ptr(self.k_cross[i]), hb.ptr(self.v_cross[i]), self.kv_code, d.cross_query_heads, self.S, 0),
                             strip_map=DL["smap_ckv"], row_map=True)
                if keep_encoder_out:
                    for b in live:
                        Lb, o = self.lens[b], offs[b]
                        inv = torch.rsqrt(ssq[:, o: o + Lb].sum(dim=0) / E + eps)
                        self.enc_out[b] = (x[o: o + Lb] * inv[:, None] * w.enc_norm[None, :]).clone()
            if self._prompt_prefill_batched():
                self._prompt_prefill(st)
            ea = self._embed_args()
            hb.check(L.dia_embed_tokens(C.byref(ea), st), "dia_embed_tokens")
        self.prefilled = True

    # ------------------------------------------------------------------ audio-prompt prefill, batched
    def _prompt_prefill_batched(self) -> bool:
        """The batched MFMA prefill of the prompt rows needs bf16 caches with the blocked V layout and an
        uncompacted decoder; everything else replays the prompt rows through the decode step (first_step)."""
        return (any(f > 2 for f in self.first_steps) and self.v_blocked == 1 and self.kv_code == hb.KV_BF16 and not self.w.compacted and not self.teacher
                and self.prompt_prefill != "replay" and self.w.weight_planes == 1)

    def _prompt_prefill(self, st):
        """Decoder.forward in prefill mode (layers.py:722-766) for the audio prompts of all utterances at once, with
        the replay's semantics (token row r -> slot r, position r + 1): packed rows = both CFG rows of every
        prompted utterance; dense layers on the MFMA-tiled GEMM, K/V append / causal self-attention / cross
        attention over the caches by the dia_dec_prefill_* kernels.  Afterwards cur[b] = first_step[b]."""
        L = hb.lib()
        cfg, w, dev = self.cfg, self.w, self.dev
        d = cfg.model.decoder
        D, F = self.D, self.F
        QH, KVH, CH = d.gqa_query_heads, d.kv_heads, d.cross_query_heads
        eps = float(cfg.model.normalization_layer_epsilon)
        segs = []                                   # (cache row 2b+c, prompt rows)
        for b, fs in enumerate(self.first_steps):
            if fs > 2:                              # rows 0..fs-2 are prefilled; fs == 2 (one frame) is left to the replay path
                segs += [(2 * b, fs - 1), (2 * b + 1, fs - 1)]
        offs, tot = [], 0
        for _, n in segs:
            offs.append(tot)
            tot += _ceil(n, 32)
        Mp, mt = tot, tot // 16
        rs = np.full((Mp,), -1, dtype=np.int32)
        for i, (_, n) in enumerate(segs):
            rs[offs[i]: offs[i] + n] = i
        i32 = lambda a: torch.tensor(a, dtype=torch.int32, device=dev)
        row_seg = torch.from_numpy(rs).to(dev)
        seg_off, seg_len, seg_row = i32(offs), i32([n for _, n in segs]), i32([r for r, _ in segs])
        z = lambda *sh, dt=torch.float32: torch.zeros(*sh, dtype=dt, device=dev)
        xkt, akt, hkt = D // 32, max(QH, CH) * HEAD_DIM // 32, F // 32
        x = z(Mp, D)
        px, pa, ph = (z(3, mt, kt_, 64, 8, dt=torch.bfloat16) for kt_ in (xkt, akt, hkt))
        ssq = z(D // 16, Mp)
        qkv, qc = z(Mp, self.nqkv), z(Mp, CH * HEAD_DIM)

        def pargs():
            a = hb.DecPrefillArgs()
            a.row_seg, a.seg_off, a.seg_len, a.seg_row, a.rows = hb.ptr(row_seg), hb.ptr(seg_off), hb.ptr(seg_len), hb.ptr(seg_row), Mp
            a.cos_t, a.sin_t = hb.ptr(w.cos_t), hb.ptr(w.sin_t)
            return a

        def gemm(A, a_kt, W: TiledW, epi, *, ssq_in=False, out=None, ldo=0, gnext=None, P=None, p_kt=0, ssq_out=False):
            g = hb.GemmArgs()
            g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), a_kt, Mp
            g.W, g.KT, g.nstrips, g.epi = hb.ptr(W.t), W.kt, W.ns, epi
            g.w_planes = self.w.weight_planes
            if ssq_in:
                g.ssq_in, g.ssq_in_n, g.inv_d, g.eps = hb.ptr(ssq), D // 16, 1.0 / D, eps
            g.ssq_ld = Mp
            g.out, g.ldo, g.gnext = hb.ptr(out), ldo, hb.ptr(gnext)
            if P is not None:
                g.P, g.p_plane_stride, g.p_ktiles = hb.ptr(P), P[0].numel(), p_kt
            if ssq_out:
                g.ssq_out = hb.ptr(ssq)
            hb.check(L.dia_gemm(C.byref(g), st), "dia_gemm")

        a = pargs()
        a.tokens, a.T, a.C, a.V, a.D = hb.ptr(self.tokens), self.T, self.C, self.V, D
        a.emb, a.g, a.x = hb.ptr(w.dec_emb), hb.ptr(w.dec_layers[0]["g_sa"]), hb.ptr(x)
        a.P, a.p_plane_stride, a.p_ktiles, a.ssq, a.ssq_ld = hb.ptr(px), px[0].numel(), xkt, hb.ptr(ssq), Mp
        hb.check(L.dia_dec_prefill_embed(C.byref(a), st), "dia_dec_prefill_embed")
        for i, DL in enumerate(w.dec_layers):
            gemm(px, xkt, DL["qkv"], hb.EPI_SCALE_STORE, ssq_in=True, out=qkv, ldo=self.nqkv)
            a = pargs()
            a.q, a.ldq, a.q_off, a.k_off, a.v_off = hb.ptr(qkv), self.nqkv, 0, QH * HEAD_DIM, (QH + KVH) * HEAD_DIM
            a.q_heads, a.kv_heads, a.kv_cap, a.causal = QH, KVH, self.T, 1
            a.kc, a.vc = hb.ptr(self.k_self[i]), hb.ptr(self.v_self[i])
            a.P, a.p_plane_stride, a.p_ktiles = hb.ptr(pa), pa[0].numel(), akt
            hb.check(L.dia_dec_prefill_kv(C.byref(a), st), "dia_dec_prefill_kv")
            hb.check(L.dia_dec_prefill_attn(C.byref(a), st), "dia_dec_prefill_attn(self)")
            gemm(pa, akt, DL["o"], hb.EPI_RESID_EMIT, out=x, ldo=D, gnext=DL["g_ca"], P=px, p_kt=xkt, ssq_out=True)
            gemm(px, xkt, DL["cq"], hb.EPI_SCALE_STORE, ssq_in=True, out=qc, ldo=CH * HEAD_DIM)
            a = pargs()
            a.q, a.ldq, a.q_off = hb.ptr(qc), CH * HEAD_DIM, 0
            a.q_heads, a.kv_heads, a.kv_cap, a.causal = CH, CH, self.S, 0
            a.kc, a.vc, a.text_len = hb.ptr(self.k_cross[i]), hb.ptr(self.v_cross[i]), hb.ptr(self.text_len)
            a.P, a.p_plane_stride, a.p_ktiles = hb.ptr(pa), pa[0].numel(), akt
            hb.check(L.dia_dec_prefill_attn(C.byref(a), st), "dia_dec_prefill_attn(cross)")
            gemm(pa, akt, DL["co"], hb.EPI_RESID_EMIT, out=x, ldo=D, gnext=DL["g_mlp"], P=px, p_kt=xkt, ssq_out=True)
            gemm(px, xkt, DL["wi"], hb.EPI_SWIGLU_EMIT, ssq_in=True, P=ph, p_kt=hkt)
            gnext = w.dec_layers[i + 1]["g_sa"] if i + 1 < len(w.dec_layers) else w.dec_norm
            gemm(ph, hkt, DL["wo"], hb.EPI_RESID_EMIT, out=x, ldo=D, gnext=gnext, P=px, p_kt=xkt, ssq_out=True)
        cur = [fs if fs > 2 else 1 for fs in self.first_steps]
        self.cur.copy_(torch.tensor(cur, dtype=torch.int32))

    # ------------------------------------------------------------------ decode
    def ensure_noise(self, rows: int):
        """Make the Exp(1) rows [0, rows) resident (no-op for explicit noise / greedy sampling).  Step number s of
        an utterance reads row s - first_step, so `rows` = decode steps enqueued so far is always enough."""
        rows = min(int(rows), self.noise_steps)
        if self._gens is None or rows <= self._noise_rows:
            return
        r0, n = self._noise_rows, rows - self._noise_rows

        def draw(b):
            t = torch.empty(n, self.C, self.V, dtype=torch.float32)
            t.exponential_(1.0, generator=self._gens[b])
            return t

        if self.B > 1:
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(self.B, 8)) as pool:      # torch releases the GIL while it draws
                parts = list(pool.map(draw, range(self.B)))
        else:
            parts = [draw(0)]
        host = torch.stack(parts)
        with torch.cuda.stream(self.stream):
            self.noise[:, r0:rows].copy_(host, non_blocking=False)
        self._noise_rows = rows

    def decode(self, n_steps: int, use_graph: bool = True):
        if not self.prefilled:
            raise hb.DiaHipError("decode() before prefill()")
        self.ensure_noise(self._issued + int(n_steps))
        hb.check(hb.lib().dia_engine_decode(self._engine, int(n_steps), int(bool(use_graph))), "dia_engine_decode")
        self._issued += int(n_steps)

    def set_prefetch(self, lookahead: int):
        hb.check(hb.lib().dia_engine_set_prefetch(self._engine, int(lookahead)), "dia_engine_set_prefetch")

    def step_logits_only(self):
        hb.check(hb.lib().dia_engine_step_logits_only(self._engine), "dia_engine_step_logits_only")

    def profile_step(self) -> np.ndarray:
        """per-launch milliseconds of one eager decode step (HIP events on the engine's stream)."""
        n = hb.lib().dia_engine_launches_per_step(self._engine)
        buf = (C.c_float * n)()
        self.ensure_noise(self._issued + 1)
        hb.check(hb.lib().dia_engine_profile_step(self._engine, buf, n), "dia_engine_profile_step")
        self._issued += 1
        return np.array(buf[:], dtype=np.float64)

    def launches_per_step(self) -> int:
        return int(hb.lib().dia_engine_launches_per_step(self._engine))

    def seg_error(self) -> int:
        """0, or the code of a persistent-segment wait that timed out (synchronises the stream)"""
        if not self.seg:
            return 0
        return int(hb.lib().dia_seg_error(hb.ptr(self.seg_ws), C.c_void_p(self.stream.cuda_stream)))

    def time_step(self) -> np.ndarray:
        """kernel durations (milliseconds, launch order) of one eager decode step, each kernel bracketed by its own
        dispatch-level start / stop events — what rocprofv3 --kernel-trace reports per kernel."""
        n = hb.lib().dia_engine_launches_per_step(self._engine)
        buf, ivl = (C.c_float * n)(), (C.c_float * n)()
        self.ensure_noise(self._issued + 1)
        got = hb.lib().dia_engine_time_step(self._engine, buf, ivl, n)
        if got < 0:
            hb.check(got, "dia_engine_time_step")
        self._issued += 1
        self.last_kernel_names = [hb.lib().dia_timed_kernel_name(i).decode() for i in range(got)]
        self.last_intervals_ms = np.array(ivl[:got], dtype=np.float64)      # end of launch i-1 -> end of launch i
        return np.array(buf[:got], dtype=np.float64)

    def time_wi_launches(self, reps: int = 5) -> float:
        """Average seconds per launch of the dominant kernel — the wi_fused GEMV with SwiGLU epilogue —
        each launch bracketed by dispatch-level HIP start/stop events on the engine's stream
        (hipExtLaunchKernelGGL), cycling through every layer's matrix (1.2 GB, HBM-cold like in a real
        step) with exactly the arguments the engine uses.  Outputs go to planes_h, which the next step overwrites."""
        L = hb.lib()
        st = C.c_void_p(self.stream.cuda_stream)
        args = []
        for DL in self.w.dec_layers:
            g = hb.GemmArgs()
            g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(self.planes_x), self.planes_x[0].numel(), self.xkt, self.R
            g.W, g.KT, g.nstrips, g.epi = hb.ptr(DL["wi"].t), DL["wi"].kt, DL["wi"].ns, hb.EPI_SWIGLU_EMIT
            g.ssq_in, g.ssq_in_n, g.ssq_ld = hb.ptr(self.ssq), self.D // 16, self.rows_pad
            g.inv_d, g.eps = 1.0 / self.D, float(self.cfg.model.normalization_layer_epsilon)
            g.P, g.p_plane_stride, g.p_ktiles = hb.ptr(self.planes_h), self.planes_h[0].numel(), self.hkt
            g.act_f32 = 3 * self.act_f32             # as in the step
            g.w_planes = self.w.weight_planes
            args.append(g)
        for g in args:                                    # warm
            hb.check(L.dia_gemm(C.byref(g), st), "dia_gemm(wi)")
        self.stream.synchronize()
        ms = C.c_float()
        tot = 0.0
        for _ in range(reps):
            for g in args:
                hb.check(L.dia_gemm_timed(C.byref(g), st, C.byref(ms)), "dia_gemm_timed(wi)")
                tot += ms.value
        return tot * 1e-3 / (reps * len(args))

    def mlp_fused(self) -> bool:
        return bool(hb.lib().dia_engine_mlp_fused(self._engine))

    def time_mlp_launches(self, reps: int = 5) -> float:
        """Average seconds per launch of the fused MLP kernel (wi + wo, batch 1), dispatch-level events, cycling
        through every layer's matrices with the engine's own arguments (outputs are scratch: x is restored)."""
        L = hb.lib()
        st = C.c_void_p(self.stream.cuda_stream)
        x_keep = self.x.clone()
        pairs = []
        for i, DL in enumerate(self.w.dec_layers):
            a = hb.GemmArgs()
            a.A, a.a_plane_stride, a.a_ktiles, a.M = hb.ptr(self.planes_x), self.planes_x[0].numel(), self.xkt, self.R
            a.W, a.KT, a.nstrips, a.epi = hb.ptr(DL["wi"].t), DL["wi"].kt, DL["wi"].ns, hb.EPI_SWIGLU_EMIT
            a.ssq_in, a.ssq_in_n, a.ssq_ld = hb.ptr(self.ssq), self.D // 16, self.rows_pad
            a.inv_d, a.eps = 1.0 / self.D, float(self.cfg.model.normalization_layer_epsilon)
            a.P, a.p_plane_stride, a.p_ktiles = hb.ptr(self.planes_h), self.planes_h[0].numel(), self.hkt
            a.w_planes = self.w.weight_planes
            b = hb.GemmArgs()
            b.A, b.a_plane_stride, b.a_ktiles, b.M = hb.ptr(self.planes_h), self.planes_h[0].numel(), self.hkt, self.R
            b.W, b.KT, b.nstrips, b.epi = hb.ptr(DL["wo"].t), DL["wo"].kt, DL["wo"].ns, hb.EPI_RESID_EMIT
            b.ssq_ld, b.out, b.ldo = self.rows_pad, hb.ptr(self.x), self.D
            b.gnext = hb.ptr(self.w.dec_layers[i + 1]["g_sa"] if i + 1 < len(self.w.dec_layers) else self.w.dec_norm)
            b.P, b.p_plane_stride, b.p_ktiles, b.ssq_out = hb.ptr(self.planes_x), self.planes_x[0].numel(), self.xkt, hb.ptr(self.ssq)
            b.sk_scratch, b.sk_tickets, b.sk = hb.ptr(self.sk_scratch), hb.ptr(self.sk_tickets), 2
            b.w_planes = self.w.weight_planes
            pairs.append((a, b))
        bar = hb.ptr(self.mlp_barrier)
        for a, b in pairs:
            hb.check(L.dia_mlp_fused(C.byref(a), C.byref(b), bar, st), "dia_mlp_fused")
        self.stream.synchronize()
        ms = C.c_float()
        tot = 0.0
        for _ in range(reps):
            for a, b in pairs:
                hb.check(L.dia_mlp_fused_timed(C.byref(a), C.byref(b), bar, st, C.byref(ms)), "dia_mlp_fused_timed")
                tot += ms.value
        self.x.copy_(x_keep)
        return tot * 1e-3 / (reps * len(pairs))

    def sync(self):
        self.stream.synchronize()

    def steps_total(self) -> int:
        return self.max_tokens - self.prefill_step

    def run(self, use_graph: bool = True, poll: int = 64):
        """Run until every utterance has finished (EOS countdown or max_tokens); the host looks at
        the device-side `done` flags every `poll` steps only."""
        remaining = self.steps_total()
        first = True
        while remaining > 0:
            n = min(8 if first else poll, remaining)        # a short first chunk: its noise rows are the only ones drawn
            first = False                                   # before anything runs (0.23 ms of host time per row)
            self.decode(n, use_graph)
            remaining -= n
            self.ensure_noise(self._issued + min(poll, remaining))      # next chunk's noise is drawn while this one runs
            self.sync()
            if self.seg and self.seg_error():
                self.results()                               # raises with the give-up code
            if bool((self.fsm[:, 3] != 0).all().item()):
                break

    def logits_host(self) -> np.ndarray:
        """fp32 [B, 2, C, V] logits of the last executed step."""
        self.sync()
        lg = self.logits[: self.R, : self.C * self.V].reshape(self.B, 2, self.C, self.V)
        return lg.cpu().numpy()

    def results(self) -> List[UtteranceResult]:
        self.sync()
        if int(self.mlp_barrier[1].item()) != 0:
            raise hb.DiaHipError("fused MLP kernel: a workgroup gave up waiting at the grid barrier; results are invalid")
        code = self.seg_error()
        if code:
            self.seg_ws[: int(hb.lib().dia_seg_workspace_control_bytes())].zero_()       # counters are inconsistent after a give-up
            raise hb.DiaHipError(f"persistent MLP segment: an in-kernel wait timed out (code {code}: 1 x1, 2 hidden, 3 wo partials, "
                                 f"4 x2) — were all 256 workgroups resident?  Results are invalid; set DIA_TUNE=seg=0 to run the launches")
        tok = self.tokens.cpu().numpy()
        prd = self.pred.cpu().numpy()
        fsm = self.fsm.cpu().numpy()
        cur = self.cur.cpu().numpy()
        out = []
        for b in range(self.B):
            last = int(fsm[b, 4]) if fsm[b, 3] else int(cur[b]) - 1
            out.append(UtteranceResult(tok[b], tok[b, self.first_steps[b]: last + 1].copy(), last, prd[b], self.lens[b]))   # model.py:831
        return out

    # ------------------------------------------------------------------ accounting
    def step_bytes(self, n_keys: Optional[int] = None) -> int:
        """Algorithmic HBM bytes of one decode step at self-KV length `n_keys` (default: current)."""
        d = self.cfg.model.decoder
        kvb = 2 if self.kv_code == hb.KV_BF16 else 4       # fp32, or two bf16 planes
        if n_keys is None:
            n_keys = int(self.cur.max().item())
        kv_self = 2 * d.n_layer * 2 * d.kv_heads * HEAD_DIM * kvb * n_keys      # both rows, K and V
        kv_cross = d.n_layer * 2 * d.cross_query_heads * HEAD_DIM * kvb          # cond row, per text byte
        return self.w.decode_weight_bytes() + self.B * kv_self + kv_cross * sum(self.lens)
