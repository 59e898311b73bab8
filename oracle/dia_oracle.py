"""CPU oracle for the Dia decode path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; the product package (``dia-tts-prune_amd/``) never does, and its HIP
path raises when the extension is missing instead of falling back to anything here.

What it is: a float32 torch-CPU restatement of the reference's algorithm for the
``Dia.generate()`` path (one utterance = the reference's hard-coded CFG row pair), written
from the reference's behaviour, each function citing the reference file:line it follows.
The arithmetic of the reference lives in a third-party dependency, PyTorch (pinned
``torch==2.6.0`` in the reference's pyproject.toml:19; this image has 2.10.0+rocm7.0 CPU),
so the restatement uses the same library for matmul / exp / RNG.

Parity pin: the reference holds no golden vectors or tests.  This oracle is
pinned against the reference itself, imported in the build container with the three
import-time shims S1-S3 of SURVEY.md App. A (``tests/golden/make_golden.py``): logits agree
to <= 1e-5 max-abs and token buffers bit-exactly on the tiny and mid configs; the vectors
are committed under ``tests/golden/`` and re-checked by ``tests/test_oracle_golden.py``.

As-written quirks of the reference that are reproduced on purpose:
RoPE position of cache slot t is t+1; the uncond row's cross-attention is fully masked and
contributes 0; prefill rows 1..14 hold PAD so the first 14 predictions are discarded;
CFG is ``cond + s*(cond-uncond)``.

Two execution modes:
  * ``mirror=True``  – op-for-op the reference's sequence, including the per-step cross
    K/V re-projection + RoPE whose result is discarded (layers.py:273-279 vs 284-287) and
    the padded 2x1024 encoder pass.  This is "the reference CPU path" for timing.
  * ``mirror=False`` – lean: same results, dead work skipped, encoder on the packed
    non-pad tokens of the cond row only (exact).
"""

from __future__ import annotations

import math
import time
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------------------
# small config view (works with the product DiaConfig or the reference's pydantic object)
# --------------------------------------------------------------------------------------

@dataclass
class Dims:
    S: int            # text_length
    T: int            # audio_length
    C: int            # channels
    text_pad: int
    eos: int
    pad: int
    bos: int
    delay: List[int]
    enc_layers: int
    enc_d: int
    enc_ffn: int
    enc_heads: int
    enc_hd: int
    dec_layers: int
    dec_d: int
    dec_ffn: int
    q_heads: int
    kv_heads: int
    hd: int
    cq_heads: int
    chd: int
    src_vocab: int
    tgt_vocab: int
    eps: float
    rope_min: int
    rope_max: int

    @classmethod
    def of(cls, cfg) -> "Dims":
        m, e, d, da = cfg.model, cfg.model.encoder, cfg.model.decoder, cfg.data
        return cls(
            S=da.text_length, T=da.audio_length, C=da.channels, text_pad=da.text_pad_value,
            eos=da.audio_eos_value, pad=da.audio_pad_value, bos=da.audio_bos_value,
            delay=list(da.delay_pattern),
            enc_layers=e.n_layer, enc_d=e.n_embd, enc_ffn=e.n_hidden, enc_heads=e.n_head, enc_hd=e.head_dim,
            dec_layers=d.n_layer, dec_d=d.n_embd, dec_ffn=d.n_hidden, q_heads=d.gqa_query_heads,
            kv_heads=d.kv_heads, hd=d.gqa_head_dim, cq_heads=d.cross_query_heads, chd=d.cross_head_dim,
            src_vocab=m.src_vocab_size, tgt_vocab=m.tgt_vocab_size, eps=m.normalization_layer_epsilon,
            rope_min=m.rope_min_timescale, rope_max=m.rope_max_timescale,
        )


# --------------------------------------------------------------------------------------
# text / token-layout helpers (integer work)
# --------------------------------------------------------------------------------------

def effective_text(text: str, audio_prompt_text: Optional[str] = None) -> str:
    """reference model.py:686-696 — strip, then append the opposite speaker tag."""
    t = (audio_prompt_text.strip() + " " + text.strip()) if audio_prompt_text else text.strip()
    s1, s2 = t.rfind("[S1]"), t.rfind("[S2]")
    if s1 > s2 and not t.endswith("[S2]"):
        t += " [S2]"
    elif s2 > s1 and not t.endswith("[S1]"):
        t += " [S1]"
    elif s1 == -1 and s2 == -1 and t:
        t += " [S2]"
    return t


def text_tokens(text: str, dm: Dims) -> np.ndarray:
    """reference model.py:254-289 — UTF-8 bytes, [S1]->1, [S2]->2, truncate, pad. int64 [S]."""
    b = text.encode("utf-8").replace(b"[S1]", b"\x01").replace(b"[S2]", b"\x02")
    ids = list(b)[: dm.S]
    out = np.full((dm.S,), dm.text_pad, dtype=np.int64)
    out[: len(ids)] = ids
    return out


def delayed_prefill(dm: Dims, prompt: Optional[np.ndarray] = None):
    """reference model.py:291-353 with audio.py:6-85.

    One BOS row (+ prompt rows) + max_delay PAD rows, then out[t,c] = in[t-d_c,c] with BOS
    where t-d_c < 0 (PAD where >= T never triggers: t-d_c < T always).  Returns
    (int32 [T0+max_delay, C], prefill_step)."""
    md = max(dm.delay)
    rows = [np.full((1, dm.C), dm.bos, dtype=np.int32)]
    step = 1
    if prompt is not None:
        rows.append(np.asarray(prompt, dtype=np.int32))
        step += prompt.shape[0]
    rows.append(np.full((md, dm.C), dm.pad, dtype=np.int32))
    src = np.concatenate(rows, axis=0)
    Tn = src.shape[0]
    out = np.empty_like(src)
    for c, d in enumerate(dm.delay):
        for t in range(Tn):
            ts = t - d
            if ts < 0:
                out[t, c] = dm.bos
            elif ts >= Tn:
                out[t, c] = dm.pad
            else:
                out[t, c] = src[ts, c]
    return out, step


def revert_delay_and_trim(codes: np.ndarray, dm: Dims, codebook_size: int = 1024) -> np.ndarray:
    """reference audio.py:88-163 + model.py:498-533.

    out[t,c] = in[min(t+d_c, T-1), c] (PAD if t+d_c >= T, which the clamp makes unreachable),
    drop the last max_delay rows, codes outside [0, codebook_size-1] -> 0, transpose to
    [1, C, T'] (what the codec receives)."""
    Tn = codes.shape[0]
    md = max(dm.delay)
    out = np.empty_like(codes)
    for c, d in enumerate(dm.delay):
        for t in range(Tn):
            ts = min(t + d, Tn - 1)
            out[t, c] = dm.pad if ts >= Tn else codes[ts, c]
    out = out[: Tn - md] if Tn - md > 0 else out[:0]
    out = out.copy()
    out[(out < 0) | (out > codebook_size - 1)] = 0
    return out.T[None, :, :]


# --------------------------------------------------------------------------------------
# float building blocks
# --------------------------------------------------------------------------------------

def rope_inv_freq(head_dim: int, min_ts: int, max_ts: int) -> torch.Tensor:
    """reference layers.py:126-132."""
    half = head_dim // 2
    fraction = (2.0 * torch.arange(0, half)) / head_dim
    return (1.0 / (min_ts * (max_ts / min_ts) ** fraction)).to(torch.float32)


def rope(x: torch.Tensor, pos: torch.Tensor, inv_freq: torch.Tensor) -> torch.Tensor:
    """Half-split rotary embedding = reference layers.py:135-173 under shim S1
    (freqs = position[...,None,None] * inv_freq).  x [B,T,N,H], pos [B,T] (int32 or fp32)."""
    f = pos.unsqueeze(-1).unsqueeze(-1) * inv_freq
    sin, cos = torch.sin(f.to(torch.float32)), torch.cos(f.to(torch.float32))
    h = x.shape[-1] // 2
    x1, x2 = x[..., :h].to(torch.float32), x[..., h:].to(torch.float32)
    return torch.cat((x1 * cos - x2 * sin, x1 * sin + x2 * cos), dim=-1)


def dense(x: torch.Tensor, w: torch.Tensor, n_in: int = 1) -> torch.Tensor:
    """reference layers.py:55-66 — contract the last ``n_in`` axes of x with the first
    ``n_in`` axes of the kernel."""
    k = 1
    for v in w.shape[:n_in]:
        k *= v
    lead = x.shape[: x.dim() - n_in]
    y = x.reshape(-1, k) @ w.reshape(k, -1)
    return y.reshape(*lead, *w.shape[n_in:])


def rmsnorm(x: torch.Tensor, g: torch.Tensor, eps: float) -> torch.Tensor:
    """torch.nn.RMSNorm as used at reference layers.py:393,411,461,541,560,579,714."""
    return F.rms_norm(x.to(torch.float32), (x.shape[-1],), g, eps)


def swiglu_mlp(x: torch.Tensor, wi: torch.Tensor, wo: torch.Tensor) -> torch.Tensor:
    """reference layers.py:92-105."""
    f = dense(x, wi)                      # [..., 2, F]
    return dense(F.silu(f[..., 0, :]) * f[..., 1, :], wo)


def sdpa(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, mask: Optional[torch.Tensor]) -> torch.Tensor:
    """softmax(q k^T / sqrt(H) [+mask]) v with q [B,N,Tq,H], k/v [B,N,Tk,H], mask bool
    broadcastable to [B,N,Tq,Tk] (True = attend).  A query row whose mask is all False
    yields 0 — the behaviour of F.scaled_dot_product_attention the reference relies on at
    layers.py:329-337 for the uncond row."""
    s = (q @ k.transpose(-1, -2)) * (1.0 / math.sqrt(q.shape[-1]))
    if mask is not None:
        s = s.masked_fill(~mask, float("-inf"))
        dead = ~mask.any(dim=-1, keepdim=True)
        s = s.masked_fill(dead.expand_as(s), 0.0)
    p = torch.softmax(s, dim=-1)
    if mask is not None:
        p = p.masked_fill(dead.expand_as(p), 0.0)
    return p @ v


# --------------------------------------------------------------------------------------
# encoder + cross K/V
# --------------------------------------------------------------------------------------

def encoder_forward(w: Dict[str, torch.Tensor], dm: Dims, ids: torch.Tensor, pos: torch.Tensor,
                    mask: Optional[torch.Tensor]) -> torch.Tensor:
    """reference layers.py:385-462.  ids [B,T] int64, pos [B,T] fp32, mask [B,1,T,T] or None."""
    inv = rope_inv_freq(dm.enc_hd, dm.rope_min, dm.rope_max)
    x = w["encoder.embedding.weight"][ids]
    for i in range(dm.enc_layers):
        p = f"encoder.layers.{i}."
        h = rmsnorm(x, w[p + "pre_sa_norm.weight"], dm.eps)
        q = rope(dense(h, w[p + "self_attention.q_proj.weight"]), pos, inv)
        k = rope(dense(h, w[p + "self_attention.k_proj.weight"]), pos, inv)
        v = dense(h, w[p + "self_attention.v_proj.weight"])
        a = sdpa(q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2), mask).transpose(1, 2)
        x = x + dense(a, w[p + "self_attention.o_proj.weight"], n_in=2)
        h = rmsnorm(x, w[p + "post_sa_norm.weight"], dm.eps)
        x = x + swiglu_mlp(h, w[p + "mlp.wi_fused.weight"], w[p + "mlp.wo.weight"])
    return rmsnorm(x, w["encoder.norm.weight"], dm.eps)


def encoder_mask(ids: torch.Tensor, dm: Dims) -> torch.Tensor:
    """reference state.py:8-39,60-61: attend iff query and key are both non-pad or both pad."""
    nonpad = ids != dm.text_pad
    q, k = nonpad.unsqueeze(2), nonpad.unsqueeze(1)
    return ((q & k) | (~q & ~k)).unsqueeze(1)


def cross_kv(w: Dict[str, torch.Tensor], dm: Dims, enc_out: torch.Tensor, pos: torch.Tensor):
    """reference layers.py:632-669 — per layer K = RoPE(k_proj(enc_out), enc pos), V = v_proj(enc_out),
    both returned as [B, N, T, H]."""
    inv = rope_inv_freq(dm.chd, dm.rope_min, dm.rope_max)
    out = []
    for i in range(dm.dec_layers):
        p = f"decoder.layers.{i}.cross_attention."
        k = rope(dense(enc_out, w[p + "k_proj.weight"]), pos, inv).transpose(1, 2)
        v = dense(enc_out, w[p + "v_proj.weight"]).transpose(1, 2)
        out.append((k.contiguous(), v.contiguous()))
    return out


# --------------------------------------------------------------------------------------
# decoder state + one decode step
# --------------------------------------------------------------------------------------

@dataclass
class DecodeState:
    dm: Dims
    mirror: bool
    L: int                                   # non-pad text bytes of the cond row
    cross: list                              # per layer (K, V): mirror [2,N,S,H]; lean [1,N,L,H] (cond row)
    cross_mask: Optional[torch.Tensor]       # mirror: bool [2,1,1,S]
    enc_out: Optional[torch.Tensor]          # mirror only (feeds the dead re-projection)
    enc_pos: Optional[torch.Tensor]
    self_k: list = field(default_factory=list)   # per layer [2, kv, T, H]
    self_v: list = field(default_factory=list)
    n_cached: int = 0


def prepare(w: Dict[str, torch.Tensor], dm: Dims, ids_cond: np.ndarray, mirror: bool) -> DecodeState:
    """reference model.py:355-427 (no audio prompt): encoder on [uncond(all pad), cond],
    cross K/V precompute, zeroed self-KV caches (state.py:126-162)."""
    ids_cond_t = torch.from_numpy(np.asarray(ids_cond, dtype=np.int64))[None, :]
    nonpad = ids_cond_t[0] != dm.text_pad
    if not bool(torch.equal(nonpad, torch.arange(dm.S) < int(nonpad.sum()))):
        # text_pad (0) cannot occur inside UTF-8 text after tag replacement; keep the fast
        # packed path honest
        raise ValueError("cond text tokens must be a non-pad prefix followed by padding")
    L = int(nonpad.sum())
    if mirror:
        ids = torch.cat([torch.full_like(ids_cond_t, dm.text_pad), ids_cond_t], dim=0)      # model.py:360-362
        pos = torch.arange(dm.S, dtype=torch.float32).unsqueeze(0).expand(2, -1)           # state.py:57-59
        enc = encoder_forward(w, dm, ids, pos, encoder_mask(ids, dm))
        cross = cross_kv(w, dm, enc, pos)
        cmask = (ids != dm.text_pad)[:, None, None, :]                                      # state.py:139-140
        st = DecodeState(dm, True, L, cross, cmask, enc, pos)
    else:
        ids = ids_cond_t[:, :L]
        pos = torch.arange(L, dtype=torch.float32).unsqueeze(0)
        enc = encoder_forward(w, dm, ids, pos, None)
        st = DecodeState(dm, False, L, cross_kv(w, dm, enc, pos), None, None, None)
    for _ in range(dm.dec_layers):
        st.self_k.append(torch.zeros(2, dm.kv_heads, dm.T, dm.hd))
        st.self_v.append(torch.zeros(2, dm.kv_heads, dm.T, dm.hd))
    return st


def decode_step(w: Dict[str, torch.Tensor], st: DecodeState, tokens_c: np.ndarray, position: int) -> torch.Tensor:
    """reference layers.py:671-720 driven as at model.py:755-759: both CFG rows get the same
    9 tokens, RoPE position ``position``, K/V appended at slot ``st.n_cached``.
    Returns fp32 logits [2, C, V]."""
    dm = st.dm
    tok = torch.as_tensor(np.asarray(tokens_c, dtype=np.int64))
    x = None
    for c in range(dm.C):                                                  # layers.py:691-696 (sequential sum)
        e = w[f"decoder.embeddings.{c}.weight"][tok[c]]
        x = e if x is None else x + e
    x = x[None, None, :].expand(2, 1, -1).contiguous()
    pos = torch.full((2, 1), position, dtype=torch.int32)                 # state.py:164-169
    inv_s = rope_inv_freq(dm.hd, dm.rope_min, dm.rope_max)
    inv_c = rope_inv_freq(dm.chd, dm.rope_min, dm.rope_max)
    slot = st.n_cached
    g = dm.q_heads // dm.kv_heads
    for i in range(dm.dec_layers):
        p = f"decoder.layers.{i}."
        # --- self attention (layers.py:541-555, 238-346; cache update state.py:99-103)
        h = rmsnorm(x, w[p + "pre_sa_norm.weight"], dm.eps)
        q = rope(dense(h, w[p + "self_attention.q_proj.weight"]), pos, inv_s)
        k = rope(dense(h, w[p + "self_attention.k_proj.weight"]), pos, inv_s)
        v = dense(h, w[p + "self_attention.v_proj.weight"])
        st.self_k[i][:, :, slot, :] = k[:, 0]
        st.self_v[i][:, :, slot, :] = v[:, 0]
        kk = st.self_k[i][:, :, : slot + 1].repeat_interleave(g, dim=1)   # layers.py:319-320
        vv = st.self_v[i][:, :, : slot + 1].repeat_interleave(g, dim=1)
        a = sdpa(q.transpose(1, 2), kk, vv, None).transpose(1, 2)
        x = x + dense(a, w[p + "self_attention.o_proj.weight"], n_in=2)
        # --- cross attention (layers.py:560-574)
        h = rmsnorm(x, w[p + "pre_ca_norm.weight"], dm.eps)
        q = rope(dense(h, w[p + "cross_attention.q_proj.weight"]), pos, inv_c)
        if st.mirror:
            # the reference projects + rotates the whole encoder output here and then uses the
            # cache instead (layers.py:273-279 vs 284-287); kept for timing fidelity only
            _dk = rope(dense(st.enc_out, w[p + "cross_attention.k_proj.weight"]), st.enc_pos, inv_c)
            _dv = dense(st.enc_out, w[p + "cross_attention.v_proj.weight"])
            ck, cv = st.cross[i]
            a = sdpa(q.transpose(1, 2), ck, cv, st.cross_mask).transpose(1, 2)
        else:
            ck, cv = st.cross[i]                                           # [1,N,L,H]: cond row only
            a_c = sdpa(q[1:2].transpose(1, 2), ck, cv, None).transpose(1, 2)
            a = torch.cat([torch.zeros_like(a_c), a_c], dim=0)            # uncond row: fully masked -> 0
        x = x + dense(a, w[p + "cross_attention.o_proj.weight"], n_in=2)
        # --- MLP (layers.py:579-582)
        h = rmsnorm(x, w[p + "pre_mlp_norm.weight"], dm.eps)
        x = x + swiglu_mlp(h, w[p + "mlp.wi_fused.weight"], w[p + "mlp.wo.weight"])
    st.n_cached = slot + 1
    h = rmsnorm(x, w["decoder.norm.weight"], dm.eps)
    return dense(h, w["decoder.logits_dense.weight"])[:, 0].to(torch.float32)   # [2, C, V]


# --------------------------------------------------------------------------------------
# CFG + constraints + sampling
# --------------------------------------------------------------------------------------

def guided_logits(logits_2cv: torch.Tensor, cfg_scale: float, dm: Dims) -> torch.Tensor:
    """reference model.py:447-478: cond + s*(cond-uncond); -inf on EOS for channels >= 1 and on
    PAD/BOS for all channels."""
    un, co = logits_2cv[0], logits_2cv[1]
    lg = co + cfg_scale * (co - un)
    if lg.shape[0] > 1:
        lg[1:, dm.eos] = float("-inf")
    lg[:, dm.pad] = float("-inf")
    lg[:, dm.bos] = float("-inf")
    if dm.tgt_vocab <= dm.eos + 1:
        lg[:, dm.tgt_vocab:] = float("-inf")
    return lg


def sample_next_token(lg: torch.Tensor, temperature: float, top_p: float, top_k: Optional[int],
                      noise: Optional[torch.Tensor] = None,
                      generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """reference model.py:32-82.  ``noise`` (Exp(1) variates, same shape as lg) switches the
    final draw to argmax(p / noise), which is what torch.multinomial(p, 1) computes on CPU
    from the same generator stream (SURVEY.md §7 'Hard parts'); with ``noise=None`` the
    draw is torch.multinomial itself."""
    if temperature == 0.0:
        return torch.argmax(lg, dim=-1)
    lg = lg / temperature
    if top_k is not None and top_k > 0:
        kth = torch.topk(lg, k=top_k, dim=-1).values[..., -1:]
        lg = lg.masked_fill(lg < kth, float("-inf"))
    if top_p < 1.0:
        pr = torch.softmax(lg, dim=-1)
        sp, si = torch.sort(pr, dim=-1, descending=True)
        rm = torch.cumsum(sp, dim=-1) > top_p
        rm = torch.roll(rm, shifts=1, dims=-1)
        rm[..., 0] = False
        drop = torch.zeros_like(rm).scatter(dim=-1, index=si, src=rm)
        lg = lg.masked_fill(drop, float("-inf"))
    pr = torch.softmax(lg, dim=-1)
    if torch.all(torch.isclose(pr.sum(dim=-1), torch.tensor(0.0))):
        return torch.argmax(lg, dim=-1)
    if noise is not None:
        return torch.argmax(pr / noise, dim=-1)
    return torch.multinomial(pr, num_samples=1, generator=generator).squeeze(-1)


def exp_noise(seed: int, steps: int, C: int, V: int) -> torch.Tensor:
    """The Exp(1) variates torch.multinomial consumes, step by step, after
    ``torch.manual_seed(seed)`` (reference model.py:679-683): one [C,V] draw per step."""
    g = torch.Generator().manual_seed(seed)
    out = torch.empty(steps, C, V, dtype=torch.float32)
    for s in range(steps):
        out[s].exponential_(1.0, generator=g)
    return out


# --------------------------------------------------------------------------------------
# the generate loop (token FSM)
# --------------------------------------------------------------------------------------

@dataclass
class GenResult:
    tokens: np.ndarray                 # int32 [T, C] — DecoderOutput.generated_tokens
    prefill_step: int
    last_step: int                     # dec_step at loop exit
    codes: np.ndarray                  # rows [prefill_step : last_step+1] (model.py:831)
    logits: List[np.ndarray]           # per executed step, fp32 [2,C,V] (if keep_logits)
    preds: List[np.ndarray]            # per executed step, sampled tokens before the EOS FSM
    step_ms: List[float]
    prep_s: float


def generate(w: Dict[str, torch.Tensor], cfg, text: str, *, max_tokens: Optional[int] = None,
             cfg_scale: float = 3.0, temperature: float = 1.3, top_p: float = 0.95,
             cfg_filter_top_k: int = 35, seed: Optional[int] = None, mirror: bool = False,
             noise: Optional[torch.Tensor] = None, forced_tokens: Optional[np.ndarray] = None,
             keep_logits: bool = True, max_steps: Optional[int] = None,
             ignore_eos: bool = False, audio_prompt: Optional[np.ndarray] = None,
             audio_prompt_text: Optional[str] = None) -> GenResult:
    """reference model.py:631-846, up to (not including) the codec.

    ``audio_prompt`` [Tp, C] codes (model.py:311-353, 406-422): SEMANTIC DECISION, parity unpinned for this
    argument.  The reference cannot run it (KVCache.prefill returns nothing, state.py:105-109 vs
    layers.py:297 — SURVEY.md App. A4) and what it sketches is inconsistent with its own decode loop (prefill
    positions r for token row r where decode uses r+1; current_idx = len-1 so the first decoded step would
    overwrite the last prefilled slot — App. B6).  Here the prompt rows are REPLAYED through the decode step:
    token row r -> cache slot r at RoPE position r+1 for r = 0..P-2 (P = 1 + Tp), exactly what the loop does
    for every generated row, with nothing sampled or written and no RNG draw; generation proper then starts
    at step P with the reference's own bookkeeping (bos_countdown, masked write, slice [P : dec_step+1]).
    The token-buffer preparation itself (model.py:291-353) runs in the reference and is pinned by
    tests/golden/ref_prompt.npz.

    ``noise`` [steps,C,V]: explicit Exp(1) variates (argmax(p/q) draw); else the torch global
    generator seeded with ``seed`` is used through torch.multinomial exactly like the
    reference.  ``forced_tokens`` [T,C]: teacher forcing — the token written at each step is
    taken from this buffer instead of the sample (the sample is still recorded in ``preds``).
    ``max_steps`` stops the loop early (timing on a bounded sample).  ``ignore_eos`` disables
    the natural-EOS trigger (perf runs on random weights)."""
    dm = Dims.of(cfg)
    if seed is not None:
        torch.manual_seed(seed)
        np.random.seed(seed)
    text = effective_text(text, audio_prompt_text)
    T = dm.T if max_tokens is None else max_tokens
    md = max(dm.delay)
    t0 = time.time()
    st = prepare(w, dm, text_tokens(text, dm), mirror)
    prefill, prefill_step = delayed_prefill(dm, audio_prompt)
    tokens = np.full((dm.T, dm.C), -1, dtype=np.int32)                     # state.py:178-188
    tokens[: prefill.shape[0]] = prefill                                  # state.py:205-208
    for cur in range(1, prefill_step):                                     # prompt replay (see docstring)
        decode_step(w, st, tokens[cur - 1], cur)
    prep_s = time.time() - t0

    dec_step = prefill_step - 1
    bos_countdown, eos_detected, eos_countdown = md, False, -1
    logits_log, preds_log, step_ms = [], [], []
    n_done = 0
    while dec_step < T - 1:
        cur = dec_step + 1
        t1 = time.time()
        lg2 = decode_step(w, st, tokens[cur - 1], cur)
        if keep_logits:
            logits_log.append(lg2.numpy().copy())
        lg = guided_logits(lg2.clone(), cfg_scale, dm)
        nz = None if noise is None else noise[n_done]
        pred = sample_next_token(lg.to(torch.float32), temperature, top_p, cfg_filter_top_k, noise=nz).numpy().astype(np.int64)
        preds_log.append(pred.copy())
        step_ms.append((time.time() - t1) * 1e3)
        if forced_tokens is not None:
            pred = forced_tokens[cur].astype(np.int64).copy()
        else:
            # EOS handling, model.py:771-788
            if not eos_detected and pred[0] == dm.eos and not ignore_eos:
                eos_detected, eos_countdown = True, md
            if eos_countdown > 0:
                after = md - eos_countdown
                for i, d in enumerate(dm.delay):
                    if after == d:
                        pred[i] = dm.eos
                    elif after > d and pred[i] != dm.eos:
                        pred[i] = dm.pad
                eos_countdown -= 1
        # masked write, model.py:791-792 + state.py:195-203
        bos_countdown = max(0, bos_countdown - 1)
        if forced_tokens is not None:
            tokens[cur] = pred
        elif bos_countdown > 0:
            m = tokens[cur] == -1
            tokens[cur] = np.where(m, pred, tokens[cur])
        else:
            tokens[cur] = pred
        n_done += 1
        if forced_tokens is None:
            if eos_countdown == 0:
                break
            if cur >= T - md - 1 and not eos_detected:
                eos_detected, eos_countdown = True, md
        dec_step += 1
        if max_steps is not None and n_done >= max_steps:
            break
    codes = tokens[prefill_step: dec_step + 1].copy()
    return GenResult(tokens, prefill_step, dec_step, codes, logits_log, preds_log, step_ms, prep_s)
