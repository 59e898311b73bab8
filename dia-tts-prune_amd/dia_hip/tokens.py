"""Integer host work around the decode path: text -> byte tokens, delay-pattern tables.

Vectorised NumPy; the arrays are tiny ([16,9], [T,9]) so none of this goes to the GPU.
"""

from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from .config import DiaConfig


def effective_text(text: str, audio_prompt_text: Optional[str] = None) -> str:
    """Prompt assembly of ``Dia.generate`` (reference dia/model.py:686-696): strip, prepend the
    prompt transcript, and close with the speaker tag opposite to the last one used."""
    t = f"{audio_prompt_text.strip()} {text.strip()}" if audio_prompt_text else text.strip()
    i1, i2 = t.rfind("[S1]"), t.rfind("[S2]")
    if i1 > i2:
        if not t.endswith("[S2]"):
            t += " [S2]"
    elif i2 > i1:
        if not t.endswith("[S1]"):
            t += " [S1]"
    elif t:                      # no tag at all (both -1)
        t += " [S2]"
    return t


def encode_text(text: str, cfg: DiaConfig) -> np.ndarray:
    """Byte tokens of the *non-pad prefix* (reference dia/model.py:254-289: UTF-8 bytes, ``[S1]``->1,
    ``[S2]``->2, truncated to text_length).  The reference pads with text_pad_value up to text_length;
    the pad tail carries no information for the cond row and is not materialised."""
    raw = text.encode("utf-8").replace(b"[S1]", b"\x01").replace(b"[S2]", b"\x02")
    n = cfg.data.text_length
    if len(raw) > n:
        print(f"Warning: Input text truncated from {len(raw)} to {n} bytes.")
        raw = raw[:n]
    return np.frombuffer(raw, dtype=np.uint8).astype(np.int32)


SYNTH_SENTENCE = "[S1] Dia is an open weights text to dialogue model. [S2] You get full control over scripts and voices. "


def synthetic_text(n_bytes: int, cfg: DiaConfig) -> str:
    """A dialogue prompt that encodes to exactly `n_bytes` byte tokens after ``effective_text`` (the fixed
    [S1]/[S2] sentence of SURVEY.md §8d repeated and cut, filled up with '.' where the cut would fall short:
    tags count one byte each, the closing tag appended by ``generate`` two).  Bench / test input only."""
    if n_bytes < 8:
        raise ValueError("synthetic_text: at least 8 bytes")
    if n_bytes > cfg.data.text_length:
        raise ValueError("synthetic_text: longer than text_length")
    reps = SYNTH_SENTENCE * (n_bytes // 40 + 2)
    best = None
    for cut in range(len(reps), 3, -1):
        t = reps[:cut].rstrip()
        if t.endswith("]") or t.endswith("[") or t[-2:] in ("[S", "S1", "S2"):
            continue                                 # never end inside or on a tag
        n = len(effective_text(t).encode("utf-8").replace(b"[S1]", b"\x01").replace(b"[S2]", b"\x02"))    # untruncated
        if n <= n_bytes:
            best = t + "." * (n_bytes - n)
            break
    if best is None or len(encode_text(effective_text(best), cfg)) != n_bytes:
        raise ValueError(f"synthetic_text: cannot build {n_bytes} bytes")
    return best


def padded_text_ids(ids: np.ndarray, cfg: DiaConfig) -> np.ndarray:
    out = np.full((cfg.data.text_length,), cfg.data.text_pad_value, dtype=np.int64)
    out[: len(ids)] = ids
    return out


def delayed_prefill(cfg: DiaConfig, prompt: Optional[np.ndarray] = None) -> Tuple[np.ndarray, int]:
    """BOS row (+ prompt rows) + max_delay PAD rows with the delay pattern applied:
    ``out[t, c] = in[t - d_c, c]``, BOS where ``t - d_c < 0`` (reference dia/model.py:291-353 and
    dia/audio.py:6-85).  Returns (int32 [rows, C], prefill_step)."""
    da = cfg.data
    delay = np.asarray(da.delay_pattern, dtype=np.int64)
    parts = [np.full((1, da.channels), da.audio_bos_value, dtype=np.int32)]
    if prompt is not None:
        parts.append(np.asarray(prompt, dtype=np.int32).reshape(-1, da.channels))
    step = sum(p.shape[0] for p in parts)
    parts.append(np.full((int(delay.max()), da.channels), da.audio_pad_value, dtype=np.int32))
    src = np.concatenate(parts, axis=0)
    n = src.shape[0]
    t_src = np.arange(n)[:, None] - delay[None, :]                     # [rows, C]
    gathered = src[np.clip(t_src, 0, n - 1), np.arange(da.channels)[None, :]]
    out = np.where(t_src < 0, da.audio_bos_value, np.where(t_src >= n, da.audio_pad_value, gathered))
    return out.astype(np.int32), step


def codes_for_codec(codes: np.ndarray, cfg: DiaConfig, codebook_size: int = 1024) -> np.ndarray:
    """Undo the delay pattern and trim, i.e. what ``Dia._generate_output`` hands to the codec
    (reference dia/audio.py:88-163 + dia/model.py:498-533): ``out[t,c] = in[min(t+d_c, T-1), c]``,
    drop the last max_delay rows, out-of-codebook ids -> 0, layout [1, C, T']."""
    da = cfg.data
    delay = np.asarray(da.delay_pattern, dtype=np.int64)
    n = codes.shape[0]
    if n == 0:
        return np.zeros((1, da.channels, 0), dtype=codes.dtype)
    t_src = np.minimum(np.arange(n)[:, None] + delay[None, :], n - 1)
    out = codes[t_src, np.arange(da.channels)[None, :]]
    keep = max(n - int(delay.max()), 0)
    out = out[:keep].copy()
    out[(out < 0) | (out > codebook_size - 1)] = 0
    return np.ascontiguousarray(out.T)[None]
