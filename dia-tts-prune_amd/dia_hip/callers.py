"""Host-side callers above ``Dia.generate``: the text chunking of the reference's
Gradio front-end and its chunk chaining, where every batch of chunks is generated with the previous
batch as the audio prompt (reference app.py:79-122 helpers, app.py:200-248 loop).

The reference chains through a WAV file (decode -> write -> ``load_audio`` -> encode).  Here the chain
carries the *codes*: the codec input of batch i, transposed to [T, C], is the prompt of batch i+1 — the
same information without the codec round trip, so it also works offline where no codec exists.  Audio is
produced at the end if (and only if) a codec is loaded.
"""

from __future__ import annotations

from typing import Iterable, Iterator, List, Optional, Sequence

import numpy as np

DEFAULT_SAMPLE_RATE = 44100


def count_effective_length(text: str) -> int:
    """length with ``[S1]`` / ``[S2]`` counted as one character each (app.py:79-81)"""
    return len(text.replace("[S1]", "¤").replace("[S2]", "¤"))


def auto_adjust_chunk_size(text: str, user_chunk_size: int) -> int:
    """a positive user value wins; otherwise 48 / 64 / 96 effective characters by input size (app.py:83-97)"""
    if user_chunk_size > 0:
        return int(user_chunk_size)
    n = count_effective_length(text)
    return 48 if n <= 1024 else (64 if n <= 4096 else 96)


def split_by_words_respecting_special_tokens(text: str, max_effective_chars: int = 64) -> List[str]:
    """greedy word packing up to ``max_effective_chars`` effective characters; a single over-long word
    becomes its own chunk (app.py:100-122)"""
    chunks: List[str] = []
    cur = ""
    for word in text.split():
        cand = f"{cur} {word}".strip() if cur else word
        if count_effective_length(cand) > max_effective_chars:
            if cur:
                chunks.append(cur.strip())
                cur = word
            else:
                chunks.append(word)
                cur = ""
        else:
            cur = cand
    if cur:
        chunks.append(cur.strip())
    return chunks


def batch_chunks(chunks: Sequence[str], batch_size: int) -> Iterator[Sequence[str]]:
    for i in range(0, len(chunks), batch_size):            # app.py:124-127
        yield chunks[i: i + batch_size]


def plan_batches(text: str, chunk_size: int = 0, max_new_tokens: int = 3072, batch_size: int = 4):
    """[(batch text, max_tokens)] exactly as the front-end schedules them (app.py:198-218): chunks joined by
    newlines, token budget scaled by the batch's effective length, never below 256."""
    cs = auto_adjust_chunk_size(text, chunk_size)
    chunks = split_by_words_respecting_special_tokens(text, max_effective_chars=cs)
    plan = []
    for batch in batch_chunks(chunks, batch_size):
        bt = "\n".join(c.strip() for c in batch).strip()
        if not bt:
            raise ValueError("All chunks in this batch were empty after trimming.")
        plan.append((bt, max(256, int(max_new_tokens * (count_effective_length(bt) / cs)))))
    return plan


def generate_long_codes(dia, text: str, *, chunk_size: int = 0, max_new_tokens: int = 3072, cfg_scale: float = 3.0,
                        temperature: float = 1.3, top_p: float = 0.95, cfg_filter_top_k: int = 35,
                        audio_prompt: Optional[np.ndarray] = None, audio_prompt_text: Optional[str] = None,
                        seed: Optional[int] = None, verbose: bool = False) -> List[np.ndarray]:
    """Chunk chaining on codes: one [1, C, T'] codec input per batch of chunks; batch i+1 is prompted with the
    codes and text of batch i (app.py:220-244)."""
    if not text or text.isspace():
        raise ValueError("Text input cannot be empty.")
    if audio_prompt is not None and (not audio_prompt_text or audio_prompt_text.isspace()):
        raise ValueError("Audio Prompt Text input cannot be empty.")
    T_cap = dia.config.data.audio_length
    prompt, ptext = audio_prompt, audio_prompt_text
    out: List[np.ndarray] = []
    for i, (bt, budget) in enumerate(plan_batches(text, chunk_size, max_new_tokens)):
        n_prompt = 0 if prompt is None else int(np.asarray(prompt).reshape(-1, dia.config.data.channels).shape[0])
        # The front-end hands its "max new tokens" budget straight to generate(max_tokens=...), where the
        # prompt rows count too, so a chained batch would be left with budget - prompt frames (none, typically).
        # That loop never ran in the reference (its audio-prompt prefill raises); here the
        # budget means NEW frames, as its name says: rows = BOS + prompt + budget, capped by the buffer.
        mt = min(T_cap, n_prompt + 1 + budget)
        codes = dia.generate_batch([bt], max_tokens=mt, cfg_scale=cfg_scale, temperature=temperature, top_p=top_p,
                                   cfg_filter_top_k=cfg_filter_top_k, seeds=None if seed is None else [seed + i],
                                   verbose=verbose, audio_prompts=[prompt], audio_prompt_texts=[ptext])[0]
        if codes.shape[-1] == 0:
            continue
        out.append(codes)
        prompt, ptext = np.ascontiguousarray(codes[0].T), bt          # next prompt = this batch (codes [T, C])
    return out


def generate_long(dia, text: str, **kw) -> Optional[np.ndarray]:
    """as the front-end: decode every batch, join with 0.2 s of silence (app.py:238-250).  Needs the codec."""
    segs: List[np.ndarray] = []
    parts = generate_long_codes(dia, text, **kw)
    for i, codes in enumerate(parts):
        segs.append(dia._generate_output(codes))
        if i + 1 < len(parts):
            segs.append(np.zeros(int(DEFAULT_SAMPLE_RATE * 0.2), dtype=np.float32))
    return np.concatenate(segs) if segs else None
