"""Shape / special-token configuration of the Dia decode path.

Field names, defaults, validation and JSON layout mirror the reference's
``dia/config.py:24-153`` (DataConfig, EncoderConfig, DecoderConfig, ModelConfig,
DiaConfig) and its ``save``/``load`` pair (``dia/config.py:156-207``) so that a
``config.json`` written by either side is readable by the other.  Nothing here
touches the GPU.
"""

from __future__ import annotations

import json
import os
from pathlib import Path
from typing import Annotated, List, Optional, Union

from pydantic import BaseModel, BeforeValidator, Field, ValidationError


def _ceil128(v: int) -> int:
    # reference config.py:38-39 rounds text/audio lengths up to a multiple of 128
    return (int(v) + 127) // 128 * 128


_Len128 = Annotated[int, BeforeValidator(_ceil128)]


class DataConfig(BaseModel, frozen=True):
    text_length: _Len128 = Field(gt=0, multiple_of=128)
    audio_length: _Len128 = Field(gt=0, multiple_of=128)
    channels: int = Field(default=9, gt=0, multiple_of=1)
    text_pad_value: int = 0
    audio_eos_value: int = 1024
    audio_pad_value: int = 1025
    audio_bos_value: int = 1026
    delay_pattern: List[Annotated[int, Field(ge=0)]] = Field(
        default_factory=lambda: [0, 8, 9, 10, 11, 12, 13, 14, 15]
    )

    def __hash__(self) -> int:
        return hash(
            (
                self.text_length,
                self.audio_length,
                self.channels,
                self.text_pad_value,
                self.audio_pad_value,
                self.audio_bos_value,
                self.audio_eos_value,
                tuple(self.delay_pattern),
            )
        )


class EncoderConfig(BaseModel, frozen=True):
    n_layer: int = Field(gt=0)
    n_embd: int = Field(gt=0)
    n_hidden: int = Field(gt=0)
    n_head: int = Field(gt=0)
    head_dim: int = Field(gt=0)


class DecoderConfig(BaseModel, frozen=True):
    n_layer: int = Field(gt=0)
    n_embd: int = Field(gt=0)
    n_hidden: int = Field(gt=0)
    gqa_query_heads: int = Field(gt=0)
    kv_heads: int = Field(gt=0)
    gqa_head_dim: int = Field(gt=0)
    cross_query_heads: int = Field(gt=0)
    cross_head_dim: int = Field(gt=0)


class ModelConfig(BaseModel, frozen=True):
    encoder: EncoderConfig
    decoder: DecoderConfig
    src_vocab_size: int = Field(default=128, gt=0)
    tgt_vocab_size: int = Field(default=1028, gt=0)
    dropout: float = Field(default=0.0, ge=0.0, lt=1.0)
    normalization_layer_epsilon: float = Field(default=1.0e-5, ge=0.0)
    weight_dtype: str = Field(default="float32")
    rope_min_timescale: int = Field(default=1)
    rope_max_timescale: int = Field(default=10_000)


class DiaConfig(BaseModel, frozen=True):
    version: str = "1.0"
    model: ModelConfig
    data: DataConfig
    model_type: str = "dia"
    architectures: List[str] = Field(default_factory=lambda: ["DiaModel"])

    def save(self, path: Union[str, Path]) -> None:
        """Write JSON; a missing ``.json`` suffix is added (reference config.py:156-172)."""
        p = Path(path)
        if p.suffix != ".json":
            p = p.with_suffix(".json")
        os.makedirs(p.parent, exist_ok=True)
        p.write_text(self.model_dump_json(indent=2), encoding="utf-8")

    @classmethod
    def load(cls, path: Union[str, Path]) -> Optional["DiaConfig"]:
        """Return the validated config, ``None`` when the file is absent, and re-raise
        validation errors — the contract of reference config.py:174-207."""
        p = Path(path)
        if not p.is_file():
            print(f"Config file not found at: {p}")
            return None
        if p.suffix != ".json":
            print(f"Warning: Config file does not have .json extension: {p}")
        try:
            return cls.model_validate_json(p.read_text(encoding="utf-8"))
        except ValidationError as e:
            print(f"Configuration validation error loading {p}: {e}")
            raise


def dia_1_6b_config() -> DiaConfig:
    """Dia-1.6B shapes: 1 611.2 M parameters."""
    return DiaConfig(
        model=ModelConfig(
            encoder=EncoderConfig(n_layer=12, n_embd=1024, n_hidden=4096, n_head=16, head_dim=128),
            decoder=DecoderConfig(
                n_layer=18, n_embd=2048, n_hidden=8192, gqa_query_heads=16, kv_heads=4,
                gqa_head_dim=128, cross_query_heads=16, cross_head_dim=128,
            ),
            src_vocab_size=256,
            tgt_vocab_size=1028,
        ),
        data=DataConfig(text_length=1024, audio_length=3072),
    )


def tiny_config() -> DiaConfig:
    """SURVEY.md App. D 'tiny' fixture config (head_dim 16)."""
    return DiaConfig(
        model=ModelConfig(
            encoder=EncoderConfig(n_layer=2, n_embd=64, n_hidden=128, n_head=4, head_dim=16),
            decoder=DecoderConfig(
                n_layer=2, n_embd=96, n_hidden=192, gqa_query_heads=4, kv_heads=2,
                gqa_head_dim=16, cross_query_heads=4, cross_head_dim=16,
            ),
            src_vocab_size=256,
            tgt_vocab_size=1028,
        ),
        data=DataConfig(text_length=128, audio_length=128),
    )


def mid_config() -> DiaConfig:
    """SURVEY.md App. D 'mid' fixture config: real head_dim 128 and GQA ratio 4."""
    return DiaConfig(
        model=ModelConfig(
            encoder=EncoderConfig(n_layer=2, n_embd=256, n_hidden=512, n_head=4, head_dim=128),
            decoder=DecoderConfig(
                n_layer=3, n_embd=512, n_hidden=1024, gqa_query_heads=8, kv_heads=2,
                gqa_head_dim=128, cross_query_heads=8, cross_head_dim=128,
            ),
            src_vocab_size=256,
            tgt_vocab_size=1028,
        ),
        data=DataConfig(text_length=256, audio_length=256),
    )


def config_from_json_dict(d: dict) -> DiaConfig:
    return DiaConfig.model_validate(d)


def config_to_json_dict(c: DiaConfig) -> dict:
    return json.loads(c.model_dump_json())
