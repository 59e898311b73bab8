"""``Dia`` — the drop-in boundary of the decode path (reference dia/model.py:101-846).

Same constructor / ``from_local`` / ``from_pretrained`` / ``generate`` / ``save_audio`` /
``load_audio`` surface and error behaviour as the reference class; underneath, text goes to byte
tokens on the host and everything from the encoder to the sampled token buffer runs in
``libdia_hip.so`` on one MI355X.  Extensions that the reference lacks are additive:
``generate_codes`` (the [1, 9, T'] code tensor the codec would receive — the Descript Audio Codec
itself is a third-party network outside this path), ``generate_batch`` (B utterances per call, the
reference hard-codes B=1, dia/state.py:83-84) and a ``load_dac`` switch for offline use.
"""

from __future__ import annotations

import time
from enum import Enum
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import binding as hb
from .config import DiaConfig
from .engine import DecodeSession, DeviceWeights, UtteranceResult
from .tokens import codes_for_codec, effective_text, encode_text
from . import weights as W

DEFAULT_SAMPLE_RATE = 44100


class ComputeDtype(str, Enum):
    FLOAT32 = "float32"
    FLOAT16 = "float16"
    BFLOAT16 = "bfloat16"

    def to_dtype(self) -> torch.dtype:
        return {"float32": torch.float32, "float16": torch.float16, "bfloat16": torch.bfloat16}[self.value]


def _default_device() -> torch.device:
    if torch.cuda.is_available():
        return torch.device("cuda")
    raise hb.DiaHipError("no HIP device visible: this build of Dia runs on MI355X only (no CPU path)")


class Dia:
    # what compute_dtype="float32" does with a checkpoint that bf16 cannot hold: "exact" = three bf16 planes per weight
    # (the reference's fp32 path, ~3x slower), "round" = one rounded bf16 tile set (fast; the load prints a warning)
    fp32_weights = "exact"

    def __init__(self, config: DiaConfig, compute_dtype: Union[str, ComputeDtype] = ComputeDtype.FLOAT32,
                 device: Optional[torch.device] = None):
        """compute_dtype selects the K/V-cache precision: float32 keeps K/V in fp32 (the parity
        configuration against the reference's fp32 CPU path), bfloat16/float16 keep them in bf16
        (the reference's GPU configuration, state.py:142-151).  Weights are streamed as bf16 and all
        accumulation is fp32 in both cases."""
        self.config = config
        self.device = torch.device(device) if device is not None else _default_device()
        if self.device.type != "cuda":
            raise hb.DiaHipError(f"device {self.device} is not a HIP device; there is no CPU fallback")
        if isinstance(compute_dtype, str):
            compute_dtype = ComputeDtype(compute_dtype)
        self.compute_dtype = compute_dtype.to_dtype()
        self.model: Optional[DeviceWeights] = None      # resident weights (reference: DiaModel nn.Module)
        self.dac_model = None
        self.last_codes: Optional[np.ndarray] = None
        self.weights_rounded = False
        self.weights_exact_planes = False
        hb.lib()                                        # fail now, not at first generate()

    # ------------------------------------------------------------------ loaders
    def _install(self, sd: Dict[str, torch.Tensor], adapter_path: Optional[str] = None) -> None:
        if any("lora_" in k for k in sd):           # model.py:170-172: adapter keys of a half-merged checkpoint are dropped
            sd = {k: v for k, v in sd.items() if "lora_" not in k}
        if adapter_path:
            from .lora import merge_lora_state_dict
            sd = merge_lora_state_dict(sd, adapter_path)
        missing, unexpected = W.check_state_dict(self.config, sd)
        if unexpected:
            print(f"Warning: Unexpected keys found in checkpoint: {unexpected}")
        if missing:
            # deliberate tightening: the reference loads with strict=False and only warns (model.py:173-177), which
            # leaves the missing tensors at their random initialisation; listed in INTEGRATION.md
            raise RuntimeError(f"Missing keys in checkpoint: {missing}")
        with torch.cuda.device(self.device):
            self.model = DeviceWeights(self.config, sd, self.device)
            # DenseGeneral kernels are streamed as ONE bf16 tile set by the fast kernels.  A checkpoint whose values are
            # bf16-representable (bf16-trained weights stored as fp32, the synthetic ones) loses nothing.  A genuine fp32
            # checkpoint would be rounded once at load = the reference's bfloat16 configuration, NOT its float32 path:
            #   compute_dtype="float32": the weights are kept as three bf16 planes (hi + mid + lo == w exactly) and run
            #       through the generic kernel — the reference's fp32 arithmetic for ANY checkpoint, at about a third
            #       of the speed (Dia.fp32_weights = "round" keeps the single rounded tile set and says so);
            #   bfloat16 / float16: rounded once, like the reference's own low-precision modules.
            self.weights_rounded = self.model.max_weight_rounding > 0.0
            self.weights_exact_planes = False
            if self.weights_rounded and self.compute_dtype == torch.float32:
                if self.fp32_weights == "exact":
                    rounding = self.model.max_weight_rounding
                    self.model = DeviceWeights(self.config, sd, self.device, weight_planes=3)
                    self.weights_rounded, self.weights_exact_planes = False, True
                    print(f"Note: checkpoint weights are not bf16-representable (largest relative rounding {rounding:.2e}): "
                          f"compute_dtype='float32' keeps them exact as three bf16 planes (3x the weight traffic, generic kernel).")
                else:
                    print(f"Warning: checkpoint weights are not bf16-representable (largest relative rounding "
                          f"{self.model.max_weight_rounding:.2e}); they are streamed as bf16, so compute_dtype='float32' here means "
                          f"fp32 activations / accumulation / K/V over bf16-rounded weights, not the reference's fp32 weights.")

    @classmethod
    def from_state_dict(cls, config: DiaConfig, state_dict: Dict[str, torch.Tensor],
                        compute_dtype: Union[str, ComputeDtype] = ComputeDtype.FLOAT32,
                        device: Optional[torch.device] = None) -> "Dia":
        dia = cls(config, compute_dtype, device)
        dia._install(state_dict)
        return dia

    @classmethod
    def from_local(cls, config_path: str, checkpoint_path: str,
                   compute_dtype: Union[str, ComputeDtype] = ComputeDtype.FLOAT32,
                   device: Optional[torch.device] = None, load_dac: bool = True,
                   adapter_path: Optional[str] = None) -> "Dia":
        """reference model.py:139-187: config JSON + pickled/safetensors state_dict (this is the
        loader for ``offline_prune.py`` outputs).  ``adapter_path``: LoRA adapter directory merged into the
        dense weights before they are tiled (reference cli.py:166-174 wraps with PEFT at run time)."""
        config = DiaConfig.load(config_path)
        if config is None:
            raise FileNotFoundError(f"Config file not found at {config_path}")
        dia = cls(config, compute_dtype, device)
        try:
            sd = W.load_state_dict_file(checkpoint_path)
        except FileNotFoundError:
            raise FileNotFoundError(f"Checkpoint file not found at {checkpoint_path}")
        except Exception as e:
            raise RuntimeError(f"Error loading checkpoint from {checkpoint_path}") from e
        dia._install(sd, adapter_path)
        if load_dac:
            dia._load_dac_model()
        return dia

    @classmethod
    def from_pretrained(cls, model_name: str = "nari-labs/Dia-1.6B",
                        compute_dtype: Union[str, ComputeDtype] = ComputeDtype.FLOAT32,
                        device: Optional[torch.device] = None, load_dac: bool = True,
                        adapter_path: Optional[str] = None, **kwargs) -> "Dia":
        """reference model.py:189-236.  ``model_name`` must be a local directory holding
        ``config.json`` and ``model.safetensors`` / ``pytorch_model.bin`` (hub download needs a network)."""
        p = Path(model_name)
        if not p.is_dir():
            raise FileNotFoundError(
                f"{model_name!r} is not a local model directory; fetching from the Hugging Face Hub is not supported here")
        cfg_path, ckpt = W.find_checkpoint_in_dir(str(p))
        config = W.read_hub_config(cfg_path)
        dia = cls(config, compute_dtype, device)
        dia._install(W.load_state_dict_file(ckpt), adapter_path)
        if load_dac:
            dia._load_dac_model()
        return dia

    def _load_dac_model(self):
        """reference model.py:238-252 — raises RuntimeError when the codec cannot be loaded."""
        try:
            import dac  # type: ignore

            print("Loading DAC model...")
            path = dac.utils.download()
            m = dac.DAC.load(path).to(self.device)
            m.eval()
        except Exception as e:
            raise RuntimeError(f"Failed to load DAC model: {e}") from e
        self.dac_model = m

    # ------------------------------------------------------------------ generation
    def _kv_dtype(self) -> str:
        return "f32" if self.compute_dtype == torch.float32 else "bf16"

    def _run(self, texts: Sequence[str], max_tokens, cfg_scale, temperature, top_p, top_k,
             seeds: Optional[Sequence[Optional[int]]], verbose: bool, ignore_eos: bool = False,
             use_graph: bool = True, audio_prompts: Optional[Sequence[Optional[np.ndarray]]] = None) -> List[UtteranceResult]:
        if self.model is None:
            raise RuntimeError("no weights loaded")
        ids = [encode_text(t, self.config) for t in texts]
        t0 = time.time()
        with torch.cuda.device(self.device):
            s = DecodeSession(self.model, ids, kv_dtype=self._kv_dtype(), max_tokens=max_tokens, cfg_scale=cfg_scale,
                              temperature=temperature, top_p=top_p, top_k=top_k, seeds=seeds, ignore_eos=ignore_eos,
                              audio_prompts=audio_prompts)
            try:
                s.prefill()
                s.sync()
                t1 = time.time()
                if verbose:
                    print(f"generate: prefill {t1 - t0:.3f}s; prefill audio steps {s.first_steps}; starting generation loop")
                s.run(use_graph=use_graph)
                res = s.results()
                if verbose:
                    dt = time.time() - t1
                    n = sum(r.last_step for r in res)
                    print(f"generate: {n} frames in {dt:.3f}s = {n / max(dt, 1e-9):.1f} frames/s")
            finally:
                s.close()
        return res

    @torch.inference_mode()
    def generate_batch(self, texts: Sequence[str], max_tokens: Optional[int] = None, cfg_scale: float = 3.0,
                       temperature: float = 1.3, top_p: float = 0.95, cfg_filter_top_k: int = 35,
                       seeds: Optional[Sequence[Optional[int]]] = None, verbose: bool = False,
                       ignore_eos: bool = False, audio_prompts: Optional[Sequence[Optional[np.ndarray]]] = None,
                       audio_prompt_texts: Optional[Sequence[Optional[str]]] = None) -> List[np.ndarray]:
        """B utterances in one decode loop; returns the codec inputs [1, C, T'_b] per utterance.
        audio_prompts: per utterance None or codes [Tp, C]; audio_prompt_texts: their transcripts."""
        apt = audio_prompt_texts or [None] * len(texts)
        eff = [effective_text(t, a) for t, a in zip(texts, apt)]
        res = self._run(eff, max_tokens, cfg_scale, temperature, top_p, cfg_filter_top_k, seeds, verbose, ignore_eos,
                        audio_prompts=audio_prompts)
        return [codes_for_codec(r.codes, self.config) for r in res]

    @torch.inference_mode()
    def generate_codes(self, text: str, max_tokens: Optional[int] = None, cfg_scale: float = 3.0,
                       temperature: float = 1.3, top_p: float = 0.95, cfg_filter_top_k: int = 35,
                       seed: Optional[int] = None, verbose: bool = False) -> Optional[np.ndarray]:
        out = self.generate_batch([text], max_tokens, cfg_scale, temperature, top_p, cfg_filter_top_k,
                                  None if seed is None else [seed], verbose)
        return out[0]

    @torch.inference_mode()
    def generate(self, text: str, max_tokens: Optional[int] = None, cfg_scale: float = 3.0, temperature: float = 1.3,
                 top_p: float = 0.95, use_torch_compile: bool = False, cfg_filter_top_k: int = 35,
                 audio_prompt: Union[str, torch.Tensor, None] = None, audio_prompt_text: Optional[str] = None,
                 seed: Optional[int] = None, verbose: bool = False) -> Optional[np.ndarray]:
        """reference model.py:631-846.  Failures inside generation are printed and turned into
        ``None`` exactly like the reference (model.py:729-733, 817-821, 841-845); ``use_torch_compile``
        is accepted and ignored (there is nothing to compile)."""
        if audio_prompt is not None and not audio_prompt_text:
            raise ValueError("`audio_prompt_text` is required when `audio_prompt` is provided.")
        if seed is not None:
            torch.manual_seed(seed)
            np.random.seed(seed)
        eff = effective_text(text, audio_prompt_text)
        prompt = None
        if audio_prompt is not None:
            # model.py:372-379: a path is encoded by the codec, a tensor is taken as codes [T, C] / [1, T, C].
            # The prompt rows are replayed through the decode step (semantic decision: DESIGN.md, audio prompt).
            try:
                pt = self.load_audio(audio_prompt) if isinstance(audio_prompt, str) else audio_prompt
                prompt = pt.detach().cpu().numpy() if isinstance(pt, torch.Tensor) else np.asarray(pt)
            except Exception as e:
                print(f"Error during preparation: {e}")             # model.py:729-733
                return None
        try:
            res = self._run([eff], max_tokens, cfg_scale, temperature, top_p, cfg_filter_top_k,
                            None if seed is None else [seed], verbose,
                            audio_prompts=None if prompt is None else [prompt])[0]
        except Exception as e:
            print(f"Error during generation loop: {e}")
            import traceback

            traceback.print_exc()
            return None
        if res.codes.shape[0] == 0:
            print("Warning: No new tokens were generated after prefill.")
            return None
        self.last_codes = codes_for_codec(res.codes, self.config)
        try:
            return self._generate_output(self.last_codes)
        except Exception as e:
            print(f"Error during final decoding: {e}")
            return None

    def _generate_output(self, codec_input: np.ndarray) -> Optional[np.ndarray]:
        """codes [1, C, T'] -> waveform through the third-party codec (model.py:535-544)."""
        if self.dac_model is None:
            raise RuntimeError("DAC model not loaded. Cannot decode audio.")
        codes = torch.from_numpy(codec_input.astype(np.int64)).to(self.device)
        with torch.inference_mode():
            z = self.dac_model.quantizer.from_codes(codes)
            audio = self.dac_model.decode(z[0])
        return audio.squeeze().cpu().numpy()

    # ------------------------------------------------------------------ audio file IO (host side)
    # Both need third-party pieces that do not exist offline (the DAC codec, torchaudio, soundfile): row (f)-3 of
    # SURVEY.md section 8 stays outside this build.  The methods keep the reference's names, arguments and failure
    # behaviour (reference model.py:546-595) so that callers written against it run unchanged once a codec is attached.
    def load_audio(self, audio_path: str) -> torch.Tensor:
        """waveform file -> codec frames [T, C] for ``generate(audio_prompt=...)`` (mono, 44.1 kHz)."""
        codec = self.dac_model
        if codec is None:
            raise RuntimeError("DAC model not loaded. Cannot encode audio.")
        try:
            import torchaudio  # type: ignore
            wave, rate = torchaudio.load(audio_path)
        except FileNotFoundError:
            raise FileNotFoundError(f"Audio file not found: {audio_path}")
        except Exception as e:
            raise RuntimeError(f"Error loading or encoding audio file {audio_path}: {e}") from e
        try:
            mono = wave if wave.shape[0] == 1 else wave.mean(dim=0, keepdim=True)
            if rate != DEFAULT_SAMPLE_RATE:
                mono = torchaudio.functional.resample(mono, rate, DEFAULT_SAMPLE_RATE)
            with torch.inference_mode():
                prepared = codec.preprocess(mono.to(self.device)[None], DEFAULT_SAMPLE_RATE)
                frames = codec.encode(prepared)[1]              # (z, codes, latents, ...): the codes
            return frames[0].T.contiguous()
        except Exception as e:
            raise RuntimeError(f"Error loading or encoding audio file {audio_path}: {e}") from e

    def save_audio(self, path: str, audio: np.ndarray, sample_rate: int = DEFAULT_SAMPLE_RATE):
        """waveform -> file; failures are printed, not raised, like every failure behind ``generate``."""
        if audio is None:
            print("Warning: Cannot save None audio.")
            return
        try:
            import soundfile  # type: ignore
            out = Path(path)
            out.parent.mkdir(parents=True, exist_ok=True)
            samples = np.asarray(audio)
            if samples.dtype.kind in "iu":
                samples = samples.astype(np.float32) / float(np.iinfo(samples.dtype).max)
            soundfile.write(str(out), samples.clip(-1.0, 1.0), sample_rate)
        except Exception as e:
            print(f"Error saving audio to {path}: {e}")
