#!/usr/bin/env python3
"""Command-line front-end on the MI355X decode path — same flags as the reference's cli.py (36-98), same
flow (validate -> seed -> load -> optional adapters -> generate -> save, 100-229).

Differences, all forced by the platform: the device is always the HIP device (there is no CPU path);
``--compute-dtype`` selects the K/V cache dtype (weights are bf16 tiles either way; float32 = parity mode);
LoRA adapters are merged into the dense weights at load (dia_hip/lora.py) instead of wrapped by PEFT;
``--codes-output`` (build-only) saves the codec input ``[1, 9, T]`` as .npy, which is the only possible
output where the Descript Audio Codec is not installed; ``--no-dac`` skips loading it.
"""

from __future__ import annotations

import argparse
import os
import random
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dia-tts-prune_amd"))


def set_seed(seed: int):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Dia text-to-dialogue on MI355X: text in, audio (or codec codes) out.")
    p.add_argument("text", type=str, help="text to synthesise ([S1]/[S2] speaker tags); with --audio-prompt give only the new text here, the prompt transcript goes to --audio-prompt-text")
    p.add_argument("--output", type=str, default=None, help="audio file to write (needs the codec)")
    p.add_argument("--codes-output", type=str, default=None, help="(build-only) path for the codec input codes [1, C, T] as .npy")
    g = p.add_argument_group("Model Loading")
    g.add_argument("--model-path", type=str, default="nari-labs/Dia-1.6B", help="model directory holding config.json and the checkpoint (hub ids cannot be fetched offline)")
    g.add_argument("--config", type=str, default=None, help="config.json to use instead of the one in --model-path")
    g.add_argument("--pruned-checkpoint", type=str, default=None, help="checkpoint file to load instead of the one in --model-path, e.g. an offline_prune.py output")
    g.add_argument("--adapter-path", type=str, default=None, help="LoRA adapter directory; folded into the dense weights while loading")
    g.add_argument("--no-dac", action="store_true", help="(build-only) do not load the audio codec; requires --codes-output")
    g = p.add_argument_group("Audio Prompting (Voice Cloning)")
    g.add_argument("--audio-prompt", type=str, default=None, help="voice to clone: an audio file (needs the codec) or a .npy of codec codes [T, 9]")
    g.add_argument("--audio-prompt-text", type=str, default=None, help="what is said in --audio-prompt (mandatory with it)")
    g = p.add_argument_group("Generation Parameters")
    g.add_argument("--max-tokens", type=int, default=None, help="cap on generated frames incl. the prompt (default: audio_length of the config)")
    g.add_argument("--cfg-scale", type=float, default=3.0, help="classifier-free guidance strength")
    g.add_argument("--temperature", type=float, default=1.3, help="softmax temperature of the sampler; 0 = greedy")
    g.add_argument("--top-p", type=float, default=0.95, help="nucleus (top-p) mass kept by the sampler")
    g.add_argument("--cfg-filter-top-k", type=int, default=35, help="keep only the k best logits after guidance; 0 switches the filter off")
    g.add_argument("--seed", type=int, default=None, help="seed of the sampling noise")
    g = p.add_argument_group("Infrastructure")
    g.add_argument("--device", type=str, default=None, help="HIP device such as cuda:0 (default: the current one)")
    g.add_argument("--compute-dtype", type=str, default="bfloat16", choices=["float16", "bfloat16", "float32"], help="K/V cache dtype: bfloat16 (default; float16 is accepted and mapped to it) or float32.")
    g.add_argument("--verbose", action="store_true", help="report prefill and generation timing")
    return p


def main(argv=None) -> int:
    parser = build_parser()
    args = parser.parse_args(argv)
    if args.audio_prompt and not args.audio_prompt_text:
        parser.error("--audio-prompt needs its transcript: pass --audio-prompt-text")
    if args.pruned_checkpoint and not args.config and not Path(args.model_path).is_dir():
        parser.error("--pruned-checkpoint needs --config unless --model-path is a local directory with a config.json")
    if not args.output and not args.codes_output:
        parser.error("one of --output / --codes-output is required.")
    if args.no_dac and args.output:
        parser.error("--output needs the audio codec; use --codes-output with --no-dac.")

    from dia_hip.model import Dia

    if args.seed is not None:
        set_seed(args.seed)
        print(f"Using seed: {args.seed}")
    device = torch.device(args.device) if args.device else None
    print("Loading model...")
    try:
        load_dac = not args.no_dac
        if args.pruned_checkpoint:
            cfg_path = args.config or str(Path(args.model_path) / "config.json")
            if not Path(cfg_path).exists():
                parser.error(f"Config file not found in {args.model_path} and --config not provided.")
            print(f"Loading specific checkpoint: {args.pruned_checkpoint}\nUsing config: {cfg_path}")
            dia = Dia.from_local(cfg_path, args.pruned_checkpoint, args.compute_dtype, device, load_dac=load_dac,
                                 adapter_path=args.adapter_path)
        else:
            print(f"Loading model from: {args.model_path}")
            dia = Dia.from_pretrained(args.model_path, args.compute_dtype, device, load_dac=load_dac,
                                      adapter_path=args.adapter_path)
        if args.adapter_path:
            print("LoRA adapters merged successfully.")
        print("Model loaded successfully.")
    except Exception as e:
        print(f"Error loading model: {e}")
        import traceback
        traceback.print_exc()
        return 1

    full_text = (args.audio_prompt_text.strip() + " " + args.text.strip()) if args.audio_prompt else args.text.strip()   # cli.py:186-190
    prompt = args.audio_prompt
    if prompt and prompt.endswith(".npy"):
        prompt = torch.from_numpy(np.load(prompt).astype(np.int64))
    print("Generating audio...")
    try:
        audio = dia.generate(text=full_text, audio_prompt=prompt, audio_prompt_text=args.audio_prompt_text,
                             max_tokens=args.max_tokens, cfg_scale=args.cfg_scale, temperature=args.temperature,
                             top_p=args.top_p, cfg_filter_top_k=args.cfg_filter_top_k, seed=args.seed, verbose=args.verbose)
        if args.codes_output and dia.last_codes is not None:
            Path(args.codes_output).parent.mkdir(parents=True, exist_ok=True)
            np.save(args.codes_output, dia.last_codes)
            print(f"Codes saved to {args.codes_output}: shape {tuple(dia.last_codes.shape)}")
        if args.output:
            if audio is None:
                print("Generation failed to produce audio.")
                return 1
            Path(args.output).parent.mkdir(parents=True, exist_ok=True)
            print(f"Saving audio to {args.output}...")
            dia.save_audio(args.output, audio)
            print(f"Audio successfully saved to {args.output}")
        elif dia.last_codes is None:
            print("Generation failed to produce codes.")
            return 1
        print("Audio generation complete.")
    except Exception as e:
        print(f"Error during audio generation or saving: {e}")
        import traceback
        traceback.print_exc()
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())
