#!/usr/bin/env python3
"""Offline pruning of a Dia checkpoint — same flags and outputs as the reference's offline_prune.py
(30-56: --model-path --output-dir --prune-mode --prune-amount --prune-dim --prune-norm --device
--compute-dtype; writes ``pytorch_model.bin`` + ``config.json`` with permanent zeros and no masks, 153-156).

This is an offline CPU tool (fp32 tensor arithmetic on the checkpoint, no model execution); its output is
what ``Dia.from_local`` / ``cli.py --pruned-checkpoint`` load, where a structured-pruned checkpoint is
repacked into physically smaller tensors (dia_hip/compact.py).
"""

from __future__ import annotations

import argparse
import os
import sys
from pathlib import Path

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "dia-tts-prune_amd"))


def main(argv=None) -> int:
    p = argparse.ArgumentParser(description="Prune a Dia checkpoint on the CPU and write it back with the zeros baked in.")
    p.add_argument("--model-path", type=str, required=True, help="directory of the dense model: config.json plus its checkpoint")
    p.add_argument("--output-dir", type=str, required=True, help="where pytorch_model.bin and config.json of the pruned model go")
    p.add_argument("--prune-mode", type=str, required=True, choices=["unstructured", "structured"], help="global magnitude pruning of single weights, or whole slices along --prune-dim")
    p.add_argument("--prune-amount", type=float, required=True, help="share of weights (unstructured) or of slices per matrix (structured) to zero, strictly between 0 and 1")
    p.add_argument("--prune-dim", type=int, default=0, help="structured mode: the axis whose slices are ranked and zeroed")
    p.add_argument("--prune-norm", type=int, default=2, choices=[1, 2], help="structured mode: rank slices by their L1 or L2 norm")
    p.add_argument("--device", type=str, default="cpu", help="accepted for compatibility; the tool runs on the CPU")
    p.add_argument("--compute-dtype", type=str, default="float32", choices=["float32"], help="pruning arithmetic is float32")
    a = p.parse_args(argv)
    if not (0.0 < a.prune_amount < 1.0):
        print("Error: --prune-amount must be between 0.0 and 1.0 (exclusive).")
        return 1

    from dia_hip import weights as W
    from dia_hip.pruning import sparsity, structured_prune_state_dict, unstructured_prune_state_dict

    out = Path(a.output_dir)
    out.mkdir(parents=True, exist_ok=True)
    print(f"Loading base model from {a.model_path}...")
    try:
        cfg_path, ckpt = W.find_checkpoint_in_dir(a.model_path)
        cfg = W.read_hub_config(cfg_path)
        sd = {k: v.float() for k, v in W.load_state_dict_file(ckpt).items()}
        missing, _ = W.check_state_dict(cfg, sd)
        if missing:
            raise RuntimeError(f"Missing keys in checkpoint: {missing}")
    except Exception as e:
        print(f"Error loading model: {e}")
        return 1
    print(f"\nApplying {a.prune_mode} pruning...")
    if a.prune_mode == "unstructured":
        psd = unstructured_prune_state_dict(cfg, sd, a.prune_amount)
    else:
        psd, _ = structured_prune_state_dict(cfg, sd, a.prune_amount, dim=a.prune_dim, n=a.prune_norm)
    print(f"Achieved sparsity: {sparsity(cfg, psd):.4f}")
    print(f"\nSaving pruned model to {a.output_dir}...")
    torch.save(dict(psd), out / "pytorch_model.bin")
    cfg.save(str(out / "config.json"))
    print("Pruned model state dict and config saved.")
    print("\nOffline pruning finished successfully.")
    return 0


if __name__ == "__main__":
    sys.exit(main())
