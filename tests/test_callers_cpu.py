"""Host logic of the callers above Dia.generate: text chunking against the
reference's own helpers, LoRA merge arithmetic, CLI argument surface.  No GPU."""
import json
import os
import sys

import numpy as np
import pytest
import torch

from dia_hip import callers as CL
from dia_hip import config as C
from dia_hip.lora import merge_lora_state_dict
from dia_hip.weights import synthetic_state_dict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_chunking_matches_reference(golden):
    g = golden("ref_chunks.npz")
    for i in range(int(g["n"])):
        t = str(g[f"text_{i}"])
        assert CL.count_effective_length(t) == int(g[f"len_{i}"])
        for user in (0, 40):
            cs = CL.auto_adjust_chunk_size(t, user)
            assert cs == int(g[f"cs_{i}_{user}"])
            chunks = CL.split_by_words_respecting_special_tokens(t, max_effective_chars=cs)
            want = str(g[f"chunks_{i}_{user}"][0])
            assert chunks == (want.split("\x00") if want else [])
            assert [len(b) for b in CL.batch_chunks(chunks, 4)] == g[f"batches_{i}_{user}"].tolist()


def test_plan_batches_budget():
    base = "[S1] Dia is an open weights text to dialogue model. [S2] You get full control over scripts and voices. "
    plan = CL.plan_batches(base * 6, chunk_size=0, max_new_tokens=1000)
    assert len(plan) >= 2 and all(mt >= 256 for _, mt in plan)
    bt, mt = plan[0]
    assert bt.count("\n") == 3 and mt == max(256, int(1000 * CL.count_effective_length(bt) / 48))      # app.py:209-212
    with pytest.raises(ValueError):
        CL.generate_long_codes(object(), "   ")


def _adapter(tmp_path, tensors, r=4, alpha=8, **extra):
    d = tmp_path / "adapter"
    d.mkdir(exist_ok=True)
    json.dump(dict(r=r, lora_alpha=alpha, target_modules=["q_proj", "o_proj"], **extra), open(d / "adapter_config.json", "w"))
    torch.save(tensors, d / "adapter_model.bin")
    return str(d)


def test_lora_merge_arithmetic(tmp_path):
    cfg = C.tiny_config()
    sd = synthetic_state_dict(cfg, seed=1, std=0.05)
    torch.manual_seed(0)
    qn, on = "decoder.layers.0.self_attention.q_proj", "decoder.layers.1.cross_attention.o_proj"
    wq, wo = sd[qn + ".weight"], sd[on + ".weight"]                 # [D, H, hd] and [H, hd, D]
    r = 4
    Aq, Bq = torch.randn(r, wq.shape[0]), torch.randn(wq[0].numel(), r)
    Ao, Bo = torch.randn(r, wo.shape[0] * wo.shape[1]), torch.randn(wo.shape[2], r)
    ad = _adapter(tmp_path, {f"base_model.model.{qn}.lora_A.weight": Aq, f"base_model.model.{qn}.lora_B.weight": Bq,
                             f"base_model.model.{on}.lora_A.default.weight": Ao, f"base_model.model.{on}.lora_B.default.weight": Bo})
    out = merge_lora_state_dict(sd, ad)
    # y = x W_flat + (alpha/r) (x A^T) B^T  for any x: the merged kernel reproduces base + adapter path
    x = torch.randn(3, wq.shape[0]).double()
    want = x @ wq.double().reshape(wq.shape[0], -1) + 2.0 * (x @ Aq.double().t()) @ Bq.double().t()
    got = x @ out[qn + ".weight"].double().reshape(wq.shape[0], -1)
    assert out[qn + ".weight"].shape == wq.shape and (got - want).abs().max() <= 1e-5
    xo = torch.randn(3, Ao.shape[1]).double()
    want = xo @ wo.double().reshape(-1, wo.shape[2]) + 2.0 * (xo @ Ao.double().t()) @ Bo.double().t()
    assert (xo @ out[on + ".weight"].double().reshape(-1, wo.shape[2]) - want).abs().max() <= 1e-5
    untouched = [k for k in sd if k not in (qn + ".weight", on + ".weight")]
    assert all(out[k] is sd[k] for k in untouched)
    # errors: unknown module, rank mismatch, missing half, empty adapter
    with pytest.raises(RuntimeError):
        merge_lora_state_dict(sd, _adapter(tmp_path, {"base_model.model.nope.lora_A.weight": Aq, "base_model.model.nope.lora_B.weight": Bq}))
    with pytest.raises(RuntimeError):
        merge_lora_state_dict(sd, _adapter(tmp_path, {f"{qn}.lora_A.weight": Aq, f"{qn}.lora_B.weight": Bq}, r=8))
    with pytest.raises(RuntimeError):
        merge_lora_state_dict(sd, _adapter(tmp_path, {f"{qn}.lora_A.weight": Aq}))
    with pytest.raises(RuntimeError):
        merge_lora_state_dict(sd, _adapter(tmp_path, {"something.else": Aq}))
    with pytest.raises(FileNotFoundError):
        merge_lora_state_dict(sd, str(tmp_path / "missing"))


def test_cli_argument_surface(capsys):
    sys.path.insert(0, ROOT)
    import cli
    p = cli.build_parser()
    a = p.parse_args(["[S1] hi", "--output", "o.wav"])
    # the reference's flags and defaults (cli.py:36-98)
    assert (a.model_path, a.cfg_scale, a.temperature, a.top_p, a.cfg_filter_top_k, a.max_tokens, a.seed) == \
        ("nari-labs/Dia-1.6B", 3.0, 1.3, 0.95, 35, None, None)
    for flag in ("--config", "--pruned-checkpoint", "--adapter-path", "--audio-prompt", "--audio-prompt-text", "--device", "--compute-dtype", "--verbose"):
        assert any(flag in act.option_strings for act in p._actions), flag
    with pytest.raises(SystemExit):
        cli.main(["x", "--output", "o.wav", "--audio-prompt", "p.wav"])          # transcript required (cli.py:103-104)
    with pytest.raises(SystemExit):
        cli.main(["x"])                                                          # no output of any kind
    with pytest.raises(SystemExit):
        cli.main(["x", "--output", "o.wav", "--pruned-checkpoint", "m.bin"])     # config required (cli.py:105-106)


def test_unstructured_pruning_matches_reference(golden, tmp_path):
    """pruning_utils.py:42-62 through the reference itself (tests/golden/make_golden.py): identical zero
    pattern, and the offline_prune.py tool end to end on a local model directory."""
    from dia_hip.pruning import sparsity, unstructured_prune_state_dict
    from dia_hip.weights import param_shapes
    g = golden("ref_unstructured.npz")
    cfg = C.tiny_config()
    gen = torch.Generator().manual_seed(int(g["seed"]))
    sd = {k: (torch.randn(shp, generator=gen) * 0.05 if not k.endswith("norm.weight") else torch.ones(shp)) for k, shp in param_shapes(cfg).items()}
    psd = unstructured_prune_state_dict(cfg, sd, float(g["amount"]))
    for k in [k[len("zero__"):] for k in g.files if k.startswith("zero__")]:
        assert np.array_equal(np.packbits((psd[k] == 0).numpy().reshape(-1)), g["zero__" + k]), k
    assert abs(sparsity(cfg, psd) - float(g["sparsity"])) < 1e-9
    assert all(torch.equal(psd[k], sd[k]) for k in sd if k.endswith("norm.weight") or "embedding" in k)
    # the tool: directory in, directory out (offline_prune.py:82-156)
    sys.path.insert(0, ROOT)
    import offline_prune
    src = tmp_path / "m"; src.mkdir()
    torch.save(sd, src / "pytorch_model.bin"); cfg.save(str(src / "config.json"))
    assert offline_prune.main(["--model-path", str(src), "--output-dir", str(tmp_path / "u"), "--prune-mode", "unstructured", "--prune-amount", "0.3"]) == 0
    got = torch.load(tmp_path / "u" / "pytorch_model.bin", weights_only=True)
    assert all(torch.equal(got[k], psd[k]) for k in psd) and (tmp_path / "u" / "config.json").exists()
    assert offline_prune.main(["--model-path", str(src), "--output-dir", str(tmp_path / "s"), "--prune-mode", "structured", "--prune-amount", "0.5"]) == 0
    from dia_hip.pruning import structured_prune_state_dict
    want, _ = structured_prune_state_dict(cfg, sd, 0.5, dim=0, n=2)
    got = torch.load(tmp_path / "s" / "pytorch_model.bin", weights_only=True)
    assert all(torch.equal(got[k], want[k]) for k in want)
    assert offline_prune.main(["--model-path", str(src), "--output-dir", str(tmp_path / "x"), "--prune-mode", "structured", "--prune-amount", "1.5"]) == 1
    assert offline_prune.main(["--model-path", str(tmp_path / "none"), "--output-dir", str(tmp_path / "x"), "--prune-mode", "structured", "--prune-amount", "0.5"]) == 1
