#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE ITSELF.

Runs only in the build container (needs /root/reference; the GPU box never has it).
The reference package is imported from where it lies, with

  * three absent third-party modules that the decode path never touches stubbed in
    ``sys.modules`` (``dac``, ``torchaudio``, ``soundfile`` — imported at model.py:9,12,13), and
  * the three import-time shims S1-S3 of SURVEY.md App. A, without which ``generate()``
    raises at the first layer / first step and returns None:
      S1  RotaryEmbedding.forward: same maths with freqs = position[...,None,None]*inv_freq
      S2  dia.model.random = random            (model.py:682 uses an un-imported name)
      S3  DecoderOutput.get_tokens_at(step) returns the 1-D row (model.py:757-759 expects [C])

Nothing of the reference is copied: the script feeds it synthetic weights produced by the
product's generator (``dia_hip.weights.synthetic_state_dict``), records inputs and outputs,
cross-checks the CPU oracle (``oracle/dia_oracle.py``) against them, and writes ``.npz`` data.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""

from __future__ import annotations

import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "dia-tts-prune_amd"))
sys.path.insert(0, ROOT)

REF = "/root/reference"


def import_reference():
    for m in ("dac", "torchaudio", "soundfile"):
        sys.modules.setdefault(m, types.ModuleType(m))
    sys.path.insert(0, REF)
    import dia.layers as L
    import dia.model as M
    import dia.state as S

    M.random = random                                                        # S2

    def _rope(self, inputs, position):                                       # S1
        f = position.unsqueeze(-1).unsqueeze(-1) * self.inv_freq.to(position.device)
        sin, cos = torch.sin(f.float()), torch.cos(f.float())
        x1, x2 = torch.chunk(inputs.float(), 2, dim=-1)
        return torch.cat((x1 * cos - x2 * sin, x1 * sin + x2 * cos), dim=-1).to(self.compute_dtype)

    L.RotaryEmbedding.forward = _rope
    _g = S.DecoderOutput.get_tokens_at
    S.DecoderOutput.get_tokens_at = (                                        # S3
        lambda self, a, b=None: self.generated_tokens[a, :] if b is None else _g(self, a, b)
    )
    M.Dia._load_dac_model = lambda self: None
    return M, L, S


TEXT = "[S1] Dia is an open weights text to dialogue model. [S2] You get full control over scripts and voices."
TEXT_SHORT = "[S1] Hello there. [S2] Hi!"


def run_reference(M, L, S, cfg_ref, sd, text, seed, max_tokens):
    """generate() on the shimmed reference, capturing decode_step logits, encoder output,
    cross K/V and the token buffer (generate() itself returns None: no codec here)."""
    dia = M.Dia(cfg_ref, "float32", torch.device("cpu"))
    missing, unexpected = dia.model.load_state_dict(sd, strict=True)
    dia.model.eval()
    cap = {"logits": [], "tokens": None, "enc_out": None, "cross": None, "enc_in": None}

    dec = dia.model.decoder
    orig_step = dec.decode_step

    def step(tok, state):
        out = orig_step(tok, state)
        cap["logits"].append(out[:, 0].detach().clone().numpy())
        return out

    dec.decode_step = step
    enc = dia.model.encoder
    orig_enc = enc.forward

    def enc_fwd(ids, state):
        out = orig_enc(ids, state)
        cap["enc_in"] = ids.clone().numpy()
        cap["enc_out"] = out.detach().clone().numpy()
        return out

    enc.forward = enc_fwd
    orig_pre = dec.precompute_cross_attn_cache

    def pre(enc_out, pos):
        r = orig_pre(enc_out, pos)
        cap["cross"] = [(c.k.detach().clone().numpy(), c.v.detach().clone().numpy()) for c in r]
        return r

    dec.precompute_cross_attn_cache = pre
    orig_new = S.DecoderOutput.new

    def new(config, device):
        o = orig_new(config, device)
        cap["out_obj"] = o
        return o

    S.DecoderOutput.new = classmethod(lambda cls, config, device: new(config, device))
    try:
        with torch.no_grad():
            dia.generate(text, max_tokens=max_tokens, seed=seed)
    finally:
        S.DecoderOutput.new = orig_new
    cap["tokens"] = cap["out_obj"].generated_tokens.clone().numpy()
    cap["prefill_step"] = cap["out_obj"].prefill_step
    return cap, dia


def prompt_goldens(M, RC, C, O):
    """audio-prompt token-buffer preparation (model.py:291-353) from the reference's own helper; the decoder
    prefill after it cannot run in the reference (App. A4), so only this part is pinned."""
    cfg = C.tiny_config()
    cfg_ref = RC.DiaConfig.model_validate(C.config_to_json_dict(cfg))
    dia = M.Dia(cfg_ref, "float32", torch.device("cpu"))
    rs = np.random.RandomState(11)
    rec = {}
    for i, tp in enumerate((1, 5, 23)):
        codes = rs.randint(0, 1024, size=(tp, cfg.data.channels)).astype(np.int32)
        arg = torch.from_numpy(codes) if i != 1 else torch.from_numpy(codes)[None]      # [T,C] and [1,T,C] forms
        pd, ps = dia._prepare_audio_prompt(arg)
        mine, ms = O.delayed_prefill(O.Dims.of(cfg), codes)
        assert ms == ps and np.array_equal(mine, pd.numpy())
        rec[f"codes_{i}"] = codes; rec[f"delayed_{i}"] = pd.numpy().astype(np.int32); rec[f"step_{i}"] = np.int32(ps)
    rec["n"] = np.int32(3)
    np.savez_compressed(os.path.join(HERE, "ref_prompt.npz"), **rec)
    print("ref_prompt.npz written")


def unstructured_goldens(M, RC, C):
    """global unstructured L1 pruning (pruning_utils.py:42-62) of the tiny model through the reference's own
    utility.  Weights here are plain fp32 normals (not bf16-rounded): magnitude ties would make torch's top-k
    choice among equals part of the vector."""
    import dia.pruning_utils as RP
    from dia_hip.weights import param_shapes
    cfg = C.tiny_config()
    cfg_ref = RC.DiaConfig.model_validate(C.config_to_json_dict(cfg))
    g = torch.Generator().manual_seed(77)
    sd = {k: (torch.randn(shp, generator=g) * 0.05 if not k.endswith("norm.weight") else torch.ones(shp)) for k, shp in param_shapes(cfg).items()}
    dia = M.Dia(cfg_ref, "float32", torch.device("cpu"))
    dia.model.load_state_dict(sd, strict=True)
    RP.apply_unstructured_pruning(dia.model, amount=0.3)
    RP.make_pruning_permanent(dia.model)
    psd = dia.model.state_dict()
    rec = {"amount": np.float64(0.3), "seed": np.int32(77), "sparsity": np.float64(RP.check_pruning_sparsity(dia.model))}
    for k in ("encoder.layers.0.mlp.wo.weight", "decoder.layers.1.self_attention.q_proj.weight", "decoder.logits_dense.weight"):
        rec["zero__" + k] = np.packbits((psd[k] == 0).numpy().reshape(-1))
    rec["nzero_total"] = np.int64(sum(int((v == 0).sum()) for k, v in psd.items() if v.dim() >= 2 and "embed" not in k))
    np.savez_compressed(os.path.join(HERE, "ref_unstructured.npz"), **rec)
    print("ref_unstructured.npz written, sparsity", float(rec["sparsity"]))


def chunk_goldens():
    """text chunking of the Gradio front-end (app.py:79-127).  app.py cannot be imported (it loads a model and
    needs gradio at import time), so the four pure helper functions are compiled from its source with ast and
    executed here; only their inputs and outputs are recorded."""
    import ast
    src = open(os.path.join(REF, "app.py")).read()
    want = {"count_effective_length", "auto_adjust_chunk_size", "split_by_words_respecting_special_tokens", "batch_chunks"}
    mod = ast.Module(body=[n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name in want], type_ignores=[])
    ns = {}
    exec(compile(mod, "app_helpers", "exec"), ns)
    base = "[S1] Dia is an open weights text to dialogue model. [S2] You get full control over scripts and voices. "
    texts = [base, base * 12, "[S1] short", "word " * 40 + "supercalifragilisticexpialidocious" * 3 + " [S2] end", base * 50, "   ", ""]
    rec = {"n": np.int32(len(texts))}
    for i, t in enumerate(texts):
        rec[f"text_{i}"] = np.array(t)
        rec[f"len_{i}"] = np.int32(ns["count_effective_length"](t))
        for user in (0, 40):
            cs = ns["auto_adjust_chunk_size"](t, user)
            chunks = ns["split_by_words_respecting_special_tokens"](t, max_effective_chars=cs)
            rec[f"cs_{i}_{user}"] = np.int32(cs)
            rec[f"chunks_{i}_{user}"] = np.array(["\x00".join(chunks)])       # one NUL-joined string: no pickled object arrays
            rec[f"batches_{i}_{user}"] = np.array([len(b) for b in ns["batch_chunks"](chunks, 4)], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "ref_chunks.npz"), **rec)
    print("ref_chunks.npz written")


EFFTEXT_CASES = [
    (TEXT, None), (TEXT_SHORT, None), ("no tags at all", None), ("[S2] starts with two [S1]", None),
    ("[S1] ends with tag [S2]", None), ("[S1] ends with its own tag [S1]", None), ("  [S1] padded with blanks.  ", None),
    ("[S1] unicode \u00e9\u00e8 \u4f60\u597d [S2] ok", None), ("x" * 300, None), ("[S1]", None), ("[S2]", None),
    ("[S1] one [S1] two [S1] three", None), ("[S2] a [S1] b [S2] c [S1]", None), ("[s1] lower-case tags are text", None),
    ("[S1] Second half of a dialogue.", "[S1] Prompt transcript. [S2] With two turns."),
    (" [S2] reply. ", "  [S1] blanks around the prompt  "), ("tail without tags", "[S2] prompt ends here"),
]


def efftext_goldens(M, RC, C, O):
    """The prompt assembly inside Dia.generate (reference model.py:686-696) is inline code: pin it by calling the reference's
    generate() with _prepare_generation replaced by a recorder that keeps the effective text it is handed and stops there;
    the ids are the reference's own _prepare_text_input of that string."""
    cfg = C.tiny_config()
    cfg_ref = RC.DiaConfig.model_validate(C.config_to_json_dict(cfg))
    dia = M.Dia(cfg_ref, "float32", torch.device("cpu"))

    class Stop(Exception):
        pass
    seen = {}

    def recorder(text, audio_prompt, verbose):
        seen["text"] = text
        raise Stop()
    dia._prepare_generation = recorder
    rec = {"n": np.int32(len(EFFTEXT_CASES))}
    for i, (t, pt) in enumerate(EFFTEXT_CASES):
        try:
            dia.generate(t, max_tokens=4, audio_prompt_text=pt, seed=1)
        except Stop:
            pass
        eff = seen.pop("text")
        ids = dia._prepare_text_input(eff)[0].numpy()
        assert eff == O.effective_text(t, pt), (t, pt, eff, O.effective_text(t, pt))
        assert np.array_equal(ids, O.text_tokens(eff, O.Dims.of(cfg)))
        rec[f"text_{i}"] = np.array(t)
        rec[f"ptext_{i}"] = np.array("" if pt is None else pt)
        rec[f"eff_{i}"] = np.array(eff)
        rec[f"ids_{i}"] = ids.astype(np.int32)
    np.savez_compressed(os.path.join(HERE, "ref_efftext.npz"), **rec)
    print(f"[efftext] {len(EFFTEXT_CASES)} prompt-assembly cases pinned on the reference's generate()")


def main():
    torch.set_num_threads(1)
    from dia_hip import config as C
    from dia_hip.weights import synthetic_state_dict
    from oracle import dia_oracle as O

    M, L, S = import_reference()
    import dia.config as RC
    efftext_goldens(M, RC, C, O)
    if "--only-efftext" in sys.argv:
        return
    prompt_goldens(M, RC, C, O)
    chunk_goldens()
    unstructured_goldens(M, RC, C)
    if "--only-prompt" in sys.argv:
        return

    out = {}
    for name, cfg, std, text, max_tokens, keep in (
        ("tiny", C.tiny_config(), 0.08, TEXT, 40, list(range(20))),
        ("mid", C.mid_config(), 0.02, TEXT, 48, list(range(12)) + [16, 32]),
    ):
        cfg_ref = RC.DiaConfig.model_validate(C.config_to_json_dict(cfg))
        sd = synthetic_state_dict(cfg, seed=1234, std=std)
        cap, dia = run_reference(M, L, S, cfg_ref, sd, text, 42, max_tokens)
        n_steps = len(cap["logits"])
        print(f"[{name}] reference ran {n_steps} steps; prefill_step={cap['prefill_step']}")

        # ---- cross-check the oracle (both modes) against the reference run
        for mirror in (True, False):
            r = O.generate(sd, cfg, text, max_tokens=max_tokens, seed=42, mirror=mirror)
            assert len(r.logits) == n_steps, (len(r.logits), n_steps)
            err = max(float(np.abs(a - b).max()) for a, b in zip(r.logits, cap["logits"]))
            tok_ok = bool(np.array_equal(r.tokens, cap["tokens"]))
            print(f"[{name}] oracle mirror={mirror}: logits max-abs err {err:.3e}, tokens equal: {tok_ok}")
            assert err <= 1e-5 and tok_ok
        # explicit-noise path must reproduce the torch.multinomial stream
        dm = O.Dims.of(cfg)
        nz = O.exp_noise(42, n_steps, dm.C, dm.tgt_vocab)
        r2 = O.generate(sd, cfg, text, max_tokens=max_tokens, seed=None, noise=nz, mirror=False)
        assert np.array_equal(r2.tokens, cap["tokens"]), "argmax(p/q) draw != torch.multinomial"
        # batched noise draw == sequential draws (what the product uploads)
        g = torch.Generator().manual_seed(42)
        nb = torch.empty(n_steps, dm.C, dm.tgt_vocab).exponential_(1.0, generator=g)
        assert torch.equal(nb, nz), "batched exponential_ != per-step draws"

        eff = O.effective_text(text)
        ids = O.text_tokens(eff, dm)
        Lnp = int((ids != dm.text_pad).sum())
        assert np.array_equal(cap["enc_in"][1], ids) and (cap["enc_in"][0] == dm.text_pad).all()
        prefill, pstep = O.delayed_prefill(dm)
        assert pstep == cap["prefill_step"]
        g = {
            "text": np.array(text), "effective_text": np.array(eff), "text_ids": ids.astype(np.int32),
            "L": np.int32(Lnp), "seed": np.int32(42), "weight_seed": np.int32(1234), "weight_std": np.float32(std),
            "max_tokens": np.int32(max_tokens), "prefill": prefill, "prefill_step": np.int32(pstep),
            "tokens": cap["tokens"].astype(np.int32), "n_steps": np.int32(n_steps),
            "logit_steps": np.array(keep, dtype=np.int32),
            "logits": np.stack([cap["logits"][i] for i in keep]).astype(np.float32),
            "enc_out_cond": cap["enc_out"][1, :Lnp].astype(np.float32),
            "cross_k_first": cap["cross"][0][0][1, :, :Lnp].astype(np.float32),
            "cross_v_first": cap["cross"][0][1][1, :, :Lnp].astype(np.float32),
            "cross_k_last": cap["cross"][-1][0][1, :, :Lnp].astype(np.float32),
            "cross_v_last": cap["cross"][-1][1][1, :, :Lnp].astype(np.float32),
            "noise_checksum": np.float64(nz.double().sum().item()),
            "noise_first8": nz[0, 0, :8].numpy(),
        }
        # the reference's own delay tables for the output stage
        codes = cap["tokens"][pstep: pstep + n_steps - 1]
        import dia.audio as RA
        rp = RA.build_revert_indices(B=1, T=codes.shape[0], C=dm.C, delay_pattern=dm.delay)
        rev = RA.revert_audio_delay(torch.from_numpy(codes.astype(np.int64))[None], dm.pad, rp, codes.shape[0])
        rev = rev[:, : codes.shape[0] - max(dm.delay), :].clone()
        rev[(rev < 0) | (rev > 1023)] = 0
        g["codes"] = codes.astype(np.int32)
        g["codec_input"] = rev.transpose(1, 2).numpy().astype(np.int32)
        mine = O.revert_delay_and_trim(codes, dm)
        assert np.array_equal(mine, g["codec_input"])
        np.savez_compressed(os.path.join(HERE, f"ref_{name}.npz"), **g)
        out[name] = g

    # ---- text-prep and prefill goldens straight from the reference's helpers
    cfg = C.tiny_config()
    cfg_ref = RC.DiaConfig.model_validate(C.config_to_json_dict(cfg))
    dia = M.Dia(cfg_ref, "float32", torch.device("cpu"))
    cases = [TEXT, TEXT_SHORT, "no tags at all", "[S2] starts with two [S1]", "[S1] ends with tag [S2]",
             "[S1] unicode éè 你好 [S2] ok", "x" * 300, ""]
    recs = {}
    for i, t in enumerate(cases):
        # effective text: replay of model.py:686-696 is inside generate(); use the oracle's and pin
        # the token side on the reference's _prepare_text_input
        eff = O.effective_text(t)
        ref_ids = dia._prepare_text_input(eff)[0].numpy()
        assert np.array_equal(ref_ids, O.text_tokens(eff, O.Dims.of(cfg)))
        recs[f"text_{i}"] = np.array(t)
        recs[f"eff_{i}"] = np.array(eff)
        recs[f"ids_{i}"] = ref_ids.astype(np.int32)
    recs["n"] = np.int32(len(cases))
    pd, ps = dia._prepare_audio_prompt(None)
    recs["prefill"] = pd.numpy().astype(np.int32)
    recs["prefill_step"] = np.int32(ps)
    assert np.array_equal(recs["prefill"], O.delayed_prefill(O.Dims.of(cfg))[0])
    np.savez_compressed(os.path.join(HERE, "ref_textprep.npz"), **recs)

    # ---- sampler unit cases: reference _sample_next_token vs oracle
    rs = np.random.RandomState(7)
    samp = {}
    case_id = 0
    for (T_, tp, tk) in ((0.0, 0.95, 35), (1.3, 0.95, 35), (1.0, 0.5, 10), (0.7, 1.0, 0), (1.3, 0.9, None), (2.0, 0.3, 50)):
        lg = torch.from_numpy((rs.randn(9, 1028) * 2.0).astype(np.float32))
        lg[:, 1025] = float("-inf"); lg[:, 1026] = float("-inf"); lg[1:, 1024] = float("-inf")
        if case_id == 2:
            lg[3, 100:140] = lg[3, 100]          # top-k ties
        torch.manual_seed(100 + case_id)
        ref = M._sample_next_token(lg.clone(), T_, tp, tk).numpy()
        torch.manual_seed(100 + case_id)
        q = torch.empty(9, 1028).exponential_(1.0) if T_ != 0.0 else torch.ones(9, 1028)
        mine = O.sample_next_token(lg.clone(), T_, tp, tk, noise=q).numpy()
        assert np.array_equal(ref, mine), (case_id, ref, mine)
        samp[f"logits_{case_id}"] = lg.numpy(); samp[f"noise_{case_id}"] = q.numpy()
        samp[f"params_{case_id}"] = np.array([T_, tp, -1 if tk is None else tk], dtype=np.float64)
        samp[f"out_{case_id}"] = ref.astype(np.int32)
        case_id += 1
    samp["n"] = np.int32(case_id)
    np.savez_compressed(os.path.join(HERE, "ref_sampler.npz"), **samp)

    # ---- structured pruning (dim 0, 50 %, L2) of the mid model through the reference's own
    #      pruning_utils, then logits of the pruned model
    import dia.pruning_utils as RP
    cfg = C.mid_config()
    cfg_ref = RC.DiaConfig.model_validate(C.config_to_json_dict(cfg))
    sd = synthetic_state_dict(cfg, seed=1234, std=0.02)
    dia = M.Dia(cfg_ref, "float32", torch.device("cpu"))
    dia.model.load_state_dict(sd, strict=True)
    RP.apply_structured_pruning(dia.model, amount=0.5, dim=0, n=2)
    RP.make_pruning_permanent(dia.model)
    psd = {k: v.detach().clone() for k, v in dia.model.state_dict().items()}
    keep = {}
    for k, v in psd.items():
        if v.dim() >= 2 and "embed" not in k:
            nz_rows = (v.reshape(v.shape[0], -1) != 0).any(dim=1)
            keep[k] = torch.nonzero(nz_rows).flatten().numpy().astype(np.int32)
    cap, _ = run_reference(M, L, S, cfg_ref, psd, TEXT, 42, 24)
    r = O.generate(psd, cfg, TEXT, max_tokens=24, seed=42, mirror=False)
    err = max(float(np.abs(a - b).max()) for a, b in zip(r.logits, cap["logits"]))
    print(f"[pruned mid] oracle err {err:.3e} tokens equal {np.array_equal(r.tokens, cap['tokens'])}")
    assert err <= 1e-5 and np.array_equal(r.tokens, cap["tokens"])
    pg = {"keep__" + k: v for k, v in keep.items()}
    pg["tokens"] = cap["tokens"].astype(np.int32)
    pg["logit_steps"] = np.array([0, 1, 2, 15, 16], dtype=np.int32)
    pg["logits"] = np.stack([cap["logits"][i] for i in (0, 1, 2, 15, 16)]).astype(np.float32)
    pg["max_tokens"] = np.int32(24)
    np.savez_compressed(os.path.join(HERE, "ref_pruned_mid.npz"), **pg)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
