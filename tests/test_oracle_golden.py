"""The CPU oracle against the vectors recorded from the (shimmed) reference — tests/golden/make_golden.py.
Runs on CPU in well under a minute."""
import numpy as np
import pytest
import torch

from dia_hip import config as C
from dia_hip.weights import synthetic_state_dict
from oracle import dia_oracle as O

CFGS = {"tiny": (C.tiny_config, 0.08), "mid": (C.mid_config, 0.02)}


@pytest.mark.parametrize("name", ["tiny", "mid"])
@pytest.mark.parametrize("mirror", [False, True])
def test_generate_matches_reference(golden, name, mirror):
    g = golden(f"ref_{name}.npz")
    mk, std = CFGS[name]
    cfg = mk()
    torch.set_num_threads(1)
    sd = synthetic_state_dict(cfg, seed=int(g["weight_seed"]), std=float(g["weight_std"]))
    if mirror and name == "mid":
        pytest.skip("mirror mode on mid is covered by make_golden.py; keep the CPU suite short")
    r = O.generate(sd, cfg, str(g["text"]), max_tokens=int(g["max_tokens"]), seed=int(g["seed"]), mirror=mirror)
    assert len(r.logits) == int(g["n_steps"])
    assert np.array_equal(r.tokens, g["tokens"])                      # token ids bit-exact
    for i, s in enumerate(g["logit_steps"]):
        assert np.abs(r.logits[int(s)] - g["logits"][i]).max() <= 1e-5
    assert np.array_equal(r.codes, g["codes"])
    assert np.array_equal(O.revert_delay_and_trim(r.codes, O.Dims.of(cfg)), g["codec_input"])


@pytest.mark.parametrize("name", ["tiny", "mid"])
def test_encoder_and_cross_kv(golden, name):
    g = golden(f"ref_{name}.npz")
    mk, std = CFGS[name]
    cfg = mk()
    dm = O.Dims.of(cfg)
    sd = synthetic_state_dict(cfg, seed=1234, std=std)
    ids = g["text_ids"].astype(np.int64)
    st = O.prepare(sd, dm, ids, mirror=False)
    L = int(g["L"])
    assert st.L == L
    k0, v0 = st.cross[0]
    k1, v1 = st.cross[-1]
    assert np.abs(k0[0].numpy() - g["cross_k_first"]).max() <= 1e-5
    assert np.abs(v0[0].numpy() - g["cross_v_first"]).max() <= 1e-5
    assert np.abs(k1[0].numpy() - g["cross_k_last"]).max() <= 1e-5
    assert np.abs(v1[0].numpy() - g["cross_v_last"]).max() <= 1e-5
    pos = torch.arange(L, dtype=torch.float32)[None]
    enc = O.encoder_forward(sd, dm, torch.from_numpy(ids[:L])[None], pos, None)
    assert np.abs(enc[0].numpy() - g["enc_out_cond"]).max() <= 1e-5


def test_noise_stream(golden):
    g = golden("ref_tiny.npz")
    nz = O.exp_noise(42, int(g["n_steps"]), 9, 1028)
    assert np.array_equal(nz[0, 0, :8].numpy(), g["noise_first8"])
    assert abs(nz.double().sum().item() - float(g["noise_checksum"])) < 1e-6


def test_sampler_cases(golden):
    g = golden("ref_sampler.npz")
    for i in range(int(g["n"])):
        T_, tp, tk = g[f"params_{i}"]
        tk = None if tk < 0 else int(tk)
        out = O.sample_next_token(torch.from_numpy(g[f"logits_{i}"]), float(T_), float(tp), tk,
                                  noise=torch.from_numpy(g[f"noise_{i}"]))
        assert np.array_equal(out.numpy(), g[f"out_{i}"]), i


def test_text_prep(golden):
    g = golden("ref_textprep.npz")
    dm = O.Dims.of(C.tiny_config())
    for i in range(int(g["n"])):
        eff = O.effective_text(str(g[f"text_{i}"]))
        assert eff == str(g[f"eff_{i}"])
        assert np.array_equal(O.text_tokens(eff, dm), g[f"ids_{i}"])
    pre, step = O.delayed_prefill(dm)
    assert step == int(g["prefill_step"]) and np.array_equal(pre, g["prefill"])


def test_prompt_assembly_pinned_on_reference_generate(golden):
    g = golden("ref_efftext.npz")
    dm = O.Dims.of(C.tiny_config())
    for i in range(int(g["n"])):
        eff = O.effective_text(str(g[f"text_{i}"]), str(g[f"ptext_{i}"]) or None)
        assert eff == str(g[f"eff_{i}"]), i
        assert np.array_equal(O.text_tokens(eff, dm), g[f"ids_{i}"]), i


def test_pruned_model(golden):
    g = golden("ref_pruned_mid.npz")
    from dia_hip.pruning import structured_prune_state_dict
    cfg = C.mid_config()
    sd = synthetic_state_dict(cfg, seed=1234, std=0.02)
    psd, keep = structured_prune_state_dict(cfg, sd, amount=0.5, dim=0, n=2)
    for k in keep:
        assert np.array_equal(keep[k], g["keep__" + k]), k
    r = O.generate(psd, cfg, str(golden("ref_mid.npz")["text"]), max_tokens=int(g["max_tokens"]), seed=42, mirror=False)
    assert np.array_equal(r.tokens, g["tokens"])
    for i, s in enumerate(g["logit_steps"]):
        assert np.abs(r.logits[int(s)] - g["logits"][i]).max() <= 1e-5


def test_audio_prompt_token_buffer(golden):
    """model.py:291-353 with a prompt: the reference's own _prepare_audio_prompt outputs"""
    from dia_hip.tokens import delayed_prefill
    g = golden("ref_prompt.npz")
    cfg = C.tiny_config()
    for i in range(int(g["n"])):
        codes = g[f"codes_{i}"]
        for got, step in (O.delayed_prefill(O.Dims.of(cfg), codes), delayed_prefill(cfg, codes)):
            assert step == int(g[f"step_{i}"]) and np.array_equal(got, g[f"delayed_{i}"])


def test_audio_prompt_replay_semantics():
    """The prompt rows go through the decode step exactly like generated rows (oracle.generate docstring):
    a run prompted with rows the model generated itself continues with the same K/V cache contents as the
    run that generated them — checked through the logits of the first sampled step."""
    cfg = C.tiny_config()
    dm = O.Dims.of(cfg)
    torch.set_num_threads(1)
    sd = synthetic_state_dict(cfg, seed=1234, std=0.08)
    text = "[S1] Hello there. [S2] Hi!"
    tp = 6
    codes = np.random.RandomState(3).randint(0, 1024, size=(tp, dm.C)).astype(np.int32)
    mt = tp + 1 + 20
    nz = O.exp_noise(9, mt - 1, dm.C, dm.tgt_vocab)
    r = O.generate(sd, cfg, text, max_tokens=mt, noise=nz, audio_prompt=codes, audio_prompt_text="[S1] Before.")
    assert r.prefill_step == tp + 1 and len(r.logits) == mt - 1 - tp
    # teacher-forced run over the same buffer without a prompt: step P of it sees the same cache and input row
    eff = O.effective_text(text, "[S1] Before.")
    st = O.prepare(sd, dm, O.text_tokens(eff, dm), False)
    for cur in range(1, tp + 2):
        lg = O.decode_step(sd, st, r.tokens[cur - 1], cur)
    assert np.abs(lg.numpy() - r.logits[0]).max() <= 1e-6
    # rows of the prompt are never overwritten; the slice handed on starts after them (model.py:831)
    pre, P = O.delayed_prefill(dm, codes)
    assert np.array_equal(r.tokens[:P], pre[:P]) and r.codes.shape[0] == r.last_step - P + 1
