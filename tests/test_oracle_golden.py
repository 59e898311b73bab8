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
