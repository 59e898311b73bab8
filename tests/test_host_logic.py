"""Host-side logic of the product package (no GPU): token helpers against the reference's vectors,
config round trip, layouts, synthetic weights."""
import json
import os

import numpy as np
import pytest
import torch

from dia_hip import config as C
from dia_hip import layout as lay
from dia_hip import tokens as T
from dia_hip import weights as W


def test_text_helpers_match_reference(golden):
    g = golden("ref_textprep.npz")
    cfg = C.tiny_config()
    for i in range(int(g["n"])):
        eff = T.effective_text(str(g[f"text_{i}"]))
        assert eff == str(g[f"eff_{i}"])
        ids = T.encode_text(eff, cfg)
        assert np.array_equal(T.padded_text_ids(ids, cfg), g[f"ids_{i}"])
    pre, step = T.delayed_prefill(cfg)
    assert step == int(g["prefill_step"]) and np.array_equal(pre, g["prefill"])


def test_prompt_assembly_matches_reference_generate(golden):
    """ref_efftext.npz: the effective text the REFERENCE's generate() hands to _prepare_generation (model.py:686-696 is inline
    code there) for 17 tag / blank / prompt-transcript cases, and the reference's ids of it"""
    g = golden("ref_efftext.npz")
    cfg = C.tiny_config()
    assert int(g["n"]) >= 17
    for i in range(int(g["n"])):
        pt = str(g[f"ptext_{i}"]) or None
        eff = T.effective_text(str(g[f"text_{i}"]), pt)
        assert eff == str(g[f"eff_{i}"]), i
        assert np.array_equal(T.padded_text_ids(T.encode_text(eff, cfg), cfg), g[f"ids_{i}"]), i


def test_effective_text_edges():
    assert T.effective_text("") == ""
    assert T.effective_text("  hello ") == "hello [S2]"
    assert T.effective_text("[S1] a [S2]") == "[S1] a [S2] [S1]"
    assert T.effective_text("[S1] a [S2] b [S1]") == "[S1] a [S2] b [S1] [S2]"
    assert T.effective_text("x", "[S1] p") == "[S1] p x [S2]"


@pytest.mark.parametrize("name", ["tiny", "mid"])
def test_codec_input_matches_reference(golden, name):
    g = golden(f"ref_{name}.npz")
    cfg = C.tiny_config() if name == "tiny" else C.mid_config()
    assert np.array_equal(T.codes_for_codec(g["codes"], cfg), g["codec_input"])
    assert T.codes_for_codec(g["codes"][:0], cfg).shape == (1, 9, 0)
    assert T.codes_for_codec(g["codes"][:10], cfg).shape == (1, 9, 0)     # shorter than max delay


def test_prefill_with_prompt_rows():
    cfg = C.tiny_config()
    prompt = np.arange(5 * 9, dtype=np.int32).reshape(5, 9) % 1000
    pre, step = T.delayed_prefill(cfg, prompt)
    assert step == 6 and pre.shape == (21, 9)
    d = cfg.data.delay_pattern
    for c in range(9):
        for t in range(21):
            ts = t - d[c]
            want = 1026 if ts < 0 else (1026 if ts == 0 else (prompt[ts - 1, c] if ts <= 5 else 1025))
            assert pre[t, c] == want


def test_config_roundtrip(tmp_path):
    cfg = C.dia_1_6b_config()
    assert W.param_count(cfg) == 1_611_196_416 or W.param_count(cfg) // 10**6 == 1611
    p = tmp_path / "sub" / "cfg"
    cfg.save(p)
    assert (tmp_path / "sub" / "cfg.json").is_file()
    back = C.DiaConfig.load(tmp_path / "sub" / "cfg.json")
    assert back == cfg
    assert C.DiaConfig.load(tmp_path / "nope.json") is None
    d = C.config_to_json_dict(cfg)
    d["data"]["text_length"] = 1000                      # rounded up to a multiple of 128
    assert C.config_from_json_dict(d).data.text_length == 1024
    with pytest.raises(Exception):
        C.DataConfig(text_length=0, audio_length=128)
    with pytest.raises(Exception):                       # frozen
        cfg.data.channels = 3
    # hub-mixin style wrapper {"config": {...}}
    (tmp_path / "config.json").write_text(json.dumps({"config": C.config_to_json_dict(cfg)}))
    assert W.read_hub_config(str(tmp_path / "config.json")) == cfg


def test_tile_and_plane_layouts():
    torch.manual_seed(0)
    w = torch.randn(70, 45).bfloat16().float()
    t, kt, ns = lay.tile_weight(w)
    assert (kt, ns) == (3, 3) and t.shape == (3, 3, 64, 8)
    assert torch.equal(lay.untile_weight(t, 70, 45), w)
    # lane l of tile (strip, kt) holds W[32kt + 8(l>>4) + j][16 strip + (l&15)]
    for (s, k, l, j) in [(0, 0, 0, 0), (2, 1, 37, 5), (1, 2, 63, 7)]:
        kk, nn = 32 * k + 8 * (l >> 4) + j, 16 * s + (l & 15)
        want = w[kk, nn] if kk < 70 and nn < 45 else 0.0
        assert float(t[s, k, l, j]) == float(want)
    x = torch.randn(19, 70) * 3
    p = lay.pack_planes(x)
    assert p.shape == (3, 2, 3, 64, 8)
    assert torch.equal(lay.unpack_planes(p, 19, 70), x)           # hi+mid+lo is exact
    for (m, k) in [(0, 0), (17, 69), (5, 33)]:
        mt, ktile, lane, j = m >> 4, k >> 5, (m & 15) + 16 * ((k & 31) >> 3), k & 7
        assert float(p[0, mt, ktile, lane, j]) == float(x[m, k].bfloat16())


def test_interleave_and_rope_perm():
    wi = torch.arange(4 * 2 * 16, dtype=torch.float32).reshape(4, 2, 16)
    out = lay.interleave_gate_up(wi)
    assert out.shape == (4, 32)
    assert torch.equal(out[:, 0:8], wi[:, 0, 0:8]) and torch.equal(out[:, 8:16], wi[:, 1, 0:8])
    assert torch.equal(out[:, 16:24], wi[:, 0, 8:16]) and torch.equal(out[:, 24:32], wi[:, 1, 8:16])
    perm = lay.rope_pair_perm(128)
    assert perm[:4].tolist() == [0, 64, 1, 65] and sorted(perm.tolist()) == list(range(128))


def test_synthetic_weights_deterministic_and_bf16():
    cfg = C.tiny_config()
    a = W.synthetic_state_dict(cfg, seed=1234, std=0.08)
    b = W.synthetic_state_dict(cfg, seed=1234, std=0.08)
    c = W.synthetic_state_dict(cfg, seed=1235, std=0.08)
    assert list(a.keys()) == list(W.param_shapes(cfg).keys())
    for k in a:
        assert torch.equal(a[k], b[k])
        assert torch.equal(a[k], a[k].bfloat16().float())          # bf16-representable
    k = "decoder.layers.0.mlp.wo.weight"
    assert not torch.equal(a[k], c[k])
    assert abs(a[k].std().item() - 0.08) < 0.01
    # chunking does not change the stream
    t1 = W.synthetic_tensor("x", (1000,), 7, 0.02, chunk=1 << 24)
    t2 = W.synthetic_tensor("x", (1000,), 7, 0.02, chunk=64)
    assert torch.equal(t1, t2)
    assert torch.all(a["decoder.norm.weight"] == 1)


def test_checkpoint_io(tmp_path):
    cfg = C.tiny_config()
    sd = W.synthetic_state_dict(cfg, seed=1, std=0.05)
    torch.save(dict(sd, **{"x.lora_A.weight": torch.zeros(1)}), tmp_path / "pytorch_model.bin")
    cfg.save(tmp_path / "config.json")
    cpath, wpath = W.find_checkpoint_in_dir(str(tmp_path))
    back = W.load_state_dict_file(wpath)
    assert "x.lora_A.weight" not in back                           # model.py:172
    assert W.check_state_dict(cfg, back) == ([], [])
    from safetensors.torch import save_file
    save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / "model.safetensors"))
    assert W.find_checkpoint_in_dir(str(tmp_path))[1].endswith("model.safetensors")
    bad = dict(sd)
    bad["decoder.norm.weight"] = torch.ones(3)
    with pytest.raises(RuntimeError):
        W.check_state_dict(cfg, bad)
    with pytest.raises(FileNotFoundError):
        W.find_checkpoint_in_dir(str(tmp_path / "missing"))


def test_compaction_plan_from_zeros():
    from dia_hip import compact as cpt
    from dia_hip.pruning import structured_prune_state_dict
    cfg = C.mid_config()
    sd = W.synthetic_state_dict(cfg, seed=1234, std=0.02)
    psd, keep = structured_prune_state_dict(cfg, sd, amount=0.5, dim=0, n=2)
    d = cfg.model.decoder
    pre = "decoder.layers.1."
    P = cpt.plan_decoder_layer(psd, pre, d.gqa_query_heads, d.kv_heads, d.cross_query_heads)
    assert cpt.is_pruned(P)
    # structure recovered from zeros == the kept indices of the pruning pass
    kq = set(keep[pre + "self_attention.q_proj.weight"].tolist()) | set(keep[pre + "self_attention.k_proj.weight"].tolist()) \
        | set(keep[pre + "self_attention.v_proj.weight"].tolist())
    got = set(torch.nonzero(P.keep_qkv).flatten().tolist())
    assert got >= kq and len(got) % 32 == 0 and len(got) - len(kq) < 32                          # padded with zero rows
    assert all((psd[pre + f"self_attention.{n}_proj.weight"][sorted(got - kq)] == 0).all() for n in "qkv")
    assert torch.nonzero(P.live_q_heads).flatten().tolist() == keep[pre + "self_attention.o_proj.weight"].tolist()
    assert torch.nonzero(P.live_hidden).flatten().tolist() == keep[pre + "mlp.wo.weight"].tolist()
    assert int(P.live_q_heads.sum()) == d.gqa_query_heads // 2 and int(P.live_hidden.sum()) == d.n_hidden // 2
    cm = cpt._cmap(P.keep_cq)
    assert cm.max().item() == int(P.keep_cq.sum()) - 1 and (cm[~P.keep_cq] == -1).all()
    assert cpt.strips_of_heads(torch.tensor([False, True, False, True]), 256) == list(range(24, 32)) + list(range(40, 48))
    idx = cpt.pad_hidden_keep(torch.tensor([True] * 5 + [False] * 11))
    assert idx.tolist() == [0, 1, 2, 3, 4, -1, -1, -1]
    big = torch.zeros(4096, dtype=torch.bool); big[:300] = True
    assert cpt.pad_hidden_keep(big).numel() == 1024 and int((cpt.pad_hidden_keep(big) < 0).sum()) == 724   # 4096-wide: 1024 granule
    mid_ = torch.zeros(2048, dtype=torch.bool); mid_[:300] = True
    assert cpt.pad_hidden_keep(mid_).numel() == 512
    k = torch.zeros(2048, dtype=torch.bool); k[100:300] = True
    pk = cpt.pad_keep(k)                                  # full-size K: 256-row granule
    assert int(pk.sum()) == 256 and pk[100:300].all() and pk[:56].all() and not pk[56:100].any()
    w_ = torch.randn(1408, 8)
    pw = cpt.pad_rows(w_)                                 # 11 heads x 128 rows -> 1536: zero rows appended
    assert pw.shape == (1536, 8) and torch.equal(pw[:1408], w_) and (pw[1408:] == 0).all() and cpt.pad_rows(pw) is pw
    k2 = torch.zeros(512, dtype=torch.bool); k2[100:300] = True
    assert int(cpt.pad_keep(k2).sum()) == 224             # small models: whole k-tiles only
    dense = cpt.plan_decoder_layer(sd, pre, d.gqa_query_heads, d.kv_heads, d.cross_query_heads)
    assert not cpt.is_pruned(dense)


def test_cross_kv_tile_sets_are_one_weight_for_the_prefill():
    """DeviceWeights.ckv_all(): the cross-K/V tile sets of all decoder layers sit back to back in the arena and are handed to the prefill as ONE
    weight (dia_gemm_args.kv_layer_strips / kv_layer_stride); a compacted decoder brings the strip map of that merged launch; three-plane
    weights keep one launch per layer"""
    from dia_hip.engine import DeviceWeights
    from dia_hip.pruning import structured_prune_state_dict
    cfg = C.mid_config()
    d = cfg.model.decoder
    cpu = torch.device("cpu")
    sd = W.synthetic_state_dict(cfg, seed=7, std=0.02)
    w = DeviceWeights(cfg, sd, cpu)
    allw = w.ckv_all()
    per = [L["ckv"] for L in w.dec_layers]
    assert allw is not None and allw.kt == per[0].kt and allw.ns == sum(t.ns for t in per) == d.n_layer * d.cross_query_heads * 16
    assert torch.equal(allw.t, torch.cat([t.t.reshape(-1) for t in per]))
    assert allw.t.data_ptr() == per[0].t.data_ptr()                      # a view of the arena, not a copy
    lo = w.flat.data_ptr()
    assert lo <= allw.t.data_ptr() and allw.t.data_ptr() + allw.t.numel() * 2 <= lo + w.flat.numel()
    assert w.smap_ckv_all is None
    assert DeviceWeights(cfg, sd, cpu, weight_planes=3).ckv_all() is None
    # structured-pruned checkpoint -> compacted decoder: merged strip map = layer * (heads * 16) + the layer's original strips
    psd, _ = structured_prune_state_dict(cfg, sd, amount=0.5, dim=0, n=2)
    wp = DeviceWeights(cfg, psd, cpu)
    assert wp.compacted and wp.smap_ckv_all is not None and wp.smap_ckv_all.dtype == torch.int32
    pa = wp.ckv_all()
    assert pa is not None and pa.ns == wp.smap_ckv_all.numel() == sum(L["ckv"].ns for L in wp.dec_layers)
    per_layer = d.cross_query_heads * 16
    o = 0
    for i, L in enumerate(wp.dec_layers):
        seg = wp.smap_ckv_all[o: o + L["ckv"].ns]
        o += L["ckv"].ns
        assert (seg // per_layer == i).all()
        if L["smap_ckv"] is not None:
            assert torch.equal(seg - i * per_layer, L["smap_ckv"])
