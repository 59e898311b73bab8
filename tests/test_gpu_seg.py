"""The persistent MLP segment (csrc/seg.hip: cross o_proj -> wi + SwiGLU -> wo -> next q/k/v in ONE launch, M <= 4 rows)
against a float64 restatement of reference DecoderLayer.forward layers.py:574-584 (+ 541, 273-275 of the next layer),
through the C ABI; then the same inputs through the launches it replaces."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dia_hip import binding as hb
from dia_hip import layout as lay

D, F, KA, NQ = 2048, 8192, 2048, 3072
EPS = 1e-5


def bf16r(t):
    return t.bfloat16().float()


def make_case(M, seed, has_qkv, dev):
    g = torch.Generator(device="cpu").manual_seed(seed)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
    W = dict(co=bf16r(rn(KA, D, sc=0.03)), gate=bf16r(rn(D, F, sc=0.03)), up=bf16r(rn(D, F, sc=0.03)), wo=bf16r(rn(F, D, sc=0.02)),
             qkv=bf16r(rn(D, NQ, sc=0.03)) if has_qkv else None)
    a = rn(M, KA)
    x0 = rn(M, D)
    g_mlp = bf16r(1.0 + 0.1 * rn(D))
    g_next = bf16r(1.0 + 0.1 * rn(D))
    return W, a, x0, g_mlp, g_next


def reference(W, a, x0, g_mlp, g_next):
    a, x0 = a.double(), x0.double()
    x1 = x0 + a @ W["co"].double()
    n1 = x1 * torch.rsqrt((x1 ** 2).mean(-1, keepdim=True) + EPS) * g_mlp.double()
    gt, up = n1 @ W["gate"].double(), n1 @ W["up"].double()
    h = torch.nn.functional.silu(gt) * up
    x2 = x1 + h @ W["wo"].double()
    q = None
    if W["qkv"] is not None:
        n2 = x2 * torch.rsqrt((x2 ** 2).mean(-1, keepdim=True) + EPS) * g_next.double()
        q = n2 @ W["qkv"].double()
    return x2, q


def run_seg(W, a, x0, g_mlp, g_next, M, ws, reps=1):
    L = hb.lib()
    dev = a.device
    has_qkv = W["qkv"] is not None
    ring = lay.seg_ring(W["co"], W["gate"], W["up"], W["wo"], W["qkv"])
    assert ring.shape[1] == L.dia_seg_slots(int(has_qkv))
    a_t = lay.pack_f32_tiles(a, ktiles=KA // 32, mtiles=1)
    x = torch.zeros(16, D, device=dev); x[:M] = x0
    planes_x = torch.full((3, 1, D // 32, 64, 8), 0, dtype=torch.bfloat16, device=dev)      # fp32 tiles live in the planes buffer
    ssq = torch.full((D // 16, 16), float("nan"), device=dev)
    qkv = torch.full((16, NQ), float("nan"), device=dev)
    s = hb.SegArgs()
    s.a_in, s.a_ktiles, s.M, s.W, s.nslots, s.has_qkv, s.D, s.F = hb.ptr(a_t), KA // 32, M, hb.ptr(ring), ring.shape[1], int(has_qkv), D, F
    s.x, s.ldx, s.g_mlp, s.g_next, s.qkv_out, s.ldq = hb.ptr(x), D, hb.ptr(g_mlp), hb.ptr(g_next), hb.ptr(qkv), NQ
    s.planes_x, s.xkt, s.ssq, s.ssq_ld, s.eps, s.ws = hb.ptr(planes_x), D // 32, hb.ptr(ssq), 16, EPS, hb.ptr(ws)
    outs = []
    for _ in range(reps):
        x[:M] = x0
        hb.check(L.dia_seg_mlp(C.byref(s), None), "dia_seg_mlp")
        torch.cuda.synchronize()
        assert L.dia_seg_error(hb.ptr(ws), None) == 0
        outs.append((x[:M].clone(), qkv[:M].clone(), planes_x.clone(), ssq.clone()))
    return outs


@pytest.mark.parametrize("M,has_qkv", [(2, True), (2, False), (4, True), (1, True), (3, False)])
def test_seg_mlp_vs_float64(M, has_qkv):
    L = hb.lib()
    if not L.dia_seg_supported(D, F, KA, NQ):
        pytest.skip("no persistent segment on this device (needs 256 CUs)")
    dev = torch.device("cuda:0")
    W, a, x0, g_mlp, g_next = make_case(M, 100 + M + int(has_qkv), has_qkv, dev)
    ws = torch.zeros(int(L.dia_seg_workspace_bytes()), dtype=torch.uint8, device=dev)
    outs = run_seg(W, a, x0, g_mlp, g_next, M, ws, reps=3)        # three launches on one workspace: the counters carry over
    x2, q = reference(W, a, x0, g_mlp, g_next)
    for x_o, q_o, planes, ssq in outs:
        err = (x_o.double() - x2).abs().max().item()
        assert err <= 2e-5 * x2.abs().max().item(), err
        if has_qkv:
            errq = (q_o.double() - q).abs().max().item()
            assert errq <= 2e-5 * max(1.0, q.abs().max().item()), errq
        # what a launched consumer reads: fp32 tiles of x2 * g_next, strip sums of squares
        xt = planes.view(torch.uint8).view(-1)[: 1 * (D // 32) * 64 * 8 * 4].view(torch.float32).reshape(1, D // 32, 64, 8)
        assert torch.equal(lay.unpack_f32_tiles(xt, M, D), x_o * g_next)
        want = (x_o.double() ** 2).reshape(M, D // 16, 16).sum(-1).T
        assert (ssq[:, :M].double() - want).abs().max().item() <= 1e-5 * want.max().item()
    # bit-reproducible from launch to launch
    assert all(torch.equal(outs[0][i], o[i]) for o in outs[1:] for i in range(2 if has_qkv else 1))


def test_seg_mlp_alternating_forms_share_one_workspace():
    """a decode step runs 17 launches with the q/k/v stage and one without on ONE workspace: every counter must advance once
    per launch whatever the form"""
    L = hb.lib()
    if not L.dia_seg_supported(D, F, KA, NQ):
        pytest.skip("no persistent segment on this device (needs 256 CUs)")
    dev = torch.device("cuda:0")
    ws = torch.zeros(int(L.dia_seg_workspace_bytes()), dtype=torch.uint8, device=dev)
    for i, has_qkv in enumerate([True, False, True, True, False, False, True]):
        W, a, x0, g_mlp, g_next = make_case(2, 300 + i, has_qkv, dev)
        (x_o, q_o, _, _), = run_seg(W, a, x0, g_mlp, g_next, 2, ws)
        x2, q = reference(W, a, x0, g_mlp, g_next)
        assert (x_o.double() - x2).abs().max().item() <= 2e-5 * x2.abs().max().item()
        if has_qkv:
            assert (q_o.double() - q).abs().max().item() <= 2e-5 * max(1.0, q.abs().max().item())


def test_seg_mlp_rejects_what_it_was_not_built_for():
    L = hb.lib()
    s = hb.SegArgs()
    assert L.dia_seg_mlp(C.byref(s), None) == -1
    assert L.dia_seg_slots(1) == 29 and L.dia_seg_slots(0) == 26
    assert L.dia_seg_supported(512, 1024, 1024, 1536) == 0


def test_seg_engine_full_size_vs_oracle():
    """the opt-in engine path (DeviceWeights(seg="on") + knob seg=1): Dia-1.6B shapes, batch 1, four teacher-forced steps against
    the lean oracle (logits <= 1e-3, samples identical) — the launches co, wi, wo and the next layer's qkv run as one
    persistent launch per layer (93 launches per step instead of 146)."""
    import os, sys
    from dia_hip import config as Cf
    from dia_hip.engine import DecodeSession, DeviceWeights
    from dia_hip.tokens import effective_text, encode_text
    from dia_hip.weights import synthetic_state_dict
    from oracle import dia_oracle as O
    L = hb.lib()
    if not L.dia_seg_supported(D, F, KA, NQ):
        pytest.skip("no persistent segment on this device (needs 256 CUs)")
    cfg = Cf.dia_1_6b_config()
    dev = torch.device("cuda:0")
    sd_gpu = synthetic_state_dict(cfg, seed=1234, std=0.02, device=dev)
    w = DeviceWeights(cfg, sd_gpu, dev, seg="on")
    assert len(w.seg_layers) == cfg.model.decoder.n_layer
    sd = {k: v.cpu() for k, v in sd_gpu.items()}
    del sd_gpu
    text = "[S1] Dia is an open weights text to dialogue model. [S2] You get full control over scripts and voices."
    steps, mt = 4, 5
    dm = O.Dims.of(cfg)
    nz = O.exp_noise(42, mt - 1, dm.C, dm.tgt_vocab)
    torch.set_num_threads(max(1, min(16, len(os.sched_getaffinity(0)))))
    r = O.generate(sd, cfg, text, max_tokens=mt, seed=None, noise=nz, mirror=False, max_steps=steps)
    hb.set_tuning("seg", 1)
    try:
        s = DecodeSession(w, [encode_text(effective_text(text), cfg)], kv_dtype="f32", max_tokens=mt, noise=nz[None, : mt - 1],
                          teacher_tokens=[r.tokens])
        assert s.seg and s.launches_per_step() == cfg.model.decoder.n_layer * 5 + 3
        s.prefill()
        worst = 0.0
        for i in range(steps):
            s.decode(1, use_graph=False)
            worst = max(worst, float(np.abs(s.logits_host()[0] - r.logits[i]).max()))
        res = s.results(); s.close()
    finally:
        hb.set_tuning("seg", -1)
    print(f"persistent segments, Dia-1.6B teacher-forced: {steps} steps, logits max-abs err {worst:.3e}")
    assert worst <= 1e-3
    for i, p in enumerate(r.preds):
        assert np.array_equal(res[0].preds[1 + i], p), i
