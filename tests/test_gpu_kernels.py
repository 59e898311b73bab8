"""Kernel-level parity on a real MI355X, through the C ABI, against float64 torch-CPU restatements of
each op (and the oracle's sampler).  Tolerances: 2e-5 relative to the output scale for the fp32
paths; token ids / integer outputs bit-exact."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dia_hip import binding as hb
from dia_hip import layout as lay


def dev():
    assert torch.cuda.is_available(), "GPU tests need the MI355X box"
    return torch.device("cuda:0")


def bf16r(t):
    return t.bfloat16().float()


def run_gemm(X, Wt, kt, ns, epi, *, akt=None, ssq_in=None, inv_d=0.0, eps=0.0, out=None, ldo=0, gnext=None, P=None,
             p_kt=0, ssq_out=None, ssq_ld=0, kv=None, cos=None, sin=None, nw=0, sk=None, reps=1, kv_vblocked=0):
    L = hb.lib()
    M = X.shape[0]
    A = lay.pack_planes(X, ktiles=akt)
    g = hb.GemmArgs()
    g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], M
    g.W, g.KT, g.nstrips, g.epi, g.nw = hb.ptr(Wt), kt, ns, epi, nw
    if ssq_in is not None:
        g.ssq_in, g.ssq_in_n, g.inv_d, g.eps = hb.ptr(ssq_in), ssq_in.shape[0], inv_d, eps
    g.ssq_ld = ssq_ld
    g.out, g.ldo, g.gnext = hb.ptr(out), ldo, hb.ptr(gnext)
    if P is not None:
        g.P, g.p_plane_stride, g.p_ktiles = hb.ptr(P), P[0].numel(), p_kt
    g.ssq_out = hb.ptr(ssq_out)
    if kv is not None:
        g.kc, g.vc, g.kv_dtype, g.kv_heads, g.kv_cap, g.kv_batch_index = kv
        g.cos_t, g.sin_t = hb.ptr(cos), hb.ptr(sin)
        g.kv_vblocked = kv_vblocked
    if sk is not None:
        scr = torch.zeros(ns * sk * 256, device=X.device)
        tk = torch.zeros(ns, dtype=torch.int32, device=X.device)
        g.sk_scratch, g.sk_tickets, g.sk = hb.ptr(scr), hb.ptr(tk), sk
    for _ in range(reps):
        hb.check(L.dia_gemm(C.byref(g), None), "dia_gemm")
    torch.cuda.synchronize()
    if sk is not None:
        assert (tk == 0).all()                  # the last arriver re-arms the tickets
    return A


def strip_ssq(x, mpad):
    M, D = x.shape
    s = torch.zeros(D // 16, mpad, dtype=torch.float32, device=x.device)
    s[:, :M] = (x.double() ** 2).reshape(M, D // 16, 16).sum(-1).T.float()
    return s


@pytest.mark.parametrize("M,K,N,nw", [(2, 2048, 2048, 0), (2, 2048, 3072, 16), (16, 512, 1024, 8), (20, 96, 80, 0),
                                      (75, 256, 300, 4), (130, 1024, 64, 0), (2, 8192, 256, 16), (5, 2048, 9252, 4)])
def test_gemm_scale_store(M, K, N, nw):
    d = dev()
    torch.manual_seed(M * 7 + K)
    x = torch.randn(M, K, device=d) * 2.0
    gw = bf16r(1.0 + 0.1 * torch.randn(K, device=d))
    W = bf16r(torch.randn(K, N, device=d) * 0.05)
    Wt, kt, ns = lay.tile_weight(W)
    mpad = (M + 15) // 16 * 16
    has_norm = K % 16 == 0
    ssq = strip_ssq(x, mpad) if has_norm else None
    out = torch.full((M, ns * 16), float("nan"), device=d)
    run_gemm(x * gw, Wt, kt, ns, hb.EPI_SCALE_STORE, ssq_in=ssq, inv_d=1.0 / K, eps=1e-5, out=out, ldo=ns * 16,
             ssq_ld=mpad, nw=nw)
    xd = x.double()
    inv = torch.rsqrt((xd ** 2).mean(-1, keepdim=True) + 1e-5) if has_norm else 1.0
    ref = ((xd * gw.double()) @ W.double()) * inv
    err = (out[:, :N].double() - ref).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err
    if ns * 16 > N:
        assert (out[:, N:] == 0).all()


@pytest.mark.parametrize("M,K,D", [(2, 2048, 2048), (16, 8192, 512), (37, 256, 96)])
def test_gemm_resid_emit(M, K, D):
    d = dev()
    torch.manual_seed(K + D)
    a = torch.randn(M, K, device=d)
    W = bf16r(torch.randn(K, D, device=d) * 0.03)
    x0 = torch.randn(M, D, device=d)
    gn = bf16r(1.0 + 0.1 * torch.randn(D, device=d))
    Wt, kt, ns = lay.tile_weight(W)
    mpad = (M + 15) // 16 * 16
    x = x0.clone()
    P = torch.zeros(3, mpad // 16, (D + 31) // 32, 64, 8, dtype=torch.bfloat16, device=d)
    ssq = torch.zeros(ns, mpad, device=d)
    run_gemm(a, Wt, kt, ns, hb.EPI_RESID_EMIT, out=x, ldo=D, gnext=gn, P=P, p_kt=P.shape[2], ssq_out=ssq, ssq_ld=mpad)
    ref = x0.double() + a.double() @ W.double()
    assert (x.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    assert torch.equal(lay.unpack_planes(P, M, D), x * gn)            # planes carry x*g exactly (3x bf16 == fp32)
    want = (x.double() ** 2).reshape(M, D // 16, 16).sum(-1).T
    assert (ssq[:, :M].double() - want).abs().max().item() <= 1e-5 * want.max().item()


@pytest.mark.parametrize("M,K,D,sk", [(2, 8192, 2048, 4), (4, 8192, 512, 4), (16, 8192, 2048, 4), (9, 4096, 256, 2)])
def test_gemm_split_k_resid(M, K, D, sk):
    """cross-workgroup split-K (wo): partial tiles combined by the last arriver, fixed order"""
    d = dev()
    torch.manual_seed(K + D + M)
    a = torch.randn(M, K, device=d)
    W = bf16r(torch.randn(K, D, device=d) * 0.03)
    x0 = torch.randn(M, D, device=d)
    gn = bf16r(1.0 + 0.1 * torch.randn(D, device=d))
    Wt, kt, ns = lay.tile_weight(W)
    outs = []
    for _ in range(2):
        x = x0.clone()
        P = torch.zeros(3, 1, D // 32, 64, 8, dtype=torch.bfloat16, device=d)
        ssq = torch.zeros(ns, 16, device=d)
        run_gemm(a, Wt, kt, ns, hb.EPI_RESID_EMIT, out=x, ldo=D, gnext=gn, P=P, p_kt=D // 32, ssq_out=ssq, ssq_ld=16, sk=sk)
        outs.append((x, P.clone(), ssq))
    ref = x0.double() + a.double() @ W.double()
    assert (outs[0][0].double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])      # bit-reproducible
    assert torch.equal(lay.unpack_planes(outs[0][1], M, D), outs[0][0] * gn)


@pytest.mark.parametrize("M", [17, 23, 32, 40, 64, 100, 128])
def test_gemm_paired_mtiles(M):
    """17..64 rows (batch 9-32), default path: the one-m-tile kernel launched over all m-tiles (gridDim.z = 2..4).
    Every epilogue, persistent and one-strip forms, split-K 4; rows 0..15 must equal the 16-row launch bit for bit
    (same kernel, same summation order), rows 16.. the 16-row launch on those rows."""
    d = dev()
    torch.manual_seed(200 + M)
    L = hb.lib()

    def args(A, M_, Wt, kt, ns, epi):
        g = hb.GemmArgs()
        g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], M_
        g.W, g.KT, g.nstrips, g.epi = hb.ptr(Wt), kt, ns, epi
        return g

    # SCALE_STORE: K = 2048, N = 3072 (192 strips -> two strips per workgroup) and N = 592 (37 strips)
    K = 2048
    x = torch.randn(M, K, device=d) * 2.0
    gw = bf16r(1.0 + 0.1 * torch.randn(K, device=d))
    xd = x.double()
    inv = torch.rsqrt((xd ** 2).mean(-1, keepdim=True) + 1e-5)
    MP = (M + 15) // 16 * 16
    MT_ = MP // 16
    ss = strip_ssq(x, MP)
    for N in (3072, 592):
        W = bf16r(torch.randn(K, N, device=d) * 0.05)
        Wt, kt, ns = lay.tile_weight(W)
        outs = []
        for rows in (slice(0, M), slice(0, 16), slice(MP - 16, M)):
            xs = x[rows]
            A = lay.pack_planes(xs * gw)
            out = torch.full((xs.shape[0], N), float("nan"), device=d)
            sr = strip_ssq(xs, MP)
            g = args(A, xs.shape[0], Wt, kt, ns, hb.EPI_SCALE_STORE)
            g.ssq_in, g.ssq_in_n, g.inv_d, g.eps, g.ssq_ld = hb.ptr(sr), K // 16, 1.0 / K, 1e-5, MP
            g.out, g.ldo = hb.ptr(out), N
            hb.check(L.dia_gemm(C.byref(g), None), "dia_gemm")
            torch.cuda.synchronize()
            outs.append(out)
        ref = ((xd * gw.double()) @ W.double()) * inv
        assert (outs[0].double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
        assert torch.equal(outs[0][:16], outs[1])
        if M - (MP - 16) > 4:           # (up to 4 rows take the GEMV kernel: another K split, another summation order)
            assert torch.equal(outs[0][MP - 16:], outs[2])
    # SWIGLU_EMIT: F = 8192 (1024 strips, eight per workgroup) and F = 1024
    A = lay.pack_planes(x * gw)
    h = (xd * inv) * gw.double()
    for F in (8192, 1024):
        wi = bf16r(torch.randn(K, 2, F, device=d) * 0.05)
        Wt, kt, ns = lay.tile_weight(lay.interleave_gate_up(wi))
        P = torch.zeros(3, MT_, F // 32, 64, 8, dtype=torch.bfloat16, device=d)
        g = args(A, M, Wt, kt, ns, hb.EPI_SWIGLU_EMIT)
        g.ssq_in, g.ssq_in_n, g.inv_d, g.eps, g.ssq_ld = hb.ptr(ss), K // 16, 1.0 / K, 1e-5, MP
        g.P, g.p_plane_stride, g.p_ktiles = hb.ptr(P), P[0].numel(), F // 32
        hb.check(L.dia_gemm(C.byref(g), None), "dia_gemm")
        torch.cuda.synchronize()
        f = torch.einsum("mk,kgf->mgf", h, wi.double())
        ref = torch.nn.functional.silu(f[:, 0]) * f[:, 1]
        assert (lay.unpack_planes(P, M, F).double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # RESID_EMIT: (K, D, sk) = o-like, wo-like with split-K 4 over both m-tiles, a small ragged one
    for K2, D, sk in ((2048, 2048, 0), (8192, 2048, 4), (512, 272, 0), (4096, 256, 2)):
        a = torch.randn(M, K2, device=d)
        W2 = bf16r(torch.randn(K2, D, device=d) * 0.03)
        x0 = torch.randn(M, D, device=d)
        gn = bf16r(1.0 + 0.1 * torch.randn(D, device=d))
        Wt, kt, ns = lay.tile_weight(W2)
        A2 = lay.pack_planes(a)
        pkt = (D + 31) // 32
        scr = torch.zeros(max(1, MT_ * ns * max(sk, 1) * 256), device=d); tk = torch.zeros(MT_ * ns, dtype=torch.int32, device=d)
        res = []
        for _ in range(2):
            xr = x0.clone()
            P = torch.zeros(3, MT_, pkt, 64, 8, dtype=torch.bfloat16, device=d)
            ssq = torch.zeros(ns, MP, device=d)
            g = args(A2, M, Wt, kt, ns, hb.EPI_RESID_EMIT)
            g.ssq_ld, g.out, g.ldo, g.gnext = MP, hb.ptr(xr), D, hb.ptr(gn)
            g.P, g.p_plane_stride, g.p_ktiles, g.ssq_out = hb.ptr(P), P[0].numel(), pkt, hb.ptr(ssq)
            if sk:
                g.sk, g.sk_scratch, g.sk_tickets, g.sk_scratch_floats = sk, hb.ptr(scr), hb.ptr(tk), scr.numel()
            hb.check(L.dia_gemm(C.byref(g), None), "dia_gemm")
            torch.cuda.synchronize()
            assert (tk == 0).all()
            if sk:
                assert scr[(MT_ - 1) * ns * sk * 256:].abs().sum().item() > 0      # the last m-tile used its own slabs
            res.append((xr, P, ssq))
        ref = x0.double() + a.double() @ W2.double()
        assert (res[0][0].double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
        assert torch.equal(lay.unpack_planes(res[0][1], M, pkt * 32)[:, :D], res[0][0] * gn)
        want = (res[0][0].double() ** 2).reshape(M, D // 16, 16).sum(-1).T
        assert (res[0][2][:, :M].double() - want).abs().max().item() <= 1e-5 * want.max().item()
    # split-K without a stated capacity for both m-tiles is refused (as before), not run out of bounds
    g = args(A2, M, Wt, kt, ns, hb.EPI_RESID_EMIT)
    g.ssq_ld, g.out, g.ldo, g.gnext = MP, hb.ptr(xr), D, hb.ptr(gn)
    g.P, g.p_plane_stride, g.p_ktiles, g.ssq_out = hb.ptr(P), P[0].numel(), pkt, hb.ptr(ssq)
    g.sk, g.sk_scratch, g.sk_tickets = 2, hb.ptr(scr), hb.ptr(tk)
    assert L.dia_gemm(C.byref(g), None) != 0


@pytest.mark.parametrize("M,K,F", [(2, 2048, 8192), (16, 256, 512), (33, 512, 1024)])
def test_gemm_swiglu_emit(M, K, F):
    d = dev()
    torch.manual_seed(F)
    x = torch.randn(M, K, device=d)
    gw = bf16r(1.0 + 0.1 * torch.randn(K, device=d))
    wi = bf16r(torch.randn(K, 2, F, device=d) * 0.05)
    Wt, kt, ns = lay.tile_weight(lay.interleave_gate_up(wi))
    mpad = (M + 15) // 16 * 16
    P = torch.zeros(3, mpad // 16, F // 32, 64, 8, dtype=torch.bfloat16, device=d)
    run_gemm(x * gw, Wt, kt, ns, hb.EPI_SWIGLU_EMIT, ssq_in=strip_ssq(x, mpad), inv_d=1.0 / K, eps=1e-5, P=P, p_kt=F // 32,
             ssq_ld=mpad)
    xd = x.double()
    h = (xd * torch.rsqrt((xd ** 2).mean(-1, keepdim=True) + 1e-5)) * gw.double()
    f = torch.einsum("mk,kgf->mgf", h, wi.double())
    ref = torch.nn.functional.silu(f[:, 0]) * f[:, 1]
    got = lay.unpack_planes(P, M, F).double()
    assert (got - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("kvd", ["f32", "bf16"])
def test_gemm_crosskv(kvd):
    d = dev()
    torch.manual_seed(3)
    Lq, E, H, cap = 37, 256, 4, 48
    x = torch.randn(Lq, E, device=d)
    gw = bf16r(1.0 + 0.1 * torch.randn(E, device=d))
    wk = bf16r(torch.randn(E, H, 128, device=d) * 0.05)
    wv = bf16r(torch.randn(E, H, 128, device=d) * 0.05)
    perm = lay.rope_pair_perm(128).to(d)
    Wt, kt, ns = lay.tile_weight(torch.cat([wk[:, :, perm].reshape(E, -1), wv.reshape(E, -1)], dim=1))
    cos, sin = [t.to(d) for t in lay.rope_tables(64, 128, 1, 10000)]
    kdt = torch.float32 if kvd == "f32" else torch.bfloat16
    kc = torch.zeros(2, H, cap, 128, dtype=kdt, device=d)
    vc = torch.zeros(2, H, cap, 128, dtype=kdt, device=d)
    mpad = (Lq + 15) // 16 * 16
    code = hb.KV_F32 if kvd == "f32" else hb.KV_BF16
    blocked = kvd == "bf16"
    if blocked:
        cap = 64
        kc = torch.zeros(2, H, cap, 128, dtype=kdt, device=d)
        vc = torch.zeros(2, H, cap, 128, dtype=kdt, device=d)
    run_gemm(x * gw, Wt, kt, ns, hb.EPI_CROSSKV, ssq_in=strip_ssq(x, mpad), inv_d=1.0 / E, eps=1e-5, ssq_ld=mpad,
             kv=(hb.ptr(kc), hb.ptr(vc), code, H, cap, 1), cos=cos, sin=sin, kv_vblocked=int(blocked))
    if blocked:
        vc = lay.v_from_blocked(vc.reshape(2, H, cap // 32, 128, 32))
    xd = x.double()
    h = (xd * torch.rsqrt((xd ** 2).mean(-1, keepdim=True) + 1e-5)) * gw.double()
    k = torch.einsum("me,ehd->mhd", h, wk.double())
    v = torch.einsum("me,ehd->mhd", h, wv.double())
    c, s = cos[:Lq].double()[:, None, :], sin[:Lq].double()[:, None, :]
    kr = torch.cat([k[..., :64] * c - k[..., 64:] * s, k[..., :64] * s + k[..., 64:] * c], dim=-1)
    tol = 2e-5 if kvd == "f32" else 1e-2
    assert (kc[1, :, :Lq].double().transpose(0, 1) - kr).abs().max().item() <= tol * kr.abs().max().item()
    assert (vc[1, :, :Lq].double().transpose(0, 1) - v).abs().max().item() <= tol * v.abs().max().item()
    assert (kc[0] == 0).all() and (kc[1, :, Lq:] == 0).all()


def attn_ref(q, K, V):
    """q [G,128] f64, K/V [t,128] f64"""
    s = (q @ K.T) / math.sqrt(128.0)
    p = torch.softmax(s, dim=-1)
    return p @ V


@pytest.mark.parametrize("kvd", ["f32", "bf16", "bf16x2"])
@pytest.mark.parametrize("B,cur,nz", [(1, 1, 0), (1, 37, 0), (1, 128, 0), (1, 129, 0), (2, 300, 0), (1, 1025, 0), (1, 1280, 0),
                                      (1, 1025, 1), (2, 300, 2), (1, 1280, 3), (1, 640, 2)])
def test_attn_self(kvd, B, cur, nz, tuning):
    """nz > 0 forces the key-split factor: few workgroups -> several rounds per wave (prefetch path, the
    new slot in a later round), many -> one granule per wave."""
    if nz:
        tuning("attn_nz", nz)
    d = dev()
    torch.manual_seed(cur)
    R, QH, KVH, T = 2 * B, 16, 4, 1280
    nq = (QH + 2 * KVH) * 128
    qkv = torch.randn(R, nq, device=d)
    kdt = torch.float32 if kvd == "f32" else torch.bfloat16
    two = kvd == "bf16x2"                         # every value as hi + lo bf16 in two planes (16 significand bits)
    kf, vf = torch.randn(R, KVH, T, 128, device=d), torch.randn(R, KVH, T, 128, device=d)
    blocked = kvd != "f32"                        # bf16 caches: V blocked, MFMA kernel
    if two:
        khi, vhi = kf.bfloat16(), vf.bfloat16()
        klo, vlo = (kf - khi.float()).bfloat16(), (vf - vhi.float()).bfloat16()
        kc = torch.stack([khi, klo]).contiguous()
        vc = torch.stack([lay.v_to_blocked(vhi), lay.v_to_blocked(vlo)]).contiguous()
    else:
        kc, vc = kf.to(kdt), vf.to(kdt)
        if blocked:
            vc = lay.v_to_blocked(vc)
    kc0, vc0 = kc.clone(), vc.clone()
    cos, sin = [t.to(d) for t in lay.rope_tables(T + 1, 128, 1, 10000)]
    curs = torch.full((B,), cur, dtype=torch.int32, device=d)
    mt = (R + 15) // 16
    P = torch.zeros(3, mt, QH * 128 // 32, 64, 8, dtype=torch.bfloat16, device=d)
    a = hb.AttnArgs()
    a.mode, a.kv_dtype, a.n_kv_heads, a.group, a.n_rows, a.kv_cap = hb.ATTN_SELF, {"f32": 0, "bf16": 1, "bf16x2": 2}[kvd], KVH, 4, R, T
    a.q, a.ldq, a.q_off, a.k_off, a.v_off = hb.ptr(qkv), nq, 0, QH * 128, (QH + KVH) * 128
    a.kc, a.vc, a.cur = hb.ptr(kc), hb.ptr(vc), hb.ptr(curs)
    a.cos_t, a.sin_t = hb.ptr(cos), hb.ptr(sin)
    a.P, a.p_plane_stride, a.p_ktiles = hb.ptr(P), P[0].numel(), P.shape[2]
    scr = torch.zeros(hb.lib().dia_attn_scratch_floats(R, KVH, T), device=d)
    tk = torch.zeros(R * KVH, dtype=torch.int32, device=d)
    a.scratch, a.tickets = hb.ptr(scr), hb.ptr(tk)
    a.v_blocked = int(blocked)
    a.kv_plane_stride = kc[0].numel() if two else 0
    for _ in range(2):                                  # second launch: tickets were re-zeroed by the kernel
        kc.copy_(kc0); vc.copy_(vc0)
        hb.check(hb.lib().dia_attn(C.byref(a), None), "dia_attn")
    torch.cuda.synchronize()
    assert (tk == 0).all()
    if two:        # the value a two-plane cache holds is hi + lo
        kc, kc0 = kc[0].float() + kc[1].float(), kc0[0].float() + kc0[1].float()
        vc = lay.v_from_blocked(vc[0]).float() + lay.v_from_blocked(vc[1]).float()
        vc0 = lay.v_from_blocked(vc0[0]).float() + lay.v_from_blocked(vc0[1]).float()
    elif blocked:
        vc, vc0 = lay.v_from_blocked(vc), lay.v_from_blocked(vc0)
    out = lay.unpack_planes(P, R, QH * 128).double().reshape(R, QH, 128)

    def rope(x, pos):
        c, s = cos[pos].double(), sin[pos].double()
        return torch.cat([x[..., :64] * c - x[..., 64:] * s, x[..., :64] * s + x[..., 64:] * c], dim=-1)

    q = rope(qkv[:, : QH * 128].double().reshape(R, QH, 128), cur)
    knew = rope(qkv[:, QH * 128: (QH + KVH) * 128].double().reshape(R, KVH, 128), cur)
    vnew = qkv[:, (QH + KVH) * 128:].double().reshape(R, KVH, 128)
    if kvd == "bf16":
        knew, vnew = knew.float().bfloat16().double(), vnew.float().bfloat16().double()
    slot = cur - 1
    # cache append (state.py:99-103): slot written, everything else untouched
    tol_new = 6e-5 if two else 1e-6            # hi + lo keeps 16 significand bits of values up to ~8
    assert (kc[:, :, slot].double() - knew).abs().max().item() <= (1e-6 if kvd == "f32" else 0.0) + tol_new
    assert (vc[:, :, slot].double() - vnew).abs().max().item() <= tol_new
    keep = torch.ones(T, dtype=torch.bool, device=d); keep[slot] = False
    assert torch.equal(kc[:, :, keep], kc0[:, :, keep]) and torch.equal(vc[:, :, keep], vc0[:, :, keep])
    worst = 0.0
    for r in range(R):
        for h in range(KVH):
            K = kc[r, h, :cur].double(); V = vc[r, h, :cur].double()
            ref = attn_ref(q[r, 4 * h: 4 * h + 4], K, V)
            worst = max(worst, (out[r, 4 * h: 4 * h + 4] - ref).abs().max().item())
    assert worst <= 2e-5, worst


@pytest.mark.parametrize("kvd", ["f32", "bf16"])
def test_attn_cross_and_uncond_zero(kvd):
    d = dev()
    torch.manual_seed(11)
    B, H, S = 3, 16, 224
    lens = [75, 224, 0]
    R = 2 * B
    qc = torch.randn(R, H * 128, device=d)
    kdt = torch.float32 if kvd == "f32" else torch.bfloat16
    kc = torch.randn(B, H, S, 128, device=d).to(kdt)
    vc = torch.randn(B, H, S, 128, device=d).to(kdt)
    cos, sin = [t.to(d) for t in lay.rope_tables(512, 128, 1, 10000)]
    cur = torch.tensor([5, 17, 9], dtype=torch.int32, device=d)
    ln = torch.tensor(lens, dtype=torch.int32, device=d)
    P = torch.full((3, 1, H * 128 // 32, 64, 8), 7.0, dtype=torch.bfloat16, device=d)
    a = hb.AttnArgs()
    a.mode, a.kv_dtype, a.n_kv_heads, a.group, a.n_rows, a.kv_cap = hb.ATTN_CROSS, (0 if kvd == "f32" else 1), H, 1, B, S
    a.q, a.ldq = hb.ptr(qc), H * 128
    a.kc, a.vc, a.cur, a.len = hb.ptr(kc), hb.ptr(vc), hb.ptr(cur), hb.ptr(ln)
    a.cos_t, a.sin_t = hb.ptr(cos), hb.ptr(sin)
    a.P, a.p_plane_stride, a.p_ktiles = hb.ptr(P), P[0].numel(), P.shape[2]
    scr = torch.zeros(hb.lib().dia_attn_scratch_floats(B, H, S), device=d)
    tk = torch.zeros(B * H, dtype=torch.int32, device=d)
    a.scratch, a.tickets = hb.ptr(scr), hb.ptr(tk)
    vc_row = vc
    if kvd == "bf16":
        vc = lay.v_to_blocked(vc); a.vc = hb.ptr(vc); a.v_blocked = 1
    hb.check(hb.lib().dia_attn(C.byref(a), None), "dia_attn")
    torch.cuda.synchronize()
    vc = vc_row
    out = lay.unpack_planes(P, R, H * 128).double().reshape(R, H, 128)
    for b in range(B):
        assert (out[2 * b] == 0).all()                      # uncond row: fully masked -> exactly 0
        c, s = cos[int(cur[b])].double(), sin[int(cur[b])].double()
        q = qc[2 * b + 1].double().reshape(H, 128)
        q = torch.cat([q[:, :64] * c - q[:, 64:] * s, q[:, :64] * s + q[:, 64:] * c], dim=-1)
        if lens[b] == 0:
            assert (out[2 * b + 1] == 0).all()
            continue
        for h in range(H):
            ref = attn_ref(q[h: h + 1], kc[b, h, : lens[b]].double(), vc[b, h, : lens[b]].double())
            assert (out[2 * b + 1, h] - ref[0]).abs().max().item() <= 2e-5


def test_attn_encoder_mode_and_kv_prep():
    d = dev()
    torch.manual_seed(5)
    Lq, H = 45, 4
    nq = 3 * H * 128
    qkv = torch.randn(Lq, nq, device=d)
    cos, sin = [t.to(d) for t in lay.rope_tables(64, 128, 1, 10000)]
    cap = 48
    kc = torch.zeros(H, cap, 128, device=d); vc = torch.zeros(H, cap, 128, device=d)
    Lb = hb.lib()
    hb.check(Lb.dia_enc_kv_prep(hb.ptr(qkv), nq, H * 128, 2 * H * 128, H, Lq, cap, hb.ptr(cos), hb.ptr(sin), hb.ptr(kc), hb.ptr(vc), None), "prep")
    P = torch.zeros(3, 3, H * 128 // 32, 64, 8, dtype=torch.bfloat16, device=d)
    a = hb.AttnArgs()
    a.mode, a.kv_dtype, a.n_kv_heads, a.group, a.n_rows, a.kv_cap = hb.ATTN_ENC, 0, H, 1, Lq, cap
    a.q, a.ldq, a.enc_len = hb.ptr(qkv), nq, Lq
    a.kc, a.vc = hb.ptr(kc), hb.ptr(vc)
    a.cos_t, a.sin_t = hb.ptr(cos), hb.ptr(sin)
    a.P, a.p_plane_stride, a.p_ktiles = hb.ptr(P), P[0].numel(), P.shape[2]
    hb.check(Lb.dia_attn(C.byref(a), None), "dia_attn")
    torch.cuda.synchronize()
    out = lay.unpack_planes(P, Lq, H * 128).double().reshape(Lq, H, 128)
    c, s = cos[:Lq].double()[:, None], sin[:Lq].double()[:, None]

    def rope(x):
        return torch.cat([x[..., :64] * c - x[..., 64:] * s, x[..., :64] * s + x[..., 64:] * c], dim=-1)

    q = rope(qkv[:, : H * 128].double().reshape(Lq, H, 128))
    k = rope(qkv[:, H * 128: 2 * H * 128].double().reshape(Lq, H, 128))
    v = qkv[:, 2 * H * 128:].double().reshape(Lq, H, 128)
    assert (kc[:, :Lq].double().transpose(0, 1) - k).abs().max().item() <= 1e-6
    for h in range(H):
        ref = attn_ref(q[:, h], k[:, h], v[:, h])
        assert (out[:, h] - ref).abs().max().item() <= 2e-5


def test_enc_attn_packed_batch():
    """dia_enc_attn: bidirectional attention of several utterances packed into one row range (prefill),
    MFMA with three planes per operand, against float64 per utterance; padding rows stay untouched."""
    d = dev()
    torch.manual_seed(9)
    H, lens = 4, [45, 32, 0, 70, 1]
    offs, tot = [], 0
    for Lb in lens:
        offs.append(tot); tot += (Lb + 31) // 32 * 32
    nq = 3 * H * 128
    qkv = torch.randn(tot, nq, device=d)
    cos, sin = [t.to(d) for t in lay.rope_tables(128, 128, 1, 10000)]
    rb = np.full((tot,), -1, dtype=np.int32)
    for b, Lb in enumerate(lens):
        rb[offs[b]: offs[b] + Lb] = b
    row_b = torch.from_numpy(rb).to(d)
    seg_off = torch.tensor(offs, dtype=torch.int32, device=d); seg_len = torch.tensor(lens, dtype=torch.int32, device=d)
    kp = torch.zeros(3, H, tot, 128, dtype=torch.bfloat16, device=d); vp = torch.zeros_like(kp)
    sentinel = 7.0
    P = lay.pack_planes(torch.full((tot, H * 128), sentinel, device=d))
    a = hb.EncAttnArgs()
    a.qkv, a.ldq, a.q_off, a.k_off, a.v_off, a.heads, a.rows = hb.ptr(qkv), nq, 0, H * 128, 2 * H * 128, H, tot
    a.row_b, a.seg_off, a.seg_len, a.cos_t, a.sin_t = hb.ptr(row_b), hb.ptr(seg_off), hb.ptr(seg_len), hb.ptr(cos), hb.ptr(sin)
    a.kp, a.vp, a.P, a.p_plane_stride, a.p_ktiles = hb.ptr(kp), hb.ptr(vp), hb.ptr(P), P[0].numel(), P.shape[2]
    hb.check(hb.lib().dia_enc_attn(C.byref(a), None), "dia_enc_attn")
    torch.cuda.synchronize()
    out = lay.unpack_planes(P, tot, H * 128).double().reshape(tot, H, 128)
    worst = 0.0
    for b, Lb in enumerate(lens):
        if Lb == 0:
            continue
        o = offs[b]
        c, s_ = cos[:Lb].double()[:, None], sin[:Lb].double()[:, None]
        rope = lambda x: torch.cat([x[..., :64] * c - x[..., 64:] * s_, x[..., :64] * s_ + x[..., 64:] * c], dim=-1)
        q = rope(qkv[o: o + Lb, : H * 128].double().reshape(Lb, H, 128))
        k = rope(qkv[o: o + Lb, H * 128: 2 * H * 128].double().reshape(Lb, H, 128))
        v = qkv[o: o + Lb, 2 * H * 128:].double().reshape(Lb, H, 128)
        for h in range(H):
            worst = max(worst, (out[o: o + Lb, h] - attn_ref(q[:, h], k[:, h], v[:, h])).abs().max().item())
    assert worst <= 2e-5, worst
    pad = torch.from_numpy(rb < 0).to(d)
    assert (out[pad] == sentinel).all()                       # padding rows are not written
    # the three planes reproduce fp32 K exactly (k rows of utterance 0, head 1)
    kk = (kp[0].float() + kp[1].float() + kp[2].float())[1, : lens[0]]
    c, s_ = cos[: lens[0]], sin[: lens[0]]
    x = qkv[: lens[0], H * 128 + 128: H * 128 + 256]
    assert (kk.double() - torch.cat([x[:, :64] * c - x[:, 64:] * s_, x[:, :64] * s_ + x[:, 64:] * c], dim=-1).double()).abs().max().item() <= 1e-6


def _sampler_session(B, T, C_, V, D, logits_rows, noise, *, temperature, top_p, top_k, cfg_scale=0.0, cur=1,
                     teacher=0, ignore_eos=0, max_tokens=None, tokens=None, fsm=None, delay=None):
    d = dev()
    ld = (C_ * V + 15) // 16 * 16
    lg = torch.zeros((2 * B + 15) // 16 * 16, ld, device=d)
    lg[: 2 * B, : C_ * V] = logits_rows.reshape(2 * B, C_ * V)
    tok = torch.full((B, T, C_), -1, dtype=torch.int32, device=d) if tokens is None else tokens.clone()
    pred = torch.full((B, T, C_), -1, dtype=torch.int32, device=d)
    curs = torch.full((B,), cur, dtype=torch.int32, device=d)
    if fsm is None:
        fsm = torch.zeros(B, 8, dtype=torch.int32, device=d); fsm[:, 1] = -1; fsm[:, 2] = 0
    emb = bf16r(torch.randn(C_, V, D, device=d) * 0.1)
    gw = bf16r(1 + 0.1 * torch.randn(D, device=d))
    x = torch.zeros(16, D, device=d)
    P = torch.zeros(3, 1, D // 32, 64, 8, dtype=torch.bfloat16, device=d)
    ssq = torch.zeros(D // 16, 16, device=d)
    dl = torch.tensor(delay or [0, 8, 9, 10, 11, 12, 13, 14, 15][:C_], dtype=torch.int32, device=d)
    s = hb.SampleArgs()
    s.logits, s.ld_logits, s.B, s.T, s.C, s.V = hb.ptr(lg), ld, B, T, C_, V
    s.max_tokens = T if max_tokens is None else max_tokens
    s.cfg_scale, s.temperature, s.top_p, s.top_k = cfg_scale, temperature, top_p, top_k
    s.eos, s.pad, s.bos, s.max_delay, s.ignore_eos, s.teacher = 1024, 1025, 1026, 15, ignore_eos, teacher
    s.delay, s.noise, s.noise_steps = hb.ptr(dl), hb.ptr(noise), (0 if noise is None else noise.shape[1])
    s.tokens, s.pred, s.cur, s.fsm = hb.ptr(tok), hb.ptr(pred), hb.ptr(curs), hb.ptr(fsm)
    e = s.embed
    e.D, e.emb, e.g, e.x = D, hb.ptr(emb), hb.ptr(gw), hb.ptr(x)
    e.P, e.p_plane_stride, e.p_ktiles, e.ssq_ld, e.ssq = hb.ptr(P), P[0].numel(), D // 32, 16, hb.ptr(ssq)
    keep = (lg, tok, pred, curs, fsm, emb, gw, x, P, ssq, dl, noise)
    return s, keep


def test_sampler_matches_reference_cases(golden):
    """the reference's _sample_next_token outputs (tests/golden/ref_sampler.npz) on the device sampler"""
    g = golden("ref_sampler.npz")
    d = dev()
    for i in range(int(g["n"])):
        T_, tp, tk = g[f"params_{i}"]
        lgc = torch.from_numpy(g[f"logits_{i}"]).to(d)
        lgc = torch.where(torch.isinf(lgc), torch.zeros_like(lgc), lgc)       # masked entries are re-masked by the kernel
        rows = torch.stack([lgc, lgc])                                         # uncond == cond, cfg_scale 0 -> guided == cond
        steps = 40
        nz = torch.ones(1, steps, 9, 1028, device=d)
        nz[0, 0] = torch.from_numpy(g[f"noise_{i}"]).to(d)
        s, keep = _sampler_session(1, 64, 9, 1028, 64, rows, nz, temperature=float(T_), top_p=float(tp),
                                   top_k=(0 if tk < 0 else int(tk)), ignore_eos=1, max_tokens=41)
        hb.check(hb.lib().dia_sample(C.byref(s), None), "dia_sample")
        torch.cuda.synchronize()
        pred = keep[2][0, 1].cpu().numpy()
        assert np.array_equal(pred, g[f"out_{i}"]), (i, pred, g[f"out_{i}"])


@pytest.mark.parametrize("quantized", [False, True])
def test_sampler_randomized_vs_oracle(quantized):
    """Random logits through every route of the device sampler — the lane-maxima candidate cut (k <= 64, <= 64 candidates),
    the bitwise k-th-value search (k > 64, or more candidates than lanes), the one-per-lane path and the 17-register path
    (no top-k), top-p off — against the oracle's sample_next_token (itself pinned on the reference's function).
    `quantized`: logits on a 1/4 grid, so equal values and equal probabilities are everywhere.  Equal probabilities at the
    top-p boundary are the one place where the reference is not a function of its inputs alone (torch.sort(descending=True)
    on CPU is not stable: ties come out in an order of its own); the device breaks them by vocabulary index, so that
    variant is compared with the oracle under a STABLE sort."""
    from unittest import mock
    from oracle import dia_oracle as O
    real_sort = torch.sort
    stable_sort = lambda t, **kw: real_sort(t, stable=True, **kw)
    d = dev()
    g = torch.Generator().manual_seed(11 if quantized else 7)
    B, C_, V = 16, 9, 1028
    cases = [(1.3, 0.95, 35), (1.0, 0.5, 10), (0.7, 1.0, 0), (1.3, 0.9, 0), (2.0, 0.3, 50), (1.0, 0.95, 64), (1.3, 0.8, 100),
             (0.5, 0.95, 1), (1.3, 1.0, 35), (1.7, 0.99, 63), (1.0, 0.6, 5)]
    bad = []
    for T_, tp, tk in cases:
        lg = torch.randn(B, C_, V, generator=g) * float(torch.empty(1).uniform_(0.5, 4.0, generator=g))
        if quantized:
            lg = (lg * 4).round() / 4
        noise = torch.ones(B, 7, C_, V)                                         # max_tokens - 1 steps; step 0 is the one drawn
        noise[:, 0] = torch.empty(B, C_, V).exponential_(1.0, generator=g)
        rows = torch.stack([lg, lg], dim=1).reshape(2 * B, C_, V).to(d)          # uncond == cond, cfg_scale 0
        s, keep = _sampler_session(B, 8, C_, V, 64, rows, noise.to(d), temperature=T_, top_p=tp, top_k=tk, ignore_eos=1, max_tokens=8)
        hb.check(hb.lib().dia_sample(C.byref(s), None), "dia_sample")
        torch.cuda.synchronize()
        pred = keep[2][:, 1].cpu()
        masked = lg.clone()
        masked[:, :, 1025] = -math.inf; masked[:, :, 1026] = -math.inf         # PAD, BOS (model.py:466-472); 1027 stays a candidate
        masked[:, 1:, 1024] = -math.inf                                        # EOS only on channel 0
        with mock.patch.object(torch, "sort", stable_sort if quantized else real_sort):
            want = O.sample_next_token(masked, T_, tp, tk if tk > 0 else None, noise=noise[:, 0])
        if not torch.equal(pred.long(), want.long()):
            bad.append((T_, tp, tk, int((pred.long() != want.long()).sum())))
    assert not bad, bad


def test_sampler_fsm_and_embedding():
    """EOS countdown / masked write / next-step embedding against a direct restatement of
    model.py:771-807 + layers.py:691-696."""
    d = dev()
    torch.manual_seed(0)
    B, T, C_, V, D = 2, 64, 9, 1028, 96
    # utterance 0: natural EOS on channel 0 at step cur=20; utterance 1: inside the BOS window (cur=3)
    rows = torch.randn(2 * B, C_, V, device=d)
    rows[1, 0, 1024] = 50.0; rows[0, 0, 1024] = 50.0
    tok = torch.full((B, T, C_), -1, dtype=torch.int32, device=d)
    tok[1, 3, :] = torch.tensor([-1, 1026, 1026, 1026, 1026, 1026, 1026, 1026, 1026], dtype=torch.int32)
    fsm = torch.zeros(B, 8, dtype=torch.int32, device=d); fsm[:, 1] = -1
    fsm[0, 2] = 0; fsm[1, 2] = 13
    s, keep = _sampler_session(B, T, C_, V, D, rows, None, temperature=0.0, top_p=0.95, top_k=35, cfg_scale=3.0, cur=20,
                               tokens=tok, fsm=fsm)
    keep[3][1] = 3
    hb.check(hb.lib().dia_sample(C.byref(s), None), "dia_sample")
    torch.cuda.synchronize()
    lg, tok2, pred, curs, fsm2, emb, gw, x, P, ssq = keep[:10]
    guided = rows[1::2] + 3.0 * (rows[1::2] - rows[0::2])
    guided[:, :, 1025] = -math.inf; guided[:, :, 1026] = -math.inf; guided[:, 1:, 1024] = -math.inf
    am = guided.argmax(-1).int()
    assert torch.equal(pred[0, 20], am[0]) and torch.equal(pred[1, 3], am[1])
    assert int(am[0, 0]) == 1024
    # utt 0: eos detected, countdown 15 -> step_after_eos 0: channel 0 (delay 0) forced EOS, others keep samples
    assert fsm2[0, :4].tolist() == [1, 14, 0, 0] and int(curs[0]) == 21
    assert torch.equal(tok2[0, 20], am[0])
    # utt 1: bos window: only the -1 entry is written
    want = tok[1, 3].clone(); want[0] = am[1, 0]
    assert torch.equal(tok2[1, 3], want) and fsm2[1, :4].tolist() == [0, -1, 12, 0] and int(curs[1]) == 4
    # next-step embedding of the written rows
    for b, row in ((0, tok2[0, 20]), (1, tok2[1, 3])):
        e = emb[0, int(row[0])].clone()
        for c in range(1, C_):
            e = e + emb[c, int(row[c])]
        assert torch.equal(x[2 * b], e) and torch.equal(x[2 * b + 1], e)
        assert torch.equal(lay.unpack_planes(P, 4, D)[2 * b], e * gw)
        assert (ssq[:, 2 * b].double() - (e.double() ** 2).reshape(-1, 16).sum(-1)).abs().max() < 1e-5


def test_error_paths_on_device():
    L = hb.lib()
    d = dev()
    g = hb.GemmArgs()
    A = torch.zeros(3, 1, 2, 64, 8, dtype=torch.bfloat16, device=d)
    Wt = torch.zeros(1, 4, 64, 8, dtype=torch.bfloat16, device=d)
    g.A, g.a_plane_stride, g.a_ktiles, g.M, g.W, g.KT, g.nstrips = hb.ptr(A), A[0].numel(), 2, 2, hb.ptr(Wt), 4, 1
    assert L.dia_gemm(C.byref(g), None) == -1 and b"exceeds" in L.dia_last_error()


# ---------------------------------------------------------------------------------------------------
# Compaction maps of structured-pruned checkpoints (dia_gemm_args.cmap / strip_map, dia_attn_args.head_map):
# every row-count class the decode step dispatches to — k_gemv_small (M <= 4), k_gemm16 (5..16 rows, one strip
# per workgroup and the persistent multi-strip form), the z-form over 2..3 m-tiles — against float64.
# ---------------------------------------------------------------------------------------------------
def _gemm_args(X, Wt, kt, ns, epi, akt=None):
    A = lay.pack_planes(X, ktiles=akt)
    g = hb.GemmArgs()
    g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], X.shape[0]
    g.W, g.KT, g.nstrips, g.epi = hb.ptr(Wt), kt, ns, epi
    return g, A


@pytest.mark.parametrize("M,K,D,sk", [(2, 2048, 2048, 0), (4, 1024, 512, 0), (6, 2048, 2048, 0), (16, 2048, 2048, 0), (16, 1024, 2048, 0),
                                      (16, 4096, 2048, 4), (20, 2048, 2048, 0), (32, 2048, 2048, 0), (32, 4096, 2048, 4), (40, 512, 512, 0),
                                      (2, 4096, 2048, 2)])
def test_gemm_resid_emit_cmap(M, K, D, sk):
    """RESID_EMIT into a compacted consumer: residual column n is emitted at plane position cmap[n] (or dropped),
    x and the strip sums of squares stay at full width."""
    d = dev()
    torch.manual_seed(M + K + D)
    a = torch.randn(M, K, device=d)
    W = bf16r(torch.randn(K, D, device=d) * 0.03)
    x0 = torch.randn(M, D, device=d)
    gn = bf16r(1.0 + 0.1 * torch.randn(D, device=d))
    keep = torch.rand(D, device=d) < 0.5
    keep[:3] = torch.tensor([True, False, True], device=d)
    nk = int(keep.sum())
    ckt = D // 32                                   # the engine's planes keep the full width; the consumer reads KT < ckt
    cmap = torch.where(keep, torch.cumsum(keep.int(), 0) - 1, torch.full((D,), -1, device=d, dtype=torch.int64)).to(torch.int32)
    Wt, kt, ns = lay.tile_weight(W)
    mpad = (M + 15) // 16 * 16
    x = x0.clone()
    sentinel = 3.0
    P = lay.pack_planes(torch.full((mpad, ckt * 32), sentinel, device=d))
    ssq = torch.zeros(ns, mpad, device=d)
    g, A = _gemm_args(a, Wt, kt, ns, hb.EPI_RESID_EMIT)
    g.ssq_ld, g.out, g.ldo, g.gnext = mpad, hb.ptr(x), D, hb.ptr(gn)
    g.P, g.p_plane_stride, g.p_ktiles, g.ssq_out, g.cmap = hb.ptr(P), P[0].numel(), ckt, hb.ptr(ssq), hb.ptr(cmap)
    if sk:
        scr = torch.zeros(mpad // 16 * ns * sk * 256, device=d)
        tk = torch.zeros(mpad // 16 * ns, dtype=torch.int32, device=d)
        g.sk_scratch, g.sk_tickets, g.sk, g.sk_scratch_floats = hb.ptr(scr), hb.ptr(tk), sk, scr.numel()
    hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
    torch.cuda.synchronize()
    ref = x0.double() + a.double() @ W.double()
    assert (x.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    got = lay.unpack_planes(P, mpad, ckt * 32)
    assert torch.equal(got[:M, :nk], (x * gn)[:, keep])               # compacted order, fp32-exact
    assert (got[:M, nk:] == sentinel).all() and (got[M:] == sentinel).all()   # nothing else is written
    want = (x.double() ** 2).reshape(M, D // 16, 16).sum(-1).T
    assert (ssq[:, :M].double() - want).abs().max().item() <= 1e-5 * want.max().item()


@pytest.mark.parametrize("M,K,heads,live", [(2, 2048, 16, 9), (4, 512, 8, 3), (6, 2048, 16, 9), (16, 2048, 24, 13), (16, 1792, 16, 7),
                                            (20, 2048, 16, 9), (32, 1792, 24, 13), (48, 512, 8, 5)])
def test_gemm_scale_store_strip_map(M, K, heads, live):
    """SCALE_STORE of a matrix whose dead heads were dropped: compact strip s lands at the 16 columns of original
    strip strip_map[s]; columns of dropped heads are not written."""
    d = dev()
    torch.manual_seed(M * 3 + K + heads)
    N = heads * 128
    x = torch.randn(M, K, device=d)
    W = bf16r(torch.randn(K, N, device=d) * 0.05)
    lh = torch.zeros(heads, dtype=torch.bool)
    lh[torch.randperm(heads)[:live]] = True
    strips = []
    for h in torch.nonzero(lh).flatten().tolist():
        strips += list(range(h * 8, h * 8 + 8))
    cols = (torch.tensor(strips)[:, None] * 16 + torch.arange(16)[None, :]).reshape(-1).to(d)
    Wt, kt, ns = lay.tile_weight(W[:, cols])
    assert ns == live * 8
    smap = torch.tensor(strips, dtype=torch.int32, device=d)
    mpad = (M + 15) // 16 * 16
    ssq = strip_ssq(x, mpad)
    out = torch.full((M, N), float("nan"), device=d)
    g, A = _gemm_args(x, Wt, kt, ns, hb.EPI_SCALE_STORE)
    g.ssq_in, g.ssq_in_n, g.inv_d, g.eps, g.ssq_ld = hb.ptr(ssq), ssq.shape[0], 1.0 / K, 1e-5, mpad
    g.out, g.ldo, g.strip_map = hb.ptr(out), N, hb.ptr(smap)
    hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
    torch.cuda.synchronize()
    xd = x.double()
    ref = (xd @ W.double()) * torch.rsqrt((xd ** 2).mean(-1, keepdim=True) + 1e-5)
    livec = torch.zeros(N, dtype=torch.bool, device=d)
    livec[cols] = True
    err = (out[:, livec].double() - ref[:, livec]).abs().max().item()
    assert err <= 2e-5 * max(1.0, ref.abs().max().item()), err
    assert torch.isnan(out[:, ~livec]).all()


@pytest.mark.parametrize("M,K,F,live", [(2, 2048, 8192, 4096), (16, 1024, 8192, 3072), (16, 2048, 2048, 1024), (32, 1024, 4096, 2048)])
def test_gemm_swiglu_emit_compacted_hidden(M, K, F, live):
    """wi_fused with dead hidden units dropped (gate/up columns taken by index, -1 -> zero column) and a K-compacted
    input: the hidden planes come out in the compacted order wo's compacted rows expect."""
    d = dev()
    torch.manual_seed(M + K + F)
    x = torch.randn(M, K, device=d)
    wi3 = bf16r(torch.randn(K, 2, F, device=d) * 0.03)
    idx = torch.sort(torch.randperm(F, device=d)[:live])[0]
    pad = (-live) % 256
    idxp = torch.cat([idx, torch.full((pad,), -1, device=d, dtype=idx.dtype)])
    take = lambda w2: torch.where((idxp >= 0)[None, :], w2[:, idxp.clamp(min=0)], torch.zeros((), device=d))
    wc = torch.stack([take(wi3[:, 0]), take(wi3[:, 1])], dim=1)
    Wt, kt, ns = lay.tile_weight(lay.interleave_gate_up(wc))
    Fc = idxp.numel()
    mpad = (M + 15) // 16 * 16
    ssq = strip_ssq(x, mpad)
    P = torch.zeros(3, mpad // 16, Fc // 32, 64, 8, dtype=torch.bfloat16, device=d)
    g, A = _gemm_args(x, Wt, kt, ns, hb.EPI_SWIGLU_EMIT)
    g.ssq_in, g.ssq_in_n, g.inv_d, g.eps, g.ssq_ld = hb.ptr(ssq), ssq.shape[0], 1.0 / K, 1e-5, mpad
    g.P, g.p_plane_stride, g.p_ktiles = hb.ptr(P), P[0].numel(), Fc // 32
    hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
    torch.cuda.synchronize()
    xd = x.double() * torch.rsqrt((x.double() ** 2).mean(-1, keepdim=True) + 1e-5)
    gate, up = xd @ wi3[:, 0].double(), xd @ wi3[:, 1].double()
    ref = (torch.nn.functional.silu(gate) * up)[:, idx]
    got = lay.unpack_planes(P, M, Fc).double()
    assert (got[:, :live] - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    assert (got[:, live:] == 0).all()                                  # padded hidden units: silu(0) * 0


@pytest.mark.parametrize("kvd", ["f32", "bf16"])
@pytest.mark.parametrize("B,cur", [(1, 40), (4, 300), (8, 77)])
def test_attn_self_head_map(kvd, B, cur):
    """self-attention of a head-pruned layer: live query head h is emitted at head position head_map[h] of the
    compacted o_proj input, dead heads emit nothing; stale plane columns beyond the live heads stay untouched."""
    d = dev()
    torch.manual_seed(cur + B)
    R, QH, KVH, T = 2 * B, 16, 4, 512
    live = torch.tensor([1, 0, 1, 1, 0, 0, 0, 0, 1, 1, 1, 1, 0, 1, 0, 0], dtype=torch.bool)      # kv head 1 fully dead
    hmap = torch.where(live, torch.cumsum(live.int(), 0) - 1, torch.full((QH,), -1)).to(torch.int32).to(d)
    nl = int(live.sum())
    nq = (QH + 2 * KVH) * 128
    qkv = torch.randn(R, nq, device=d)
    kdt = torch.float32 if kvd == "f32" else torch.bfloat16
    kc = torch.randn(R, KVH, T, 128, device=d).to(kdt)
    vc = torch.randn(R, KVH, T, 128, device=d).to(kdt)
    blocked = kvd == "bf16"
    if blocked:
        vc = lay.v_to_blocked(vc)
    cos, sin = [t.to(d) for t in lay.rope_tables(T + 1, 128, 1, 10000)]
    curs = torch.full((B,), cur, dtype=torch.int32, device=d)
    mt = (R + 15) // 16
    sentinel = 5.0
    P = lay.pack_planes(torch.full((mt * 16, QH * 128), sentinel, device=d))
    a = hb.AttnArgs()
    a.mode, a.kv_dtype, a.n_kv_heads, a.group, a.n_rows, a.kv_cap = hb.ATTN_SELF, (0 if kvd == "f32" else 1), KVH, 4, R, T
    a.q, a.ldq, a.q_off, a.k_off, a.v_off = hb.ptr(qkv), nq, 0, QH * 128, (QH + KVH) * 128
    a.kc, a.vc, a.cur = hb.ptr(kc), hb.ptr(vc), hb.ptr(curs)
    a.cos_t, a.sin_t = hb.ptr(cos), hb.ptr(sin)
    a.P, a.p_plane_stride, a.p_ktiles = hb.ptr(P), P[0].numel(), P.shape[2]
    scr = torch.zeros(hb.lib().dia_attn_scratch_floats(R, KVH, T), device=d)
    tk = torch.zeros(R * KVH, dtype=torch.int32, device=d)
    a.scratch, a.tickets, a.head_map, a.v_blocked = hb.ptr(scr), hb.ptr(tk), hb.ptr(hmap), int(blocked)
    hb.check(hb.lib().dia_attn(C.byref(a), None), "dia_attn")
    torch.cuda.synchronize()
    if blocked:
        vc = lay.v_from_blocked(vc)
    out = lay.unpack_planes(P, mt * 16, QH * 128).double().reshape(mt * 16, QH, 128)

    def rope(x, pos):
        c, s = cos[pos].double(), sin[pos].double()
        return torch.cat([x[..., :64] * c - x[..., 64:] * s, x[..., :64] * s + x[..., 64:] * c], dim=-1)

    q = rope(qkv[:, : QH * 128].double().reshape(R, QH, 128), cur)
    worst = 0.0
    for r in range(R):
        for h in range(QH):
            if not live[h]:
                continue
            kvh = h // 4
            ref = attn_ref(q[r, h: h + 1], kc[r, kvh, :cur].double(), vc[r, kvh, :cur].double())
            worst = max(worst, (out[r, int(hmap[h])] - ref[0]).abs().max().item())
    assert worst <= 2e-5, worst
    assert (out[:R, nl:] == sentinel).all() and (out[R:] == sentinel).all()      # head positions >= live count: not written


@pytest.mark.parametrize("M,K,D,sk", [(16, 8192, 2048, 4), (9, 4096, 2048, 2), (16, 4096, 512, 4), (32, 8192, 2048, 4)])
def test_gemm_split_k_paired_strips(M, K, D, sk):
    """split-K with two strips per workgroup (spw = 2): both tiles handed over in ONE slab publication / ticket / merge.
    Bit-identical to the one-strip-per-workgroup split-K (same partial tiles, same slab order), twice, tickets re-armed."""
    d = dev()
    torch.manual_seed(K + D + M)
    a = torch.randn(M, K, device=d)
    W = bf16r(torch.randn(K, D, device=d) * 0.03)
    x0 = torch.randn(M, D, device=d)
    gn = bf16r(1.0 + 0.1 * torch.randn(D, device=d))
    Wt, kt, ns = lay.tile_weight(W)
    mpad = (M + 15) // 16 * 16
    mt = mpad // 16
    outs = []
    for spw in (1, 2, 2):
        x = x0.clone()
        P = torch.zeros(3, mt, D // 32, 64, 8, dtype=torch.bfloat16, device=d)
        ssq = torch.zeros(ns, mpad, device=d)
        g, A = _gemm_args(a, Wt, kt, ns, hb.EPI_RESID_EMIT)
        g.ssq_ld, g.out, g.ldo, g.gnext = mpad, hb.ptr(x), D, hb.ptr(gn)
        g.P, g.p_plane_stride, g.p_ktiles, g.ssq_out = hb.ptr(P), P[0].numel(), D // 32, hb.ptr(ssq)
        scr = torch.zeros(mt * ns * sk * 256, device=d)
        tk = torch.zeros(mt * ns, dtype=torch.int32, device=d)
        g.sk_scratch, g.sk_tickets, g.sk, g.sk_scratch_floats, g.spw = hb.ptr(scr), hb.ptr(tk), sk, scr.numel(), spw
        hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
        torch.cuda.synchronize()
        assert (tk == 0).all()
        outs.append((x, P.clone(), ssq))
    ref = x0.double() + a.double() @ W.double()
    assert (outs[0][0].double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2])
    # the pair kernel runs two epilogues over one LDS staging area (a barrier separates them: without it 96 threads of the first
    # could still be reading the staged planes while the second overwrites them — seen once as an order-dependent parity failure):
    # the same launch, many times back to back, must keep producing the same bits
    if M > 4:
        for _ in range(150):
            x = x0.clone()
            P.zero_()
            g.out = hb.ptr(x)
            hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
        torch.cuda.synchronize()
        assert torch.equal(x, outs[0][0]) and torch.equal(P, outs[0][1])


@pytest.mark.parametrize("M", [2, 16])
def test_padded_o_rows_ignore_stale_planes(M):
    """compact.pad_rows: the zero rows appended to a head-pruned o_proj meet activation-plane columns that this layer's
    attention never wrote (stale but finite leftovers of other layers).  Zero weights times finite values add exactly
    nothing: the result is bit-identical whatever those columns hold."""
    d = dev()
    torch.manual_seed(17 + M)
    live, K, D = 11 * 128, 12 * 128, 2048                       # 11 live heads, K padded to 1536
    a = torch.randn(M, K, device=d)
    W = bf16r(torch.randn(K, D, device=d) * 0.03)
    W[live:] = 0                                                # the padding rows
    x0 = torch.randn(M, D, device=d)
    gn = bf16r(1.0 + 0.1 * torch.randn(D, device=d))
    Wt, kt, ns = lay.tile_weight(W)
    mpad = (M + 15) // 16 * 16
    outs = []
    for stale in (0.0, 3.0e30):
        aa = a.clone()
        aa[:, live:] = stale
        x = x0.clone()
        P = torch.zeros(3, mpad // 16, D // 32, 64, 8, dtype=torch.bfloat16, device=d)
        ssq = torch.zeros(ns, mpad, device=d)
        run_gemm(aa, Wt, kt, ns, hb.EPI_RESID_EMIT, out=x, ldo=D, gnext=gn, P=P, p_kt=D // 32, ssq_out=ssq, ssq_ld=mpad)
        outs.append((x, P.clone(), ssq))
    assert torch.isfinite(outs[1][0]).all()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1]) and torch.equal(outs[0][2], outs[1][2])


# ---------------------------------------------------------------------------------------------------
# fp32 activation tiles (dia_gemm_args.act_f32): the 5..128-row kernels read / write the activations between the
# kernels of a step as 4-byte values in the fragment order of one plane and split the three bf16 planes in
# registers.  The arithmetic is the planes path's, so every output must be IDENTICAL bit for bit.
# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,K,N,epi,sk,mapped", [
    (5, 2048, 2048, "resid", 0, False), (16, 2048, 2048, "resid", 0, True), (16, 8192, 2048, "resid", 4, False),
    (16, 4096, 2048, "resid", 2, True), (40, 2048, 2048, "resid", 0, True), (128, 8192, 2048, "resid", 4, False),
    (16, 2048, 16384, "swiglu", 0, False), (24, 1024, 4096, "swiglu", 0, False),
    (16, 2048, 3072, "store", 0, False), (7, 1792, 1024, "store", 0, False), (64, 2048, 9264, "store", 0, False),
    (16, 96, 80, "resid_generic", 0, False),
    # M <= 4: the LDS-staged GEMV (fp32 image split by the staging threads, one 4-byte store per emitted value)
    (2, 2048, 2048, "resid", 0, False), (2, 2048, 2048, "resid", 0, True), (2, 2048, 3072, "store", 0, False),
    (2, 2048, 16384, "swiglu", 0, False), (2, 8192, 2048, "resid", 2, False), (2, 2048, 9264, "store", 0, False),
    (4, 2048, 2048, "resid", 0, False), (4, 8192, 2048, "resid", 2, True), (3, 1024, 4096, "swiglu", 0, False),
    (1, 1792, 1024, "store", 0, False)])
def test_act_f32_tiles_equal_planes_bitwise(M, K, N, epi, sk, mapped, fmt=3):
    """dia_gemm_args.act_f32 = 3: the A operand and the emitted activations are fp32 tiles (bit 0 / bit 1; mixed formats
    run the generic kernel, whose summation order differs from the 16-row kernel's)"""
    d = dev()
    torch.manual_seed(M + K + N)
    x_in = torch.randn(M, K, device=d) * 3
    Npad = (N + 15) // 16 * 16
    W = torch.zeros(K, Npad, device=d)
    W[:, :N] = bf16r(torch.randn(K, N, device=d) * 0.03)
    Wt, kt, ns = lay.tile_weight(W)
    mpad = (M + 15) // 16 * 16
    mt = mpad // 16
    resid = epi.startswith("resid")
    D_out = Npad if resid else (Npad // 2 if epi == "swiglu" else 0)
    pkt = (D_out + 31) // 32 if D_out else 1
    keep = torch.rand(max(Npad, 1), device=d) < 0.6
    cmap = torch.where(keep, torch.cumsum(keep.int(), 0) - 1, torch.full((Npad,), -1, device=d, dtype=torch.int64)).to(torch.int32)
    gn = bf16r(1.0 + 0.1 * torch.randn(Npad, device=d))
    ssq_in = strip_ssq(x_in, mpad) if not resid else None
    x0 = torch.randn(mpad, Npad, device=d)
    res = []
    for f32 in (0, 1):
        A = lay.pack_planes(x_in)
        af, pf = bool(f32 and fmt & 1), bool(f32 and fmt & 2)
        if af:
            A.view(torch.float32).reshape(-1)[: A[0].numel()] = lay.pack_f32_tiles(x_in).reshape(-1)
        g = hb.GemmArgs()
        g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], M
        g.W, g.KT, g.nstrips = hb.ptr(Wt), kt, ns
        g.epi = {"resid": hb.EPI_RESID_EMIT, "resid_generic": hb.EPI_RESID_EMIT, "swiglu": hb.EPI_SWIGLU_EMIT, "store": hb.EPI_SCALE_STORE}[epi]
        g.act_f32 = fmt * f32
        out = x0.clone() if resid else torch.zeros(mpad, Npad, device=d)
        P = torch.full((3, mt, pkt, 64, 8), 7.0, dtype=torch.bfloat16, device=d)       # (7.0 in bf16 = 0x40E0: as fp32 pairs a finite sentinel too)
        ssq_o = torch.zeros(ns, mpad, device=d)
        g.ssq_ld = mpad
        if ssq_in is not None:
            g.ssq_in, g.ssq_in_n, g.inv_d, g.eps = hb.ptr(ssq_in), ssq_in.shape[0], 1.0 / K, 1e-5
        if epi != "swiglu":
            g.out, g.ldo = hb.ptr(out), Npad
        if resid:
            g.gnext, g.ssq_out = hb.ptr(gn), hb.ptr(ssq_o)
            if mapped:
                g.cmap = hb.ptr(cmap)
        if D_out:
            g.P, g.p_plane_stride, g.p_ktiles = hb.ptr(P), P[0].numel(), pkt
        if sk:
            scr = torch.zeros(mt * ns * sk * 256, device=d)
            tk = torch.zeros(mt * ns, dtype=torch.int32, device=d)
            g.sk_scratch, g.sk_tickets, g.sk, g.sk_scratch_floats = hb.ptr(scr), hb.ptr(tk), sk, scr.numel()
        hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
        torch.cuda.synchronize()
        if D_out:
            if pf:
                emitted = lay.unpack_f32_tiles(P.view(torch.float32).reshape(-1)[: mt * pkt * 512].reshape(mt, pkt, 64, 8), mpad, pkt * 32)
            else:
                emitted = lay.unpack_planes(P, mpad, pkt * 32)
            nlive = int(keep[:Npad].sum()) if (resid and mapped) else D_out
            emitted = emitted[:M, :nlive].clone()
        else:
            emitted = None
        res.append((out[:M].clone(), ssq_o[:, :M].clone(), emitted))
    (o0, s0, e0), (o1, s1, e1) = res
    assert torch.equal(o0, o1) and torch.equal(s0, s1)
    if e0 is not None:
        assert torch.equal(e0, e1)
        assert e0.abs().max().item() > 0
    # and the values are right (float64), not merely equal
    if epi == "store":
        xd = x_in.double()
        ref = (xd @ W.double()) * torch.rsqrt((xd ** 2).mean(-1, keepdim=True) + 1e-5)
        assert (o1.double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    elif resid:
        ref = x0[:M].double() + x_in.double() @ W.double()
        assert (o1.double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("kvd,B,cur", [("bf16", 4, 300), ("f32", 3, 70), ("bf16", 8, 33)])
def test_act_f32_attention_output_equals_planes(kvd, B, cur):
    """dia_attn_args.act_f32: the attention output as fp32 tiles holds exactly what the three planes sum to"""
    d = dev()
    torch.manual_seed(cur + B)
    R, QH, KVH, T = 2 * B, 16, 4, 512
    nq = (QH + 2 * KVH) * 128
    qkv = torch.randn(R, nq, device=d)
    kdt = torch.float32 if kvd == "f32" else torch.bfloat16
    kc0 = torch.randn(R, KVH, T, 128, device=d).to(kdt)
    vc0 = torch.randn(R, KVH, T, 128, device=d).to(kdt)
    blocked = kvd == "bf16"
    if blocked:
        vc0 = lay.v_to_blocked(vc0)
    cos, sin = [t.to(d) for t in lay.rope_tables(T + 1, 128, 1, 10000)]
    curs = torch.full((B,), cur, dtype=torch.int32, device=d)
    mt = (R + 15) // 16
    outs = []
    for f32 in (0, 1):
        kc, vc = kc0.clone(), vc0.clone()
        P = torch.zeros(3, mt, QH * 4, 64, 8, dtype=torch.bfloat16, device=d)
        a = hb.AttnArgs()
        a.mode, a.kv_dtype, a.n_kv_heads, a.group, a.n_rows, a.kv_cap = hb.ATTN_SELF, (0 if kvd == "f32" else 1), KVH, 4, R, T
        a.q, a.ldq, a.q_off, a.k_off, a.v_off = hb.ptr(qkv), nq, 0, QH * 128, (QH + KVH) * 128
        a.kc, a.vc, a.cur = hb.ptr(kc), hb.ptr(vc), hb.ptr(curs)
        a.cos_t, a.sin_t = hb.ptr(cos), hb.ptr(sin)
        a.P, a.p_plane_stride, a.p_ktiles, a.act_f32 = hb.ptr(P), P[0].numel(), P.shape[2], f32
        scr = torch.zeros(hb.lib().dia_attn_scratch_floats(R, KVH, T), device=d)
        tk = torch.zeros(R * KVH, dtype=torch.int32, device=d)
        a.scratch, a.tickets, a.v_blocked = hb.ptr(scr), hb.ptr(tk), int(blocked)
        hb.check(hb.lib().dia_attn(C.byref(a), None), "dia_attn")
        torch.cuda.synchronize()
        if f32:
            outs.append(lay.unpack_f32_tiles(P.view(torch.float32).reshape(-1)[: mt * QH * 4 * 512].reshape(mt, QH * 4, 64, 8), R, QH * 128))
        else:
            outs.append(lay.unpack_planes(P, R, QH * 128))
    assert torch.equal(outs[0], outs[1]) and outs[0].abs().max().item() > 0


@pytest.mark.parametrize("M,K,N,epi", [(2, 2048, 2048, "resid"), (2, 512, 1024, "store"), (16, 1024, 4096, "swiglu"), (40, 512, 512, "resid"),
                                       (130, 256, 320, "store")])
def test_gemm_fp32_weights_as_three_planes(M, K, N, epi):
    """dia_gemm_args.w_planes = 3: fp32 weights that bf16 cannot hold, streamed as hi / mid / lo tile sets — the products are
    those of fp32 x fp32 (float64 reference to 2e-6 of the output scale; the single rounded tile set misses by ~4e-3)."""
    d = dev()
    torch.manual_seed(M + K + N)
    x = torch.randn(M, K, device=d)
    W = torch.randn(K, N, device=d) * 0.05                      # NOT bf16-representable
    Wt3, kt, ns = lay.tile_weight_planes(W)
    mpad = (M + 15) // 16 * 16
    A = lay.pack_planes(x)
    g = hb.GemmArgs()
    g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], M
    g.W, g.KT, g.nstrips, g.w_planes = hb.ptr(Wt3), kt, ns, 3
    g.ssq_ld = mpad
    xd, Wd = x.double(), W.double()
    if epi == "store":
        ssq = strip_ssq(x, mpad)
        out = torch.zeros(mpad, N, device=d)
        g.epi, g.ssq_in, g.ssq_in_n, g.inv_d, g.eps, g.out, g.ldo = hb.EPI_SCALE_STORE, hb.ptr(ssq), ssq.shape[0], 1.0 / K, 1e-5, hb.ptr(out), N
        ref = (xd @ Wd) * torch.rsqrt((xd ** 2).mean(-1, keepdim=True) + 1e-5)
        get = lambda: out[:M].double()
    elif epi == "resid":
        x0 = torch.randn(mpad, N, device=d)
        out = x0.clone()
        gn = torch.ones(N, device=d)
        P = torch.zeros(3, mpad // 16, N // 32, 64, 8, dtype=torch.bfloat16, device=d)
        so = torch.zeros(ns, mpad, device=d)
        g.epi, g.out, g.ldo, g.gnext, g.P, g.p_plane_stride, g.p_ktiles, g.ssq_out = hb.EPI_RESID_EMIT, hb.ptr(out), N, hb.ptr(gn), hb.ptr(P), P[0].numel(), N // 32, hb.ptr(so)
        ref = x0[:M].double() + xd @ Wd
        get = lambda: out[:M].double()
    else:
        ssq = strip_ssq(x, mpad)
        P = torch.zeros(3, mpad // 16, N // 64, 64, 8, dtype=torch.bfloat16, device=d)
        g.epi, g.ssq_in, g.ssq_in_n, g.inv_d, g.eps = hb.EPI_SWIGLU_EMIT, hb.ptr(ssq), ssq.shape[0], 1.0 / K, 1e-5
        g.P, g.p_plane_stride, g.p_ktiles = hb.ptr(P), P[0].numel(), N // 64
        xn = xd * torch.rsqrt((xd ** 2).mean(-1, keepdim=True) + 1e-5)
        Wg = Wd.reshape(K, N // 16, 2, 8)                        # strips of 8 gate + 8 up columns
        ref = (torch.nn.functional.silu(xn @ Wg[:, :, 0].reshape(K, -1)) * (xn @ Wg[:, :, 1].reshape(K, -1)))
        get = lambda: lay.unpack_planes(P, M, N // 2).double()
    hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
    torch.cuda.synchronize()
    err = (get() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    assert err <= 2e-6, err
    # the same launch with fp32 activation tiles on both sides (what the engine passes from 5 rows on): identical bits
    if M > 4:
        first = get().clone()
        A32 = lay.pack_planes(x)
        A32.view(torch.float32).reshape(-1)[: A32[0].numel()] = lay.pack_f32_tiles(x).reshape(-1)
        g.A, g.act_f32 = hb.ptr(A32), 3
        if epi == "resid":
            out.copy_(x0)
        if epi != "store":
            P.zero_()
        hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
        torch.cuda.synchronize()
        if epi == "swiglu":
            again = lay.unpack_f32_tiles(P.view(torch.float32).reshape(-1)[: (mpad // 16) * (N // 64) * 512].reshape(mpad // 16, N // 64, 64, 8), M, N // 2).double()
        else:
            again = get()
        assert torch.equal(again, first)
        g.A, g.act_f32 = hb.ptr(A), 0
    # the rounded single tile set on the same problem, for scale
    Wt1, _, _ = lay.tile_weight(W)
    g.W, g.w_planes = hb.ptr(Wt1), 0
    if epi == "resid":
        out.copy_(x0)
    hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
    torch.cuda.synchronize()
    err1 = (get() - ref).abs().max().item() / max(1.0, ref.abs().max().item())
    assert err1 > 20 * err, (err, err1)


@pytest.mark.parametrize("M,K,D", [(2, 8192, 2048), (4, 4096, 2048), (1, 8192, 2048), (2, 2048, 2048), (3, 4096, 512)])
def test_gemm_diagonal_layout_resid(M, K, D):
    """wo at <= 4 rows from the diagonal weight layout (dia_gemm_args.w_layout = 1: 4-column groups, 256 workgroups with the
    whole K, no split-K): x += a . W, fp32 tile of x * g_next, one sum of squares per 8-column half strip — against float64."""
    d = dev()
    torch.manual_seed(K + D + M)
    a = torch.randn(M, K, device=d)
    W = bf16r(torch.randn(K, D, device=d) * 0.03)
    x0 = torch.randn(M, D, device=d)
    gn = bf16r(1.0 + 0.1 * torch.randn(D, device=d))
    Wd = lay.diag_tile_weight(W)
    assert tuple(Wd.shape) == (D // 4, K // 128, 64, 8)
    A = lay.pack_f32_tiles(a, ktiles=K // 32, mtiles=1)
    x = torch.zeros(16, D, device=d); x[:M] = x0
    P = torch.zeros(3, 1, D // 32, 64, 8, dtype=torch.bfloat16, device=d)          # fp32 tiles live in the planes buffer
    ssq = torch.full((D // 8, 16), float("nan"), device=d)
    g = hb.GemmArgs()
    g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel() * 2, K // 32, M
    g.W, g.KT, g.nstrips, g.epi = hb.ptr(Wd), K // 32, D // 8, hb.EPI_RESID_EMIT
    g.ssq_ld, g.out, g.ldo, g.gnext = 16, hb.ptr(x), D, hb.ptr(gn)
    g.P, g.p_plane_stride, g.p_ktiles, g.ssq_out = hb.ptr(P), P[0].numel(), D // 32, hb.ptr(ssq)
    g.act_f32, g.w_layout = 3, 1
    outs = []
    for _ in range(2):
        x[:M] = x0
        hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm(diag)")
        torch.cuda.synchronize()
        outs.append(x[:M].clone())
    assert torch.equal(outs[0], outs[1])
    ref = x0.double() + a.double() @ W.double()
    assert (outs[0].double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    xt = P.view(torch.uint8).view(-1)[: (D // 32) * 64 * 8 * 4].view(torch.float32).reshape(1, D // 32, 64, 8)
    assert torch.equal(lay.unpack_f32_tiles(xt, M, D), outs[0] * gn)
    want = (outs[0].double() ** 2).reshape(M, D // 8, 8).sum(-1).T
    assert (ssq[:, :M].double() - want).abs().max().item() <= 1e-5 * want.max().item()
    # what it is not built for is refused
    g.epi = hb.EPI_SCALE_STORE
    assert hb.lib().dia_gemm(C.byref(g), None) == -1


@pytest.mark.parametrize("M,K,N,epi,f32", [
    # planes in and out (the short-prompt prefill): K = 1024 -> the 256-thread form, K = 2048 -> the 512-thread form
    (98, 1024, 6144, "store", 0), (128, 1024, 8192, "swiglu", 0), (40, 1024, 1024, "swiglu", 0), (98, 2048, 1024, "resid", 0),
    (128, 2048, 2048, "resid", 0),
    # fp32 tiles in and out (the decode step at 33..128 rows)
    (64, 2048, 9264, "store", 1), (128, 2048, 16384, "swiglu", 1), (48, 2048, 16384, "swiglu", 1), (100, 2048, 2048, "resid", 1)])
def test_gemm2t_uniform_tails_equal_run_time_tails(M, K, N, epi, f32):
    """k_gemm2t with a compile-time epilogue and stores issued by every thread (idle lanes into a sink) against the tails that
    branch on the epilogue kind at run time (knob gemm_2t=5): same bits in out, ssq and the emitted activations"""
    d = dev()
    torch.manual_seed(M + K + N)
    x_in = torch.randn(M, K, device=d) * 2
    Npad = (N + 15) // 16 * 16
    W = torch.zeros(K, Npad, device=d)
    W[:, :N] = bf16r(torch.randn(K, N, device=d) * 0.03)
    Wt, kt, ns = lay.tile_weight(W)
    mpad = (M + 15) // 16 * 16
    mt = mpad // 16
    resid = epi == "resid"
    D_out = Npad if resid else (Npad // 2 if epi == "swiglu" else 0)
    pkt = (D_out + 31) // 32 if D_out else 1
    gn = bf16r(1.0 + 0.1 * torch.randn(Npad, device=d))
    ssq_in = strip_ssq(x_in, mpad) if not resid else None
    x0 = torch.randn(mpad, Npad, device=d)
    res = []
    try:
        for knob in (5, -1):
            hb.set_tuning("gemm_2t", knob)
            A = lay.pack_planes(x_in)
            if f32:
                A.view(torch.float32).reshape(-1)[: A[0].numel()] = lay.pack_f32_tiles(x_in).reshape(-1)
            g = hb.GemmArgs()
            g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], M
            g.W, g.KT, g.nstrips = hb.ptr(Wt), kt, ns
            g.epi = {"resid": hb.EPI_RESID_EMIT, "swiglu": hb.EPI_SWIGLU_EMIT, "store": hb.EPI_SCALE_STORE}[epi]
            g.act_f32 = 3 * f32
            out = x0.clone() if resid else torch.zeros(mpad, Npad, device=d)
            P = torch.full((3, mt, pkt, 64, 8), 7.0, dtype=torch.bfloat16, device=d)
            ssq_o = torch.zeros(ns, mpad, device=d)
            g.ssq_ld = mpad
            if ssq_in is not None:
                g.ssq_in, g.ssq_in_n, g.inv_d, g.eps = hb.ptr(ssq_in), ssq_in.shape[0], 1.0 / K, 1e-5
            if epi != "swiglu":
                g.out, g.ldo = hb.ptr(out), Npad
            if resid:
                g.gnext, g.ssq_out = hb.ptr(gn), hb.ptr(ssq_o)
            if D_out:
                g.P, g.p_plane_stride, g.p_ktiles = hb.ptr(P), P[0].numel(), pkt
            hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
            torch.cuda.synchronize()
            res.append((out.clone(), ssq_o.clone(), P.clone()))
    finally:
        hb.set_tuning("gemm_2t", -1)
    (o0, s0, p0), (o1, s1, p1) = res
    assert torch.equal(o0[:M], o1[:M]) and torch.equal(s0[:, :M], s1[:, :M])
    if D_out:
        un = (lambda P_: lay.unpack_f32_tiles(P_.view(torch.float32).reshape(-1)[: mt * pkt * 512].reshape(mt, pkt, 64, 8), mpad, pkt * 32)) if f32 \
            else (lambda P_: lay.unpack_planes(P_, mpad, pkt * 32))
        e0, e1 = un(p0)[:M, :D_out], un(p1)[:M, :D_out]
        assert torch.equal(e0, e1) and e0.abs().max().item() > 0
    # rows past M and the padding of the buffers are untouched by the sink stores
    assert torch.equal(o1[M:], (x0 if resid else torch.zeros_like(x0))[M:])
    if epi == "store":
        xd = x_in.double()
        ref = (xd @ W.double()) * torch.rsqrt((xd ** 2).mean(-1, keepdim=True) + 1e-5)
        assert (o1[:M].double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    elif resid:
        ref = x0[:M].double() + x_in.double() @ W.double()
        assert (o1[:M].double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("kvd,lens", [("bf16", (98,)), ("bf16", (40, 0, 57)), ("f32", (70, 33))])
def test_gemm_crosskv_layers_merged_equals_per_layer(kvd, lens):
    """dia_gemm_args.kv_layer_strips / kv_layer_stride: the cross-K/V projections of several decoder layers as ONE launch over their common
    input (tile sets back to back, caches of all layers in one allocation) write the bits of one launch per layer; bf16 + blocked V at
    K = 1024 and 33..128 rows runs the all-thread tail (k_gemm2t<8, false, false, 4, true>), fp32 caches the shared epilogue"""
    d = dev()
    torch.manual_seed(11)
    E, H, NL = 1024, 16, 3
    B = len(lens)
    offs, tot = [], 0
    for Lb in lens:
        offs.append(tot)
        tot += (Lb + 31) // 32 * 32
    Mp = tot
    cap = 128
    x = torch.randn(Mp, E, device=d)
    rb = np.full((Mp,), -1, dtype=np.int32)
    for b, Lb in enumerate(lens):
        rb[offs[b]: offs[b] + Lb] = b
    row_b = torch.from_numpy(rb).to(d)
    seg_off = torch.tensor(offs, dtype=torch.int32, device=d)
    perm = lay.rope_pair_perm(128).to(d)
    tiles = []
    for _ in range(NL):
        wk = bf16r(torch.randn(E, H, 128, device=d) * 0.05)
        wv = bf16r(torch.randn(E, H, 128, device=d) * 0.05)
        Wt, kt, ns = lay.tile_weight(torch.cat([wk[:, :, perm].reshape(E, -1), wv.reshape(E, -1)], dim=1))
        tiles.append(Wt)
    W_all = torch.cat([t.reshape(-1) for t in tiles])
    cos, sin = [t.to(d) for t in lay.rope_tables(cap + 1, 128, 1, 10000)]
    kdt = torch.float32 if kvd == "f32" else torch.bfloat16
    code = hb.KV_F32 if kvd == "f32" else hb.KV_BF16
    blocked = int(kvd == "bf16")
    A = lay.pack_planes(x)
    ssq = strip_ssq(x, Mp)

    def launch(Wptr, nstrips, kc, vc, layers):
        g = hb.GemmArgs()
        g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], Mp
        g.W, g.KT, g.nstrips, g.epi = Wptr, kt, nstrips, hb.EPI_CROSSKV
        g.ssq_in, g.ssq_in_n, g.inv_d, g.eps, g.ssq_ld = hb.ptr(ssq), E // 16, 1.0 / E, 1e-5, Mp
        g.kc, g.vc, g.kv_dtype, g.kv_heads, g.kv_cap, g.kv_batch_index = hb.ptr(kc), hb.ptr(vc), code, H, cap, 0
        g.cos_t, g.sin_t, g.kv_vblocked = hb.ptr(cos), hb.ptr(sin), blocked
        g.row_b, g.seg_off = hb.ptr(row_b), hb.ptr(seg_off)
        if layers:
            g.kv_layer_strips, g.kv_layer_stride = H * 16, kc[0].numel()
        hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")

    k1 = torch.zeros(NL, B, H, cap, 128, dtype=kdt, device=d)
    v1 = torch.zeros_like(k1)
    k2, v2 = torch.zeros_like(k1), torch.zeros_like(k1)
    for l in range(NL):
        launch(hb.ptr(tiles[l]), ns, k1[l], v1[l], False)
    launch(hb.ptr(W_all), NL * ns, k2, v2, True)
    torch.cuda.synchronize()
    assert torch.equal(k1, k2) and torch.equal(v1, v2)
    assert k1.abs().max().item() > 0 and v1[NL - 1].abs().max().item() > 0
    for b, Lb in enumerate(lens):                       # nothing behind an utterance's text
        assert (k2[:, b, :, Lb:] == 0).all()
    # and one layer against float64
    xd = x.double()
    h = xd * torch.rsqrt((xd ** 2).mean(-1, keepdim=True) + 1e-5)
    Wl = lay.untile_weight(tiles[NL - 1], E, 2 * H * 128).double() if hasattr(lay, "untile_weight") else None
    if Wl is not None:
        inv = torch.argsort(perm)
        kfull = (h @ Wl[:, : H * 128]).reshape(Mp, H, 128)[:, :, inv]
        b0 = next(b for b, Lb in enumerate(lens) if Lb > 0)
        rows = slice(offs[b0], offs[b0] + lens[b0])
        c, s = cos[: lens[b0]].double()[:, None, :], sin[: lens[b0]].double()[:, None, :]
        kk = kfull[rows]
        kr = torch.cat([kk[..., :64] * c - kk[..., 64:] * s, kk[..., :64] * s + kk[..., 64:] * c], dim=-1)
        got = k2[NL - 1, b0, :, : lens[b0]].double().transpose(0, 1)
        tol = 2e-5 if kvd == "f32" else 1e-2
        assert (got - kr).abs().max().item() <= tol * kr.abs().max().item()


@pytest.mark.parametrize("M,K,D,sk,f32", [(128, 8192, 2048, 4, 1), (100, 8192, 2048, 4, 1), (64, 8192, 2048, 4, 1), (32, 8192, 2048, 4, 1),
                                          (98, 4096, 1024, 2, 0), (128, 4096, 1024, 2, 0), (40, 8192, 1024, 4, 0)])
def test_gemm2t_grouped_handoff_equals_per_strip(M, K, D, sk, f32):
    """k_gemm2t split-K: the K slices handed over once per group of four / two strips (SK2) against once per strip (knob gemm_2t=6) and
    against pairs only (knob 7): same bits in x, the sums of squares and the emitted activations; tickets re-armed; float64 reference"""
    d = dev()
    torch.manual_seed(M + K + D + sk)
    a = torch.randn(M, K, device=d)
    W = bf16r(torch.randn(K, D, device=d) * 0.02)
    Wt, kt, ns = lay.tile_weight(W)
    mpad = (M + 15) // 16 * 16
    mt = mpad // 16
    pkt = D // 32
    gn = bf16r(1.0 + 0.1 * torch.randn(D, device=d))
    x0 = torch.randn(mpad, D, device=d)
    res = []
    try:
        for knob in (6, 7, -1):
            hb.set_tuning("gemm_2t", knob)
            A = lay.pack_planes(a)
            if f32:
                A.view(torch.float32).reshape(-1)[: A[0].numel()] = lay.pack_f32_tiles(a).reshape(-1)
            g = hb.GemmArgs()
            g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], M
            g.W, g.KT, g.nstrips, g.epi, g.act_f32 = hb.ptr(Wt), kt, ns, hb.EPI_RESID_EMIT, 3 * f32
            xr = x0.clone()
            P = torch.full((3, mt, pkt, 64, 8), 7.0, dtype=torch.bfloat16, device=d)
            ssq = torch.zeros(ns, mpad, device=d)
            g.ssq_ld, g.out, g.ldo, g.gnext = mpad, hb.ptr(xr), D, hb.ptr(gn)
            g.P, g.p_plane_stride, g.p_ktiles, g.ssq_out = hb.ptr(P), P[0].numel(), pkt, hb.ptr(ssq)
            scr = torch.zeros(mt * ns * sk * 256, device=d)
            tk = torch.zeros(mt * ns, dtype=torch.int32, device=d)
            g.sk, g.sk_scratch, g.sk_tickets, g.sk_scratch_floats = sk, hb.ptr(scr), hb.ptr(tk), scr.numel()
            for _ in range(2):                           # twice: the tickets must come back to zero
                xr.copy_(x0)
                hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
            torch.cuda.synchronize()
            assert (tk == 0).all()
            res.append((xr.clone(), ssq.clone(), P.clone()))
    finally:
        hb.set_tuning("gemm_2t", -1)
    for other in res[:2]:
        assert torch.equal(other[0][:M], res[2][0][:M]) and torch.equal(other[1][:, :M], res[2][1][:, :M])
        un = (lambda P_: lay.unpack_f32_tiles(P_.view(torch.float32).reshape(-1)[: mt * pkt * 512].reshape(mt, pkt, 64, 8), mpad, D)) if f32 \
            else (lambda P_: lay.unpack_planes(P_, mpad, D))
        assert torch.equal(un(other[2])[:M], un(res[2][2])[:M])
    ref = x0[:M].double() + a.double() @ W.double()
    assert (res[2][0][:M].double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(res[2][0][M:], x0[M:])
