"""Kernel-level tests of the measured-and-rejected GEMM forms (csrc/gemm_experiments.hip).  They only run against a
library built with `make -C dia-tts-prune_amd/csrc EXPERIMENTS=1`; the product build does not contain these kernels and
the whole module is skipped."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from dia_hip import binding as hb
from dia_hip import layout as lay
from test_gpu_kernels import bf16r, dev, run_gemm, strip_ssq


@pytest.fixture(autouse=True)
def _needs_experiments():
    if not hb.has_experiments():
        pytest.skip("library built without EXPERIMENTS=1")


@pytest.mark.parametrize("M,K,N,epi", [(2, 2048, 4096, "swiglu"), (4, 2048, 2048, "resid"), (1, 1024, 1024, "store"), (2, 512, 4096, "swiglu")])
def test_gemv_sparse_stream(M, K, N, epi):
    """unstructured-pruned matrix as a zero-skipping stream (layout.sparse_tile_weight): bit-identical to the dense
    tiles of the same matrix, through every epilogue; a dense corner exercises the raw-tile escape."""
    d = dev()
    torch.manual_seed(K + N + M)
    W = bf16r(torch.randn(K, N, device=d) * 0.05)
    W[torch.rand_like(W) < 0.5] = 0
    W[: 64, : 32] = bf16r(torch.randn(64, 32, device=d) * 0.05 + 0.5)          # fully dense tiles -> stored raw
    W[: 64, N // 2: N // 2 + 32] = bf16r(torch.randn(64, 32, device=d) * 0.05 + 0.5)   # (the "up" half too, for the interleaved wi layout)
    x = torch.randn(M, K, device=d)
    mpad = 16
    ssq = strip_ssq(x, mpad)
    A = lay.pack_planes(x)
    if epi == "swiglu":
        Wt, kt, ns = lay.tile_weight(lay.interleave_gate_up(W.reshape(K, 2, N // 2)))
    else:
        Wt, kt, ns = lay.tile_weight(W)
    blocks, toff = lay.sparse_tile_weight(Wt)
    assert int(((toff & 255) == 0).sum()) >= 1 and blocks.numel() < 0.7 * Wt.numel() * 2
    gn = bf16r(1.0 + 0.1 * torch.randn(N, device=d))
    outs = []
    for sparse in (False, True):
        g = hb.GemmArgs()
        g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], M
        g.KT, g.nstrips = kt, ns
        if sparse:
            g.sp_blocks, g.sp_toff = hb.ptr(blocks), hb.ptr(toff)
        else:
            g.W, g.nw = hb.ptr(Wt), 16                 # the same 16-wave K split as the sparse kernel: identical summation order
        g.ssq_ld = mpad
        out = torch.zeros(mpad, N, device=d)
        P = torch.zeros(3, 1, max(N // 32, 1), 64, 8, dtype=torch.bfloat16, device=d)
        sso = torch.zeros(ns, mpad, device=d)
        if epi == "swiglu":
            g.epi = hb.EPI_SWIGLU_EMIT
            g.ssq_in, g.ssq_in_n, g.inv_d, g.eps = hb.ptr(ssq), K // 16, 1.0 / K, 1e-5
            g.P, g.p_plane_stride, g.p_ktiles = hb.ptr(P), P[0].numel(), P.shape[2]
        elif epi == "resid":
            g.epi = hb.EPI_RESID_EMIT
            out[:M] = torch.randn(M, N, device=d, generator=torch.Generator(device=d).manual_seed(1))
            g.out, g.ldo, g.gnext = hb.ptr(out), N, hb.ptr(gn)
            g.P, g.p_plane_stride, g.p_ktiles, g.ssq_out = hb.ptr(P), P[0].numel(), P.shape[2], hb.ptr(sso)
        else:
            g.epi = hb.EPI_SCALE_STORE
            g.ssq_in, g.ssq_in_n, g.inv_d, g.eps = hb.ptr(ssq), K // 16, 1.0 / K, 1e-5
            g.out, g.ldo = hb.ptr(out), N
        hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
        torch.cuda.synchronize()
        outs.append((out.clone(), P.clone(), sso.clone()))
    for a_, b_ in zip(outs[0], outs[1]):
        assert torch.equal(a_, b_)
    if epi == "store":
        xd = x.double()
        ref = (xd @ W.double()) * torch.rsqrt((xd ** 2).mean(-1, keepdim=True) + 1e-5)
        assert (outs[1][0][:M].double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("M", [17, 20, 32])
def test_gemm_two_mtiles(M, tuning):
    """17..32 rows (batch 9-16): k_gemm32 — both m-tiles' A fragments in registers; K = 2048 in one workgroup
    (selected by a debug knob only: it loses to the generic kernel), K = 8192 split four ways by dia_gemm itself
    when the scratch capacity is stated."""
    tuning("g32_all", 1)
    tuning("gemm_mz_max", 0)        # not the paired one-m-tile kernel (test_gemm_paired_mtiles)
    d = dev()
    torch.manual_seed(M)
    # SCALE_STORE with the row norm, K = 2048
    K, N = 2048, 512
    x = torch.randn(M, K, device=d) * 2.0
    gw = bf16r(1.0 + 0.1 * torch.randn(K, device=d))
    W = bf16r(torch.randn(K, N, device=d) * 0.05)
    Wt, kt, ns = lay.tile_weight(W)
    out = torch.full((M, N), float("nan"), device=d)
    run_gemm(x * gw, Wt, kt, ns, hb.EPI_SCALE_STORE, ssq_in=strip_ssq(x, 32), inv_d=1.0 / K, eps=1e-5, out=out, ldo=N, ssq_ld=32)
    xd = x.double()
    ref = ((xd * gw.double()) @ W.double()) * torch.rsqrt((xd ** 2).mean(-1, keepdim=True) + 1e-5)
    assert (out.double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # SWIGLU, K = 2048
    F = 1024
    wi = bf16r(torch.randn(K, 2, F, device=d) * 0.05)
    Wt, kt, ns = lay.tile_weight(lay.interleave_gate_up(wi))
    P = torch.zeros(3, 2, F // 32, 64, 8, dtype=torch.bfloat16, device=d)
    run_gemm(x * gw, Wt, kt, ns, hb.EPI_SWIGLU_EMIT, ssq_in=strip_ssq(x, 32), inv_d=1.0 / K, eps=1e-5, P=P, p_kt=F // 32, ssq_ld=32)
    h = (xd * torch.rsqrt((xd ** 2).mean(-1, keepdim=True) + 1e-5)) * gw.double()
    f = torch.einsum("mk,kgf->mgf", h, wi.double())
    ref = torch.nn.functional.silu(f[:, 0]) * f[:, 1]
    assert (lay.unpack_planes(P, M, F).double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # persistent two-half form (k_gemm32m): K = 2048, >= 128 strips, scratch lent -> SWIGLU again + RESID_EMIT
    tuning("g32_all", -1)
    tuning("g32m", 1)
    F2 = 2048
    wi2 = bf16r(torch.randn(K, 2, F2, device=d) * 0.05)
    Wt, kt, ns = lay.tile_weight(lay.interleave_gate_up(wi2))
    A = lay.pack_planes(x * gw)
    scr = torch.zeros(ns * 2 * 512, device=d); tk = torch.zeros(ns, dtype=torch.int32, device=d)
    ss = strip_ssq(x, 32)
    for _ in range(2):
        P = torch.zeros(3, 2, F2 // 32, 64, 8, dtype=torch.bfloat16, device=d)
        g = hb.GemmArgs()
        g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], M
        g.W, g.KT, g.nstrips, g.epi = hb.ptr(Wt), kt, ns, hb.EPI_SWIGLU_EMIT
        g.ssq_in, g.ssq_in_n, g.inv_d, g.eps, g.ssq_ld = hb.ptr(ss), K // 16, 1.0 / K, 1e-5, 32
        g.P, g.p_plane_stride, g.p_ktiles = hb.ptr(P), P[0].numel(), F2 // 32
        g.sk_scratch, g.sk_tickets, g.sk_scratch_floats = hb.ptr(scr), hb.ptr(tk), scr.numel()
        hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
    torch.cuda.synchronize()
    assert (tk == 0).all()
    f = torch.einsum("mk,kgf->mgf", h, wi2.double())
    ref = torch.nn.functional.silu(f[:, 0]) * f[:, 1]
    assert (lay.unpack_planes(P, M, F2).double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    Dn = 2048
    Wn = bf16r(torch.randn(K, Dn, device=d) * 0.03)
    xn0 = torch.randn(M, Dn, device=d); gnn = bf16r(1.0 + 0.1 * torch.randn(Dn, device=d))
    Wt, kt, ns = lay.tile_weight(Wn)
    An = lay.pack_planes(x)
    xr = xn0.clone()
    Pn = torch.zeros(3, 2, Dn // 32, 64, 8, dtype=torch.bfloat16, device=d); ssqn = torch.zeros(ns, 32, device=d)
    scr = torch.zeros(ns * 2 * 512, device=d); tk = torch.zeros(ns, dtype=torch.int32, device=d)
    g = hb.GemmArgs()
    g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(An), An[0].numel(), An.shape[2], M
    g.W, g.KT, g.nstrips, g.epi = hb.ptr(Wt), kt, ns, hb.EPI_RESID_EMIT
    g.ssq_ld, g.out, g.ldo, g.gnext = 32, hb.ptr(xr), Dn, hb.ptr(gnn)
    g.P, g.p_plane_stride, g.p_ktiles, g.ssq_out = hb.ptr(Pn), Pn[0].numel(), Dn // 32, hb.ptr(ssqn)
    g.sk_scratch, g.sk_tickets, g.sk_scratch_floats = hb.ptr(scr), hb.ptr(tk), scr.numel()
    hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
    torch.cuda.synchronize()
    refn = xn0.double() + x.double() @ Wn.double()
    assert (xr.double() - refn).abs().max().item() <= 2e-5 * refn.abs().max().item()
    assert torch.equal(lay.unpack_planes(Pn, M, Dn), xr * gnn)
    want = (xr.double() ** 2).reshape(M, Dn // 16, 16).sum(-1).T
    assert (ssqn[:, :M].double() - want).abs().max().item() <= 1e-5 * want.max().item()
    # RESID_EMIT, K = 8192: split-K 4 inside dia_gemm, twice for reproducibility
    K2, D = 8192, 256
    a = torch.randn(M, K2, device=d)
    W2 = bf16r(torch.randn(K2, D, device=d) * 0.03)
    x0 = torch.randn(M, D, device=d)
    gn = bf16r(1.0 + 0.1 * torch.randn(D, device=d))
    Wt, kt, ns = lay.tile_weight(W2)
    A = lay.pack_planes(a)
    scr = torch.zeros(ns * 4 * 512, device=d); tk = torch.zeros(ns, dtype=torch.int32, device=d)
    outs = []
    for _ in range(2):
        xr = x0.clone()
        P = torch.zeros(3, 2, D // 32, 64, 8, dtype=torch.bfloat16, device=d)
        ssq = torch.zeros(ns, 32, device=d)
        g = hb.GemmArgs()
        g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], M
        g.W, g.KT, g.nstrips, g.epi = hb.ptr(Wt), kt, ns, hb.EPI_RESID_EMIT
        g.ssq_ld, g.out, g.ldo, g.gnext = 32, hb.ptr(xr), D, hb.ptr(gn)
        g.P, g.p_plane_stride, g.p_ktiles, g.ssq_out = hb.ptr(P), P[0].numel(), D // 32, hb.ptr(ssq)
        g.sk_scratch, g.sk_tickets, g.sk_scratch_floats = hb.ptr(scr), hb.ptr(tk), scr.numel()
        hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
        torch.cuda.synchronize()
        outs.append((xr, P, ssq))
    assert (tk == 0).all()
    ref = x0.double() + a.double() @ W2.double()
    assert (outs[0][0].double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.equal(lay.unpack_planes(outs[0][1], M, D), outs[0][0] * gn)
    want = (outs[0][0].double() ** 2).reshape(M, D // 16, 16).sum(-1).T
    assert (outs[0][2][:, :M].double() - want).abs().max().item() <= 1e-5 * want.max().item()


@pytest.mark.parametrize("form", ["16,2", "16,1", "8,2", "8,1"])
@pytest.mark.parametrize("M", [17, 23, 32])
def test_gemm_blk32(M, form, tuning):
    """17..32 rows (batch 9-16): k_gemm_blk32 — column blocks x K ranges, one slab hand-off per block.  All four
    forms (k-tiles per range, strips per wave), the three decode epilogues, strip counts that are not whole
    blocks, twice for bit-reproducibility, tickets re-armed."""
    kr, ws = (int(v) for v in form.split(","))
    tuning("blk32_kr", kr)
    tuning("blk32_ws", ws)
    tuning("gemm_mz_max", 0)
    d = dev()
    torch.manual_seed(100 + M)

    def lend(g, ns, kt):
        blocks = (ns + 8 * ws - 1) // (8 * ws)
        scr = torch.zeros(blocks * (kt // kr) * 512 * 8 * ws, device=d)
        tk = torch.zeros(ns, dtype=torch.int32, device=d)
        g.sk_scratch, g.sk_tickets, g.sk_scratch_floats = hb.ptr(scr), hb.ptr(tk), scr.numel()
        return scr, tk

    # SCALE_STORE with the row norm: K = 2048, 37 strips (a partial last block)
    K, N = 2048, 37 * 16
    x = torch.randn(M, K, device=d) * 2.0
    gw = bf16r(1.0 + 0.1 * torch.randn(K, device=d))
    W = bf16r(torch.randn(K, N, device=d) * 0.05)
    Wt, kt, ns = lay.tile_weight(W)
    A = lay.pack_planes(x * gw)
    ss = strip_ssq(x, 32)
    outs = []
    for _ in range(2):
        out = torch.full((M, N), float("nan"), device=d)
        g = hb.GemmArgs()
        g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], M
        g.W, g.KT, g.nstrips, g.epi = hb.ptr(Wt), kt, ns, hb.EPI_SCALE_STORE
        g.ssq_in, g.ssq_in_n, g.inv_d, g.eps, g.ssq_ld = hb.ptr(ss), K // 16, 1.0 / K, 1e-5, 32
        g.out, g.ldo = hb.ptr(out), N
        scr, tk = lend(g, ns, kt)
        hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
        torch.cuda.synchronize()
        assert (tk == 0).all() and scr.abs().sum().item() > 0        # the split kernel ran and re-armed its tickets
        outs.append(out)
    xd = x.double()
    ref = ((xd * gw.double()) @ W.double()) * torch.rsqrt((xd ** 2).mean(-1, keepdim=True) + 1e-5)
    assert (outs[0].double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    assert torch.equal(outs[0], outs[1])
    # SWIGLU_EMIT: K = 2048, F = 2048 (256 strips)
    F = 2048
    wi = bf16r(torch.randn(K, 2, F, device=d) * 0.05)
    Wt, kt, ns = lay.tile_weight(lay.interleave_gate_up(wi))
    P = torch.zeros(3, 2, F // 32, 64, 8, dtype=torch.bfloat16, device=d)
    g = hb.GemmArgs()
    g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], M
    g.W, g.KT, g.nstrips, g.epi = hb.ptr(Wt), kt, ns, hb.EPI_SWIGLU_EMIT
    g.ssq_in, g.ssq_in_n, g.inv_d, g.eps, g.ssq_ld = hb.ptr(ss), K // 16, 1.0 / K, 1e-5, 32
    g.P, g.p_plane_stride, g.p_ktiles = hb.ptr(P), P[0].numel(), F // 32
    scr, tk = lend(g, ns, kt)
    hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
    torch.cuda.synchronize()
    assert (tk == 0).all()
    h = (xd * torch.rsqrt((xd ** 2).mean(-1, keepdim=True) + 1e-5)) * gw.double()
    f = torch.einsum("mk,kgf->mgf", h, wi.double())
    ref = torch.nn.functional.silu(f[:, 0]) * f[:, 1]
    assert (lay.unpack_planes(P, M, F).double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    # RESID_EMIT: K = 8192 (16 or 32 ranges), D = 2048; and K = 512 into D = 272 (17 strips)
    for K2, D in ((8192, 2048), (512, 272)):
        a = torch.randn(M, K2, device=d)
        W2 = bf16r(torch.randn(K2, D, device=d) * 0.03)
        x0 = torch.randn(M, D, device=d)
        gn = bf16r(1.0 + 0.1 * torch.randn(D, device=d))
        Wt, kt, ns = lay.tile_weight(W2)
        A2 = lay.pack_planes(a)
        pkt = (D + 31) // 32
        res = []
        for _ in range(2):
            xr = x0.clone()
            P = torch.zeros(3, 2, pkt, 64, 8, dtype=torch.bfloat16, device=d)
            ssq = torch.zeros(ns, 32, device=d)
            g = hb.GemmArgs()
            g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A2), A2[0].numel(), A2.shape[2], M
            g.W, g.KT, g.nstrips, g.epi = hb.ptr(Wt), kt, ns, hb.EPI_RESID_EMIT
            g.ssq_ld, g.out, g.ldo, g.gnext = 32, hb.ptr(xr), D, hb.ptr(gn)
            g.P, g.p_plane_stride, g.p_ktiles, g.ssq_out = hb.ptr(P), P[0].numel(), pkt, hb.ptr(ssq)
            scr, tk = lend(g, ns, kt)
            hb.check(hb.lib().dia_gemm(C.byref(g), None), "dia_gemm")
            torch.cuda.synchronize()
            assert (tk == 0).all()
            res.append((xr, P, ssq))
        ref = x0.double() + a.double() @ W2.double()
        assert (res[0][0].double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])
        assert torch.equal(lay.unpack_planes(res[0][1], M, pkt * 32)[:, :D], res[0][0] * gn)
        want = (res[0][0].double() ** 2).reshape(M, D // 16, 16).sum(-1).T
        assert (res[0][2][:, :M].double() - want).abs().max().item() <= 1e-5 * want.max().item()


@pytest.mark.parametrize("M,D,F", [(2, 512, 1024), (1, 512, 1024), (2, 2048, 8192)])
def test_mlp_fused(M, D, F):
    """dia_mlp_fused: wi (SwiGLU) and wo (residual + planes + ssq) in one persistent launch with a grid barrier,
    against float64 and against the two separate launches; repeated launches reuse the barrier counter."""
    d = dev()
    torch.manual_seed(D + M)
    x = torch.randn(M, D, device=d)
    gw = bf16r(1.0 + 0.1 * torch.randn(D, device=d))
    wi = bf16r(torch.randn(D, 2, F, device=d) * 0.05)
    wo = bf16r(torch.randn(F, D, device=d) * 0.03)
    gn = bf16r(1.0 + 0.1 * torch.randn(D, device=d))
    x0 = torch.randn(M, D, device=d)
    Wi, kti, nsi = lay.tile_weight(lay.interleave_gate_up(wi))
    Wo, kto, nso = lay.tile_weight(wo)
    A = lay.pack_planes(x * gw)
    ssq_in = strip_ssq(x, 16)
    L = hb.lib()

    def args(xres, Ph, Px, ssq_out, skscr, sktk):
        a = hb.GemmArgs()
        a.A, a.a_plane_stride, a.a_ktiles, a.M = hb.ptr(A), A[0].numel(), A.shape[2], M
        a.W, a.KT, a.nstrips, a.epi = hb.ptr(Wi), kti, nsi, hb.EPI_SWIGLU_EMIT
        a.ssq_in, a.ssq_in_n, a.inv_d, a.eps, a.ssq_ld = hb.ptr(ssq_in), D // 16, 1.0 / D, 1e-5, 16
        a.P, a.p_plane_stride, a.p_ktiles = hb.ptr(Ph), Ph[0].numel(), F // 32
        b = hb.GemmArgs()
        b.A, b.a_plane_stride, b.a_ktiles, b.M = hb.ptr(Ph), Ph[0].numel(), F // 32, M
        b.W, b.KT, b.nstrips, b.epi = hb.ptr(Wo), kto, nso, hb.EPI_RESID_EMIT
        b.ssq_ld, b.out, b.ldo, b.gnext = 16, hb.ptr(xres), D, hb.ptr(gn)
        b.P, b.p_plane_stride, b.p_ktiles, b.ssq_out = hb.ptr(Px), Px[0].numel(), D // 32, hb.ptr(ssq_out)
        b.sk_scratch, b.sk_tickets, b.sk = hb.ptr(skscr), hb.ptr(sktk), 2
        return a, b

    def fresh():
        return (x0.clone(), torch.zeros(3, 1, F // 32, 64, 8, dtype=torch.bfloat16, device=d),
                torch.zeros(3, 1, D // 32, 64, 8, dtype=torch.bfloat16, device=d), torch.zeros(D // 16, 16, device=d),
                torch.zeros((D // 16) * 4 * 256, device=d), torch.zeros(D // 16, dtype=torch.int32, device=d))

    bar = torch.zeros(2, dtype=torch.int32, device=d)
    outs = []
    for rep in range(3):                                   # the counter keeps counting across launches
        bufs = fresh()
        a, b = args(*bufs)
        hb.check(L.dia_mlp_fused(C.byref(a), C.byref(b), hb.ptr(bar), None), "dia_mlp_fused")
        torch.cuda.synchronize()
        outs.append(bufs)
    assert bar.tolist() == [3 * 2 * nso, 0] and (outs[0][5] == 0).all()
    xd = x.double()
    hdn = (xd * torch.rsqrt((xd ** 2).mean(-1, keepdim=True) + 1e-5)) * gw.double()
    f = torch.einsum("mk,kgf->mgf", hdn, wi.double())
    h_ref = torch.nn.functional.silu(f[:, 0]) * f[:, 1]
    xr, Ph, Px, ssq_o = outs[0][:4]
    h_got = lay.unpack_planes(Ph, M, F)
    assert (h_got.double() - h_ref).abs().max().item() <= 2e-5 * max(1.0, h_ref.abs().max().item())
    ref = x0.double() + h_got.double() @ wo.double()       # wo consumes the fp32 h the planes carry
    assert (xr.double() - ref).abs().max().item() <= 2e-5 * ref.abs().max().item()
    assert torch.equal(lay.unpack_planes(Px, M, D), xr * gn)
    want = (xr.double() ** 2).reshape(M, D // 16, 16).sum(-1).T
    assert (ssq_o[:, :M].double() - want).abs().max().item() <= 1e-5 * want.max().item()
    for o in outs[1:]:
        assert torch.equal(o[0], xr) and torch.equal(o[2], Px)            # bit-reproducible
    # the two separate launches agree to fp32 rounding (their waves split K differently)
    bufs = fresh()
    a, b = args(*bufs)
    hb.check(L.dia_gemm(C.byref(a), None), "wi"); hb.check(L.dia_gemm(C.byref(b), None), "wo")
    torch.cuda.synchronize()
    assert (lay.unpack_planes(bufs[1], M, F) - h_got).abs().max().item() <= 1e-5 * max(1.0, h_ref.abs().max().item())
    assert (bufs[0] - xr).abs().max().item() <= 1e-5 * ref.abs().max().item()
    # shapes that do not chain are refused, nothing is launched
    b.KT = kto // 2
    assert L.dia_mlp_fused(C.byref(a), C.byref(b), hb.ptr(bar), None) == -1
