"""in-chain phase timeline of the 16-row GEMM (k_gemm16) on decode shapes: o, wi, wo (split-K 4)"""
import ctypes as C, sys, os
sys.path.insert(0, "dia-tts-prune_amd")
import numpy as np, torch
from dia_hip import binding as hb, layout as lay
d = torch.device("cuda:0"); L = hb.lib()
L.dia_dbg_stamps.argtypes = [C.c_void_p, C.c_int]
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16
mpad = (M + 15) // 16 * 16
for shape, K, N, epi, sk in (("o", 2048, 2048, hb.EPI_RESID_EMIT, 0), ("qkv", 2048, 3072, hb.EPI_SCALE_STORE, 0)):
    Ws = [torch.randint(-30000, 30000, (N // 16, K // 32, 64, 8), dtype=torch.int16, device=d).view(torch.bfloat16) for _ in range(6)]
    A = lay.pack_planes(torch.randn(M, K, device=d))
    ssq = torch.ones(K // 16, mpad, device=d); out = torch.zeros(mpad, N, device=d)
    P = torch.zeros(3, mpad // 16, max(N // 32, 1), 64, 8, dtype=torch.bfloat16, device=d); ssq_out = torch.zeros(N // 16, mpad, device=d); gn = torch.ones(N, device=d)
    scr = torch.zeros((N // 16) * 8 * 256, device=d); tk = torch.zeros(N // 16, dtype=torch.int32, device=d)
    def launch(W):
        g = hb.GemmArgs()
        g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], M
        g.W, g.KT, g.nstrips, g.epi = hb.ptr(W), K // 32, N // 16, epi
        if epi != hb.EPI_RESID_EMIT: g.ssq_in, g.ssq_in_n, g.inv_d, g.eps = hb.ptr(ssq), K // 16, 1.0 / K, 1e-5
        g.ssq_ld = mpad; g.out, g.ldo, g.gnext = hb.ptr(out), N, hb.ptr(gn)
        g.P, g.p_plane_stride, g.p_ktiles, g.ssq_out = hb.ptr(P), P[0].numel(), P.shape[2], hb.ptr(ssq_out)
        if sk: g.sk, g.sk_scratch, g.sk_tickets = sk, hb.ptr(scr), hb.ptr(tk)
        hb.check(L.dia_gemm(C.byref(g), None), "gemm")
    for rep in range(2):
        torch.cuda.synchronize()
        for i in range(24): launch(Ws[i % 6])
        torch.cuda.synchronize()
        buf = np.zeros(4096 * 8, dtype=np.int64)
        assert L.dia_dbg_stamps(buf.ctypes.data_as(C.c_void_p), 4096 * 8) == 0
        st = buf.reshape(4096, 8)[:, :6].astype(np.float64)
        st = st[st[:, 0] > 0]
        st = st[st[:, 0] > st[:, 0].max() - 5000]            # the last launch only
        t0 = st[:, 0].min(); us = (st - t0) / 100.0
        f = lambda c: f"{np.median(us[:, c]):5.2f}/{us[:, c].max():5.2f}"
        print(f"{shape} M={M} sk={sk} rep{rep} WGs {len(st)}: start {f(0)} loads issued {f(1)} MFMA done {f(2)} " + (f"reduced {f(3)} combined {f(4)} " if sk else f"summed {f(3)} staged {f(4)} ") + f"end {f(5)}  (median/max us)")
