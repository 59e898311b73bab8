import ctypes as C, sys, os
sys.path.insert(0, "dia-tts-prune_amd")
import numpy as np, torch
from dia_hip import binding as hb, layout as lay
d = torch.device("cuda:0"); L = hb.lib()
L.dia_dbg_stamps.argtypes = [C.c_void_p, C.c_int]
for shape, K, N, epi in (("o", 2048, 2048, hb.EPI_RESID_EMIT), ("wi", 2048, 16384, hb.EPI_SWIGLU_EMIT), ("wo", 8192, 2048, hb.EPI_RESID_EMIT)):
    M, mpad = 2, 16
    Ws = [torch.randint(-30000, 30000, (N // 16, K // 32, 64, 8), dtype=torch.int16, device=d).view(torch.bfloat16) for _ in range(4)]
    A = lay.pack_planes(torch.randn(M, K, device=d))
    ssq = torch.ones(K // 16, mpad, device=d); out = torch.zeros(mpad, N, device=d)
    P = torch.zeros(3, 1, max(N // 32, 1), 64, 8, dtype=torch.bfloat16, device=d); ssq_out = torch.zeros(N // 16, mpad, device=d); gn = torch.ones(N, device=d)
    def launch(W):
        g = hb.GemmArgs()
        g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], M
        g.W, g.KT, g.nstrips, g.epi = hb.ptr(W), K // 32, N // 16, epi
        if epi != hb.EPI_RESID_EMIT: g.ssq_in, g.ssq_in_n, g.inv_d, g.eps = hb.ptr(ssq), K // 16, 1.0 / K, 1e-5
        g.ssq_ld = mpad; g.out, g.ldo, g.gnext = hb.ptr(out), N, hb.ptr(gn)
        g.P, g.p_plane_stride, g.p_ktiles, g.ssq_out = hb.ptr(P), P[0].numel(), P.shape[2], hb.ptr(ssq_out)
        hb.check(L.dia_gemm(C.byref(g), None), "gemm")
    for W in Ws[:3]: launch(W)
    torch.cuda.synchronize(); launch(Ws[3]); torch.cuda.synchronize()
    nb = 4096
    buf = np.zeros(nb * 8, dtype=np.int64)
    assert L.dia_dbg_stamps(buf.ctypes.data_as(C.c_void_p), nb * 8) == 0
    st = buf.reshape(nb, 8)[:, :6]
    grid = (N // 16) if N < 16384 else (N // 16) // 4
    st = st[:grid].astype(np.float64)
    t0 = st[:, 0].min()
    us = (st - t0) / 100.0      # 100 MHz wall clock -> us
    names = ["start", "B issued", "A staged+barrier", "MFMA done(last strip)", "reduced", "end"]
    print(shape, "grid", grid)
    for i, n in enumerate(names):
        c = us[:, i]
        print(f"   {n:24s} min {c.min():6.2f}  median {np.median(c):6.2f}  max {c.max():6.2f} us")
