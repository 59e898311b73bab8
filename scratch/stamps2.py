"""workgroup start spread / phase timeline of the o GEMV (K = N = 2048, M = 2) for several waves-per-workgroup choices"""
import ctypes as C, sys, os
sys.path.insert(0, "dia-tts-prune_amd")
import numpy as np, torch
from dia_hip import binding as hb, layout as lay
d = torch.device("cuda:0"); L = hb.lib()
L.dia_dbg_stamps.argtypes = [C.c_void_p, C.c_int]
K, N, epi, M, mpad = 2048, 2048, hb.EPI_RESID_EMIT, 2, 16
Ws = [torch.randint(-30000, 30000, (N // 16, K // 32, 64, 8), dtype=torch.int16, device=d).view(torch.bfloat16) for _ in range(4)]
A = lay.pack_planes(torch.randn(M, K, device=d))
out = torch.zeros(mpad, N, device=d)
P = torch.zeros(3, 1, N // 32, 64, 8, dtype=torch.bfloat16, device=d); ssq_out = torch.zeros(N // 16, mpad, device=d); gn = torch.ones(N, device=d)
for nw in (0,):
    def launch(W):
        g = hb.GemmArgs()
        g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], M
        g.W, g.KT, g.nstrips, g.epi, g.nw = hb.ptr(W), K // 32, N // 16, epi, nw
        g.ssq_ld = mpad; g.out, g.ldo, g.gnext = hb.ptr(out), N, hb.ptr(gn)
        g.P, g.p_plane_stride, g.p_ktiles, g.ssq_out = hb.ptr(P), P[0].numel(), P.shape[2], hb.ptr(ssq_out)
        hb.check(L.dia_gemm(C.byref(g), None), "gemm")
    for rep in range(3):
        torch.cuda.synchronize()
        for i in range(24): launch(Ws[i % 4])          # queued back to back: the stamps left are those of the LAST launch, in-chain
        torch.cuda.synchronize()
        buf = np.zeros(4096 * 8, dtype=np.int64)
        assert L.dia_dbg_stamps(buf.ctypes.data_as(C.c_void_p), 4096 * 8) == 0
        st = buf.reshape(4096, 8)[:128, :6].astype(np.float64)
        us = (st - st[:, 0].min()) / 100.0
        print(f"nw={nw} rep{rep}: start spread {us[:,0].max():.2f} | per-WG medians: B issued +{np.median(us[:,1]-us[:,0]):.2f}, A staged +{np.median(us[:,2]-us[:,1]):.2f}, MFMA +{np.median(us[:,3]-us[:,2]):.2f}, reduce +{np.median(us[:,4]-us[:,3]):.2f}, end +{np.median(us[:,5]-us[:,4]):.2f} | kernel {us[:,5].max():.2f} us")
