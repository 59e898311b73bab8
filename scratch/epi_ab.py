"""dump the outputs of the M <= 4 GEMV epilogues for fixed inputs (run once per library build, compare the files)"""
import ctypes as C, sys
sys.path.insert(0, "dia-tts-prune_amd")
import numpy as np, torch
from dia_hip import binding as hb, layout as lay
d = torch.device("cuda:0"); L = hb.lib()
res = {}
for name, K, N, epi, sk in (("o", 2048, 2048, hb.EPI_RESID_EMIT, 0), ("wo", 8192, 2048, hb.EPI_RESID_EMIT, 2), ("wi", 2048, 16384, hb.EPI_SWIGLU_EMIT, 0),
                            ("small", 512, 272, hb.EPI_RESID_EMIT, 0), ("wis", 1024, 1024, hb.EPI_SWIGLU_EMIT, 0)):
    for M in (1, 2, 3, 4, 7, 16, 23):
        g_ = torch.Generator(device=d).manual_seed(1000 + M)
        W = (torch.randn(K, N, device=d, generator=g_) * 0.05).bfloat16().float()
        Wt, kt, ns = lay.tile_weight(W)
        x = torch.randn(M, K, device=d, generator=g_) * 2
        A = lay.pack_planes(x)
        mp = (M + 15) // 16 * 16
        ssq = torch.zeros(K // 16, mp, device=d); ssq[:, :M] = (x.double() ** 2).reshape(M, K // 16, 16).sum(-1).T.float()
        out = torch.zeros(mp, N, device=d); out[:M] = torch.randn(M, N, device=d, generator=g_)
        nP = N // 2 if epi == hb.EPI_SWIGLU_EMIT else N
        P = torch.zeros(3, mp // 16, (nP + 31) // 32, 64, 8, dtype=torch.bfloat16, device=d)
        sso = torch.zeros(ns, mp, device=d)
        gn = (1.0 + 0.1 * torch.randn(N, device=d, generator=g_)).bfloat16().float()
        scr = torch.zeros(2 * ns * 4 * 256, device=d); tk = torch.zeros(2 * ns, dtype=torch.int32, device=d)
        g = hb.GemmArgs()
        g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], M
        g.W, g.KT, g.nstrips, g.epi = hb.ptr(Wt), kt, ns, epi
        if epi != hb.EPI_RESID_EMIT: g.ssq_in, g.ssq_in_n, g.inv_d, g.eps = hb.ptr(ssq), K // 16, 1.0 / K, 1e-5
        g.ssq_ld = mp; g.out, g.ldo, g.gnext = hb.ptr(out), N, hb.ptr(gn)
        g.P, g.p_plane_stride, g.p_ktiles, g.ssq_out = hb.ptr(P), P[0].numel(), P.shape[2], hb.ptr(sso)
        if sk: g.sk, g.sk_scratch, g.sk_tickets, g.sk_scratch_floats = (sk if M <= 4 else 4), hb.ptr(scr), hb.ptr(tk), scr.numel()
        hb.check(L.dia_gemm(C.byref(g), None), "gemm"); torch.cuda.synchronize()
        res[f"{name}_{M}_out"] = out.cpu().numpy(); res[f"{name}_{M}_P"] = P.view(torch.int16).cpu().numpy(); res[f"{name}_{M}_ssq"] = sso.cpu().numpy()
np.savez(sys.argv[1], **res)
print("saved", sys.argv[1], len(res))
