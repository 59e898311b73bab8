"""Kernel time (dia_gemm_timed: dispatch-level start / stop) of the short-prompt prefill's K = 1024 GEMMs: planes in, M rows,
SCALE_STORE into fp32 (qkv-like, N = 6144) and SWIGLU_EMIT into planes (wi-like, N = 8192), median of 20 launches."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dia-tts-prune_amd"))
import torch
from dia_hip import binding as hb, layout as lay
L = hb.lib(); d = torch.device("cuda:0")
torch.manual_seed(0)
def tile(w): return lay.tile_weight(w.to(d))[0]
def run(M, K, N, epi, label, reps=20):
    X = torch.randn(M, K, device=d)
    A = lay.pack_planes(X)
    Wt = tile(torch.randn(K, N) * 0.02)
    mp = (M + 15) // 16 * 16
    ssq = torch.ones(K // 16, mp, device=d)
    g = hb.GemmArgs()
    g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], M
    g.W, g.KT, g.nstrips, g.epi = hb.ptr(Wt), K // 32, N // 16, epi
    g.ssq_in, g.ssq_in_n, g.inv_d, g.eps, g.ssq_ld = hb.ptr(ssq), K // 16, 1.0 / K, 1e-5, mp
    keep = []
    if epi == hb.EPI_SCALE_STORE:
        out = torch.zeros(mp, N, device=d); g.out, g.ldo = hb.ptr(out), N; keep.append(out)
    else:
        P = torch.zeros(3, mp // 16, N // 2 // 32, 64, 8, dtype=torch.bfloat16, device=d)
        g.P, g.p_plane_stride, g.p_ktiles = hb.ptr(P), P[0].numel(), N // 2 // 32; keep.append(P)
    ts = []
    ms = C.c_float()
    for _ in range(reps):
        hb.check(L.dia_gemm_timed(C.byref(g), None, C.byref(ms)), "dia_gemm_timed")
        ts.append(ms.value * 1e3)
    ts.sort()
    print(f"{label:28s} M {M:4d} K {K} N {N}: median {ts[len(ts)//2]:7.2f} us  min {ts[0]:7.2f}  ({K * N * 2 / 1e6:.1f} MB of weights)")
for M in (int(a) for a in (sys.argv[1:] or ["128"])):
    run(M, 1024, 6144, hb.EPI_SCALE_STORE, "qkv-like SCALE_STORE")
    run(M, 1024, 8192, hb.EPI_SWIGLU_EMIT, "wi-like SWIGLU_EMIT")
    run(M, 1024, 73728, hb.EPI_SCALE_STORE, "18 x ckv-sized SCALE_STORE")
