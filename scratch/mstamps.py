import ctypes as C, sys
sys.path.insert(0, "dia-tts-prune_amd")
import numpy as np, torch
from dia_hip import binding as hb, layout as lay
d = torch.device("cuda:0"); L = hb.lib()
L.dia_dbg_stamps.argtypes = [C.c_void_p, C.c_int]
M, D, F = 2, 2048, 8192
mk = lambda *s: torch.randint(-30000, 30000, s, dtype=torch.int16, device=d).view(torch.bfloat16)
Wis = [mk(2 * F // 16, D // 32, 64, 8) for _ in range(4)]; Wos = [mk(D // 16, F // 32, 64, 8) for _ in range(4)]
A = lay.pack_planes(torch.randn(M, D, device=d)); ssq = torch.ones(D // 16, 16, device=d); gn = torch.ones(D, device=d)
x = torch.zeros(16, D, device=d)
Ph = torch.zeros(3, 1, F // 32, 64, 8, dtype=torch.bfloat16, device=d); Px = torch.zeros(3, 1, D // 32, 64, 8, dtype=torch.bfloat16, device=d)
so = torch.zeros(D // 16, 16, device=d); skscr = torch.zeros((D // 16) * 4 * 256, device=d); sktk = torch.zeros(D // 16, dtype=torch.int32, device=d)
bar = torch.zeros(2, dtype=torch.int32, device=d)
def launch(i):
    a = hb.GemmArgs()
    a.A, a.a_plane_stride, a.a_ktiles, a.M = hb.ptr(A), A[0].numel(), A.shape[2], M
    a.W, a.KT, a.nstrips, a.epi = hb.ptr(Wis[i]), D // 32, 2 * F // 16, hb.EPI_SWIGLU_EMIT
    a.ssq_in, a.ssq_in_n, a.inv_d, a.eps, a.ssq_ld = hb.ptr(ssq), D // 16, 1.0 / D, 1e-5, 16
    a.P, a.p_plane_stride, a.p_ktiles = hb.ptr(Ph), Ph[0].numel(), F // 32
    b = hb.GemmArgs()
    b.A, b.a_plane_stride, b.a_ktiles, b.M = hb.ptr(Ph), Ph[0].numel(), F // 32, M
    b.W, b.KT, b.nstrips, b.epi = hb.ptr(Wos[i]), F // 32, D // 16, hb.EPI_RESID_EMIT
    b.ssq_ld, b.out, b.ldo, b.gnext = 16, hb.ptr(x), D, hb.ptr(gn)
    b.P, b.p_plane_stride, b.p_ktiles, b.ssq_out = hb.ptr(Px), Px[0].numel(), D // 32, hb.ptr(so)
    b.sk_scratch, b.sk_tickets, b.sk = hb.ptr(skscr), hb.ptr(sktk), 2
    hb.check(L.dia_mlp_fused(C.byref(a), C.byref(b), hb.ptr(bar), None), "mlp")
for i in range(3): launch(i)
torch.cuda.synchronize(); launch(3); torch.cuda.synchronize()
buf = np.zeros(4096 * 8, dtype=np.int64)
assert L.dia_dbg_stamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
st = buf.reshape(4096, 8)[:256, :6].astype(np.float64)
t0 = st[:, 0].min()
for i, n in enumerate(["start", "phase 1 done", "b2 issued (stores acked)", "barrier passed", "A2 staged", "end"]):
    c = (st[:, i][st[:, i] > 0] - t0) / 100
    print(f"   {n:26s} n {len(c)} min {c.min():6.2f}  median {np.median(c):6.2f}  max {c.max():6.2f} us")
print("bar", bar.tolist())
