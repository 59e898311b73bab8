"""micro-benchmark of dia_gemm on decode shapes: 18 distinct weight buffers (HBM-cold like a real step)"""
import argparse, ctypes as C, sys, os, time
sys.path.insert(0, "dia-tts-prune_amd")
import torch
from dia_hip import binding as hb, layout as lay

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="wi")
ap.add_argument("--nw", type=int, default=0)
ap.add_argument("--M", type=int, default=2)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--prefetch", type=int, default=0)
ap.add_argument("--same", type=int, default=0)
ap.add_argument("--sk", type=int, default=0)
ap.add_argument("--K", type=int, default=0, help="with --N: a custom RESID_EMIT shape instead of --shape")
ap.add_argument("--N", type=int, default=0)
ap.add_argument("--graph", type=int, default=0, help="capture 8 x 18 launches into a graph and replay it (per-launch time without the eager launch rate limit)")
ap.add_argument("--lend", type=int, default=0, help="lend split-K scratch (floats per strip and k-tile/8) so dia_gemm may pick k_gemm_blk32 at 17..32 rows")
ap.add_argument("--f32", type=int, default=0, help="1: A and P as fp32 activation tiles (act_f32)")
ap.add_argument("--sparse", type=float, default=0.0, help="fraction of zero weights -> zero-skipping stream")
a = ap.parse_args()
d = torch.device("cuda:0")
K, N, epi = {"wi": (2048, 16384, hb.EPI_SWIGLU_EMIT), "wo": (8192, 2048, hb.EPI_RESID_EMIT), "o": (2048, 2048, hb.EPI_RESID_EMIT),
             "qkv": (2048, 3072, hb.EPI_SCALE_STORE), "logits": (2048, 9264, hb.EPI_SCALE_STORE),
             "eqkv": (1024, 3072, hb.EPI_SCALE_STORE), "eo": (1024, 1024, hb.EPI_RESID_EMIT), "ewi": (1024, 8192, hb.EPI_SWIGLU_EMIT), "ewo": (4096, 1024, hb.EPI_RESID_EMIT),
             "qkvp": (1024, 3072, hb.EPI_SCALE_STORE), "op": (1024, 2048, hb.EPI_RESID_EMIT), "wip": (1024, 8192, hb.EPI_SWIGLU_EMIT), "wop": (4096, 2048, hb.EPI_RESID_EMIT)}[a.shape]
if a.K and a.N:
    K, N, epi = a.K, a.N, hb.EPI_RESID_EMIT
M = a.M
mpad = (M + 15) // 16 * 16
Ws = [torch.randint(-30000, 30000, (N // 16, K // 32, 64, 8), dtype=torch.int16, device=d).view(torch.bfloat16) for _ in range(18)]
SP = None
if a.sparse > 0:
    SP = []
    for Wd in Ws:
        Wd.view(torch.int16)[torch.rand(Wd.shape, device=d) < a.sparse] = 0
        SP.append(lay.sparse_tile_weight(Wd))
    print("stream bytes", SP[0][0].numel(), "dense", Ws[0].numel() * 2)
x = torch.randn(M, K, device=d)
A = lay.pack_planes(x)
if a.f32:          # fp32 tiles live in a buffer sized for three planes, like in the engine
    A.view(torch.float32).reshape(-1)[: A[0].numel()] = lay.pack_f32_tiles(x).reshape(-1)
ssq = torch.ones(K // 16, mpad, device=d)
out = torch.zeros(mpad, max(N, 16), device=d)
P = torch.zeros(3, mpad // 16, max(N // 32, 1) if epi != hb.EPI_SWIGLU_EMIT else N // 64, 64, 8, dtype=torch.bfloat16, device=d)
ssq_out = torch.zeros(N // 16, mpad, device=d)
gn = torch.ones(N, device=d)
L = hb.lib()
hb.get_tuning("act_f32")          # (reads DIA_TUNE: nothing else in this script initialises the tuning table)
st = torch.cuda.Stream()
skscr = torch.zeros(2 * (N // 16) * 8 * 256, device=d); sktk = torch.zeros(2 * (N // 16), dtype=torch.int32, device=d)
lendscr = torch.zeros((N // 16) * (K // 256) * 512, device=d) if a.lend else None
def launch(W):
    g = hb.GemmArgs()
    g.A, g.a_plane_stride, g.a_ktiles, g.M = hb.ptr(A), A[0].numel(), A.shape[2], M
    g.W, g.KT, g.nstrips, g.epi, g.nw = hb.ptr(W), K // 32, N // 16, epi, a.nw
    g.act_f32 = a.f32
    if SP is not None:
        i = next(j for j, x_ in enumerate(Ws) if x_ is W)
        g.W, g.sp_blocks, g.sp_toff = None, hb.ptr(SP[i][0]), hb.ptr(SP[i][1])
    if epi != hb.EPI_RESID_EMIT:
        g.ssq_in, g.ssq_in_n, g.inv_d, g.eps = hb.ptr(ssq), K // 16, 1.0 / K, 1e-5
    g.ssq_ld = mpad
    g.out, g.ldo, g.gnext = hb.ptr(out), out.shape[1], hb.ptr(gn)
    g.P, g.p_plane_stride, g.p_ktiles, g.ssq_out = hb.ptr(P), P[0].numel(), P.shape[2], hb.ptr(ssq_out)
    if a.sk > 1:
        g.sk_scratch, g.sk_tickets, g.sk, g.sk_scratch_floats = hb.ptr(skscr), hb.ptr(sktk), a.sk, skscr.numel()
    elif a.lend:
        g.sk_scratch, g.sk_tickets, g.sk_scratch_floats = hb.ptr(lendscr), hb.ptr(sktk), lendscr.numel()
    hb.check(L.dia_gemm(C.byref(g), C.c_void_p(st.cuda_stream)), "gemm")
if a.same: Ws = Ws[:1] * 18
for W in Ws: launch(W)
torch.cuda.synchronize()
if a.graph:
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g_, stream=st):
            for _ in range(8):
                for W in Ws: launch(W)
        for _ in range(3): g_.replay()
        st.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        for _ in range(a.reps): g_.replay()
        e1.record(st); st.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (a.reps * 8 * 18)
    print(f"{a.shape} M={M} nw={a.nw} graph: {us:.2f} us/launch, {K*N*2/us/1e3:.0f} GB/s")
    sys.exit(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(st)
for _ in range(a.reps):
    for W in Ws:
        if a.prefetch:
            hb.check(L.dia_prefetch(hb.ptr(W), W.numel() * 2, a.prefetch, C.c_void_p(st.cuda_stream)), "prefetch")
        launch(W)
e1.record(st); st.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (a.reps * 18)
print(f"{a.shape} M={M} nw={a.nw}: {us:.2f} us/launch (back-to-back incl. gaps), {K*N*2/us/1e3:.0f} GB/s, {2*M*K*N/us/1e6:.1f} TFLOP/s algorithmic (x3 planes on the MFMA pipe)")
