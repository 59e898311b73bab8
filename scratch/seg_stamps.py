"""In-kernel timeline of the persistent MLP segment: the sync wave of every workgroup stamps the wall clock (100 MHz) at
each phase; 18 different layers' worth of weights are cycled (HBM-cold like a real step), the last of 24 back-to-back
launches is reported.  Usage: python scratch/seg_stamps.py [M]"""
import ctypes as C, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dia-tts-prune_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from dia_hip import binding as hb, layout as lay
import test_gpu_seg as T

M = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device("cuda:0")
L = hb.lib()
W, a, x0, g_mlp, g_next = T.make_case(M, 1, True, dev)
rings = [lay.seg_ring(W["co"], W["gate"], W["up"], W["wo"], W["qkv"]) for _ in range(1)]
rings += [rings[0].clone() for _ in range(11)]                      # 12 x 121.6 MB > the 256 MiB Infinity Cache
a_t = lay.pack_f32_tiles(a, ktiles=64, mtiles=1)
x = torch.zeros(16, 2048, device=dev); x[:M] = x0
planes_x = torch.zeros(3, 1, 64, 64, 8, dtype=torch.bfloat16, device=dev)
ssq = torch.zeros(128, 16, device=dev); qkv = torch.zeros(16, 3072, device=dev)
ws = torch.zeros(int(L.dia_seg_workspace_bytes()), dtype=torch.uint8, device=dev)
stamps = torch.zeros(256, 16, dtype=torch.int64, device=dev)
def args(ring, st):
    s = hb.SegArgs()
    s.a_in, s.a_ktiles, s.M, s.W, s.nslots, s.has_qkv, s.D, s.F = hb.ptr(a_t), 64, M, hb.ptr(ring), 29, 1, 2048, 8192
    s.x, s.ldx, s.g_mlp, s.g_next, s.qkv_out, s.ldq = hb.ptr(x), 2048, hb.ptr(g_mlp), hb.ptr(g_next), hb.ptr(qkv), 3072
    s.planes_x, s.xkt, s.ssq, s.ssq_ld, s.eps, s.ws = hb.ptr(planes_x), 64, hb.ptr(ssq), 16, 1e-5, hb.ptr(ws)
    s.stamps = hb.ptr(st) if st is not None else None
    return s
# an idle GPU sits at its low clocks: ~0.3 s of load first (the clock ramp)
big = torch.randn(8192, 8192, device=dev)
for _ in range(40):
    big = (big @ big) * 1e-4
torch.cuda.synchronize()
for i in range(3000):
    hb.check(L.dia_seg_mlp(C.byref(args(rings[i % 12], None)), None), "seg")
torch.cuda.synchronize()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(3):
    ev0.record()
    for i in range(24):
        hb.check(L.dia_seg_mlp(C.byref(args(rings[i % 12], stamps if i == 23 else None)), None), "seg")
    ev1.record(); torch.cuda.synchronize()
    print(f"24 eager launches: {ev0.elapsed_time(ev1) / 24 * 1e3:.2f} us per launch (host-paced)")
assert L.dia_seg_error(hb.ptr(ws), None) == 0
st = stamps.cpu().numpy().astype(np.float64) / 100.0      # us
t0 = st[:, 0].min()
names = ["start", "attn rows in LDS", "planes(co) ready", "co computed", "x1 published", "x1 complete (all CUs)", "x1 gathered",
         "wi computed", "h published", "h quarter complete", "wo computed", "wo partial out / reduced+x2 published", "x2 complete", "x2 gathered",
         "qkv computed", "end"]
print(f"{'phase':42s} {'min':>7s} {'median':>7s} {'max':>7s}   (us after the first workgroup's start; reducers = CUs 0..63)")
for i, nme in enumerate(names):
    v = st[:, i] - t0
    print(f"{i:2d} {nme:39s} {v.min():7.2f} {np.median(v):7.2f} {v.max():7.2f}")

