"""cross-attention microbench + stamps: 18 layers of distinct bf16 K/V caches"""
import ctypes as C, sys, os
sys.path.insert(0, "dia-tts-prune_amd")
import numpy as np, torch
from dia_hip import binding as hb, layout as lay
d = torch.device("cuda:0"); L = hb.lib()
NL = 18
stamps = hasattr(L, "dia_dbg_astamps") and os.environ.get("STAMPS")
names = ["start", "cur known", "q in LDS", "key loop done", "wave merge barrier", "ticket known", "end"]
for lens in ([75], [32, 64, 96, 128, 192, 256, 384, 512], [512] * 8, [64] * 8):
    B, H, S = len(lens), 16, 512
    R = 2 * B
    qc = torch.randn(R, H * 128, device=d)
    cos, sin = [t.to(d) for t in lay.rope_tables(3073, 128, 1, 10000)]
    curs = torch.full((B,), 500, dtype=torch.int32, device=d)
    lt = torch.tensor(lens, dtype=torch.int32, device=d)
    P = torch.zeros(3, (R + 15) // 16, H * 128 // 32, 64, 8, dtype=torch.bfloat16, device=d)
    scr = torch.zeros(max(1, L.dia_attn_scratch_floats(B, H, S)), device=d); tk = torch.zeros(B * H, dtype=torch.int32, device=d)
    args, keep = [], []
    for l in range(NL):
        kc = torch.randn(B, H, S, 128, device=d).bfloat16(); vc = torch.randn(B, H, S, 128, device=d).bfloat16()
        keep.append((kc, vc))
        a = hb.AttnArgs()
        a.mode, a.kv_dtype, a.n_kv_heads, a.group, a.n_rows, a.kv_cap = hb.ATTN_CROSS, 1, H, 1, B, S
        a.q, a.ldq = hb.ptr(qc), H * 128
        a.kc, a.vc, a.cur, a.len = hb.ptr(kc), hb.ptr(vc), hb.ptr(curs), hb.ptr(lt)
        a.cos_t, a.sin_t = hb.ptr(cos), hb.ptr(sin)
        a.P, a.p_plane_stride, a.p_ktiles = hb.ptr(P), P[0].numel(), P.shape[2]
        a.scratch, a.tickets, a.v_blocked = hb.ptr(scr), hb.ptr(tk), 1
        args.append(a)
    def sweep(n):
        for _ in range(n):
            for a in args: hb.check(L.dia_attn(C.byref(a), None), "attn")
    sweep(2); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record(); sweep(reps); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * NL)
    print(f"cross B={B} lens={lens}: {us:7.2f} us/launch", flush=True)
    if stamps:
        L.dia_dbg_astamps.argtypes = [C.c_void_p, C.c_int]
        L.dia_dbg_aclear(); hb.check(L.dia_attn(C.byref(args[0]), None), "attn"); torch.cuda.synchronize()
        buf = np.zeros(8192 * 8, dtype=np.int64); L.dia_dbg_astamps(buf.ctypes.data_as(C.c_void_p), buf.size)
        st = buf.reshape(8192, 8)[:, :7].astype(np.float64)
        t0 = st[:, 0][st[:, 0] > 0].min()
        for i, n in enumerate(names):
            c = st[:, i]; c = c[c > 0]
            if len(c): print(f"   {n:20s} n {len(c):5d} min {(c.min()-t0)/100:6.2f}  median {(np.median(c)-t0)/100:6.2f}  max {(c.max()-t0)/100:6.2f} us")
    del keep, args
