"""attention microbench: 18 layers of distinct bf16 K/V caches (HBM-cold like a real step), back-to-back launches"""
import ctypes as C, sys, os
sys.path.insert(0, "dia-tts-prune_amd")
import numpy as np, torch
from dia_hip import binding as hb, layout as lay
d = torch.device("cuda:0"); L = hb.lib()
NL = 18
cases = [(1, 1040), (8, 300), (8, 1040), (8, 2500)]
if len(sys.argv) > 1: cases = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for B, cur in cases:
    R, QH, KVH, T = 2 * B, 16, 4, 3072
    nq = (QH + 2 * KVH) * 128
    qkv = torch.randn(R, nq, device=d)
    cos, sin = [t.to(d) for t in lay.rope_tables(T + 1, 128, 1, 10000)]
    curs = torch.full((B,), cur, dtype=torch.int32, device=d)
    P = torch.zeros(3, (R + 15) // 16, QH * 128 // 32, 64, 8, dtype=torch.bfloat16, device=d)
    scr = torch.zeros(L.dia_attn_scratch_floats(R, KVH, T), device=d); tk = torch.zeros(R * KVH, dtype=torch.int32, device=d)
    args, keep = [], []
    for l in range(NL):
        kc = torch.randn(R, KVH, T, 128, device=d).bfloat16(); vc = torch.randn(R, KVH, T, 128, device=d).bfloat16()
        keep.append((kc, vc))
        a = hb.AttnArgs()
        a.mode, a.kv_dtype, a.n_kv_heads, a.group, a.n_rows, a.kv_cap = hb.ATTN_SELF, 1, KVH, 4, R, T
        a.q, a.ldq, a.q_off, a.k_off, a.v_off = hb.ptr(qkv), nq, 0, QH * 128, (QH + KVH) * 128
        a.kc, a.vc, a.cur = hb.ptr(kc), hb.ptr(vc), hb.ptr(curs)
        a.cos_t, a.sin_t = hb.ptr(cos), hb.ptr(sin)
        a.P, a.p_plane_stride, a.p_ktiles = hb.ptr(P), P[0].numel(), P.shape[2]
        a.scratch, a.tickets, a.v_blocked = hb.ptr(scr), hb.ptr(tk), int(os.environ.get("VBLOCKED", "1"))
        args.append(a)
    def sweep(n):
        for _ in range(n):
            for a in args: hb.check(L.dia_attn(C.byref(a), None), "attn")
    sweep(2); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record(); sweep(reps); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (reps * NL)
    mb = R * KVH * cur * 128 * 2 * 2 / 1e6
    print(f"B={B} cur={cur}: {us:7.2f} us/launch   K+V read {mb:6.1f} MB -> {mb / us * 1e-3 * 1e3:7.1f} GB/s", flush=True)
    del keep, args
