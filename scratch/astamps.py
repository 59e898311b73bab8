import ctypes as C, sys, os
sys.path.insert(0, "dia-tts-prune_amd")
import numpy as np, torch
from dia_hip import binding as hb, layout as lay
d = torch.device("cuda:0"); L = hb.lib()
L.dia_dbg_astamps.argtypes = [C.c_void_p, C.c_int]
names = ["start", "cur known", "q in LDS", "key loop done", "wave merge barrier", "ticket known", "end"]
def report(tag, nwg):
    buf = np.zeros(8192 * 8, dtype=np.int64)
    assert L.dia_dbg_astamps(buf.ctypes.data_as(C.c_void_p), buf.size) == 0
    st = buf.reshape(8192, 8)[:nwg, :7].astype(np.float64)
    live = st[:, 1] > 0
    t0 = st[:, 0][st[:, 0] > 0].min()
    print(tag, "workgroups", nwg, "live", int(live.sum()))
    for i, n in enumerate(names):
        c = st[:, i]; c = c[c > 0]
        if i == 3 and len(c): print("      key loop done percentiles", [round((v - t0) / 100, 2) for v in np.percentile(c, [10, 25, 50, 75, 90, 99])])
        if len(c): print(f"   {n:20s} n {len(c):5d} min {(c.min()-t0)/100:6.2f}  median {(np.median(c)-t0)/100:6.2f}  max {(c.max()-t0)/100:6.2f} us")
def clear():
    pass
for B, cur, T in ((1, 1040, 3072), (1, 300, 3072), (8, 1040, 3072), (8, 300, 3072)):
    R, QH, KVH = 2 * B, 16, 4
    nq = (QH + 2 * KVH) * 128
    qkv = torch.randn(R, nq, device=d)
    kc = torch.randn(R, KVH, T, 128, device=d).bfloat16(); vc = torch.randn(R, KVH, T, 128, device=d).bfloat16()
    cos, sin = [t.to(d) for t in lay.rope_tables(T + 1, 128, 1, 10000)]
    curs = torch.full((B,), cur, dtype=torch.int32, device=d)
    P = torch.zeros(3, (R + 15) // 16, QH * 128 // 32, 64, 8, dtype=torch.bfloat16, device=d)
    a = hb.AttnArgs()
    a.mode, a.kv_dtype, a.n_kv_heads, a.group, a.n_rows, a.kv_cap = hb.ATTN_SELF, 1, KVH, 4, R, T
    a.q, a.ldq, a.q_off, a.k_off, a.v_off = hb.ptr(qkv), nq, 0, QH * 128, (QH + KVH) * 128
    a.kc, a.vc, a.cur = hb.ptr(kc), hb.ptr(vc), hb.ptr(curs)
    a.cos_t, a.sin_t = hb.ptr(cos), hb.ptr(sin)
    a.P, a.p_plane_stride, a.p_ktiles = hb.ptr(P), P[0].numel(), P.shape[2]
    scr = torch.zeros(L.dia_attn_scratch_floats(R, KVH, T), device=d); tk = torch.zeros(R * KVH, dtype=torch.int32, device=d)
    a.scratch, a.tickets, a.v_blocked = hb.ptr(scr), hb.ptr(tk), 1
    nz = min((512 + R * KVH - 1) // (R * KVH), T // 128)
    if hb.lib().dia_get_tuning(b"attn_nz") > 0: nz = hb.lib().dia_get_tuning(b"attn_nz")      # DIA_TUNE=attn_nz=N
    big = torch.empty(1 << 28, dtype=torch.uint8, device=d)
    for _ in range(3):
        hb.check(L.dia_attn(C.byref(a), None), "attn")
    torch.cuda.synchronize()
    L.dia_dbg_aclear()
    big.fill_(1)          # push K/V out of L2 / MALL like the weight stream does in a real step; NO sync after it: the
    hb.check(L.dia_attn(C.byref(a), None), "attn"); torch.cuda.synchronize()      # stamped launch starts in-chain (an isolated launch starts its XCDs up to 1.4 us apart)
    report(f"self B={B} cur={cur} T={T} nz={nz}", KVH * R * nz)
