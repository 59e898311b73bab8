import ctypes as C, sys
sys.path.insert(0, "dia-tts-prune_amd")
import torch
from dia_hip import binding as hb, layout as lay
d = torch.device("cuda:0")
torch.manual_seed(5)
Lq, H = 45, 4
nq = 3 * H * 128
qkv = torch.randn(Lq, nq, device=d)
cos, sin = [t.to(d) for t in lay.rope_tables(64, 128, 1, 10000)]
cap = 48
kc = torch.zeros(H, cap, 128, device=d); vc = torch.zeros(H, cap, 128, device=d)
Lb = hb.lib()
hb.check(Lb.dia_enc_kv_prep(hb.ptr(qkv), nq, H * 128, 2 * H * 128, H, Lq, cap, hb.ptr(cos), hb.ptr(sin), hb.ptr(kc), hb.ptr(vc), None), "prep")
P = torch.zeros(3, 3, H * 128 // 32, 64, 8, dtype=torch.bfloat16, device=d)
a = hb.AttnArgs()
a.mode, a.kv_dtype, a.n_kv_heads, a.group, a.n_rows, a.kv_cap = hb.ATTN_ENC, 0, H, 1, Lq, cap
a.q, a.ldq, a.enc_len = hb.ptr(qkv), nq, Lq
a.kc, a.vc = hb.ptr(kc), hb.ptr(vc)
a.cos_t, a.sin_t = hb.ptr(cos), hb.ptr(sin)
a.P, a.p_plane_stride, a.p_ktiles = hb.ptr(P), P[0].numel(), P.shape[2]
hb.check(Lb.dia_attn(C.byref(a), None), "dia_attn")
torch.cuda.synchronize()
out = lay.unpack_planes(P, Lq, H * 128)
print("rows with nonzero output:", torch.nonzero(out.abs().sum(1) > 0).flatten().tolist())
