import sys, os
sys.path.insert(0, "dia-tts-prune_amd"); sys.path.insert(0, ".")
import torch
from tests.test_gpu_kernels import *
d = dev()
M, K, D = 2, 2048, 2048
torch.manual_seed(K + D)
a = torch.randn(M, K, device=d)
W = bf16r(torch.randn(K, D, device=d) * 0.03)
x0 = torch.randn(M, D, device=d)
gn = bf16r(1.0 + 0.1 * torch.randn(D, device=d))
Wt, kt, ns = lay.tile_weight(W)
mpad = 16
x = x0.clone()
P = torch.zeros(3, 1, D // 32, 64, 8, dtype=torch.bfloat16, device=d)
ssq = torch.zeros(ns, mpad, device=d)
run_gemm(a, Wt, kt, ns, hb.EPI_RESID_EMIT, out=x, ldo=D, gnext=gn, P=P, p_kt=P.shape[2], ssq_out=ssq, ssq_ld=mpad)
got = lay.unpack_planes(P, M, D); want = x * gn
diff = (got - want).abs()
print("max diff", diff.max().item(), "n bad", (diff > 0).sum().item(), "of", diff.numel())
idx = torch.nonzero(diff > 0)[:10]
for i in idx:
    m, n = int(i[0]), int(i[1])
    print(m, n, got[m, n].item(), want[m, n].item(), x[m, n].item(), gn[n].item(), P[:, 0, n // 32, (m & 15) + 16 * ((n & 31) >> 3), n & 7].float().tolist())
