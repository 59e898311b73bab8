"""HBM data layouts of the decode path — pure tensor reshuffles, no arithmetic.

* weight tiles: bf16 ``[strip=n/16][ktile=k/32][lane][8]`` with lane ``l`` holding
  ``W[32*ktile + 8*(l>>4) + j][16*strip + (l&15)]`` — the B operand of ``v_mfma_f32_16x16x32_bf16``;
* activation planes: bf16 ``[3][mtile=m/16][ktile][lane][8]`` with lane ``l`` holding
  ``X[16*mtile + (l&15)][32*ktile + 8*(l>>4) + j]`` — the A operand; the three planes sum to the
  fp32 value exactly.

The reference stores DenseGeneral kernels as ``in_shapes + out_features`` (dia/layers.py:47-51); the
functions below flatten them to ``[K, N]`` first.
"""

from __future__ import annotations

from typing import Tuple

import torch


def _ceil(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def tile_weight(w2d: torch.Tensor) -> Tuple[torch.Tensor, int, int]:
    """[K, N] float -> (bf16 tiles [N/16, K/32, 64, 8], K/32, N/16); K, N zero-padded to 32 / 16."""
    K, N = w2d.shape
    Kp, Np = _ceil(K, 32), _ceil(N, 16)
    if (Kp, Np) != (K, N):
        wp = torch.zeros(Kp, Np, dtype=w2d.dtype, device=w2d.device)
        wp[:K, :N] = w2d
        w2d = wp
    kt, ns = Kp // 32, Np // 16
    t = w2d.reshape(kt, 4, 8, ns, 16).permute(3, 0, 1, 4, 2)        # (strip, kt, kq, c, j)
    return t.reshape(ns, kt, 64, 8).to(torch.bfloat16).contiguous(), kt, ns


def untile_weight(tiles: torch.Tensor, K: int, N: int) -> torch.Tensor:
    ns, kt = tiles.shape[0], tiles.shape[1]
    w = tiles.float().reshape(ns, kt, 4, 16, 8).permute(1, 2, 4, 0, 3).reshape(kt * 32, ns * 16)
    return w[:K, :N].contiguous()


def split3(x: torch.Tensor):
    hi = x.to(torch.bfloat16)
    r = x - hi.float()
    mid = r.to(torch.bfloat16)
    lo = (r - mid.float()).to(torch.bfloat16)
    return hi, mid, lo


def pack_planes(x: torch.Tensor, ktiles: int | None = None, mtiles: int | None = None) -> torch.Tensor:
    """fp32 [M, K] -> bf16 planes [3, mtiles, ktiles, 64, 8]."""
    M, K = x.shape
    mt = mtiles if mtiles is not None else _ceil(M, 16) // 16
    kt = ktiles if ktiles is not None else _ceil(K, 32) // 32
    xp = torch.zeros(mt * 16, kt * 32, dtype=torch.float32, device=x.device)
    xp[:M, :K] = x
    out = []
    for pl in split3(xp):
        out.append(pl.reshape(mt, 16, kt, 4, 8).permute(0, 2, 3, 1, 4).reshape(mt, kt, 64, 8))
    return torch.stack(out).contiguous()


def unpack_planes(p: torch.Tensor, M: int, K: int) -> torch.Tensor:
    """inverse of pack_planes (sums the three planes in fp32)."""
    _, mt, kt = p.shape[:3]
    s = p[0].float() + p[1].float() + p[2].float()
    x = s.reshape(mt, kt, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(mt * 16, kt * 32)
    return x[:M, :K].contiguous()


def interleave_gate_up(wi: torch.Tensor) -> torch.Tensor:
    """wi_fused kernel [D, 2, F] (layers.py:77-82) -> [D, 2F] where every 16-column strip holds
    8 gate columns followed by the 8 matching up columns."""
    D, two, F = wi.shape
    assert two == 2 and F % 8 == 0
    g = wi[:, 0, :].reshape(D, F // 8, 8)
    u = wi[:, 1, :].reshape(D, F // 8, 8)
    return torch.cat([g, u], dim=2).reshape(D, 2 * F)


def rope_pair_perm(head_dim: int = 128) -> torch.Tensor:
    """column order inside a head so that a RoPE pair (d, d+head_dim/2) sits in adjacent columns."""
    c = torch.arange(head_dim)
    return (c // 2) + (head_dim // 2) * (c % 2)


def rope_tables(npos: int, head_dim: int, min_ts: int, max_ts: int):
    """cos/sin [npos, head_dim/2] fp32 built with the reference's op sequence
    (dia/layers.py:126-132, 145-146, 161-162): inv_freq = 1/(min*(max/min)**(2i/H)), theta = pos*inv_freq."""
    half = head_dim // 2
    fraction = (2.0 * torch.arange(0, half)) / head_dim
    inv_freq = (1.0 / (min_ts * (max_ts / min_ts) ** fraction)).to(torch.float32)
    pos = torch.arange(npos, dtype=torch.float32)
    f = pos.unsqueeze(-1) * inv_freq
    return torch.cos(f.to(torch.float32)).contiguous(), torch.sin(f.to(torch.float32)).contiguous()


def v_to_blocked(v: torch.Tensor) -> torch.Tensor:
    """V cache [..., T, 128] -> blocked [..., T/32, 128, 32] (the MFMA attention kernel's B operand wants
    8 consecutive KEYS of one dim in 16 bytes); same number of elements, T % 32 == 0."""
    *lead, T, H = v.shape
    return v.reshape(*lead, T // 32, 32, H).transpose(-1, -2).contiguous()


def v_from_blocked(vb: torch.Tensor) -> torch.Tensor:
    *lead, nb, H, k = vb.shape
    return vb.transpose(-1, -2).reshape(*lead, nb * k, H).contiguous()
