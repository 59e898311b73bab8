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


def tile_weight_planes(w2d: torch.Tensor) -> Tuple[torch.Tensor, int, int]:
    """[K, N] fp32 -> (bf16 tiles [3, N/16, K/32, 64, 8], K/32, N/16): the hi / mid / lo planes of the weights
    (hi + mid + lo == w exactly), one tile set each, back to back — dia_gemm_args.w_planes = 3."""
    planes = [tile_weight(pl.float()) for pl in split3(w2d.float())]
    return torch.stack([t for t, _, _ in planes]).contiguous(), planes[0][1], planes[0][2]


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


def pack_f32_tiles(x: torch.Tensor, ktiles: int | None = None, mtiles: int | None = None) -> torch.Tensor:
    """fp32 [M, K] -> fp32 activation tiles [mtiles, ktiles, 64, 8]: the fragment order of one plane with 4-byte elements
    (dia_gemm_args.act_f32; the 5..128-row kernels split the planes in registers)."""
    M, K = x.shape
    mt = mtiles if mtiles is not None else _ceil(M, 16) // 16
    kt = ktiles if ktiles is not None else _ceil(K, 32) // 32
    xp = torch.zeros(mt * 16, kt * 32, dtype=torch.float32, device=x.device)
    xp[:M, :K] = x
    return xp.reshape(mt, 16, kt, 4, 8).permute(0, 2, 3, 1, 4).reshape(mt, kt, 64, 8).contiguous()


def unpack_f32_tiles(t: torch.Tensor, M: int, K: int) -> torch.Tensor:
    """inverse of pack_f32_tiles"""
    mt, kt = t.shape[:2]
    return t.reshape(mt, kt, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(mt * 16, kt * 32)[:M, :K].contiguous()


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


SPARSE_HEADER = 80          # bytes: 64 lane masks + 4 x uint16 row prefixes + 8 pad


def sparse_tile_weight(tiles: torch.Tensor):
    """bf16 weight tiles [ns, kt, 64, 8] -> (blocks uint8 [bytes], toff int32 [ns*kt]) — the zero-skipping stream
    format for unstructured-pruned matrices (offline_prune.py --prune-mode unstructured):

      block of a tile = [64 x uint8 lane masks (bit j = element j of the lane's 8-element fragment is non-zero)]
                        [4 x uint16: non-zeros in lanes < 16*q]  [8 pad bytes]
                        [non-zero bf16 values, lane-major]                 padded to 16 bytes, <= 1024 bytes
      a tile with more than 472 non-zeros is stored raw (1024 bytes, the dense fragment order)
      toff[tile] = (block offset / 16) << 8 | (block size / 16, or 0 for a raw tile)

    One 16-byte load per lane still fetches a whole tile (lanes past the block re-read its last chunk), so the
    kernel keeps the dense kernel's prefetch structure and simply moves fewer bytes."""
    ns, kt = tiles.shape[0], tiles.shape[1]
    dev = tiles.device
    raw = tiles.contiguous().view(torch.int16).reshape(ns * kt, 64, 8)
    nzm = raw != 0                                                    # [T, 64, 8]
    cnt = nzm.sum(dim=2)                                              # [T, 64]
    nnz = cnt.sum(dim=1)                                              # [T]
    dense = nnz > 472
    size = torch.where(dense, torch.full_like(nnz, 1024), (SPARSE_HEADER + 2 * nnz + 15) // 16 * 16)
    off = torch.cumsum(size, 0) - size
    total = int(size.sum().item())
    blocks = torch.zeros(total, dtype=torch.uint8, device=dev)
    T = ns * kt
    # raw tiles
    if bool(dense.any()):
        idx = torch.nonzero(dense).flatten()
        dst = (off[idx][:, None] + torch.arange(1024, device=dev)[None, :]).reshape(-1)
        blocks[dst] = raw[idx].reshape(len(idx), -1).view(torch.uint8).reshape(-1)
    sp = ~dense
    if bool(sp.any()):
        idx = torch.nonzero(sp).flatten()
        bits = (nzm[idx].to(torch.int32) * (1 << torch.arange(8, device=dev, dtype=torch.int32))[None, None, :]).sum(dim=2)   # [S, 64]
        dst = (off[idx][:, None] + torch.arange(64, device=dev)[None, :]).reshape(-1)
        blocks[dst] = bits.to(torch.uint8).reshape(-1)
        rowpre = torch.cumsum(cnt[idx].reshape(len(idx), 4, 16).sum(dim=2), dim=1) - cnt[idx].reshape(len(idx), 4, 16).sum(dim=2)   # [S, 4]
        rp = rowpre.to(torch.int16).contiguous().view(torch.uint8).reshape(len(idx), 8)
        dst = (off[idx][:, None] + 64 + torch.arange(8, device=dev)[None, :]).reshape(-1)
        blocks[dst] = rp.reshape(-1)
        # values: global order of the non-zeros is already (tile, lane, j)
        sel = nzm[idx]
        vals = raw[idx][sel]                                           # int16 [sum nnz]
        tile_of = torch.repeat_interleave(torch.arange(len(idx), device=dev), nnz[idx])
        start = torch.cumsum(nnz[idx], 0) - nnz[idx]
        within = torch.arange(vals.numel(), device=dev) - start[tile_of]
        dstb = off[idx][tile_of] + SPARSE_HEADER + 2 * within
        vb = vals.contiguous().view(torch.uint8).reshape(-1, 2)
        blocks[dstb] = vb[:, 0]
        blocks[dstb + 1] = vb[:, 1]
    chunks = torch.where(dense, torch.zeros_like(size), size // 16)
    toff = ((off // 16) << 8 | chunks).to(torch.int32)
    return blocks, toff.contiguous()


def sparse_untile(blocks: torch.Tensor, toff: torch.Tensor, ns: int, kt: int) -> torch.Tensor:
    """inverse of sparse_tile_weight (test helper, slow): bf16 tiles [ns, kt, 64, 8]"""
    b = blocks.cpu().numpy()
    t = toff.cpu().numpy().astype("int64") & 0xFFFFFFFF
    import numpy as np
    out = np.zeros((ns * kt, 64, 8), dtype=np.int16)
    for i in range(ns * kt):
        o, ch = int(t[i] >> 8) * 16, int(t[i] & 255)
        if ch == 0:
            out[i] = b[o: o + 1024].view(np.int16).reshape(64, 8)
            continue
        masks = b[o: o + 64]
        vals = b[o + SPARSE_HEADER: o + ch * 16].view(np.int16)
        k = 0
        for l in range(64):
            for j in range(8):
                if masks[l] >> j & 1:
                    out[i, l, j] = vals[k]; k += 1
    return torch.from_numpy(out).view(torch.bfloat16).reshape(ns, kt, 64, 8)


# ---- ring layout of the persistent MLP segment (csrc/seg.hip) ------------------------------------------------------
SEG_CUS = 256            # one workgroup per CU of an MI355X
SEG_K = 2048             # contraction length of one slot (16 tiles of 128 k x 4 columns)


def _seg_slots(w2d: torch.Tensor) -> torch.Tensor:
    """[2048, N] float (N % 4 == 0) -> bf16 [N/4, 16, 64, 8]: one 16 KiB slot per group of 4 columns.  Tile t, lane l,
    element j = W[128 t + 32 ((l & 15) >> 2) + 8 (l >> 4) + j][4 group + (l & 3)] — four k-tiles of the four columns side
    by side in the 16 B-operand columns of v_mfma_f32_16x16x32_bf16 (the matching A operand carries the rows of k-tile g in
    rows 4g..4g+3, so the product is block diagonal)."""
    K, N = w2d.shape
    if K != SEG_K or N % 4:
        raise ValueError("seg slots: K must be 2048 and N a multiple of 4")
    lane = torch.arange(64, device=w2d.device)
    t = torch.arange(16, device=w2d.device)
    j = torch.arange(8, device=w2d.device)
    k = (128 * t[:, None, None] + (32 * ((lane & 15) >> 2) + 8 * (lane >> 4))[None, :, None] + j[None, None, :])   # [16, 64, 8]
    c = (lane & 3)[None, :, None].expand(16, 64, 8)
    g = w2d.reshape(K, N // 4, 4)
    out = g[k[None], torch.arange(N // 4, device=w2d.device)[:, None, None, None], c[None]]     # [N/4, 16, 64, 8]
    return out.to(torch.bfloat16).contiguous()


def seg_ring(co: torch.Tensor, wi_gate: torch.Tensor, wi_up: torch.Tensor, wo: torch.Tensor, qkv_next) -> torch.Tensor:
    """The weights one persistent segment streams, per CU in consumption order: bf16 [256][slots][16][64][8].
      co       [2048, 2048]  cross-attention o_proj ([heads*128, D]): CU c owns columns [8c, 8c+8)           -> 2 slots
      wi_gate / wi_up [2048, 8192]: CU c owns hidden units [32c, 32c+32), gate slots then up slots          -> 16 slots
      wo       [8192, 2048]: CU c owns K quarter c >> 6 of columns [32 (c & 63), +32)                        -> 8 slots
      qkv_next [2048, 3072] or None (last layer): CU c owns columns [12c, 12c+12)                            -> 3 slots"""
    D, F = co.shape[1], wi_gate.shape[1]
    if co.shape != (SEG_K, 2048) or wi_gate.shape != (2048, 8192) or wi_up.shape != (2048, 8192) or wo.shape != (8192, 2048):
        raise ValueError("seg_ring: built for the Dia-1.6B decoder shapes")
    parts = [_seg_slots(co).reshape(SEG_CUS, 2, 16, 64, 8),
             _seg_slots(wi_gate).reshape(SEG_CUS, 8, 16, 64, 8), _seg_slots(wi_up).reshape(SEG_CUS, 8, 16, 64, 8)]
    # wo: [quarter q][K 2048][2048 columns] -> slots [q][512 groups] -> CU (q, c) takes groups [8c, 8c+8)
    wo_s = torch.stack([_seg_slots(wo[q * SEG_K:(q + 1) * SEG_K]) for q in range(4)])            # [4, 512, 16, 64, 8]
    parts.append(wo_s.reshape(4, 64, 8, 16, 64, 8).reshape(SEG_CUS, 8, 16, 64, 8))
    if qkv_next is not None:
        if qkv_next.shape != (2048, 3072):
            raise ValueError("seg_ring: q/k/v projection must be [2048, 3072]")
        parts.append(_seg_slots(qkv_next).reshape(SEG_CUS, 3, 16, 64, 8))
    return torch.cat(parts, dim=1).contiguous()


def diag_tile_weight(w2d: torch.Tensor) -> torch.Tensor:
    """[K, N] float (K % 128 == 0, N % 4 == 0) -> bf16 [N/4, K/128, 64, 8]: the diagonal layout of dia_gemm_args.w_layout = 1 (k_gemv_diag).
    Tile t of column group g, lane l, element j = W[128 t + 32 ((l & 15) >> 2) + 8 (l >> 4) + j][4 g + (l & 3)]: four k-tiles of the group's
    four columns side by side in the B operand of v_mfma_f32_16x16x32_bf16 (as the slots of seg_ring, for any K)."""
    K, N = w2d.shape
    if K % 128 or N % 4:
        raise ValueError("diag_tile_weight: K must be a multiple of 128 and N of 4")
    lane = torch.arange(64, device=w2d.device)
    t = torch.arange(K // 128, device=w2d.device)
    j = torch.arange(8, device=w2d.device)
    k = (128 * t[:, None, None] + (32 * ((lane & 15) >> 2) + 8 * (lane >> 4))[None, :, None] + j[None, None, :])      # [K/128, 64, 8]
    c = (lane & 3)[None, :, None].expand(K // 128, 64, 8)
    g = w2d.reshape(K, N // 4, 4)
    out = g[k[None], torch.arange(N // 4, device=w2d.device)[:, None, None, None], c[None]]
    return out.to(torch.bfloat16).contiguous()
