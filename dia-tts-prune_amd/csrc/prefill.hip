// Decoder prefill over an audio prompt, batched: instead of replaying the prompt rows one
// decode step at a time (1.1 ms each), all prompt rows of all utterances — both CFG rows — run as ONE packed
// batch through the MFMA-tiled GEMMs, with the three kernels below for what is not a GEMM.  Semantics are the
// replay's (oracle.generate docstring): token row r -> cache slot r at RoPE position r + 1; the K/V written to
// the caches are rounded to the cache dtype first and attention reads them back from the caches, exactly what
// the decode steps will do afterwards.  bf16 caches with the blocked V layout only (the perf configuration);
// fp32 caches keep the replay.
//
// Packing: segment s = (utterance b, CFG row c) owns packed rows [seg_off[s], seg_off[s] + seg_len[s]),
// seg_off a multiple of 32; row_seg[m] = segment of packed row m or -1 (padding); seg_row[s] = 2b + c.
#include "common.hpp"
#include "../../include/dia_hip.h"
#include "errors.hpp"
#include "launch.hpp"

namespace {

constexpr int HD = 128;

struct PrefK {
  const int* row_seg; const int* seg_off; const int* seg_len; const int* seg_row; int rows;
  // embedding
  const int* tokens; int T, C, V, D; const float* emb; const float* g; float* x;
  bf16_raw* P; long p_plane_stride; int p_ktiles; float* ssq; int ssq_ld;
  // K/V append + attention
  const float* q; int ldq, q_off, k_off, v_off;
  int q_heads, kv_heads, kv_cap;         // attention: q_heads query heads, kv_heads cache heads (group = q/kv)
  bf16_raw* kc; bf16_raw* vc;            // caches [cache row][kv_heads][kv_cap][128], V blocked
  const float* cos_t; const float* sin_t;
  int causal;                            // 1 = self (keys 0..r of the segment's own cache row), 0 = cross
  const int* text_len;                   // cross: keys of utterance b
};

// x[m] = sum_c emb[c][tokens[b][r][c]] (layers.py:691-696), planes(x * g), strip ssq — one packed row per workgroup
__global__ __launch_bounds__(256) void k_prefill_embed(PrefK p) {
  __shared__ int tok[16];
  const int m = blockIdx.x, s = p.row_seg[m];
  if (s < 0) return;
  const int r = m - p.seg_off[s], b = p.seg_row[s] >> 1;
  if (threadIdx.x < p.C) tok[threadIdx.x] = p.tokens[((long)b * p.T + r) * p.C + threadIdx.x];
  __syncthreads();
  for (int d0 = threadIdx.x * 8; d0 < p.D; d0 += 256 * 8) {
    float v[8];
    for (int c = 0; c < p.C; ++c) {          // sequential sum in channel order
      const float* e = p.emb + ((long)c * p.V + tok[c]) * p.D + d0;
      const float4 a = *reinterpret_cast<const float4*>(e), bq = *reinterpret_cast<const float4*>(e + 4);
      if (c == 0) { v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w; }
      else { v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += bq.x; v[5] += bq.y; v[6] += bq.z; v[7] += bq.w; }
    }
    float ss = 0.f, vg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { ss += v[j] * v[j]; vg[j] = p.g ? v[j] * p.g[d0 + j] : v[j]; }
    const float other = __shfl_xor(ss, 1, 64);
    float* xo = p.x + (long)m * p.D + d0;
    *reinterpret_cast<float4*>(xo) = float4{v[0], v[1], v[2], v[3]};
    *reinterpret_cast<float4*>(xo + 4) = float4{v[4], v[5], v[6], v[7]};
    emit_planes8(p.P, p.p_plane_stride, p.p_ktiles, m, d0, vg);
    if (((d0 >> 3) & 1) == 0) p.ssq[(long)(d0 >> 4) * p.ssq_ld + m] = ss + other;
  }
}

// K = RoPE(k, r + 1) and V of every packed row into the self caches (KVCache.update for slots 0..Tp-1)
__global__ __launch_bounds__(256) void k_prefill_kv(PrefK p) {
  const int h = blockIdx.x, blk = blockIdx.y, tid = threadIdx.x;
  for (int t = tid; t < 32 * 64; t += 256) {
    const int rr = t >> 6, d = t & 63, m = blk * 32 + rr;
    const int s = p.row_seg[m];
    if (s < 0) continue;
    const int r = m - p.seg_off[s], pos = r + 1;
    const float* kr = p.q + (long)m * p.ldq + p.k_off + h * HD;
    const float x1 = kr[d], x2 = kr[d + 64];
    const float c = p.cos_t[(long)pos * 64 + d], sn = p.sin_t[(long)pos * 64 + d];
    bf16_raw* ko = p.kc + (((long)p.seg_row[s] * p.kv_heads + h) * p.kv_cap + r) * HD;
    KVElem<bf16_raw>::store(ko + d, x1 * c - x2 * sn);
    KVElem<bf16_raw>::store(ko + d + 64, x1 * sn + x2 * c);
  }
  for (int t = tid; t < 32 * HD; t += 256) {
    const int rr = t & 31, d = t >> 5, m = blk * 32 + rr;
    const int s = p.row_seg[m];
    if (s < 0) continue;
    const int r = m - p.seg_off[s];
    bf16_raw* vo = p.vc + ((long)p.seg_row[s] * p.kv_heads + h) * p.kv_cap * HD + (long)(r >> 5) * HD * 32 + (long)d * 32 + (r & 31);
    KVElem<bf16_raw>::store(vo, p.q[(long)m * p.ldq + p.v_off + h * HD + d]);
  }
}

__device__ __forceinline__ float row16_max_p(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
  return v;
}

// Many-query attention over the bf16 caches: grid (query head, 16-row query tile).  q as three planes (exact),
// K / V straight from the caches as single bf16 operands, p as three planes; 4 waves split the keys in 32-key
// granules, online softmax per query row, waves merged through LDS (k_attn_enc_mfma with the caches as K/V).
// causal: query row r sees keys 0..r of its own cache row; else the cond segment sees the utterance's text keys
// and the uncond segment gets exact zeros.
__global__ __launch_bounds__(256) void k_prefill_attn(PrefK p) {
  constexpr int NWV = 4;
  __shared__ __attribute__((aligned(16))) bf16_raw qf[4][DIA_NPLANES][16][4][8];
  __shared__ __attribute__((aligned(16))) bf16_raw pbuf[NWV][DIA_NPLANES][16][32];
  __shared__ float part[NWV * 16 * HD];
  __shared__ float pm_s[NWV * 16], pl_s[NWV * 16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = blockIdx.x, qt = blockIdx.y;
  const int arow = lane & 15, akq = lane >> 4;
  const int s = p.row_seg[qt * 16];
  if (s < 0) return;
  const int off = p.seg_off[s], len = p.seg_len[s], crow = p.seg_row[s];
  const int r0 = qt * 16 - off;                       // first prompt row of this tile
  const int group = p.q_heads / p.kv_heads;
  int nkeys; const bf16_raw* Kh; const bf16_raw* Vh;
  if (p.causal) {
    nkeys = min(len, r0 + 16);
    const long base = ((long)crow * p.kv_heads + h / group) * p.kv_cap * HD;
    Kh = p.kc + base; Vh = p.vc + base;
  } else {
    if ((crow & 1) == 0) {                            // uncond row: cross-attention output is exactly 0
      const int r = tid >> 4, d0 = (tid & 15) * 8, m = qt * 16 + r;
      const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (m - off < len) emit_planes8(p.P, p.p_plane_stride, p.p_ktiles, m, h * HD + d0, z);
      return;
    }
    nkeys = p.text_len[crow >> 1];
    const long base = ((long)(crow >> 1) * p.kv_heads + h / group) * p.kv_cap * HD;
    Kh = p.kc + base; Vh = p.vc + base;
  }
  {
    const int r = tid >> 4, m = min(qt * 16 + r, off + len - 1), pos = (m - off) + 1;
    const float* qh = p.q + (long)m * p.ldq + p.q_off + h * HD;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = (tid & 15) * 4 + i;
      const float x1 = qh[d], x2 = qh[d + 64];
      const float c = p.cos_t[(long)pos * 64 + d], sn = p.sin_t[(long)pos * 64 + d];
      const float qv[2] = {x1 * c - x2 * sn, x1 * sn + x2 * c};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int D = d + 64 * e;
        __bf16 a, bb, c3;
        split3(qv[e], a, bb, c3);
        qf[D >> 5][0][r][(D >> 3) & 3][D & 7] = *reinterpret_cast<const bf16_raw*>(&a);
        qf[D >> 5][1][r][(D >> 3) & 3][D & 7] = *reinterpret_cast<const bf16_raw*>(&bb);
        qf[D >> 5][2][r][(D >> 3) & 3][D & 7] = *reinterpret_cast<const bf16_raw*>(&c3);
      }
    }
  }
  __syncthreads();
  const float scale = 0.08838834764831845f;
  f32x4 O[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) O[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mrun[4], lrun[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) { mrun[r] = -INFINITY; lrun[r] = 0.f; }
  const int klast = max(nkeys - 1, 0);
  const int ngran = (nkeys + 31) >> 5;
  for (int g = w; g < ngran; g += NWV) {
    const int key0 = g << 5;
    f32x4 S[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 qa[DIA_NPLANES];
#pragma unroll
      for (int pl = 0; pl < DIA_NPLANES; ++pl) qa[pl] = *reinterpret_cast<const bf16x8*>(&qf[ks][pl][arow][akq][0]);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const bf16x8 kb = *reinterpret_cast<const bf16x8*>(Kh + (long)min(key0 + 16 * t + arow, klast) * HD + 32 * ks + 8 * akq);
#pragma unroll
        for (int pl = 0; pl < DIA_NPLANES; ++pl) S[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[pl], kb, S[t], 0, 0, 0);
      }
    }
    float alpha[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int lim = p.causal ? min(nkeys, r0 + 4 * akq + r + 1) : nkeys;      // keys this query row may see
      const bool v0 = key0 + arow < lim, v1 = key0 + 16 + arow < lim;
      const float s0 = v0 ? S[0][r] * scale : -INFINITY, s1 = v1 ? S[1][r] * scale : -INFINITY;
      const float mnew = fmaxf(mrun[r], row16_max_p(fmaxf(s0, s1)));
      alpha[r] = (mrun[r] == -INFINITY) ? 0.f : expf(mrun[r] - mnew);
      const float p0 = (s0 == -INFINITY) ? 0.f : expf(s0 - mnew), p1 = (s1 == -INFINITY) ? 0.f : expf(s1 - mnew);
      lrun[r] = lrun[r] * alpha[r] + row16_sum(p0 + p1);
      mrun[r] = mnew;
      __bf16 a, bb, c3;
      split3(p0, a, bb, c3);
      pbuf[w][0][4 * akq + r][arow] = *reinterpret_cast<const bf16_raw*>(&a);
      pbuf[w][1][4 * akq + r][arow] = *reinterpret_cast<const bf16_raw*>(&bb);
      pbuf[w][2][4 * akq + r][arow] = *reinterpret_cast<const bf16_raw*>(&c3);
      split3(p1, a, bb, c3);
      pbuf[w][0][4 * akq + r][16 + arow] = *reinterpret_cast<const bf16_raw*>(&a);
      pbuf[w][1][4 * akq + r][16 + arow] = *reinterpret_cast<const bf16_raw*>(&bb);
      pbuf[w][2][4 * akq + r][16 + arow] = *reinterpret_cast<const bf16_raw*>(&c3);
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int r = 0; r < 4; ++r) O[nb][r] *= alpha[r];
    bf16x8 pa[DIA_NPLANES];
#pragma unroll
    for (int pl = 0; pl < DIA_NPLANES; ++pl) pa[pl] = *reinterpret_cast<const bf16x8*>(&pbuf[w][pl][arow][8 * akq]);
    const bf16_raw* Vblk = Vh + (long)g * HD * 32;
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) {
      const bf16x8 vb = *reinterpret_cast<const bf16x8*>(Vblk + (long)(16 * nb + arow) * 32 + 8 * akq);
#pragma unroll
      for (int pl = 0; pl < DIA_NPLANES; ++pl) O[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[pl], vb, O[nb], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) part[(w * 16 + 4 * akq + r) * HD + 16 * nb + arow] = O[nb][r];
    if (arow == 0) { pm_s[w * 16 + 4 * akq + r] = mrun[r]; pl_s[w * 16 + 4 * akq + r] = lrun[r]; }
  }
  __syncthreads();
  {
    const int r = tid >> 4, d0 = (tid & 15) * 8, m = qt * 16 + r;
    if (m - off >= len) return;
    float mm = -INFINITY;
#pragma unroll
    for (int ww = 0; ww < NWV; ++ww) mm = fmaxf(mm, pm_s[ww * 16 + r]);
    float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, Ls = 0.f;
#pragma unroll
    for (int ww = 0; ww < NWV; ++ww) {
      const float pmw = pm_s[ww * 16 + r];
      const float f = (pmw == -INFINITY) ? 0.f : expf(pmw - mm);
      Ls += pl_s[ww * 16 + r] * f;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += part[(ww * 16 + r) * HD + d0 + j] * f;
    }
    const float inv = Ls > 0.f ? 1.0f / Ls : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] *= inv;
    emit_planes8(p.P, p.p_plane_stride, p.p_ktiles, m, h * HD + d0, o);
  }
}

int fill(const dia_dec_prefill_args* a, PrefK& k) {
  if (!a || !a->row_seg || !a->seg_off || !a->seg_len || !a->seg_row || a->rows <= 0 || a->rows % 32 != 0)
    return dia_fail(DIA_E_ARG, "dia_dec_prefill: packing arrays missing or rows not a multiple of 32");
  k.row_seg = a->row_seg; k.seg_off = a->seg_off; k.seg_len = a->seg_len; k.seg_row = a->seg_row; k.rows = a->rows;
  k.tokens = a->tokens; k.T = a->T; k.C = a->C; k.V = a->V; k.D = a->D; k.emb = a->emb; k.g = a->g; k.x = a->x;
  k.P = (bf16_raw*)a->P; k.p_plane_stride = a->p_plane_stride; k.p_ktiles = a->p_ktiles; k.ssq = a->ssq; k.ssq_ld = a->ssq_ld;
  k.q = a->q; k.ldq = a->ldq; k.q_off = a->q_off; k.k_off = a->k_off; k.v_off = a->v_off;
  k.q_heads = a->q_heads; k.kv_heads = a->kv_heads; k.kv_cap = a->kv_cap;
  k.kc = (bf16_raw*)a->kc; k.vc = (bf16_raw*)a->vc; k.cos_t = a->cos_t; k.sin_t = a->sin_t; k.causal = a->causal; k.text_len = a->text_len;
  return DIA_OK;
}

}  // namespace

extern "C" int dia_dec_prefill_embed(const dia_dec_prefill_args* a, void* stream) {
  PrefK k; int rc = fill(a, k); if (rc) return rc;
  if (!a->tokens || !a->emb || !a->x || !a->P || !a->ssq || a->C <= 0 || a->C > 16 || a->D % 16 != 0 || a->p_ktiles * 32 < a->D || a->p_plane_stride % 8 != 0)
    return dia_fail(DIA_E_ARG, "dia_dec_prefill_embed: bad argument");
  dia_launch<k_prefill_embed>(dim3(a->rows), dim3(256), 0, (hipStream_t)stream, k);
  return dia_check_launch("k_prefill_embed");
}

extern "C" int dia_dec_prefill_kv(const dia_dec_prefill_args* a, void* stream) {
  PrefK k; int rc = fill(a, k); if (rc) return rc;
  if (!a->q || !a->kc || !a->vc || !a->cos_t || !a->sin_t || a->kv_heads <= 0 || a->kv_cap % 32 != 0)
    return dia_fail(DIA_E_ARG, "dia_dec_prefill_kv: bad argument");
  dia_launch<k_prefill_kv>(dim3(a->kv_heads, a->rows / 32), dim3(256), 0, (hipStream_t)stream, k);
  return dia_check_launch("k_prefill_kv");
}

extern "C" int dia_dec_prefill_attn(const dia_dec_prefill_args* a, void* stream) {
  PrefK k; int rc = fill(a, k); if (rc) return rc;
  if (!a->q || !a->kc || !a->vc || !a->cos_t || !a->sin_t || !a->P || a->q_heads <= 0 || a->kv_heads <= 0 || a->q_heads % a->kv_heads != 0 ||
      a->kv_cap % 32 != 0 || (!a->causal && !a->text_len) || a->p_plane_stride % 8 != 0 || (a->q_heads * 128 + 31) / 32 > a->p_ktiles)
    return dia_fail(DIA_E_ARG, "dia_dec_prefill_attn: bad argument");
  dia_launch<k_prefill_attn>(dim3(a->q_heads, a->rows / 16), dim3(256), 0, (hipStream_t)stream, k);
  return dia_check_launch("k_prefill_attn");
}
