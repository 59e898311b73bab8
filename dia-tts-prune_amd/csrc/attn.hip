// Single-query attention for the decode path, head_dim 128, fp32 arithmetic.
//
// Replaces, per reference call site: RotaryEmbedding.forward on q and the new k (dia/layers.py:135-173,
// 278-279), KVCache.update (dia/state.py:99-103), the GQA repeat_interleave (layers.py:319-320, never
// materialised here: one workgroup serves a kv head and its `G` query heads) and
// F.scaled_dot_product_attention (layers.py:329-337).
//
// One workgroup = (kv head, query row).  Two passes over the keys:
//   pass 1  16 lanes per key (8 dims each, one 16 B load for bf16 K / two for fp32), G dot products,
//           xor-butterfly over the 16 lanes, score -> LDS
//   softmax one wave per query head: max, exp, sum (the 1/sum is applied once at the end)
//   pass 2  same key->lane-group mapping on V, p from LDS, partial outputs reduced over the 4 groups
//           of a wave by shuffles and over the waves through LDS in a fixed order (deterministic).
// The output leaves as three bf16 planes in MFMA A-operand order for the o_proj GEMM.
#include "common.hpp"
#include "../../include/dia_hip.h"
#include "errors.hpp"

namespace {

constexpr int HD = 128;
constexpr int NT = 512;            // threads per workgroup
constexpr int NGRP = NT / 16;      // key groups
constexpr int U = 4;               // keys per lane group per round (independent loads in flight)

struct AttnK {
  int mode, n_kv_heads, n_rows, kv_cap;
  const float* q; int ldq, q_off, k_off, v_off;
  void* kc; void* vc;
  const int* cur; const int* len; int enc_len;
  const float* cos_t; const float* sin_t;
  bf16_raw* P; long p_plane_stride; int p_ktiles;
};

template <typename KVT, int G>
__global__ __launch_bounds__(NT) void k_attn(AttnK p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* q_s = smem;                       // [G][128]
  float* knew = q_s + G * HD;              // [128]
  float* vnew = knew + HD;                 // [128]
  float* lsum = vnew + HD;                 // [G] (padded to 8)
  float* part = lsum + 8;                  // [NT/64][G][128]
  float* sc = part + (NT / 64) * G * HD;   // [G][nkeys_cap]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kvh = blockIdx.x;
  int qrow, kvrow, pos, nkeys, slot = -1;
  if (p.mode == DIA_ATTN_SELF) {
    qrow = blockIdx.y; kvrow = qrow;
    const int c = p.cur[qrow >> 1];
    pos = c; nkeys = c; slot = c - 1;
  } else if (p.mode == DIA_ATTN_CROSS) {
    const int b = blockIdx.y;
    qrow = 2 * b + 1; kvrow = b;
    pos = p.cur[b]; nkeys = p.len[b];
  } else {
    qrow = blockIdx.y; kvrow = 0; pos = qrow; nkeys = p.enc_len;
  }
  const int sc_ld = (p.mode == DIA_ATTN_ENC) ? p.enc_len : p.kv_cap;

  // ---- prologue: RoPE(q) (and RoPE(k_new), v_new -> cache) ------------------------------------
  const float* qr = p.q + (long)qrow * p.ldq;
  for (int t = tid; t < G * 64; t += NT) {
    const int g = t >> 6, d = t & 63;
    const float* qh = qr + p.q_off + (kvh * G + g) * HD;
    const float x1 = qh[d], x2 = qh[d + 64];
    const float c = p.cos_t[(long)pos * 64 + d], s = p.sin_t[(long)pos * 64 + d];
    q_s[g * HD + d] = x1 * c - x2 * s;
    q_s[g * HD + d + 64] = x1 * s + x2 * c;
  }
  KVT* Kc = reinterpret_cast<KVT*>(p.kc) + ((long)kvrow * p.n_kv_heads + kvh) * p.kv_cap * HD;
  KVT* Vc = reinterpret_cast<KVT*>(p.vc) + ((long)kvrow * p.n_kv_heads + kvh) * p.kv_cap * HD;
  if (p.mode == DIA_ATTN_SELF && tid >= NT - 64) {
    const int d = tid - (NT - 64);
    const float* kh = qr + p.k_off + kvh * HD;
    const float* vh = qr + p.v_off + kvh * HD;
    const float x1 = kh[d], x2 = kh[d + 64];
    const float c = p.cos_t[(long)pos * 64 + d], s = p.sin_t[(long)pos * 64 + d];
    const float k1 = KVElem<KVT>::round(x1 * c - x2 * s), k2 = KVElem<KVT>::round(x1 * s + x2 * c);
    const float v1 = KVElem<KVT>::round(vh[d]), v2 = KVElem<KVT>::round(vh[d + 64]);
    KVElem<KVT>::store(Kc + (long)slot * HD + d, k1);
    KVElem<KVT>::store(Kc + (long)slot * HD + d + 64, k2);
    KVElem<KVT>::store(Vc + (long)slot * HD + d, v1);
    KVElem<KVT>::store(Vc + (long)slot * HD + d + 64, v2);
  }
  __syncthreads();

  // ---- pass 1: scores -------------------------------------------------------------------------
  const int grp = tid >> 4, sub = tid & 15;
  float qreg[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int j = 0; j < 8; ++j) qreg[g][j] = q_s[g * HD + sub * 8 + j];
  const float scale = 0.08838834764831845f;   // 1/sqrt(128)
  // U keys per lane group per round, all U loads issued before the first use.  The key written by
  // this workgroup in the prologue is read back from memory like any other (the barrier above orders it).
  for (int key0 = grp; key0 < nkeys; key0 += NGRP * U) {
    float kv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) KVElem<KVT>::load8(Kc + (long)min(key0 + u * NGRP, nkeys - 1) * HD + sub * 8, kv[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int key = key0 + u * NGRP;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += qreg[g][j] * kv[u][j];
        s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 4, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 1, 64);
        if (sub == 0 && key < nkeys) sc[g * sc_ld + key] = s * scale;
      }
    }
  }
  __syncthreads();

  // ---- softmax numerators (wave g handles query head g) ------------------------------------------
  if (w < G) {
    float* s = sc + w * sc_ld;
    float m = -INFINITY;
    for (int k = lane; k < nkeys; k += 64) m = fmaxf(m, s[k]);
    m = wave_max(m);
    float l = 0.f;
    for (int k = lane; k < nkeys; k += 64) {
      const float e = expf(s[k] - m);
      s[k] = e;
      l += e;
    }
    l = wave_sum(l);
    if (lane == 0) lsum[w] = l;
  }
  __syncthreads();

  // ---- pass 2: P.V ----------------------------------------------------------------------------
  float acc[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;
  for (int key0 = grp; key0 < nkeys; key0 += NGRP * U) {
    float vv[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) KVElem<KVT>::load8(Vc + (long)min(key0 + u * NGRP, nkeys - 1) * HD + sub * 8, vv[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int key = key0 + u * NGRP;
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float pk = key < nkeys ? sc[g * sc_ld + key] : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[g][j] += pk * vv[u][j];
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = acc[g][j];
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      acc[g][j] = a;
    }
  if (lane < 16) {
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int j = 0; j < 8; ++j) part[(w * G + g) * HD + sub * 8 + j] = acc[g][j];
  }
  __syncthreads();

  // ---- output: G*16 threads, 8 dims each; planes for the o_proj GEMM ---------------------------
  if (tid < G * 16) {
    const int g = tid >> 4, d0 = (tid & 15) * 8;
    const float invl = lsum[g] > 0.f ? 1.0f / lsum[g] : 0.f;   // no keys (empty text) -> 0, like a fully masked row
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = 0.f;
#pragma unroll
      for (int ww = 0; ww < NT / 64; ++ww) a += part[(ww * G + g) * HD + d0 + j];
      o[j] = a * invl;
    }
    const int col = (kvh * G + g) * HD + d0;
    emit_planes8(p.P, p.p_plane_stride, p.p_ktiles, qrow, col, o);
    if (p.mode == DIA_ATTN_CROSS) {
      // the uncond row's cross-attention mask is all False -> SDPA returns 0
      const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      emit_planes8(p.P, p.p_plane_stride, p.p_ktiles, qrow - 1, col, z);
    }
  }
}

__global__ void k_enc_kv_prep(const float* qkv, int ldq, int k_off, int v_off, int heads, int L, int cap,
                              const float* cos_t, const float* sin_t, float* kc, float* vc) {
  // grid (heads, L), 64 threads: thread d handles the RoPE pair (d, d+64)
  const int h = blockIdx.x, m = blockIdx.y, d = threadIdx.x;
  const float* kr = qkv + (long)m * ldq + k_off + h * 128;
  const float* vr = qkv + (long)m * ldq + v_off + h * 128;
  const float x1 = kr[d], x2 = kr[d + 64];
  const float c = cos_t[(long)m * 64 + d], s = sin_t[(long)m * 64 + d];
  float* ko = kc + ((long)h * cap + m) * 128;
  float* vo = vc + ((long)h * cap + m) * 128;
  ko[d] = x1 * c - x2 * s;
  ko[d + 64] = x1 * s + x2 * c;
  vo[d] = vr[d];
  vo[d + 64] = vr[d + 64];
}

template <typename KVT, int G>
int launch_attn(const AttnK& k, int grid_y, int sc_ld, hipStream_t st) {
  size_t smem = sizeof(float) * ((size_t)G * HD + 2 * HD + 8 + (NT / 64) * G * HD + (size_t)G * sc_ld);
  if (smem > 160 * 1024) return dia_fail(DIA_E_ARG, "dia_attn: score buffer exceeds LDS");
  if (smem > 64 * 1024) {
    int rc = dia_kernels_init_once();
    if (rc) return rc;
  }
  hipLaunchKernelGGL((k_attn<KVT, G>), dim3(k.n_kv_heads, grid_y), dim3(NT), smem, st, k);
  return dia_check_launch("k_attn");
}

template <typename KVT, int G>
int set_attr() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn<KVT, G>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  return e == hipSuccess ? DIA_OK : dia_fail_hip(e, "hipFuncSetAttribute(k_attn)");
}

}  // namespace

int dia_attn_init() {
  int rc = 0;
  rc |= set_attr<float, 1>(); rc |= set_attr<float, 2>(); rc |= set_attr<float, 4>();
  rc |= set_attr<bf16_raw, 1>(); rc |= set_attr<bf16_raw, 2>(); rc |= set_attr<bf16_raw, 4>();
  return rc ? DIA_E_HIP : DIA_OK;
}

extern "C" int dia_attn(const dia_attn_args* a, void* stream) {
  if (!a || !a->q || !a->kc || !a->vc || !a->P || !a->cos_t || !a->sin_t) return dia_fail(DIA_E_ARG, "dia_attn: null argument");
  if (a->n_rows <= 0 || a->n_kv_heads <= 0) return dia_fail(DIA_E_ARG, "dia_attn: empty problem");
  if (a->p_plane_stride % 8 != 0) return dia_fail(DIA_E_ARG, "dia_attn: plane stride must be a multiple of 8");
  AttnK k;
  k.mode = a->mode; k.n_kv_heads = a->n_kv_heads; k.n_rows = a->n_rows; k.kv_cap = a->kv_cap;
  k.q = a->q; k.ldq = a->ldq; k.q_off = a->q_off; k.k_off = a->k_off; k.v_off = a->v_off;
  k.kc = a->kc; k.vc = a->vc; k.cur = a->cur; k.len = a->len; k.enc_len = a->enc_len;
  k.cos_t = a->cos_t; k.sin_t = a->sin_t;
  k.P = (bf16_raw*)a->P; k.p_plane_stride = a->p_plane_stride; k.p_ktiles = a->p_ktiles;
  if ((a->n_kv_heads * a->group * 128 + 31) / 32 > a->p_ktiles) return dia_fail(DIA_E_ARG, "dia_attn: output planes too narrow");
  hipStream_t st = (hipStream_t)stream;
  const bool f32 = a->kv_dtype == DIA_KV_F32;
  switch (a->mode) {
    case DIA_ATTN_SELF:
      if (!a->cur) return dia_fail(DIA_E_ARG, "dia_attn: SELF needs cur");
      if (a->group == 4) return f32 ? launch_attn<float, 4>(k, a->n_rows, a->kv_cap, st) : launch_attn<bf16_raw, 4>(k, a->n_rows, a->kv_cap, st);
      if (a->group == 2) return f32 ? launch_attn<float, 2>(k, a->n_rows, a->kv_cap, st) : launch_attn<bf16_raw, 2>(k, a->n_rows, a->kv_cap, st);
      if (a->group == 1) return f32 ? launch_attn<float, 1>(k, a->n_rows, a->kv_cap, st) : launch_attn<bf16_raw, 1>(k, a->n_rows, a->kv_cap, st);
      return dia_fail(DIA_E_ARG, "dia_attn: GQA group must be 1, 2 or 4");
    case DIA_ATTN_CROSS:
      if (!a->cur || !a->len || a->group != 1) return dia_fail(DIA_E_ARG, "dia_attn: CROSS needs cur, len and group 1");
      return f32 ? launch_attn<float, 1>(k, a->n_rows, a->kv_cap, st) : launch_attn<bf16_raw, 1>(k, a->n_rows, a->kv_cap, st);
    case DIA_ATTN_ENC:
      if (a->group != 1 || a->enc_len <= 0 || a->enc_len > a->kv_cap || !f32) return dia_fail(DIA_E_ARG, "dia_attn: ENC needs group 1, fp32 scratch K/V, 0 < L <= cap");
      return launch_attn<float, 1>(k, a->n_rows, a->enc_len, st);
    default:
      return dia_fail(DIA_E_ARG, "dia_attn: unknown mode");
  }
}

extern "C" int dia_enc_kv_prep(const float* qkv, int ldq, int k_off, int v_off, int heads, int L, int cap,
                               const float* cos_t, const float* sin_t, float* kc, float* vc, void* stream) {
  if (!qkv || !kc || !vc || !cos_t || !sin_t || heads <= 0 || L <= 0 || L > cap) return dia_fail(DIA_E_ARG, "dia_enc_kv_prep: bad argument");
  hipLaunchKernelGGL(k_enc_kv_prep, dim3(heads, L), dim3(64), 0, (hipStream_t)stream, qkv, ldq, k_off, v_off, heads, L, cap, cos_t, sin_t, kc, vc);
  return dia_check_launch("k_enc_kv_prep");
}
