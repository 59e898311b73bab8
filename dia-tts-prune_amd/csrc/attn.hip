// Single-query attention for the decode path, head_dim 128, fp32 arithmetic, split over the keys.
//
// Replaces, per reference call site: RotaryEmbedding.forward on q and the new k (dia/layers.py:135-173,
// 278-279), KVCache.update (dia/state.py:99-103), the GQA repeat_interleave (layers.py:319-320, never
// materialised here: one workgroup serves a kv head and its `G` query heads) and
// F.scaled_dot_product_attention (layers.py:329-337).
//
// Grid = (kv head, query row, key split).  A workgroup is 4 waves; every 16-lane group of a wave runs
// its OWN online softmax over its keys (8 dims of K/V per lane: one 16 B load per key and lane for bf16,
// two for fp32; four keys per group per round, all eight loads in flight together), so the key loop has
// no LDS traffic and no barrier.  Keys are dealt in units of 64 (one per wave-round); workgroup z of NZ
// takes units z, z+NZ, ...  At the end the 4 groups of a wave merge by shuffles, the 4 waves through LDS,
// and — when NZ > 1 — the workgroups through a scratch slab: every workgroup publishes {max, sum,
// unnormalised output} and the LAST arriver (agent-scope release / ticket / acquire, guide §6 G16) merges
// the slabs in split order — deterministic regardless of arrival order — and emits.  The grid is static
// (hipGraph): splits beyond the current length exit at once.
// The output leaves as three bf16 planes in MFMA A-operand order for the o_proj GEMM.
#include "common.hpp"
#include "../../include/dia_hip.h"
#include "errors.hpp"
#include <cstdlib>

namespace {

constexpr int HD = 128;
constexpr int NT = 256;            // threads per workgroup (4 waves)
constexpr int NWV = NT / 64;
constexpr int U = 4;               // keys per lane group per round (independent loads in flight)
constexpr int UNIT = NWV * 16;     // 64 keys per workgroup round: 16 per wave, 4 per lane group
constexpr int CHUNK = 128;         // scratch sizing granule: at most one split per 128 keys of capacity
constexpr int SLAB = 2 * 8 + 4 * HD;   // floats per (row, head, chunk): m[G<=4], l[G<=4] (padded to 8 each), o[G][128]

struct AttnK {
  int mode, n_kv_heads, n_rows, kv_cap;
  const float* q; int ldq, q_off, k_off, v_off;
  void* kc; void* vc;
  const int* cur; const int* len; int enc_len;
  const float* cos_t; const float* sin_t;
  bf16_raw* P; long p_plane_stride; int p_ktiles;
  float* scratch; int* tickets; int max_chunks;
  const int* head_map;
};

template <typename KVT, int G>
__global__ __launch_bounds__(NT) void k_attn(AttnK p) {
  __shared__ __attribute__((aligned(16))) float q_s[G * HD];
  __shared__ float part[NWV * G * HD];
  __shared__ float pm_s[NWV * 8], pl_s[NWV * 8];
  __shared__ float knew_s[HD], vnew_s[HD];
  __shared__ int last_s;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kvh = blockIdx.x, chunk = blockIdx.z;
  int qrow, kvrow, pos, nkeys, slot = -1, head_row;
  if (p.mode == DIA_ATTN_SELF) {
    qrow = blockIdx.y; kvrow = qrow;
    const int c = p.cur[qrow >> 1];
    pos = c; nkeys = c; slot = c - 1;
    head_row = qrow;
  } else if (p.mode == DIA_ATTN_CROSS) {
    const int b = blockIdx.y;
    qrow = 2 * b + 1; kvrow = b;
    pos = p.cur[b]; nkeys = p.len[b];
    head_row = b;
  } else {
    qrow = blockIdx.y; kvrow = 0; pos = qrow; nkeys = p.enc_len;
    head_row = qrow;
  }
  const int NZ = gridDim.z;
  const int nunits = max(1, (nkeys + UNIT - 1) / UNIT);
  if (chunk >= nunits) return;                        // uniform: nothing to do for this workgroup yet
  const int nchunks = min(NZ, nunits);                // workgroups that publish a partial for this pair
  if (p.head_map) {                                   // every query head of this kv head pruned: no work
    bool any_live = false;
#pragma unroll
    for (int g = 0; g < G; ++g) any_live |= p.head_map[kvh * G + g] >= 0;
    if (!any_live && !(p.mode == DIA_ATTN_SELF)) return;      // (self: the k/v append below must still happen)
  }

  KVT* Kc = reinterpret_cast<KVT*>(p.kc) + ((long)kvrow * p.n_kv_heads + kvh) * p.kv_cap * HD;
  KVT* Vc = reinterpret_cast<KVT*>(p.vc) + ((long)kvrow * p.n_kv_heads + kvh) * p.kv_cap * HD;
  const int gr = lane >> 4, sub = lane & 15;           // lane group inside the wave, 8-dim slice
  const int klast = max(nkeys - 1, 0);
  float kv[U][8], vv[U][8];
  // unit u, wave w, group gr, round-key uu  ->  key u*64 + w*16 + uu*4 + gr (4 consecutive keys per load instruction)
  auto load_unit = [&](int unit) {
    const int base = unit * UNIT + w * 16 + gr;
#pragma unroll
    for (int u = 0; u < U; ++u) KVElem<KVT>::load8(Kc + (long)min(base + u * 4, klast) * HD + sub * 8, kv[u]);
#pragma unroll
    for (int u = 0; u < U; ++u) KVElem<KVT>::load8(Vc + (long)min(base + u * 4, klast) * HD + sub * 8, vv[u]);
  };
  load_unit(chunk);      // K/V do not depend on q: in flight during the prologue

  // ---- prologue: RoPE(q); the workgroup that owns the new slot also ropes k and appends k, v -------
  const float* qr = p.q + (long)qrow * p.ldq;
  for (int t = tid; t < G * 64; t += NT) {
    const int g = t >> 6, d = t & 63;
    const float* qh = qr + p.q_off + (kvh * G + g) * HD;
    const float x1 = qh[d], x2 = qh[d + 64];
    const float c = p.cos_t[(long)pos * 64 + d], s = p.sin_t[(long)pos * 64 + d];
    q_s[g * HD + d] = x1 * c - x2 * s;
    q_s[g * HD + d + 64] = x1 * s + x2 * c;
  }
  const int slot_unit = slot >= 0 ? slot / UNIT : -1;
  const bool owns_slot = p.mode == DIA_ATTN_SELF && slot_unit >= 0 && (slot_unit % NZ) == chunk;
  if (owns_slot && tid >= NT - 64) {
    const int d = tid - (NT - 64);
    const float* kh = qr + p.k_off + kvh * HD;
    const float* vh = qr + p.v_off + kvh * HD;
    const float x1 = kh[d], x2 = kh[d + 64];
    const float c = p.cos_t[(long)pos * 64 + d], s = p.sin_t[(long)pos * 64 + d];
    const float k1v = KVElem<KVT>::round(x1 * c - x2 * s), k2v = KVElem<KVT>::round(x1 * s + x2 * c);
    const float v1v = KVElem<KVT>::round(vh[d]), v2v = KVElem<KVT>::round(vh[d + 64]);
    knew_s[d] = k1v; knew_s[d + 64] = k2v; vnew_s[d] = v1v; vnew_s[d + 64] = v2v;
    KVElem<KVT>::store(Kc + (long)slot * HD + d, k1v);
    KVElem<KVT>::store(Kc + (long)slot * HD + d + 64, k2v);
    KVElem<KVT>::store(Vc + (long)slot * HD + d, v1v);
    KVElem<KVT>::store(Vc + (long)slot * HD + d + 64, v2v);
  }
  __syncthreads();

  float qreg[G][8];
#pragma unroll
  for (int g = 0; g < G; ++g)
#pragma unroll
    for (int j = 0; j < 8; ++j) qreg[g][j] = q_s[g * HD + sub * 8 + j];
  const float scale = 0.08838834764831845f;   // 1/sqrt(128)
  float acc[G][8], mrun[G], lrun[G];           // this lane group's running softmax state
#pragma unroll
  for (int g = 0; g < G; ++g) {
    mrun[g] = -INFINITY; lrun[g] = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[g][j] = 0.f;
  }

  constexpr bool PREFETCH = G <= 2;            // 64 more VGPRs: fits next to G <= 2 accumulators only
  float kvn[PREFETCH ? U : 1][8], vvn[PREFETCH ? U : 1][8];
  for (int unit = chunk; unit < nunits; unit += NZ) {
    const int nu = unit + NZ;
    if constexpr (PREFETCH) {
      if (nu < nunits) {                       // next unit's K/V stream while this one computes
        const int nb = nu * UNIT + w * 16 + gr;
#pragma unroll
        for (int u = 0; u < U; ++u) KVElem<KVT>::load8(Kc + (long)min(nb + u * 4, klast) * HD + sub * 8, kvn[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) KVElem<KVT>::load8(Vc + (long)min(nb + u * 4, klast) * HD + sub * 8, vvn[u]);
      }
    } else {
      if (unit != chunk) load_unit(unit);
    }
    const int base = unit * UNIT + w * 16 + gr;
    if (owns_slot && unit == slot_unit) {      // loads of the slot written this step are stale: take the LDS copy
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (base + u * 4 == slot) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { kv[u][j] = knew_s[sub * 8 + j]; vv[u][j] = vnew_s[sub * 8 + j]; }
        }
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      float sc[U];
      float mnew = mrun[g];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += qreg[g][j] * kv[u][j];
        s = row16_sum(s);
        sc[u] = (base + u * 4 < nkeys) ? s * scale : -INFINITY;
        mnew = fmaxf(mnew, sc[u]);
      }
      const float alpha = (mrun[g] == -INFINITY) ? 0.f : __expf(mrun[g] - mnew);
      float pu[U], ls = 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) { pu[u] = (sc[u] == -INFINITY) ? 0.f : __expf(sc[u] - mnew); ls += pu[u]; }
      lrun[g] = lrun[g] * alpha + ls;
      mrun[g] = mnew;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = acc[g][j] * alpha;
#pragma unroll
        for (int u = 0; u < U; ++u) a += pu[u] * vv[u][j];
        acc[g][j] = a;
      }
    }
    if constexpr (PREFETCH) {
      if (nu < nunits) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int j = 0; j < 8; ++j) { kv[u][j] = kvn[u][j]; vv[u][j] = vvn[u][j]; }
      }
    }
  }

  // ---- merge the 4 lane groups of the wave (shuffles), then the waves (LDS), both in fixed order -----
#pragma unroll
  for (int g = 0; g < G; ++g) {
    float mw = mrun[g];
    mw = fmaxf(mw, __shfl_xor(mw, 16, 64));
    mw = fmaxf(mw, __shfl_xor(mw, 32, 64));
    const float f = (mrun[g] == -INFINITY) ? 0.f : expf(mrun[g] - mw);
    float lw = lrun[g] * f;
    lw += __shfl_xor(lw, 16, 64);
    lw += __shfl_xor(lw, 32, 64);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float a = acc[g][j] * f;
      a += __shfl_xor(a, 16, 64);
      a += __shfl_xor(a, 32, 64);
      acc[g][j] = a;
    }
    mrun[g] = mw; lrun[g] = lw;
  }
  if (lane < 16) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int j = 0; j < 8; ++j) part[(w * G + g) * HD + sub * 8 + j] = acc[g][j];
      if (sub == 0) { pm_s[w * 8 + g] = mrun[g]; pl_s[w * 8 + g] = lrun[g]; }
    }
  }
  __syncthreads();

  // ---- G*16 threads own 8 output dims each ------------------------------------------------------------
  const bool o_thread = tid < G * 16;
  const int og = tid >> 4, od0 = (tid & 15) * 8;
  float o[8];
  float M = 0.f, Lsum = 1.f;
  if (o_thread) {
    float mm = -INFINITY;
#pragma unroll
    for (int ww = 0; ww < NWV; ++ww) mm = fmaxf(mm, pm_s[ww * 8 + og]);
    float L2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
#pragma unroll
    for (int ww = 0; ww < NWV; ++ww) {
      const float pmw = pm_s[ww * 8 + og];
      const float f = (pmw == -INFINITY) ? 0.f : expf(pmw - mm);
      L2 += pl_s[ww * 8 + og] * f;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += part[(ww * G + og) * HD + od0 + j] * f;
    }
    M = mm; Lsum = L2;
  }

  if (nchunks > 1) {
    // publish this chunk's slab, take a ticket; the last arriver merges
    const long pair = (long)head_row * p.n_kv_heads + kvh;
    float* slab = p.scratch + (pair * p.max_chunks + chunk) * SLAB;
    if (o_thread) {
      if ((tid & 15) == 0) { slab[og] = M; slab[8 + og] = Lsum; }
      *reinterpret_cast<float4*>(slab + 16 + og * HD + od0) = float4{o[0], o[1], o[2], o[3]};
      *reinterpret_cast<float4*>(slab + 16 + og * HD + od0 + 4) = float4{o[4], o[5], o[6], o[7]};
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int ticket = __hip_atomic_fetch_add(p.tickets + pair, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == nchunks - 1;
      if (last) {
        __hip_atomic_store(p.tickets + pair, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      last_s = last;
    }
    __syncthreads();
    if (!last_s) return;
    if (o_thread) {
      const float* base = p.scratch + pair * p.max_chunks * SLAB;
      constexpr int MAXC = 24;               // 3072 keys / 128
      float mc[MAXC];
#pragma unroll
      for (int c = 0; c < MAXC; ++c) mc[c] = base[(long)min(c, nchunks - 1) * SLAB + og];   // independent loads
      float mm = -INFINITY;
#pragma unroll
      for (int c = 0; c < MAXC; ++c) mm = fmaxf(mm, mc[c]);
      for (int c = MAXC; c < nchunks; ++c) mm = fmaxf(mm, base[(long)c * SLAB + og]);        // capacity > 3072 only
      float L2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = 0.f;
      for (int c0 = 0; c0 < nchunks; c0 += 8) {                     // chunk order: deterministic
        float lc[8], fm[8]; float4 oa[8], ob[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float* sl = base + (long)min(c0 + u, nchunks - 1) * SLAB;
          fm[u] = sl[og]; lc[u] = sl[8 + og];
          oa[u] = *reinterpret_cast<const float4*>(sl + 16 + og * HD + od0);
          ob[u] = *reinterpret_cast<const float4*>(sl + 16 + og * HD + od0 + 4);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const float f = (c0 + u < nchunks && fm[u] != -INFINITY) ? expf(fm[u] - mm) : 0.f;
          L2 += lc[u] * f;
          o[0] += oa[u].x * f; o[1] += oa[u].y * f; o[2] += oa[u].z * f; o[3] += oa[u].w * f;
          o[4] += ob[u].x * f; o[5] += ob[u].y * f; o[6] += ob[u].z * f; o[7] += ob[u].w * f;
        }
      }
      Lsum = L2;
    }
  }

  if (o_thread) {
    const float invl = Lsum > 0.f ? 1.0f / Lsum : 0.f;   // no keys (empty text) -> 0, like a fully masked row
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] *= invl;
    const int hpos = p.head_map ? p.head_map[kvh * G + og] : kvh * G + og;   // compacted o_proj input
    const int col = hpos * HD + od0;
    if (hpos >= 0) emit_planes8(p.P, p.p_plane_stride, p.p_ktiles, qrow, col, o);
    if (p.mode == DIA_ATTN_CROSS && hpos >= 0) {
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
int launch_attn(const AttnK& k, int grid_y, int grid_z, hipStream_t st) {
  hipLaunchKernelGGL((k_attn<KVT, G>), dim3(k.n_kv_heads, grid_y, grid_z), dim3(NT), 0, st, k);
  return dia_check_launch("k_attn");
}

}  // namespace

int dia_attn_init() { return DIA_OK; }   // static LDS only since the split-key rewrite

extern "C" int dia_attn_scratch_floats(int n_rows, int n_kv_heads, int kv_cap) {
  const long chunks = (kv_cap + CHUNK - 1) / CHUNK;
  const long n = (long)n_rows * n_kv_heads * chunks * SLAB;
  return n > 0x7fffffffL ? -1 : (int)n;
}

extern "C" int dia_attn(const dia_attn_args* a, void* stream) {
  if (!a || !a->q || !a->kc || !a->vc || !a->P || !a->cos_t || !a->sin_t) return dia_fail(DIA_E_ARG, "dia_attn: null argument");
  if (a->n_rows <= 0 || a->n_kv_heads <= 0) return dia_fail(DIA_E_ARG, "dia_attn: empty problem");
  if (a->p_plane_stride % 8 != 0) return dia_fail(DIA_E_ARG, "dia_attn: plane stride must be a multiple of 8");
  const int cap_keys = a->mode == DIA_ATTN_ENC ? a->enc_len : a->kv_cap;
  const int cap_chunks = (cap_keys + CHUNK - 1) / CHUNK;
  // key-split factor: ~2 workgroups of 4 waves per CU, never more than one split per 128 keys of
  // capacity (the scratch sizing granule), and each split at least one 64-key unit
  int max_chunks = (512 + a->n_rows * a->n_kv_heads - 1) / (a->n_rows * a->n_kv_heads);
  if (const char* e = getenv("DIA_DBG_NZ")) { if (atoi(e) > 0) max_chunks = atoi(e); }
  if (max_chunks > cap_chunks) max_chunks = cap_chunks;
  if (max_chunks < 1) max_chunks = 1;
  if (max_chunks > 1 && (!a->scratch || !a->tickets)) return dia_fail(DIA_E_ARG, "dia_attn: more than 128 keys possible: scratch and tickets are required");
  AttnK k;
  k.mode = a->mode; k.n_kv_heads = a->n_kv_heads; k.n_rows = a->n_rows; k.kv_cap = a->kv_cap;
  k.q = a->q; k.ldq = a->ldq; k.q_off = a->q_off; k.k_off = a->k_off; k.v_off = a->v_off;
  k.kc = a->kc; k.vc = a->vc; k.cur = a->cur; k.len = a->len; k.enc_len = a->enc_len;
  k.cos_t = a->cos_t; k.sin_t = a->sin_t;
  k.P = (bf16_raw*)a->P; k.p_plane_stride = a->p_plane_stride; k.p_ktiles = a->p_ktiles;
  k.head_map = a->head_map;
  k.scratch = a->scratch; k.tickets = a->tickets; k.max_chunks = (a->kv_cap + CHUNK - 1) / CHUNK;
  if ((a->n_kv_heads * a->group * 128 + 31) / 32 > a->p_ktiles) return dia_fail(DIA_E_ARG, "dia_attn: output planes too narrow");
  hipStream_t st = (hipStream_t)stream;
  const bool f32 = a->kv_dtype == DIA_KV_F32;
  switch (a->mode) {
    case DIA_ATTN_SELF:
      if (!a->cur) return dia_fail(DIA_E_ARG, "dia_attn: SELF needs cur");
      if (a->group == 4) return f32 ? launch_attn<float, 4>(k, a->n_rows, max_chunks, st) : launch_attn<bf16_raw, 4>(k, a->n_rows, max_chunks, st);
      if (a->group == 2) return f32 ? launch_attn<float, 2>(k, a->n_rows, max_chunks, st) : launch_attn<bf16_raw, 2>(k, a->n_rows, max_chunks, st);
      if (a->group == 1) return f32 ? launch_attn<float, 1>(k, a->n_rows, max_chunks, st) : launch_attn<bf16_raw, 1>(k, a->n_rows, max_chunks, st);
      return dia_fail(DIA_E_ARG, "dia_attn: GQA group must be 1, 2 or 4");
    case DIA_ATTN_CROSS:
      if (!a->cur || !a->len || a->group != 1) return dia_fail(DIA_E_ARG, "dia_attn: CROSS needs cur, len and group 1");
      return f32 ? launch_attn<float, 1>(k, a->n_rows, max_chunks, st) : launch_attn<bf16_raw, 1>(k, a->n_rows, max_chunks, st);
    case DIA_ATTN_ENC:
      if (a->group != 1 || a->enc_len <= 0 || a->enc_len > a->kv_cap || !f32) return dia_fail(DIA_E_ARG, "dia_attn: ENC needs group 1, fp32 scratch K/V, 0 < L <= cap");
      return launch_attn<float, 1>(k, a->n_rows, max_chunks, st);
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
