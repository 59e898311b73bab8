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
#include "launch.hpp"
#include "tuning.hpp"
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
  int v_blocked;
  int gpw;          // MFMA kernel: granules a wave takes before the keys are split over another workgroup
  int act_f32;      // output as fp32 activation tiles instead of three bf16 planes
  long kv_plane_stride;   // DIA_KV_BF16X2: elements between the hi and the lo plane of a cache (0 = one plane)
};

#ifdef DIA_DBG_STAMPS
__device__ long long g_astamps[8192 * 8];
#define ASTAMP(i) do { if (threadIdx.x == 0) g_astamps[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define ASTAMP(i) do {} while (0)
#endif

// Shared tail of both attention kernels: merge the NWV per-wave partials (LDS, fixed order), then —
// when the pair was split over several workgroups — publish to the slab, take a ticket, and let the
// last arriver merge all splits in split order; finally normalise and emit the planes.
template <int G>
__device__ __forceinline__ void attn_finish(const AttnK& p, const float* part, const float* pm_s, const float* pl_s,
                                            int* last_s_ptr, int tid, int qrow, int kvh, int head_row, int chunk, int nchunks) {
  int& last_s = *last_s_ptr;
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
    // Publish this chunk's slab, take a ticket; the last arriver merges.  The hand-off uses agent-scope
    // RELAXED atomics (sc1 stores/loads: written through to / read from the device coherence point) with the
    // ordering made explicit — stores acknowledged (vmcnt 0) before the ticket is taken, slab loads issued
    // after the ticket came back — instead of release/acquire fences: on gfx950 those are a whole-L2
    // write-back (buffer_wbl2) and invalidate (buffer_inv) per workgroup, microseconds under load.
    const long pair = (long)head_row * p.n_kv_heads + kvh;
    float* slab = p.scratch + (pair * p.max_chunks + chunk) * SLAB;
    if (o_thread) {
      if ((tid & 15) == 0) st2_agent(slab + 2 * og, M, Lsum);
      const __amdgpu_buffer_rsrc_t sr = agent_rsrc(p.scratch);
      const int so = (int)(((pair * p.max_chunks + chunk) * SLAB + 16 + og * HD + od0) * 4);
      st4_agent(sr, so, f32x4{o[0], o[1], o[2], o[3]});
      st4_agent(sr, so + 16, f32x4{o[4], o[5], o[6], o[7]});
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const int ticket = __hip_atomic_fetch_add(p.tickets + pair, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == nchunks - 1;
      if (last) __hip_atomic_store(p.tickets + pair, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      last_s = last;
    }
    __syncthreads();
    ASTAMP(5);
    if (!last_s) return;
    if (o_thread) {
      // the last arriver merges the slabs in chunk order (deterministic).  Only the LIVE slabs are read, MB per round
      // trip, every load of a round issued before the first is waited for: (m, l) as one 8-byte and the 8 output dims
      // as two 16-byte coherent loads per slab (72 eight-byte loads per lane for 24 slabs, live or not, took 2.8 us of
      // an 8.5 us launch at batch 1)
      const float* base = p.scratch + pair * p.max_chunks * SLAB;
      const __amdgpu_buffer_rsrc_t sr = agent_rsrc(p.scratch);
      const int bo = (int)((pair * p.max_chunks * SLAB + 16 + og * HD + od0) * 4);
      constexpr int MB = 12;                 // slabs per round trip
      float mrun = -INFINITY, L2 = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = 0.f;
      for (int c0 = 0; c0 < nchunks; c0 += MB) {
        float2 ml[MB]; f32x4 oa[MB], ob[MB];
#pragma unroll
        for (int u = 0; u < MB; ++u)
          if (c0 + u < nchunks) {            // (uniform over the workgroup)
            const float* sl = base + (long)(c0 + u) * SLAB;
            ml[u] = ld2_agent(sl + 2 * og);
            oa[u] = ld4_agent(sr, bo + (c0 + u) * (SLAB * 4));
            ob[u] = ld4_agent(sr, bo + (c0 + u) * (SLAB * 4) + 16);
          }
        float mm = mrun;
#pragma unroll
        for (int u = 0; u < MB; ++u)
          if (c0 + u < nchunks) mm = fmaxf(mm, ml[u].x);
        const float fr = (mrun == -INFINITY) ? 0.f : expf(mrun - mm);      // earlier rounds (capacity > 1536 keys only)
        L2 *= fr;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] *= fr;
#pragma unroll
        for (int u = 0; u < MB; ++u)
          if (c0 + u < nchunks) {
            const float f = (ml[u].x != -INFINITY) ? expf(ml[u].x - mm) : 0.f;
            L2 += ml[u].y * f;
#pragma unroll
            for (int j = 0; j < 4; ++j) { o[j] += oa[u][j] * f; o[4 + j] += ob[u][j] * f; }
          }
        mrun = mm;
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
    auto emit = [&](int row, const float* v8) {       // the o-projection's A operand: three planes, or fp32 tiles (common.hpp)
      if (p.act_f32) emit_f32x8(reinterpret_cast<float*>(p.P), p.p_ktiles, row, col, v8);
      else emit_planes8(p.P, p.p_plane_stride, p.p_ktiles, row, col, v8);
    };
    if (hpos >= 0) emit(qrow, o);
    if (p.mode == DIA_ATTN_CROSS && hpos >= 0) {
      // the uncond row's cross-attention mask is all False -> SDPA returns 0
      const float z[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      emit(qrow - 1, z);
    }
  }
  ASTAMP(6);
}

template <typename KVT, int G>
__global__ __launch_bounds__(NT) void k_attn(AttnK p) {
  __shared__ __attribute__((aligned(16))) float q_s[G * HD];
  __shared__ float part[NWV * G * HD];
  __shared__ float pm_s[NWV * 8], pl_s[NWV * 8];
  __shared__ float knew_s[HD], vnew_s[HD];
  __shared__ int last_s;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kvh = blockIdx.x, chunk = blockIdx.z;
  // mode decode, branch-free (cur/len are never null here: dia_attn substitutes a readable dummy).
  // History (round 1, gpurun_out/t5-t8.log): while this tail was being factored out into attn_finish, an UNCOMMITTED
  // intermediate of the then branchy decode (`int qrow, kvrow, pos, nkeys, slot = -1, head_row;` assigned per mode
  // branch) left a row index unset on the ENC path.  test_attn_encoder_mode_and_kv_prep faulted on a non-null heap
  // address just past an allocation (an out-of-range row into q / P, not a null cur / len), the next build computed
  // wrong values without faulting (max error 1.74), and the committed states on either side (665b5d6, 8e4f8ca) pass:
  // a source-level slip of the work in progress, not a compiler problem.  Since then the decode is a set of selects,
  // every variable is const-initialised, and dia_attn checks n_rows / enc_len / kv_cap / rope_rows on the host.
  //   SELF  row = grid y, kv row = row, position = cur[row/2], keys = cur, new slot = cur-1
  //   CROSS utterance b = grid y, row = 2b+1 (cond), kv row = b, position = cur[b], keys = len[b]
  //   ENC   row = grid y, kv row = 0, position = row, keys = enc_len
  const int by = blockIdx.y;
  const bool is_self = p.mode == DIA_ATTN_SELF, is_cross = p.mode == DIA_ATTN_CROSS;
  const int c_cur = p.cur[is_self ? (by >> 1) : (is_cross ? by : 0)];
  const int c_len = p.len[is_cross ? by : 0];
  const int qrow = is_cross ? 2 * by + 1 : by;
  const int kvrow = (is_self || is_cross) ? by : 0;
  const int pos = (is_self || is_cross) ? c_cur : by;
  const int nkeys = is_self ? c_cur : (is_cross ? c_len : p.enc_len);
  const int slot = is_self ? c_cur - 1 : -1;
  const int head_row = by;
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

  attn_finish<G>(p, part, pm_s, pl_s, &last_s, tid, qrow, kvh, head_row, chunk, nchunks);
}

__device__ __forceinline__ float row16_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
  return v;
}

// bf16 K/V caches, MFMA arithmetic.  Keys are dealt to WAVES in granules of 32: granule g goes to wave
// g mod (4 * live workgroups), so the work is balanced to one granule whatever the length.  Per granule:
//   S = Q.K^T   A = q as three bf16 planes (exact fp32 q), rows = query heads; B = K rows straight from the
//               cache ([key][128], 16 B per lane: key l&15, dims 32*ks + 8*(l>>4)..); 2 key tiles x 4 k-steps x
//               3 planes = 24 MFMAs into two 16x16 score tiles (lanes 0..15 = keys, registers = heads)
//   online softmax on those tiles (row-of-16 DPP reductions), p written to LDS as hi+lo bf16 and read
//               back in A-operand order (rows = heads, k = 32 keys)
//   O += P.V    B = V from the BLOCKED cache layout [key/32][128 dims][32 keys] (16 B per lane: dim l&15 of
//               dim block nb, keys 8*(l>>4)..): 8 dim blocks x 2 planes = 16 MFMAs
// Latency plan (the kernel is a chain of dependent round trips, not a throughput problem): the q / new-k /
// new-v loads do not depend on cur[] and are issued first; then cur -> cos/sin and the first granule's K/V
// (vmcnt retires in order, so the small loads go ahead of the 16 KB granule); RoPE and the LDS hand-off run
// under the K/V latency; the next granule is prefetched before the current one is consumed.  The wave that
// owns the granule of the slot written this step appends k/v itself and patches its fragments from LDS
// instead of waiting for its own store.  V of keys beyond the length is multiplied by p == 0 exactly:
// the caches must hold finite values (the engine allocates them zeroed).
#ifdef DIA_DBG_KV_NT
#define KVLOAD(ptr) __builtin_nontemporal_load(ptr)
#else
#define KVLOAD(ptr) (*(ptr))
#endif
// PL = bf16 planes per cache: 1 = bf16 K/V (the reference's GPU cache precision), 2 = DIA_KV_BF16X2 — every K / V value is kept
// as hi + lo bf16 (16 significand bits, the bytes of an fp32 cache; written split by the producers), so the MFMA kernel
// serves the parity configuration too: q (3 planes, exact) x K (2 planes) and p (2 planes) x V (2 planes), fp32 accumulation.
// Two planes take the registers of both granule buffers: one granule at a time, no prefetch behind the one being consumed.
template <int G, int PL>
struct MfmaFrag { bf16x8 kb[PL][2][4]; bf16x8 vb[PL][8]; };

template <int G, int PL = 1>
__global__ __launch_bounds__(NT, 2) void k_attn_mfma(AttnK p) {
  __shared__ __attribute__((aligned(16))) bf16_raw qf[4][DIA_NPLANES][G][4][8];   // q planes in A-fragment order
  __shared__ float part[NWV * G * HD];
  __shared__ float pm_s[NWV * 8], pl_s[NWV * 8];
  __shared__ __attribute__((aligned(16))) bf16_raw pbuf[NWV][2][16][32];
  __shared__ __attribute__((aligned(16))) bf16_raw knew_s[PL][HD], vnew_s[PL][HD];
  __shared__ int last_s;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kvh = blockIdx.x, chunk = blockIdx.z;
  const int arow = lane & 15, akq = lane >> 4;
  ASTAMP(0);
  // SELF or CROSS only (see k_attn for the decode)
  const int by = blockIdx.y;
  const bool is_self = p.mode == DIA_ATTN_SELF;
  const int qrow = is_self ? by : 2 * by + 1;
  // ---- loads that do not need cur[]: this wave's q head (waves >= G re-read head G-1, branch-free) and
  //      the new k / v of the kv head (CROSS: re-reads q, unused)
  const float* qr = p.q + (long)qrow * p.ldq;
  const float* qh = qr + p.q_off + (kvh * G + min(w, G - 1)) * HD;
  const float* kh = qr + (is_self ? p.k_off : p.q_off) + kvh * HD;
  const float* vh = qr + (is_self ? p.v_off : p.q_off) + kvh * HD;
  const float qx1 = qh[lane], qx2 = qh[lane + 64];
  const float kx1 = kh[lane], kx2 = kh[lane + 64];
  const float vx1 = vh[lane], vx2 = vh[lane + 64];

  const int c_cur = p.cur[is_self ? (by >> 1) : by];
  const int c_len = p.len[is_self ? 0 : by];
  const int kvrow = by, pos = c_cur;
  const int nkeys = is_self ? c_cur : c_len;
  const int slot = is_self ? c_cur - 1 : -1;
  const int head_row = by;
  const int NZ = gridDim.z;
  const int ngran = (nkeys + 31) >> 5;
  // up to 256 keys stay in ONE workgroup (two granules per wave, the second prefetched): no slab hand-off at all,
  // 7.1 vs 8.1 us at 200 keys; beyond that the keys are split, one granule per wave and round
  const int gpw = (ngran <= 2 * NWV) ? max(p.gpw, 2) : p.gpw;
  const int nchunks = min(NZ, max(1, (ngran + NWV * gpw - 1) / (NWV * gpw)));   // workgroups that take keys
  if (chunk >= nchunks) return;
  ASTAMP(1);
  if (p.head_map) {
    bool any_live = false;
#pragma unroll
    for (int g = 0; g < G; ++g) any_live |= p.head_map[kvh * G + g] >= 0;
    if (!any_live && !is_self) return;
  }
  const float rc = p.cos_t[(long)pos * 64 + lane], rs = p.sin_t[(long)pos * 64 + lane];

  bf16_raw* Kc = reinterpret_cast<bf16_raw*>(p.kc) + ((long)kvrow * p.n_kv_heads + kvh) * p.kv_cap * HD;
  bf16_raw* Vc = reinterpret_cast<bf16_raw*>(p.vc) + ((long)kvrow * p.n_kv_heads + kvh) * p.kv_cap * HD;
  const int klast = max(nkeys - 1, 0);
  const int gstride = NWV * nchunks;
  const int g_first = chunk * NWV + w;
  auto load_gran = [&](MfmaFrag<G, PL>& f, int gi) {
    const int key0 = gi << 5;
    // all K planes first (S = Q.K^T runs before P.V, and vmcnt retires in order), then the V planes
#pragma unroll
    for (int pl = 0; pl < PL; ++pl) {
      const bf16_raw* Kp = Kc + pl * p.kv_plane_stride;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
          f.kb[pl][t][ks] = KVLOAD(reinterpret_cast<const bf16x8*>(Kp + (long)min(key0 + 16 * t + arow, klast) * HD + 32 * ks + 8 * akq));
    }
#pragma unroll
    for (int pl = 0; pl < PL; ++pl) {
      const bf16_raw* Vblk = Vc + pl * p.kv_plane_stride + (long)gi * HD * 32;
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) f.vb[pl][nb] = KVLOAD(reinterpret_cast<const bf16x8*>(Vblk + (long)(16 * nb + arow) * 32 + 8 * akq));
    }
  };
  MfmaFrag<G, PL> fa, fb;        // (fb: one-plane caches only — two planes take the registers of both buffers)
  const bool have0 = g_first < ngran;
  load_gran(fa, have0 ? g_first : 0);          // idle waves re-read granule 0 (cache hit) and never use it

  // ---- RoPE(q) -> LDS; the wave that owns the new slot's granule ropes k, appends k (row layout) and
  //      v (blocked layout) and keeps a bf16 copy in LDS for its own fragments
  if (w < G) {      // dim D of head w -> k-step D>>5, lane quarter (D>>3)&3, element D&7 of row w
    const float qv[2] = {qx1 * rc - qx2 * rs, qx1 * rs + qx2 * rc};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int D = lane + 64 * e;
      __bf16 a, b, c;
      split3(qv[e], a, b, c);
      qf[D >> 5][0][w][(D >> 3) & 3][D & 7] = *reinterpret_cast<const bf16_raw*>(&a);
      qf[D >> 5][1][w][(D >> 3) & 3][D & 7] = *reinterpret_cast<const bf16_raw*>(&b);
      qf[D >> 5][2][w][(D >> 3) & 3][D & 7] = *reinterpret_cast<const bf16_raw*>(&c);
    }
  }
  const int gslot = slot >> 5;
  const bool owner = slot >= 0 && (gslot % gstride) == g_first;      // wave-uniform
  if (owner) {
    float kv4[4] = {kx1 * rc - kx2 * rs, kx1 * rs + kx2 * rc, vx1, vx2};      // k dims lane, lane + 64 (roped); v dims lane, lane + 64
#pragma unroll
    for (int pl = 0; pl < PL; ++pl) {       // plane pl = bf16 of what the planes before it left over (hi, then lo)
      bf16_raw r4[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const __bf16 b = (__bf16)kv4[e];
        r4[e] = *reinterpret_cast<const bf16_raw*>(&b);
        kv4[e] -= (float)b;
      }
      knew_s[pl][lane] = r4[0]; knew_s[pl][lane + 64] = r4[1]; vnew_s[pl][lane] = r4[2]; vnew_s[pl][lane + 64] = r4[3];
      bf16_raw* Kp = Kc + pl * p.kv_plane_stride;
      Kp[(long)slot * HD + lane] = r4[0]; Kp[(long)slot * HD + lane + 64] = r4[1];
      bf16_raw* vdst = Vc + pl * p.kv_plane_stride + (long)gslot * HD * 32 + (slot & 31);
      vdst[(long)lane * 32] = r4[2]; vdst[(long)(lane + 64) * 32] = r4[3];
    }
  }
  lds_barrier();     // q (and the owner's k/v copy) in LDS; global loads stay in flight
  ASTAMP(2);

  // q fragments are re-read from LDS per k-step (rows >= G alias row G-1: their score rows are unused)
  const int qrow_l = min(arow, G - 1);
  const float scale = 0.08838834764831845f;
  f32x4 O[8];
#pragma unroll
  for (int nb = 0; nb < 8; ++nb) O[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mrun[4], lrun[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) { mrun[g] = -INFINITY; lrun[g] = 0.f; }

  auto consume = [&](MfmaFrag<G, PL>& f, int gi) {
    const int key0 = gi << 5;
    if (owner && gi == gslot) {                 // the slot written this step: fragments from the LDS copy
      const int ts = (slot >> 4) & 1, ns = slot & 15, js = slot & 7, kqs = (slot & 31) >> 3;
#pragma unroll
      for (int pl = 0; pl < PL; ++pl) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const bf16x8 kn = *reinterpret_cast<const bf16x8*>(&knew_s[pl][32 * ks + 8 * akq]);
#pragma unroll
          for (int t = 0; t < 2; ++t)
            if (t == ts && arow == ns) f.kb[pl][t][ks] = kn;
        }
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) {
          const bf16_raw vr = vnew_s[pl][16 * nb + arow];
          const __bf16 vn = *reinterpret_cast<const __bf16*>(&vr);
#pragma unroll
          for (int j = 0; j < 8; ++j)
            if (j == js && akq == kqs) f.vb[pl][nb][j] = vn;
        }
      }
    }
    f32x4 S[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int pl = 0; pl < DIA_NPLANES; ++pl) {
        const bf16x8 qa = *reinterpret_cast<const bf16x8*>(&qf[ks][pl][qrow_l][akq][0]);
        // two K planes: q_hi.(K_hi + K_lo) + q_mid.K_hi — the products left out (q_mid.K_lo, q_lo.K) are below 2^-16 of the
        // score, the resolution two K planes have anyway; one plane: all three q planes (exact q against bf16 K)
#pragma unroll
        for (int kp = 0; kp < PL; ++kp) {
          if (PL == 2 && (pl + kp > 1)) continue;
#pragma unroll
          for (int t = 0; t < 2; ++t)
            S[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa, f.kb[kp][t][ks], S[t], 0, 0, 0);
        }
      }
    // online softmax: in lanes 0..15, S[t][g] is the score of head g against key key0 + 16t + lane
    float alpha[4];
    const bool v0 = key0 + arow < nkeys, v1 = key0 + 16 + arow < nkeys;
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float s0 = v0 ? S[0][g] * scale : -INFINITY, s1 = v1 ? S[1][g] * scale : -INFINITY;
      const float mnew = fmaxf(mrun[g], row16_max(fmaxf(s0, s1)));
      alpha[g] = (mrun[g] == -INFINITY) ? 0.f : __expf(mrun[g] - mnew);
      const float p0 = (s0 == -INFINITY) ? 0.f : __expf(s0 - mnew), p1 = (s1 == -INFINITY) ? 0.f : __expf(s1 - mnew);
      lrun[g] = lrun[g] * alpha[g] + row16_sum(p0 + p1);
      mrun[g] = mnew;
      if (lane < 16) {      // p -> LDS as hi + lo bf16, [plane][head][key]
        const __bf16 h0 = (__bf16)p0, h1 = (__bf16)p1;
        const __bf16 l0 = (__bf16)(p0 - (float)h0), l1 = (__bf16)(p1 - (float)h1);
        pbuf[w][0][g][lane] = *reinterpret_cast<const bf16_raw*>(&h0);
        pbuf[w][0][g][16 + lane] = *reinterpret_cast<const bf16_raw*>(&h1);
        pbuf[w][1][g][lane] = *reinterpret_cast<const bf16_raw*>(&l0);
        pbuf[w][1][g][16 + lane] = *reinterpret_cast<const bf16_raw*>(&l1);
      }
    }
    __builtin_amdgcn_wave_barrier();
    // rescale what is accumulated (register r of lanes 0..15 = head r)
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int g = 0; g < G; ++g) O[nb][g] *= alpha[g];
    const bf16x8 pa0 = *reinterpret_cast<const bf16x8*>(&pbuf[w][0][arow][8 * akq]);
    const bf16x8 pa1 = *reinterpret_cast<const bf16x8*>(&pbuf[w][1][arow][8 * akq]);
#pragma unroll
    for (int nb = 0; nb < 8; ++nb)
#pragma unroll
      for (int vp = 0; vp < PL; ++vp) {
        O[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa0, f.vb[vp][nb], O[nb], 0, 0, 0);
        if (vp == 0) O[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa1, f.vb[vp][nb], O[nb], 0, 0, 0);     // (p_lo.V_lo: below 2^-16)
      }
    __builtin_amdgcn_wave_barrier();     // pbuf is rewritten by the next granule
  };

  // two granules per trip: the other register set is loading while one is consumed.  Steady state first — three granules
  // known to exist, NO branch around a prefetch: behind an `if` the waits of the consuming MFMAs have to serve the path
  // without the new loads too, the compiler then counts as if they had not been issued and every granule waited for the
  // one requested just before it (no overlap at all; the GEMM kernels had the same: gemm.hip DIA_PREFETCH_CLAMP) — then a
  // tail of at most two granules in the old, branchy form.
  if constexpr (PL == 1) {
    int gi = g_first;
    for (; gi + 2 * gstride < ngran; gi += 2 * gstride) {
      load_gran(fb, gi + gstride);
      consume(fa, gi);
      load_gran(fa, gi + 2 * gstride);
      consume(fb, gi + gstride);
    }
    if (gi < ngran) {                // fa holds granule gi
      const int g1 = gi + gstride;
      if (g1 < ngran) load_gran(fb, g1);
      consume(fa, gi);
      if (g1 < ngran) consume(fb, g1);
    }
  } else {
    for (int gi = g_first; gi < ngran; gi += gstride) {
      if (gi != g_first) load_gran(fa, gi);
      consume(fa, gi);
    }
  }
  ASTAMP(3);
  // per-wave partial -> LDS (lanes 0..15: dim 16*nb + lane, register g = head)
  if (lane < 16) {
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int nb = 0; nb < 8; ++nb) part[(w * G + g) * HD + 16 * nb + lane] = O[nb][g];
      if (lane == 0) { pm_s[w * 8 + g] = mrun[g]; pl_s[w * 8 + g] = lrun[g]; }
    }
  }
  __syncthreads();
  ASTAMP(4);
  attn_finish<G>(p, part, pm_s, pl_s, &last_s, tid, qrow, kvh, head_row, chunk, nchunks);
}

// ---------------------------------------------------------------------------------------------------
// Encoder self-attention for the packed prefill batch (layers.py:385-462: bidirectional over the L non-pad
// positions of each utterance, RoPE positions = byte indices).  Two launches per layer for ALL utterances:
//   k_enc_kv_planes   K = RoPE(k) and V of every packed row as three bf16 planes each (hi+mid+lo == fp32
//                     exactly): K in row layout [plane][head][row][128], V blocked [plane][head][row/32][128][32]
//   k_attn_enc_mfma   grid (head, 16-row query tile): S = Q.K^T and O = P.V on MFMA, every operand as three
//                     planes -> 9 MFMAs per 16x16x32 product, fp32-exact products; 4 waves split the keys in
//                     32-key granules, online softmax per query row (rows of a wave's 16x16 tiles live in the
//                     four 16-lane groups), waves merged through LDS.
// Rows are addressed through row_b[] (utterance of a packed row, -1 = padding), seg_off[] / seg_len[].
struct EncAttnK {
  const float* qkv; int ldq, q_off, k_off, v_off, heads, rows;
  const int* row_b; const int* seg_off; const int* seg_len;
  const float* cos_t; const float* sin_t;
  bf16_raw* kp; bf16_raw* vp;          // [3][heads][rows][128], [3][heads][rows/32][128][32]
  bf16_raw* P; long p_plane_stride; int p_ktiles;
};

__global__ __launch_bounds__(256) void k_enc_kv_planes(EncAttnK p) {
  // grid (heads, rows/32): one 32-row key block of one head.  One pass: every load of the thread is requested up front (the first form
  // walked 24 loop rounds of dependent 4-byte loads and 2-byte stores: 11.7 us per launch at any size), stores are 16 bytes per plane
  const int h = blockIdx.x, blk = blockIdx.y, tid = threadIdx.x;
  const long plane = (long)p.heads * p.rows * HD;
  bf16_raw* kp = p.kp + ((long)h * p.rows + blk * 32) * HD;
  bf16_raw* vp = p.vp + ((long)h * (p.rows >> 5) + blk) * HD * 32;
  // K: thread (row r, dims d0..d0+7 and their RoPE partners d0+64..): rows of padding (row_b < 0) are still rows of the qkv buffer
  const int r = tid >> 3, d0 = (tid & 7) * 8, m = blk * 32 + r;
  const float* kr = p.qkv + (long)m * p.ldq + p.k_off + h * HD + d0;
  const float4 xa0 = *reinterpret_cast<const float4*>(kr), xa1 = *reinterpret_cast<const float4*>(kr + 4);
  const float4 xb0 = *reinterpret_cast<const float4*>(kr + 64), xb1 = *reinterpret_cast<const float4*>(kr + 68);
  // V: two items per thread, item (dim d, keys r0..r0+7) -> one 16-byte store per plane at [d][r0]
  float vv[2][8];
  int4 vb[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int it = tid + 256 * j, d = it >> 2, r0 = (it & 3) * 8;
    const float* vr = p.qkv + (long)(blk * 32 + r0) * p.ldq + p.v_off + h * HD + d;
#pragma unroll
    for (int e = 0; e < 8; ++e) vv[j][e] = vr[(long)e * p.ldq];
    vb[j][0] = *reinterpret_cast<const int4*>(p.row_b + blk * 32 + r0);
    vb[j][1] = *reinterpret_cast<const int4*>(p.row_b + blk * 32 + r0 + 4);
  }
  const int b = p.row_b[m];
  const int pos = b >= 0 ? m - p.seg_off[b] : 0;
  const float* ct = p.cos_t + (long)pos * 64 + d0;
  const float* st = p.sin_t + (long)pos * 64 + d0;
  const float4 c0 = *reinterpret_cast<const float4*>(ct), c1 = *reinterpret_cast<const float4*>(ct + 4);
  const float4 s0 = *reinterpret_cast<const float4*>(st), s1 = *reinterpret_cast<const float4*>(st + 4);
  {
    const float x1[8] = {xa0.x, xa0.y, xa0.z, xa0.w, xa1.x, xa1.y, xa1.z, xa1.w};
    const float x2[8] = {xb0.x, xb0.y, xb0.z, xb0.w, xb1.x, xb1.y, xb1.z, xb1.w};
    const float c[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
    const float s[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    bf16x8 k1[3], k2[3];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float a1 = b >= 0 ? x1[e] * c[e] - x2[e] * s[e] : 0.f;
      const float a2 = b >= 0 ? x1[e] * s[e] + x2[e] * c[e] : 0.f;
      __bf16 u0, u1, u2;
      split3(a1, u0, u1, u2); k1[0][e] = u0; k1[1][e] = u1; k1[2][e] = u2;
      split3(a2, u0, u1, u2); k2[0][e] = u0; k2[1][e] = u1; k2[2][e] = u2;
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      *reinterpret_cast<bf16x8*>(kp + pl * plane + (long)r * HD + d0) = k1[pl];
      *reinterpret_cast<bf16x8*>(kp + pl * plane + (long)r * HD + d0 + 64) = k2[pl];
    }
  }
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int it = tid + 256 * j, d = it >> 2, r0 = (it & 3) * 8;
    const int rb[8] = {vb[j][0].x, vb[j][0].y, vb[j][0].z, vb[j][0].w, vb[j][1].x, vb[j][1].y, vb[j][1].z, vb[j][1].w};
    bf16x8 v3[3];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      __bf16 u0, u1, u2;
      split3(rb[e] >= 0 ? vv[j][e] : 0.f, u0, u1, u2);
      v3[0][e] = u0; v3[1][e] = u1; v3[2][e] = u2;
    }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<bf16x8*>(vp + pl * plane + d * 32 + r0) = v3[pl];
  }
}

__global__ __launch_bounds__(NT) void k_attn_enc_mfma(EncAttnK p) {
  __shared__ __attribute__((aligned(16))) bf16_raw qf[4][DIA_NPLANES][16][4][8];     // q planes, A-fragment order
  __shared__ __attribute__((aligned(16))) bf16_raw pbuf[NWV][DIA_NPLANES][16][32];   // p planes per wave
  __shared__ float part[NWV * 16 * HD];
  __shared__ float pm_s[NWV * 16], pl_s[NWV * 16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int h = blockIdx.x, qt = blockIdx.y;
  const int arow = lane & 15, akq = lane >> 4;
  const int b = p.row_b[qt * 16];
  if (b < 0) return;                                   // a tile of padding rows (tiles never straddle utterances)
  const int off = p.seg_off[b], len = p.seg_len[b];
  const long plane = (long)p.heads * p.rows * HD;

  // ---- RoPE(q) of the 16 rows -> three planes in LDS: thread (row = tid>>4, 8 dim pairs)
  {
    const int r = tid >> 4, m = qt * 16 + r, pos = min(m - off, len - 1);       // rows past the end: recomputed, never emitted
    const float* qh = p.qkv + (long)min(m, off + len - 1) * p.ldq + p.q_off + h * HD;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int d = (tid & 15) * 4 + i;
      const float x1 = qh[d], x2 = qh[d + 64];
      const float c = p.cos_t[(long)pos * 64 + d], s = p.sin_t[(long)pos * 64 + d];
      const float qv[2] = {x1 * c - x2 * s, x1 * s + x2 * c};
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
  float mrun[4], lrun[4];                               // rows 4*akq + r of this lane group
#pragma unroll
  for (int r = 0; r < 4; ++r) { mrun[r] = -INFINITY; lrun[r] = 0.f; }
  const bf16_raw* Kh = p.kp + ((long)h * p.rows + off) * HD;
  const bf16_raw* Vh = p.vp + ((long)h * (p.rows >> 5) + (off >> 5)) * HD * 32;
  const int ngran = (len + 31) >> 5;
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
        const bf16_raw* kr = Kh + (long)(key0 + 16 * t + arow) * HD + 32 * ks + 8 * akq;   // rows < seg padded to 32: in bounds
#pragma unroll
        for (int kpl = 0; kpl < DIA_NPLANES; ++kpl) {
          const bf16x8 kb = *reinterpret_cast<const bf16x8*>(kr + kpl * plane);
#pragma unroll
          for (int pl = 0; pl < DIA_NPLANES; ++pl) S[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[pl], kb, S[t], 0, 0, 0);
        }
      }
    }
    // online softmax per query row: S[t][r] = score(row 4*akq + r, key key0 + 16t + arow)
    float alpha[4];
    const bool v0 = key0 + arow < len, v1 = key0 + 16 + arow < len;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float s0 = v0 ? S[0][r] * scale : -INFINITY, s1 = v1 ? S[1][r] * scale : -INFINITY;
      const float mnew = fmaxf(mrun[r], row16_max(fmaxf(s0, s1)));
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
      const bf16_raw* vr = Vblk + (long)(16 * nb + arow) * 32 + 8 * akq;
#pragma unroll
      for (int vpl = 0; vpl < DIA_NPLANES; ++vpl) {
        const bf16x8 vb = *reinterpret_cast<const bf16x8*>(vr + vpl * plane);
#pragma unroll
        for (int pl = 0; pl < DIA_NPLANES; ++pl) O[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa[pl], vb, O[nb], 0, 0, 0);
      }
    }
    __builtin_amdgcn_wave_barrier();     // pbuf is rewritten by the next granule
  }
  // ---- merge the waves: lane holds dims 16*nb + arow of rows 4*akq + r
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int nb = 0; nb < 8; ++nb) part[(w * 16 + 4 * akq + r) * HD + 16 * nb + arow] = O[nb][r];
    if (arow == 0) { pm_s[w * 16 + 4 * akq + r] = mrun[r]; pl_s[w * 16 + 4 * akq + r] = lrun[r]; }
  }
  __syncthreads();
  {
    const int r = tid >> 4, d0 = (tid & 15) * 8, m = qt * 16 + r;
    if (m - off >= len) return;                         // padding row inside the last tile of an utterance
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
  dia_launch<k_attn<KVT, G>>(dim3(k.n_kv_heads, grid_y, grid_z), dim3(NT), 0, st, k);
  return dia_check_launch("k_attn");
}

template <int G>
int launch_attn_mfma(const AttnK& k, int grid_y, int grid_z, hipStream_t st) {
  if (k.kv_plane_stride > 0) dia_launch<k_attn_mfma<G, 2>>(dim3(k.n_kv_heads, grid_y, grid_z), dim3(NT), 0, st, k);
  else dia_launch<k_attn_mfma<G, 1>>(dim3(k.n_kv_heads, grid_y, grid_z), dim3(NT), 0, st, k);
  return dia_check_launch("k_attn_mfma");
}

}  // namespace

#ifdef DIA_DBG_STAMPS
extern "C" int dia_dbg_aclear() {
  static long long zeros[8192 * 8];
  return hipMemcpyToSymbol(HIP_SYMBOL(g_astamps), zeros, sizeof(zeros)) == hipSuccess ? 0 : -2;
}
extern "C" int dia_dbg_astamps(long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_astamps), sizeof(long long) * n) == hipSuccess ? 0 : -2;
}
#endif

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
  // every index the kernels derive from the launch shape is checked here against what the caller states about its
  // buffers: ENC reads q row / RoPE row / emits plane row blockIdx.y for blockIdx.y < n_rows, and keys 0..enc_len-1 of
  // a kv_cap-row scratch cache; SELF reads RoPE row cur <= kv_cap
  if (a->mode == DIA_ATTN_ENC && (a->n_rows != a->enc_len || a->enc_len <= 0 || a->enc_len > a->kv_cap))
    return dia_fail(DIA_E_ARG, "dia_attn: ENC needs n_rows == enc_len and 0 < enc_len <= kv_cap");
  if (a->rope_rows > 0 && ((a->mode == DIA_ATTN_ENC && a->enc_len > a->rope_rows) || (a->mode == DIA_ATTN_SELF && a->kv_cap + 1 > a->rope_rows)))
    return dia_fail(DIA_E_ARG, "dia_attn: RoPE tables shorter than the positions this launch can reach");
  if (a->kv_cap <= 0) return dia_fail(DIA_E_ARG, "dia_attn: kv_cap must be positive");
  const int cap_keys = a->mode == DIA_ATTN_ENC ? a->enc_len : a->kv_cap;
  const int cap_chunks = (cap_keys + CHUNK - 1) / CHUNK;
  // key-split factor: ~2 workgroups of 4 waves per CU, never more than one split per 128 keys of
  // capacity (the scratch sizing granule), and each split at least one 64-key unit
  int max_chunks = (512 + a->n_rows * a->n_kv_heads - 1) / (a->n_rows * a->n_kv_heads);
  if (dia_tune(DIA_TUNE_ATTN_NZ) > 0) max_chunks = dia_tune(DIA_TUNE_ATTN_NZ);
  if (max_chunks > cap_chunks) max_chunks = cap_chunks;
  if (max_chunks < 1) max_chunks = 1;
  if (max_chunks > 1 && (!a->scratch || !a->tickets)) return dia_fail(DIA_E_ARG, "dia_attn: more than 128 keys possible: scratch and tickets are required");
  AttnK k;
  k.mode = a->mode; k.n_kv_heads = a->n_kv_heads; k.n_rows = a->n_rows; k.kv_cap = a->kv_cap;
  k.q = a->q; k.ldq = a->ldq; k.q_off = a->q_off; k.k_off = a->k_off; k.v_off = a->v_off;
  k.kc = a->kc; k.vc = a->vc; k.cur = a->cur; k.len = a->len; k.enc_len = a->enc_len;
  k.cos_t = a->cos_t; k.sin_t = a->sin_t;
  k.P = (bf16_raw*)a->P; k.p_plane_stride = a->p_plane_stride; k.p_ktiles = a->p_ktiles; k.act_f32 = a->act_f32;
  k.head_map = a->head_map; k.v_blocked = a->v_blocked;
  k.gpw = 1;
  { const int g = dia_tune(a->mode == DIA_ATTN_CROSS ? DIA_TUNE_ATTN_GPW_CROSS : DIA_TUNE_ATTN_GPW); if (g > 0) k.gpw = g; }
  // the kernels decode the mode branch-free and load cur[]/len[] unconditionally (index 0 when the mode
  // does not use them): never hand them a null pointer
  if (!k.cur) k.cur = reinterpret_cast<const int*>(a->cos_t);
  if (!k.len) k.len = reinterpret_cast<const int*>(a->cos_t);
  k.scratch = a->scratch; k.tickets = a->tickets; k.max_chunks = (a->kv_cap + CHUNK - 1) / CHUNK;
  if ((a->n_kv_heads * a->group * 128 + 31) / 32 > a->p_ktiles) return dia_fail(DIA_E_ARG, "dia_attn: output planes too narrow");
  hipStream_t st = (hipStream_t)stream;
  const bool f32 = a->kv_dtype == DIA_KV_F32;
  k.kv_plane_stride = 0;
  if (a->kv_dtype == DIA_KV_BF16X2) {
    if (!a->v_blocked || a->kv_plane_stride <= 0 || a->kv_plane_stride % 8 != 0)
      return dia_fail(DIA_E_ARG, "dia_attn: two-plane bf16 K/V needs the blocked V layout and the plane stride of the caches");
    k.kv_plane_stride = a->kv_plane_stride;
  }
  if (a->v_blocked) {      // bf16 caches with the blocked V layout: MFMA kernel
    if (f32 || a->mode == DIA_ATTN_ENC || a->kv_cap % 32 != 0) return dia_fail(DIA_E_ARG, "dia_attn: v_blocked needs bf16 K/V, SELF/CROSS mode and kv_cap % 32 == 0");
    if (a->mode == DIA_ATTN_SELF && !a->cur) return dia_fail(DIA_E_ARG, "dia_attn: SELF needs cur");
    if (a->mode == DIA_ATTN_CROSS && (!a->cur || !a->len || a->group != 1)) return dia_fail(DIA_E_ARG, "dia_attn: CROSS needs cur, len and group 1");
    if (a->group == 4) return launch_attn_mfma<4>(k, a->n_rows, max_chunks, st);
    if (a->group == 2) return launch_attn_mfma<2>(k, a->n_rows, max_chunks, st);
    if (a->group == 1) return launch_attn_mfma<1>(k, a->n_rows, max_chunks, st);
    return dia_fail(DIA_E_ARG, "dia_attn: GQA group must be 1, 2 or 4");
  }
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
  dia_launch<k_enc_kv_prep>(dim3(heads, L), dim3(64), 0, (hipStream_t)stream, qkv, ldq, k_off, v_off, heads, L, cap, cos_t, sin_t, kc, vc);
  return dia_check_launch("k_enc_kv_prep");
}

extern "C" int dia_enc_attn(const dia_enc_attn_args* a, void* stream) {
  if (!a || !a->qkv || !a->row_b || !a->seg_off || !a->seg_len || !a->cos_t || !a->sin_t || !a->kp || !a->vp || !a->P)
    return dia_fail(DIA_E_ARG, "dia_enc_attn: null argument");
  if (a->heads <= 0 || a->rows <= 0 || a->rows % 32 != 0) return dia_fail(DIA_E_ARG, "dia_enc_attn: rows must be a positive multiple of 32");
  if (a->p_plane_stride % 8 != 0 || (a->heads * 128 + 31) / 32 > a->p_ktiles) return dia_fail(DIA_E_ARG, "dia_enc_attn: output planes too narrow");
  // k_enc_kv_planes reads q/k/v rows as float4, row_b as int4 and stores 16-byte plane chunks
  if (a->ldq % 4 != 0 || a->k_off % 4 != 0 || a->v_off % 4 != 0 || ((uintptr_t)a->qkv & 15) || ((uintptr_t)a->row_b & 15) || ((uintptr_t)a->kp & 15) ||
      ((uintptr_t)a->vp & 15) || ((uintptr_t)a->cos_t & 15) || ((uintptr_t)a->sin_t & 15))
    return dia_fail(DIA_E_ARG, "dia_enc_attn: qkv / row_b / kp / vp / RoPE tables must be 16-byte aligned, ldq and the k / v offsets multiples of 4");
  EncAttnK k;
  k.qkv = a->qkv; k.ldq = a->ldq; k.q_off = a->q_off; k.k_off = a->k_off; k.v_off = a->v_off; k.heads = a->heads; k.rows = a->rows;
  k.row_b = a->row_b; k.seg_off = a->seg_off; k.seg_len = a->seg_len; k.cos_t = a->cos_t; k.sin_t = a->sin_t;
  k.kp = (bf16_raw*)a->kp; k.vp = (bf16_raw*)a->vp; k.P = (bf16_raw*)a->P; k.p_plane_stride = a->p_plane_stride; k.p_ktiles = a->p_ktiles;
  hipStream_t st = (hipStream_t)stream;
  dia_launch<k_enc_kv_planes>(dim3(a->heads, a->rows / 32), dim3(256), 0, st, k);
  int rc = dia_check_launch("k_enc_kv_planes");
  if (rc) return rc;
  dia_launch<k_attn_enc_mfma>(dim3(a->heads, a->rows / 16), dim3(NT), 0, st, k);
  return dia_check_launch("k_attn_enc_mfma");
}
