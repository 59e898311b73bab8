// Shared by the GEMM translation units (gemm.hip: the kernels the decode step and the prefill run;
// gemm_experiments.hip: measured-and-rejected alternatives, built only with EXPERIMENTS=1): kernel argument block,
// launch helpers, the four epilogues, the in-workgroup and cross-workgroup split-K reductions.
#pragma once
#include "common.hpp"
#include <type_traits>
#include "../../include/dia_hip.h"
#include "errors.hpp"
#include <hip/hip_ext.h>
#include <cstdlib>
#include <cstring>
#include "tuning.hpp"
#include "launch.hpp"

namespace {


// weights are read once per launch: non-temporal loads (measured 21.6 vs 26.2 us on wi_fused)
#ifdef DIA_DBG_PLAIN_LOAD
#define DIA_WLOAD(ptr) (*(ptr))
#else
#define DIA_WLOAD(ptr) __builtin_nontemporal_load(ptr)
#endif

// strip index of an unconditional prefetch (see k_gemv_small): past the end, the last strip once more
#ifdef DIA_DBG_COND_PREFETCH
#define DIA_PREFETCH_CLAMP(next, n) (next)
#else
#define DIA_PREFETCH_CLAMP(next, n) min((next), (n) - 1)
#endif

template <auto Kern, typename Arg>
void launch_kernel(dim3 grid, dim3 block, size_t smem, hipStream_t st, const Arg& arg) {
  dia_launch<Kern>(grid, block, smem, st, arg);
}

struct GemmK;
template <auto Kern>
void launch_small_kernel(dim3 grid, dim3 block, size_t smem, hipStream_t st, const GemmK& k);

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));   // plain vector: HIP's uint4 struct defeats SROA in register arrays

struct GemmK {
  const bf16_raw* A; long a_plane_stride; int a_ktiles; int M;
  const bf16_raw* W; int KT; int nstrips; int epi;
  const float* ssq_in; int ssq_in_n; int ssq_ld; float inv_d; float eps;
  float* out; int ldo;
  const float* gnext;
  bf16_raw* P; long p_plane_stride; int p_ktiles;
  float* ssq_out;
  void* kc; void* vc; int kv_dtype; int kv_heads; int kv_cap; int kv_batch_index;
  const float* cos_t; const float* sin_t;
  int spw;
  const int* cmap; const int* strip_map;
  float* sk_scratch; int* sk_tickets;     // cross-workgroup split-K (gridDim.y > 1)
  int kv_vblocked;
  const int* row_b; const int* seg_off;   // CROSSKV over a packed batch
  const unsigned char* sp_blocks; const unsigned int* sp_toff;   // zero-skipping weight stream (k_gemv_sparse)
  int mz;                                  // host side only: m-tiles a k_gemm16 launch covers through gridDim.z (0/1 = one)
  int a_f32, p_f32;                        // A / P are fp32 activation tiles (common.hpp) instead of three bf16 planes
  long kv_plane_stride;                    // CROSSKV, DIA_KV_BF16X2
  long kv_layer_stride; int kv_layer_strips;   // CROSSKV over several layers (dia_gemm_args)
  int sk2_quads;                                // k_gemm2t SK2: groups of four strips per hand-off (knob gemm_2t=7: pairs only)
  int w_planes; long w_plane_stride;       // k_gemm only: hi / mid / lo planes of fp32 weights, one tile set each
};

// the three plane fragments of the A operand at element offset `off` inside a plane / an fp32 tile set
__device__ __forceinline__ void load_afrag3(const GemmK& p, long off, bf16x8& a0, bf16x8& a1, bf16x8& a2) {
  if (p.a_f32) {
    const float4* s = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.A) + off);
    split3x8(s[0], s[1], a0, a1, a2);
  } else {
    a0 = *reinterpret_cast<const bf16x8*>(p.A + off);
    a1 = *reinterpret_cast<const bf16x8*>(p.A + p.a_plane_stride + off);
    a2 = *reinterpret_cast<const bf16x8*>(p.A + 2 * p.a_plane_stride + off);
  }
}
// 8 consecutive-column activations of row m leave in the consumer's format
__device__ __forceinline__ void emit_act8(const GemmK& p, int m, int n0, const float* v, const int* cmap) {
  if (p.p_f32) {
    float* Pf = reinterpret_cast<float*>(p.P);
    if (cmap) emit_f32x8_mapped(Pf, p.p_ktiles, m, n0, v, cmap);
    else emit_f32x8(Pf, p.p_ktiles, m, n0, v);
  } else {
    if (cmap) emit_planes8_mapped(p.P, p.p_plane_stride, p.p_ktiles, m, n0, v, cmap);
    else emit_planes8(p.P, p.p_plane_stride, p.p_ktiles, m, n0, v);
  }
}

template <auto Kern>
void launch_small_kernel(dim3 grid, dim3 block, size_t smem, hipStream_t st, const GemmK& k) {
  dia_launch<Kern>(grid, block, smem, st, k.A, k.a_plane_stride, k.W, k.KT, k.M, k.epi, k.nstrips, k.out, k.ldo, k.gnext, k);
}

__device__ __forceinline__ void kv_store(void* base, int dtype, long idx, float v, long plane_stride = 0) {
  if (dtype == DIA_KV_F32) reinterpret_cast<float*>(base)[idx] = v;
  else if (dtype == DIA_KV_BF16X2) {      // hi + lo bf16 planes (16 significand bits)
    const __bf16 hi = (__bf16)v, lo = (__bf16)(v - (float)hi);
    reinterpret_cast<bf16_raw*>(base)[idx] = *reinterpret_cast<const bf16_raw*>(&hi);
    reinterpret_cast<bf16_raw*>(base)[idx + plane_stride] = *reinterpret_cast<const bf16_raw*>(&lo);
  } else KVElem<bf16_raw>::store(reinterpret_cast<bf16_raw*>(base) + idx, v);
}

// Everything the epilogue needs from memory is requested early, behind the weight loads, so that its
// latency overlaps theirs instead of adding dependent round trips at the end of the kernel.
template <int MT, int NT>
__device__ __forceinline__ void prefetch_epilogue(const GemmK& p, int tid, int mt0, int m, int n0, bool live,
                                                  float* xpre, float* gpre, float* inv_s) {
  if (p.epi == DIA_EPI_RESID_EMIT && live) {
    const float* o = p.out + (long)m * p.ldo + n0;
    const float4 xa = *reinterpret_cast<const float4*>(o), xb = *reinterpret_cast<const float4*>(o + 4);
    xpre[0] = xa.x; xpre[1] = xa.y; xpre[2] = xa.z; xpre[3] = xa.w;
    xpre[4] = xb.x; xpre[5] = xb.y; xpre[6] = xb.z; xpre[7] = xb.w;
#pragma unroll
    for (int j = 0; j < 8; ++j) gpre[j] = p.gnext ? p.gnext[n0 + j] : 1.0f;
  }
  for (int t = tid; t < MT * 128; t += NT) {       // 8 threads per row sum the strip partials
    const int r = t >> 3, part = t & 7;
    const int row = mt0 * 16 + r;
    float sA = 0.f, sB = 0.f;
    if (p.ssq_in != nullptr && row < p.M) {
      int i = part;
      for (; i + 8 < p.ssq_in_n; i += 16) {
        sA += p.ssq_in[(long)i * p.ssq_ld + row];
        sB += p.ssq_in[(long)(i + 8) * p.ssq_ld + row];
      }
      if (i < p.ssq_in_n) sA += p.ssq_in[(long)i * p.ssq_ld + row];
    }
    float sq = sA + sB;
    sq += __shfl_xor(sq, 1, 64);
    sq += __shfl_xor(sq, 2, 64);
    sq += __shfl_xor(sq, 4, 64);
    if (part == 0) inv_s[r] = (p.ssq_in != nullptr) ? rsqrtf(sq * p.inv_d + p.eps) : 1.0f;
  }
}

// One thread = one row x 8 consecutive columns of the finished 16x16 tile.
// pre_cs (CROSSKV, K strips): xpre[0..3] / gpre[0..3] already hold cos / sin of (position m, pairs i0..i0+3) — callers that finish many strips
// for the same row look the row up once and request the table entries a strip ahead (k_gemm2t)
__device__ __forceinline__ void run_epilogue(const GemmK& p, const float* trow, float inv, int m, int n0, int half,
                                             int strip, bool live, const float* xpre, const float* gpre, bool pre_cs = false) {
  // (n0 and strip are by-value copies: the compaction maps below redirect them)
  if (p.epi == DIA_EPI_SCALE_STORE) {
    if (!live) return;
    if (p.strip_map) n0 = p.strip_map[strip] * 16 + half * 8;     // compacted output: whole heads dropped
    float4 a = {trow[half * 8 + 0] * inv, trow[half * 8 + 1] * inv, trow[half * 8 + 2] * inv, trow[half * 8 + 3] * inv};
    float4 b = {trow[half * 8 + 4] * inv, trow[half * 8 + 5] * inv, trow[half * 8 + 6] * inv, trow[half * 8 + 7] * inv};
    float* o = p.out + (long)m * p.ldo + n0;
    *reinterpret_cast<float4*>(o) = a;
    *reinterpret_cast<float4*>(o + 4) = b;
  } else if (p.epi == DIA_EPI_RESID_EMIT) {
    float v[8];
    float ss = 0.f;
    if (live) {
      float* o = p.out + (long)m * p.ldo + n0;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = xpre[j] + trow[half * 8 + j];
      *reinterpret_cast<float4*>(o) = float4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<float4*>(o + 4) = float4{v[4], v[5], v[6], v[7]};
#pragma unroll
      for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = mul_rn(v[j], gpre[j]);
      emit_act8(p, m, n0, v, p.cmap);
    }
    float other = __shfl_xor(ss, 1, 64);
    if (half == 0 && live) p.ssq_out[(long)strip * p.ssq_ld + m] = ss + other;
  } else if (p.epi == DIA_EPI_SWIGLU_EMIT) {
    if (!live || half != 0) return;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float g = trow[j] * inv, u = trow[8 + j] * inv;
      v[j] = (g / (1.0f + expf(-g))) * u;
    }
    emit_act8(p, m, strip * 8, v, nullptr);
  } else {  // DIA_EPI_CROSSKV: strips [0, heads*8) hold K as RoPE pairs (d, d+64), the rest hold V
    if (!live) return;
    if (p.strip_map) strip = p.strip_map[strip];                  // compacted cross K/V: original strip index
    long lofs = 0;                                                // several layers in one launch: this strip's layer
    if (p.kv_layer_strips > 0) {
      const int layer = strip / p.kv_layer_strips;
      strip -= layer * p.kv_layer_strips;
      lofs = (long)layer * p.kv_layer_stride;
    }
    int kvb = p.kv_batch_index;
    if (p.row_b) {                                                // packed batch: row -> (utterance, position)
      kvb = p.row_b[m];
      if (kvb < 0) return;
      m -= p.seg_off[kvb];
    }
    const int nk = p.kv_heads * 8;
    if (strip < nk) {
      const int head = strip >> 3, i0 = (strip & 7) * 8 + half * 4;
      const long base = lofs + (((long)kvb * p.kv_heads + head) * p.kv_cap + m) * 128;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = i0 + t;
        const float x1 = trow[half * 8 + 2 * t] * inv, x2 = trow[half * 8 + 2 * t + 1] * inv;
        const float c = pre_cs ? xpre[t] : p.cos_t[(long)m * 64 + i], s = pre_cs ? gpre[t] : p.sin_t[(long)m * 64 + i];
        kv_store(p.kc, p.kv_dtype, base + i, x1 * c - x2 * s, p.kv_plane_stride);
        kv_store(p.kc, p.kv_dtype, base + i + 64, x1 * s + x2 * c, p.kv_plane_stride);
      }
    } else {
      const int sv = strip - nk, head = sv >> 3, d0 = (sv & 7) * 16 + half * 8;
      if (p.kv_vblocked) {      // [key/32][128 dims][32 keys] (MFMA attention reads 8 consecutive keys per lane)
        const long hb = lofs + ((long)kvb * p.kv_heads + head) * p.kv_cap * 128;
        const long blk = hb + (long)(m >> 5) * 128 * 32 + (m & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) kv_store(p.vc, p.kv_dtype, blk + (long)(d0 + j) * 32, trow[half * 8 + j] * inv, p.kv_plane_stride);
      } else {
        const long base = lofs + (((long)kvb * p.kv_heads + head) * p.kv_cap + m) * 128 + d0;
#pragma unroll
        for (int j = 0; j < 8; ++j) kv_store(p.vc, p.kv_dtype, base + j, trow[half * 8 + j] * inv, p.kv_plane_stride);
      }
    }
  }
}

// DPP moves inside rows of 16 lanes (no LDS crossbar, a few cycles): lane i takes lane i - n / i + n of its row
#define DIA_ROW_SHR(v, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), 0x110 + (n), 0xF, 0xF, true))
#define DIA_ROW_SHL(v, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), 0x100 + (n), 0xF, 0xF, true))

// M <= RS rows (k_gemv_small): every decode epilogue with ONE element per thread (16*RS threads: row = tid / 16, column =
// tid % 16; `v` = that element of the finished 16x16 tile) instead of eight per thread on half a wave — the 32-thread form
// spends 0.5 us of single-wave issue time (8 three-way bf16 splits per thread) at the very end of every launch.  Results are
// bit-identical to run_epilogue: the strip's sum of squares is accumulated in the same order (columns 0..7 of each half in
// sequence, rounded squares and plain adds, then half 0 + half 1), carried from lane to lane by DPP row shifts.
template <int RS, bool PF32 = false>
__device__ __forceinline__ void run_epilogue_rows(const GemmK& p, float v_tile, const float* inv_s, int tid, int strip,
                                                  float xpre1, float gpre1) {
  const int m = tid >> 4, c = tid & 15;
  const bool live = m < p.M;
  if (p.epi == DIA_EPI_RESID_EMIT) {
    const int n = strip * 16 + c;
    const float v = xpre1 + v_tile;
    if (live) p.out[(long)m * p.ldo + n] = v;
    const float sq = mul_rn(v, v);      // (the 32-thread form squares with packed multiplies and adds in sequence: no FMA)
    float acc = sq;
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      const float t = DIA_ROW_SHR(acc, 1);
      if ((c & 7) == j) acc = add_rn(t, sq);
    }
    const float h0 = DIA_ROW_SHR(acc, 8);                               // lane 15 of the row: the sum of columns 0..7
    if (live && c == 15) p.ssq_out[(long)strip * p.ssq_ld + m] = h0 + acc;
    const float vg = mul_rn(v, gpre1);
    int cc = n;
    if (p.cmap) cc = p.cmap[n];
    if (live && cc >= 0) {
      const long off = plane_frag_off(m, cc & ~7, p.p_ktiles) + (cc & 7);
      if constexpr (PF32) {
        reinterpret_cast<float*>(p.P)[off] = vg;        // fp32 tile: one 4-byte store instead of three 2-byte ones
      } else {
        __bf16 a, b, d;
        split3(vg, a, b, d);
        p.P[off] = *reinterpret_cast<bf16_raw*>(&a);
        p.P[p.p_plane_stride + off] = *reinterpret_cast<bf16_raw*>(&b);
        p.P[2 * p.p_plane_stride + off] = *reinterpret_cast<bf16_raw*>(&d);
      }
    }
  } else if (p.epi == DIA_EPI_SWIGLU_EMIT) {    // columns 0..7 gate, 8..15 up
    const float up_raw = DIA_ROW_SHL(v_tile, 8);                        // lane c < 8 takes column c + 8
    if (!live || c >= 8) return;
    const float inv = inv_s[m];
    const float g = v_tile * inv, u = up_raw * inv;
    const float v = (g / (1.0f + expf(-g))) * u;
    const long off = plane_frag_off(m, strip * 8, p.p_ktiles) + c;
    if constexpr (PF32) {
      reinterpret_cast<float*>(p.P)[off] = v;
    } else {
      __bf16 a, b, d;
      split3(v, a, b, d);
      p.P[off] = *reinterpret_cast<bf16_raw*>(&a);
      p.P[p.p_plane_stride + off] = *reinterpret_cast<bf16_raw*>(&b);
      p.P[2 * p.p_plane_stride + off] = *reinterpret_cast<bf16_raw*>(&d);
    }
  } else {                                      // DIA_EPI_SCALE_STORE
    if (!live) return;
    const int s_out = p.strip_map ? p.strip_map[strip] : strip;         // compacted output: whole heads dropped
    p.out[(long)m * p.ldo + s_out * 16 + c] = v_tile * inv_s[m];
  }
}

// split-K partials -> LDS -> fixed-order sum -> 16x16 tile(s) in LDS
template <int MT, int NW, bool RAW = false>
__device__ __forceinline__ void reduce_to_tile(const f32x4* acc, f32x4* red, float* tile, int tid, int lane, int w) {
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) red[(w * MT + mt) * 64 + lane] = acc[mt];
  if constexpr (RAW) lds_barrier(); else __syncthreads();
  if (tid < MT * 64) {
    const int mt = tid >> 6;
    f32x4 s = red[(0 * MT + mt) * 64 + lane];
#pragma unroll
    for (int ww = 1; ww < NW; ++ww) {
      f32x4 t = red[(ww * MT + mt) * 64 + lane];
      s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
    }
    const int col = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[(mt * 16 + r0 + r) * 17 + col] = s[r];
  }
  if constexpr (RAW) lds_barrier(); else __syncthreads();
}

// Cross-workgroup split-K: gridDim.y workgroups hold partial 16x16 tiles of one strip.  Each publishes
// its tile to a slab; the LAST arriver (agent-scope release / ticket / acquire, guide §6 G16) sums the
// slabs in split order — bit-reproducible regardless of arrival order — and alone runs the epilogue.
// Returns true for the workgroup that must run the epilogue (always true when gridDim.y == 1).
__device__ __forceinline__ bool splitk_combine(const GemmK& p, float* tile, int strip, int tid, int* flag_s) {
  const int SK = gridDim.y;
  if (SK == 1) return true;
  const int ks = blockIdx.y;
#ifdef DIA_X_NOHANDOFF
  return ks == SK - 1;                                            // TIMING ONLY: no slab, no ticket, no merge (wrong results)
#endif
  // hand-off through device-coherent (sc1) accesses with explicit ordering, no cache-wide fences — see
  // attn_finish in attn.hip
  // 16-byte coherent accesses: 64 threads carry the 16 x 16 tile (one row quarter each)
  const __amdgpu_buffer_rsrc_t sr = agent_rsrc(p.sk_scratch);
  const int row = tid >> 2, c4 = (tid & 3) * 4;                   // tid < 64
  if (tid < 64) {
    const float* t = tile + row * 17 + c4;
    st4_agent(sr, (int)((((long)strip * SK + ks) * 256 + row * 16 + c4) * 4), f32x4{t[0], t[1], t[2], t[3]});
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const int ticket = __hip_atomic_fetch_add(p.sk_tickets + strip, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int last = ticket == SK - 1;
    if (last) __hip_atomic_store(p.sk_tickets + strip, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    *flag_s = last;
  }
  __syncthreads();
  if (!*flag_s) return false;
  if (tid < 64) {
    f32x4 v[4];                                                   // SK <= 4: every slab requested before the first is used
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < SK) v[k] = ld4_agent(sr, (int)((((long)strip * SK + k) * 256 + row * 16 + c4) * 4));
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (k < SK) { a[0] += v[k][0]; a[1] += v[k][1]; a[2] += v[k][2]; a[3] += v[k][3]; }
    for (int k = 4; k < SK; ++k) {
      const f32x4 t = ld4_agent(sr, (int)((((long)strip * SK + k) * 256 + row * 16 + c4) * 4));
      a[0] += t[0]; a[1] += t[1]; a[2] += t[2]; a[3] += t[3];
    }
    float* t = tile + row * 17 + c4;
    t[0] = a[0]; t[1] = a[1]; t[2] = a[2]; t[3] = a[3];
  }
  __syncthreads();
  return true;
}

#ifdef DIA_DBG_STAMPS
__device__ long long g_stamps[4096 * 8];
#define STAMP(i) do { if (tid == 0) g_stamps[((blockIdx.y * gridDim.x + blockIdx.x) & 4095) * 8 + (i)] = wall_clock64(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

constexpr int GT_MT = 4, GT_WM = 2, GT_WS = 4;   // workgroup: 4 m-tiles; per wave: 2 m-tiles x 4 strips
constexpr size_t gt_abuf(int kc) { return (size_t)kc * DIA_NPLANES * GT_MT * 64 * 16; }    // bytes of one staged chunk (24 KiB at 2 k-tiles)
constexpr size_t gt_smem(int kc, int nw) { return 2 * gt_abuf(kc) + sizeof(float) * (nw * 2 * 16 * 17 + 64); }

inline int fill_gemmk(const dia_gemm_args* a, GemmK& k) {
  k.A = (const bf16_raw*)a->A; k.a_plane_stride = a->a_plane_stride; k.a_ktiles = a->a_ktiles; k.M = a->M;
  k.W = (const bf16_raw*)a->W; k.KT = a->KT; k.nstrips = a->nstrips; k.epi = a->epi;
  k.ssq_in = a->ssq_in; k.ssq_in_n = a->ssq_in_n; k.ssq_ld = a->ssq_ld; k.inv_d = a->inv_d; k.eps = a->eps;
  k.out = a->out; k.ldo = a->ldo; k.gnext = a->gnext;
  k.P = (bf16_raw*)a->P; k.p_plane_stride = a->p_plane_stride; k.p_ktiles = a->p_ktiles; k.ssq_out = a->ssq_out;
  k.kc = a->kc; k.vc = a->vc; k.kv_dtype = a->kv_dtype; k.kv_heads = a->kv_heads; k.kv_cap = a->kv_cap;
  k.kv_batch_index = a->kv_batch_index; k.cos_t = a->cos_t; k.sin_t = a->sin_t; k.spw = a->spw;
  k.cmap = a->cmap; k.strip_map = a->strip_map;
  k.sk_scratch = a->sk_scratch; k.sk_tickets = a->sk_tickets; k.kv_vblocked = a->kv_vblocked;
  k.row_b = a->row_b; k.seg_off = a->seg_off;
  k.sp_blocks = (const unsigned char*)a->sp_blocks; k.sp_toff = (const unsigned int*)a->sp_toff;
  k.mz = 0;
  k.a_f32 = a->act_f32 & 1; k.p_f32 = (a->act_f32 >> 1) & 1;
  k.kv_plane_stride = a->kv_plane_stride;
  k.kv_layer_strips = a->kv_layer_strips; k.kv_layer_stride = a->kv_layer_stride;
  k.sk2_quads = dia_tune(DIA_TUNE_GEMM_2T) != 7;
  k.w_planes = a->w_planes > 1 ? a->w_planes : 1; k.w_plane_stride = (long)a->KT * a->nstrips * 512;
  return DIA_OK;
}

}  // namespace
