// Skinny GEMM for the decode path: out[M][N] = X[M][K] . W[K][N], M small (2..64 rows per launch
// group), W streamed from HBM exactly once per launch.
//
// Replaces DenseGeneral.forward (reference dia/layers.py:55-66, torch.tensordot) plus, via the
// epilogues, RMSNorm scaling, residual add, SwiGLU and the cross-K/V RoPE + cache store.
//
// Mapping (CDNA4): one workgroup owns one 16-column strip of W and all of K; its NW waves split K
// into contiguous ranges of KPW k-tiles.  A k-tile of the strip is one contiguous 1 KiB block
// (64 lanes x 16 B) that is loaded (non-temporal: read once) straight into the B operand of
// v_mfma_f32_16x16x32_bf16 — no LDS staging, no conversion (guide §5 'GEMV / M <= 16 decode
// weights').  All B loads of a wave are issued before anything else.  X arrives as three bf16
// planes (hi+mid+lo == fp32 value); three MFMAs per k-tile accumulate them into one fp32
// accumulator, which makes the product exact w.r.t. the fp32 activations of the reference.
// Cross-wave (split-K) partials are summed through LDS in a fixed order: results are
// bit-reproducible run to run.
//
// Two kernels share the epilogue:
//   k_gemm<MT,NW,KPW>       A fragments straight from L2 (any M; rows >= M alias the last valid row)
//   k_gemv_small<NW,KPW,RS> M <= RS in {2,4} (batch 1-2): the few valid rows of X are staged ONCE per
//                           workgroup into LDS in compact fragment order, so a wave issues 3*KT*4*RS/256
//                           staging loads instead of 3*KPW full-wave fragment loads per wave — the
//                           texture-address path, not HBM, bounded the direct form (measured 21.6 -> 14.5 us
//                           on the 64 MiB wi_fused matrix with the A loads removed).
#include "common.hpp"
#include <type_traits>
#include "../../include/dia_hip.h"
#include "errors.hpp"
#include <hip/hip_ext.h>
#include <cstdlib>
#include <cstring>

namespace {

// weights are read once per launch: non-temporal loads (measured 21.6 vs 26.2 us on wi_fused)
#ifdef DIA_DBG_PLAIN_LOAD
#define DIA_WLOAD(ptr) (*(ptr))
#else
#define DIA_WLOAD(ptr) __builtin_nontemporal_load(ptr)
#endif

// when set (dia_gemm_timed), the next launch is bracketed by these events via hipExtLaunchKernelGGL:
// the timestamps come from the dispatch packet itself (kernel begin/end), like rocprofv3's durations
thread_local hipEvent_t g_ev_start = nullptr, g_ev_stop = nullptr;

template <typename Kern, typename Arg>
void launch_kernel(Kern kern, dim3 grid, dim3 block, size_t smem, hipStream_t st, const Arg& arg) {
  if (g_ev_start) hipExtLaunchKernelGGL(kern, grid, block, smem, st, g_ev_start, g_ev_stop, 0, arg);
  else hipLaunchKernelGGL(kern, grid, block, smem, st, arg);
}

struct GemmK;
template <typename Kern>
void launch_small_kernel(Kern kern, dim3 grid, dim3 block, size_t smem, hipStream_t st, const GemmK& k);

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
};

template <typename Kern>
void launch_small_kernel(Kern kern, dim3 grid, dim3 block, size_t smem, hipStream_t st, const GemmK& k) {
  if (g_ev_start) hipExtLaunchKernelGGL(kern, grid, block, smem, st, g_ev_start, g_ev_stop, 0, k.A, k.a_plane_stride, k.W, k.KT, k.M, k.epi,
                                        k.nstrips, k.out, k.ldo, k.gnext, k);
  else hipLaunchKernelGGL(kern, grid, block, smem, st, k.A, k.a_plane_stride, k.W, k.KT, k.M, k.epi, k.nstrips, k.out, k.ldo, k.gnext, k);
}

__device__ __forceinline__ void kv_store(void* base, int dtype, long idx, float v) {
  if (dtype == DIA_KV_F32) reinterpret_cast<float*>(base)[idx] = v;
  else KVElem<bf16_raw>::store(reinterpret_cast<bf16_raw*>(base) + idx, v);
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
__device__ __forceinline__ void run_epilogue(const GemmK& p, const float* trow, float inv, int m, int n0, int half,
                                             int strip, bool live, const float* xpre, const float* gpre) {
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
      if (p.cmap) emit_planes8_mapped(p.P, p.p_plane_stride, p.p_ktiles, m, n0, v, p.cmap);
      else emit_planes8(p.P, p.p_plane_stride, p.p_ktiles, m, n0, v);
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
    emit_planes8(p.P, p.p_plane_stride, p.p_ktiles, m, strip * 8, v);
  } else {  // DIA_EPI_CROSSKV: strips [0, heads*8) hold K as RoPE pairs (d, d+64), the rest hold V
    if (!live) return;
    if (p.strip_map) strip = p.strip_map[strip];                  // compacted cross K/V: original strip index
    int kvb = p.kv_batch_index;
    if (p.row_b) {                                                // packed batch: row -> (utterance, position)
      kvb = p.row_b[m];
      if (kvb < 0) return;
      m -= p.seg_off[kvb];
    }
    const int nk = p.kv_heads * 8;
    if (strip < nk) {
      const int head = strip >> 3, i0 = (strip & 7) * 8 + half * 4;
      const long base = (((long)kvb * p.kv_heads + head) * p.kv_cap + m) * 128;
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int i = i0 + t;
        const float x1 = trow[half * 8 + 2 * t] * inv, x2 = trow[half * 8 + 2 * t + 1] * inv;
        const float c = p.cos_t[(long)m * 64 + i], s = p.sin_t[(long)m * 64 + i];
        kv_store(p.kc, p.kv_dtype, base + i, x1 * c - x2 * s);
        kv_store(p.kc, p.kv_dtype, base + i + 64, x1 * s + x2 * c);
      }
    } else {
      const int sv = strip - nk, head = sv >> 3, d0 = (sv & 7) * 16 + half * 8;
      if (p.kv_vblocked) {      // [key/32][128 dims][32 keys] (MFMA attention reads 8 consecutive keys per lane)
        const long hb = ((long)kvb * p.kv_heads + head) * p.kv_cap * 128;
        const long blk = hb + (long)(m >> 5) * 128 * 32 + (m & 31);
#pragma unroll
        for (int j = 0; j < 8; ++j) kv_store(p.vc, p.kv_dtype, blk + (long)(d0 + j) * 32, trow[half * 8 + j] * inv);
      } else {
        const long base = (((long)kvb * p.kv_heads + head) * p.kv_cap + m) * 128 + d0;
#pragma unroll
        for (int j = 0; j < 8; ++j) kv_store(p.vc, p.kv_dtype, base + j, trow[half * 8 + j] * inv);
      }
    }
  }
}

// M <= RS rows (k_gemv_small): RESID_EMIT and SWIGLU_EMIT with ONE element per thread (16*RS threads: row = tid / 16,
// column = tid % 16) instead of eight per thread on half a wave — the 32-thread form spends 0.5 us of single-wave issue time
// (8 three-way bf16 splits per thread) at the very end of every o / co / wo / wi launch.  Results are bit-identical
// to run_epilogue: the strip's sum of squares is accumulated in the same order (columns 0..7 of each half in
// sequence, rounded squares and plain adds, then half 0 + half 1) through lane shifts.
template <int RS>
__device__ __forceinline__ void run_epilogue_rows(const GemmK& p, const float* tile, const float* inv_s, int tid, int strip,
                                                  float xpre1, float gpre1) {
  const int m = tid >> 4, c = tid & 15;
  const bool live = m < p.M;
  if (p.epi == DIA_EPI_RESID_EMIT) {
    const int n = strip * 16 + c;
    const float v = xpre1 + tile[m * 17 + c];
    if (live) p.out[(long)m * p.ldo + n] = v;
    const float sq = mul_rn(v, v);      // (the 32-thread form squares with packed multiplies and adds in sequence: no FMA)
    float acc = sq;
#pragma unroll
    for (int j = 1; j < 8; ++j) {
      const float t = __shfl_up(acc, 1, 64);
      if ((c & 7) == j) acc = add_rn(t, sq);
    }
    const int lane = tid & 63;
    const float h0 = __shfl(acc, (lane & ~15) | 7, 64), h1 = __shfl(acc, (lane & ~15) | 15, 64);
    if (live && c == 0) p.ssq_out[(long)strip * p.ssq_ld + m] = h0 + h1;
    const float vg = mul_rn(v, gpre1);
    int cc = n;
    if (p.cmap) cc = p.cmap[n];
    if (live && cc >= 0) {
      __bf16 a, b, d;
      split3(vg, a, b, d);
      const long off = plane_frag_off(m, cc & ~7, p.p_ktiles) + (cc & 7);
      p.P[off] = *reinterpret_cast<bf16_raw*>(&a);
      p.P[p.p_plane_stride + off] = *reinterpret_cast<bf16_raw*>(&b);
      p.P[2 * p.p_plane_stride + off] = *reinterpret_cast<bf16_raw*>(&d);
    }
  } else {  // DIA_EPI_SWIGLU_EMIT: columns 0..7 gate, 8..15 up
    if (!live || c >= 8) return;
    const float inv = inv_s[m];
    const float g = tile[m * 17 + c] * inv, u = tile[m * 17 + 8 + c] * inv;
    const float v = (g / (1.0f + expf(-g))) * u;
    __bf16 a, b, d;
    split3(v, a, b, d);
    const long off = plane_frag_off(m, strip * 8, p.p_ktiles) + c;
    p.P[off] = *reinterpret_cast<bf16_raw*>(&a);
    p.P[p.p_plane_stride + off] = *reinterpret_cast<bf16_raw*>(&b);
    p.P[2 * p.p_plane_stride + off] = *reinterpret_cast<bf16_raw*>(&d);
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
  // hand-off through device-coherent (sc1) accesses with explicit ordering, no cache-wide fences — see
  // attn_finish in attn.hip
  float* slab = p.sk_scratch + ((long)strip * SK + ks) * 256;
  if (tid < 128) {
    const int e = tid * 2;
    st2_agent(slab + e, tile[(e >> 4) * 17 + (e & 15)], tile[(e >> 4) * 17 + (e & 15) + 1]);
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
  if (tid < 128) {
    const int e = tid * 2;
    const float* base = p.sk_scratch + (long)strip * SK * 256 + e;
    float a = 0.f, b = 0.f;
    for (int k = 0; k < SK; ++k) { const float2 v = ld2_agent(base + k * 256); a += v.x; b += v.y; }
    tile[(e >> 4) * 17 + (e & 15)] = a; tile[(e >> 4) * 17 + (e & 15) + 1] = b;
  }
  __syncthreads();
  return true;
}

template <int MT, int NW, int KPW>
__global__ __launch_bounds__(NW * 64) void k_gemm(GemmK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                         // [NW][MT][64]
  float* tile = reinterpret_cast<float*>(smem_raw + sizeof(f32x4) * NW * MT * 64);   // [MT][16][17]
  float* inv_s = tile + MT * 16 * 17;                                      // [MT*16]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int strip = blockIdx.x;
  const int mt0 = blockIdx.y * MT;                 // first m-tile of this group
  const int kpw = (KPW > 0) ? KPW : (p.KT + NW - 1) / NW;
  const int kt0 = w * kpw;

  f32x4 acc[MT];
  long aoff[MT];   // per m-tile element offset of this lane's A fragment at k-tile 0
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    // rows >= M: the lane re-reads the last valid row of the (clamped) tile — same 16 B as its
    // neighbour, so it costs no extra L2 traffic; its results are never stored
    const int mtile = min(mt0 + i, (p.M - 1) >> 4);
    const int rlast = min(15, p.M - 1 - mtile * 16);
    const int alane = (lane & 48) | min(lane & 15, rlast);
    aoff[i] = ((long)mtile * p.a_ktiles * 64 + alane) * 8;
  }

  const bf16x8* Wt = reinterpret_cast<const bf16x8*>(p.W) + ((long)strip * p.KT) * 64 + lane;
  const bf16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  // epilogue geometry (threads 0 .. MT*32-1: one row, 8 consecutive columns each)
  const int e_mt = tid >> 5, e_r = (tid >> 1) & 15, half = tid & 1;
  const int m = (mt0 + e_mt) * 16 + e_r;
  const int n0 = strip * 16 + half * 8;
  const bool e_thread = tid < MT * 32;
  const bool live = e_thread && m < p.M;
  float xpre[8], gpre[8];

  if constexpr (KPW > 0) {
    bf16x8 b[KPW];
#pragma unroll
    for (int i = 0; i < KPW; ++i) b[i] = DIA_WLOAD(Wt + (long)(kt0 + i) * 64);
    __builtin_amdgcn_sched_barrier(0);   // every HBM load of this wave is in flight before anything else
    // k-tiles of A fetched up front (registers: 12*MT per k-tile; a 16-wave workgroup has 128 VGPRs)
    constexpr int AP = (KPW >= 8) ? 1 : ((KPW * MT <= 4) ? KPW : ((4 / MT) > 0 ? (4 / MT) : 1));
    bf16x8 a0[AP][MT][DIA_NPLANES];
#pragma unroll
    for (int i = 0; i < AP; ++i)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int pl = 0; pl < DIA_NPLANES; ++pl)
          a0[i][mt][pl] = *reinterpret_cast<const bf16x8*>(p.A + pl * p.a_plane_stride + aoff[mt] + (long)(kt0 + i) * 512);
    prefetch_epilogue<MT, NW * 64>(p, tid, mt0, m, n0, live, xpre, gpre, inv_s);
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int pl = 0; pl < DIA_NPLANES; ++pl) {
          bf16x8 a;
          if (i < AP) a = a0[i < AP ? i : 0][mt][pl];
          else a = *reinterpret_cast<const bf16x8*>(p.A + pl * p.a_plane_stride + aoff[mt] + (long)(kt0 + i) * 512);
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[i], acc[mt], 0, 0, 0);
        }
      }
    }
  } else {
    prefetch_epilogue<MT, NW * 64>(p, tid, mt0, m, n0, live, xpre, gpre, inv_s);
    const int kt1 = min(kt0 + kpw, p.KT);
    for (int kt = kt0; kt < kt1; kt += 4) {
      bf16x8 b[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) b[i] = (kt + i < kt1) ? DIA_WLOAD(Wt + (long)(kt + i) * 64) : zero8;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (kt + i < kt1) {
#pragma unroll
          for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
            for (int pl = 0; pl < DIA_NPLANES; ++pl) {
              bf16x8 a = *reinterpret_cast<const bf16x8*>(p.A + pl * p.a_plane_stride + aoff[mt] + (long)(kt + i) * 512);
              acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[i], acc[mt], 0, 0, 0);
            }
          }
        }
      }
    }
  }

  reduce_to_tile<MT, NW>(acc, red, tile, tid, lane, w);
  if (!e_thread) return;
  run_epilogue(p, tile + (e_mt * 16 + e_r) * 17, inv_s[e_mt * 16 + e_r], m, n0, half, strip, live, xpre, gpre);
}

// M <= RS rows (RS = 2 or 4): the valid rows of all three planes are staged once per workgroup into
// LDS as [plane][ktile][kq 0..3][row 0..RS-1] x 16 B; lane l of a wave then reads its A fragment for
// k-tile kt at ((plane*KT + kt)*4 + (l>>4))*RS + min(l&15, RS-1) (rows >= RS alias row RS-1: broadcast).
#ifdef DIA_DBG_STAMPS
__device__ long long g_stamps[4096 * 8];
#define STAMP(i) do { if (tid == 0) g_stamps[blockIdx.x * 8 + (i)] = wall_clock64(); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

// The first ten arguments repeat fields of p: they fill the first 64 bytes of the argument block, which the command
// processor hands over in SGPRs at wave launch (kernarg preload, -mllvm -amdgpu-kernarg-preload-count=16) — the operand
// and weight loads of the prologue then need no scalar load from the argument block, whose lines every CU of the grid
// otherwise requests at the same moment (in-kernel stamps: 0.8 us from the start of a wave to its first weight load).
template <int NW, int KPW, int RS, bool MULTI, bool MZ = false>
__global__ __launch_bounds__(NW * 64) void k_gemv_small(const bf16_raw* a_A, long a_aps, const bf16_raw* a_W, int a_KT, int a_M, int a_epi,
                                                        int a_nstrips, float* a_out, int a_ldo, const float* a_gnext, GemmK p) {
  p.A = a_A; p.a_plane_stride = a_aps; p.W = a_W; p.KT = a_KT; p.M = a_M; p.epi = a_epi; p.nstrips = a_nstrips;
  p.out = a_out; p.ldo = a_ldo; p.gnext = a_gnext;
  // 5..16 rows (batch 3-8), short K, few strips: gridDim.z row groups of 4, each the 4-row kernel on rows 4z..4z+3 of
  // the one m-tile (a row shift is a pointer shift in every layout involved).  Two workgroups fit a CU, so strips x
  // groups <= 512 are all resident and the groups of a strip share its weights through L2 — against k_gemm16, whose
  // one workgroup per strip pulls the whole 16-row image (192 KB at K = 2048) through one CU for 64 KB of weights.
  if constexpr (MZ) {
    const int r0 = 4 * blockIdx.z;
    p.A += r0 * 8;
    p.M = min(4, p.M - r0);
    if (p.ssq_in) p.ssq_in += r0;
    if (p.out) p.out += (long)r0 * p.ldo;
    if (p.P) p.P += r0 * 8;
    if (p.ssq_out) p.ssq_out += r0;
  }
  // (the compiler loads the fields of the 250-byte argument block where they are first used: four s_load round trips
  // lie between the start of a wave and its first weight load.  Fetching every field up front in one batch —
  // asm volatile("" :: "s"(p.A), "s"(p.W), ...) — was measured: the step got 4 % SLOWER, the first wait then covers four
  // cold lines instead of one)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                         // [NW][64]
  float* tile = reinterpret_cast<float*>(smem_raw + sizeof(f32x4) * NW * 64);   // [16][17]
  float* inv_s = tile + 16 * 17;                                           // [16]
  bf16x8* As = reinterpret_cast<bf16x8*>(smem_raw + sizeof(f32x4) * NW * 64 + sizeof(float) * (16 * 17 + 16));

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kt0 = w * KPW;                 // k-tile inside this workgroup's K range
  const int ktg = blockIdx.y * (NW * KPW); // first global k-tile of that range (split-K over gridDim.y)
  const int G = gridDim.x;                 // the workgroup walks strips blockIdx.x, +G, +2G, ...
  const bf16x8* Wl = reinterpret_cast<const bf16x8*>(p.W) + (long)(ktg + kt0) * 64 + lane;
  auto load_strip = [&](bf16x8* b, int strip) {
    const bf16x8* Wt = Wl + (long)strip * p.KT * 64;
#pragma unroll
    for (int i = 0; i < KPW; ++i) b[i] = DIA_WLOAD(Wt + (long)i * 64);
  };
  __shared__ int sk_flag;

  STAMP(0);
  bf16x8 b0[KPW], b1[MULTI ? KPW : 1];
  constexpr int KT = NW * KPW;             // the dispatcher only picks this kernel when p.KT == NW*KPW
  constexpr int NT = NW * 64;

  const int e_r = (tid >> 1) & 15, half = tid & 1;
  const int m = e_r;
  const bool e_thread = tid < 32;
  const bool live = e_thread && m < p.M;
  float xpre[8], gpre[8];

  // ---- every small, L2-resident operand is requested BEFORE the weight stream, branch-free: vmcnt
  // retires in order, so anything queued behind 16-32 KiB of HBM loads per wave would stall its first
  // use (and with it the barrier below) until the whole strip has arrived.
  // (1) compact A image: chunk c = ((plane*KT + kt)*4 + kq)*RS + row, 16 bytes each
  constexpr int CH = (3 * KPW * RS + 15) / 16;        // chunks per thread = 3*KT*4*RS / NT
  constexpr int nchunks = DIA_NPLANES * KT * 4 * RS;
  bf16x8 v0[CH];
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    const int c = min(tid + u * NT, nchunks - 1);
    const int row = c % RS, kq = (c / RS) & 3, kt = (c / (4 * RS)) % KT, pl = c / (4 * RS * KT);
    v0[u] = *reinterpret_cast<const bf16x8*>(p.A + pl * p.a_plane_stride + ((long)(ktg + kt) * 64 + row + 16 * kq) * 8);
  }
  // (2) strip sums of squares for the row scale: 8 threads per row, up to 16 strips each per round
  const bool has_norm = p.ssq_in != nullptr;
  const int s_row = tid >> 3, s_part = tid & 7;
  const bool s_thread = tid < 128 && has_norm;
  float sq[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) sq[i] = 0.f;
  if (s_thread) {        // one exec-masked region, 16 unconditional loads on clamped addresses
    const float* sp = p.ssq_in + min(s_row, p.M - 1);
#pragma unroll
    for (int i = 0; i < 16; ++i) sq[i] = sp[(long)min(s_part + 8 * i, p.ssq_in_n - 1) * p.ssq_ld];
  }
  // (3) residual row + next norm weight of the first strip (RESID_EMIT only)
  const bool resid = p.epi == DIA_EPI_RESID_EMIT;
  const bool rows_epi = resid || p.epi == DIA_EPI_SWIGLU_EMIT;      // one element per thread (run_epilogue_rows)
  const bool r_thread = tid < 16 * RS;
  float xpre1 = 0.f, gpre1 = 1.f;
  auto load_resid = [&](int strip) {
    const int r_m = tid >> 4, n = strip * 16 + (tid & 15);
    xpre1 = p.out[(long)(r_m < p.M ? r_m : 0) * p.ldo + n];
    gpre1 = p.gnext[n];
  };
  if (resid && r_thread) load_resid(blockIdx.x);
  __builtin_amdgcn_sched_barrier(0);
  load_strip(b0, blockIdx.x);                       // the HBM stream starts here
  __builtin_amdgcn_sched_barrier(0);
  STAMP(1);
#pragma unroll
  for (int u = 0; u < CH; ++u)
    if (tid + u * NT < nchunks) As[tid + u * NT] = v0[u];
  {
    float s0 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s0 += (s_part + 8 * i < p.ssq_in_n && s_row < p.M) ? sq[i] : 0.f;
    if (s_thread && s_row < p.M)
      for (int idx = s_part + 128; idx < p.ssq_in_n; idx += 8) s0 += p.ssq_in[(long)idx * p.ssq_ld + s_row];   // D > 2048 only
    s0 += __shfl_xor(s0, 1, 64);
    s0 += __shfl_xor(s0, 2, 64);
    s0 += __shfl_xor(s0, 4, 64);
    if (tid < 128 && s_part == 0) inv_s[s_row] = has_norm ? rsqrtf(s0 * p.inv_d + p.eps) : 1.0f;
  }
  lds_barrier();      // A image + row scales visible; the weight loads stay in flight
  STAMP(2);

  const int arow = min(lane & 15, RS - 1), akq = lane >> 4;
  auto body = [&](bf16x8* bc, bf16x8* bn, int strip) {
    const int next = strip + G;
    if constexpr (MULTI) { if (next < p.nstrips) load_strip(bn, next); }     // next strip's weights stream while this one computes
    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
#pragma unroll
      for (int pl = 0; pl < DIA_NPLANES; ++pl) {
        const bf16x8 a = As[((pl * KT + kt0 + i) * 4 + akq) * RS + arow];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bc[i], acc[0], 0, 0, 0);
      }
    }
    STAMP(3);
    reduce_to_tile<1, NW, true>(acc, red, tile, tid, lane, w);
    STAMP(4);
    if (!MULTI) { if (!splitk_combine(p, tile, strip, tid, &sk_flag)) return; }
    if (rows_epi) {
      if (r_thread) {
        run_epilogue_rows<RS>(p, tile, inv_s, tid, strip, xpre1, gpre1);
        if (MULTI && next < p.nstrips && resid) load_resid(next);      // residual operands of the next strip
      }
    } else if (e_thread) {
      const int n0 = strip * 16 + half * 8;
      run_epilogue(p, tile + e_r * 17, inv_s[e_r], m, n0, half, strip, live, xpre, gpre);
    }
  };
  if constexpr (MULTI) {
    for (int strip = blockIdx.x; strip < p.nstrips; strip += 2 * G) {
      body(b0, b1, strip);
      if (strip + G < p.nstrips) body(b1, b0, strip + G);
    }
  } else {
    body(b0, b1, blockIdx.x);
  }
  STAMP(5);
}



// ---------------------------------------------------------------------------------------------------
// Fused SwiGLU MLP for M <= 2 rows (batch 1): wi_fused and wo in ONE persistent launch, the two phases
// separated by a grid barrier.  What it buys: the weight stream never stops.  Between two separate launches
// HBM idles for the tail of the first kernel (reduce + epilogue), the launch gap and the head of the
// second (dispatch, operand staging, first-byte latency) — about 7 us per layer; here every workgroup
// requests its share of the wo tiles BEFORE it arrives at the barrier, so the barrier's round trips are
// covered by that stream.
//   phase 1  = k_gemv_small<16, KPW1, 2, MULTI> with the SWIGLU epilogue; the hidden planes are written
//              with device-coherent (sc1) stores
//   barrier  = one relaxed agent-scope counter (stores acknowledged first, vmcnt 0), bounded spin
//   phase 2  = k_gemv_small<16, KPW2, 2> with two workgroups per strip (split-K 2, fence-free combine) and
//              the RESID_EMIT epilogue; the hidden planes are staged with sc1 loads
// The grid (2 * wo strips = 256 workgroups of 16 waves, one per CU) must be fully resident: the host checks it
// against the CU count; a spin that outlasts its bound raises an error word instead of hanging the GPU.
// MEASURED (in-kernel stamps, Dia-1.6B shapes): it LOSES to the two launches, 37 vs 27 us.  Phase 1 ends at
// 15 us (median), but the write-through stores of the hidden planes are acknowledged only at 17 us median /
// 26 us worst under the saturating weight stream, the barrier completes 3.6 us after the last arrival and the
// coherent re-read of the planes takes another 3.3 us.  A graph-replayed kernel boundary does the same hand-off
// in about 5 us.  The kernel stays as a tested experiment (engine: DIA_MLP_FUSE=1), not as the default.
struct MlpK { GemmK wi, wo; int* bar; };     // bar[0]: arrivals (monotonic, zeroed by the host per session), bar[1]: error flag

__device__ __forceinline__ void st16_agent(bf16_raw* dst, bf16x8 v) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(&v);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst), q[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst) + 1, q[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bf16x8 ld16_agent(const bf16_raw* src) {
  unsigned long long q[2];
  q[0] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  q[1] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(src) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *reinterpret_cast<const bf16x8*>(q);
}

template <int KPW1, int KPW2>
__global__ __launch_bounds__(1024) void k_mlp_fused(MlpK q) {
  constexpr int NW = 16, RS = 2, NT = NW * 64;
  constexpr int KT1 = NW * KPW1, KT2 = NW * KPW2;          // k-tiles of phase 1; k-tiles of ONE K half of phase 2
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                         // [NW][64]
  float* tile = reinterpret_cast<float*>(smem_raw + sizeof(f32x4) * NW * 64);   // [16][17]
  float* inv_s = tile + 16 * 17;                                           // [16]
  bf16x8* As = reinterpret_cast<bf16x8*>(smem_raw + sizeof(f32x4) * NW * 64 + sizeof(float) * (16 * 17 + 16));
  __shared__ int sk_flag;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int G = gridDim.x, wg = blockIdx.x;
  const int e_r = (tid >> 1) & 15, half = tid & 1, m = e_r;
  const bool e_thread = tid < 32;
  const int arow = min(lane & 15, RS - 1), akq = lane >> 4;
  STAMP(0);

  // phase 2's residual row and next-norm weight do not depend on phase 1: requested first (x is only written by
  // the phase-2 epilogue)
  const GemmK& p = q.wo;
  const int strip2 = wg % p.nstrips, ks = wg / p.nstrips;     // two workgroups per strip: K halves
  const bool live = e_thread && m < p.M;
  float xpre[8], gpre[8];
  if (e_thread) {
    const int n0 = strip2 * 16 + half * 8;
    const float* o = p.out + (long)(live ? m : 0) * p.ldo + n0;
    const float4 xa = *reinterpret_cast<const float4*>(o), xb = *reinterpret_cast<const float4*>(o + 4);
    xpre[0] = xa.x; xpre[1] = xa.y; xpre[2] = xa.z; xpre[3] = xa.w;
    xpre[4] = xb.x; xpre[5] = xb.y; xpre[6] = xb.z; xpre[7] = xb.w;
    const float4 ga = *reinterpret_cast<const float4*>(p.gnext + n0), gb = *reinterpret_cast<const float4*>(p.gnext + n0 + 4);
    gpre[0] = ga.x; gpre[1] = ga.y; gpre[2] = ga.z; gpre[3] = ga.w;
    gpre[4] = gb.x; gpre[5] = gb.y; gpre[6] = gb.z; gpre[7] = gb.w;
  }

  // ================= phase 1: h = silu(gate) * up,  [gate|up] = norm(x) . wi =================
  {
    const GemmK& p = q.wi;
    const bool live1 = e_thread && m < p.M;
    const bf16x8* Wl = reinterpret_cast<const bf16x8*>(p.W) + (long)(w * KPW1) * 64 + lane;
    auto load_strip = [&](bf16x8 (&b)[KPW1], int strip) {
      const bf16x8* Wt = Wl + (long)strip * p.KT * 64;
#pragma unroll
      for (int i = 0; i < KPW1; ++i) b[i] = DIA_WLOAD(Wt + (long)i * 64);
    };
    bf16x8 b0[KPW1], b1[KPW1];
    constexpr int nch = DIA_NPLANES * KT1 * 4 * RS, CH = (nch + NT - 1) / NT;
    bf16x8 v0[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int c = min(tid + u * NT, nch - 1);
      const int row = c % RS, kq = (c / RS) & 3, kt = (c / (4 * RS)) % KT1, pl = c / (4 * RS * KT1);
      v0[u] = *reinterpret_cast<const bf16x8*>(p.A + pl * p.a_plane_stride + ((long)kt * 64 + row + 16 * kq) * 8);
    }
    const int s_row = tid >> 3, s_part = tid & 7;
    float s0 = 0.f;
    if (tid < 128 && s_row < p.M)
      for (int i = s_part; i < p.ssq_in_n; i += 8) s0 += p.ssq_in[(long)i * p.ssq_ld + s_row];
    __builtin_amdgcn_sched_barrier(0);
    load_strip(b0, wg);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < CH; ++u)
      if (tid + u * NT < nch) As[tid + u * NT] = v0[u];
    s0 += __shfl_xor(s0, 1, 64);
    s0 += __shfl_xor(s0, 2, 64);
    s0 += __shfl_xor(s0, 4, 64);
    if (tid < 128 && s_part == 0) inv_s[s_row] = rsqrtf(s0 * p.inv_d + p.eps);
    lds_barrier();
    auto body = [&](bf16x8 (&bc)[KPW1], bf16x8 (&bn)[KPW1], int strip) {
      const int next = strip + G;
      if (next < p.nstrips) load_strip(bn, next);
      f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int i = 0; i < KPW1; ++i)
#pragma unroll
        for (int pl = 0; pl < DIA_NPLANES; ++pl) {
          const bf16x8 a = As[((pl * KT1 + w * KPW1 + i) * 4 + akq) * RS + arow];
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bc[i], acc[0], 0, 0, 0);
        }
      reduce_to_tile<1, NW, true>(acc, red, tile, tid, lane, w);
      if (live1 && half == 0) {          // SWIGLU (layers.py:95-101): 8 gate columns then 8 up columns per strip
        const float* trow = tile + e_r * 17;
        const float inv = inv_s[e_r];
        bf16x8 h, mi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float gte = trow[j] * inv, up = trow[8 + j] * inv;
          const float v = (gte / (1.0f + expf(-gte))) * up;
          __bf16 a, b, c;
          split3(v, a, b, c);
          h[j] = a; mi[j] = b; lo[j] = c;
        }
        const long off = plane_frag_off(m, strip * 8, p.p_ktiles);
        st16_agent(p.P + off, h);
        st16_agent(p.P + p.p_plane_stride + off, mi);
        st16_agent(p.P + 2 * p.p_plane_stride + off, lo);
      }
    };
    for (int strip = wg; strip < p.nstrips; strip += 2 * G) {
      body(b0, b1, strip);
      if (strip + G < p.nstrips) body(b1, b0, strip + G);
    }
  }

  STAMP(1);
  // ================= phase 2 weights: requested now, they stream while the barrier below completes =================
  // (the hidden-plane stores of phase 1 are acknowledged first, so that the barrier's arrival needs no further wait)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bf16x8 b2[KPW2];
  {
    const bf16x8* Wt = reinterpret_cast<const bf16x8*>(p.W) + ((long)strip2 * p.KT + ks * KT2 + w * KPW2) * 64 + lane;
#pragma unroll
    for (int i = 0; i < KPW2; ++i) b2[i] = DIA_WLOAD(Wt + (long)i * 64);
  }

  // ================= grid barrier: every hidden plane is written (and acknowledged) before anyone reads =================
  STAMP(2);
  lds_barrier();                                     // every wave's stores are acknowledged (waited above); b2 stays in flight
  if (tid == 0) {
    const int v = __hip_atomic_fetch_add(q.bar, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int target = (v / G + 1) * G;              // arrivals of this launch complete the current multiple of G
    int spins = 0;
    while (__hip_atomic_load(q.bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target < 0) {
      if (++spins > 400000) { __hip_atomic_store(q.bar + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      __builtin_amdgcn_s_sleep(1);
    }
  }
  lds_barrier();

  STAMP(3);
  // ================= phase 2: x += h . wo (this workgroup: one K half of one strip) =================
  {
    constexpr int nch = DIA_NPLANES * KT2 * 4 * RS, CH = (nch + NT - 1) / NT;
    bf16x8 v0[CH];
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int c = min(tid + u * NT, nch - 1);
      const int row = c % RS, kq = (c / RS) & 3, kt = (c / (4 * RS)) % KT2, pl = c / (4 * RS * KT2);
      v0[u] = ld16_agent(p.A + pl * p.a_plane_stride + ((long)(ks * KT2 + kt) * 64 + row + 16 * kq) * 8);
    }
#pragma unroll
    for (int u = 0; u < CH; ++u)
      if (tid + u * NT < nch) As[tid + u * NT] = v0[u];
    if (tid < 16) inv_s[tid] = 1.0f;
    lds_barrier();
    STAMP(4);
    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < KPW2; ++i)
#pragma unroll
      for (int pl = 0; pl < DIA_NPLANES; ++pl) {
        const bf16x8 a = As[((pl * KT2 + w * KPW2 + i) * 4 + akq) * RS + arow];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b2[i], acc[0], 0, 0, 0);
      }
    reduce_to_tile<1, NW, true>(acc, red, tile, tid, lane, w);
    // split-K combine over the two K halves (fence-free slab hand-off, as splitk_combine)
    const int SK = G / p.nstrips;
    if (SK > 1) {
      float* slab = p.sk_scratch + ((long)strip2 * SK + ks) * 256;
      if (tid < 128) {
        const int e = tid * 2;
        st2_agent(slab + e, tile[(e >> 4) * 17 + (e & 15)], tile[(e >> 4) * 17 + (e & 15) + 1]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(p.sk_tickets + strip2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == SK - 1;
        if (last) __hip_atomic_store(p.sk_tickets + strip2, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sk_flag = last;
      }
      __syncthreads();
      if (!sk_flag) return;
      if (tid < 128) {
        const int e = tid * 2;
        const float* base = p.sk_scratch + (long)strip2 * SK * 256 + e;
        float a = 0.f, b = 0.f;
        for (int k = 0; k < SK; ++k) { const float2 v = ld2_agent(base + k * 256); a += v.x; b += v.y; }
        tile[(e >> 4) * 17 + (e & 15)] = a; tile[(e >> 4) * 17 + (e & 15) + 1] = b;
      }
      __syncthreads();
    }
    if (e_thread) run_epilogue(p, tile + e_r * 17, 1.0f, m, strip2 * 16 + half * 8, half, strip2, live, xpre, gpre);
    STAMP(5);
  }
}

constexpr size_t mlp_smem(int kt1, int kt2) {
  return sizeof(f32x4) * 16 * 64 + sizeof(float) * (16 * 17 + 16) + (size_t)DIA_NPLANES * (kt1 > kt2 ? kt1 : kt2) * 4 * 2 * 16;
}

// ---------------------------------------------------------------------------------------------------
// M <= 4 GEMV over the ZERO-SKIPPING weight stream of an unstructured-pruned matrix (layout.sparse_tile_weight:
// per tile 64 lane masks + 4 row prefixes + the non-zero bf16 values, at most 1024 bytes; denser tiles raw).
// The persistent multi-strip form of k_gemv_small with a different B producer: one 16-byte load per lane still
// fetches a whole tile (lanes past the block re-read its last chunk — same cache line, no traffic), the loaded
// chunks are prefetched one strip ahead exactly like dense tiles, and each wave expands them through its own LDS
// scratch (write the chunks, read mask byte + row prefix, 4-step DPP scan for the lane's offset, eight 2-byte
// reads) into the MFMA B fragment.  The arithmetic and its order are those of the dense kernel: results are
// bit-identical to dia_gemm on the same (zero-holding) matrix, the stream is 0.59x the bytes at 50 % zeros.
// MEASURED (wi_fused 2048 x 16384, M = 2): 22.6 us at 50 % zeros, 21.8 us at 70 %, against 15.7 us for the dense
// stream — the expansion, not the bytes, is the limit: eight 2-byte LDS gathers per lane and tile (bank-conflicted,
// 2 300 LDS instructions per workgroup) cost more than the 28-41 MB they save at 4.4 TB/s.  Kept as a tested
// kernel-level experiment for SURVEY.md §8(f)-4; the engine streams unstructured-pruned checkpoints dense.
template <int KPW, int RS, int MAXS>
__global__ __launch_bounds__(1024) void k_gemv_sparse(GemmK p) {
  constexpr int NW = 16, KT = NW * KPW, NT = NW * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                         // [NW][64]
  float* tile = reinterpret_cast<float*>(smem_raw + sizeof(f32x4) * NW * 64);   // [16][17]
  float* inv_s = tile + 16 * 17;                                           // [16]
  bf16x8* As = reinterpret_cast<bf16x8*>(smem_raw + sizeof(f32x4) * NW * 64 + sizeof(float) * (16 * 17 + 16));
  unsigned char* dec = reinterpret_cast<unsigned char*>(As) + (size_t)DIA_NPLANES * KT * 4 * RS * 16;   // [NW][KPW][1024]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kt0 = w * KPW;
  const int G = gridDim.x;
  const int e_r = (tid >> 1) & 15, half = tid & 1, m = e_r;
  const bool e_thread = tid < 32;
  const bool live = e_thread && m < p.M;
  float xpre[8], gpre[8];

  // tile offsets of every strip this workgroup walks (a handful of words: no dependent load in the loop)
  unsigned int toff[MAXS][KPW];
#pragma unroll
  for (int sI = 0; sI < MAXS; ++sI) {
    const int strip = min(blockIdx.x + sI * G, p.nstrips - 1);
#pragma unroll
    for (int i = 0; i < KPW; ++i) toff[sI][i] = p.sp_toff[(long)strip * KT + kt0 + i];
  }
  // A image, row scales, residual operands: as k_gemv_small
  constexpr int CH = (3 * KPW * RS + 15) / 16;
  constexpr int nchunks = DIA_NPLANES * KT * 4 * RS;
  bf16x8 v0[CH];
#pragma unroll
  for (int u = 0; u < CH; ++u) {
    const int c = min(tid + u * NT, nchunks - 1);
    const int row = c % RS, kq = (c / RS) & 3, kt = (c / (4 * RS)) % KT, pl = c / (4 * RS * KT);
    v0[u] = *reinterpret_cast<const bf16x8*>(p.A + pl * p.a_plane_stride + ((long)kt * 64 + row + 16 * kq) * 8);
  }
  const bool has_norm = p.ssq_in != nullptr;
  const int s_row = tid >> 3, s_part = tid & 7;
  float s0 = 0.f;
  if (tid < 128 && has_norm && s_row < p.M)
    for (int i = s_part; i < p.ssq_in_n; i += 8) s0 += p.ssq_in[(long)i * p.ssq_ld + s_row];
  const bool resid = p.epi == DIA_EPI_RESID_EMIT;
  auto load_resid = [&](int strip) {
    const int n0 = strip * 16 + half * 8;
    const float* o = p.out + (long)(live ? m : 0) * p.ldo + n0;
    const float4 xa = *reinterpret_cast<const float4*>(o), xb = *reinterpret_cast<const float4*>(o + 4);
    xpre[0] = xa.x; xpre[1] = xa.y; xpre[2] = xa.z; xpre[3] = xa.w;
    xpre[4] = xb.x; xpre[5] = xb.y; xpre[6] = xb.z; xpre[7] = xb.w;
    const float4 ga = *reinterpret_cast<const float4*>(p.gnext + n0), gb = *reinterpret_cast<const float4*>(p.gnext + n0 + 4);
    gpre[0] = ga.x; gpre[1] = ga.y; gpre[2] = ga.z; gpre[3] = ga.w;
    gpre[4] = gb.x; gpre[5] = gb.y; gpre[6] = gb.z; gpre[7] = gb.w;
  };
  if (resid && e_thread) load_resid(blockIdx.x);
  auto load_blocks = [&](u32x4 (&b)[KPW], const unsigned int (&t)[KPW]) {
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
      const unsigned int nch = t[i] & 255u;
      const int l = min(lane, (int)(nch ? nch : 64u) - 1);
      b[i] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p.sp_blocks) + (long)(t[i] >> 8) + l);
    }
  };
  u32x4 b0[KPW], b1[KPW];
  load_blocks(b0, toff[0]);
#pragma unroll
  for (int u = 0; u < CH; ++u)
    if (tid + u * NT < nchunks) As[tid + u * NT] = v0[u];
  s0 += __shfl_xor(s0, 1, 64);
  s0 += __shfl_xor(s0, 2, 64);
  s0 += __shfl_xor(s0, 4, 64);
  if (tid < 128 && s_part == 0) inv_s[s_row] = has_norm ? rsqrtf(s0 * p.inv_d + p.eps) : 1.0f;
  lds_barrier();

  const int arow = min(lane & 15, RS - 1), akq = lane >> 4;
  unsigned char* dw = dec + (size_t)w * KPW * 1024;
  auto body = [&](u32x4 (&bc)[KPW], u32x4 (&bn)[KPW], const unsigned int (&tc)[KPW], const unsigned int (&tn)[KPW], int strip) {
    const int next = strip + G;
    if (next < p.nstrips) load_blocks(bn, tn);
    // expand this strip's tiles: chunks -> this wave's LDS scratch -> fragments
#pragma unroll
    for (int i = 0; i < KPW; ++i) *reinterpret_cast<u32x4*>(dw + i * 1024 + lane * 16) = bc[i];
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
      bf16x8 bfrag;
      if ((tc[i] & 255u) == 0u) {                       // raw tile (wave-uniform)
        bfrag = __builtin_bit_cast(bf16x8, bc[i]);
      } else {
        const unsigned char* blk = dw + i * 1024;
        const unsigned int mk = blk[lane];
        const int cnt = __popc(mk);
        int incl = cnt;                                 // inclusive scan over the 16-lane row
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);   // row_shr:1
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);   // row_shr:2
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);   // row_shr:4
        incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);   // row_shr:8
        const int base = reinterpret_cast<const unsigned short*>(blk + 64)[lane >> 4] + incl - cnt;
        const unsigned short* vals = reinterpret_cast<const unsigned short*>(blk + 80);
        unsigned short e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int rk = __popc(mk & ((1u << j) - 1u));
          const unsigned short v = vals[min(base + rk, 471)];
          e[j] = ((mk >> j) & 1u) ? v : (unsigned short)0;
        }
        u32x4 packed;
        packed[0] = e[0] | ((unsigned int)e[1] << 16); packed[1] = e[2] | ((unsigned int)e[3] << 16);
        packed[2] = e[4] | ((unsigned int)e[5] << 16); packed[3] = e[6] | ((unsigned int)e[7] << 16);
        bfrag = __builtin_bit_cast(bf16x8, packed);
      }
#pragma unroll
      for (int pl = 0; pl < DIA_NPLANES; ++pl) {
        const bf16x8 a = As[((pl * KT + kt0 + i) * 4 + akq) * RS + arow];
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bfrag, acc[0], 0, 0, 0);
      }
    }
    reduce_to_tile<1, NW, true>(acc, red, tile, tid, lane, w);
    if (e_thread) {
      const int n0 = strip * 16 + half * 8;
      run_epilogue(p, tile + e_r * 17, inv_s[e_r], m, n0, half, strip, live, xpre, gpre);
      if (next < p.nstrips && resid) load_resid(next);
    }
  };
#pragma unroll
  for (int sI = 0; sI < MAXS; sI += 2) {
    const int strip = blockIdx.x + sI * G;
    if (strip < p.nstrips) body(b0, b1, toff[sI], toff[sI + 1 < MAXS ? sI + 1 : sI], strip);
    if (strip + G < p.nstrips && sI + 1 < MAXS) body(b1, b0, toff[sI + 1], toff[sI + 2 < MAXS ? sI + 2 : sI + 1], strip + G);
  }
}

template <int KPW, int RS>
int launch_sparse(const GemmK& k, hipStream_t st) {
  constexpr int MAXS = 8;
  const size_t smem = sizeof(f32x4) * 16 * 64 + sizeof(float) * (16 * 17 + 16) + (size_t)DIA_NPLANES * (16 * KPW) * 4 * RS * 16 + (size_t)16 * KPW * 1024;
  int rc = dia_kernels_init_once();
  if (rc) return rc;
  int grid = (k.nstrips + MAXS - 1) / MAXS;
  if (grid < 256 && k.nstrips >= 256) grid = 256;
  if (grid > k.nstrips) grid = k.nstrips;
  if ((k.nstrips + grid - 1) / grid > MAXS) return dia_fail(DIA_E_ARG, "dia_gemm: too many strips for the sparse kernel");
  launch_kernel(k_gemv_sparse<KPW, RS, MAXS>, dim3(grid), dim3(1024), smem, st, k);
  return dia_check_launch("k_gemv_sparse");
}

// 5..16 rows (batch 3-8): one m-tile, A fragments held in registers for the workgroup's whole life
// (each wave owns a fixed K range of KPW k-tiles = 12*KPW VGPRs) and reused for every strip the
// workgroup walks; weight tiles double-buffered across strips like k_gemv_small.
#ifndef DIA_Z_TEMPORAL
#define DIA_Z_TEMPORAL 1
#endif
constexpr bool ZTEMPORAL = DIA_Z_TEMPORAL != 0;
template <int NW, int KPW, bool MULTI, bool MZ = false>
__global__ __launch_bounds__(NW * 64) void k_gemm16(const bf16_raw* a_A, long a_aps, const bf16_raw* a_W, int a_KT, int a_M, int a_epi,
                                                    int a_nstrips, float* a_out, int a_ldo, const float* a_gnext, GemmK p) {
  // (leading arguments = fields of p, preloaded into SGPRs: see k_gemv_small)
  p.A = a_A; p.a_plane_stride = a_aps; p.W = a_W; p.KT = a_KT; p.M = a_M; p.epi = a_epi; p.nstrips = a_nstrips;
  p.out = a_out; p.ldo = a_ldo; p.gnext = a_gnext;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                         // [NW][64]
  float* tile = reinterpret_cast<float*>(smem_raw + sizeof(f32x4) * NW * 64);   // [16][17]
  float* inv_s = tile + 16 * 17;                                           // [16]
  constexpr int NT = NW * 64;

  // 17..128 rows (batch 9-64): gridDim.z = 2..8, one m-tile per z.  The workgroups of a group stream the same weights
  // at the same time from different CUs of ONE XCD (gridDim.x * gridDim.y is a multiple of 8), so HBM sees each
  // byte once and the second reader is served by that XCD's L2; everything row-indexed is shifted by 16 rows.
  // (a separate instantiation: the 16-row kernels stay exactly as they were — the extra prologue cost 2 % of the
  // batch-8 step)
  if constexpr (MZ) {
    const int z = blockIdx.z;
    p.A += (long)z * p.a_ktiles * 512;
    p.M = min(16, p.M - 16 * z);
    if (p.ssq_in) p.ssq_in += 16 * z;
    if (p.out) p.out += (long)16 * z * p.ldo;
    if (p.P) p.P += (long)z * p.p_ktiles * 512;
    if (p.ssq_out) p.ssq_out += 16 * z;
    if (p.sk_scratch) { p.sk_scratch += (long)z * p.nstrips * gridDim.y * 256; p.sk_tickets += z * p.nstrips; }
  }
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int kt0 = blockIdx.y * (NW * KPW) + w * KPW;      // split-K over gridDim.y
  const int G = gridDim.x;
  __shared__ int sk_flag;
  const bf16x8* Wl = reinterpret_cast<const bf16x8*>(p.W) + (long)kt0 * 64 + lane;
  auto load_strip = [&](bf16x8* b, int strip) {
    const bf16x8* Wt = Wl + (long)strip * p.KT * 64;
#pragma unroll
    for (int i = 0; i < KPW; ++i) b[i] = (MZ && ZTEMPORAL) ? *(Wt + (long)i * 64) : DIA_WLOAD(Wt + (long)i * 64);   // z-form: the other m-tiles re-read these lines from L2
  };
  bf16x8 b0[KPW], b1[MULTI ? KPW : 1];

  const int e_r = (tid >> 1) & 15, half = tid & 1;
  const int m = e_r;
  const bool e_thread = tid < 32;
  const bool live = e_thread && m < p.M;
  float xpre[8], gpre[8];

  // A fragments (rows >= M alias the last valid row: no extra L2 traffic, results never stored)
  const int alane = (lane & 48) | min(lane & 15, p.M - 1);
  bf16x8 a[KPW][DIA_NPLANES];
#pragma unroll
  for (int i = 0; i < KPW; ++i)
#pragma unroll
    for (int pl = 0; pl < DIA_NPLANES; ++pl)
      a[i][pl] = *reinterpret_cast<const bf16x8*>(p.A + pl * p.a_plane_stride + ((long)(kt0 + i) * 64 + alane) * 8);
  // strip sums of squares: 8 threads per row
  const bool has_norm = p.ssq_in != nullptr;
  const int s_row = tid >> 3, s_part = tid & 7;
  const bool s_thread = tid < 128 && has_norm;
  float sq[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) sq[i] = 0.f;
  if (s_thread) {
    const float* sp = p.ssq_in + min(s_row, p.M - 1);
#pragma unroll
    for (int i = 0; i < 16; ++i) sq[i] = sp[(long)min(s_part + 8 * i, p.ssq_in_n - 1) * p.ssq_ld];
  }
  const bool resid = p.epi == DIA_EPI_RESID_EMIT;
  auto load_resid = [&](int strip) {
    const int n0 = strip * 16 + half * 8;
    const float* o = p.out + (long)(live ? m : 0) * p.ldo + n0;
    const float4 xa = *reinterpret_cast<const float4*>(o), xb = *reinterpret_cast<const float4*>(o + 4);
    xpre[0] = xa.x; xpre[1] = xa.y; xpre[2] = xa.z; xpre[3] = xa.w;
    xpre[4] = xb.x; xpre[5] = xb.y; xpre[6] = xb.z; xpre[7] = xb.w;
    const float4 ga = *reinterpret_cast<const float4*>(p.gnext + n0), gb = *reinterpret_cast<const float4*>(p.gnext + n0 + 4);
    gpre[0] = ga.x; gpre[1] = ga.y; gpre[2] = ga.z; gpre[3] = ga.w;
    gpre[4] = gb.x; gpre[5] = gb.y; gpre[6] = gb.z; gpre[7] = gb.w;
  };
  if (resid && e_thread) load_resid(blockIdx.x);
  __builtin_amdgcn_sched_barrier(0);
  load_strip(b0, blockIdx.x);
  __builtin_amdgcn_sched_barrier(0);
  {
    float s0 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s0 += (s_part + 8 * i < p.ssq_in_n && s_row < p.M) ? sq[i] : 0.f;
    if (s_thread && s_row < p.M)
      for (int idx = s_part + 128; idx < p.ssq_in_n; idx += 8) s0 += p.ssq_in[(long)idx * p.ssq_ld + s_row];
    s0 += __shfl_xor(s0, 1, 64);
    s0 += __shfl_xor(s0, 2, 64);
    s0 += __shfl_xor(s0, 4, 64);
    if (tid < 128 && s_part == 0) inv_s[s_row] = has_norm ? rsqrtf(s0 * p.inv_d + p.eps) : 1.0f;
  }

  auto body = [&](bf16x8* bc, bf16x8* bn, int strip) {
    const int next = strip + G;
    if constexpr (MULTI) { if (next < p.nstrips) load_strip(bn, next); }
    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < KPW; ++i)
#pragma unroll
      for (int pl = 0; pl < DIA_NPLANES; ++pl)
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][pl], bc[i], acc[0], 0, 0, 0);
    reduce_to_tile<1, NW, true>(acc, red, tile, tid, lane, w);
    const bool last_slice = splitk_combine(p, tile, strip, tid, &sk_flag);      // workgroup-uniform; true without split-K
    // (one element per thread as in k_gemv_small was measured here too: bit-identical, 1.3 % SLOWER at batch 8 - 768
    // two-byte plane stores per tile instead of 96 sixteen-byte ones)
    if (e_thread) {
      const int n0 = strip * 16 + half * 8;
      if (last_slice) run_epilogue(p, tile + e_r * 17, inv_s[e_r], m, n0, half, strip, live, xpre, gpre);
      if (MULTI && next < p.nstrips && resid) load_resid(next);
    }
  };
  if constexpr (MULTI) {
    for (int strip = blockIdx.x; strip < p.nstrips; strip += 2 * G) {
      body(b0, b1, strip);
      if (strip + G < p.nstrips) body(b1, b0, strip + G);
    }
  } else {
    body(b0, b1, blockIdx.x);
  }
}

// 17..32 rows (batch 9-16): two m-tiles, the k_gemm16 scheme with both tiles' A fragments in registers
// (24 * KPW VGPRs: 8 waves x 4 k-tiles), one strip per workgroup, K split over gridDim.y workgroups whose
// partial tiles meet through the fence-free slab hand-off (2 x 256 floats per slab).
template <int KPW>
__global__ __launch_bounds__(512) void k_gemm32(GemmK p) {
  constexpr int NW = 8, MT = 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                              // [NW][MT][64]
  float* tile = reinterpret_cast<float*>(smem_raw + sizeof(f32x4) * NW * MT * 64);   // [MT][16][17]
  float* inv_s = tile + MT * 16 * 17;                                           // [32]
  __shared__ int sk_flag;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int strip = blockIdx.x, ks = blockIdx.y, SK = gridDim.y;
  const int kt0 = ks * (NW * KPW) + w * KPW;
  const int e_mt = tid >> 5, e_r = (tid >> 1) & 15, half = tid & 1;
  const int m = e_mt * 16 + e_r;
  const bool e_thread = tid < 32 * MT;
  const bool live = e_thread && m < p.M;
  float xpre[8], gpre[8];
  // A fragments of both m-tiles (rows >= M alias the last valid row)
  bf16x8 a[MT][KPW][DIA_NPLANES];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = min(mt * 16 + (lane & 15), p.M - 1);
    const long aoff = ((long)(row >> 4) * p.a_ktiles * 64 + ((lane & 48) | (row & 15))) * 8;
#pragma unroll
    for (int i = 0; i < KPW; ++i)
#pragma unroll
      for (int pl = 0; pl < DIA_NPLANES; ++pl)
        a[mt][i][pl] = *reinterpret_cast<const bf16x8*>(p.A + pl * p.a_plane_stride + aoff + (long)(kt0 + i) * 512);
  }
  // row scales: 8 threads per row, 32 rows
  {
    const int s_row = tid >> 3, s_part = tid & 7;
    float s0 = 0.f;
    if (tid < 256 && p.ssq_in != nullptr && s_row < p.M)
      for (int i = s_part; i < p.ssq_in_n; i += 8) s0 += p.ssq_in[(long)i * p.ssq_ld + s_row];
    s0 += __shfl_xor(s0, 1, 64);
    s0 += __shfl_xor(s0, 2, 64);
    s0 += __shfl_xor(s0, 4, 64);
    if (tid < 256 && s_part == 0) inv_s[s_row] = (p.ssq_in != nullptr) ? rsqrtf(s0 * p.inv_d + p.eps) : 1.0f;
  }
  if (p.epi == DIA_EPI_RESID_EMIT && e_thread) {
    const int n0 = strip * 16 + half * 8;
    const float* o = p.out + (long)(live ? m : 0) * p.ldo + n0;
    const float4 xa = *reinterpret_cast<const float4*>(o), xb = *reinterpret_cast<const float4*>(o + 4);
    xpre[0] = xa.x; xpre[1] = xa.y; xpre[2] = xa.z; xpre[3] = xa.w;
    xpre[4] = xb.x; xpre[5] = xb.y; xpre[6] = xb.z; xpre[7] = xb.w;
#pragma unroll
    for (int j = 0; j < 8; ++j) gpre[j] = p.gnext ? p.gnext[n0 + j] : 1.0f;
  }
  __builtin_amdgcn_sched_barrier(0);
  bf16x8 b[KPW];
  {
    const bf16x8* Wt = reinterpret_cast<const bf16x8*>(p.W) + ((long)strip * p.KT + kt0) * 64 + lane;
#pragma unroll
    for (int i = 0; i < KPW; ++i) b[i] = DIA_WLOAD(Wt + (long)i * 64);
  }
  f32x4 acc[MT] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int i = 0; i < KPW; ++i)
#pragma unroll
    for (int pl = 0; pl < DIA_NPLANES; ++pl)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][i][pl], b[i], acc[mt], 0, 0, 0);
  reduce_to_tile<MT, NW, true>(acc, red, tile, tid, lane, w);
  if (SK > 1) {        // as splitk_combine, two tiles per slab
    float* slab = p.sk_scratch + ((long)strip * SK + ks) * (MT * 256);
    if (tid < MT * 128) {
      const int t = tid >> 7, e = (tid & 127) * 2;
      st2_agent(slab + t * 256 + e, tile[(t * 16 + (e >> 4)) * 17 + (e & 15)], tile[(t * 16 + (e >> 4)) * 17 + (e & 15) + 1]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const int ticket = __hip_atomic_fetch_add(p.sk_tickets + strip, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == SK - 1;
      if (last) __hip_atomic_store(p.sk_tickets + strip, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      sk_flag = last;
    }
    __syncthreads();
    if (!sk_flag) return;
    if (tid < MT * 128) {
      const int t = tid >> 7, e = (tid & 127) * 2;
      const float* base = p.sk_scratch + (long)strip * SK * (MT * 256) + t * 256 + e;
      float x0 = 0.f, x1 = 0.f;
      for (int k = 0; k < SK; ++k) { const float2 v = ld2_agent(base + (long)k * (MT * 256)); x0 += v.x; x1 += v.y; }
      tile[(t * 16 + (e >> 4)) * 17 + (e & 15)] = x0; tile[(t * 16 + (e >> 4)) * 17 + (e & 15) + 1] = x1;
    }
    __syncthreads();
  }
  if (e_thread) run_epilogue(p, tile + (e_mt * 16 + e_r) * 17, inv_s[e_mt * 16 + e_r], m, strip * 16 + half * 8, half, strip, live, xpre, gpre);
}

template <int KPW>
int launch_g32(const GemmK& k, int sk, hipStream_t st) {
  const size_t smem = sizeof(f32x4) * 8 * 2 * 64 + sizeof(float) * (2 * 16 * 17 + 32);
  launch_kernel(k_gemm32<KPW>, dim3(k.nstrips, sk), dim3(512), smem, st, k);
  return dia_check_launch("k_gemm32");
}

// 17..32 rows, K = 2048-class shapes with many strips (qkv, wi, logits at batch 9-16): the persistent form of
// k_gemm32.  A one-strip workgroup would re-read the whole 393 KB activation image per strip (measured: wi 47 us,
// worse than the generic kernel's 38); here a workgroup keeps its K half of both m-tiles in registers (8 waves x
// 4 k-tiles, 96 VGPRs) and walks strips blockIdx.x, +gridDim.x, ... with double-buffered weight tiles; the two
// K halves of a strip (gridDim.y = 2) meet through the fence-free slab hand-off, strip by strip.
__global__ __launch_bounds__(512) void k_gemm32m(GemmK p) {
  constexpr int NW = 8, MT = 2, KPW = 4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                              // [NW][MT][64]
  float* tile = reinterpret_cast<float*>(smem_raw + sizeof(f32x4) * NW * MT * 64);   // [MT][16][17]
  float* inv_s = tile + MT * 16 * 17;                                           // [32]
  __shared__ int sk_flag;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int ks = blockIdx.y, SK = gridDim.y, G = gridDim.x;
  const int kt0 = ks * (NW * KPW) + w * KPW;
  const int e_mt = tid >> 5, e_r = (tid >> 1) & 15, half = tid & 1;
  const int m = e_mt * 16 + e_r;
  const bool e_thread = tid < 32 * MT;
  const bool live = e_thread && m < p.M;
  const bool resid = p.epi == DIA_EPI_RESID_EMIT;
  float xpre[8], gpre[8];
  bf16x8 a[MT][KPW][DIA_NPLANES];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int row = min(mt * 16 + (lane & 15), p.M - 1);
    const long aoff = ((long)(row >> 4) * p.a_ktiles * 64 + ((lane & 48) | (row & 15))) * 8;
#pragma unroll
    for (int i = 0; i < KPW; ++i)
#pragma unroll
      for (int pl = 0; pl < DIA_NPLANES; ++pl)
        a[mt][i][pl] = *reinterpret_cast<const bf16x8*>(p.A + pl * p.a_plane_stride + aoff + (long)(kt0 + i) * 512);
  }
  {
    const int s_row = tid >> 3, s_part = tid & 7;
    float s0 = 0.f;
    if (tid < 256 && p.ssq_in != nullptr && s_row < p.M)
      for (int i = s_part; i < p.ssq_in_n; i += 8) s0 += p.ssq_in[(long)i * p.ssq_ld + s_row];
    s0 += __shfl_xor(s0, 1, 64);
    s0 += __shfl_xor(s0, 2, 64);
    s0 += __shfl_xor(s0, 4, 64);
    if (tid < 256 && s_part == 0) inv_s[s_row] = (p.ssq_in != nullptr) ? rsqrtf(s0 * p.inv_d + p.eps) : 1.0f;
  }
  const bf16x8* Wl = reinterpret_cast<const bf16x8*>(p.W) + (long)kt0 * 64 + lane;
  auto load_strip = [&](bf16x8 (&b)[KPW], int strip) {
    const bf16x8* Wt = Wl + (long)strip * p.KT * 64;
#pragma unroll
    for (int i = 0; i < KPW; ++i) b[i] = DIA_WLOAD(Wt + (long)i * 64);
  };
  auto load_resid = [&](int strip) {
    const int n0 = strip * 16 + half * 8;
    const float* o = p.out + (long)(live ? m : 0) * p.ldo + n0;
    const float4 xa = *reinterpret_cast<const float4*>(o), xb = *reinterpret_cast<const float4*>(o + 4);
    xpre[0] = xa.x; xpre[1] = xa.y; xpre[2] = xa.z; xpre[3] = xa.w;
    xpre[4] = xb.x; xpre[5] = xb.y; xpre[6] = xb.z; xpre[7] = xb.w;
#pragma unroll
    for (int j = 0; j < 8; ++j) gpre[j] = p.gnext ? p.gnext[n0 + j] : 1.0f;
  };
  bf16x8 b0[KPW], b1[KPW];
  if (resid && e_thread) load_resid(blockIdx.x);
  __builtin_amdgcn_sched_barrier(0);
  load_strip(b0, blockIdx.x);
  __builtin_amdgcn_sched_barrier(0);
  auto body = [&](bf16x8 (&bc)[KPW], bf16x8 (&bn)[KPW], int strip) {
    const int next = strip + G;
    if (next < p.nstrips) load_strip(bn, next);
    f32x4 acc[MT] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < KPW; ++i)
#pragma unroll
      for (int pl = 0; pl < DIA_NPLANES; ++pl)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mt][i][pl], bc[i], acc[mt], 0, 0, 0);
    reduce_to_tile<MT, NW, true>(acc, red, tile, tid, lane, w);
    bool last_slice = true;
    if (SK > 1) {
      float* slab = p.sk_scratch + ((long)strip * SK + ks) * (MT * 256);
      if (tid < MT * 128) {
        const int t = tid >> 7, e = (tid & 127) * 2;
        st2_agent(slab + t * 256 + e, tile[(t * 16 + (e >> 4)) * 17 + (e & 15)], tile[(t * 16 + (e >> 4)) * 17 + (e & 15) + 1]);
      }
      // only the slab stores need their acknowledgement here; the weight tiles of the next strip, requested
      // before them, are older and complete first (in-order), so this costs the store latency only
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(p.sk_tickets + strip, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == SK - 1;
        if (last) __hip_atomic_store(p.sk_tickets + strip, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sk_flag = last;
      }
      __syncthreads();
      last_slice = sk_flag != 0;
      if (last_slice) {
        if (tid < MT * 128) {
          const int t = tid >> 7, e = (tid & 127) * 2;
          const float* base = p.sk_scratch + (long)strip * SK * (MT * 256) + t * 256 + e;
          float x0 = 0.f, x1 = 0.f;
          for (int k = 0; k < SK; ++k) { const float2 v = ld2_agent(base + (long)k * (MT * 256)); x0 += v.x; x1 += v.y; }
          tile[(t * 16 + (e >> 4)) * 17 + (e & 15)] = x0; tile[(t * 16 + (e >> 4)) * 17 + (e & 15) + 1] = x1;
        }
        __syncthreads();
      }
    }
    if (e_thread) {
      if (last_slice) run_epilogue(p, tile + (e_mt * 16 + e_r) * 17, inv_s[e_mt * 16 + e_r], m, strip * 16 + half * 8, half, strip, live, xpre, gpre);
      if (next < p.nstrips && resid) load_resid(next);
    }
  };
  for (int strip = blockIdx.x; strip < p.nstrips; strip += 2 * G) {
    body(b0, b1, strip);
    if (strip + G < p.nstrips) body(b1, b0, strip + G);
  }
}

int launch_g32m(const GemmK& k, int sk, hipStream_t st) {
  const size_t smem = sizeof(f32x4) * 8 * 2 * 64 + sizeof(float) * (2 * 16 * 17 + 32);
  const int gx = k.nstrips < 128 ? k.nstrips : 128;
  launch_kernel(k_gemm32m, dim3(gx, sk), dim3(512), smem, st, k);
  return dia_check_launch("k_gemm32m");
}

// ---------------------------------------------------------------------------------------------------
// Prefill GEMM (encoder layers, cross-K/V projections: M = text bytes, tens to thousands of rows).
// Here the contraction is dense and MFMA is the roofline, not HBM: a workgroup owns a 64-row x 256-column
// output block; its 8 waves form a 2 x 4 grid, each wave = 2 m-tiles x 4 strips of 16 columns, running the
// WHOLE K loop for its sub-block (no cross-wave reduction).  The activation planes of the 64 rows
// (3 planes x 4 m-tiles x 1 KiB per k-tile) are staged through LDS in chunks of GT_KC k-tiles,
// double-buffered, loaded two chunks ahead; every A fragment read from LDS feeds FOUR MFMAs (one per
// strip), so LDS read time (6 KiB per k-tile and wave) is half the MFMA time (24 x 16 cycles) — with two
// strips per fragment the two were equal and the kernel ran at a third of the MFMA bound.  Weight tiles come
// straight from global memory into the B operand registers, PD chunks ahead; the two wave rows and the row
// groups of the same column block re-read them from L2.  fp32-exact like every other GEMM here: 3 planes
// x bf16 weights.
constexpr int GT_MT = 4, GT_WM = 2, GT_WS = 4;   // workgroup: 4 m-tiles; per wave: 2 m-tiles x 4 strips
constexpr size_t gt_abuf(int kc) { return (size_t)kc * DIA_NPLANES * GT_MT * 64 * 16; }    // bytes of one staged chunk (24 KiB at 2 k-tiles)
constexpr size_t gt_smem(int kc, int nw) { return 2 * gt_abuf(kc) + sizeof(float) * (nw * 2 * 16 * 17 + 64); }

// GT_KC k-tiles per staged chunk, weight tiles PD chunks ahead, WPE waves per SIMD (2 = one workgroup per CU)
template <int GT_KC, int PD, int WPE, int GT_NW>
__global__ __launch_bounds__(GT_NW * 64) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void k_gemm_tile(GemmK p) {
  constexpr size_t GT_ABUF = gt_abuf(GT_KC);
  constexpr int GT_NT = GT_NW * 64, GT_WC = GT_NW / 2;                 // wave grid 2 x GT_WC
  constexpr int NPIECE = GT_KC * DIA_NPLANES * GT_MT * 64 / GT_NT;     // 16-byte pieces per thread and chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* abuf = reinterpret_cast<u32x4*>(smem_raw);                                     // [2][KC][3][MT][64] x 16 B
  float* tiles = reinterpret_cast<float*>(smem_raw + 2 * GT_ABUF);                      // [NW][2][16][17]
  float* inv_s = tiles + GT_NW * 2 * 16 * 17;                                           // [64]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int wr = w / GT_WC, wc = w % GT_WC;                                                    // wave row (m-tiles 2wr, 2wr+1), wave column
  const int mt0 = blockIdx.y * GT_MT;
  const int mtiles = (p.M + 15) >> 4;
  const int s0 = blockIdx.x * (GT_WC * GT_WS) + wc * GT_WS;
  const int nchunks = p.KT / GT_KC;                                                     // KT % 8 == 0 (dispatcher)

  // ---- A staging: pieces of 16 bytes, NPIECE per thread: piece -> (k-tile, plane, m-tile, lane)
  const u32x4* asrc[NPIECE];
  int adst[NPIECE];
#pragma unroll
  for (int j = 0; j < NPIECE; ++j) {
    const int i = tid + GT_NT * j;
    const int ln = i & 63, blk = i >> 6, mt = blk & 3, pl = (blk >> 2) % 3, kk = blk / 12;
    const int mtile = min(mt0 + mt, mtiles - 1);
    asrc[j] = reinterpret_cast<const u32x4*>(p.A + pl * p.a_plane_stride + (((long)mtile * p.a_ktiles + kk) * 64 + ln) * 8);
    adst[j] = ((kk * DIA_NPLANES + pl) * GT_MT + mt) * 64 + ln;
  }
  auto a_load = [&](u32x4 (&r)[NPIECE], int chunk) {
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) r[j] = asrc[j][(long)chunk * GT_KC * 64];          // k-tile stride = 64 pieces
  };
  auto a_store = [&](const u32x4 (&r)[NPIECE], int buf) {
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) abuf[buf * (GT_ABUF / 16) + adst[j]] = r[j];
  };
  const bf16x8* Wl = reinterpret_cast<const bf16x8*>(p.W) + lane;
  long woff[GT_WS];
#pragma unroll
  for (int j = 0; j < GT_WS; ++j) woff[j] = (long)min(s0 + j, p.nstrips - 1) * p.KT * 64;   // clamped for the loads
  auto b_load = [&](bf16x8 (&b)[GT_KC][GT_WS], int chunk) {
#pragma unroll
    for (int kk = 0; kk < GT_KC; ++kk)
#pragma unroll
      for (int j = 0; j < GT_WS; ++j) b[kk][j] = Wl[woff[j] + (long)(chunk * GT_KC + kk) * 64];
  };

  u32x4 areg0[NPIECE], areg1[NPIECE];
  bf16x8 bq[PD][GT_KC][GT_WS];
  a_load(areg0, 0);
  if (nchunks > 1) a_load(areg1, 1);
#pragma unroll
  for (int j = 0; j < PD; ++j) if (j < nchunks) b_load(bq[j], j);

  // RMSNorm scale of the 64 rows (8 threads per row sum the strip partials in fixed order)
  for (int t = tid; t < 64 * 8; t += GT_NT) {
    const int r = t >> 3, part = t & 7, row = mt0 * 16 + r;
    float sA = 0.f;
    if (p.ssq_in != nullptr && row < p.M)
      for (int i = part; i < p.ssq_in_n; i += 8) sA += p.ssq_in[(long)i * p.ssq_ld + row];
    sA += __shfl_xor(sA, 1, 64);
    sA += __shfl_xor(sA, 2, 64);
    sA += __shfl_xor(sA, 4, 64);
    if (part == 0) inv_s[r] = (p.ssq_in != nullptr) ? rsqrtf(sA * p.inv_d + p.eps) : 1.0f;
  }
  a_store(areg0, 0);
  __syncthreads();

  f32x4 acc[GT_WM][GT_WS];
#pragma unroll
  for (int i = 0; i < GT_WM; ++i)
#pragma unroll
    for (int j = 0; j < GT_WS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto chunk_body = [&](int c, auto Q) {
    constexpr int q = decltype(Q)::value;           // q = c mod PD, a literal at every call: buffer parities are compile-time
    if (c + 2 < nchunks) { if constexpr ((q & 1) == 0) a_load(areg0, c + 2); else a_load(areg1, c + 2); }
    const u32x4* ab = abuf + (q & 1) * (GT_ABUF / 16);
#pragma unroll
    for (int kk = 0; kk < GT_KC; ++kk)
#pragma unroll
      for (int pl = 0; pl < DIA_NPLANES; ++pl)
#pragma unroll
        for (int i = 0; i < GT_WM; ++i) {
          const u32x4 av = ab[((kk * DIA_NPLANES + pl) * GT_MT + wr * GT_WM + i) * 64 + lane];
          const bf16x8 a = __builtin_bit_cast(bf16x8, av);
#pragma unroll
          for (int j = 0; j < GT_WS; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq[q][kk][j], acc[i][j], 0, 0, 0);
        }
    if (c + PD < nchunks) b_load(bq[q], c + PD);
    if (c + 1 < nchunks) { if constexpr ((q & 1) == 0) a_store(areg1, 1); else a_store(areg0, 0); }
    lds_barrier();
  };
  static_assert(PD == 2 || PD == 4, "ring depth");
  for (int c0 = 0; c0 < nchunks; c0 += PD) {
    chunk_body(c0, std::integral_constant<int, 0>{});
    if (c0 + 1 < nchunks) chunk_body(c0 + 1, std::integral_constant<int, 1>{});
    if constexpr (PD == 4) {
      if (c0 + 2 < nchunks) chunk_body(c0 + 2, std::integral_constant<int, 2>{});
      if (c0 + 3 < nchunks) chunk_body(c0 + 3, std::integral_constant<int, 3>{});
    }
  }

  // ---- epilogue, per wave: two 16x16 tiles (a strip pair of one m-tile) at a time through this wave's LDS tiles
  float* tw = tiles + w * (2 * 16 * 17);
  const int et = lane >> 5, e_r = (lane >> 1) & 15, half = lane & 1;
  const int col = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < GT_WM; ++i) {
#pragma unroll
    for (int pr = 0; pr < GT_WS / 2; ++pr) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) tw[j * (16 * 17) + (r0 + r) * 17 + col] = acc[i][2 * pr + j][r];
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int mtl = wr * GT_WM + i;
      const int m = (mt0 + mtl) * 16 + e_r;
      const int strip = s0 + 2 * pr + et;
      const bool live = m < p.M && strip < p.nstrips;
      const int n0 = strip * 16 + half * 8;
      float xpre[8], gpre[8];
      if (p.epi == DIA_EPI_RESID_EMIT && live) {
        const float* o = p.out + (long)m * p.ldo + n0;
        const float4 xa = *reinterpret_cast<const float4*>(o), xb = *reinterpret_cast<const float4*>(o + 4);
        xpre[0] = xa.x; xpre[1] = xa.y; xpre[2] = xa.z; xpre[3] = xa.w;
        xpre[4] = xb.x; xpre[5] = xb.y; xpre[6] = xb.z; xpre[7] = xb.w;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) gpre[jj] = p.gnext ? p.gnext[n0 + jj] : 1.0f;
      }
      run_epilogue(p, tw + et * (16 * 17) + e_r * 17, inv_s[mtl * 16 + e_r], m, n0, half, min(strip, p.nstrips - 1), live, xpre, gpre);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// Wave-specialised form of k_gemm_tile: 8 consumer waves (the 2 x 4 grid above: weight tiles + MFMA only) and
// 2 producer waves that do nothing but stage the activation planes global -> registers -> LDS.  vmcnt retires in
// order, so in the plain form the wait for a staged chunk also waits for the weight tiles requested just before
// it, and a chunk lasts about one memory latency; here the producers' counter sees A loads only and the
// consumers never wait for A at all — the two kinds of wave meet at the chunk barrier.
template <int GT_KC, int PD, int NPW>
__global__ __launch_bounds__((8 + NPW) * 64) void k_gemm_tile_ws(GemmK p) {
  constexpr int NWC = 8, NPROD = NPW * 64;       // NPW producer waves
  constexpr size_t ABUF = gt_abuf(GT_KC);
  constexpr int NPIECE = GT_KC * DIA_NPLANES * GT_MT * 64 / NPROD;     // 16-byte pieces per producer thread and chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* abuf = reinterpret_cast<u32x4*>(smem_raw);                     // [2][KC][3][MT][64] x 16 B
  float* tiles = reinterpret_cast<float*>(smem_raw + 2 * ABUF);          // [NWC][2][16][17]
  float* inv_s = tiles + NWC * 2 * 16 * 17;                              // [64]

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int mt0 = blockIdx.y * GT_MT;
  const int mtiles = (p.M + 15) >> 4;
  const int nchunks = p.KT / GT_KC;

  // RMSNorm scale of the 64 rows (8 threads per row sum the strip partials in fixed order)
  for (int t = tid; t < 64 * 8; t += (NWC + NPW) * 64) {
    const int r = t >> 3, part = t & 7, row = mt0 * 16 + r;
    float sA = 0.f;
    if (p.ssq_in != nullptr && row < p.M)
      for (int i = part; i < p.ssq_in_n; i += 8) sA += p.ssq_in[(long)i * p.ssq_ld + row];
    sA += __shfl_xor(sA, 1, 64);
    sA += __shfl_xor(sA, 2, 64);
    sA += __shfl_xor(sA, 4, 64);
    if (part == 0) inv_s[r] = (p.ssq_in != nullptr) ? rsqrtf(sA * p.inv_d + p.eps) : 1.0f;
  }

  if (w >= NWC) {
    // ================= producers =================
    const int pt = tid - NWC * 64;
    const u32x4* asrc[NPIECE];
    int adst[NPIECE];
#pragma unroll
    for (int j = 0; j < NPIECE; ++j) {
      const int i = pt + NPROD * j;
      const int ln = i & 63, blk = i >> 6, mt = blk & 3, pl = (blk >> 2) % 3, kk = blk / 12;
      const int mtile = min(mt0 + mt, mtiles - 1);
      asrc[j] = reinterpret_cast<const u32x4*>(p.A + pl * p.a_plane_stride + (((long)mtile * p.a_ktiles + kk) * 64 + ln) * 8);
      adst[j] = ((kk * DIA_NPLANES + pl) * GT_MT + mt) * 64 + ln;
    }
    auto a_load = [&](u32x4 (&r)[NPIECE], int chunk) {
#pragma unroll
      for (int j = 0; j < NPIECE; ++j) r[j] = asrc[j][(long)chunk * GT_KC * 64];
    };
    auto a_store = [&](const u32x4 (&r)[NPIECE], int buf) {
#pragma unroll
      for (int j = 0; j < NPIECE; ++j) abuf[buf * (ABUF / 16) + adst[j]] = r[j];
    };
    u32x4 r0[NPIECE], r1[NPIECE];
    a_load(r0, 0);
    if (nchunks > 1) a_load(r1, 1);
    a_store(r0, 0);
    lds_barrier();                                    // chunk 0 staged (all 10 waves)
    for (int c = 0; c < nchunks; c += 2) {
      // during chunk c: chunk c+1 goes to buffer 1, chunk c+2 is requested
      if (c + 2 < nchunks) a_load(r0, c + 2);
      if (c + 1 < nchunks) a_store(r1, 1);
      lds_barrier();                                  // end of chunk c
      if (c + 1 < nchunks) {
        if (c + 3 < nchunks) a_load(r1, c + 3);
        if (c + 2 < nchunks) a_store(r0, 0);
        lds_barrier();                                // end of chunk c+1
      }
    }
    return;
  }

  // ================= consumers =================
  const int wr = w >> 2, wc = w & 3;
  const int s0 = blockIdx.x * 16 + wc * GT_WS;
  const bf16x8* Wl = reinterpret_cast<const bf16x8*>(p.W) + lane;
  long woff[GT_WS];
#pragma unroll
  for (int j = 0; j < GT_WS; ++j) woff[j] = (long)min(s0 + j, p.nstrips - 1) * p.KT * 64;
  auto b_load = [&](bf16x8 (&b)[GT_KC][GT_WS], int chunk) {
#pragma unroll
    for (int kk = 0; kk < GT_KC; ++kk)
#pragma unroll
      for (int j = 0; j < GT_WS; ++j) b[kk][j] = Wl[woff[j] + (long)(chunk * GT_KC + kk) * 64];
  };
  bf16x8 bq[PD][GT_KC][GT_WS];
#pragma unroll
  for (int j = 0; j < PD; ++j) if (j < nchunks) b_load(bq[j], j);
  f32x4 acc[GT_WM][GT_WS];
#pragma unroll
  for (int i = 0; i < GT_WM; ++i)
#pragma unroll
    for (int j = 0; j < GT_WS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  lds_barrier();                                      // chunk 0 staged; the weight loads stay in flight
  auto chunk_body = [&](int c, auto Q) {
    constexpr int q = decltype(Q)::value;
    const u32x4* ab = abuf + (q & 1) * (ABUF / 16);
#pragma unroll
    for (int kk = 0; kk < GT_KC; ++kk)
#pragma unroll
      for (int pl = 0; pl < DIA_NPLANES; ++pl)
#pragma unroll
        for (int i = 0; i < GT_WM; ++i) {
          const u32x4 av = ab[((kk * DIA_NPLANES + pl) * GT_MT + wr * GT_WM + i) * 64 + lane];
          const bf16x8 a = __builtin_bit_cast(bf16x8, av);
#pragma unroll
          for (int j = 0; j < GT_WS; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bq[q][kk][j], acc[i][j], 0, 0, 0);
        }
    if (c + PD < nchunks) b_load(bq[q], c + PD);
    lds_barrier();
  };
  static_assert(PD == 2 || PD == 4, "ring depth");
  for (int c0 = 0; c0 < nchunks; c0 += PD) {
    chunk_body(c0, std::integral_constant<int, 0>{});
    if (c0 + 1 < nchunks) chunk_body(c0 + 1, std::integral_constant<int, 1>{});
    if constexpr (PD == 4) {
      if (c0 + 2 < nchunks) chunk_body(c0 + 2, std::integral_constant<int, 2>{});
      if (c0 + 3 < nchunks) chunk_body(c0 + 3, std::integral_constant<int, 3>{});
    }
  }
  // ---- epilogue (as k_gemm_tile)
  float* tw = tiles + w * (2 * 16 * 17);
  const int et = lane >> 5, e_r = (lane >> 1) & 15, half = lane & 1;
  const int col = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
  for (int i = 0; i < GT_WM; ++i) {
#pragma unroll
    for (int pr = 0; pr < GT_WS / 2; ++pr) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) tw[j * (16 * 17) + (r0 + r) * 17 + col] = acc[i][2 * pr + j][r];
      __builtin_amdgcn_wave_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int mtl = wr * GT_WM + i;
      const int m = (mt0 + mtl) * 16 + e_r;
      const int strip = s0 + 2 * pr + et;
      const bool live = m < p.M && strip < p.nstrips;
      const int n0 = strip * 16 + half * 8;
      float xpre[8], gpre[8];
      if (p.epi == DIA_EPI_RESID_EMIT && live) {
        const float* o = p.out + (long)m * p.ldo + n0;
        const float4 xa = *reinterpret_cast<const float4*>(o), xb = *reinterpret_cast<const float4*>(o + 4);
        xpre[0] = xa.x; xpre[1] = xa.y; xpre[2] = xa.z; xpre[3] = xa.w;
        xpre[4] = xb.x; xpre[5] = xb.y; xpre[6] = xb.z; xpre[7] = xb.w;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) gpre[jj] = p.gnext ? p.gnext[n0 + jj] : 1.0f;
      }
      run_epilogue(p, tw + et * (16 * 17) + e_r * 17, inv_s[mtl * 16 + e_r], m, n0, half, min(strip, p.nstrips - 1), live, xpre, gpre);
      __builtin_amdgcn_wave_barrier();
    }
  }
}

template <int KC, int PD, int NPW>
int launch_tile_ws(const GemmK& k, hipStream_t st) {
  const int mgroups = ((k.M + 15) / 16 + GT_MT - 1) / GT_MT;
  launch_kernel(k_gemm_tile_ws<KC, PD, NPW>, dim3((k.nstrips + 15) / 16, mgroups), dim3((8 + NPW) * 64), gt_smem(KC, 8), st, k);
  return dia_check_launch("k_gemm_tile_ws");
}

// 17..32 rows in decode (batch 9-16).  With two m-tiles the activation image (3 planes x 32 rows x K) is 393 KB at
// K = 2048: a workgroup that splits K over its waves for ONE strip pulls all of it through L2 -> CU for 64 KB of
// weights.  Here the waves own STRIPS (WS each: a 32-row x 128*WS-column block per workgroup) and K is cut into
// ranges of KR k-tiles, one workgroup each (gridDim.z), so a workgroup needs only its range of the image, shared
// by its 8 waves through LDS.  A range is short (8-16 k-tiles), so nothing is pipelined: every load of the
// workgroup — the image pieces first, then all KR*WS weight tiles of each wave — is issued at once, the pieces go
// to LDS while the weights are still in flight (vmcnt retires in order), one barrier, then the MFMAs.  The partial
// blocks (same lane <-> same output element in every range) meet once: coherent slab stores, a ticket per column
// block, the last arriver adds the slabs in range order (bit-reproducible) and runs the epilogues.
template <int KR, int WS>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k_gemm_blk32(GemmK p) {
  constexpr int NW = 8, MT = 2;
  constexpr int NA = KR * DIA_NPLANES * MT * 64;                               // 16-byte pieces of one K range of the image
  constexpr int NPIECE = NA / 512;
  constexpr int FPT = MT * WS * 4;                                             // partial sums per thread
  static_assert(NA % 512 == 0, "pieces per thread");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* abuf = reinterpret_cast<u32x4*>(smem_raw);                            // [KR][3][MT][64] x 16 B
  float* tiles = reinterpret_cast<float*>(smem_raw + (size_t)NA * 16);         // [NW][2][16][17]
  float* inv_s = tiles + NW * 2 * 16 * 17;                                     // [32]
  __shared__ int sk_flag;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int SK = gridDim.z, ks = blockIdx.z;
  const int ktb = ks * KR;                                                     // first k-tile of this workgroup's range
  const int s0 = blockIdx.x * (NW * WS) + w * WS;

  u32x4 areg[NPIECE];
#pragma unroll
  for (int j = 0; j < NPIECE; ++j) {                                           // piece i -> (k-tile, plane, m-tile, lane) = its LDS slot
    const int i = tid + 512 * j;
    const int ln = i & 63, blk = i >> 6, mt = blk % MT, pl = (blk / MT) % DIA_NPLANES, kk = blk / (MT * DIA_NPLANES);
    const int row = min(mt * 16 + (ln & 15), p.M - 1);                          // rows >= M alias the last valid row
    areg[j] = *reinterpret_cast<const u32x4*>(p.A + pl * p.a_plane_stride + (((long)(row >> 4) * p.a_ktiles + ktb + kk) * 64 + ((ln & 48) | (row & 15))) * 8);
  }
  const bf16x8* Wl = reinterpret_cast<const bf16x8*>(p.W) + lane;
  bf16x8 b[KR][WS];
  long woff[WS];
#pragma unroll
  for (int j = 0; j < WS; ++j) woff[j] = ((long)min(s0 + j, p.nstrips - 1) * p.KT + ktb) * 64;   // clamped for the loads
#pragma unroll
  for (int kk = 0; kk < KR; ++kk)                 // in the order the MFMAs consume them
#pragma unroll
    for (int j = 0; j < WS; ++j) b[kk][j] = DIA_WLOAD(Wl + woff[j] + (long)kk * 64);
  if (tid < 256) {                              // row scales of the 32 rows: 8 threads per row, fixed order
    const int r = tid >> 3, part = tid & 7;
    float sA = 0.f;
    if (p.ssq_in != nullptr && r < p.M)
      for (int i = part; i < p.ssq_in_n; i += 8) sA += p.ssq_in[(long)i * p.ssq_ld + r];
    sA += __shfl_xor(sA, 1, 64);
    sA += __shfl_xor(sA, 2, 64);
    sA += __shfl_xor(sA, 4, 64);
    if (part == 0) inv_s[r] = (p.ssq_in != nullptr) ? rsqrtf(sA * p.inv_d + p.eps) : 1.0f;
  }
#pragma unroll
  for (int j = 0; j < NPIECE; ++j) abuf[tid + 512 * j] = areg[j];
  lds_barrier();                                // image range + row scales visible; the weight loads stay in flight

  f32x4 acc[MT][WS];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < WS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kk = 0; kk < KR; ++kk)
#pragma unroll
    for (int pl = 0; pl < DIA_NPLANES; ++pl)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, abuf[((kk * DIA_NPLANES + pl) * MT + i) * 64 + lane]);
#pragma unroll
        for (int j = 0; j < WS; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[kk][j], acc[i][j], 0, 0, 0);
      }

  // ---- one hand-off for the whole block: every lane's FPT partial sums, same mapping in every K range
  if (SK > 1) {
    // slab = [FPT / 2][512 threads][2 floats]: a wave's store instruction covers 512 contiguous bytes
    float* slab = p.sk_scratch + ((long)blockIdx.x * SK + ks) * (512 * FPT) + tid * 2;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < WS; ++j) {
        st2_agent(slab + ((i * WS + j) * 2) * 1024, acc[i][j][0], acc[i][j][1]);
        st2_agent(slab + ((i * WS + j) * 2 + 1) * 1024, acc[i][j][2], acc[i][j][3]);
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const int ticket = __hip_atomic_fetch_add(p.sk_tickets + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = ticket == SK - 1;
      if (last) __hip_atomic_store(p.sk_tickets + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      sk_flag = last;
    }
    __syncthreads();
    if (!sk_flag) return;
    const float* base = p.sk_scratch + (long)blockIdx.x * SK * (512 * FPT) + tid * 2;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < WS; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < SK; ++k)                                   // range order: deterministic
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < WS; ++j) {
          const float* s = base + (long)k * (512 * FPT) + ((i * WS + j) * 2) * 1024;
          const float2 lo = ld2_agent(s), hi = ld2_agent(s + 1024);
          acc[i][j][0] += lo.x; acc[i][j][1] += lo.y; acc[i][j][2] += hi.x; acc[i][j][3] += hi.y;
        }
  }
  // ---- epilogue per wave, two 16x16 tiles at a time through this wave's LDS tiles: the two strips of one m-tile
  // (WS = 2) or the two m-tiles of the one strip (WS = 1)
  float* tw = tiles + w * (2 * 16 * 17);
  const int et = lane >> 5, e_r = (lane >> 1) & 15, half = lane & 1;
  const int col = lane & 15, r0 = (lane >> 4) * 4;
#pragma unroll
  for (int it = 0; it < MT * WS / 2; ++it) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int i = WS == 2 ? it : t, j = WS == 2 ? t : 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) tw[t * (16 * 17) + (r0 + r) * 17 + col] = acc[i][j][r];
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int mtl = WS == 2 ? it : et;
    const int m = mtl * 16 + e_r;
    const int strip = s0 + (WS == 2 ? et : 0);
    const bool live = m < p.M && strip < p.nstrips;
    const int n0 = strip * 16 + half * 8;
    float xpre[8], gpre[8];
    if (p.epi == DIA_EPI_RESID_EMIT && live) {
      const float* o = p.out + (long)m * p.ldo + n0;
      const float4 xa = *reinterpret_cast<const float4*>(o), xb = *reinterpret_cast<const float4*>(o + 4);
      xpre[0] = xa.x; xpre[1] = xa.y; xpre[2] = xa.z; xpre[3] = xa.w;
      xpre[4] = xb.x; xpre[5] = xb.y; xpre[6] = xb.z; xpre[7] = xb.w;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) gpre[jj] = p.gnext ? p.gnext[n0 + jj] : 1.0f;
    }
    run_epilogue(p, tw + et * (16 * 17) + e_r * 17, inv_s[m], m, n0, half, min(strip, p.nstrips - 1), live, xpre, gpre);
    __builtin_amdgcn_wave_barrier();
  }
}

template <int KR, int WS>
int launch_blk32(const GemmK& k, hipStream_t st) {
  constexpr size_t smem = (size_t)KR * DIA_NPLANES * 2 * 64 * 16 + sizeof(float) * (8 * 2 * 16 * 17 + 32);
  launch_kernel(k_gemm_blk32<KR, WS>, dim3((k.nstrips + 8 * WS - 1) / (8 * WS), 1, k.KT / KR), dim3(512), smem, st, k);
  return dia_check_launch("k_gemm_blk32");
}

template <int KC, int PD, int WPE, int NWT>
int launch_tile_v(const GemmK& k, hipStream_t st) {
  const int mgroups = ((k.M + 15) / 16 + GT_MT - 1) / GT_MT;
  constexpr int SPB = (NWT / 2) * GT_WS;                 // strips per workgroup
  launch_kernel(k_gemm_tile<KC, PD, WPE, NWT>, dim3((k.nstrips + SPB - 1) / SPB, mgroups), dim3(NWT * 64), gt_smem(KC, NWT), st, k);
  return dia_check_launch("k_gemm_tile");
}

int launch_tile(const GemmK& k, hipStream_t st) {
  int rc = dia_kernels_init_once();
  if (rc) return rc;
  // default: the wave-specialised form (8 consumer + 4 producer waves): wi 113 us, wo 79, qkv 52, o 29 at 1696
  // rows, against 137 / 97 / 61 / 34 for the plain 8-wave form and 192 / 86 / 79 / 30 for 4-wave 64 x 128
  // blocks (kept behind DIA_DBG_TILE_V = 0 / 1 for comparison)
  int v = 3;
  if (const char* e = getenv("DIA_DBG_TILE_V")) v = atoi(e);
  if (v == 1) return launch_tile_v<2, 4, 1, 4>(k, st);
  if (v == 2) return launch_tile_v<2, 2, 1, 4>(k, st);
  if (v == 3) return launch_tile_ws<2, 2, 4>(k, st);
  if (v == 4) return launch_tile_ws<2, 2, 2>(k, st);
  if (v == 5) return launch_tile_ws<2, 4, 4>(k, st);
  return launch_tile_v<2, 4, 2, 8>(k, st);
}

template <int NW, int KPW>
int launch_g16(const GemmK& k, hipStream_t st) {
  const size_t smem = sizeof(f32x4) * NW * 64 + sizeof(float) * (16 * 17 + 16);
  const int sk = k.KT / (NW * KPW);
  int spw = k.spw > 0 ? k.spw : (k.nstrips >= 1024 ? 4 : 1);
  const int mz = k.mz > 1 ? k.mz : 1;
  // between one and four rounds of workgroups: walk the strips with about one workgroup per CU instead (logits, 579
  // strips at 16 rows: 16.1 -> 10.9 us with three strips per workgroup)
  if (mz == 1 && k.spw <= 0 && sk == 1 && k.nstrips > 256 && k.nstrips < 1024) spw = (k.nstrips + 255) / 256;
  // two m-tiles: about 256 workgroups in all (one per CU, both halves of every pair resident together)
  if (mz >= 2 && k.spw <= 0) {
    int per = 256 / mz / sk;
    per = per >= 8 ? per / 8 * 8 : (per > 0 ? per : 1);
    spw = (k.nstrips + per - 1) / per;
  }
  if (const char* e = getenv("DIA_DBG_SPW")) spw = atoi(e);
  if constexpr (!(NW == 16 && KPW >= 4)) {
    if (spw > 1) {      // persistent multi-strip form, with or without split-K: A fragments loaded once per workgroup
      int gx = (k.nstrips + spw - 1) / spw;
      if (mz >= 2 && (gx * sk) % 8 != 0 && (gx + 7) / 8 * 8 <= k.nstrips) gx = (gx + 7) / 8 * 8;   // pairs on one XCD
      if (mz > 1) launch_small_kernel(k_gemm16<NW, KPW, true, true>, dim3(gx, sk, mz), dim3(NW * 64), smem, st, k);
      else launch_small_kernel(k_gemm16<NW, KPW, true>, dim3(gx, sk), dim3(NW * 64), smem, st, k);
      return dia_check_launch("k_gemm16");
    }
  }
  if (mz > 1) launch_small_kernel(k_gemm16<NW, KPW, false, true>, dim3(k.nstrips, sk, mz), dim3(NW * 64), smem, st, k);
  else launch_small_kernel(k_gemm16<NW, KPW, false>, dim3(k.nstrips, sk), dim3(NW * 64), smem, st, k);
  return dia_check_launch("k_gemm16");
}

// returns true when a k_gemm16 instantiation exists for (nw, KT)
int launch_g16_any(const GemmK& k, int nw, int sk, hipStream_t st, bool& handled) {
  handled = true;
  const int kpw = (k.KT % (nw * sk) == 0) ? k.KT / (nw * sk) : 0;
  if (nw == 16) {
    if (kpw == 1) return launch_g16<16, 1>(k, st);
    if (kpw == 2) return launch_g16<16, 2>(k, st);
    if (kpw == 4) return launch_g16<16, 4>(k, st);
  } else if (nw == 8) {
    if (kpw == 1) return launch_g16<8, 1>(k, st);
    if (kpw == 2) return launch_g16<8, 2>(k, st);
    if (kpw == 3) return launch_g16<8, 3>(k, st);
    if (kpw == 4) return launch_g16<8, 4>(k, st);
    if (kpw == 5) return launch_g16<8, 5>(k, st);
    if (kpw == 6) return launch_g16<8, 6>(k, st);
    if (kpw == 7) return launch_g16<8, 7>(k, st);
    if (kpw == 8) return launch_g16<8, 8>(k, st);
  } else if (nw == 4) {
    if (kpw == 4) return launch_g16<4, 4>(k, st);
    if (kpw == 8) return launch_g16<4, 8>(k, st);
  }
  handled = false;
  return DIA_OK;
}

template <int MT, int NW, int KPW>
int launch(const GemmK& k, int mgroups, hipStream_t st) {
  size_t smem = sizeof(f32x4) * NW * MT * 64 + sizeof(float) * (MT * 16 * 17 + MT * 16);
  launch_kernel(k_gemm<MT, NW, KPW>, dim3(k.nstrips, mgroups), dim3(NW * 64), smem, st, k);
  return dia_check_launch("k_gemm");
}

template <int MT, int NW>
int launch_kpw(const GemmK& k, int mgroups, hipStream_t st) {
  if (k.KT % NW == 0) {
    switch (k.KT / NW) {
      case 1: return launch<MT, NW, 1>(k, mgroups, st);
      case 2: return launch<MT, NW, 2>(k, mgroups, st);
      case 4: return launch<MT, NW, 4>(k, mgroups, st);
      case 8: return launch<MT, NW, 8>(k, mgroups, st);
      case 16: if constexpr (NW <= 8) return launch<MT, NW, 16>(k, mgroups, st); else break;
      case 32: if constexpr (NW <= 8) return launch<MT, NW, 32>(k, mgroups, st); else break;
      default: break;
    }
  }
  return launch<MT, NW, 0>(k, mgroups, st);
}

template <int MT>
int launch_nw(const GemmK& k, int nw, int mgroups, hipStream_t st) {
  switch (nw) {
    case 4: return launch_kpw<MT, 4>(k, mgroups, st);
    case 8: return launch_kpw<MT, 8>(k, mgroups, st);
    case 16: return launch_kpw<MT, 16>(k, mgroups, st);
    default: return dia_fail(DIA_E_ARG, "dia_gemm: nw must be 4, 8 or 16");
  }
}

size_t small_smem(int nw, int KT, int rs) {     // KT = k-tiles one workgroup stages (its own K range)
  return sizeof(f32x4) * nw * 64 + sizeof(float) * (16 * 17 + 16) + (size_t)DIA_NPLANES * KT * 4 * rs * 16;
}

template <int NW, int KPW, int RS>
int launch_small(const GemmK& k, hipStream_t st) {
  size_t smem = small_smem(NW, NW * KPW, RS);
  if (const char* pad = getenv("DIA_DBG_LDS_PAD")) smem += (size_t)atoi(pad) * 1024;   // experiments: throttle residency
  if (smem > 64 * 1024) {
    int rc = dia_kernels_init_once();     // raises the dynamic-LDS limit of every large-LDS kernel, once
    if (rc) return rc;
  }
  // strips per workgroup: enough workgroups to cover every CU, few enough that each streams several
  // strips back to back (next strip's loads overlap this strip's reduce + epilogue)
  int spw = k.spw > 0 ? k.spw : (k.nstrips >= 1024 ? 4 : 1);
  if (const char* e = getenv("DIA_DBG_SPW")) spw = atoi(e);
  const int grid = (k.nstrips + spw - 1) / spw;
  const int sk = k.KT / (NW * KPW);          // cross-workgroup split-K factor (1 = none)
  if (sk > 1) {
    launch_small_kernel(k_gemv_small<NW, KPW, RS, false>, dim3(k.nstrips, sk), dim3(NW * 64), smem, st, k);
    return dia_check_launch("k_gemv_small");
  }
  if (spw > 1) {
    if constexpr (KPW <= 16 && !(NW == 16 && KPW > 4))
      launch_small_kernel(k_gemv_small<NW, KPW, RS, true>, dim3(grid), dim3(NW * 64), smem, st, k);
    else
      launch_small_kernel(k_gemv_small<NW, KPW, RS, false>, dim3(k.nstrips), dim3(NW * 64), smem, st, k);
  } else {
    launch_small_kernel(k_gemv_small<NW, KPW, RS, false>, dim3(k.nstrips), dim3(NW * 64), smem, st, k);
  }
  return dia_check_launch("k_gemv_small");
}

// row groups of 4 over gridDim.z (5..16 rows): 8 waves, K = 8 * KPW k-tiles, one strip per workgroup, no split-K
template <int KPW>
int launch_small_z(const GemmK& k, hipStream_t st) {
  const size_t smem = small_smem(8, 8 * KPW, 4);
  launch_small_kernel(k_gemv_small<8, KPW, 4, false, true>, dim3(k.nstrips, 1, k.mz), dim3(8 * 64), smem, st, k);
  return dia_check_launch("k_gemv_small(z)");
}

template <int RS>
int launch_small_rs(const GemmK& k, int nw, int sk, hipStream_t st, bool& handled) {
  handled = true;
  const int kpw = (k.KT % (nw * sk) == 0) ? k.KT / (nw * sk) : 0;
  if (nw == 4) {
    if (kpw == 4) return launch_small<4, 4, RS>(k, st);
    if (kpw == 8) return launch_small<4, 8, RS>(k, st);
    if (kpw == 16) return launch_small<4, 16, RS>(k, st);
  } else if (nw == 8) {
    if (kpw == 2) return launch_small<8, 2, RS>(k, st);
    if (kpw == 3) return launch_small<8, 3, RS>(k, st);      // 3, 5, 6, 7: K-compacted (pruned) shapes, K % 256 == 0
    if (kpw == 4) return launch_small<8, 4, RS>(k, st);
    if (kpw == 5) return launch_small<8, 5, RS>(k, st);
    if (kpw == 6) return launch_small<8, 6, RS>(k, st);
    if (kpw == 7) return launch_small<8, 7, RS>(k, st);
    if (kpw == 8) return launch_small<8, 8, RS>(k, st);
    if (kpw == 10) return launch_small<8, 10, RS>(k, st);    // 10, 12, 14: compacted hidden widths (multiples of 1024) under split-K 2
    if (kpw == 12) return launch_small<8, 12, RS>(k, st);
    if (kpw == 14) return launch_small<8, 14, RS>(k, st);
    if (kpw == 16) return launch_small<8, 16, RS>(k, st);
    if (kpw == 32) return launch_small<8, 32, RS>(k, st);
  } else if (nw == 16) {
    if (kpw == 1) return launch_small<16, 1, RS>(k, st);
    if (kpw == 2) return launch_small<16, 2, RS>(k, st);
    if (kpw == 4) return launch_small<16, 4, RS>(k, st);
    if (kpw == 8) return launch_small<16, 8, RS>(k, st);
  }
  handled = false;
  return DIA_OK;
}

template <int NW, int KPW>
int small_attr() {
  hipError_t e[4];
  e[0] = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_small<NW, KPW, 2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
  e[1] = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_small<NW, KPW, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
  e[2] = e[3] = hipSuccess;
  if constexpr (KPW <= 16 && !(NW == 16 && KPW > 4)) {
    e[2] = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_small<NW, KPW, 2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    e[3] = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_small<NW, KPW, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
  }
  for (int i = 0; i < 4; ++i) if (e[i] != hipSuccess) return dia_fail_hip(e[i], "hipFuncSetAttribute(k_gemv_small)");
  return DIA_OK;
}

}  // namespace

#ifdef DIA_DBG_STAMPS
extern "C" int dia_dbg_stamps(long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(long long) * n) == hipSuccess ? 0 : -2;
}
#endif

static int fill_gemmk(const dia_gemm_args* a, GemmK& k) {
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
  return DIA_OK;
}

// large-LDS attribute of every small-M instantiation, set once outside any graph capture
template <int KPW, int RS>
static int sparse_attr() {
  const size_t smem = sizeof(f32x4) * 16 * 64 + sizeof(float) * (16 * 17 + 16) + (size_t)DIA_NPLANES * (16 * KPW) * 4 * RS * 16 + (size_t)16 * KPW * 1024;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_sparse<KPW, RS, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess;
}

int dia_gemm_init() {
  int rc = 0;
  rc |= sparse_attr<4, 2>(); rc |= sparse_attr<4, 4>(); rc |= sparse_attr<2, 2>(); rc |= sparse_attr<2, 4>(); rc |= sparse_attr<1, 2>(); rc |= sparse_attr<1, 4>();
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_fused<4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlp_smem(64, 128)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_fused<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlp_smem(16, 16)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tile_ws<2, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gt_smem(2, 8)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tile_ws<2, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gt_smem(2, 8)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tile_ws<2, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gt_smem(2, 8)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tile<2, 4, 2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gt_smem(2, 8)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tile<2, 4, 1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gt_smem(2, 4)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tile<2, 2, 1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gt_smem(2, 4)) != hipSuccess) rc = 1;
  rc |= small_attr<4, 4>(); rc |= small_attr<4, 8>(); rc |= small_attr<4, 16>();
  rc |= small_attr<8, 2>(); rc |= small_attr<8, 3>(); rc |= small_attr<8, 4>(); rc |= small_attr<8, 5>(); rc |= small_attr<8, 6>(); rc |= small_attr<8, 7>(); rc |= small_attr<8, 8>(); rc |= small_attr<8, 10>(); rc |= small_attr<8, 12>(); rc |= small_attr<8, 14>(); rc |= small_attr<8, 16>(); rc |= small_attr<8, 32>();
  rc |= small_attr<16, 1>(); rc |= small_attr<16, 2>(); rc |= small_attr<16, 4>(); rc |= small_attr<16, 8>();
  return rc ? DIA_E_HIP : DIA_OK;
}

extern "C" int dia_gemm(const dia_gemm_args* a, void* stream) {
  if (!a || !a->A || (!a->W && !a->sp_blocks)) return dia_fail(DIA_E_ARG, "dia_gemm: null argument");
  if (a->M <= 0 || a->KT <= 0 || a->nstrips <= 0) return dia_fail(DIA_E_ARG, "dia_gemm: empty problem");
  if (a->KT > a->a_ktiles) return dia_fail(DIA_E_ARG, "dia_gemm: weight K exceeds the plane layout's K");
  if (a->a_plane_stride % 8 != 0 || a->p_plane_stride % 8 != 0) return dia_fail(DIA_E_ARG, "dia_gemm: plane stride must be a multiple of 8");
  if ((a->epi == DIA_EPI_SCALE_STORE || a->epi == DIA_EPI_RESID_EMIT) && (!a->out || (!a->strip_map && a->ldo < a->nstrips * 16) || a->ldo % 4 != 0))
    return dia_fail(DIA_E_ARG, "dia_gemm: output leading dimension too small");
  if ((a->epi == DIA_EPI_RESID_EMIT) && (!a->P || !a->ssq_out || a->p_ktiles * 32 < a->nstrips * 16))
    return dia_fail(DIA_E_ARG, "dia_gemm: RESID_EMIT needs planes and ssq_out covering N");
  if ((a->epi == DIA_EPI_SWIGLU_EMIT) && (!a->P || a->p_ktiles * 32 < a->nstrips * 8))
    return dia_fail(DIA_E_ARG, "dia_gemm: SWIGLU_EMIT needs planes covering N/2");
  if (a->epi == DIA_EPI_CROSSKV && (!a->kc || !a->vc || !a->cos_t || !a->sin_t || (!a->strip_map && a->nstrips != a->kv_heads * 16) || (!a->row_b && a->M > a->kv_cap) || (!a->row_b != !a->seg_off) || (a->kv_vblocked && a->kv_cap % 32 != 0)))
    return dia_fail(DIA_E_ARG, "dia_gemm: CROSSKV shape mismatch");
  if (a->epi < 0 || a->epi > DIA_EPI_CROSSKV) return dia_fail(DIA_E_ARG, "dia_gemm: unknown epilogue");
  if (a->ssq_in && a->ssq_ld < ((a->M + 15) / 16) * 16) return dia_fail(DIA_E_ARG, "dia_gemm: ssq_ld smaller than padded rows");

  GemmK k;
  fill_gemmk(a, k);

  if (a->sp_blocks || a->sp_toff) {       // zero-skipping stream: M <= 4, K = 16 * {1, 2, 4} k-tiles, no split-K
    if (!a->sp_blocks || !a->sp_toff || a->M > 4 || a->epi == DIA_EPI_CROSSKV || (a->epi == DIA_EPI_RESID_EMIT && !a->gnext) || a->sk > 1)
      return dia_fail(DIA_E_ARG, "dia_gemm: the sparse stream serves M <= 4 without split-K");
    const int rs = a->M <= 2 ? 2 : 4;
    hipStream_t st0 = (hipStream_t)stream;
    if (a->KT == 64) return rs == 2 ? launch_sparse<4, 2>(k, st0) : launch_sparse<4, 4>(k, st0);
    if (a->KT == 32) return rs == 2 ? launch_sparse<2, 2>(k, st0) : launch_sparse<2, 4>(k, st0);
    if (a->KT == 16) return rs == 2 ? launch_sparse<1, 2>(k, st0) : launch_sparse<1, 4>(k, st0);
    return dia_fail(DIA_E_ARG, "dia_gemm: no sparse kernel for this K");
  }
  int nw = a->nw;
  if (nw == 0) {
    // many strips -> few fat waves (deep load queues); few strips -> many waves per strip.
    // A 16-wave workgroup has 128 VGPRs per lane: it keeps at most 8 weight tiles (32 VGPRs) in flight
    // per wave; longer K ranges go to 8-wave workgroups (256 VGPRs).
    if (a->M <= 4 && a->nstrips >= 1024 && a->KT % 16 == 0 && a->KT / 16 <= 4) nw = 16;   // persistent multi-strip form
    else if (a->M <= 4 && a->KT % 8 == 0 && a->KT / 8 <= 8) nw = 8;          // measured: 8 waves x 8 k-tiles beats 16 x 4 and 4 x 16 (5.0 vs 5.2-5.8 us)
    else if (a->nstrips >= 512 && a->KT % 4 == 0 && a->KT / 4 <= 32) nw = 4;
    else if (a->KT % 16 == 0 && a->KT / 16 <= 8) nw = 16;
    else if (a->KT % 8 == 0) nw = 8;
    else nw = 4;
  }
  const int mtiles = (a->M + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  // cross-workgroup split-K (opt-in, a->sk > 1): sk workgroups per strip, each 1/sk of K, combined by the
  // last arriver.  The engine uses sk = 2 for wo at M <= 4 (11.2 vs 12.4 us); more splits or rows lose to the seam.
  int sk = 1;
  if (a->sk > 1) sk = a->sk;
  if (sk > 1 && (!a->sk_scratch || !a->sk_tickets || a->KT % sk != 0)) return dia_fail(DIA_E_ARG, "dia_gemm: split-K needs sk_scratch, sk_tickets and KT % sk == 0");
  if (sk > 1 && !a->nw) { const int ktl = a->KT / sk; nw = (ktl % 16 == 0 && ktl / 16 <= 4) ? 16 : ((ktl % 8 == 0) ? 8 : 4); }
  if (a->M <= 4 && a->epi != DIA_EPI_CROSSKV && !(a->epi == DIA_EPI_RESID_EMIT && !a->gnext)) {
    const int rs = a->M <= 2 ? 2 : 4;
    if (small_smem(nw, a->KT / sk, rs) <= 150 * 1024) {
      bool handled = false;
      int rc = (rs == 2) ? launch_small_rs<2>(k, nw, sk, st, handled) : launch_small_rs<4>(k, nw, sk, st, handled);
      if (handled) return rc;
    }
  }
  // 5..16 rows, K <= 2048, few strips: the 4-row kernel over row groups (k_gemv_small<..., MZ>), when every group of
  // every strip is resident at once (two workgroups per CU).  DIA_DBG_ZSMALL = highest strips x groups.  OFF by
  // default — measured slower in the step: o 9.2 -> 11.3 us at batch 8 (four groups), qkv 8.6 -> 11.2 at batch 3
  // (two groups): the image traffic of k_gemm16 is not what bounds these launches, the workgroup count is.
  if (mtiles == 1 && a->M > 4 && sk == 1 && !a->nw && a->epi != DIA_EPI_CROSSKV && !(a->epi == DIA_EPI_RESID_EMIT && !a->gnext) &&
      (a->KT == 64 || a->KT == 32 || a->KT == 16)) {
    int zmax = 0;
    if (const char* e = getenv("DIA_DBG_ZSMALL")) zmax = atoi(e);
    const int groups = (a->M + 3) / 4;
    if (a->nstrips * groups <= zmax) {
      k.mz = groups;
      const int rc = a->KT == 64 ? launch_small_z<8>(k, st) : (a->KT == 32 ? launch_small_z<4>(k, st) : launch_small_z<2>(k, st));
      k.mz = 0;
      return rc;
    }
  }
  if (mtiles == 1 && a->epi != DIA_EPI_CROSSKV && !(a->epi == DIA_EPI_RESID_EMIT && !a->gnext)) {
    // registers hold the A fragments: 12*KPW VGPRs, so only short per-wave K ranges qualify
    const int ktl16 = a->KT / sk;
    // 8 waves x up to 8 k-tiles each measured slightly ahead of 16 x 4 (6.3 vs 6.7 us on qkv at 16 rows)
    int nw16 = a->nw ? a->nw : ((ktl16 % 8 == 0 && ktl16 / 8 <= 8) ? 8 : ((ktl16 % 16 == 0 && ktl16 / 16 <= 4) ? 16 : 0));
    // the persistent multi-strip form double-buffers the weight tiles: 8 waves x 8 k-tiles fit, 16 x 4 spill
    if (!a->nw && a->nstrips >= 1024 && a->KT % 8 == 0 && a->KT / 8 <= 8) nw16 = 8;
    if (nw16) {
      bool handled = false;
      int rc = launch_g16_any(k, nw16, sk, st, handled);
      if (handled) return rc;
    }
  }
  // 17..128 rows: the one-m-tile kernel over all m-tiles at once (gridDim.z = 2..8, the workgroups of a group share their
  // weight stream through L2).  Split-K (a->sk > 1) needs scratch for every m-tile: mtiles * nstrips * sk * 256 floats,
  // mtiles * nstrips tickets.  DIA_DBG_PAIR16 = highest m-tile count served this way (0 = off).
  int pair_max = 8;
  if (const char* e = getenv("DIA_DBG_PAIR16")) pair_max = atoi(e) == 1 ? 8 : atoi(e);
  if (mtiles >= 2 && mtiles <= pair_max && a->epi != DIA_EPI_CROSSKV && !(a->epi == DIA_EPI_RESID_EMIT && !a->gnext) &&
      (sk == 1 || a->sk_scratch_floats >= (int64_t)mtiles * a->nstrips * sk * 256)) {
    const int ktl16 = a->KT / sk;
    int nw16 = a->nw ? a->nw : ((ktl16 % 8 == 0 && ktl16 / 8 <= 8) ? 8 : ((ktl16 % 16 == 0 && ktl16 / 16 <= 4) ? 16 : 0));
    if (nw16) {
      bool handled = false;
      k.mz = mtiles;
      int rc = launch_g16_any(k, nw16, sk, st, handled);
      k.mz = 0;
      if (handled) return rc;
    }
  }
  // 17..32 rows: column blocks x K ranges (k_gemm_blk32) when the caller lends split-K scratch holding
  // (column blocks) * (K ranges) * 512 * 8*WS floats and a ticket per column block.  Opt-in: DIA_DBG_BLK32 = "KR,WS"
  // (measured at 32 rows: wi 27.4 us, wo 20.1, o 10.5 against 20.9 / 21.3 / 6.9 for the paired kernel above).
  if (mtiles == 2 && a->epi != DIA_EPI_CROSSKV && a->sk <= 1 && a->sk_scratch && a->sk_tickets && getenv("DIA_DBG_BLK32")) {
    int kr = 16, ws = a->nstrips >= 512 ? 2 : 1;
    { const char* e = getenv("DIA_DBG_BLK32"); kr = atoi(e); const char* c = strchr(e, ','); if (c) ws = atoi(c + 1); }
    if ((kr == 8 || kr == 16) && (ws == 1 || ws == 2) && a->KT % kr == 0) {
      const int64_t need = (int64_t)((a->nstrips + 8 * ws - 1) / (8 * ws)) * (a->KT / kr) * 512 * 8 * ws;
      if (a->KT == kr || a->sk_scratch_floats >= need) {
        if (kr == 16) return ws == 2 ? launch_blk32<16, 2>(k, st) : launch_blk32<16, 1>(k, st);
        return ws == 2 ? launch_blk32<8, 2>(k, st) : launch_blk32<8, 1>(k, st);
      }
    }
  }
  // 17..32 rows: two m-tiles with register-resident A (8 waves x 8 k-tiles = K 2048 per workgroup); longer K is
  // split over KT / 64 workgroups per strip when the caller's scratch holds nstrips * sk * 512 floats
  if (mtiles == 2 && a->KT % 64 == 0 && a->epi != DIA_EPI_CROSSKV && !(a->epi == DIA_EPI_RESID_EMIT && !a->gnext) && a->sk <= 1 &&
      !getenv("DIA_DBG_NO_G32")) {
    const int sk32 = a->KT / 64;
    // only the long-K case pays (wo at batch 16: 36 -> 25 us): with K = 2048 every one-strip workgroup re-reads the
    // whole 393 KB activation image and loses to the generic kernel (wi 47 vs 38 us), so that case needs
    // DIA_DBG_G32_ALL to be selected
    if (sk32 == 1 && getenv("DIA_DBG_G32_ALL")) return launch_g32<8>(k, 1, st);
    // K = 2048 with many strips: the persistent two-half form (k_gemm32m) is NOT selected by default — measured at
    // batch 16: wi 36.4 us (generic 37.3), logits 25.8 (30.0), o 10.9 (12.4) but qkv 17.0 (12.6), cq 14.0 (11.9), and
    // the step as a whole slower (7 302 vs 7 597 frames/s): a split-K hand-off per strip inside the persistent loop
    // is a 3-4 us dependent chain that the next strip cannot hide.  DIA_DBG_G32M=1 selects it.
    if (sk32 == 1 && a->KT == 64 && a->nstrips >= 128 && a->sk_scratch && a->sk_tickets && a->sk_scratch_floats >= (int64_t)a->nstrips * 2 * 512 &&
        getenv("DIA_DBG_G32M"))
      return launch_g32m(k, 2, st);
    if (sk32 > 1 && a->sk_scratch && a->sk_tickets && a->sk_scratch_floats >= (int64_t)a->nstrips * sk32 * 512) return launch_g32<8>(k, sk32, st);
  }
  if (sk > 1) return dia_fail(DIA_E_ARG, "dia_gemm: no split-K kernel for this shape");
  // prefill shapes: the MFMA-tiled kernel (64 x 256 blocks) from 3 m-tiles on
  //   (only when the 64 x 256 blocks fill a good part of the chip: a lone short utterance is better off
  //   with the K-split kernels below)
  {
    const int blocks = ((mtiles + GT_MT - 1) / GT_MT) * ((a->nstrips + 15) / 16);
    int min_blocks = 48;
    if (const char* e = getenv("DIA_DBG_TILE_MIN")) min_blocks = atoi(e);
    if (mtiles >= 3 && a->KT % 8 == 0 && blocks >= min_blocks) return launch_tile(k, st);
  }
  if (mtiles == 1) return launch_nw<1>(k, nw, 1, st);
  if (mtiles == 2) return launch_nw<2>(k, nw, 1, st);
  return launch_nw<4>(k, nw, (mtiles + 3) / 4, st);
}

extern "C" int dia_gemm_timed(const dia_gemm_args* a, void* stream, float* ms_out) {
  if (!ms_out) return dia_fail(DIA_E_ARG, "dia_gemm_timed: null output");
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return dia_fail(DIA_E_HIP, "hipEventCreate");
  g_ev_start = e0; g_ev_stop = e1;
  int rc = dia_gemm(a, stream);
  g_ev_start = g_ev_stop = nullptr;
  if (rc == DIA_OK) {
    hipError_t he = hipEventSynchronize(e1);
    if (he == hipSuccess) he = hipEventElapsedTime(ms_out, e0, e1);
    if (he != hipSuccess) rc = dia_fail_hip(he, "dia_gemm_timed");
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return rc;
}

extern "C" int dia_mlp_fused(const dia_gemm_args* wi, const dia_gemm_args* wo, int32_t* barrier, void* stream) {
  if (!wi || !wo || !barrier) return dia_fail(DIA_E_ARG, "dia_mlp_fused: null argument");
  if (wi->M < 1 || wi->M > 2 || wo->M != wi->M) return dia_fail(DIA_E_ARG, "dia_mlp_fused: 1 or 2 rows only");
  if (wi->epi != DIA_EPI_SWIGLU_EMIT || wo->epi != DIA_EPI_RESID_EMIT || !wi->ssq_in || !wi->P || wo->A != wi->P || !wo->gnext ||
      !wo->out || !wo->P || !wo->ssq_out || !wo->sk_scratch || !wo->sk_tickets || wi->cmap || wi->strip_map)
    return dia_fail(DIA_E_ARG, "dia_mlp_fused: wi must be SWIGLU_EMIT into the planes wo reads, wo RESID_EMIT with split-K scratch");
  if (wi->nstrips * 8 != wo->KT * 32 || wo->KT % 32 != 0 || wi->KT % 16 != 0 || wi->p_ktiles < wo->KT || wo->a_ktiles != wi->p_ktiles ||
      wo->a_plane_stride != wi->p_plane_stride)
    return dia_fail(DIA_E_ARG, "dia_mlp_fused: shapes do not chain");
  const int G = 2 * wo->nstrips;
  static int n_cu = 0;
  if (!n_cu) { int dev = 0; if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = -1; }
  if (n_cu < G) return dia_fail(DIA_E_ARG, "dia_mlp_fused: the grid barrier needs every workgroup resident (2 * wo strips <= CUs)");
  if (wi->nstrips % G != 0 && wi->nstrips < G) return dia_fail(DIA_E_ARG, "dia_mlp_fused: too few wi strips");
  int rc = dia_kernels_init_once();
  if (rc) return rc;
  MlpK q;
  rc = fill_gemmk(wi, q.wi); if (rc) return rc;
  rc = fill_gemmk(wo, q.wo); if (rc) return rc;
  q.bar = barrier;
  const int kpw1 = wi->KT / 16, kpw2 = wo->KT / 32;
  hipStream_t st = (hipStream_t)stream;
  if (kpw1 == 4 && kpw2 == 8) { launch_kernel(k_mlp_fused<4, 8>, dim3(G), dim3(1024), mlp_smem(64, 128), st, q); return dia_check_launch("k_mlp_fused"); }
  if (kpw1 == 1 && kpw2 == 1) { launch_kernel(k_mlp_fused<1, 1>, dim3(G), dim3(1024), mlp_smem(16, 16), st, q); return dia_check_launch("k_mlp_fused"); }
  return dia_fail(DIA_E_ARG, "dia_mlp_fused: no instantiation for these K sizes");
}

extern "C" int dia_mlp_fused_timed(const dia_gemm_args* wi, const dia_gemm_args* wo, int32_t* barrier, void* stream, float* ms_out) {
  if (!ms_out) return dia_fail(DIA_E_ARG, "dia_mlp_fused_timed: null output");
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return dia_fail(DIA_E_HIP, "hipEventCreate");
  g_ev_start = e0; g_ev_stop = e1;
  int rc = dia_mlp_fused(wi, wo, barrier, stream);
  g_ev_start = g_ev_stop = nullptr;
  if (rc == DIA_OK) {
    hipError_t he = hipEventSynchronize(e1);
    if (he == hipSuccess) he = hipEventElapsedTime(ms_out, e0, e1);
    if (he != hipSuccess) rc = dia_fail_hip(he, "dia_mlp_fused_timed");
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return rc;
}
