// Persistent MLP segment of one decoder layer at M <= 4 rows (batch 1-2): cross o_proj -> wi_fused + SwiGLU -> wo
// (-> the NEXT layer's q/k/v projection) in ONE launch.  Replaces four launches of the decode step
// (reference DecoderLayer.forward, dia/layers.py:574-584: residual add after cross-attention, pre_mlp_norm, MlpBlock
// layers.py:95-104, residual add; and layers.py:541 + 273-275 of the following layer).
//
// Why: at batch 1 every launch of the chain pays ~3.3 us of fixed cost (boundary, first byte, reduce + epilogue) during
// which HBM idles; the four matrices of this segment are 121.6 MB = 19 us of streaming at the rate the chip sustains.
// Here the weight stream never stops at a dependency edge: every CU runs ONE workgroup of 7 streaming waves + 1 sync
// wave; a streaming wave keeps three 16 KiB slots of weights in flight in its REGISTERS (7 x 48 KiB per CU) and keeps
// requesting the slots of the following ops while the sync wave hands the activations over.
//
//   * weights ("ring layout", built at load: dia_hip/layout.py seg_ring): per CU one contiguous run of 16 KiB slots in the
//     order they are consumed.  A slot = ONE group of 4 output columns over K = 2048 = 16 tiles of 1 KiB; a tile is the B
//     operand of v_mfma_f32_16x16x32_bf16 holding FOUR k-tiles of those 4 columns: lane l, element j =
//     W[128 t + 32 ((l & 15) >> 2) + 8 (l >> 4) + j][n0 + (l & 3)].  The A operand carries the M <= 4 rows of the matching
//     k-tile in each group of 4 rows, so the 16 x 16 product is block diagonal: D[4g + r][4g + c] is k-tile g's
//     contribution to (row r, column c), the off-diagonal blocks are ignored.  Column granularity 4 lets all 256 CUs
//     take an equal share of every matrix (o: 8 columns per CU, wi: 32 gate + 32 up, wo: 32 columns x one K quarter,
//     qkv: 12) and every slot yields finished column sums: no cross-wave or cross-CU reduction except wo's K quarters.
//   * hand-offs (guide §6 Guideline 16, form R1; MI355X_MICROARCH.md visibility table, first row): the sync wave stores
//     its CU's outputs write-through (sc1), waits vmcnt(0), ONE lane adds 1 to a monotonic device-scope counter; the
//     consumer's sync wave polls that counter with sc1 loads (bounded), then pulls the vector with sc1 loads.  The sync
//     wave issues no weight load, so its vmcnt never waits behind the stream.  Counters are never reset: the expected
//     value is arrivals x (launch number + 1), the launch number lives in a per-CU word the kernel itself advances.
//   * no spin is unbounded: a poll that outlives `timeout_ticks` (100 MHz) sets the error word, and every later wait of
//     that workgroup falls through (the outputs are then garbage; the host checks the word and raises).
#include "common.hpp"
#include "../../include/dia_hip.h"
#include "errors.hpp"
#include "launch.hpp"
#include "tuning.hpp"

namespace {

constexpr int SG_CUS = 256;            // workgroups = CUs of an MI355X; all must be resident together
constexpr int SG_NCW = 7;              // streaming (compute) waves
constexpr int SG_NT = (SG_NCW + 1) * 64;
constexpr int SG_TILES = 16;           // 1 KiB tiles per slot
constexpr int SG_K = 2048;             // contraction length of one slot
constexpr int SG_D = 2048, SG_F = 8192, SG_NQ = 3072;
constexpr int SG_S_CO = SG_D / SG_CUS / 4;          // 2 slots
constexpr int SG_S_WI = 2 * SG_F / SG_CUS / 4;      // 16
constexpr int SG_S_WO = 8;                          // 32 columns of one K quarter
constexpr int SG_S_QKV = SG_NQ / SG_CUS / 4;        // 3
constexpr int SG_FIRST_WI = SG_S_CO, SG_FIRST_WO = SG_S_CO + SG_S_WI, SG_FIRST_QKV = SG_FIRST_WO + SG_S_WO;
constexpr int SG_SLOTS_FULL = SG_FIRST_QKV + SG_S_QKV;      // 29
constexpr int SG_KMAX = (SG_SLOTS_FULL + SG_NCW - 1) / SG_NCW;   // slots a streaming wave may own: 5
constexpr int SG_MAXM = 4;

// workspace (one allocation, caller-owned): [control words, zeroed once by the caller][payload]
constexpr int SG_CNT_STRIDE = 16;                   // one counter per 64-byte line
constexpr int SG_CNT_X1 = 0, SG_CNT_H = 8, SG_CNT_P = 16, SG_CNT_X2 = 80, SG_NCNT = 88;
constexpr size_t SG_OFF_LAUNCH = (size_t)SG_NCNT * SG_CNT_STRIDE * 4;            // uint32 [256]
constexpr size_t SG_OFF_ERR = SG_OFF_LAUNCH + SG_CUS * 4;                        // uint32 [16]
constexpr size_t SG_CTRL_BYTES = (SG_OFF_ERR + 64 + 255) / 256 * 256;
constexpr size_t SG_OFF_V1 = SG_CTRL_BYTES;                                      // float [4][2048]
constexpr size_t SG_OFF_X1 = SG_OFF_V1 + SG_MAXM * SG_D * 4;                     // float [4][2048]
constexpr size_t SG_OFF_SSQ1 = SG_OFF_X1 + SG_MAXM * SG_D * 4;                   // float [4][256]
constexpr size_t SG_OFF_H = SG_OFF_SSQ1 + SG_MAXM * SG_CUS * 4;                  // float [4 quarters][4][2048]
constexpr size_t SG_OFF_P = SG_OFF_H + 4 * SG_MAXM * SG_K * 4;                   // float [64][3][4][32]
constexpr size_t SG_OFF_V2 = SG_OFF_P + 64 * 3 * SG_MAXM * 32 * 4;               // float [4][2048]
constexpr size_t SG_OFF_SSQ2 = SG_OFF_V2 + SG_MAXM * SG_D * 4;                   // float [4][64]
constexpr size_t SG_WS_BYTES = SG_OFF_SSQ2 + SG_MAXM * 64 * 4;

// LDS
constexpr size_t SG_L_PLANES = 0;                                                // bf16x8 [3][16][64]   48 KiB
constexpr size_t SG_L_RAW = SG_L_PLANES + 3 * SG_TILES * 64 * 16;                // float [4][2048]      32 KiB
constexpr size_t SG_L_PART = SG_L_RAW + SG_MAXM * SG_K * 4;                      // f32x4 [16 slots][4 g][4 c]   4 KiB
constexpr size_t SG_L_MISC = SG_L_PART + 16 * 16 * 16;                           // float inv[4]; int abort
constexpr size_t SG_LDS_BYTES = SG_L_MISC + 64;

struct SegK {
  const float* a_in; int a_ktiles;                  // attention output, fp32 activation tiles (common.hpp), rows in m-tile 0
  const bf16_raw* W; int nslots; int M; int has_qkv;
  float* x; int ldx;                                // residual stream
  const float* g_mlp; const float* g_next;
  float* qkv_out; int ldq;
  float* planes_x; int xkt; float* ssq; int ssq_ld; // standard-format outputs of wo (RESID_EMIT's): fp32 tiles of x*g_next, strip ssq
  float inv_d, eps;
  unsigned char* ws;
  long long timeout_ticks;
  int dbg;                                          // debug bits: 1 = the streaming waves load nothing (timing of the hand-offs alone; wrong results)
  int head_sleep;                                   // s_sleep units (64 clocks) the waves of the later ops wait before their first loads
  long long* stamps;                                // debug: [256][16] wall-clock stamps of the sync wave (NULL = off)
};

__device__ __forceinline__ long long sg_clock() { return (long long)__builtin_amdgcn_s_memrealtime(); }

__device__ __forceinline__ void sg_bar() { lds_barrier(); }

__device__ __forceinline__ int sg_op_of(int s) { return s < SG_FIRST_WI ? 0 : (s < SG_FIRST_WO ? 1 : (s < SG_FIRST_QKV ? 2 : 3)); }
__device__ __forceinline__ int sg_op_first(int op) { return op == 0 ? 0 : (op == 1 ? SG_FIRST_WI : (op == 2 ? SG_FIRST_WO : SG_FIRST_QKV)); }

// all threads: raw fp32 rows in LDS -> the three bf16 planes in the block-diagonal fragment order
__device__ __forceinline__ void sg_convert(unsigned char* smem, int tid, int M) {
  const float* raw = reinterpret_cast<const float*>(smem + SG_L_RAW);
  bf16x8* planes = reinterpret_cast<bf16x8*>(smem + SG_L_PLANES);
#pragma unroll
  for (int u = 0; u < (SG_TILES * 64) / SG_NT; ++u) {
    const int i = tid + u * SG_NT;
    const int t = i >> 6, l = i & 63;
    const int r = l & 3, g = (l & 15) >> 2, q = l >> 4;
    if (r < M) {
      const float4* src = reinterpret_cast<const float4*>(raw + r * SG_K + 128 * t + 32 * g + 8 * q);
      bf16x8 h, mi, lo;
      split3x8(src[0], src[1], h, mi, lo);
      planes[i] = h; planes[SG_TILES * 64 + i] = mi; planes[2 * SG_TILES * 64 + i] = lo;
    }
  }
}

__device__ __forceinline__ void sg_load_slot(bf16x8 (&b)[SG_TILES], const bf16_raw* W, int cu, int nslots, int s, int lane, int dbg = 0) {
  if (dbg & 1) return;
  const bf16x8* src = reinterpret_cast<const bf16x8*>(W) + ((long)(cu * nslots + s) * SG_TILES) * 64 + lane;
#pragma unroll
  for (int t = 0; t < SG_TILES; ++t) b[t] = __builtin_nontemporal_load(src + t * 64);
}

__device__ __forceinline__ void sg_consume(const bf16x8 (&b)[SG_TILES], unsigned char* smem, int slot_in_op, int lane) {
  const bf16x8* planes = reinterpret_cast<const bf16x8*>(smem + SG_L_PLANES);
  f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
#pragma unroll
  for (int t = 0; t < SG_TILES; ++t) {
    const bf16x8 h = planes[t * 64 + lane], mi = planes[SG_TILES * 64 + t * 64 + lane], lo = planes[2 * SG_TILES * 64 + t * 64 + lane];
    a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(h, b[t], a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(mi, b[t], a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo, b[t], a2, 0, 0, 0);
  }
  f32x4 a;
#pragma unroll
  for (int r = 0; r < 4; ++r) a[r] = (a0[r] + a1[r]) + a2[r];
  // diagonal block g lives in the lanes with l >> 4 == (l & 15) >> 2: rows 4g..4g+3 in the registers, column l & 3
  const int g = lane >> 4;
  if (((lane & 15) >> 2) == g) reinterpret_cast<f32x4*>(smem + SG_L_PART)[(slot_in_op * 4 + g) * 4 + (lane & 3)] = a;
}

// sum of the four k-tile blocks of (slot, column c), row r — fixed order
__device__ __forceinline__ float sg_part(const unsigned char* smem, int slot_in_op, int c, int r) {
  const float* p = reinterpret_cast<const float*>(smem + SG_L_PART) + (slot_in_op * 16 + c) * 4 + r;
  return ((p[0] + p[16]) + p[32]) + p[48];
}

struct SgSync {
  unsigned char* ws; unsigned* err; int* abort_s; long long timeout; int lane;
  __device__ __forceinline__ unsigned* cnt(int i) const { return reinterpret_cast<unsigned*>(ws) + i * SG_CNT_STRIDE; }
  // lanes [0, n) poll counters first .. first+n-1 until each equals `want` (bounded)
  __device__ __forceinline__ void wait(int first, int n, unsigned want, unsigned code) const {
    if (*abort_s) return;
    const long long t0 = sg_clock();
    const unsigned* p = cnt(first + (lane < n ? lane : 0));
    for (;;) {
      const unsigned v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__all(lane >= n || v == want)) break;
      __builtin_amdgcn_s_sleep(2);
      if (sg_clock() - t0 > timeout) {
        if (lane == 0) { __hip_atomic_store(err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *abort_s = 1; }
        break;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");     // (no instruction: the payload loads stay behind the poll)
  }
  // every store of this wave has been acknowledged, then ONE lane signals
  __device__ __forceinline__ void arrive(int c) const {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(cnt(c), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};

__device__ __forceinline__ void st1_agent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld1_agent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// sync wave: M rows x 2048 floats (row stride `ld` floats, device-coherent) -> raw LDS rows
__device__ __forceinline__ void sg_gather_rows(const float* src, int ld, int M, unsigned char* smem, int lane) {
  const __amdgpu_buffer_rsrc_t r = agent_rsrc(src);
  float4* raw = reinterpret_cast<float4*>(smem + SG_L_RAW);
  f32x4 v[SG_MAXM * 8];
#pragma unroll
  for (int u = 0; u < SG_MAXM * 8; ++u) {
    const int row = min(u >> 3, M - 1), c = (u & 7) * 64 + lane;     // float4 index inside the row (rows >= M: a clamped
    v[u] = ld4_agent(r, (row * ld + c * 4) * 4);                     //  address, loaded unconditionally, never stored)
  }
#pragma unroll
  for (int u = 0; u < SG_MAXM * 8; ++u) {
    const int row = u >> 3, c = (u & 7) * 64 + lane;
    if (row < M) raw[row * (SG_K / 4) + c] = float4{v[u][0], v[u][1], v[u][2], v[u][3]};
  }
}

template <int SG_NB>
__global__ __launch_bounds__(SG_NT) void k_seg_mlp(SegK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, cu = blockIdx.x;
  const int M = p.M, nslots = p.nslots;
  const int nops = p.has_qkv ? 4 : 3;
  float* inv_s = reinterpret_cast<float*>(smem + SG_L_MISC);
  int* abort_s = reinterpret_cast<int*>(smem + SG_L_MISC + 32);

  if (w < SG_NCW) {
    // ------------------------------------------------------------------ streaming waves
    bf16x8 buf[SG_NB][SG_TILES];
    // the sync wave's first loads (16 KiB of attention output + the residual row) must be in the CU's memory queue AHEAD of
    // the weight stream — that queue is served in order, and 336 KiB of weights in front of them cost 11 us — and so must
    // the slots of the first op: a short sleep instead of a barrier (arrays kept in registers across a barrier went to scratch)
    for (int i = 0; i < (w >= SG_S_CO ? 2 * p.head_sleep : p.head_sleep); ++i) __builtin_amdgcn_s_sleep(1);
#pragma unroll
    for (int k = 0; k < SG_NB; ++k) {
      const int s = w + SG_NCW * k;
      if (s < nslots) sg_load_slot(buf[k], p.W, cu, nslots, s, lane, p.dbg);
    }
    {   // rows >= M of every group of 4 stay zero for the whole launch
      bf16x8* planes = reinterpret_cast<bf16x8*>(smem + SG_L_PLANES);
      const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      for (int i = tid; i < SG_TILES * 64; i += SG_NCW * 64)
        if ((i & 3) >= M) { planes[i] = z; planes[SG_TILES * 64 + i] = z; planes[2 * SG_TILES * 64 + i] = z; }
    }
    int cur = -1;                     // op whose planes are ready (its barrier B has been passed)
    auto advance_to = [&](int op) {
      while (cur < op) {
        if (cur >= 0) sg_bar();       // C(cur): partial sums of op `cur` are in LDS
        sg_bar();                     // A(cur+1): raw rows of the next op are in LDS
        sg_convert(smem, tid, M);
        sg_bar();                     // B(cur+1): planes ready
        ++cur;
      }
    };
#pragma unroll
    for (int k = 0; k < SG_KMAX; ++k) {
      const int s = w + SG_NCW * k;
      if (s < nslots) {
        const int op = sg_op_of(s);
        advance_to(op);
        sg_consume(buf[k % SG_NB], smem, s - sg_op_first(op), lane);
        const int s2 = s + SG_NCW * SG_NB;
        if (s2 < nslots) sg_load_slot(buf[k % SG_NB], p.W, cu, nslots, s2, lane, p.dbg);
      }
    }
    advance_to(nops - 1);
    sg_bar();                         // C(last)
    return;
  }

  // -------------------------------------------------------------------- sync wave
  SgSync sy{p.ws, reinterpret_cast<unsigned*>(p.ws + SG_OFF_ERR), abort_s, p.timeout_ticks, lane};
  if (lane == 0) *abort_s = 0;
  unsigned* launch_no = reinterpret_cast<unsigned*>(p.ws + SG_OFF_LAUNCH);
  const unsigned L1 = launch_no[cu] + 1u;                 // this launch's number + 1
  float* v1 = reinterpret_cast<float*>(p.ws + SG_OFF_V1);
  float* x1b = reinterpret_cast<float*>(p.ws + SG_OFF_X1);
  float* ssq1 = reinterpret_cast<float*>(p.ws + SG_OFF_SSQ1);
  float* hb = reinterpret_cast<float*>(p.ws + SG_OFF_H);
  float* pb = reinterpret_cast<float*>(p.ws + SG_OFF_P);
  float* v2 = reinterpret_cast<float*>(p.ws + SG_OFF_V2);
  float* ssq2 = reinterpret_cast<float*>(p.ws + SG_OFF_SSQ2);

#define SG_STAMP(i) do { if (p.stamps && lane == 0) p.stamps[cu * 16 + (i)] = sg_clock(); } while (0)
  SG_STAMP(0);
  // operands of the first epilogue, requested before anything else: lane = (row r, column c8 of this CU's 8)
  const int r8 = lane >> 3, c8 = lane & 7;
  const bool live8 = r8 < M;
  const int n8 = 8 * cu + c8;
  const float xold = p.x[(long)(live8 ? r8 : 0) * p.ldx + n8];
  const float gm = p.g_mlp[n8];
  {   // attention output of the previous kernel (plain loads: a kernel boundary lies in between): fragment f = row * 256 + kt * 4 + q
    float4* raw = reinterpret_cast<float4*>(smem + SG_L_RAW);
    float4 fa[SG_MAXM * 4], fb[SG_MAXM * 4];
#pragma unroll
    for (int u = 0; u < SG_MAXM * 4; ++u) {
      const int f = u * 64 + lane, row = min(f >> 8, M - 1), idx = f & 255, kt = idx >> 2, q = idx & 3;
      const float4* src = reinterpret_cast<const float4*>(p.a_in + ((long)kt * 64 + row + 16 * q) * 8);
      fa[u] = src[0]; fb[u] = src[1];
    }
#pragma unroll
    for (int u = 0; u < SG_MAXM * 4; ++u) {
      const int f = u * 64 + lane, row = f >> 8, idx = f & 255;
      if (row < M) { raw[row * (SG_K / 4) + idx * 2] = fa[u]; raw[row * (SG_K / 4) + idx * 2 + 1] = fb[u]; }
    }
  }
  SG_STAMP(1);
  sg_bar();                           // A(0)
  sg_convert(smem, tid, M);
  sg_bar();                           // B(0)
  SG_STAMP(2);
  sg_bar();                           // C(0)
  SG_STAMP(3);

  // ---- cross o_proj epilogue (layers.py:574): x1 = x + acc; publish x1, x1 * g_mlp and the CU's sum of squares
  {
    float x1 = 0.f;
    if (live8) {
      x1 = xold + sg_part(smem, c8 >> 2, c8 & 3, r8);
      st1_agent(x1b + r8 * SG_D + n8, x1);
      st1_agent(v1 + r8 * SG_D + n8, mul_rn(x1, gm));
    }
    float sq = mul_rn(x1, x1);
    sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
    if (live8 && c8 == 0) st1_agent(ssq1 + r8 * SG_CUS + cu, sq);
    sy.arrive(SG_CNT_X1 + (cu & 7));
  }
  SG_STAMP(4);
  sy.wait(SG_CNT_X1, 8, 32u * L1, 1u);
  SG_STAMP(5);
  float x1own = 0.f;                  // wo reducers (CUs 0..63): x1 of their 32 columns, element e = lane (+64)
  float x1own2 = 0.f;
  {
    sg_gather_rows(v1, SG_D, M, smem, lane);
    // row scale of pre_mlp_norm: 256 partials per row, lane i holds [4i, 4i+4)
    const __amdgpu_buffer_rsrc_t sr = agent_rsrc(ssq1);
#pragma unroll
    for (int r = 0; r < SG_MAXM; ++r) {
      if (r < M) {
        const f32x4 q = ld4_agent(sr, (r * SG_CUS + lane * 4) * 4);
        const float s = wave_sum((q[0] + q[1]) + (q[2] + q[3]));
        if (lane == 0) inv_s[r] = rsqrtf(s * p.inv_d + p.eps);
      }
    }
    if (cu < 64) {
      const int e0 = lane, e1 = lane + 64;
      if (e0 < M * 32) x1own = ld1_agent(x1b + (e0 >> 5) * SG_D + 32 * cu + (e0 & 31));
      if (e1 < M * 32) x1own2 = ld1_agent(x1b + (e1 >> 5) * SG_D + 32 * cu + (e1 & 31));
    }
  }
  SG_STAMP(6);
  sg_bar();                           // A(1)
  sg_convert(smem, tid, M);
  sg_bar();                           // B(1)
  sg_bar();                           // C(1)
  SG_STAMP(7);

  // ---- SwiGLU (layers.py:95-101): hidden units [32 cu, 32 cu + 32), gate slots 0..7, up slots 8..15
  {
    const int quarter = cu >> 6;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = lane + 64 * u, r = e >> 5, hcol = e & 31;
      if (e < M * 32) {
        const float inv = inv_s[r];
        const float g = sg_part(smem, hcol >> 2, hcol & 3, r) * inv, up = sg_part(smem, 8 + (hcol >> 2), hcol & 3, r) * inv;
        st1_agent(hb + ((long)quarter * SG_MAXM + r) * SG_K + 32 * (cu & 63) + hcol, (g / (1.0f + expf(-g))) * up);
      }
    }
    sy.arrive(SG_CNT_H + quarter);
    SG_STAMP(8);
    sy.wait(SG_CNT_H + quarter, 1, 64u * L1, 2u);
    SG_STAMP(9);
    sg_gather_rows(hb + (long)quarter * SG_MAXM * SG_K, SG_K, M, smem, lane);
  }
  sg_bar();                           // A(2)
  sg_convert(smem, tid, M);
  sg_bar();                           // B(2)
  sg_bar();                           // C(2)
  SG_STAMP(10);

  // ---- wo: this CU holds K quarter (cu >> 6) of columns [32 c, 32 c + 32), c = cu & 63; CU c sums the quarters in order
  {
    const int quarter = cu >> 6, c = cu & 63;
    float acc[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int e = lane + 64 * u, r = e >> 5, col = e & 31;
      acc[u] = (e < M * 32) ? sg_part(smem, col >> 2, col & 3, r) : 0.f;
    }
    if (quarter != 0) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = lane + 64 * u;
        if (e < M * 32) st1_agent(pb + ((long)(c * 3 + quarter - 1) * SG_MAXM + (e >> 5)) * 32 + (e & 31), acc[u]);
      }
      sy.arrive(SG_CNT_P + c);
    } else {
      sy.wait(SG_CNT_P + c, 1, 3u * L1, 3u);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int e = lane + 64 * u, r = e >> 5, col = e & 31;
        const bool live = e < M * 32;
        float x2 = 0.f;
        if (live) {
          const float p1 = ld1_agent(pb + ((long)(c * 3 + 0) * SG_MAXM + r) * 32 + col);
          const float p2 = ld1_agent(pb + ((long)(c * 3 + 1) * SG_MAXM + r) * 32 + col);
          const float p3 = ld1_agent(pb + ((long)(c * 3 + 2) * SG_MAXM + r) * 32 + col);
          const int n = 32 * c + col;
          x2 = (u == 0 ? x1own : x1own2) + (((acc[u] + p1) + p2) + p3);
          p.x[(long)r * p.ldx + n] = x2;                                   // layers.py:582
          const float vg = mul_rn(x2, p.g_next[n]);
          p.planes_x[plane_frag_off(r, n & ~7, p.xkt) + (n & 7)] = vg;     // what a launched consumer (the logits head) reads
          if (p.has_qkv) st1_agent(v2 + r * SG_D + n, vg);
        }
        const float s16 = row16_sum(mul_rn(x2, x2));                       // this strip's sum of squares (16 columns)
        const float s32 = s16 + __shfl_xor(s16, 16, 64);
        if (live && (col & 15) == 0) p.ssq[(long)(2 * c + (col >> 4)) * p.ssq_ld + r] = s16;
        if (live && col == 0 && p.has_qkv) st1_agent(ssq2 + r * 64 + c, s32);
      }
      sy.arrive(SG_CNT_X2 + (c & 7));      // (also without a q/k/v stage: every counter advances once per launch)
    }
  }
  SG_STAMP(11);
  if (p.has_qkv) {
    sy.wait(SG_CNT_X2, 8, 8u * L1, 4u);
    SG_STAMP(12);
    sg_gather_rows(v2, SG_D, M, smem, lane);
#pragma unroll
    for (int r = 0; r < SG_MAXM; ++r) {
      if (r < M) {
        const float s = wave_sum(ld1_agent(ssq2 + r * 64 + lane));
        if (lane == 0) inv_s[r] = rsqrtf(s * p.inv_d + p.eps);
      }
    }
    SG_STAMP(13);
    sg_bar();                         // A(3)
    sg_convert(smem, tid, M);
    sg_bar();                         // B(3)
    sg_bar();                         // C(3)
    SG_STAMP(14);
    // ---- next layer's q/k/v projection of the pre-SA-normed row (layers.py:541, 273-275): columns [12 cu, 12 cu + 12)
    const int r = lane / 12, cc = lane % 12;
    if (r < M) p.qkv_out[(long)r * p.ldq + 12 * cu + cc] = sg_part(smem, cc >> 2, cc & 3, r) * inv_s[r];
  }
  if (lane == 0) launch_no[cu] = L1;
  SG_STAMP(15);
}

}  // namespace

static bool g_seg_ready = false;
static int seg_init() {
  if (g_seg_ready) return DIA_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_seg_mlp<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SG_LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_seg_mlp<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SG_LDS_BYTES);
  if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_seg_mlp<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SG_LDS_BYTES);
  if (e != hipSuccess) return dia_fail_hip(e, "hipFuncSetAttribute(k_seg_mlp)");
  g_seg_ready = true;
  return DIA_OK;
}

extern "C" int64_t dia_seg_workspace_bytes(void) { return (int64_t)SG_WS_BYTES; }
extern "C" int64_t dia_seg_workspace_control_bytes(void) { return (int64_t)SG_CTRL_BYTES; }
extern "C" int32_t dia_seg_slots(int with_qkv) { return with_qkv ? SG_SLOTS_FULL : SG_FIRST_QKV; }

extern "C" int dia_seg_supported(int D, int F, int attn_width, int nqkv) {
  if (D != SG_D || F != SG_F || attn_width != SG_K || nqkv != SG_NQ) return 0;
  int dev = 0;
  hipDeviceProp_t pr;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return 0;
  return pr.multiProcessorCount >= SG_CUS ? 1 : 0;
}

extern "C" int dia_seg_error(const void* ws, void* stream) {
  if (!ws) return dia_fail(DIA_E_ARG, "dia_seg_error: null workspace");
  unsigned code = 0;
  hipError_t e = hipMemcpyAsync(&code, (const unsigned char*)ws + SG_OFF_ERR, 4, hipMemcpyDeviceToHost, (hipStream_t)stream);
  if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
  if (e != hipSuccess) return dia_fail_hip(e, "dia_seg_error");
  return (int)code;
}

extern "C" int dia_seg_mlp(const dia_seg_args* a, void* stream) {
  if (!a || !a->a_in || !a->W || !a->x || !a->g_mlp || !a->g_next || !a->planes_x || !a->ssq || !a->ws)
    return dia_fail(DIA_E_ARG, "dia_seg_mlp: null argument");
  if (a->M < 1 || a->M > SG_MAXM) return dia_fail(DIA_E_ARG, "dia_seg_mlp: 1..4 rows");
  if (a->D != SG_D || a->F != SG_F || a->a_ktiles * 32 != SG_K) return dia_fail(DIA_E_ARG, "dia_seg_mlp: built for D 2048, F 8192, attention width 2048");
  if (a->has_qkv && (!a->qkv_out || a->ldq < SG_NQ)) return dia_fail(DIA_E_ARG, "dia_seg_mlp: q/k/v output missing");
  if (a->nslots != (a->has_qkv ? SG_SLOTS_FULL : SG_FIRST_QKV)) return dia_fail(DIA_E_ARG, "dia_seg_mlp: slot count does not match the ring layout");
  if (a->xkt * 32 < SG_D || a->ssq_ld < a->M || a->ldx < SG_D) return dia_fail(DIA_E_ARG, "dia_seg_mlp: output layout too narrow");
  int rc = seg_init();
  if (rc) return rc;
  SegK k;
  k.a_in = a->a_in; k.a_ktiles = a->a_ktiles; k.W = (const bf16_raw*)a->W; k.nslots = a->nslots; k.M = a->M; k.has_qkv = a->has_qkv ? 1 : 0;
  k.x = a->x; k.ldx = a->ldx; k.g_mlp = a->g_mlp; k.g_next = a->g_next; k.qkv_out = a->qkv_out; k.ldq = a->ldq;
  k.planes_x = a->planes_x; k.xkt = a->xkt; k.ssq = a->ssq; k.ssq_ld = a->ssq_ld; k.inv_d = 1.0f / SG_D; k.eps = a->eps;
  k.ws = (unsigned char*)a->ws;
  k.stamps = (long long*)a->stamps;
  k.timeout_ticks = a->timeout_us > 0 ? (long long)a->timeout_us * 100 : 2000000LL;      // default 20 ms
  k.dbg = dia_tune(DIA_TUNE_SEG_DBG) > 0 ? dia_tune(DIA_TUNE_SEG_DBG) : 0;
  k.head_sleep = dia_tune(DIA_TUNE_SEG_SLEEP) >= 0 ? dia_tune(DIA_TUNE_SEG_SLEEP) : 8;
  const int nb = dia_tune(DIA_TUNE_SEG_NB) > 0 ? dia_tune(DIA_TUNE_SEG_NB) : 3;      // 16 KiB slots in flight per streaming wave
  if (nb == 1) dia_launch<k_seg_mlp<1>>(dim3(SG_CUS), dim3(SG_NT), SG_LDS_BYTES, (hipStream_t)stream, k);
  else if (nb == 2) dia_launch<k_seg_mlp<2>>(dim3(SG_CUS), dim3(SG_NT), SG_LDS_BYTES, (hipStream_t)stream, k);
  else dia_launch<k_seg_mlp<3>>(dim3(SG_CUS), dim3(SG_NT), SG_LDS_BYTES, (hipStream_t)stream, k);
  return dia_check_launch("k_seg_mlp");
}
