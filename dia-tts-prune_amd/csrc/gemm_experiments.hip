// Measured-and-rejected GEMM forms, kept as tested kernel-level experiments (DESIGN.md §5 "measured and rejected").
// NOT part of the default build: `make EXPERIMENTS=1` compiles this unit and defines DIA_EXPERIMENTS, which lets
// dia_gemm reach these kernels through the tuning knobs named below; without it the entry points defined here
// (dia_mlp_fused*, the sparse stream of dia_gemm_args.sp_blocks) fail with DIA_E_ARG.
#include "gemm_common.hpp"
#include "gemm_experiments.hpp"

namespace {

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
  launch_kernel<k_gemv_sparse<KPW, RS, MAXS>>(dim3(grid), dim3(1024), smem, st, k);
  return dia_check_launch("k_gemv_sparse");
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
  launch_kernel<k_gemm32<KPW>>(dim3(k.nstrips, sk), dim3(512), smem, st, k);
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
  launch_kernel<k_gemm32m>(dim3(gx, sk), dim3(512), smem, st, k);
  return dia_check_launch("k_gemm32m");
}

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
  launch_kernel<k_gemm_blk32<KR, WS>>(dim3((k.nstrips + 8 * WS - 1) / (8 * WS), 1, k.KT / KR), dim3(512), smem, st, k);
  return dia_check_launch("k_gemm_blk32");
}

template <int KC, int PD, int WPE, int NWT>
int launch_tile_v(const GemmK& k, hipStream_t st) {
  const int mgroups = ((k.M + 15) / 16 + GT_MT - 1) / GT_MT;
  constexpr int SPB = (NWT / 2) * GT_WS;                 // strips per workgroup
  launch_kernel<k_gemm_tile<KC, PD, WPE, NWT>>(dim3((k.nstrips + SPB - 1) / SPB, mgroups), dim3(NWT * 64), gt_smem(KC, NWT), st, k);
  return dia_check_launch("k_gemm_tile");
}

}  // namespace

// ---------------------------------------------------------------------------------------------------
// host glue (called by dia_gemm / dia_gemm_init in gemm.hip when built with DIA_EXPERIMENTS)
// ---------------------------------------------------------------------------------------------------
template <int KPW, int RS>
static int sparse_attr() {
  const size_t smem = sizeof(f32x4) * 16 * 64 + sizeof(float) * (16 * 17 + 16) + (size_t)DIA_NPLANES * (16 * KPW) * 4 * RS * 16 + (size_t)16 * KPW * 1024;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_sparse<KPW, RS, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess;
}

int dia_exp_init() {
  int rc = 0;
  rc |= sparse_attr<4, 2>(); rc |= sparse_attr<4, 4>(); rc |= sparse_attr<2, 2>(); rc |= sparse_attr<2, 4>(); rc |= sparse_attr<1, 2>(); rc |= sparse_attr<1, 4>();
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_fused<4, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlp_smem(64, 128)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mlp_fused<1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mlp_smem(16, 16)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tile<2, 4, 2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gt_smem(2, 8)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tile<2, 4, 1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gt_smem(2, 4)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tile<2, 2, 1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gt_smem(2, 4)) != hipSuccess) rc = 1;
  return rc;
}

// zero-skipping stream: M <= 4, K = 16 * {1, 2, 4} k-tiles, no split-K
int dia_exp_gemm_sparse(const dia_gemm_args* a, void* stream) {
  if (!a->sp_blocks || !a->sp_toff || a->M > 4 || a->epi == DIA_EPI_CROSSKV || (a->epi == DIA_EPI_RESID_EMIT && !a->gnext) || a->sk > 1)
    return dia_fail(DIA_E_ARG, "dia_gemm: the sparse stream serves M <= 4 without split-K");
  GemmK k;
  fill_gemmk(a, k);
  const int rs = a->M <= 2 ? 2 : 4;
  hipStream_t st0 = (hipStream_t)stream;
  if (a->KT == 64) return rs == 2 ? launch_sparse<4, 2>(k, st0) : launch_sparse<4, 4>(k, st0);
  if (a->KT == 32) return rs == 2 ? launch_sparse<2, 2>(k, st0) : launch_sparse<2, 4>(k, st0);
  if (a->KT == 16) return rs == 2 ? launch_sparse<1, 2>(k, st0) : launch_sparse<1, 4>(k, st0);
  return dia_fail(DIA_E_ARG, "dia_gemm: no sparse kernel for this K");
}

// 17..32 rows (two m-tiles): the forms that preceded k_gemm16 over gridDim.z.  All opt-in through tuning knobs.
int dia_exp_gemm_two_mtiles(const dia_gemm_args* a, void* stream, bool& handled) {
  handled = true;
  GemmK k;
  fill_gemmk(a, k);
  hipStream_t st = (hipStream_t)stream;
  // column blocks x K ranges (k_gemm_blk32) when the caller lends split-K scratch holding (column blocks) * (K ranges) *
  // 512 * 8*WS floats and a ticket per column block: blk32_kr = 8 | 16, blk32_ws = 1 | 2
  // (measured at 32 rows: wi 27.4 us, wo 20.1, o 10.5 against 20.9 / 21.3 / 6.9 for the z-form)
  if (a->epi != DIA_EPI_CROSSKV && a->sk <= 1 && a->sk_scratch && a->sk_tickets && dia_tune(DIA_TUNE_BLK32_KR) > 0) {
    const int kr = dia_tune(DIA_TUNE_BLK32_KR);
    const int ws = dia_tune(DIA_TUNE_BLK32_WS) > 0 ? dia_tune(DIA_TUNE_BLK32_WS) : (a->nstrips >= 512 ? 2 : 1);
    if ((kr == 8 || kr == 16) && (ws == 1 || ws == 2) && a->KT % kr == 0) {
      const int64_t need = (int64_t)((a->nstrips + 8 * ws - 1) / (8 * ws)) * (a->KT / kr) * 512 * 8 * ws;
      if (a->KT == kr || a->sk_scratch_floats >= need) {
        if (kr == 16) return ws == 2 ? launch_blk32<16, 2>(k, st) : launch_blk32<16, 1>(k, st);
        return ws == 2 ? launch_blk32<8, 2>(k, st) : launch_blk32<8, 1>(k, st);
      }
    }
  }
  // two m-tiles with register-resident A (8 waves x 8 k-tiles = K 2048 per workgroup); longer K is split over KT / 64
  // workgroups per strip when the caller's scratch holds nstrips * sk * 512 floats
  if (a->KT % 64 == 0 && a->epi != DIA_EPI_CROSSKV && !(a->epi == DIA_EPI_RESID_EMIT && !a->gnext) && a->sk <= 1 && dia_tune(DIA_TUNE_NO_G32) <= 0) {
    const int sk32 = a->KT / 64;
    // only the long-K case paid (wo at batch 16: 36 -> 25 us): with K = 2048 every one-strip workgroup re-reads the
    // whole 393 KB activation image and loses to the generic kernel (wi 47 vs 38 us): g32_all selects it anyway
    if (sk32 == 1 && dia_tune(DIA_TUNE_G32_ALL) > 0) return launch_g32<8>(k, 1, st);
    // K = 2048 with many strips: the persistent two-half form (k_gemm32m, knob g32m) — measured at batch 16: wi 36.4 us
    // (generic 37.3), logits 25.8 (30.0), o 10.9 (12.4) but qkv 17.0 (12.6), cq 14.0 (11.9), and the step as a whole
    // slower (7 302 vs 7 597 frames/s): a split-K hand-off per strip inside the persistent loop is a 3-4 us dependent
    // chain that the next strip cannot hide
    if (sk32 == 1 && a->KT == 64 && a->nstrips >= 128 && a->sk_scratch && a->sk_tickets && a->sk_scratch_floats >= (int64_t)a->nstrips * 2 * 512 &&
        dia_tune(DIA_TUNE_G32M) > 0)
      return launch_g32m(k, 2, st);
    if (sk32 > 1 && a->sk_scratch && a->sk_tickets && a->sk_scratch_floats >= (int64_t)a->nstrips * sk32 * 512)
      return launch_g32<8>(k, sk32, st);
  }
  handled = false;
  return DIA_OK;
}

// prefill tile kernel without wave specialisation: 0 = 8 waves 64 x 256, 1 / 2 = 4 waves 64 x 128 (PD 4 / 2)
int dia_exp_tile_variant(const dia_gemm_args* a, void* stream, int v) {
  int rc = dia_kernels_init_once();
  if (rc) return rc;
  GemmK k;
  fill_gemmk(a, k);
  hipStream_t st = (hipStream_t)stream;
  if (v == 1) return launch_tile_v<2, 4, 1, 4>(k, st);
  if (v == 2) return launch_tile_v<2, 2, 1, 4>(k, st);
  return launch_tile_v<2, 4, 2, 8>(k, st);
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
  if (kpw1 == 4 && kpw2 == 8) { launch_kernel<k_mlp_fused<4, 8>>(dim3(G), dim3(1024), mlp_smem(64, 128), st, q); return dia_check_launch("k_mlp_fused"); }
  if (kpw1 == 1 && kpw2 == 1) { launch_kernel<k_mlp_fused<1, 1>>(dim3(G), dim3(1024), mlp_smem(16, 16), st, q); return dia_check_launch("k_mlp_fused"); }
  return dia_fail(DIA_E_ARG, "dia_mlp_fused: no instantiation for these K sizes");
}

extern "C" int dia_mlp_fused_timed(const dia_gemm_args* wi, const dia_gemm_args* wo, int32_t* barrier, void* stream, float* ms_out) {
  if (!ms_out) return dia_fail(DIA_E_ARG, "dia_mlp_fused_timed: null output");
  dia_recorder_arm();
  int rc = dia_mlp_fused(wi, wo, barrier, stream);
  float ms[2] = {0.f, 0.f};
  const int n = dia_recorder_collect(ms, 2);
  if (rc != DIA_OK) return rc;
  if (n < 1) return n < 0 ? n : dia_fail(DIA_E_STATE, "dia_mlp_fused_timed: nothing was launched");
  *ms_out = ms[0];
  return DIA_OK;
}
