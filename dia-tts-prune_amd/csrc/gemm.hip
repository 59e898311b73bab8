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
#include "gemm_common.hpp"
#include "gemm_experiments.hpp"

namespace {

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

  // fp32 weights as three bf16 planes (w_planes == 3: hi, mid, lo tile sets back to back): the K loop below once per plane,
  // into the same accumulators — 9 exact bf16 products per fp32 x fp32 one
  if (p.w_planes > 1) {
    prefetch_epilogue<MT, NW * 64>(p, tid, mt0, m, n0, live, xpre, gpre, inv_s);
    const int kt1 = min(kt0 + kpw, p.KT);
    for (int pw = 0; pw < p.w_planes; ++pw) {
      const bf16x8* Wp = Wt + pw * (p.w_plane_stride / 8);
      for (int kt = kt0; kt < kt1; ++kt) {
        const bf16x8 bb = DIA_WLOAD(Wp + (long)kt * 64);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          bf16x8 a3[DIA_NPLANES];
          load_afrag3(p, aoff[mt] + (long)kt * 512, a3[0], a3[1], a3[2]);
#pragma unroll
          for (int pl = 0; pl < DIA_NPLANES; ++pl)
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[pl], bb, acc[mt], 0, 0, 0);
        }
      }
    }
  } else if constexpr (KPW > 0) {
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
        load_afrag3(p, aoff[mt] + (long)(kt0 + i) * 512, a0[i][mt][0], a0[i][mt][1], a0[i][mt][2]);
    prefetch_epilogue<MT, NW * 64>(p, tid, mt0, m, n0, live, xpre, gpre, inv_s);
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        bf16x8 a3[DIA_NPLANES];
        if (i < AP) { a3[0] = a0[i < AP ? i : 0][mt][0]; a3[1] = a0[i < AP ? i : 0][mt][1]; a3[2] = a0[i < AP ? i : 0][mt][2]; }
        else load_afrag3(p, aoff[mt] + (long)(kt0 + i) * 512, a3[0], a3[1], a3[2]);
#pragma unroll
        for (int pl = 0; pl < DIA_NPLANES; ++pl)
          acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[pl], b[i], acc[mt], 0, 0, 0);
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
            bf16x8 a3[DIA_NPLANES];
            load_afrag3(p, aoff[mt] + (long)(kt + i) * 512, a3[0], a3[1], a3[2]);
#pragma unroll
            for (int pl = 0; pl < DIA_NPLANES; ++pl)
              acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a3[pl], b[i], acc[mt], 0, 0, 0);
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

// The first ten arguments repeat fields of p: they fill the first 64 bytes of the argument block, which the command
// processor hands over in SGPRs at wave launch (kernarg preload, -mllvm -amdgpu-kernarg-preload-count=16) — the operand
// and weight loads of the prologue then need no scalar load from the argument block, whose lines every CU of the grid
// otherwise requests at the same moment (in-kernel stamps: 0.8 us from the start of a wave to its first weight load).
template <int NW, int KPW, int RS, bool MULTI, bool AF32 = false, bool PF32 = false>
__global__ __launch_bounds__(NW * 64) void k_gemv_small(const bf16_raw* a_A, long a_aps, const bf16_raw* a_W, int a_KT, int a_M, int a_epi,
                                                        int a_nstrips, float* a_out, int a_ldo, const float* a_gnext, GemmK p) {
  p.A = a_A; p.a_plane_stride = a_aps; p.W = a_W; p.KT = a_KT; p.M = a_M; p.epi = a_epi; p.nstrips = a_nstrips;
  p.out = a_out; p.ldo = a_ldo; p.gnext = a_gnext;
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
#ifdef DIA_X_WFIRST
  load_strip(b0, blockIdx.x);                       // the HBM stream starts here
  __builtin_amdgcn_sched_barrier(0);
#endif

  // ---- every small, L2-resident operand is requested BEFORE the weight stream, branch-free: vmcnt
  // retires in order, so anything queued behind 16-32 KiB of HBM loads per wave would stall its first
  // use (and with it the barrier below) until the whole strip has arrived.
  // (1) compact A image: chunk c = ((plane*KT + kt)*4 + kq)*RS + row, 16 bytes each
  constexpr int CH = (3 * KPW * RS + 15) / 16;        // chunks per thread = 3*KT*4*RS / NT
  constexpr int nchunks = DIA_NPLANES * KT * 4 * RS;
  // AF32: the image arrives as fp32 tiles (common.hpp) — 32 bytes per (k-tile, quarter, row) entry instead of three 16-byte plane
  // chunks, a third less to pull before the barrier; the thread that loads an entry splits it into the three plane chunks
  constexpr int nentries = KT * 4 * RS;
  constexpr int CE = (KPW * RS + 15) / 16;           // entries per thread = KT*4*RS / NT
  bf16x8 v0[AF32 ? 1 : CH];
  float4 ex[AF32 ? CE : 1], ey[AF32 ? CE : 1];
  if constexpr (AF32) {
    const float* Af = reinterpret_cast<const float*>(p.A);
#pragma unroll
    for (int u = 0; u < CE; ++u) {
      const int c = min(tid + u * NT, nentries - 1);
      const int row = c % RS, kq = (c / RS) & 3, kt = c / (4 * RS);
      const float4* src = reinterpret_cast<const float4*>(Af + ((long)(ktg + kt) * 64 + row + 16 * kq) * 8);
      ex[u] = src[0]; ey[u] = src[1];
    }
  } else {
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const int c = min(tid + u * NT, nchunks - 1);
      const int row = c % RS, kq = (c / RS) & 3, kt = (c / (4 * RS)) % KT, pl = c / (4 * RS * KT);
      v0[u] = *reinterpret_cast<const bf16x8*>(p.A + pl * p.a_plane_stride + ((long)(ktg + kt) * 64 + row + 16 * kq) * 8);
    }
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
  const bool r_thread = tid < 16 * RS;                // one tile element per thread in the epilogue (run_epilogue_rows)
  float xpre1 = 0.f, gpre1 = 1.f;
  auto load_resid = [&](int strip) {
    const int r_m = tid >> 4, n = strip * 16 + (tid & 15);
    xpre1 = p.out[(long)(r_m < p.M ? r_m : 0) * p.ldo + n];
    gpre1 = p.gnext[n];
  };
  if (resid && r_thread) load_resid(blockIdx.x);
  __builtin_amdgcn_sched_barrier(0);
#ifndef DIA_X_WFIRST
  load_strip(b0, blockIdx.x);                       // the HBM stream starts here
  __builtin_amdgcn_sched_barrier(0);
#endif
  STAMP(1);
  if constexpr (AF32) {
#pragma unroll
    for (int u = 0; u < CE; ++u)
      if (tid + u * NT < nentries) {
        const int c = tid + u * NT;                    // entry c of plane pl sits at chunk pl * nentries + c
        bf16x8 h, mi, lo;
        split3x8(ex[u], ey[u], h, mi, lo);
        As[c] = h; As[nentries + c] = mi; As[2 * nentries + c] = lo;
      }
  } else {
#pragma unroll
    for (int u = 0; u < CH; ++u)
      if (tid + u * NT < nchunks) As[tid + u * NT] = v0[u];
  }
  {
    float s0 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s0 += (s_part + 8 * i < p.ssq_in_n && s_row < p.M) ? sq[i] : 0.f;
    if (s_thread && s_row < p.M)
      for (int idx = s_part + 128; idx < p.ssq_in_n; idx += 8) s0 += p.ssq_in[(long)idx * p.ssq_ld + s_row];   // D > 2048 only
    // the 8 partials of a row sit in 8 consecutive lanes: quad xor 1, quad xor 2, then the other quad of the half row
    // (after two steps a quad is uniform, so the mirror delivers what lane ^ 4 holds) — DPP, no LDS crossbar
    s0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, true));
    s0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0x4E, 0xF, 0xF, true));
    s0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0x141, 0xF, 0xF, true));
    if (tid < 128 && s_part == 0) inv_s[s_row] = has_norm ? rsqrtf(s0 * p.inv_d + p.eps) : 1.0f;
  }
  lds_barrier();      // A image + row scales visible; the weight loads stay in flight
  STAMP(2);

  const int arow = min(lane & 15, RS - 1), akq = lane >> 4;
  // cross-wave sum of the partial tiles.  Only rows 0..3 of the 16x16 tile can be valid (M <= 4): they live in lanes 0..15
  // (registers = rows), so a wave hands over 256 bytes, and thread (row m, column c) of the epilogue adds the NW
  // partials of ITS element in wave order — one barrier, no tile round trip (the old form wrote all 64 lanes, summed on
  // one wave, wrote a tile and met at a second barrier).  Double-buffered across the strips of the persistent form.
  f32x4* red16 = red;                                 // [2][NW][16]
  int sbuf = 0;
  auto body = [&](bf16x8* bc, bf16x8* bn, int strip) {
    const int next = strip + G;
    // next strip's weights stream while this one computes.  UNCONDITIONAL (past the end: the last strip once more, an L2 hit that
    // nobody reads): behind an `if` the wait before the first MFMA has to serve the path without the new loads too, the
    // compiler then emits vmcnt(KPW - 1 - i) instead of vmcnt(2 KPW - 1 - i) and every MFMA waits for the load just issued
    if constexpr (MULTI) load_strip(bn, DIA_PREFETCH_CLAMP(next, p.nstrips));
#ifdef DIA_PIN_PREFETCH
    if constexpr (MULTI) __builtin_amdgcn_sched_barrier(0);     // keep the requests IN FRONT of the MFMAs they are meant to run under
#endif
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
    float v = 0.f;
    if (MULTI || gridDim.y == 1) {
      f32x4* rb = red16 + sbuf * (NW * 16);
      sbuf ^= 1;
      if (lane < 16) rb[w * 16 + lane] = acc[0];
      lds_barrier();
      if (r_thread) {
        const float* rf = reinterpret_cast<const float*>(rb) + (tid & 15) * 4 + (tid >> 4);
        v = rf[0];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) v += rf[ww * 64];
      }
      STAMP(4);
    } else {              // cross-workgroup split-K (wo): partial tile -> slab, the last arriver sums the slabs in split order
      reduce_to_tile<1, NW, true>(acc, red, tile, tid, lane, w);
      STAMP(4);
      if (!splitk_combine(p, tile, strip, tid, &sk_flag)) return;
      if (r_thread) v = tile[(tid >> 4) * 17 + (tid & 15)];
    }
    if (r_thread) {
      run_epilogue_rows<RS, PF32>(p, v, inv_s, tid, strip, xpre1, gpre1);
      if (MULTI && next < p.nstrips && resid) load_resid(next);      // residual operands of the next strip
    }
  };
  if constexpr (MULTI) {
    // strip PAIRS, then at most one more: with `if (second strip exists) body(b1, b0, ...)` inside the loop the compiler put an
    // s_waitcnt vmcnt(0) at the loop header (the two ways into it leave different loads pending), i.e. the strip requested
    // by the second body had to land before the next trip could start — found in the ISA, round 3
    int strip = blockIdx.x;
    for (; strip + G < p.nstrips; strip += 2 * G) {
      body(b0, b1, strip);
      body(b1, b0, strip + G);
    }
    if (strip < p.nstrips) body(b0, b1, strip);
  } else {
    body(b0, b1, blockIdx.x);
  }
  STAMP(5);
}


// 5..16 rows (batch 3-8): one m-tile, A fragments held in registers for the workgroup's whole life
// (each wave owns a fixed K range of KPW k-tiles = 12*KPW VGPRs) and reused for every strip the
// workgroup walks; weight tiles double-buffered across strips like k_gemv_small.
#ifndef DIA_Z_TEMPORAL
#define DIA_Z_TEMPORAL 1
#endif
constexpr bool ZTEMPORAL = DIA_Z_TEMPORAL != 0;
#ifndef DIA_ZR_RING3
#define DIA_ZR_RING3 0
#endif
constexpr size_t g16_smem(int nw) { return sizeof(f32x4) * 2 * nw * 64 + 1536 + sizeof(float) * (2 * 16 * 17 + 16); }
constexpr size_t g16_alds(int nw, int kpw) { return (size_t)2 * nw * kpw * 64 * 16; }     // mid + lo planes of every wave's A fragments
// ALDS (17..128 rows, the z-form): only the hi plane of a wave's A fragments stays in registers; the mid and lo planes live in a
// wave-private part of LDS (2 x 8 KiB per wave) and are re-read per strip (16 ds_read_b128 under the 24 MFMAs).  The 64 VGPRs
// this frees let the z-form run the element-per-thread tail of the 16-row kernels (256 threads sum, scale and emit one tile
// element each) instead of round 1's 32-thread tail, whose serial reduce + epilogue on ONE wave was what a strip cost
// (1.7 us per strip at 128 rows against 0.32 us of MFMA time: in-kernel; a deeper weight ring alone changed nothing).
template <int NW, int KPW, bool MULTI, bool MZ = false, bool AF32 = false, bool PF32 = false, bool PAIR = false, bool ALDS = false>
__global__ __launch_bounds__(NW * 64) void k_gemm16(const bf16_raw* a_A, long a_aps, const bf16_raw* a_W, int a_KT, int a_M, int a_epi,
                                                    int a_nstrips, float* a_out, int a_ldo, const float* a_gnext, GemmK p) {
  // (leading arguments = fields of p, preloaded into SGPRs: see k_gemv_small)
  p.A = a_A; p.a_plane_stride = a_aps; p.W = a_W; p.KT = a_KT; p.M = a_M; p.epi = a_epi; p.nstrips = a_nstrips;
  p.out = a_out; p.ldo = a_ldo; p.gnext = a_gnext;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                         // [2][NW][64]
  bf16_raw* stg = reinterpret_cast<bf16_raw*>(smem_raw + sizeof(f32x4) * 2 * NW * 64);   // [3][16][16] plane staging
  float* tile = reinterpret_cast<float*>(smem_raw + sizeof(f32x4) * 2 * NW * 64 + 1536);   // [16][17]
  float* inv_s = tile + 16 * 17;                                           // [16]
  constexpr int NT = NW * 64;

  // 17..128 rows (batch 9-64): gridDim.z = 2..8, one m-tile per z.  The workgroups of a group stream the same weights
  // at the same time from different CUs of ONE XCD (gridDim.x * gridDim.y is a multiple of 8), so HBM sees each
  // byte once and the second reader is served by that XCD's L2; everything row-indexed is shifted by 16 rows.
  // (a separate instantiation: the 16-row kernels stay exactly as they were — the extra prologue cost 2 % of the
  // batch-8 step)
  int zrow0 = 0;                  // first row of this workgroup's m-tile (CROSSKV addresses rows of the packed batch globally)
  if constexpr (MZ) {
    const int z = blockIdx.z;
    zrow0 = 16 * z;
    p.A += (long)z * p.a_ktiles * 512 * (AF32 ? 2 : 1);       // (bf16_raw pointer: an fp32 tile set is twice as wide)
    p.M = min(16, p.M - 16 * z);
    if (p.ssq_in) p.ssq_in += 16 * z;
    if (p.out) p.out += (long)16 * z * p.ldo;
    if (p.P) p.P += (long)z * p.p_ktiles * 512 * (PF32 ? 2 : 1);
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
  STAMP(0);

  // epilogue geometry: thread t < 256 owns element (row t / 16, column t % 16) of the finished 16 x 16 tile
  const int r16 = tid >> 4, c16 = tid & 15;
  const bool r_thread = tid < 256;
  float xpre1 = 0.f, gpre1 = 1.f;

  // A fragments (rows >= M alias the last valid row: no extra L2 traffic, results never stored)
  const int alane = (lane & 48) | min(lane & 15, p.M - 1);
  bf16x8 a[KPW][DIA_NPLANES];
  constexpr bool ALDS_LATE = ALDS && !MZ;      // see below
  if constexpr (AF32) {     // fp32 tiles (compile-time: a second operand path behind a branch would end the basic block in which
    // all loads of the wave are issued).  32 bytes per fragment; the raw values land in the registers of planes 0 and 1 of their
    // own fragment and are split in place — no second register set beside the 12 * KPW VGPRs of `a`
    const float4* Af = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(p.A) + ((long)kt0 * 64 + alane) * 8);
    // (masking the lanes of missing rows off instead of aliasing them to the last row was measured: slower at every row count, -3.5 % at 16 rows where
    // nothing is masked — the branch costs more than the aliased lanes do)
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
#ifdef DIA_X_NOA                                             /* TIMING ONLY: one k-tile of the image eight times (wrong results) */
      a[i][0] = __builtin_bit_cast(bf16x8, Af[0]);
      a[i][1] = __builtin_bit_cast(bf16x8, Af[1]);
#else
      a[i][0] = __builtin_bit_cast(bf16x8, Af[(long)i * 128]);
      a[i][1] = __builtin_bit_cast(bf16x8, Af[(long)i * 128 + 1]);
#endif
    }
    if constexpr (!ALDS_LATE) {
#pragma unroll
      for (int i = 0; i < KPW; ++i)
        split3x8(__builtin_bit_cast(float4, a[i][0]), __builtin_bit_cast(float4, a[i][1]), a[i][0], a[i][1], a[i][2]);
    }
  } else {
#pragma unroll
    for (int i = 0; i < KPW; ++i)
#pragma unroll
      for (int pl = 0; pl < DIA_NPLANES; ++pl)
        a[i][pl] = *reinterpret_cast<const bf16x8*>(p.A + pl * p.a_plane_stride + ((long)(kt0 + i) * 64 + alane) * 8);
  }
  bf16x8* my = reinterpret_cast<bf16x8*>(smem_raw + g16_smem(NW)) + (long)w * 2 * KPW * 64 + lane;     // ALDS: [plane - 1][i][lane]
  // ALDS_LATE (the 16-row persistent form): split and LDS stores wait for the image, so they sit BEHIND the first weight request
  // (batch 8 +0.9 %).  Not in the z-form: there the same move costs qkv / cq 0.9 us at 32 rows (profiles/r03_early_wait_ab.txt)
  if constexpr (ALDS && !ALDS_LATE) {
#pragma unroll
    for (int i = 0; i < KPW; ++i) { my[i * 64] = a[i][1]; my[(KPW + i) * 64] = a[i][2]; }
  }
  // plane pl of the wave's i-th A fragment
  auto afrag = [&](int i, int pl) -> bf16x8 {
    if constexpr (ALDS) { if (pl > 0) return my[((pl - 1) * KPW + i) * 64]; }
    return a[i][pl];
  };
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
    const int n = strip * 16 + c16;
    xpre1 = p.out[(long)(r16 < p.M ? r16 : 0) * p.ldo + n];
    gpre1 = p.gnext[n];
  };
  // 17..128 rows (MZ): the tail of round 1 — 32 threads, 8 tile elements each through the shared epilogue.  The element-per-thread
  // tail below is worth 4 % at 16 rows, but it takes the persistent form from 231 VGPRs to the 256 limit with 200 B of scratch
  // per lane, and at 2..8 m-tiles EVERY launch runs that form: batch 16 8 618 -> 7 100 frames/s, batch 32 12 836 -> 9 300.
  const int e_r = (tid >> 1) & 15, e_half = tid & 1;
  const bool e_thread = tid < 32, e_live = e_thread && e_r < p.M;
  float xpre8[8], gpre8[8];
  auto load_resid8 = [&](int strip) {
    const int n0 = strip * 16 + e_half * 8;
    const float* o = p.out + (long)(e_live ? e_r : 0) * p.ldo + n0;
    const float4 xa = *reinterpret_cast<const float4*>(o), xb = *reinterpret_cast<const float4*>(o + 4);
    xpre8[0] = xa.x; xpre8[1] = xa.y; xpre8[2] = xa.z; xpre8[3] = xa.w;
    xpre8[4] = xb.x; xpre8[5] = xb.y; xpre8[6] = xb.z; xpre8[7] = xb.w;
    const float4 ga = *reinterpret_cast<const float4*>(p.gnext + n0), gb = *reinterpret_cast<const float4*>(p.gnext + n0 + 4);
    gpre8[0] = ga.x; gpre8[1] = ga.y; gpre8[2] = ga.z; gpre8[3] = ga.w;
    gpre8[4] = gb.x; gpre8[5] = gb.y; gpre8[6] = gb.z; gpre8[7] = gb.w;
  };
  // (the persistent 8 x 8 form with fp32 input takes that tail as well — unless it runs the strip-pair split-K below)
  // PAIR (host-selected instantiation): split-K with exactly two strips per workgroup, gridDim.y > 1 && 2 * gridDim.x == nstrips
  constexpr bool ZTAIL8 = MULTI && !MZ && !PAIR && KPW == 8 && AF32 && !ALDS;
  constexpr bool pair = PAIR;
  constexpr bool ZT = (MZ && !ALDS) || ZTAIL8;       // the 32-thread tail
  if constexpr (ZT) { if (resid && e_thread) load_resid8(blockIdx.x); }
  else { if (resid && r_thread) load_resid(blockIdx.x); }
  __builtin_amdgcn_sched_barrier(0);
  load_strip(b0, blockIdx.x);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (ALDS_LATE) {
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
      if constexpr (AF32) split3x8(__builtin_bit_cast(float4, a[i][0]), __builtin_bit_cast(float4, a[i][1]), a[i][0], a[i][1], a[i][2]);
      my[i * 64] = a[i][1]; my[(KPW + i) * 64] = a[i][2];
    }
  }
  STAMP(1);
  {
    float s0 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s0 += (s_part + 8 * i < p.ssq_in_n && s_row < p.M) ? sq[i] : 0.f;
    if (s_thread && s_row < p.M)
      for (int idx = s_part + 128; idx < p.ssq_in_n; idx += 8) s0 += p.ssq_in[(long)idx * p.ssq_ld + s_row];
    s0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, true));    // see k_gemv_small
    s0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0x4E, 0xF, 0xF, true));
    s0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0x141, 0xF, 0xF, true));
    if (tid < 128 && s_part == 0) inv_s[s_row] = has_norm ? rsqrtf(s0 * p.inv_d + p.eps) : 1.0f;
  }

  auto run_ztail = [&]() {
    auto body_z = [&](bf16x8* bc, bf16x8* bn, int strip) {
      const int next = strip + G;
      if constexpr (MULTI) load_strip(bn, DIA_PREFETCH_CLAMP(next, p.nstrips));     // unconditional: see k_gemv_small
#ifdef DIA_PIN_PREFETCH
      if constexpr (MULTI) __builtin_amdgcn_sched_barrier(0);     // keep the requests IN FRONT of the MFMAs they are meant to run under
#endif
      f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int i = 0; i < KPW; ++i)
#pragma unroll
        for (int pl = 0; pl < DIA_NPLANES; ++pl)
          acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][pl], bc[i], acc[0], 0, 0, 0);
      reduce_to_tile<1, NW, true>(acc, red, tile, tid, lane, w);
      const bool last_slice = splitk_combine(p, tile, strip, tid, &sk_flag);      // workgroup-uniform; true without split-K
      if (e_thread) {
        const int n0 = strip * 16 + e_half * 8;
        if (last_slice) run_epilogue(p, tile + e_r * 17, inv_s[e_r], (p.epi == DIA_EPI_CROSSKV ? zrow0 : 0) + e_r, n0, e_half, strip, e_live, xpre8, gpre8);
        if (MULTI && next < p.nstrips && resid) load_resid8(next);
      }
    };
    if constexpr (MULTI) {
      int strip = blockIdx.x;                    // strip pairs, then at most one more (see k_gemv_small)
      for (; strip + G < p.nstrips; strip += 2 * G) {
        body_z(b0, b1, strip);
        body_z(b1, b0, strip + G);
      }
      if (strip < p.nstrips) body_z(b0, b1, strip);
    } else {
      body_z(b0, b1, blockIdx.x);
    }
  };
  if constexpr (ZT) { run_ztail(); return; }

  // Cross-wave sum + epilogue, one tile element per thread (256 threads): the NW partial tiles go to LDS whole, a thread adds
  // the partials of ITS element in wave order (one barrier, double-buffered over the strips of the persistent form)
  // and runs the epilogue arithmetic for it — the three-way bf16 split of 256 elements spread over four waves instead
  // of eight per thread on half a wave.  The planes still leave as 16-byte stores: the bf16 triples are staged in LDS
  // (1.5 KB) and 96 threads store a row half of a plane each (768 two-byte stores per tile were measured slower).
  // epilogue of one finished tile: thread t < 256 holds element (t / 16, t % 16) in `v`; xp / gp = its residual and norm weight
  auto finish = [&](int strip, float v, float xpre1, float gpre1) {
    {
      const bool live = r_thread && r16 < p.M;
      if (p.epi == DIA_EPI_SCALE_STORE) {
        if (live) {
          const int s_out = p.strip_map ? p.strip_map[strip] : strip;
          p.out[(long)r16 * p.ldo + s_out * 16 + c16] = v * inv_s[r16];
        }
      } else {
        float e = 0.f;                              // the value whose planes are emitted
        int ecol = c16;                             // its column inside the emitted row segment
        [[maybe_unused]] bool emit = false;
        if (resid) {
          if (r_thread) {
            const int n = strip * 16 + c16;
            const float xv = xpre1 + v;
            if (live) p.out[(long)r16 * p.ldo + n] = xv;
            const float sqv = mul_rn(xv, xv);
            float accs = sqv;
#pragma unroll
            for (int j = 1; j < 8; ++j) {
              const float t = DIA_ROW_SHR(accs, 1);
              if ((c16 & 7) == j) accs = add_rn(t, sqv);
            }
            const float h0 = DIA_ROW_SHR(accs, 8);
            if (live && c16 == 15) p.ssq_out[(long)strip * p.ssq_ld + r16] = h0 + accs;
            e = mul_rn(xv, gpre1);
            emit = live;
          }
        } else {                                    // SWIGLU: columns 0..7 gate, 8..15 up
          if (r_thread) {
            const float up_raw = DIA_ROW_SHL(v, 8);
            const float inv = inv_s[r16];
            const float g = v * inv, u = up_raw * inv;
            e = (g / (1.0f + expf(-g))) * u;
            emit = live && c16 < 8;
          }
        }
        if constexpr (PF32) {                       // fp32 tiles: one value per thread, staged so that they leave as 16-byte stores
          float* Pf = reinterpret_cast<float*>(p.P);
          if (resid && p.cmap) {
            if (emit) {
              const int cc = p.cmap[strip * 16 + c16];
              if (cc >= 0) Pf[plane_frag_off(r16, cc & ~7, p.p_ktiles) + (cc & 7)] = e;
            }
          } else {
            float* stgf = reinterpret_cast<float*>(stg);            // [16][16] fp32 (1 KB of the 1.5 KB staging area)
            if (r_thread) stgf[r16 * 16 + ecol] = e;
            lds_barrier();
            STAMP(4);
            if (resid) {
              if (tid < 64) {
                const int mm = tid >> 2, q = tid & 3;
                if (mm < p.M)
                  *reinterpret_cast<float4*>(Pf + plane_frag_off(mm, strip * 16 + (q >> 1) * 8, p.p_ktiles) + (q & 1) * 4) =
                      *reinterpret_cast<const float4*>(&stgf[mm * 16 + q * 4]);
              }
            } else if (tid < 32) {
              const int mm = tid >> 1, q = tid & 1;
              if (mm < p.M)
                *reinterpret_cast<float4*>(Pf + plane_frag_off(mm, strip * 8, p.p_ktiles) + q * 4) =
                    *reinterpret_cast<const float4*>(&stgf[mm * 16 + q * 4]);
            }
          }
          return;
        }
        if constexpr (PF32) return;                 // (not reached: keeps the planes code below out of the fp32 instantiations)
        __bf16 ea, eb, ec;
        split3(e, ea, eb, ec);
        if (resid && p.cmap) {                      // compacted consumer: scattered two-byte stores
          if (emit) {
            const int cc = p.cmap[strip * 16 + c16];
            if (cc >= 0) {
              const long off = plane_frag_off(r16, cc & ~7, p.p_ktiles) + (cc & 7);
              p.P[off] = *reinterpret_cast<bf16_raw*>(&ea);
              p.P[p.p_plane_stride + off] = *reinterpret_cast<bf16_raw*>(&eb);
              p.P[2 * p.p_plane_stride + off] = *reinterpret_cast<bf16_raw*>(&ec);
            }
          }
        } else {
          if (r_thread) {
            stg[(0 * 16 + r16) * 16 + ecol] = *reinterpret_cast<bf16_raw*>(&ea);
            stg[(1 * 16 + r16) * 16 + ecol] = *reinterpret_cast<bf16_raw*>(&eb);
            stg[(2 * 16 + r16) * 16 + ecol] = *reinterpret_cast<bf16_raw*>(&ec);
          }
          lds_barrier();
          STAMP(4);
          if (resid) {
            if (tid < 96) {
              const int pl = tid >> 5, mm = (tid & 31) >> 1, hh = tid & 1;
              if (mm < p.M) {
                const bf16x8 t8 = *reinterpret_cast<const bf16x8*>(&stg[(pl * 16 + mm) * 16 + hh * 8]);
                *reinterpret_cast<bf16x8*>(p.P + pl * p.p_plane_stride + plane_frag_off(mm, strip * 16 + hh * 8, p.p_ktiles)) = t8;
              }
            }
          } else if (tid < 48) {
            const int pl = tid >> 4, mm = tid & 15;
            if (mm < p.M) {
              const bf16x8 t8 = *reinterpret_cast<const bf16x8*>(&stg[(pl * 16 + mm) * 16]);
              *reinterpret_cast<bf16x8*>(p.P + pl * p.p_plane_stride + plane_frag_off(mm, strip * 8, p.p_ktiles)) = t8;
            }
          }
        }
      }
    }
  };
  // Split-K with exactly two strips per workgroup (wo at 5..16 rows: 64 strip pairs x 4 K ranges = 256 workgroups, one per
  // CU, one round): both tiles are computed first and handed over TOGETHER — one slab publication, one ticket, one merge
  // by the last arriver — instead of one dependent hand-off chain per strip.  (512 one-strip workgroups at 160 VGPRs run
  // in two rounds, one workgroup per CU: 14.3 us; a hand-off per strip inside the persistent loop: 14.5 us.)
  if constexpr (PAIR) {
    {
      const int s0 = blockIdx.x, s1 = blockIdx.x + G;
      float xpreB = 0.f, gpreB = 1.f;
      if (resid && r_thread) { const float xa = xpre1, ga = gpre1; load_resid(s1); xpreB = xpre1; gpreB = gpre1; xpre1 = xa; gpre1 = ga; }
      load_strip(b1, s1);
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < KPW; ++i)
#pragma unroll
        for (int pl = 0; pl < DIA_NPLANES; ++pl) acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][pl], b0[i], acc0, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < KPW; ++i)
#pragma unroll
        for (int pl = 0; pl < DIA_NPLANES; ++pl) acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][pl], b1[i], acc1, 0, 0, 0);
      STAMP(2);
      red[w * 64 + lane] = acc0;
      red[NW * 64 + w * 64 + lane] = acc1;
      lds_barrier();
      float v0 = 0.f, v1 = 0.f;
      if (r_thread) {
        const float* rf = reinterpret_cast<const float*>(red) + (c16 + 16 * (r16 >> 2)) * 4 + (r16 & 3);
        v0 = rf[0]; v1 = rf[NW * 256];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) { v0 += rf[ww * 256]; v1 += rf[NW * 256 + ww * 256]; }
      }
      STAMP(3);
      // publish both partial tiles (element-per-thread, 4-byte coherent stores would be slow: go through LDS rows)
      const int SK = gridDim.y, ks = blockIdx.y;
      float* tile1 = tile + 16 * 17 + 16;               // second tile behind tile + inv_s
      if (r_thread) { tile[r16 * 17 + c16] = v0; tile1[r16 * 17 + c16] = v1; }
      lds_barrier();
      const __amdgpu_buffer_rsrc_t sr = agent_rsrc(p.sk_scratch);
#ifdef DIA_X_NOHANDOFF
      if (ks != SK - 1) return;                           // TIMING ONLY: no slab, no ticket, no merge (wrong results)
      finish(s0, v0, xpre1, gpre1);
      lds_barrier();
      finish(s1, v1, xpreB, gpreB);
      return;
#endif
      if (tid < 128) {
        const int tsel = tid >> 6, row = (tid & 63) >> 2, c4 = (tid & 3) * 4;
        const float* t = (tsel ? tile1 : tile) + row * 17 + c4;
        st4_agent(sr, (int)((((long)(tsel ? s1 : s0) * SK + ks) * 256 + row * 16 + c4) * 4), f32x4{t[0], t[1], t[2], t[3]});
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(p.sk_tickets + s0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = ticket == SK - 1;
        if (last) __hip_atomic_store(p.sk_tickets + s0, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sk_flag = last;
      }
      __syncthreads();
      STAMP(4);
      if (!sk_flag) return;
      if (tid < 128) {
        const int tsel = tid >> 6, row = (tid & 63) >> 2, c4 = (tid & 3) * 4;
        const long sb = (long)(tsel ? s1 : s0) * SK;
        f32x4 q[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          if (kk < SK) q[kk] = ld4_agent(sr, (int)(((sb + kk) * 256 + row * 16 + c4) * 4));
        f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
          if (kk < SK) { sum[0] += q[kk][0]; sum[1] += q[kk][1]; sum[2] += q[kk][2]; sum[3] += q[kk][3]; }
        for (int kk = 4; kk < SK; ++kk) {
          const f32x4 t = ld4_agent(sr, (int)(((sb + kk) * 256 + row * 16 + c4) * 4));
          sum[0] += t[0]; sum[1] += t[1]; sum[2] += t[2]; sum[3] += t[3];
        }
        float* t = (tsel ? tile1 : tile) + row * 17 + c4;
        t[0] = sum[0]; t[1] = sum[1]; t[2] = sum[2]; t[3] = sum[3];
      }
      lds_barrier();
      if (r_thread) { v0 = tile[r16 * 17 + c16]; v1 = tile1[r16 * 17 + c16]; }
      finish(s0, v0, xpre1, gpre1);
      lds_barrier();        // the first tile's staged planes have left LDS before the second tile's values overwrite them
      finish(s1, v1, xpreB, gpreB);
      STAMP(5);
      return;
    }
  }
  int sbuf = 0;
  constexpr int DIST = (ALDS && DIA_ZR_RING3) ? 2 : 1;      // strips in flight ahead of the one being multiplied
  auto body = [&](bf16x8* bc, bf16x8* bn, int strip) {
    const int next = strip + G;
    if constexpr (MULTI) load_strip(bn, DIA_PREFETCH_CLAMP(strip + DIST * G, p.nstrips));       // unconditional: see k_gemv_small
#ifdef DIA_PIN_PREFETCH
    if constexpr (MULTI) __builtin_amdgcn_sched_barrier(0);     // keep the requests IN FRONT of the MFMAs they are meant to run under
#endif
    f32x4 acc[1] = {f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int i = 0; i < KPW; ++i)
#pragma unroll
      for (int pl = 0; pl < DIA_NPLANES; ++pl)
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afrag(i, pl), bc[i], acc[0], 0, 0, 0);
    STAMP(2);
    float v = 0.f;
    bool last_slice = true;
    if (gridDim.y == 1) {
      f32x4* rb = red + sbuf * (NW * 64);
      sbuf ^= 1;
      rb[w * 64 + lane] = acc[0];
      lds_barrier();
      if (r_thread) {
        const float* rf = reinterpret_cast<const float*>(rb) + (c16 + 16 * (r16 >> 2)) * 4 + (r16 & 3);
        v = rf[0];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) v += rf[ww * 256];
      }
      STAMP(3);
    } else {
      reduce_to_tile<1, NW, true>(acc, red, tile, tid, lane, w);
      STAMP(3);
      last_slice = splitk_combine(p, tile, strip, tid, &sk_flag);      // workgroup-uniform
      STAMP(4);
      if (last_slice && r_thread) v = tile[r16 * 17 + c16];
    }
    if (last_slice) finish(strip, v, xpre1, gpre1);
    if (MULTI && next < p.nstrips && resid && r_thread) load_resid(next);
    STAMP(5);
  };
  if constexpr (MULTI && ALDS && DIA_ZR_RING3) {
    // ring of three (measured, not used: 17 530 vs 17 862 frames/s at batch 64 — a strip is not latency-bound): the registers the mid / lo planes gave up hold a third strip, so TWO strips of weights are in flight
    // behind the one being multiplied (a strip lasts 0.4-0.5 us of MFMA + tail, a load under the chip-wide stream 1.5-2 us)
    bf16x8 b2[KPW];
    load_strip(b1, DIA_PREFETCH_CLAMP(blockIdx.x + G, p.nstrips));
    for (int strip = blockIdx.x; strip < p.nstrips; strip += 3 * G) {
      body(b0, b2, strip);
      if (strip + G < p.nstrips) body(b1, b0, strip + G);
      if (strip + 2 * G < p.nstrips) body(b2, b1, strip + 2 * G);
    }
  } else if constexpr (MULTI) {
    // strip PAIRS, then at most one more: with `if (second strip exists) body(b1, b0, ...)` inside the loop the compiler put an
    // s_waitcnt vmcnt(0) at the loop header (the two ways into it leave different loads pending), i.e. the strip requested
    // by the second body had to land before the next trip could start — found in the ISA, round 3
    int strip = blockIdx.x;
    for (; strip + G < p.nstrips; strip += 2 * G) {
      body(b0, b1, strip);
      body(b1, b0, strip + G);
    }
    if (strip < p.nstrips) body(b0, b1, strip);
  } else {
    body(b0, b1, blockIdx.x);
  }
}


// ---------------------------------------------------------------------------------------------------
// M <= 4 rows, long K over few strips (wo: K = 8192, N = 2048): the DIAGONAL weight layout (layout.diag_tile_weight).
// With 16-column strips only 128 workgroups exist, so wo splits K over two workgroups per strip and pays the slab hand-off
// (1.75 us of a 9.5 us launch).  Here a tile carries FOUR k-tiles of FOUR columns — lane l, element j =
// W[128 t + 32 ((l & 15) >> 2) + 8 (l >> 4) + j][4 group + (l & 3)] — and the A operand carries the <= 4 rows of the matching
// k-tile in each group of four rows, so the 16 x 16 product is block diagonal (the blocks D[4g + r][4g + c] summed over g give
// row r, column c; the rest is ignored).  Column granularity 4: a workgroup owns 8 columns = two groups x K / 128 tiles,
// 256 workgroups share N = 2048 with the WHOLE K each — no cross-workgroup reduction.  16 waves: waves 0..7 the first group,
// 8..15 the second, K split eight ways inside a group.  Epilogue RESID_EMIT (x += acc; fp32 tile of x * g_next; one sum of
// squares per 8-column half strip: the consumers add twice as many partials).  fp32 activation tiles in and out.
template <int RS>
__global__ __launch_bounds__(1024) void k_gemv_diag(GemmK p) {
  constexpr int NW = 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw);                                   // [NW][16]: lane 20g + c -> entry 4g + c
  bf16x8* As = reinterpret_cast<bf16x8*>(smem_raw + sizeof(f32x4) * NW * 16);         // [3 planes][KT][4 kq][RS]
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int KT = p.KT;                       // k-tiles of 32 (K / 32); tiles of the diagonal layout: KT / 4 per column group
  const int TPG = KT >> 2;                   // tiles per column group
  const int TPW = TPG >> 3;                  // tiles per wave (8 waves per group)
  const int hs = blockIdx.x;                 // half strip: columns [8 hs, 8 hs + 8)
  const int cg = w >> 3, wk = w & 7;
  // ---- small operands first (vmcnt retires in order): the fp32 image of the <= RS rows, residual + norm weight
  const int nentries = KT * 4 * RS;          // 32-byte entries (k-tile, quarter, row)
  const float* Af = reinterpret_cast<const float*>(p.A);
  float4 ex[RS], ey[RS];                     // (K = 8192: 1024 entries per row)
#pragma unroll
  for (int u = 0; u < RS; ++u) {
    const int c = min(tid + u * 1024, nentries - 1);
    const int row = c % RS, kq = (c / RS) & 3, kt = c / (4 * RS);
    const float4* src = reinterpret_cast<const float4*>(Af + ((long)kt * 64 + row + 16 * kq) * 8);
    ex[u] = src[0]; ey[u] = src[1];
  }
  const int er = tid >> 3, ec = tid & 7;     // epilogue thread (row, column of the 8)
  const bool e_thread = tid < 8 * RS, e_live = e_thread && er < p.M;
  const int n = 8 * hs + ec;
  float xpre = 0.f, gpre = 1.f;
  if (e_thread) { xpre = p.out[(long)(e_live ? er : 0) * p.ldo + n]; gpre = p.gnext[n]; }
  __builtin_amdgcn_sched_barrier(0);
  // ---- this wave's tiles (TPW <= 8 of 1 KiB): the HBM stream
  const bf16x8* Wl = reinterpret_cast<const bf16x8*>(p.W) + ((long)(2 * hs + cg) * TPG + wk * TPW) * 64 + lane;
  bf16x8 b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) b[i] = DIA_WLOAD(Wl + (long)min(i, TPW - 1) * 64);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < RS; ++u)
    if (tid + u * 1024 < nentries) {
      const int c = tid + u * 1024;
      bf16x8 h, mi, lo;
      split3x8(ex[u], ey[u], h, mi, lo);
      As[c] = h; As[nentries + c] = mi; As[2 * nentries + c] = lo;
    }
  lds_barrier();
  const int g = (lane & 15) >> 2, r = min(lane & 3, RS - 1), kq = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i < TPW) {
      const int kt = 4 * (wk * TPW + i) + g;
#pragma unroll
      for (int pl = 0; pl < DIA_NPLANES; ++pl) {
        const bf16x8 a = As[((pl * KT + kt) * 4 + kq) * RS + r];
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b[i], acc, 0, 0, 0);
      }
    }
  }
  if (((lane & 15) >> 2) == (lane >> 4)) red[w * 16 + 4 * (lane >> 4) + (lane & 3)] = acc;       // diagonal block g, column c: rows in the registers
  lds_barrier();
  if (e_thread) {
    // column ec of group ec >> 2: waves 8 (ec >> 2) .. +7, blocks g = 0..3, in fixed order
    const float* rf = reinterpret_cast<const float*>(red) + ((ec >> 2) * 8 * 16 + (ec & 3)) * 4 + er;
    float v = 0.f;
#pragma unroll
    for (int ww = 0; ww < 8; ++ww)
#pragma unroll
      for (int gg = 0; gg < 4; ++gg) v += rf[(ww * 16 + 4 * gg) * 4];
    const float xv = xpre + v;
    if (e_live) p.out[(long)er * p.ldo + n] = xv;
    float sq = mul_rn(xv, xv);
    sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
    if (e_live && ec == 0) p.ssq_out[(long)hs * p.ssq_ld + er] = sq;
    if (e_live) reinterpret_cast<float*>(p.P)[plane_frag_off(er, n & ~7, p.p_ktiles) + (n & 7)] = mul_rn(xv, gpre);
  }
}

// ---------------------------------------------------------------------------------------------------
// 17..128 rows, K = 2048, dense decode shapes (qkv, o, cq, co, wi, logits at batch 9-64): TWO m-tiles per workgroup.
// The z-form above streams every weight byte once per m-tile through L2 -> CU (eight times at 128 rows: 536 MB per wi
// launch, the L2's practical limit: profiles/r03_pmc_zform_l2.txt) and pays the per-strip reduce / epilogue once per
// m-tile.  Here a workgroup multiplies each strip it loads against 32 rows: its waves still split K eight ways, the hi
// and mid planes of both m-tiles' A fragments stay in registers (128 VGPRs), the lo planes in a wave-private part of
// LDS (2 x 8 KiB per wave), all 512 threads take one element of the two finished 16 x 16 tiles each.  gridDim.z = pairs
// of m-tiles (the last pair may hold one).  Same arithmetic in the same order as k_gemm16: results are bit-identical.
// fp32 activation tiles in and out; epilogues SCALE_STORE / RESID_EMIT / SWIGLU_EMIT without compaction maps.
// KPW = 8 (K = 2048 per workgroup) or 4 (K = 1024: the encoder's shapes); AF32 = fp32 activation tiles in and out with the 512-thread
// tail (decode), else planes in and out with the 32-threads-per-tile tail through the shared epilogue (every epilogue incl. CROSSKV and
// the compaction maps: the short-prompt prefill, 33..128 rows).
// NW = 4 (planes, K = 1024 as 4 waves x 8 k-tiles, 256 threads, 74 KB of LDS): TWO workgroups per CU — the planes tail is 32 threads
// per tile between two barriers, and with one workgroup per CU nothing else runs meanwhile (13.9 us per launch at 98 rows for 8-17 MB).
constexpr size_t g2t_smem(int kpw, int nw = 8, bool sk2 = false) { return (size_t)2 * nw * kpw * 64 * 16 + sizeof(f32x4) * 2 * nw * 64 + 2 * 1024 + sizeof(float) * 32 + (sk2 ? 8192 : 0); }
// CKV (NW = 4, bf16 caches with the blocked V layout, no strip map): the cross-K/V tail on ALL 256 threads with UNCONDITIONAL stores.  vmcnt counts
// stores as well as loads and retires in order; the wait before a strip's first MFMA is one static count for every wave, so it is the
// count of the path that issued the FEWEST memory operations since the awaited load — with the tail on two of the waves (or behind
// `if (live)`) those waves sat out the acknowledgement of their 16 scattered 2-byte stores before every strip: 2 us per strip,
// 138 us for the merged launch of the 18 layers where the same GEMM with a plain fp32 store takes 63.  Here every thread of every
// wave issues the same two stores (rows past the end or of padding: into a sink) and the same two table loads for the next strip.
__device__ bf16_raw g_ckv_sink[1024];
// EPI >= 0 (fp32 tiles, no split-K): the epilogue is a compile-time constant and its tail issues the SAME memory operations on every thread of every
// wave (dead rows and idle threads store into g_sink16, the next strip's residual is requested on a clamped index) — for the reason above: behind
// run-time branches (`if (live)`, `if (tid < 128)`, the epilogue kind) the one static wait in front of the next strip's MFMAs is the count of the
// emptiest path, and every wave that did store sits out the acknowledgement of its stores.  EPI = -1: the epilogue kind at run time (planes, split-K).
__device__ float4 g_sink16[1024];
// SK2 (fp32 tiles, split-K, RESID_EMIT = wo at 17..128 rows): the hand-off of the K quarters once per GROUP of four (then two) strips — eight or four partial tiles, one slab
// publication, one ticket, one merge.  A hand-off is two dependent coherent round trips (slab stores acknowledged, then the ticket) that nothing
// overlaps: priced with a build that skips it, 13 of wo's 37 us at 128 rows (eight per workgroup).  Same slabs, same summation order: bit-identical.
template <int KPW, bool AF32, bool SPLITK, int NW = 8, bool CKV = false, int EPI = -1, bool SK2 = false>
__global__ __launch_bounds__(NW * 64, NW == 4 ? 2 : 1) void k_gemm2t(const bf16_raw* a_A, long a_aps, const bf16_raw* a_W, int a_KT, int a_M, int a_epi,
                                               int a_nstrips, float* a_out, int a_ldo, const float* a_gnext, GemmK p) {
  static_assert(NW == 8 || (NW == 4 && !AF32 && !SPLITK), "the 256-thread form serves the planes path without split-K");
  static_assert(!CKV || (NW == 4 && KPW == 8), "the all-thread cross-K/V tail belongs to the 256-thread form");
  static_assert(EPI < 0 || (!SPLITK && !CKV && (NW == 8 || (!AF32 && EPI != DIA_EPI_RESID_EMIT))), "the uniform tails: no split-K; 256 threads only for planes without the residual");
  static_assert(!SK2 || (SPLITK && NW == 8 && EPI < 0 && !CKV), "the grouped hand-off belongs to the split-K form");
  const int epi = EPI >= 0 ? EPI : a_epi;
  p.A = a_A; p.a_plane_stride = a_aps; p.W = a_W; p.KT = a_KT; p.M = a_M; p.epi = a_epi; p.nstrips = a_nstrips;
  p.out = a_out; p.ldo = a_ldo; p.gnext = a_gnext;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16x8* alo = reinterpret_cast<bf16x8*>(smem_raw);                                           // [2 tiles][NW][KPW][64]
  f32x4* red = reinterpret_cast<f32x4*>(smem_raw + (size_t)2 * NW * KPW * 64 * 16);             // [2 tiles][NW][64]
  float* stgf = reinterpret_cast<float*>(smem_raw + (size_t)2 * NW * KPW * 64 * 16 + sizeof(f32x4) * 2 * NW * 64);   // [2 tiles][16][16]
  float* inv_s = stgf + 2 * 256;                                                               // [32]
  [[maybe_unused]] float* m4 = inv_s + 32;                                                     // SK2: [4 strips][2 tiles][16][16]
  [[maybe_unused]] const bool dia_sk2_quads = p.sk2_quads != 0;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int mtiles = (p.M + 15) >> 4;
  const int mt0 = 2 * blockIdx.z;                          // first m-tile of this workgroup
  const int Ml = min(32, p.M - 16 * mt0);                  // rows it holds (1..32)
  const int mt1 = min(mt0 + 1, mtiles - 1);                // second m-tile, or the first once more when there is none (never stored)
  // SPLITK (its own instantiation: behind a run-time branch its stores and atomics made the waits of the plain form conservative,
  // wi 35 -> 42 us at 128 rows): split-K over gridDim.y workgroups (wo: four K quarters), merged by the last arriver
  const int SK = SPLITK ? gridDim.y : 1, ks = SPLITK ? blockIdx.y : 0;
  const int kt0 = ks * (NW * KPW) + w * KPW;
  const int G = gridDim.x;
  __shared__ int sk_flag;
  const bf16x8* Wl = reinterpret_cast<const bf16x8*>(p.W) + (long)kt0 * 64 + lane;
  auto load_strip = [&](bf16x8* b, int strip) {
    const bf16x8* Wt = Wl + (long)strip * p.KT * 64;
#pragma unroll
    for (int i = 0; i < KPW; ++i) b[i] = (gridDim.z > 1 && ZTEMPORAL) ? *(Wt + (long)i * 64) : DIA_WLOAD(Wt + (long)i * 64);
  };
  bf16x8 b0[KPW], b1[KPW];
  // A fragments of both m-tiles (fp32 tiles): hi / mid to registers, lo to this wave's own LDS region
  bf16x8 ah[2][KPW], am[2][KPW];
  bf16x8* my = alo + (long)w * KPW * 64 + lane;            // + tile * NW * KPW * 64 + i * 64
  if constexpr (AF32) {
    const float* Af = reinterpret_cast<const float*>(p.A);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int mt = t ? mt1 : mt0;
      const int rows = t ? max(Ml - 16, 1) : min(Ml, 16);
      const int alane = (lane & 48) | min(lane & 15, rows - 1);
      const float4* src = reinterpret_cast<const float4*>(Af + (((long)mt * p.a_ktiles + kt0) * 64 + alane) * 8);
      float4 x0[KPW], x1[KPW];
#pragma unroll
      for (int i = 0; i < KPW; ++i) { x0[i] = src[(long)i * 128]; x1[i] = src[(long)i * 128 + 1]; }
#pragma unroll
      for (int i = 0; i < KPW; ++i) {
        bf16x8 lo;
        split3x8(x0[i], x1[i], ah[t][i], am[t][i], lo);
        my[(t * NW * KPW + i) * 64] = lo;
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int mt = t ? mt1 : mt0;
      const int rows = t ? max(Ml - 16, 1) : min(Ml, 16);
      const int alane = (lane & 48) | min(lane & 15, rows - 1);
#pragma unroll
      for (int i = 0; i < KPW; ++i) {
        const long off = (((long)mt * p.a_ktiles + kt0 + i) * 64 + alane) * 8;
        ah[t][i] = *reinterpret_cast<const bf16x8*>(p.A + off);
        am[t][i] = *reinterpret_cast<const bf16x8*>(p.A + p.a_plane_stride + off);
        my[(t * NW * KPW + i) * 64] = *reinterpret_cast<const bf16x8*>(p.A + 2 * p.a_plane_stride + off);
      }
    }
  }
  // row scales of the 32 rows: 8 threads per row sum the strip partials in the order of k_gemm16
  const bool has_norm = p.ssq_in != nullptr;
  {
    const int s_row = tid >> 3, s_part = tid & 7;
    const bool s_thread = tid < 256 && has_norm;
    float sq[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) sq[i] = 0.f;
    if (s_thread) {
      const float* sp = p.ssq_in + 16 * mt0 + min(s_row, Ml - 1);
#pragma unroll
      for (int i = 0; i < 16; ++i) sq[i] = sp[(long)min(s_part + 8 * i, p.ssq_in_n - 1) * p.ssq_ld];
    }
    float s0 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s0 += (s_part + 8 * i < p.ssq_in_n && s_row < Ml) ? sq[i] : 0.f;
    if (s_thread && s_row < Ml)
      for (int idx = s_part + 128; idx < p.ssq_in_n; idx += 8) s0 += p.ssq_in[(long)idx * p.ssq_ld + 16 * mt0 + s_row];
    s0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0xB1, 0xF, 0xF, true));
    s0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0x4E, 0xF, 0xF, true));
    s0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s0), 0x141, 0xF, 0xF, true));
    if (tid < 256 && s_part == 0) inv_s[s_row] = has_norm ? rsqrtf(s0 * p.inv_d + p.eps) : 1.0f;
  }
  // epilogue geometry: thread t owns element (row 16 * (t >> 8) + ((t >> 4) & 15), column t & 15) of the two finished tiles
  const int ti = tid >> 8, r16 = (tid >> 4) & 15, c16 = tid & 15;
  const int rl = 16 * ti + r16;                            // row inside the workgroup's 32
  const bool live = rl < Ml;
  const long grow = 16 * mt0 + rl;                         // row of the whole batch
  const bool resid = epi == DIA_EPI_RESID_EMIT;
  float xpre1 = 0.f, gpre1 = 1.f;
  auto load_resid = [&](int strip) {
    const int n = strip * 16 + c16;
    xpre1 = p.out[(live ? grow : (long)16 * mt0) * p.ldo + n];
    gpre1 = p.gnext[n];
  };
  // planes path: 32 threads per tile, 8 tile elements each, through the shared epilogue (as the z-form's tail)
  // (tile 0: the first 32 lanes of wave 0, tile 1: of wave 1 — the tile is wave-uniform, so its shifted pointers stay scalar)
  const int e_t = __builtin_amdgcn_readfirstlane(tid >> 6) & 1, e_r = (tid >> 1) & 15, e_half = tid & 1;
  const bool e_thread = !AF32 && tid < 128 && (tid & 63) < 32, e_live = e_thread && 16 * e_t + e_r < Ml;
  GemmK pe = p;                                            // this thread's m-tile: row-indexed pointers shifted (CROSSKV addresses rows globally)
  if constexpr (!AF32) {
    const int mt = mt0 + e_t;
    if (pe.out) pe.out += (long)16 * mt * p.ldo;
    if (pe.P) pe.P += (long)mt * p.p_ktiles * 512;
    if (pe.ssq_out) pe.ssq_out += 16 * mt;
  }
  float xpre8[8], gpre8[8];
  auto load_resid8 = [&](int strip) {
    const int n0 = strip * 16 + e_half * 8;
    const float* o = pe.out + (long)(e_live ? e_r : 0) * p.ldo + n0;
    const float4 xa = *reinterpret_cast<const float4*>(o), xb = *reinterpret_cast<const float4*>(o + 4);
    xpre8[0] = xa.x; xpre8[1] = xa.y; xpre8[2] = xa.z; xpre8[3] = xa.w;
    xpre8[4] = xb.x; xpre8[5] = xb.y; xpre8[6] = xb.z; xpre8[7] = xb.w;
    const float4 ga = *reinterpret_cast<const float4*>(p.gnext + n0), gb = *reinterpret_cast<const float4*>(p.gnext + n0 + 4);
    gpre8[0] = ga.x; gpre8[1] = ga.y; gpre8[2] = ga.z; gpre8[3] = ga.w;
    gpre8[4] = gb.x; gpre8[5] = gb.y; gpre8[6] = gb.z; gpre8[7] = gb.w;
  };
  // CROSSKV: a thread finishes the SAME row for every strip its workgroup walks — utterance and text position are looked up once (inside the
  // strip loop the chain row_b -> seg_off -> cos / sin was three dependent round trips per strip, 3.5 us per (strip, m-tile pair) unit),
  // and the RoPE table entries of a strip are requested while the previous one is still being finished
  // (compile-time off where K per workgroup is not 1024 — the encoder width, the only K a cross-K/V projection has: the host sends no other here)
  const bool ckv = !AF32 && NW * KPW == 32 && p.epi == DIA_EPI_CROSSKV;
  int ep_row = e_r;                                        // row handed to the shared epilogue: inside the m-tile, or (CROSSKV) the text position
  bool ep_live = e_live;
  if (ckv && e_thread) {
    const int m = 16 * (mt0 + e_t) + e_r;
    int b = p.kv_batch_index;
    ep_row = m;
    if (p.row_b) {
      b = e_live ? p.row_b[m] : -1;
      ep_live = b >= 0;
      ep_row = ep_live ? m - p.seg_off[b] : 0;
    }
    pe.kv_batch_index = b;
  }
  if (ckv) { pe.row_b = nullptr; pe.seg_off = nullptr; }     // (workgroup-uniform: the pointers stay scalar)
  auto load_cs = [&](int strip) {
    int s = p.strip_map ? p.strip_map[strip] : strip;
    if (p.kv_layer_strips > 0) s %= p.kv_layer_strips;
    if (s < p.kv_heads * 8) {
      const int o = ep_row * 64 + (s & 7) * 8 + e_half * 4;    // (int: scalar base + 32-bit offset, one VGPR less than a 64-bit address)
      const float4 cc = *reinterpret_cast<const float4*>(p.cos_t + o), ss = *reinterpret_cast<const float4*>(p.sin_t + o);
      xpre8[0] = cc.x; xpre8[1] = cc.y; xpre8[2] = cc.z; xpre8[3] = cc.w;
      gpre8[0] = ss.x; gpre8[1] = ss.y; gpre8[2] = ss.z; gpre8[3] = ss.w;
    }
  };
  // CKV: thread (tile tt, u) finishes, for a K strip, RoPE pair u & 7 of row u >> 3 (8 threads = 16 contiguous bytes of a cache row, and
  // again 64 elements further); for a V strip, columns u >> 4 and 8 + (u >> 4) of row u & 15 (16 threads = 32 contiguous bytes of the
  // blocked layout).  Both rows are looked up once.
  [[maybe_unused]] const int ck_tt = tid >> 7, ck_rK = (tid & 127) >> 3, ck_t8 = tid & 7, ck_rV = tid & 15, ck_cV = (tid & 127) >> 4;
  [[maybe_unused]] int ck_posK = 0, ck_posV = 0;
  [[maybe_unused]] long ck_baseK = 0, ck_baseV = 0;
  [[maybe_unused]] bool ck_liveK = false, ck_liveV = false;
  [[maybe_unused]] float ck_c = 0.f, ck_s = 0.f;
  if constexpr (CKV) {
    auto look = [&](int r, int& pos, bool& lv, long& base) {
      const int rl_ = 16 * ck_tt + r, m = 16 * mt0 + rl_;
      int b = p.kv_batch_index;
      pos = m; lv = rl_ < Ml;
      if (p.row_b) {
        b = lv ? p.row_b[m] : -1;
        lv = b >= 0;
        pos = lv ? m - p.seg_off[b] : 0;
      }
      base = (long)(lv ? b : 0) * p.kv_heads * p.kv_cap * 128;
    };
    look(ck_rK, ck_posK, ck_liveK, ck_baseK);
    look(ck_rV, ck_posV, ck_liveV, ck_baseV);
  }
  auto load_cs_all = [&](int strip) {                       // unconditional (a V strip loads two entries nobody reads)
    int s = strip;
    if (p.kv_layer_strips > 0) s %= p.kv_layer_strips;
    const int o = ck_posK * 64 + (s & 7) * 8 + ck_t8;
    ck_c = p.cos_t[o]; ck_s = p.sin_t[o];
  };
  if constexpr (CKV) load_cs_all(blockIdx.x);
  else if constexpr (AF32 || EPI >= 0 || SK2) { if (resid) load_resid(blockIdx.x); }
  else { if (resid && e_thread) load_resid8(blockIdx.x); if (ckv && e_thread) load_cs(blockIdx.x); }
  __builtin_amdgcn_sched_barrier(0);
  load_strip(b0, blockIdx.x);
  __builtin_amdgcn_sched_barrier(0);
  float* Pf = reinterpret_cast<float*>(p.P);

  auto body = [&](bf16x8* bc, bf16x8* bn, int strip) {
    const int next = strip + G;
#ifndef DIA_X2T_NOWLOAD
    load_strip(bn, DIA_PREFETCH_CLAMP(next, p.nstrips));       // unconditional: see k_gemv_small
#endif
#ifndef DIA_X2T_NOPIN
    // without this the scheduler sinks the requests behind the MFMAs they are meant to run under (both weight buffers live across the MFMA
    // block is what the register budget was drawn for).  Only where it fits: the other instantiations spill 2..34 VGPRs when pinned
    if constexpr (EPI >= 0 || CKV) __builtin_amdgcn_sched_barrier(0);
#endif
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#ifndef DIA_X2T_NOMFMA
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < KPW; ++i) {
        const bf16x8 lo = my[(t * NW * KPW + i) * 64];
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t][i], bc[i], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[t][i], bc[i], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo, bc[i], acc[t], 0, 0, 0);
      }
#else
    acc[0][0] = __builtin_bit_cast(float, (int)bc[0][0]); acc[1][0] = __builtin_bit_cast(float, (int)bc[KPW - 1][0]);
#endif
    red[(0 * NW + w) * 64 + lane] = acc[0];
    red[(1 * NW + w) * 64 + lane] = acc[1];
    lds_barrier();
    float v;
    [[maybe_unused]] float v_t1 = 0.f;                      // NW == 4: 256 threads, every thread sums its element of BOTH tiles
    {
      const float* rf = reinterpret_cast<const float*>(red + ti * NW * 64) + (c16 + 16 * (r16 >> 2)) * 4 + (r16 & 3);
      v = rf[0];
#pragma unroll
      for (int ww = 1; ww < NW; ++ww) v += rf[ww * 256];
      if constexpr (NW == 4) {
        const float* rg = rf + NW * 256;
        v_t1 = rg[0];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) v_t1 += rg[ww * 256];
      }
    }
    bool last = true;
#ifdef DIA_X2T_NOHANDOFF
    if constexpr (SPLITK) last = ks == SK - 1;               // TIMING ONLY: no slab, no ticket, no merge (wrong results)
    if constexpr (false) {
#else
    if constexpr (SPLITK) {
#endif
      // both partial tiles leave together: one slab publication, one ticket, one merge by the last arriver in split order
      // (the protocol of splitk_combine: sc1 stores acknowledged before the ticket, sc1 loads after it; no fences)
      stgf[ti * 256 + r16 * 16 + c16] = v;
      lds_barrier();
      const __amdgpu_buffer_rsrc_t sr = agent_rsrc(p.sk_scratch);
      const long unit = (long)blockIdx.z * p.nstrips + strip;
      const int tt = (tid >> 6) & 1, row = (tid & 63) >> 2, c4 = (tid & 3) * 4;
      if (tid < 128) {
        const float* f = &stgf[tt * 256 + row * 16 + c4];
        st4_agent(sr, (int)(((unit * SK + ks) * 512 + tt * 256 + row * 16 + c4) * 4), f32x4{f[0], f[1], f[2], f[3]});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      lds_barrier();
      if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(p.sk_tickets + unit, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int lst = ticket == SK - 1;
        if (lst) __hip_atomic_store(p.sk_tickets + unit, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        sk_flag = lst;
      }
      lds_barrier();
      last = sk_flag != 0;
      if (last) {
        if (tid < 128) {
          // SK <= 4 in every launch the hosts sends here: every slab requested before the first is used (the plain loop took them one
          // coherent round trip at a time — 3 us on the workgroup that arrives last, a quarter of all units)
          f32x4 sv[4];
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (k < SK) sv[k] = ld4_agent(sr, (int)(((unit * SK + k) * 512 + tt * 256 + row * 16 + c4) * 4));
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (k < SK) { a[0] += sv[k][0]; a[1] += sv[k][1]; a[2] += sv[k][2]; a[3] += sv[k][3]; }
          for (int k = 4; k < SK; ++k) {
            const f32x4 t = ld4_agent(sr, (int)(((unit * SK + k) * 512 + tt * 256 + row * 16 + c4) * 4));
            a[0] += t[0]; a[1] += t[1]; a[2] += t[2]; a[3] += t[3];
          }
          float* f = &stgf[tt * 256 + row * 16 + c4];
          f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3];
        }
        lds_barrier();
        v = stgf[ti * 256 + r16 * 16 + c16];
      }
      lds_barrier();                                        // stgf is staged again below; red is free
    }
    if constexpr (CKV) {
      stgf[r16 * 16 + c16] = v;
      stgf[256 + r16 * 16 + c16] = v_t1;
      lds_barrier();
      {
        int s = strip;
        long lofs = 0;
        if (p.kv_layer_strips > 0) {
          const int layer = s / p.kv_layer_strips;
          s -= layer * p.kv_layer_strips;
          lofs = (long)layer * p.kv_layer_stride;
        }
        const int nk = p.kv_heads * 8;
        bf16_raw *d0, *d1;
        float o0, o1;
        bool lv;
        if (s < nk) {                                         // K: RoPE pair (i, i + 64) of one row  (run_epilogue's arithmetic)
          const int head = s >> 3, i = (s & 7) * 8 + ck_t8;
          const float inv = inv_s[16 * ck_tt + ck_rK];
          const float* tr = stgf + ck_tt * 256 + ck_rK * 16 + 2 * ck_t8;
          const float x1 = tr[0] * inv, x2 = tr[1] * inv;
          o0 = x1 * ck_c - x2 * ck_s; o1 = x1 * ck_s + x2 * ck_c;
          d0 = reinterpret_cast<bf16_raw*>(p.kc) + lofs + ck_baseK + ((long)head * p.kv_cap + ck_posK) * 128 + i;
          d1 = d0 + 64;
          lv = ck_liveK;
        } else {                                              // V, blocked [key / 32][128 dims][32 keys]
          const int sv = s - nk, head = sv >> 3, dd = (sv & 7) * 16 + ck_cV;
          const float inv = inv_s[16 * ck_tt + ck_rV];
          const float* tr = stgf + ck_tt * 256 + ck_rV * 16 + ck_cV;
          o0 = tr[0] * inv; o1 = tr[8] * inv;
          d0 = reinterpret_cast<bf16_raw*>(p.vc) + lofs + ck_baseV + (long)head * p.kv_cap * 128 + (long)(ck_posV >> 5) * 4096 + (ck_posV & 31) + dd * 32;
          d1 = d0 + 8 * 32;
          lv = ck_liveV;
        }
        if (!lv) { d0 = g_ckv_sink + tid; d1 = g_ckv_sink + 512 + tid; }
        KVElem<bf16_raw>::store(d0, o0);
        KVElem<bf16_raw>::store(d1, o1);
      }
      load_cs_all(DIA_PREFETCH_CLAMP(next, p.nstrips));
      lds_barrier();                                          // the staging area is rewritten by the next strip
    } else if constexpr (!AF32 && EPI >= 0) {
      // planes out, compile-time epilogue: element-per-thread arithmetic (256 threads: one element of EACH tile), then ONE 16-byte plane
      // fragment per thread — every thread of every wave issues the same stores (idle ones into the sink), see EPI above
      constexpr int NE = NW == 8 ? 1 : 2;
      float* const sinkf = reinterpret_cast<float*>(g_sink16);
      const float vv[2] = {v, v_t1};
      if constexpr (EPI == DIA_EPI_SCALE_STORE) {
#pragma unroll
        for (int j = 0; j < NE; ++j) {
          const int rlj = NW == 8 ? rl : 16 * j + r16;
          float* dst = rlj < Ml ? p.out + (long)(16 * mt0 + rlj) * p.ldo + strip * 16 + c16 : sinkf + tid + 512 * j;
          *dst = vv[j] * inv_s[rlj];
        }
        lds_barrier();                                        // red is rewritten by the next strip
      } else {
#pragma unroll
        for (int j = 0; j < NE; ++j) {
          const int te = NW == 8 ? ti : j, rlj = 16 * te + r16;
          float e;
          if constexpr (EPI == DIA_EPI_RESID_EMIT) {           // (NW == 8: one element per thread)
            const int n = strip * 16 + c16;
            const float xv = xpre1 + vv[j];
            float* dst = live ? p.out + grow * p.ldo + n : sinkf + tid;
            *dst = xv;
            const float sqv = mul_rn(xv, xv);
            float accs = sqv;
#pragma unroll
            for (int jj = 1; jj < 8; ++jj) {
              const float t = DIA_ROW_SHR(accs, 1);
              if ((c16 & 7) == jj) accs = add_rn(t, sqv);
            }
            const float h0 = DIA_ROW_SHR(accs, 8);
            float* sd = (live && c16 == 15) ? p.ssq_out + (long)strip * p.ssq_ld + grow : sinkf + 512 + tid;
            *sd = h0 + accs;
            e = mul_rn(xv, gpre1);
          } else {                                            // SWIGLU: columns 0..7 gate, 8..15 up
            const float up_raw = DIA_ROW_SHL(vv[j], 8);
            const float inv = inv_s[rlj];
            const float g = vv[j] * inv, u = up_raw * inv;
            e = (g / (1.0f + expf(-g))) * u;
          }
          stgf[te * 256 + r16 * 16 + c16] = e;
        }
        lds_barrier();                                        // (also: red is free again)
        {
          // RESID: 2 tiles x 16 rows x 2 halves x 3 planes = 192 fragments; SWIGLU: 2 x 16 x 3 = 96 (8 outputs per row and strip)
          constexpr bool RS_ = EPI == DIA_EPI_RESID_EMIT;
          const int pl = RS_ ? tid >> 6 : tid >> 5;
          const int tt = RS_ ? (tid >> 5) & 1 : (tid >> 4) & 1, mm = RS_ ? (tid >> 1) & 15 : tid & 15, hf = RS_ ? tid & 1 : 0;
          const bool on = pl < 3 && 16 * tt + mm < Ml;
          const float* src = &stgf[tt * 256 + mm * 16 + hf * 8];
          bf16x8 h, mi, lo;
          split3x8(*reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 4), h, mi, lo);
          const bf16x8 frag = pl == 0 ? h : (pl == 1 ? mi : lo);
          bf16_raw* Pt = p.P + (long)min(pl, 2) * p.p_plane_stride + (long)(mt0 + tt) * p.p_ktiles * 512;
          bf16x8* dst = on ? reinterpret_cast<bf16x8*>(Pt + plane_frag_off(mm, RS_ ? strip * 16 + hf * 8 : strip * 8, p.p_ktiles))
                           : reinterpret_cast<bf16x8*>(g_sink16 + tid);
          *dst = frag;
        }
        if constexpr (EPI == DIA_EPI_RESID_EMIT) load_resid(DIA_PREFETCH_CLAMP(next, p.nstrips));
        lds_barrier();                                        // the staging area is rewritten by the next strip
      }
    } else if constexpr (!AF32) {
      // planes / every epilogue: the finished tiles in LDS rows, 32 threads per tile run the shared epilogue
      if (last) stgf[ti * 256 + r16 * 16 + c16] = v;
      if constexpr (NW == 4) stgf[256 + r16 * 16 + c16] = v_t1;
      lds_barrier();
      if (e_thread) {
        const int n0 = strip * 16 + e_half * 8;
#ifndef DIA_X2T_NOEPI
        if (last) run_epilogue(pe, stgf + e_t * 256 + e_r * 16, inv_s[16 * e_t + e_r], ep_row, n0, e_half, strip, ep_live, xpre8, gpre8, ckv);
#endif
        if (next < p.nstrips && resid) load_resid8(next);
        if (next < p.nstrips && ckv) load_cs(next);
      }
      lds_barrier();
    } else if constexpr (EPI >= 0) {
      float* const sinkf = reinterpret_cast<float*>(g_sink16);
      if constexpr (EPI == DIA_EPI_SCALE_STORE) {
        float* dst = live ? p.out + grow * p.ldo + strip * 16 + c16 : sinkf + tid;
        *dst = v * inv_s[rl];
        lds_barrier();                                        // red is rewritten by the next strip
      } else {
        float e;
        if constexpr (EPI == DIA_EPI_RESID_EMIT) {
          const int n = strip * 16 + c16;
          const float xv = xpre1 + v;
          float* dst = live ? p.out + grow * p.ldo + n : sinkf + tid;
          *dst = xv;
          const float sqv = mul_rn(xv, xv);
          float accs = sqv;
#pragma unroll
          for (int j = 1; j < 8; ++j) {
            const float t = DIA_ROW_SHR(accs, 1);
            if ((c16 & 7) == j) accs = add_rn(t, sqv);
          }
          const float h0 = DIA_ROW_SHR(accs, 8);
          float* sd = (live && c16 == 15) ? p.ssq_out + (long)strip * p.ssq_ld + grow : sinkf + 512 + tid;
          *sd = h0 + accs;
          e = mul_rn(xv, gpre1);
        } else {                                              // SWIGLU: columns 0..7 gate, 8..15 up
          const float up_raw = DIA_ROW_SHL(v, 8);
          const float inv = inv_s[rl];
          const float g = v * inv, u = up_raw * inv;
          e = (g / (1.0f + expf(-g))) * u;
        }
        stgf[ti * 256 + r16 * 16 + c16] = e;
        lds_barrier();                                        // (also: red is free again)
        // 16-byte stores of the staged values: RESID 4 per row (64 threads per tile), SWIGLU 2 per row (32 threads per tile) — issued by ALL threads
        {
          const int tt = tid >> 6 & 1, u = tid & 63;
          float* Pt = Pf + (long)(mt0 + tt) * p.p_ktiles * 512;
          float4* dst;
          const float* src;
          if constexpr (EPI == DIA_EPI_RESID_EMIT) {
            const int mm = u >> 2, q = u & 3;
            const bool on = tid < 128 && 16 * tt + mm < Ml;
            dst = on ? reinterpret_cast<float4*>(Pt + plane_frag_off(mm, strip * 16 + (q >> 1) * 8, p.p_ktiles) + (q & 1) * 4) : g_sink16 + tid;
            src = &stgf[tt * 256 + mm * 16 + q * 4];
          } else {
            const int mm = (u >> 1) & 15, q = u & 1;
            const bool on = tid < 128 && u < 32 && 16 * tt + mm < Ml;
            dst = on ? reinterpret_cast<float4*>(Pt + plane_frag_off(mm, strip * 8, p.p_ktiles) + q * 4) : g_sink16 + tid;
            src = &stgf[tt * 256 + mm * 16 + q * 4];
          }
          *dst = *reinterpret_cast<const float4*>(src);
        }
        if constexpr (EPI == DIA_EPI_RESID_EMIT) load_resid(DIA_PREFETCH_CLAMP(next, p.nstrips));
        lds_barrier();                                        // the staging area is rewritten by the next strip
      }
    } else if (!last) {
      if (next < p.nstrips && resid) load_resid(next);
    } else if (p.epi == DIA_EPI_SCALE_STORE) {
      if (live) p.out[grow * p.ldo + strip * 16 + c16] = v * inv_s[rl];
      lds_barrier();                                        // red is rewritten by the next strip
    } else {
      float e = 0.f;
      [[maybe_unused]] bool emit = false;
      if (resid) {
        const int n = strip * 16 + c16;
        const float xv = xpre1 + v;
        if (live) p.out[grow * p.ldo + n] = xv;
        const float sqv = mul_rn(xv, xv);
        float accs = sqv;
#pragma unroll
        for (int j = 1; j < 8; ++j) {
          const float t = DIA_ROW_SHR(accs, 1);
          if ((c16 & 7) == j) accs = add_rn(t, sqv);
        }
        const float h0 = DIA_ROW_SHR(accs, 8);
        if (live && c16 == 15) p.ssq_out[(long)strip * p.ssq_ld + grow] = h0 + accs;
        e = mul_rn(xv, gpre1);
        emit = live;
      } else {                                              // SWIGLU: columns 0..7 gate, 8..15 up
        const float up_raw = DIA_ROW_SHL(v, 8);
        const float inv = inv_s[rl];
        const float g = v * inv, u = up_raw * inv;
        e = (g / (1.0f + expf(-g))) * u;
        emit = live && c16 < 8;
      }
      stgf[ti * 256 + r16 * 16 + c16] = e;
      lds_barrier();                                        // (also: red is free again)
      // the staged values leave as 16-byte stores: RESID 4 per row (64 threads per tile), SWIGLU 2 per row (32 threads per tile)
      const int tt = tid >> 6 & 1;                          // waves 0 / 1 store tile 0 / 1
      if (tid < 128) {
        const int u = tid & 63;
        const int mt = mt0 + tt;
        float* Pt = Pf + (long)mt * p.p_ktiles * 512;
        if (resid) {
          const int mm = u >> 2, q = u & 3;
          if (16 * tt + mm < Ml)
            *reinterpret_cast<float4*>(Pt + plane_frag_off(mm, strip * 16 + (q >> 1) * 8, p.p_ktiles) + (q & 1) * 4) =
                *reinterpret_cast<const float4*>(&stgf[tt * 256 + mm * 16 + q * 4]);
        } else if (u < 32) {
          const int mm = u >> 1, q = u & 1;
          if (16 * tt + mm < Ml)
            *reinterpret_cast<float4*>(Pt + plane_frag_off(mm, strip * 8, p.p_ktiles) + q * 4) =
                *reinterpret_cast<const float4*>(&stgf[tt * 256 + mm * 16 + q * 4]);
        }
      }
      if (next < p.nstrips && resid) load_resid(next);
      lds_barrier();                                        // the staging area is rewritten by the next strip
    }
  };
  lds_barrier();          // inv_s
  int strip = blockIdx.x;                                   // strip pairs, then at most one more (see k_gemv_small)
  if constexpr (SK2) {
    // weights x both tiles of one strip -> this thread's element of the K-quarter's partial tiles
    auto mm = [&](bf16x8* bc, bf16x8* bn, int pre) -> float {
      load_strip(bn, DIA_PREFETCH_CLAMP(pre, p.nstrips));
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
          const bf16x8 lo = my[(t * NW * KPW + i) * 64];
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[t][i], bc[i], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[t][i], bc[i], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(lo, bc[i], acc[t], 0, 0, 0);
        }
      red[(0 * NW + w) * 64 + lane] = acc[0];
      red[(1 * NW + w) * 64 + lane] = acc[1];
      lds_barrier();
      const float* rf = reinterpret_cast<const float*>(red + ti * NW * 64) + (c16 + 16 * (r16 >> 2)) * 4 + (r16 & 3);
      float v = rf[0];
#pragma unroll
      for (int ww = 1; ww < NW; ++ww) v += rf[ww * 256];
      return v;
    };
    // RESID_EMIT for one finished strip (the arithmetic of the run-time tail above)
    auto tail_resid = [&](int s, float v, float xp, float gp) {
      const int n = s * 16 + c16;
      const float xv = xp + v;
      if (live) p.out[grow * p.ldo + n] = xv;
      const float sqv = mul_rn(xv, xv);
      float accs = sqv;
#pragma unroll
      for (int j = 1; j < 8; ++j) {
        const float t = DIA_ROW_SHR(accs, 1);
        if ((c16 & 7) == j) accs = add_rn(t, sqv);
      }
      const float h0 = DIA_ROW_SHR(accs, 8);
      if (live && c16 == 15) p.ssq_out[(long)s * p.ssq_ld + grow] = h0 + accs;
      stgf[ti * 256 + r16 * 16 + c16] = mul_rn(xv, gp);
      lds_barrier();
      if constexpr (AF32) {
        if (tid < 128) {
          const int tt = tid >> 6 & 1, u = tid & 63, mm_ = u >> 2, q = u & 3;
          float* Pt = Pf + (long)(mt0 + tt) * p.p_ktiles * 512;
          if (16 * tt + mm_ < Ml)
            *reinterpret_cast<float4*>(Pt + plane_frag_off(mm_, s * 16 + (q >> 1) * 8, p.p_ktiles) + (q & 1) * 4) =
                *reinterpret_cast<const float4*>(&stgf[tt * 256 + mm_ * 16 + q * 4]);
        }
      } else if (tid < 192) {                               // planes: 2 tiles x 16 rows x 2 halves x 3 planes, one 16-byte fragment each
        const int pl = tid >> 6, tt = (tid >> 5) & 1, mm_ = (tid >> 1) & 15, hf = tid & 1;
        if (16 * tt + mm_ < Ml) {
          const float* src = &stgf[tt * 256 + mm_ * 16 + hf * 8];
          bf16x8 h, mi, lo;
          split3x8(*reinterpret_cast<const float4*>(src), *reinterpret_cast<const float4*>(src + 4), h, mi, lo);
          *reinterpret_cast<bf16x8*>(p.P + (long)pl * p.p_plane_stride + (long)(mt0 + tt) * p.p_ktiles * 512 + plane_frag_off(mm_, s * 16 + hf * 8, p.p_ktiles)) =
              pl == 0 ? h : (pl == 1 ? mi : lo);
        }
      }
      lds_barrier();                                        // the staging area is rewritten by the next strip
    };
    const __amdgpu_buffer_rsrc_t sr = agent_rsrc(p.sk_scratch);
    // N = 4 or 2 strips (even: the two weight buffers alternate, the next group starts in b0 again) -> one hand-off
    auto group = [&](auto n_tag) {
      constexpr int N = decltype(n_tag)::value;
      float xp[N], gp[N];
      xp[0] = xpre1; gp[0] = gpre1;                         // (requested by the previous group / the prologue)
#pragma unroll
      for (int j = 1; j < N; ++j) {
        const int n = (strip + j * G) * 16 + c16;
        xp[j] = p.out[(live ? grow : (long)16 * mt0) * p.ldo + n];
        gp[j] = p.gnext[n];
      }
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const float v = (j & 1) ? mm(b1, b0, strip + (j + 1) * G) : mm(b0, b1, strip + (j + 1) * G);
        m4[j * 512 + ti * 256 + r16 * 16 + c16] = v;
        lds_barrier();                                      // red is rewritten by the next product
      }
      // 2 N partial tiles leave together: thread t carries floats 4t..4t+3 of strip t >> 7 (tile (t >> 6) & 1, row (t & 63) >> 2)
      const int ps = tid >> 7, po = (tid & 127) * 4;
      const long u0 = (long)blockIdx.z * p.nstrips + strip;
      const long ub = u0 + (long)ps * G;
      if (ps < N) {
        const float* f = &m4[ps * 512 + po];
        st4_agent(sr, (int)(((ub * SK + ks) * 512 + po) * 4), f32x4{f[0], f[1], f[2], f[3]});
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      lds_barrier();
      if (tid == 0) {
        const int ticket = __hip_atomic_fetch_add(p.sk_tickets + u0, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int lst = ticket == SK - 1;
        if (lst) __hip_atomic_store(p.sk_tickets + u0, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        sk_flag = lst;
      }
      lds_barrier();
      const bool last = sk_flag != 0;
      if (last) {
        if (ps < N) {
          f32x4 sv[4];
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (k < SK) sv[k] = ld4_agent(sr, (int)(((ub * SK + k) * 512 + po) * 4));
          f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (k < SK) { a[0] += sv[k][0]; a[1] += sv[k][1]; a[2] += sv[k][2]; a[3] += sv[k][3]; }
          for (int k = 4; k < SK; ++k) {
            const f32x4 t = ld4_agent(sr, (int)(((ub * SK + k) * 512 + po) * 4));
            a[0] += t[0]; a[1] += t[1]; a[2] += t[2]; a[3] += t[3];
          }
          float* f = &m4[ps * 512 + po];
          f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3];
        }
        lds_barrier();
#pragma unroll
        for (int j = 0; j < N; ++j) tail_resid(strip + j * G, m4[j * 512 + ti * 256 + r16 * 16 + c16], xp[j], gp[j]);
      }
      load_resid(DIA_PREFETCH_CLAMP(strip + N * G, p.nstrips));
      lds_barrier();                                        // m4 / sk_flag are rewritten by the next group
      strip += N * G;
    };
    if (dia_sk2_quads)
      while (strip + 3 * G < p.nstrips) group(std::integral_constant<int, 4>{});
    while (strip + G < p.nstrips) group(std::integral_constant<int, 2>{});
    if (strip < p.nstrips) {
      if constexpr (!AF32) { if (resid && e_thread) load_resid8(strip); }     // (the shared planes tail takes 8 residual values per thread)
      body(b0, b1, strip);
    }
    return;
  }
  for (; strip + G < p.nstrips; strip += 2 * G) {
    body(b0, b1, strip);
    body(b1, b0, strip + G);
  }
  if (strip < p.nstrips) body(b0, b1, strip);
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
  launch_kernel<k_gemm_tile_ws<KC, PD, NPW>>(dim3((k.nstrips + 15) / 16, mgroups), dim3((8 + NPW) * 64), gt_smem(KC, 8), st, k);
  return dia_check_launch("k_gemm_tile_ws");
}


int launch_tile(const GemmK& k, hipStream_t st) {
  int rc = dia_kernels_init_once();
  if (rc) return rc;
  // the wave-specialised form (8 consumer + 4 producer waves): wi 113 us, wo 79, qkv 52, o 29 at 1696 rows, against
  // 137 / 97 / 61 / 34 for the plain 8-wave form and 192 / 86 / 79 / 30 for 4-wave 64 x 128 blocks (those two live in
  // gemm_experiments.hip, tuning knob tile_v = 0 / 1 / 2)
  const int v = dia_tune(DIA_TUNE_TILE_V);
  if (v == 4) return launch_tile_ws<2, 2, 2>(k, st);
  if (v == 5) return launch_tile_ws<2, 4, 4>(k, st);
  return launch_tile_ws<2, 2, 4>(k, st);
}

template <int NW, int KPW, bool AF32 = false, bool PF32 = false>
int launch_g16(const GemmK& k, hipStream_t st) {
  const size_t smem = g16_smem(NW);
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
  if (dia_tune(DIA_TUNE_GEMM_SPW) > 0) spw = dia_tune(DIA_TUNE_GEMM_SPW);
  if constexpr (!(NW == 16 && KPW >= 4)) {
    if (spw > 1) {      // persistent multi-strip form, with or without split-K: A fragments loaded once per workgroup
      int gx = (k.nstrips + spw - 1) / spw;
      if (mz >= 2 && (gx * sk) % 8 != 0 && (gx + 7) / 8 * 8 <= k.nstrips) gx = (gx + 7) / 8 * 8;   // pairs on one XCD
      if constexpr (NW == 8 && (KPW == 8 || KPW == 4)) {
        if (mz > 1 && sk == 1 && k.epi != DIA_EPI_CROSSKV && dia_tune(DIA_TUNE_GEMM_ZR) != 0) {       // mid / lo planes of A in LDS: room for the element-per-thread tail
          // (not with split-K: wo at 128 rows 59.9 vs 56.0 us — its hand-off drains the stream either way, the 32-thread tail is shorter there)
          launch_small_kernel<k_gemm16<NW, KPW, true, true, AF32, PF32, false, true>>(dim3(gx, sk, mz), dim3(NW * 64), g16_smem(NW) + g16_alds(NW, KPW), st, k);
          return dia_check_launch("k_gemm16");
        }
      }
      if (mz > 1) launch_small_kernel<k_gemm16<NW, KPW, true, true, AF32, PF32>>(dim3(gx, sk, mz), dim3(NW * 64), smem, st, k);
      else if (KPW == 8 && sk > 1 && 2 * gx == k.nstrips)      // strip pairs handed over together (wo at 5..16 rows)
        launch_small_kernel<k_gemm16<NW, (KPW == 8 ? 8 : KPW), true, false, AF32, PF32, KPW == 8>>(dim3(gx, sk), dim3(NW * 64), smem, st, k);
      else {
        if constexpr (NW == 8 && KPW == 8 && AF32 && PF32) {
          // 5..16 rows, persistent form with fp32 input (wi, logits at batch 3-8): with the mid / lo planes of A in LDS the registers
          // suffice for the element-per-thread tail (the 32-thread tail was what fitted before) — knob gemm_zr=0 / 3 keeps the old form
          if (sk == 1 && dia_tune(DIA_TUNE_GEMM_ZR) != 0 && dia_tune(DIA_TUNE_GEMM_ZR) != 3) {
            launch_small_kernel<k_gemm16<NW, KPW, true, false, AF32, PF32, false, true>>(dim3(gx, sk), dim3(NW * 64), g16_smem(NW) + g16_alds(NW, KPW), st, k);
            return dia_check_launch("k_gemm16");
          }
        }
        launch_small_kernel<k_gemm16<NW, KPW, true, false, AF32, PF32>>(dim3(gx, sk), dim3(NW * 64), smem, st, k);
      }
      return dia_check_launch("k_gemm16");
    }
  }
  if (mz > 1) launch_small_kernel<k_gemm16<NW, KPW, false, true, AF32, PF32>>(dim3(k.nstrips, sk, mz), dim3(NW * 64), smem, st, k);
  else launch_small_kernel<k_gemm16<NW, KPW, false, false, AF32, PF32>>(dim3(k.nstrips, sk), dim3(NW * 64), smem, st, k);
  return dia_check_launch("k_gemm16");
}

// returns true when a k_gemm16 instantiation exists for (nw, KT)
int launch_g16_any(const GemmK& k, int nw, int sk, hipStream_t st, bool& handled) {
  handled = true;
  const int kpw = (k.KT % (nw * sk) == 0) ? k.KT / (nw * sk) : 0;
  if (k.a_f32 || k.p_f32) {      // fp32 activation tiles on either side: the 8-wave forms (every decode shape from 5 rows on)
    const bool emits = k.epi == DIA_EPI_RESID_EMIT || k.epi == DIA_EPI_SWIGLU_EMIT;
    const bool pf = emits && k.p_f32;
    if (nw == 8 && k.a_f32 && pf == emits) {           // fp32 in, fp32 out (or nothing emitted)
      if (kpw == 1) return launch_g16<8, 1, true, true>(k, st);
      if (kpw == 2) return launch_g16<8, 2, true, true>(k, st);
      if (kpw == 3) return launch_g16<8, 3, true, true>(k, st);
      if (kpw == 4) return launch_g16<8, 4, true, true>(k, st);
      if (kpw == 5) return launch_g16<8, 5, true, true>(k, st);
      if (kpw == 6) return launch_g16<8, 6, true, true>(k, st);
      if (kpw == 7) return launch_g16<8, 7, true, true>(k, st);
      if (kpw == 8) return launch_g16<8, 8, true, true>(k, st);
    }
    handled = false;                                   // mixed formats: the generic kernel (run-time flags)
    return DIA_OK;
  }
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
  launch_kernel<k_gemm<MT, NW, KPW>>(dim3(k.nstrips, mgroups), dim3(NW * 64), smem, st, k);
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

template <int NW, int KPW, int RS, bool F32 = false>
int launch_small(const GemmK& k, hipStream_t st) {
  size_t smem = small_smem(NW, NW * KPW, RS);
  if (smem > 64 * 1024) {
    int rc = dia_kernels_init_once();     // raises the dynamic-LDS limit of every large-LDS kernel, once
    if (rc) return rc;
  }
  // strips per workgroup: enough workgroups to cover every CU, few enough that each streams several
  // strips back to back (next strip's loads overlap this strip's reduce + epilogue)
  int spw = k.spw > 0 ? k.spw : (k.nstrips >= 1024 ? 4 : 1);
  const int sk = k.KT / (NW * KPW);          // cross-workgroup split-K factor (1 = none)
  // between one and two rounds of resident workgroups (logits head: 579 strips, two 8-wave workgroups per CU): walk the
  // strips with about 256 persistent workgroups instead of leaving a short second round
  if (k.spw <= 0 && sk == 1 && k.nstrips > 512 && k.nstrips < 1024) spw = (k.nstrips + 255) / 256;
  if (dia_tune(DIA_TUNE_GEMM_SPW) > 0) spw = dia_tune(DIA_TUNE_GEMM_SPW);
  const int grid = (k.nstrips + spw - 1) / spw;
  if (sk > 1) {
    launch_small_kernel<k_gemv_small<NW, KPW, RS, false, F32, F32>>(dim3(k.nstrips, sk), dim3(NW * 64), smem, st, k);
    return dia_check_launch("k_gemv_small");
  }
  if (spw > 1) {
    if constexpr (KPW <= 16 && !(NW == 16 && KPW > 4))
      launch_small_kernel<k_gemv_small<NW, KPW, RS, true, F32, F32>>(dim3(grid), dim3(NW * 64), smem, st, k);
    else
      launch_small_kernel<k_gemv_small<NW, KPW, RS, false, F32, F32>>(dim3(k.nstrips), dim3(NW * 64), smem, st, k);
  } else {
    launch_small_kernel<k_gemv_small<NW, KPW, RS, false, F32, F32>>(dim3(k.nstrips), dim3(NW * 64), smem, st, k);
  }
  return dia_check_launch("k_gemv_small");
}

template <int RS, bool F32 = false>
int launch_small_rs(const GemmK& k, int nw, int sk, hipStream_t st, bool& handled) {
  handled = true;
  const int kpw = (k.KT % (nw * sk) == 0) ? k.KT / (nw * sk) : 0;
  if (nw == 4) {
    if (kpw == 4) return launch_small<4, 4, RS, F32>(k, st);
    if (kpw == 8) return launch_small<4, 8, RS, F32>(k, st);
    if (kpw == 16) return launch_small<4, 16, RS, F32>(k, st);
  } else if (nw == 8) {
    if (kpw == 2) return launch_small<8, 2, RS, F32>(k, st);
    if (kpw == 3) return launch_small<8, 3, RS, F32>(k, st);      // 3, 5, 6, 7: K-compacted (pruned) shapes, K % 256 == 0
    if (kpw == 4) return launch_small<8, 4, RS, F32>(k, st);
    if (kpw == 5) return launch_small<8, 5, RS, F32>(k, st);
    if (kpw == 6) return launch_small<8, 6, RS, F32>(k, st);
    if (kpw == 7) return launch_small<8, 7, RS, F32>(k, st);
    if (kpw == 8) return launch_small<8, 8, RS, F32>(k, st);
    if (kpw == 10) return launch_small<8, 10, RS, F32>(k, st);    // 10, 12, 14: compacted hidden widths (multiples of 1024) under split-K 2
    if (kpw == 12) return launch_small<8, 12, RS, F32>(k, st);
    if (kpw == 14) return launch_small<8, 14, RS, F32>(k, st);
    if (kpw == 16) return launch_small<8, 16, RS, F32>(k, st);
    if (kpw == 32) return launch_small<8, 32, RS, F32>(k, st);
  } else if (nw == 16) {
    if (kpw == 1) return launch_small<16, 1, RS, F32>(k, st);
    if (kpw == 2) return launch_small<16, 2, RS, F32>(k, st);
    if (kpw == 4) return launch_small<16, 4, RS, F32>(k, st);
    if (kpw == 8) return launch_small<16, 8, RS, F32>(k, st);
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
  e[0] = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_small<NW, KPW, 2, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
  e[1] = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_small<NW, KPW, 4, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
  if constexpr (KPW <= 16 && !(NW == 16 && KPW > 4)) {
    e[2] = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_small<NW, KPW, 2, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
    e[3] = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_small<NW, KPW, 4, true, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
  }
  for (int i = 0; i < 4; ++i) if (e[i] != hipSuccess) return dia_fail_hip(e[i], "hipFuncSetAttribute(k_gemv_small, fp32 tiles)");
  return DIA_OK;
}

}  // namespace

#ifdef DIA_DBG_STAMPS
extern "C" int dia_dbg_stamps(long long* host, int n) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(long long) * n) == hipSuccess ? 0 : -2;
}
#endif

int dia_gemm_init() {
  int rc = 0;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_diag<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemv_diag<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm2t<8, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2t_smem(8)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm2t<8, true, false, 8, false, DIA_EPI_SCALE_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2t_smem(8)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm2t<8, true, false, 8, false, DIA_EPI_RESID_EMIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2t_smem(8)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm2t<8, true, false, 8, false, DIA_EPI_SWIGLU_EMIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2t_smem(8)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm2t<8, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2t_smem(8)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm2t<8, true, true, 8, false, -1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2t_smem(8, 8, true)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm2t<8, false, true, 8, false, -1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2t_smem(8, 8, true)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm2t<8, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2t_smem(8)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm2t<8, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2t_smem(8)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm2t<4, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2t_smem(4)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm2t<8, false, false, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2t_smem(8, 4)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm2t<8, false, false, 4, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2t_smem(8, 4)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm2t<8, false, false, 4, false, DIA_EPI_SCALE_STORE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2t_smem(8, 4)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm2t<8, false, false, 4, false, DIA_EPI_SWIGLU_EMIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2t_smem(8, 4)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm2t<8, false, false, 8, false, DIA_EPI_RESID_EMIT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)g2t_smem(8)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm16<8, 8, true, true, true, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(g16_smem(8) + g16_alds(8, 8))) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm16<8, 8, true, false, true, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(g16_smem(8) + g16_alds(8, 8))) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm16<8, 8, true, true, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(g16_smem(8) + g16_alds(8, 8))) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm16<8, 4, true, true, true, true, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(g16_smem(8) + g16_alds(8, 4))) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm16<8, 4, true, true, false, false, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(g16_smem(8) + g16_alds(8, 4))) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tile_ws<2, 2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gt_smem(2, 8)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tile_ws<2, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gt_smem(2, 8)) != hipSuccess) rc = 1;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_tile_ws<2, 4, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)gt_smem(2, 8)) != hipSuccess) rc = 1;
  rc |= small_attr<4, 4>(); rc |= small_attr<4, 8>(); rc |= small_attr<4, 16>();
  rc |= small_attr<8, 2>(); rc |= small_attr<8, 3>(); rc |= small_attr<8, 4>(); rc |= small_attr<8, 5>(); rc |= small_attr<8, 6>(); rc |= small_attr<8, 7>(); rc |= small_attr<8, 8>(); rc |= small_attr<8, 10>(); rc |= small_attr<8, 12>(); rc |= small_attr<8, 14>(); rc |= small_attr<8, 16>(); rc |= small_attr<8, 32>();
  rc |= small_attr<16, 1>(); rc |= small_attr<16, 2>(); rc |= small_attr<16, 4>(); rc |= small_attr<16, 8>();
#ifdef DIA_EXPERIMENTS
  rc |= dia_exp_init();
#endif
  return rc ? DIA_E_HIP : DIA_OK;
}

extern "C" int dia_has_experiments(void) {
#ifdef DIA_EXPERIMENTS
  return 1;
#else
  return 0;
#endif
}

// Shape -> kernel.  Decode (M = 2 x batch rows):
//   M <= 4            k_gemv_small   (LDS-staged compact activations; persistent multi-strip form for N >= 16384)
//   5..16 rows        k_gemm16       (register-resident A fragments)
//   17..128 rows      k_gemm16 over gridDim.z m-tiles (weights shared through one XCD's L2)
// Prefill (hundreds of packed rows): k_gemm_tile_ws (MFMA-bound).  Everything else: k_gemm (any shape).
extern "C" int dia_gemm(const dia_gemm_args* a, void* stream) {
  if (!a || !a->A || (!a->W && !a->sp_blocks)) return dia_fail(DIA_E_ARG, "dia_gemm: null argument");
  if (a->M <= 0 || a->KT <= 0 || a->nstrips <= 0) return dia_fail(DIA_E_ARG, "dia_gemm: empty problem");
  if (a->w_layout == 1) {     // diagonal 4-column tiles (layout.diag_tile_weight): M <= 4, RESID_EMIT, fp32 tiles in and out
    const int groups = a->nstrips;               // here: 8-column half strips
    if (!a->out || !a->P || !a->ssq_out || a->ssq_ld < a->M || a->a_ktiles < a->KT) return dia_fail(DIA_E_ARG, "dia_gemm: diagonal layout: missing buffer");
    GemmK k;
    fill_gemmk(a, k);
    if (a->M > 4 || a->epi != DIA_EPI_RESID_EMIT || !a->gnext || a->cmap || (a->act_f32 & 3) != 3 || a->w_planes > 1 || a->sk > 1 ||
        a->KT % 32 != 0 || a->KT / 32 > 8 || a->p_ktiles * 32 < groups * 8 || a->ldo < groups * 8)
      return dia_fail(DIA_E_ARG, "dia_gemm: the diagonal weight layout serves M <= 4, RESID_EMIT, fp32 tiles, K a multiple of 1024 up to 8192");
    const int rs = a->M <= 2 ? 2 : 4;
    const size_t smem = sizeof(f32x4) * 16 * 16 + (size_t)3 * a->KT * 4 * rs * 16;
    if (smem > 152 * 1024) return dia_fail(DIA_E_ARG, "dia_gemm: diagonal layout: the activation image of these rows does not fit LDS (3-4 rows: K <= 4096)");
    if (rs == 2) launch_kernel<k_gemv_diag<2>>(dim3(groups), dim3(1024), smem, (hipStream_t)stream, k);
    else launch_kernel<k_gemv_diag<4>>(dim3(groups), dim3(1024), smem, (hipStream_t)stream, k);
    return dia_check_launch("k_gemv_diag");
  }
  if (a->KT > a->a_ktiles) return dia_fail(DIA_E_ARG, "dia_gemm: weight K exceeds the plane layout's K");
  if (a->a_plane_stride % 8 != 0 || a->p_plane_stride % 8 != 0) return dia_fail(DIA_E_ARG, "dia_gemm: plane stride must be a multiple of 8");
  if ((a->epi == DIA_EPI_SCALE_STORE || a->epi == DIA_EPI_RESID_EMIT) && (!a->out || (!a->strip_map && a->ldo < a->nstrips * 16) || a->ldo % 4 != 0))
    return dia_fail(DIA_E_ARG, "dia_gemm: output leading dimension too small");
  if ((a->epi == DIA_EPI_RESID_EMIT) && (!a->P || !a->ssq_out || a->p_ktiles * 32 < a->nstrips * 16))
    return dia_fail(DIA_E_ARG, "dia_gemm: RESID_EMIT needs planes and ssq_out covering N");
  if ((a->epi == DIA_EPI_SWIGLU_EMIT) && (!a->P || a->p_ktiles * 32 < a->nstrips * 8))
    return dia_fail(DIA_E_ARG, "dia_gemm: SWIGLU_EMIT needs planes covering N/2");
  if (a->epi == DIA_EPI_CROSSKV && (!a->kc || !a->vc || !a->cos_t || !a->sin_t || (!a->strip_map && a->nstrips != a->kv_heads * 16 && !(a->kv_layer_strips == a->kv_heads * 16 && a->nstrips % (a->kv_heads * 16) == 0)) ||
                                     a->kv_layer_strips < 0 || (a->kv_layer_strips > 0 && (a->kv_layer_strips != a->kv_heads * 16 || a->kv_layer_stride <= 0)) || (!a->row_b && a->M > a->kv_cap) || (!a->row_b != !a->seg_off) || (a->kv_vblocked && a->kv_cap % 32 != 0)))
    return dia_fail(DIA_E_ARG, "dia_gemm: CROSSKV shape mismatch");
  if (a->epi < 0 || a->epi > DIA_EPI_CROSSKV) return dia_fail(DIA_E_ARG, "dia_gemm: unknown epilogue");
  if (a->ssq_in && a->ssq_ld < ((a->M + 15) / 16) * 16) return dia_fail(DIA_E_ARG, "dia_gemm: ssq_ld smaller than padded rows");

  if (a->sp_blocks || a->sp_toff) {       // zero-skipping stream of an unstructured-pruned matrix: kernel-level experiment
#ifdef DIA_EXPERIMENTS
    return dia_exp_gemm_sparse(a, stream);
#else
    return dia_fail(DIA_E_ARG, "dia_gemm: the sparse weight stream needs a build with EXPERIMENTS=1");
#endif
  }
  GemmK k;
  fill_gemmk(a, k);
  if (a->w_planes > 1) {      // fp32 weights as three planes: the generic kernel, whatever the shape (exactness, not speed)
    if (a->w_planes != 3) return dia_fail(DIA_E_ARG, "dia_gemm: w_planes must be 0, 1 or 3");
    const int mt_ = (a->M + 15) / 16;
    const int nw_ = (a->KT % 8 == 0) ? 8 : 4;
    hipStream_t st_ = (hipStream_t)stream;
    if (mt_ == 1) return launch_nw<1>(k, nw_, 1, st_);
    if (mt_ == 2) return launch_nw<2>(k, nw_, 1, st_);
    return launch_nw<4>(k, nw_, (mt_ + 3) / 4, st_);
  }
  const bool fast_epi = a->epi != DIA_EPI_CROSSKV && !(a->epi == DIA_EPI_RESID_EMIT && !a->gnext);
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
  // cross-workgroup split-K (a->sk > 1): sk workgroups per strip, each 1/sk of K, combined by the last arriver.
  // The engine uses it for wo (K = 8192 over 128 strips): 2 at M <= 4, 4 at 5..16 rows and per m-tile above.
  const int sk = a->sk > 1 ? a->sk : 1;
  if (sk > 1 && (!a->sk_scratch || !a->sk_tickets || a->KT % sk != 0)) return dia_fail(DIA_E_ARG, "dia_gemm: split-K needs sk_scratch, sk_tickets and KT % sk == 0");
  // (M <= 4: sixteen waves x 8 k-tiles beat eight x 16 on wo, 9.4 vs 10.0 us with the fp32 image — half the loads per wave in
  // flight before its first MFMA.  The same wave count for both activation formats: their results stay identical bit for bit)
  if (sk > 1 && !a->nw) { const int ktl = a->KT / sk; nw = (ktl % 16 == 0 && ktl / 16 <= (a->M <= 4 ? 8 : 4)) ? 16 : ((ktl % 8 == 0) ? 8 : 4); }
  const bool emits_ = a->epi == DIA_EPI_RESID_EMIT || a->epi == DIA_EPI_SWIGLU_EMIT;
  const bool uni_f32 = k.a_f32 && (!emits_ || k.p_f32);          // fp32 tiles in, fp32 tiles out (or nothing emitted)
  if (a->M <= 4 && fast_epi && (!a->act_f32 || uni_f32)) {       // (a mixed-format call goes on to the generic kernel)
    const int rs = a->M <= 2 ? 2 : 4;
    if (small_smem(nw, a->KT / sk, rs) <= 150 * 1024) {
      bool handled = false;
      int rc;
      if (uni_f32) rc = (rs == 2) ? launch_small_rs<2, true>(k, nw, sk, st, handled) : launch_small_rs<4, true>(k, nw, sk, st, handled);
      else rc = (rs == 2) ? launch_small_rs<2>(k, nw, sk, st, handled) : launch_small_rs<4>(k, nw, sk, st, handled);
      if (handled) return rc;
    }
  }
  // registers hold the A fragments (12 * KPW VGPRs), so only short per-wave K ranges qualify; 8 waves x up to 8 k-tiles
  // measured slightly ahead of 16 x 4 (6.3 vs 6.7 us on qkv at 16 rows)
  const int ktl16 = a->KT / sk;
  int nw16 = a->nw ? a->nw : ((ktl16 % 8 == 0 && ktl16 / 8 <= 8) ? 8 : ((ktl16 % 16 == 0 && ktl16 / 16 <= 4) ? 16 : 0));
  if (mtiles == 1 && fast_epi && nw16) {
    bool handled = false;
    int rc = launch_g16_any(k, nw16, sk, st, handled);
    if (handled) return rc;
  }
  // 17..128 rows: the one-m-tile kernel over all m-tiles at once (gridDim.z = 2..8, the workgroups of a group share their
  // weight stream through L2).  Split-K (a->sk > 1) needs scratch for every m-tile: mtiles * nstrips * sk * 256 floats,
  // mtiles * nstrips tickets.
  const int mz_max = dia_tune(DIA_TUNE_GEMM_MZ_MAX) >= 0 ? dia_tune(DIA_TUNE_GEMM_MZ_MAX) : 8;
  // (the z-form's 32-thread tail runs the shared epilogue: the cross-K/V projections of a short prompt ride it too — the generic
  // kernel they fell to spills and took 21-25 us per launch at 98 rows)
  // two m-tiles per workgroup (k_gemm2t).  Decode (fp32 tiles on both sides, dense K = 2048 shapes): where a workgroup walks many strips —
  // measured (profiles/r03_gemm2t_ab.txt): wi at every row count (35 vs 52 us at 128 rows), the logits head from three m-tiles on, wo
  // (split-K 4, both tiles handed over together: 38 vs 58 us); the 128..192-strip projections run as fast in the z-form (more workgroups).
  // Short prompts (planes on both sides, 33..128 rows): every GEMM of the prefill incl. the cross-K/V projections.
  {
    const bool f32io = k.a_f32 && (!emits_ || k.p_f32);
    const bool planes = !k.a_f32 && !k.p_f32;
    const int ktw = a->KT / sk;                       // k-tiles per workgroup
    const bool shape = (ktw == 64 || (planes && ktw == 32)) && a->KT % sk == 0 && a->w_planes <= 1 && mtiles >= 2 && mtiles <= mz_max &&
                       (sk == 1 || a->sk_scratch_floats >= (int64_t)((mtiles + 1) / 2) * a->nstrips * sk * 512);
    const bool decode_ok = f32io && fast_epi && !a->cmap && !a->strip_map &&
                           (a->nstrips >= 1024 || (a->nstrips >= 512 && mtiles >= 2) || mtiles >= 7 || sk > 1 || dia_tune(DIA_TUNE_GEMM_2T) == 2);     // (from 7 m-tiles on the 128..192-strip projections gain as well: batch 64 +1.4 %)
    const bool prefill_ok = planes && mtiles >= 3 && (fast_epi || (a->epi == DIA_EPI_CROSSKV && ktw == 32)) && !(ktw == 32 && sk > 1);
    if (shape && (decode_ok || prefill_ok) && dia_tune(DIA_TUNE_GEMM_2T) != 0) {
      const int zp = (mtiles + 1) / 2;
      const bool half = !f32io && ktw == 32 && dia_tune(DIA_TUNE_GEMM_2T) != 3;     // K = 1024: 256-thread workgroups, two per CU (knob 3: the 512-thread form)
      int budget = half ? 512 : 256;
      if (!f32io && dia_tune(DIA_TUNE_G2T_WGS) > 0) budget = dia_tune(DIA_TUNE_G2T_WGS);
      int per = budget / zp / sk;
      per = per >= 8 ? per / 8 * 8 : (per > 0 ? per : 1);
      const int spw_ = (a->nstrips + per - 1) / per;                                              // strips per workgroup
      int gx = (a->nstrips + spw_ - 1) / spw_;
      if (gx % 8 != 0 && (gx + 7) / 8 * 8 <= a->nstrips) gx = (gx + 7) / 8 * 8;                   // the pairs of one strip group on one XCD
      const dim3 grid(gx, sk, zp), blk(half ? 256 : 512);
      // planes with a compile-time epilogue and the all-thread tail: dense shapes only (no compaction maps), knob 5 = the shared tail (A/B)
      const bool puni = !f32io && !a->cmap && !a->strip_map && (a->epi != DIA_EPI_RESID_EMIT || a->gnext) && dia_tune(DIA_TUNE_GEMM_2T) != 5;
      if (f32io) {
        const bool uni = dia_tune(DIA_TUNE_GEMM_2T) != 5;      // (knob 5: the run-time epilogue, A/B)
        if (sk > 1 && sk <= 4 && a->epi == DIA_EPI_RESID_EMIT && a->gnext && dia_tune(DIA_TUNE_GEMM_2T) != 6)     // (knob 6: one hand-off per strip, A/B)
          launch_small_kernel<k_gemm2t<8, true, true, 8, false, -1, true>>(grid, blk, g2t_smem(8, 8, true), st, k);
        else if (sk > 1) launch_small_kernel<k_gemm2t<8, true, true>>(grid, blk, g2t_smem(8), st, k);
        else if (uni && a->epi == DIA_EPI_SCALE_STORE) launch_small_kernel<k_gemm2t<8, true, false, 8, false, DIA_EPI_SCALE_STORE>>(grid, blk, g2t_smem(8), st, k);
        else if (uni && a->epi == DIA_EPI_RESID_EMIT) launch_small_kernel<k_gemm2t<8, true, false, 8, false, DIA_EPI_RESID_EMIT>>(grid, blk, g2t_smem(8), st, k);
        else if (uni && a->epi == DIA_EPI_SWIGLU_EMIT) launch_small_kernel<k_gemm2t<8, true, false, 8, false, DIA_EPI_SWIGLU_EMIT>>(grid, blk, g2t_smem(8), st, k);
        else launch_small_kernel<k_gemm2t<8, true, false>>(grid, blk, g2t_smem(8), st, k);
      } else if (ktw == 64) {
        if (sk > 1 && sk <= 4 && a->epi == DIA_EPI_RESID_EMIT && a->gnext && !a->cmap && dia_tune(DIA_TUNE_GEMM_2T) != 6)
          launch_small_kernel<k_gemm2t<8, false, true, 8, false, -1, true>>(grid, blk, g2t_smem(8, 8, true), st, k);
        else if (sk > 1) launch_small_kernel<k_gemm2t<8, false, true>>(grid, blk, g2t_smem(8), st, k);
        else if (puni && a->epi == DIA_EPI_RESID_EMIT) launch_small_kernel<k_gemm2t<8, false, false, 8, false, DIA_EPI_RESID_EMIT>>(grid, blk, g2t_smem(8), st, k);
        else launch_small_kernel<k_gemm2t<8, false, false>>(grid, blk, g2t_smem(8), st, k);
      } else if (half && a->epi == DIA_EPI_CROSSKV && a->kv_dtype == DIA_KV_BF16 && a->kv_vblocked && !a->strip_map && dia_tune(DIA_TUNE_GEMM_2T) != 4) {
        launch_small_kernel<k_gemm2t<8, false, false, 4, true>>(grid, blk, g2t_smem(8, 4), st, k);    // (knob 4: the shared tail, A/B)
      } else if (half && puni && a->epi == DIA_EPI_SCALE_STORE) {
        launch_small_kernel<k_gemm2t<8, false, false, 4, false, DIA_EPI_SCALE_STORE>>(grid, blk, g2t_smem(8, 4), st, k);
      } else if (half && puni && a->epi == DIA_EPI_SWIGLU_EMIT) {
        launch_small_kernel<k_gemm2t<8, false, false, 4, false, DIA_EPI_SWIGLU_EMIT>>(grid, blk, g2t_smem(8, 4), st, k);
      } else if (half) {
        launch_small_kernel<k_gemm2t<8, false, false, 4>>(grid, blk, g2t_smem(8, 4), st, k);
      } else {
        launch_small_kernel<k_gemm2t<4, false, false>>(grid, blk, g2t_smem(4), st, k);
      }
      return dia_check_launch("k_gemm2t");
    }
  }
  const bool z_epi = fast_epi || a->epi == DIA_EPI_CROSSKV;
  if (mtiles >= 2 && mtiles <= mz_max && z_epi && nw16 && (sk == 1 || a->sk_scratch_floats >= (int64_t)mtiles * a->nstrips * sk * 256)) {
    bool handled = false;
    k.mz = mtiles;
    int rc = launch_g16_any(k, nw16, sk, st, handled);
    k.mz = 0;
    if (handled) return rc;
  }
#ifdef DIA_EXPERIMENTS
  if (mtiles == 2 && !a->act_f32) {       // earlier forms for 17..32 rows (k_gemm_blk32, k_gemm32, k_gemm32m), selected by tuning knobs
    bool handled = false;
    int rc = dia_exp_gemm_two_mtiles(a, stream, handled);
    if (handled) return rc;
  }
#endif
  if (sk > 1) return dia_fail(DIA_E_ARG, "dia_gemm: no split-K kernel for this shape");
  // prefill shapes: the MFMA-tiled kernel (64 x 256 blocks) from 3 m-tiles on — only when the blocks fill a good part of
  // the chip: a lone short utterance is better off with the K-split kernel below
  {
    const int blocks = ((mtiles + GT_MT - 1) / GT_MT) * ((a->nstrips + 15) / 16);
    const int min_blocks = dia_tune(DIA_TUNE_TILE_MIN_BLOCKS) >= 0 ? dia_tune(DIA_TUNE_TILE_MIN_BLOCKS) : 48;
    if (mtiles >= 3 && a->KT % 8 == 0 && blocks >= min_blocks && !a->act_f32) {      // (the tiled kernel stages planes)
#ifdef DIA_EXPERIMENTS
      const int v = dia_tune(DIA_TUNE_TILE_V);
      if (v >= 0 && v <= 2) return dia_exp_tile_variant(a, stream, v);
#endif
      return launch_tile(k, st);
    }
  }
  if (mtiles == 1) return launch_nw<1>(k, nw, 1, st);
  if (mtiles == 2) return launch_nw<2>(k, nw, 1, st);
  return launch_nw<4>(k, nw, (mtiles + 3) / 4, st);
}

extern "C" int dia_gemm_timed(const dia_gemm_args* a, void* stream, float* ms_out) {
  if (!ms_out) return dia_fail(DIA_E_ARG, "dia_gemm_timed: null output");
  dia_recorder_arm();
  int rc = dia_gemm(a, stream);
  float ms[4] = {0.f, 0.f, 0.f, 0.f};
  const int n = dia_recorder_collect(ms, 4);
  if (rc != DIA_OK) return rc;
  if (n < 1) return n < 0 ? n : dia_fail(DIA_E_STATE, "dia_gemm_timed: nothing was launched");
  *ms_out = ms[0];
  return DIA_OK;
}

#ifndef DIA_EXPERIMENTS
extern "C" int dia_mlp_fused(const dia_gemm_args*, const dia_gemm_args*, int32_t*, void*) {
  return dia_fail(DIA_E_ARG, "dia_mlp_fused: experiment, needs a build with EXPERIMENTS=1");
}
extern "C" int dia_mlp_fused_timed(const dia_gemm_args*, const dia_gemm_args*, int32_t*, void*, float*) {
  return dia_fail(DIA_E_ARG, "dia_mlp_fused_timed: experiment, needs a build with EXPERIMENTS=1");
}
#endif
