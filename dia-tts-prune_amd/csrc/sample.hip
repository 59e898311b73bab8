// Per-step token selection for the 9 codebook channels of every utterance, on the device:
//   CFG combine + EOS/PAD/BOS constraints       (reference dia/model.py:447-478)
//   temperature, top-k, top-p, multinomial      (reference dia/model.py:32-82)
//   EOS countdown / delay-pattern state machine (reference dia/model.py:771-807)
//   masked token write                          (reference dia/state.py:195-203)
//   next-step input embedding (9 tables summed) (reference dia/layers.py:691-696)
// so that the decode loop needs no host synchronisation (the reference syncs at model.py:76,773,786).
//
// One workgroup per utterance, one wave per channel: every lane keeps 17 of the 1028 logits in
// registers; reductions are wave shuffles / ballots, nothing crosses waves until the state machine.
// torch.multinomial(p, 1) on the reference's CPU path equals argmax(p / q), q ~ Exp(1) drawn from the
// same generator, so the host uploads q and the kernel takes the argmax: identical
// token ids for identical logits.
//
// fp contraction is OFF in this file: the reference evaluates every elementwise op with its own
// rounding, and the top-k / top-p cut-offs compare those values.
#pragma clang fp contract(off)
#include "common.hpp"
#include "../../include/dia_hip.h"
#include "errors.hpp"
#include "launch.hpp"

namespace {

constexpr int NV = 17;                 // 64 * 17 = 1088 >= 1028
constexpr int VCAP = NV * 64;
constexpr int MAXC = 16;

struct EmbedK {
  const int* tokens; const int* cur; int B, T, C, V, D;
  const float* emb; const float* g; float* x;
  bf16_raw* P; long p_plane_stride; int p_ktiles; int ssq_ld; float* ssq;
  const int* cmap;
  int act_f32;
};

struct SampleK {
  const float* logits; int ld_logits; int B, T, C, V, max_tokens;
  float cfg_scale, temperature, top_p; int top_k;
  int eos, pad, bos, max_delay, ignore_eos, teacher;
  const int* delay; const float* noise; int noise_steps;
  const int* first_step;
  int* tokens; int* pred; int* cur; int* fsm;
  EmbedK e;
};

// x rows (2b, 2b+1) <- sum_c emb[c][tok[c]][:]; planes(x*g); strip ssq.  8 dims per thread.
__device__ __forceinline__ void embed_rows(const EmbedK& e, int b, const int* tok, int tid, int nthreads) {
  for (int d0 = tid * 8; d0 < e.D; d0 += nthreads * 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = 0.f;
    if (e.C == 9) {
      float4 ra[9], rb[9];
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        const float* r = e.emb + ((long)c * e.V + tok[c]) * e.D + d0;
        ra[c] = *reinterpret_cast<const float4*>(r);
        rb[c] = *reinterpret_cast<const float4*>(r + 4);
      }
      v[0] = ra[0].x; v[1] = ra[0].y; v[2] = ra[0].z; v[3] = ra[0].w; v[4] = rb[0].x; v[5] = rb[0].y; v[6] = rb[0].z; v[7] = rb[0].w;
#pragma unroll
      for (int c = 1; c < 9; ++c) {   // sequential sum in channel order (layers.py:696)
        v[0] += ra[c].x; v[1] += ra[c].y; v[2] += ra[c].z; v[3] += ra[c].w;
        v[4] += rb[c].x; v[5] += rb[c].y; v[6] += rb[c].z; v[7] += rb[c].w;
      }
    } else
    for (int c = 0; c < e.C; ++c) {
      const float* r = e.emb + ((long)c * e.V + tok[c]) * e.D + d0;
      const float4 a = *reinterpret_cast<const float4*>(r), bq = *reinterpret_cast<const float4*>(r + 4);
      if (c == 0) { v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w; }
      else { v[0] += a.x; v[1] += a.y; v[2] += a.z; v[3] += a.w; v[4] += bq.x; v[5] += bq.y; v[6] += bq.z; v[7] += bq.w; }
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += v[j] * v[j];
    const float other = __shfl_xor(ss, 1, 64);
    float vg[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) vg[j] = e.g ? v[j] * e.g[d0 + j] : v[j];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int m = 2 * b + r;
      float* xo = e.x + (long)m * e.D + d0;
      *reinterpret_cast<float4*>(xo) = float4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<float4*>(xo + 4) = float4{v[4], v[5], v[6], v[7]};
      if (e.act_f32) {
        if (e.cmap) emit_f32x8_mapped(reinterpret_cast<float*>(e.P), e.p_ktiles, m, d0, vg, e.cmap);
        else emit_f32x8(reinterpret_cast<float*>(e.P), e.p_ktiles, m, d0, vg);
      } else if (e.cmap) emit_planes8_mapped(e.P, e.p_plane_stride, e.p_ktiles, m, d0, vg, e.cmap);
      else emit_planes8(e.P, e.p_plane_stride, e.p_ktiles, m, d0, vg);
      if (((d0 >> 3) & 1) == 0) e.ssq[(long)(d0 >> 4) * e.ssq_ld + m] = ss + other;
    }
  }
}

__global__ __launch_bounds__(256) void k_embed_tokens(EmbedK e) {
  __shared__ int tok[MAXC];
  const int b = blockIdx.x;
  if (threadIdx.x < e.C) tok[threadIdx.x] = e.tokens[((long)b * e.T + (e.cur[b] - 1)) * e.C + threadIdx.x];
  __syncthreads();
  embed_rows(e, b, tok, threadIdx.x, 256);
}

__global__ __launch_bounds__(256) void k_embed_text(const int* ids, int L, const float* table, int D, const float* g,
                                                    float* x, bf16_raw* P, long p_plane_stride, int p_ktiles,
                                                    float* ssq, int ssq_ld, const int* cmap) {
  const int m = blockIdx.x;
  const float* row = table + (long)ids[m] * D;
  for (int d0 = threadIdx.x * 8; d0 < D; d0 += 256 * 8) {
    float v[8], vg[8];
    const float4 a = *reinterpret_cast<const float4*>(row + d0), bq = *reinterpret_cast<const float4*>(row + d0 + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = bq.x; v[5] = bq.y; v[6] = bq.z; v[7] = bq.w;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { ss += v[j] * v[j]; vg[j] = g ? v[j] * g[d0 + j] : v[j]; }
    const float other = __shfl_xor(ss, 1, 64);
    float* xo = x + (long)m * D + d0;
    *reinterpret_cast<float4*>(xo) = a;
    *reinterpret_cast<float4*>(xo + 4) = bq;
    if (cmap) emit_planes8_mapped(P, p_plane_stride, p_ktiles, m, d0, vg, cmap);
    else emit_planes8(P, p_plane_stride, p_ktiles, m, d0, vg);
    if (((d0 >> 3) & 1) == 0) ssq[(long)(d0 >> 4) * ssq_ld + m] = ss + other;
  }
}

__device__ __forceinline__ uint32_t fkey(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// lane-local (value, index) -> wave argmax, first index wins ties
__device__ __forceinline__ int wave_argmax(float bv, int bi) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  return bi;
}

#ifdef DIA_DBG_STAMPS
__device__ long long g_sstamps[16];
#define SSTAMP(i) do { if (threadIdx.x == 64) g_sstamps[i] = wall_clock64(); } while (0)
#else
#define SSTAMP(i) do {} while (0)
#endif

// ---- wave-wide reductions on DPP row operations + four v_readlane (a ds_bpermute shuffle costs an LDS round trip per step,
// and this kernel is one dependent chain of them: 60 shuffles were 3 of its 6.7 us top-p phase)
#define DIA_DPPI(x, ctrl) __builtin_amdgcn_update_dpp(0, (x), (ctrl), 0xF, 0xF, true)
#define DIA_DPPF(x, ctrl) __builtin_bit_cast(float, DIA_DPPI(__builtin_bit_cast(int, (x)), (ctrl)))
__device__ __forceinline__ float lane_f(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wsum(float v) {          // same value in every lane: (r0 + r1) + (r2 + r3) of the four row sums
  v = row16_sum(v);
  return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}
__device__ __forceinline__ float wmax(float v) {
  v = fmaxf(v, DIA_DPPF(v, 0xB1));
  v = fmaxf(v, DIA_DPPF(v, 0x4E));
  v = fmaxf(v, DIA_DPPF(v, 0x141));
  v = fmaxf(v, DIA_DPPF(v, 0x140));
  return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}
// (value, index) -> index of the largest value, smallest index among equals
__device__ __forceinline__ int wargmax(float bv, int bi) {
#define DIA_AM_STEP(ctrl) { const float ov = DIA_DPPF(bv, ctrl); const int oi = DIA_DPPI(bi, ctrl); \
                            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; } }
  DIA_AM_STEP(0xB1) DIA_AM_STEP(0x4E) DIA_AM_STEP(0x141) DIA_AM_STEP(0x140)
#undef DIA_AM_STEP
  float v = lane_f(bv, 0); int ix = __builtin_amdgcn_readlane(bi, 0);
#pragma unroll
  for (int r = 16; r < 64; r += 16) {
    const float ov = lane_f(bv, r); const int oi = __builtin_amdgcn_readlane(bi, r);
    if (ov > v || (ov == v && oi < ix)) { v = ov; ix = oi; }
  }
  return ix;
}
// inclusive prefix sum over the 64 lanes in double: Hillis-Steele inside each row of 16 on row_shr (lanes shifted in from
// outside the row read 0), then the totals of the rows below
__device__ __forceinline__ double wscan_incl(double x, int lane) {
#define DIA_SC_STEP(n) { const int lo = DIA_DPPI(__double2loint(x), 0x110 + (n)), hi = DIA_DPPI(__double2hiint(x), 0x110 + (n)); \
                         x += __hiloint2double(hi, lo); }
  DIA_SC_STEP(1) DIA_SC_STEP(2) DIA_SC_STEP(4) DIA_SC_STEP(8)
#undef DIA_SC_STEP
  const double t0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 15), __builtin_amdgcn_readlane(__double2loint(x), 15));
  const double t1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 31), __builtin_amdgcn_readlane(__double2loint(x), 31));
  const double t2 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 47), __builtin_amdgcn_readlane(__double2loint(x), 47));
  const int row = lane >> 4;
  const double below = row == 0 ? 0.0 : row == 1 ? t0 : row == 2 ? t0 + t1 : (t0 + t1) + t2;
  return x + below;
}

__global__ __launch_bounds__(MAXC * 64) void k_sample(SampleK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ int preds[MAXC];
  __shared__ int tok_next[MAXC];
  __shared__ int go_next;

  const int tid = threadIdx.x, lane = tid & 63, c = tid >> 6;
  const int b = blockIdx.x;
  SSTAMP(0);
  // the logits of this wave's channel do not depend on any device-side state: requested before cur[] / fsm[] are read,
  // on clamped indices (no per-element branches: a conditional load makes hipcc wait for it on the spot)
  const int cc = min(c, p.C - 1);
  const float* un = p.logits + (long)(2 * b) * p.ld_logits + cc * p.V;
  const float* co = p.logits + (long)(2 * b + 1) * p.ld_logits + cc * p.V;
  float cv[NV], uv[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int v = min(lane + 64 * i, p.V - 1);
    cv[i] = co[v]; uv[i] = un[v];
  }
  const int cur = p.cur[b];
  int* fsm = p.fsm + b * 8;
  const bool done = fsm[3] != 0;                    // uniform over the workgroup
  const int first = p.first_step ? p.first_step[b] : 1;
  const bool replay = cur < first;                  // audio-prompt rows: nothing sampled, nothing written
  // what the state machine reads after the barrier, requested now (every wave: no branch around loads)
  const int chl = min(lane, p.C - 1);
  const int fsm_old = p.tokens[((long)b * p.T + cur) * p.C + chl];
  const int fsm_d = p.delay[chl];
  const int fsm0 = fsm[0], fsm1 = fsm[1], fsm2 = fsm[2];

  // per-wave LDS scratch for the top-p ordering
  float* lp = reinterpret_cast<float*>(smem_raw) + (size_t)c * (2 * VCAP + VCAP);  // [VCAP] probs of survivors
  float* sp = lp + VCAP;                                                            // [VCAP] sorted probs
  unsigned short* li = reinterpret_cast<unsigned short*>(sp + VCAP);                // [VCAP] survivor indices
  unsigned short* rk = li + VCAP;                                                   // [VCAP] rank by vocab index

  float warm = 0.f;                                 // see "next-step embedding rows" below
  if (!done && !replay && c < p.C) {
    const int n = cur - first;                                                      // sampled steps so far
    float lg[NV], qn[NV];
    const bool use_noise = p.temperature != 0.0f;
    const float* q = use_noise ? p.noise + (((long)b * p.noise_steps + n) * p.C + c) * p.V : co;
#pragma unroll
    for (int i = 0; i < NV; ++i) qn[i] = q[min(lane + 64 * i, p.V - 1)];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = lane + 64 * i;
      const float t = cv[i] - uv[i];
      float x = cv[i] + p.cfg_scale * t;                                            // model.py:457
      if (v >= p.V || v == p.pad || v == p.bos || (c > 0 && v == p.eos)) x = -INFINITY;   // model.py:462-472
      lg[i] = x;
      if (!use_noise) qn[i] = 1.0f;
    }
    int choice;
    if (p.temperature == 0.0f) {                                                    // model.py:38-40
      float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = lane + 64 * i;
        if (v < p.V && (bi == 0x7fffffff || lg[i] > bv)) { bv = lg[i]; bi = v; }
      }
      choice = wargmax(bv, bi);
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) lg[i] = lg[i] / p.temperature;                   // model.py:43
      SSTAMP(1);
      // ---- top-k (model.py:46-52): everything below the k-th largest value goes to -inf.
      // k <= 64 (the reference's default is 35): the k-th largest of the 64 per-lane maxima is a lower bound T0 of the
      // k-th largest value overall (k lanes hold something >= T0), and about 47 of the 1028 values reach it.  With at
      // most 64 candidates they are compacted to one per lane, in increasing vocabulary index, and the cut is a count of
      // larger candidates.  More candidates than lanes, k > 64 or no top-k: the k-th value by bitwise search on
      // order-preserving keys, and the survivors are compacted if they fit.
      const int k = min(p.top_k, p.V);
      bool compact = false, kcut = false;
      int cnt = 0;
      auto compact_if = [&](auto pred) {               // survivors of pred(lg[i]) -> lp (value), sp (noise), li (index)
        int base = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          const bool in = pred(lg[i]);
          const unsigned long long mk = __ballot(in);
          if (in) {
            const int pos = base + __popcll(mk & ((1ull << lane) - 1ull));
            lp[pos] = lg[i]; sp[pos] = qn[i]; li[pos] = (unsigned short)(lane + 64 * i);
          }
          base += __popcll(mk);
        }
        __builtin_amdgcn_wave_barrier();
      };
      if (p.top_k > 0 && k <= 64) {
        float lm = lg[0];
#pragma unroll
        for (int i = 1; i < NV; ++i) lm = fmaxf(lm, lg[i]);
        int above = 0;                                 // lanes whose maximum is larger than this lane's
#pragma unroll 8
        for (int j = 0; j < 64; ++j) above += (lane_f(lm, j) > lm) ? 1 : 0;
        const float t0 = -wmax(above < k ? -lm : -INFINITY);     // smallest of the k largest lane maxima (ties included)
#pragma unroll
        for (int i = 0; i < NV; ++i) cnt += __popcll(__ballot(lg[i] >= t0));
        if (t0 != -INFINITY && cnt <= 64) {
          compact_if([&](float x) { return x >= t0; });
          compact = true; kcut = true;
        }
      }
      if (!compact) {
        if (p.top_k > 0) {
          uint32_t key[NV];
#pragma unroll
          for (int i = 0; i < NV; ++i) key[i] = (lane + 64 * i < p.V) ? fkey(lg[i]) : 0u;
          uint32_t pre = 0;
          for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = pre | (1u << bit);
            int cn = 0;
#pragma unroll
            for (int i = 0; i < NV; ++i) cn += __popcll(__ballot(key[i] >= cand));
            if (cn >= k) pre = cand;
            if (cn == k) break;          // exactly the k largest already separated: same mask as the full search
          }
#pragma unroll
          for (int i = 0; i < NV; ++i) if (key[i] < pre) lg[i] = -INFINITY;
        }
        cnt = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) cnt += __popcll(__ballot(lg[i] != -INFINITY));
        if (cnt >= 1 && cnt <= 64) {
          compact_if([&](float x) { return x != -INFINITY; });
          compact = true;
        }
      }
      SSTAMP(2);
      if (compact) {
        // ---- one candidate per lane: softmax, top-p ordering, final softmax and the draw on single registers; sums run
        // over the compacted lanes in the DPP tree's order
        float* srt = lp + 256;                         // [64] probabilities in sorted order
        const bool act0 = lane < cnt;
        float l = act0 ? lp[lane] : -INFINITY;
        const float q1 = act0 ? sp[lane] : 1.0f;
        const int idx = act0 ? (int)li[lane] : 0x7fff;
        bool act = act0;
        if (kcut) {                                    // keep what fewer than k candidates exceed: everything >= the k-th value
          int gt = 0;
          for (int j = 0; j < cnt; ++j) gt += (lane_f(l, j) > l) ? 1 : 0;
          act = act0 && gt < k;
          if (!act) l = -INFINITY;
        }
        if (p.top_p < 1.0f) {                          // model.py:56-70
          const float m = wmax(l);
          const float e = act ? expf(l - m) : 0.f;
          const float z = wsum(e);
          const float pr = e / z;
          const bool actp = act && pr > 0.f;           // (a survivor whose probability underflows sorts after everything)
          const int ns = __popcll(__ballot(actp));
          int r = 0, eq = 0;                           // rank by descending probability; equal ones by index, if there are any
          for (int j = 0; j < cnt; ++j) {
            const float pj = lane_f(pr, j);
            r += (pj > pr) ? 1 : 0;
            eq += (pj == pr) ? 1 : 0;
          }
          if (__ballot(actp && eq > 1)) {
            for (int j = 0; j < cnt; ++j) {
              const float pj = lane_f(pr, j);
              const int ij = __builtin_amdgcn_readlane(idx, j);
              r += (pj == pr && ij < idx) ? 1 : 0;
            }
          }
          if (actp) srt[r] = pr;
          __builtin_amdgcn_wave_barrier();
          // cumulative sum in sorted order, in double like torch.cumsum on CPU — in double the scan order moves the
          // sum by ~1e-16 relative, far below the fp32 rounding applied before the comparison
          const double cum = wscan_incl((lane < ns) ? (double)srt[lane] : 0.0, lane);
          const unsigned long long over = __ballot(lane < ns && (float)cum > p.top_p);
          const int first_over = over ? (__ffsll((long long)over) - 1) : ns;      // smallest r with cum[r] > top_p
          int keep = min(ns, first_over + 1);          // entry r+1 is removed iff float(cum[r]) > top_p; monotone in r
          if (ns == 0) keep = 0;
          if (!(actp && r < keep)) l = -INFINITY;
          __builtin_amdgcn_wave_barrier();
        }
        SSTAMP(3);
        // ---- final softmax + multinomial as argmax(p / q) (model.py:73-82)
        const float m2 = wmax(l);
        const float e2 = (l == -INFINITY) ? 0.f : expf(l - m2);
        const float z2 = wsum(e2);
        const float sc = act ? (e2 / z2) / q1 : -1.f;
        choice = wargmax(sc, act ? idx : 0x7fffffff);
      } else {
      // ---- top-p (model.py:56-70)
      if (p.top_p < 1.0f) {
        float m = -INFINITY;
#pragma unroll
        for (int i = 0; i < NV; ++i) m = fmaxf(m, lg[i]);
        m = wave_max(m);
        float e[NV], z = 0.f;
#pragma unroll
        for (int i = 0; i < NV; ++i) { e[i] = expf(lg[i] - m); z += e[i]; }
        z = wave_sum(z);
        int ns = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          e[i] = e[i] / z;
          const unsigned long long mk = __ballot(e[i] > 0.f);
          if (e[i] > 0.f) {
            const int pos = ns + __popcll(mk & ((1ull << lane) - 1ull));
            lp[pos] = e[i];
            li[pos] = (unsigned short)(lane + 64 * i);
          }
          ns += __popcll(mk);
        }
        __builtin_amdgcn_wave_barrier();
        // rank = number of survivors ordered before this one (descending prob, then index)
        for (int s = lane; s < ns; s += 64) {
          const float ps = lp[s];
          const int is = li[s];
          int r = 0;
          for (int j = 0; j < ns; ++j) {
            const float pj = lp[j];
            r += (pj > ps || (pj == ps && (int)li[j] < is)) ? 1 : 0;
          }
          sp[r] = ps;
          rk[is] = (unsigned short)r;
        }
        __builtin_amdgcn_wave_barrier();
        // cumulative sum in sorted order, accumulated in double like torch.cumsum on CPU.  Up to 64
        // survivors (the usual case after top-k): a wave inclusive scan — in double the scan order moves the
        // sum by ~1e-16 relative, far below the fp32 rounding applied before the comparison.
        int keep = ns;
        if (ns <= 64) {
          double cum = (lane < ns) ? (double)sp[lane] : 0.0;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(cum, o, 64);
            if (lane >= o) cum += up;
          }
          // entry r+1 is removed iff float(cum[r]) > top_p; removal is monotone in r
          const unsigned long long over = __ballot(lane < ns && (float)cum > p.top_p);
          const int first = over ? (__ffsll((long long)over) - 1) : ns;          // smallest r with cum[r] > top_p
          keep = min(ns, first + 1);
          if (ns == 0) keep = 0;
        } else if (lane == 0) {
          double cum = 0.0;
          keep = 1;
          for (int r = 0; r + 1 < ns; ++r) {
            cum += (double)sp[r];
            if ((float)cum > p.top_p) break;          // entry r+1 is removed, and all after it
            keep = r + 2;
          }
        }
        keep = __shfl(keep, 0, 64);
#pragma unroll
        for (int i = 0; i < NV; ++i) {
          if (e[i] > 0.f) { if ((int)rk[lane + 64 * i] >= keep) lg[i] = -INFINITY; }
          else lg[i] = -INFINITY;                       // sorts after every survivor; cum there > top_p
        }
      }
      SSTAMP(3);
      // ---- final softmax + multinomial as argmax(p / q) (model.py:73-82)
      float m2 = -INFINITY;
#pragma unroll
      for (int i = 0; i < NV; ++i) m2 = fmaxf(m2, lg[i]);
      m2 = wave_max(m2);
      float e2[NV], z2 = 0.f;
#pragma unroll
      for (int i = 0; i < NV; ++i) { e2[i] = expf(lg[i] - m2); z2 += e2[i]; }
      z2 = wave_sum(z2);
      float bv = -1.f; int bi = 0x7fffffff;
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        const int v = lane + 64 * i;
        if (v < p.V) {
          const float sc = (e2[i] / z2) / qn[i];
          if (sc > bv) { bv = sc; bi = v; }
        }
      }
      choice = wave_argmax(bv, bi);
    }
      }
    SSTAMP(4);
    if (lane == 0) preds[c] = choice;
    // next-step embedding rows: outside the EOS countdown and the BOS window the token written is the one just drawn, so
    // this wave pulls its channel's row (one dword per 64-byte line) towards the L2 of this XCD while the other waves
    // finish and the state machine runs; embed_rows() below then finds it there instead of in HBM
    // (all-NaN scores leave the argmax sentinel 0x7fffffff in `choice`: no row to warm then, and no read outside the table)
    if ((unsigned)choice < (unsigned)p.e.V) {
      const float* row = p.e.emb + ((long)c * p.e.V + choice) * p.e.D;
      for (int o = lane * 16; o < p.e.D; o += 64 * 16) warm += row[o];
    }
  }
  __syncthreads();

  SSTAMP(5);
  // ---- token state machine: wave 0, lane i = channel i (all global accesses issued in parallel) ----
  if (c == 0) {
    int go = 0;
    if (!done) {
      const bool ch = lane < p.C;
      int* prow = p.pred + ((long)b * p.T + cur) * p.C;
      int* trow = p.tokens + ((long)b * p.T + cur) * p.C;
      int eos_detected = fsm0, eos_countdown = fsm1, bos_countdown = fsm2;            // read before the sampling started
      const int old = ch ? fsm_old : 0;
      const int d = ch ? fsm_d : 0;
      int pr = (ch && !replay) ? preds[lane] : -1;
      if (ch && !replay) prow[lane] = pr;
      const int pr0 = __shfl(pr, 0, 64);
      int finished = 0, last = cur, tk = old;
      if (!p.teacher && !replay) {
        if (!eos_detected && pr0 == p.eos && !p.ignore_eos) { eos_detected = 1; eos_countdown = p.max_delay; }
        if (eos_countdown > 0) {
          const int after = p.max_delay - eos_countdown;
          if (after == d) pr = p.eos;
          else if (after > d && pr != p.eos) pr = p.pad;
          eos_countdown -= 1;
        }
        bos_countdown = max(0, bos_countdown - 1);
        tk = (bos_countdown > 0 && old != -1) ? old : pr;
        if (ch) trow[lane] = tk;
        if (eos_countdown == 0) { finished = 1; last = cur - 1; }                     // model.py:795-797 (break)
        else if (cur >= p.max_tokens - p.max_delay - 1 && !eos_detected) { eos_detected = 1; eos_countdown = p.max_delay; }
      }
      // (the row that is embedded next stays inside the table even when all-NaN scores left the argmax sentinel in the token)
      if (ch) tok_next[lane] = min(max(tk, 0), p.e.V - 1);
      if (!finished) {
        last = cur;                                                                    // dec_step += 1
        if (cur + 1 > p.max_tokens - 1) finished = 1;                                  // while dec_step < max_tokens-1
        else go = 1;
      }
      if (lane == 0) {
        if (go) p.cur[b] = cur + 1;
        fsm[0] = eos_detected; fsm[1] = eos_countdown; fsm[2] = bos_countdown; fsm[3] = finished; fsm[4] = last;
      }
    }
    if (lane == 0) go_next = go && !(warm == 1.2345e38f);      // (keeps the warm-up loads alive; never true)
  }
  __syncthreads();
  SSTAMP(6);
  if (go_next) embed_rows(p.e, b, tok_next, tid, blockDim.x);
  SSTAMP(7);
}

}  // namespace

static int fill_embed(const dia_embed_args* a, EmbedK& e) {
  if (!a->emb || !a->x || !a->P || !a->ssq) return dia_fail(DIA_E_ARG, "embed: null argument");
  if (a->D % 16 != 0 || a->C > MAXC || a->C <= 0) return dia_fail(DIA_E_ARG, "embed: D must be a multiple of 16 and C <= 16");
  if (a->p_plane_stride % 8 != 0 || a->p_ktiles * 32 < a->D) return dia_fail(DIA_E_ARG, "embed: plane layout too narrow");
  e.tokens = a->tokens; e.cur = a->cur; e.B = a->B; e.T = a->T; e.C = a->C; e.V = a->V; e.D = a->D;
  e.emb = a->emb; e.g = a->g; e.x = a->x; e.P = (bf16_raw*)a->P; e.p_plane_stride = a->p_plane_stride;
  e.p_ktiles = a->p_ktiles; e.ssq_ld = a->ssq_ld; e.ssq = a->ssq; e.cmap = a->cmap; e.act_f32 = a->act_f32;
  return DIA_OK;
}

extern "C" int dia_embed_tokens(const dia_embed_args* a, void* stream) {
  if (!a || !a->tokens || !a->cur || a->B <= 0) return dia_fail(DIA_E_ARG, "dia_embed_tokens: null argument");
  EmbedK e;
  int rc = fill_embed(a, e);
  if (rc) return rc;
  dia_launch<k_embed_tokens>(dim3(a->B), dim3(256), 0, (hipStream_t)stream, e);
  return dia_check_launch("k_embed_tokens");
}

extern "C" int dia_embed_text(const int32_t* ids, int L, const float* table, int D, const float* g, float* x,
                              void* P, int64_t p_plane_stride, int p_ktiles, float* ssq, int ssq_ld, const int32_t* cmap,
                              void* stream) {
  if (!ids || !table || !x || !P || !ssq || L <= 0 || D % 16 != 0 || (!cmap && p_ktiles * 32 < D) || p_plane_stride % 8 != 0)
    return dia_fail(DIA_E_ARG, "dia_embed_text: bad argument");
  dia_launch<k_embed_text>(dim3(L), dim3(256), 0, (hipStream_t)stream, ids, L, table, D, g, x,
                     (bf16_raw*)P, (long)p_plane_stride, p_ktiles, ssq, ssq_ld, cmap);
  return dia_check_launch("k_embed_text");
}

#ifdef DIA_DBG_STAMPS
extern "C" int dia_dbg_sstamps(long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_sstamps), sizeof(long long) * 16) == hipSuccess ? 0 : -2;
}
#endif

int dia_sample_init() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_sample), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
  if (e != hipSuccess) return dia_fail_hip(e, "hipFuncSetAttribute(k_sample)");
  return DIA_OK;
}

extern "C" int dia_sample(const dia_sample_args* a, void* stream) {
  if (!a || !a->logits || !a->tokens || !a->pred || !a->cur || !a->fsm || !a->delay) return dia_fail(DIA_E_ARG, "dia_sample: null argument");
  if (a->V > VCAP || a->C > MAXC || a->C <= 0 || a->B <= 0) return dia_fail(DIA_E_ARG, "dia_sample: vocabulary > 1088 or channels > 16");
  if (a->temperature != 0.0f && !a->noise) return dia_fail(DIA_E_ARG, "dia_sample: temperature > 0 needs the Exp(1) noise buffer");
  if (a->temperature != 0.0f && a->noise_steps < a->max_tokens - 1) return dia_fail(DIA_E_ARG, "dia_sample: noise buffer shorter than max_tokens-1 steps");
  if (a->max_tokens > a->T || a->max_tokens < 2) return dia_fail(DIA_E_ARG, "dia_sample: max_tokens out of range");
  SampleK k;
  k.logits = a->logits; k.ld_logits = a->ld_logits; k.B = a->B; k.T = a->T; k.C = a->C; k.V = a->V; k.max_tokens = a->max_tokens;
  k.cfg_scale = a->cfg_scale; k.temperature = a->temperature; k.top_p = a->top_p; k.top_k = a->top_k;
  k.eos = a->eos; k.pad = a->pad; k.bos = a->bos; k.max_delay = a->max_delay; k.ignore_eos = a->ignore_eos; k.teacher = a->teacher;
  k.delay = a->delay; k.noise = a->noise; k.noise_steps = a->noise_steps; k.first_step = a->first_step;
  k.tokens = a->tokens; k.pred = a->pred; k.cur = a->cur; k.fsm = a->fsm;
  dia_embed_args ea = a->embed;
  ea.tokens = a->tokens; ea.cur = a->cur; ea.B = a->B; ea.T = a->T; ea.C = a->C; ea.V = a->V;
  int rc = fill_embed(&ea, k.e);
  if (rc) return rc;
  const size_t smem = (size_t)a->C * (3 * VCAP) * sizeof(float);
  if (smem > 64 * 1024) {
    rc = dia_kernels_init_once();
    if (rc) return rc;
  }
  dia_launch<k_sample>(dim3(a->B), dim3(a->C * 64), smem, (hipStream_t)stream, k);
  return dia_check_launch("k_sample");
}
