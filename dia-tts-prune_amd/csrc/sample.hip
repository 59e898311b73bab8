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
      if (e.cmap) emit_planes8_mapped(e.P, e.p_plane_stride, e.p_ktiles, m, d0, vg, e.cmap);
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

__global__ __launch_bounds__(MAXC * 64) void k_sample(SampleK p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ int preds[MAXC];
  __shared__ int tok_next[MAXC];
  __shared__ int go_next;

  const int tid = threadIdx.x, lane = tid & 63, c = tid >> 6;
  const int b = blockIdx.x;
  const int cur = p.cur[b];
  int* fsm = p.fsm + b * 8;
  const bool done = fsm[3] != 0;                    // uniform over the workgroup
  const int first = p.first_step ? p.first_step[b] : 1;
  const bool replay = cur < first;                  // audio-prompt rows: nothing sampled, nothing written

  // per-wave LDS scratch for the top-p ordering
  float* lp = reinterpret_cast<float*>(smem_raw) + (size_t)c * (2 * VCAP + VCAP);  // [VCAP] probs of survivors
  float* sp = lp + VCAP;                                                            // [VCAP] sorted probs
  unsigned short* li = reinterpret_cast<unsigned short*>(sp + VCAP);                // [VCAP] survivor indices
  unsigned short* rk = li + VCAP;                                                   // [VCAP] rank by vocab index

  SSTAMP(0);
  if (!done && !replay && c < p.C) {
    const int n = cur - first;                                                      // sampled steps so far
    const float* un = p.logits + (long)(2 * b) * p.ld_logits + c * p.V;
    const float* co = p.logits + (long)(2 * b + 1) * p.ld_logits + c * p.V;
    float lg[NV], qn[NV];
    // every load of this wave is issued up front on clamped indices (no per-element branches: a
    // conditional load makes hipcc wait for it on the spot), selections happen afterwards
    const bool use_noise = p.temperature != 0.0f;
    const float* q = use_noise ? p.noise + (((long)b * p.noise_steps + n) * p.C + c) * p.V : co;
    float cv[NV], uv[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int v = min(lane + 64 * i, p.V - 1);
      cv[i] = co[v]; uv[i] = un[v]; qn[i] = q[v];
    }
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
      choice = wave_argmax(bv, bi);
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i) lg[i] = lg[i] / p.temperature;                   // model.py:43
      SSTAMP(1);
      // ---- top-k: k-th largest value by bitwise search on order-preserving keys (model.py:46-52)
      if (p.top_k > 0) {
        const int k = min(p.top_k, p.V);
        uint32_t key[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) key[i] = (lane + 64 * i < p.V) ? fkey(lg[i]) : 0u;
        uint32_t pre = 0;
        for (int bit = 31; bit >= 0; --bit) {
          const uint32_t cand = pre | (1u << bit);
          int cnt = 0;
#pragma unroll
          for (int i = 0; i < NV; ++i) cnt += __popcll(__ballot(key[i] >= cand));
          if (cnt >= k) pre = cand;
          if (cnt == k) break;          // exactly the k largest already separated: same mask as the full search
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) if (key[i] < pre) lg[i] = -INFINITY;
      }
      SSTAMP(2);
      // ---- survivors of the top-k cut.  With at most 64 of them (k = 35 plus ties: the usual case) everything that
      // follows — softmax, top-p ordering, final softmax, multinomial — runs on ONE element per lane instead of 17
      // registers of mostly -inf.  The arithmetic is the general path's, bit for bit: the softmax denominators are
      // accumulated per ORIGINAL lane in index order and then summed by the same butterfly (lanep[] below).
      int nsv = 0;
#pragma unroll
      for (int i = 0; i < NV; ++i) nsv += __popcll(__ballot(lg[i] != -INFINITY));
      if (nsv >= 1 && nsv <= 64) {
        float* lanep = lp + 128;                       // [64] per-original-lane partial sums
        float* srt = lp + 256;                         // [64] probabilities in sorted order
        int base = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) {                 // compaction in increasing vocabulary index (index = 64 i + lane)
          const unsigned long long mk = __ballot(lg[i] != -INFINITY);
          if (lg[i] != -INFINITY) {
            const int pos = base + __popcll(mk & ((1ull << lane) - 1ull));
            lp[pos] = lg[i]; sp[pos] = qn[i]; li[pos] = (unsigned short)(lane + 64 * i);
          }
          base += __popcll(mk);
        }
        __builtin_amdgcn_wave_barrier();
        const bool act = lane < nsv;
        const float l = act ? lp[lane] : -INFINITY;
        const float q1 = act ? sp[lane] : 1.0f;
        const int idx = act ? (int)li[lane] : 0x7fff;
        const int Lc = idx & 63;
        // occurrence number of this element among the survivors of its original lane (increasing index)
        int occ = 0;
        for (int j = 0; j < nsv; ++j) {
          const int lj = __builtin_amdgcn_readlane(Lc, j);
          occ += (j < lane && lj == Lc) ? 1 : 0;
        }
        int maxocc = occ;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxocc = max(maxocc, __shfl_xor(maxocc, o, 64));
        auto lane_order_sum = [&](float e) {           // sum over all elements exactly as the 17-register form does it
          lanep[lane] = 0.f;
          __builtin_amdgcn_wave_barrier();
          for (int r = 0; r <= maxocc; ++r) {
            if (act && occ == r) lanep[Lc] = lanep[Lc] + e;
            __builtin_amdgcn_wave_barrier();
          }
          const float zl = lanep[lane];
          __builtin_amdgcn_wave_barrier();
          return wave_sum(zl);
        };
        float lcur = l;
        if (p.top_p < 1.0f) {                          // model.py:56-70
          const float m = wave_max(l);
          const float e = act ? expf(l - m) : 0.f;
          const float z = lane_order_sum(e);
          const float pr = e / z;
          const bool actp = act && pr > 0.f;           // (a survivor whose probability underflows sorts after everything)
          const int ns = __popcll(__ballot(actp));
          int r = 0;
          for (int j = 0; j < nsv; ++j) {
            const float pj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pr), j));
            const int ij = __builtin_amdgcn_readlane(idx, j);
            r += (pj > 0.f && (pj > pr || (pj == pr && ij < idx))) ? 1 : 0;
          }
          if (actp) srt[r] = pr;
          __builtin_amdgcn_wave_barrier();
          // cumulative sum in sorted order, in double like torch.cumsum on CPU (wave inclusive scan)
          double cum = (lane < ns) ? (double)srt[lane] : 0.0;
#pragma unroll
          for (int o = 1; o < 64; o <<= 1) {
            const double up = __shfl_up(cum, o, 64);
            if (lane >= o) cum += up;
          }
          const unsigned long long over = __ballot(lane < ns && (float)cum > p.top_p);
          const int first_over = over ? (__ffsll((long long)over) - 1) : ns;      // smallest r with cum[r] > top_p
          int keep = min(ns, first_over + 1);
          if (ns == 0) keep = 0;
          if (!(actp && r < keep)) lcur = -INFINITY;
          __builtin_amdgcn_wave_barrier();
        }
        SSTAMP(3);
        // ---- final softmax + multinomial as argmax(p / q) (model.py:73-82)
        const float m2 = wave_max(lcur);
        const float e2 = (lcur == -INFINITY) ? 0.f : expf(lcur - m2);
        const float z2 = lane_order_sum(e2);
        const float sc = act ? (e2 / z2) / q1 : -1.f;
        choice = wave_argmax(sc, act ? idx : 0x7fffffff);
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
      int eos_detected = fsm[0], eos_countdown = fsm[1], bos_countdown = fsm[2];      // same address in every lane
      const int old = ch ? trow[lane] : 0;
      const int d = ch ? p.delay[lane] : 0;
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
      if (ch) tok_next[lane] = tk;
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
    if (lane == 0) go_next = go;
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
  e.p_ktiles = a->p_ktiles; e.ssq_ld = a->ssq_ld; e.ssq = a->ssq; e.cmap = a->cmap;
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
