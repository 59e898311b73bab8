// Feasibility probe: can a chain of dependent GEMV-like launches be software-pipelined over TWO hardware queues, the
// dependency carried by device-side counters instead of the stream order?  Launch k+1 (other queue) is resident while
// launch k computes: it requests its weight slice first, then waits for launch k's workgroups to signal, then reads
// launch k's output with device-coherent loads.  Same-queue order (k, k+2, ...) keeps at most two launches resident.
//   mode 0  one stream, one graph, plain loads / stores (what the engine does today)
//   mode 1  one stream, but with the counter protocol (its own overhead)
//   mode 2  two streams, two graphs (even / odd launches), counter protocol
// Every spin is bounded (2 ms of wall clock): a protocol error shows up as err != 0, never as a hang.
// Build: hipcc -O3 --offload-arch=gfx950 -o pipeline_probe pipeline_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned ld_agent(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u4 ld16_agent(const u4* p) {
  const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
  unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return u4{(unsigned)a, (unsigned)(a >> 32), (unsigned)b, (unsigned)(b >> 32)};
}
__device__ __forceinline__ void st16_agent(u4* p, u4 v) {
  unsigned long long* q = reinterpret_cast<unsigned long long*>(p);
  __hip_atomic_store(q, ((unsigned long long)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(q + 1, ((unsigned long long)v.w << 32) | v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct Stage {
  const u4* w; long per_wg;            // weight slice of workgroup b: w + b * per_wg (u4 units), NT * 8 u4 per round
  const u4* a_in; u4* a_out;           // 24 KB operand in / out
  unsigned* arr_prev; unsigned grid_prev; unsigned* arr_mine; unsigned* done_mine; unsigned prev_wraps;
  unsigned* err; int rounds;           // rounds of 64 KB per workgroup (1 = o-like, 4 = wi-like)
};

template <bool PIPE>
__global__ __launch_bounds__(512) void k_stage(Stage p) {
  __shared__ u4 As[1536];
  __shared__ float red[8][64];
  __shared__ unsigned s_epoch;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const u4* base = p.w + (size_t)blockIdx.x * p.per_wg + tid;
  u4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = __builtin_nontemporal_load(base + i * 512);       // the weight stream starts first
  u4 av[3];
  if (PIPE) {
    if (tid == 0) {
      const unsigned epoch = ld_agent(p.done_mine);                 // completed launches of THIS stage: stable until our last arriver bumps it
      const unsigned target = (epoch + (p.prev_wraps ? 0u : 1u)) * p.grid_prev;
      const long long t0 = wall_clock64();
      while (ld_agent(p.arr_prev) < target) {
        __builtin_amdgcn_s_sleep(2);
        if (wall_clock64() - t0 > 200000) { atomicAdd(p.err, 1u); break; }      // 2 ms at 100 MHz
      }
      s_epoch = epoch;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i) av[i] = ld16_agent(p.a_in + tid + 512 * i);
  } else {
#pragma unroll
    for (int i = 0; i < 3; ++i) av[i] = p.a_in[tid + 512 * i];
  }
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) As[tid + 512 * i] = av[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 3; ++i) { const u4 t = As[(tid * 7 + 512 * i + 13) % 1536]; s += t.x ^ t.w; }
  for (int r = 0; r < p.rounds; ++r) {
    u4 n[8];
    if (r + 1 < p.rounds) {
#pragma unroll
      for (int i = 0; i < 8; ++i) n[i] = __builtin_nontemporal_load(base + (size_t)(r + 1) * 4096 + i * 512);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
    if (r + 1 < p.rounds) {
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = n[i];
    }
  }
  red[wv][lane] = (float)s;
  __syncthreads();
  if (tid < 64) {
    float a = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) a += red[k][tid];
    red[0][tid] = a;
  }
  __syncthreads();
  if (tid < 32) {
    const float x = red[0][tid] + red[0][tid + 32];
    const unsigned xb = __float_as_uint(x) | 1u;
    const u4 o = {xb, xb + 1, xb + 2, xb + 3};
    if (tid < 12) {
      u4* dst = p.a_out + (blockIdx.x % 128) * 12 + tid;          // 128 x 192 B = 24 KB
      if (PIPE) st16_agent(dst, o); else *dst = o;
    }
  }
  if (PIPE) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      const unsigned epoch = s_epoch;
      const unsigned ticket = __hip_atomic_fetch_add(p.arr_mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ticket == (epoch + 1) * gridDim.x - 1) __hip_atomic_store(p.done_mine, epoch + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

int main(int argc, char** argv) {
  const int NK = argc > 1 ? atoi(argv[1]) : 146;
  const int steps = argc > 2 ? atoi(argv[2]) : 200;
  const int grid = argc > 3 ? atoi(argv[3]) : 128;
  const int rounds = argc > 4 ? atoi(argv[4]) : 1;
  const size_t per_wg = (size_t)4096 * rounds;                    // u4 per workgroup
  const size_t stage_u4 = per_wg * grid;
  const int NBUF = 18;                                            // cold weights: 18 distinct slices, reused round robin
  u4* w; CK(hipMalloc(&w, stage_u4 * 16 * NBUF)); CK(hipMemset(w, 1, stage_u4 * 16 * NBUF));
  u4* act[2]; for (int i = 0; i < 2; ++i) { CK(hipMalloc(&act[i], 24576)); CK(hipMemset(act[i], 0, 24576)); }
  unsigned* ctr; CK(hipMalloc(&ctr, sizeof(unsigned) * (2 * NK + 1)));
  hipStream_t st[2]; CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
  printf("NK=%d steps=%d grid=%d rounds=%d: %.1f MB per launch, %.2f GB per step\n", NK, steps, grid, rounds, stage_u4 * 16 / 1e6, stage_u4 * 16.0 * NK / 1e9);
  for (int mode = 0; mode < 3; ++mode) {
    CK(hipMemset(ctr, 0, sizeof(unsigned) * (2 * NK + 1)));
    CK(hipDeviceSynchronize());
    unsigned* arr = ctr; unsigned* done = ctr + NK; unsigned* err = ctr + 2 * NK;
    hipGraph_t g[2] = {nullptr, nullptr}; hipGraphExec_t ge[2] = {nullptr, nullptr};
    const int nstreams = mode == 2 ? 2 : 1;
    for (int q = 0; q < nstreams; ++q) {
      CK(hipStreamBeginCapture(st[q], hipStreamCaptureModeThreadLocal));
      for (int k = 0; k < NK; ++k) {
        if (nstreams == 2 && (k & 1) != q) continue;
        Stage p;
        p.w = w + (size_t)(k % NBUF) * stage_u4; p.per_wg = (long)per_wg; p.a_in = act[(k + 1) & 1]; p.a_out = act[k & 1];
        const int kp = (k + NK - 1) % NK;
        p.arr_prev = arr + kp; p.grid_prev = grid; p.arr_mine = arr + k; p.done_mine = done + k; p.prev_wraps = k == 0; p.err = err; p.rounds = rounds;
        if (mode == 0) hipLaunchKernelGGL(k_stage<false>, dim3(grid), dim3(512), 0, st[q], p);
        else hipLaunchKernelGGL(k_stage<true>, dim3(grid), dim3(512), 0, st[q], p);
      }
      CK(hipStreamEndCapture(st[q], &g[q]));
      CK(hipGraphInstantiate(&ge[q], g[q], nullptr, nullptr, 0));
    }
    hipEvent_t e0, e1[2]; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1[0])); CK(hipEventCreate(&e1[1]));
    for (int rep = 0; rep < 2; ++rep) {                            // rep 0 = warm-up
      const int n = rep == 0 ? 20 : steps;
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, st[0]));
      if (nstreams == 2) CK(hipStreamWaitEvent(st[1], e0, 0));
      for (int s = 0; s < n; ++s)
        for (int q = 0; q < nstreams; ++q) CK(hipGraphLaunch(ge[q], st[q]));
      for (int q = 0; q < nstreams; ++q) CK(hipEventRecord(e1[q], st[q]));
      CK(hipDeviceSynchronize());
      float ms = 0.f, ms2 = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1[0]));
      if (nstreams == 2) { CK(hipEventElapsedTime(&ms2, e0, e1[1])); if (ms2 > ms) ms = ms2; }
      unsigned herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      if (rep == 1)
        printf("mode %d: %.3f ms per step, %.3f us per launch, %.2f TB/s, err=%u\n", mode, ms / n, ms * 1e3 / n / NK,
               stage_u4 * 16.0 * NK / (ms / n * 1e-3) / 1e12, herr);
    }
    for (int q = 0; q < nstreams; ++q) { (void)hipGraphExecDestroy(ge[q]); (void)hipGraphDestroy(g[q]); }
  }
  return 0;
}
