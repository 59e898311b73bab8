// What does one dependent kernel launch cost inside a replayed hipGraph on this GPU, and which part of a GEMV's
// skeleton adds how much?  Chains of 146 identical launches on one stream, captured once and replayed; microseconds
// per launch.  Build: hipcc -O3 --offload-arch=gfx950 -o launchcost launchcost.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(int* p) { if (p == (int*)1) *p = 0; }

// F & 1 = stage a 24 KB operand through LDS before the stream is consumed; F & 2 = cross-wave reduce through LDS + a
// 32-thread epilogue; F & 4 = the epilogue writes the NEXT launch's 24 KB operand (192 B per workgroup); F & 8 = no
// weight stream at all
template <int F>
__global__ void k_gemvlike(const u4* w, const u4* a_in, u4* a_out, float* out, int per_wg) {
  __shared__ u4 As[1536];
  __shared__ float red[8][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  u4 av[3];
  if (F & 1) {
#pragma unroll
    for (int i = 0; i < 3; ++i) av[i] = a_in[tid + 512 * i];
  }
  const u4* base = w + (size_t)blockIdx.x * per_wg + tid;
  u4 v[8];
  if (!(F & 8)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = __builtin_nontemporal_load(base + i * 512);
  }
  unsigned s = 0;
  if (F & 1) {
#pragma unroll
    for (int i = 0; i < 3; ++i) As[tid + 512 * i] = av[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 3; ++i) { const u4 t = As[(tid * 7 + 512 * i + 13) % 1536]; s += t.x ^ t.w; }
  }
  if (!(F & 8)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  }
  if (F & 2) {
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
      if (F & 4) {
        const unsigned xb = __float_as_uint(x) | 1u;
        const u4 o = {xb, xb + 1, xb + 2, xb + 3};
        if (tid < 12) a_out[(blockIdx.x % 128) * 12 + tid] = o;          // 128 workgroups x 192 B = 24 KB
      } else if (tid == 0) out[blockIdx.x] = x;
    }
  } else if (s == 0x12345678u) out[blockIdx.x] = 1.f;
}

// (h) with the stream and the operand sized by template: NL 16-byte weight loads per thread (8 = 64 KB per workgroup,
// 4 = 32 KB: "half strips" on twice the workgroups), NI 16-byte operand loads per thread (3 = 24 KB = three bf16 planes
// of 2 x 2048 values, 2 = 16 KB = the same values as fp32)
template <int NL, int NI>
__global__ void k_gemvlike_sized(const u4* w, const u4* a_in, u4* a_out, float* out, int per_wg) {
  __shared__ u4 As[512 * NI];
  __shared__ float red[8][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  u4 av[NI];
#pragma unroll
  for (int i = 0; i < NI; ++i) av[i] = a_in[tid + 512 * i];
  const u4* base = w + (size_t)blockIdx.x * per_wg + tid;
  u4 v[NL];
#pragma unroll
  for (int i = 0; i < NL; ++i) v[i] = __builtin_nontemporal_load(base + i * 512);
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < NI; ++i) As[tid + 512 * i] = av[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < NI; ++i) { const u4 t = As[(tid * 7 + 512 * i + 13) % (512 * NI)]; s += t.x ^ t.w; }
#pragma unroll
  for (int i = 0; i < NL; ++i) s += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
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
    if (tid < 12) a_out[(blockIdx.x % 128) * 12 + tid] = o;
  }
}

// (h) again with the real kernel's resource footprint: dynamic LDS and a forced VGPR allocation
template <int F>
__global__ __attribute__((amdgpu_num_vgpr(128))) void k_gemvlike_fat(const u4* w, const u4* a_in, u4* a_out, float* out, int per_wg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dyn[];
  u4* As = reinterpret_cast<u4*>(dyn);
  float (*red)[64] = reinterpret_cast<float (*)[64]>(dyn + 1536 * 16);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  u4 av[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) av[i] = a_in[tid + 512 * i];
  const u4* base = w + (size_t)blockIdx.x * per_wg + tid;
  u4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = __builtin_nontemporal_load(base + i * 512);
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) As[tid + 512 * i] = av[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 3; ++i) { const u4 t = As[(tid * 7 + 512 * i + 13) % 1536]; s += t.x ^ t.w; }
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
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
    if (tid < 12) a_out[(blockIdx.x % 128) * 12 + tid] = o;
  }
}

// (h) once more with wall-clock stamps: when does each workgroup start, issue its stream, finish?
__global__ void k_gemvlike_stamped(const u4* w, const u4* a_in, u4* a_out, float* out, int per_wg, long long* stamps) {
  __shared__ u4 As[1536];
  __shared__ float red[8][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  if (tid == 0) stamps[blockIdx.x * 4 + 0] = wall_clock64();
  u4 av[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) av[i] = a_in[tid + 512 * i];
  const u4* base = w + (size_t)blockIdx.x * per_wg + tid;
  u4 v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = __builtin_nontemporal_load(base + i * 512);
  if (tid == 0) stamps[blockIdx.x * 4 + 1] = wall_clock64();
  unsigned s = 0;
#pragma unroll
  for (int i = 0; i < 3; ++i) As[tid + 512 * i] = av[i];
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 3; ++i) { const u4 t = As[(tid * 7 + 512 * i + 13) % 1536]; s += t.x ^ t.w; }
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i].x ^ v[i].y ^ v[i].z ^ v[i].w;
  red[wv][lane] = (float)s;
  __syncthreads();
  if (tid == 0) stamps[blockIdx.x * 4 + 2] = wall_clock64();
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
    if (tid < 12) a_out[(blockIdx.x % 128) * 12 + tid] = o;
  }
  if (tid == 0) stamps[blockIdx.x * 4 + 3] = wall_clock64();
}

template <typename F>
int run(const char* name, hipStream_t st, int n, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int i = 0; i < n; ++i) launch(i);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipStreamSynchronize(st));
  const int reps = 20;
  CK(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
  CK(hipEventRecord(e1, st));
  CK(hipStreamSynchronize(st));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%-66s %6.2f us per launch\n", name, ms * 1e3 / (reps * n));
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return 0;
}

int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  u4* big; CK(hipMalloc(&big, 544u << 20)); CK(hipMemset(big, 1, 544u << 20));   // 64 x 8 MB: beyond L2 + Infinity Cache
  float* out; CK(hipMalloc(&out, 1 << 20));
  u4 *a0, *a1; CK(hipMalloc(&a0, 24576)); CK(hipMalloc(&a1, 24576)); CK(hipMemset(a0, 1, 24576)); CK(hipMemset(a1, 1, 24576));
  int* dummy = nullptr;
  const int N = 146;
  auto wb = [&](int i) { return big + (size_t)(i % 64) * ((8u << 20) / 16); };
#define RUN(name, F, grid, ain, aout) if (run(name, st, N, [&](int i) { hipLaunchKernelGGL(k_gemvlike<F>, dim3(grid), dim3(512), 0, st, wb(i), ain, aout, out, 4096); })) return 1
  if (run("(a) 1 x 64 threads, empty", st, N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st, dummy); })) return 1;
  if (run("(b) 256 x 512 threads, empty", st, N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(512), 0, st, dummy); })) return 1;
  if (run("(c) 256 x 1024 threads, empty", st, N, [&](int) { hipLaunchKernelGGL(k_empty, dim3(256), dim3(1024), 0, st, dummy); })) return 1;
  RUN("(d) 128 x 512: 8 MB cold stream, nothing else", 0, 128, a0, a1);
  RUN("(e) (d) + cross-wave LDS reduce + one store per workgroup", 2, 128, a0, a1);
  RUN("(f) (d) + 24 KB operand (never rewritten) staged through LDS", 1, 128, a0, a1);
  RUN("(g) (e) + (f)", 3, 128, a0, a1);
  RUN("(h) (g), operand WRITTEN by the previous launch (ping-pong)", 7, 128, (i & 1) ? a1 : a0, (i & 1) ? a0 : a1);
  RUN("(i) (h) without the weight stream", 15, 128, (i & 1) ? a1 : a0, (i & 1) ? a0 : a1);
  if (run("(k) (h) with 34 KB of dynamic LDS and 128 VGPRs allocated", st, N, [&](int i) { hipLaunchKernelGGL(k_gemvlike_fat<7>, dim3(128), dim3(512), 34816, st, wb(i), (i & 1) ? a1 : a0, (i & 1) ? a0 : a1, out, 4096); })) return 1;
  if (run("(l) (k) with 60 KB of dynamic LDS", st, N, [&](int i) { hipLaunchKernelGGL(k_gemvlike_fat<7>, dim3(128), dim3(512), 61440, st, wb(i), (i & 1) ? a1 : a0, (i & 1) ? a0 : a1, out, 4096); })) return 1;
  // (j): 256 workgroups read 16 MB per launch — 32 distinct 16 MB windows of the same 512 MB buffer
  if (run("(j) (h) with 256 workgroups, 16 MB", st, N, [&](int i) { hipLaunchKernelGGL(k_gemvlike<7>, dim3(256), dim3(512), 0, st, big + (size_t)(i % 32) * ((16u << 20) / 16), (i & 1) ? a1 : a0, (i & 1) ? a0 : a1, out, 4096); })) return 1;
#define RUNS(name, NL, NI, grid) if (run(name, st, N, [&](int i) { hipLaunchKernelGGL((k_gemvlike_sized<NL, NI>), dim3(grid), dim3(512), 0, st, wb(i), (i & 1) ? a1 : a0, (i & 1) ? a0 : a1, out, 512 * NL); })) return 1
  RUNS("(m) (h) again: 128 x 64 KB weights, 24 KB operand", 8, 3, 128);
  RUNS("(n) 128 x 64 KB weights, 16 KB operand (fp32 instead of 3 planes)", 8, 2, 128);
  RUNS("(o) 256 x 32 KB weights (half strips), 24 KB operand", 4, 3, 256);
  RUNS("(p) 256 x 32 KB weights (half strips), 16 KB operand", 4, 2, 256);
  RUNS("(q) 256 x 32 KB weights, 8 KB operand", 4, 1, 256);
  RUNS("(r) 192 x 64 KB weights (qkv), 24 KB operand", 8, 3, 192);
  RUNS("(s) 192 x 64 KB weights (qkv), 16 KB operand", 8, 2, 192);
  {   // stamped skeleton: a few back-to-back launches, the last one's timeline (100 MHz clock)
    long long* st_d; CK(hipMalloc(&st_d, 128 * 4 * sizeof(long long)));
    for (int i = 0; i < 6; ++i) hipLaunchKernelGGL(k_gemvlike_stamped, dim3(128), dim3(512), 0, st, wb(i), (i & 1) ? a1 : a0, (i & 1) ? a0 : a1, out, 4096, st_d);
    CK(hipStreamSynchronize(st));
    std::vector<long long> h(128 * 4);
    CK(hipMemcpy(h.data(), st_d, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    long long t0 = h[0];
    for (int b = 0; b < 128; ++b) t0 = h[b * 4] < t0 ? h[b * 4] : t0;
    const char* names[4] = {"start", "stream issued", "stream consumed", "end"};
    for (int k = 0; k < 4; ++k) {
      double mn = 1e9, mx = 0, sum = 0;
      for (int b = 0; b < 128; ++b) { const double us = (h[b * 4 + k] - t0) / 100.0; mn = us < mn ? us : mn; mx = us > mx ? us : mx; sum += us; }
      printf("    skeleton workgroups: %-16s min %5.2f  mean %5.2f  max %5.2f us\n", names[k], mn, sum / 128, mx);
    }
  }
  return 0;
}
