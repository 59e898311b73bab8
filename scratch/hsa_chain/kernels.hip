// device code for hsa_chain.cpp (built with hipcc --genco): no blockDim / gridDim (they would come from hidden kernargs)
#include <hip/hip_runtime.h>
extern "C" __global__ void k_empty(int* p) { if (p == (int*)1) *p = 0; }

extern "C" __global__ void k_fill(unsigned* a, unsigned v) { a[blockIdx.x * 256 + threadIdx.x] = v; }

// 128 workgroups x 256 threads: every workgroup reads the whole 24 KB operand the previous launch wrote (6144 words == step),
// counts words that are not `step`, and writes its 48 words of the next operand (step + 1)
extern "C" __global__ __launch_bounds__(256) void k_pingpong(const unsigned* in, unsigned* out, unsigned* err, unsigned step) {
  unsigned bad = 0;
#pragma unroll
  for (int i = 0; i < 24; ++i) bad += in[threadIdx.x + 256 * i] != step;
  if (bad) atomicAdd(err, bad);
  if (threadIdx.x < 48) out[blockIdx.x * 48 + threadIdx.x] = step + 1;
}
// the same with device-coherent accesses (sc1): what the product kernels use for cross-workgroup hand-offs
extern "C" __global__ __launch_bounds__(256) void k_pingpong_sc1(const unsigned* in, unsigned* out, unsigned* err, unsigned step) {
  unsigned bad = 0;
#pragma unroll
  for (int i = 0; i < 24; ++i) bad += __hip_atomic_load(in + threadIdx.x + 256 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != step;
  if (bad) atomicAdd(err, bad);
  if (threadIdx.x < 48) __hip_atomic_store(out + blockIdx.x * 48 + threadIdx.x, step + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the GEMV skeleton of scratch/launchcost (variant (h)/(m)): 128 workgroups x 512 threads stream 64 KB of cold weights each, stage the
// 24 KB operand the previous launch wrote through LDS, reduce across waves and write 192 B of the next operand
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
extern "C" __global__ __launch_bounds__(512) void k_gemvlike(const u4* w, const u4* a_in, u4* a_out) {
  __shared__ u4 As[1536];
  __shared__ float red[8][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  u4 av[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) av[i] = a_in[tid + 512 * i];
  const u4* base = w + (size_t)blockIdx.x * 4096 + tid;
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

// the same skeleton with a write-through (sc1) store of the next operand and the wave waiting for its acknowledgement before it
// ends: what a chain without packet-level release fences needs from its producers
extern "C" __global__ __launch_bounds__(512) void k_gemvlike_wt(const u4* w, const u4* a_in, u4* a_out) {
  __shared__ u4 As[1536];
  __shared__ float red[8][64];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  u4 av[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) av[i] = a_in[tid + 512 * i];
  const u4* base = w + (size_t)blockIdx.x * 4096 + tid;
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
    if (tid < 12) {
      __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(a_out, 0, 0x7fffffff, 0x00020000);
      __builtin_amdgcn_raw_buffer_store_b128(o, r, ((blockIdx.x % 128) * 12 + tid) * 16, 0, 16);      // aux bit 4 = sc1
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
}
