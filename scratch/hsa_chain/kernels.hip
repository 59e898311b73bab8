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
