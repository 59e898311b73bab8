// How fast does the dispatcher start the workgroups of one launch, as a function of what a workgroup asks for?
// Every workgroup stamps the 100 MHz wall clock at entry; printed: last start - first start (us) for a grid of G
// workgroups of T threads with L bytes of dynamic LDS and a forced VGPR allocation.
// Build: hipcc -O3 --offload-arch=gfx950 -o dispatch_probe dispatch_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int V> struct Tag {};
#define PROBE_BODY \
  extern __shared__ unsigned char dyn[]; \
  const long long t0 = wall_clock64(); \
  if (threadIdx.x == 0) t[blockIdx.x] = t0; \
  if (spin) { \
    while (wall_clock64() - t0 < 200) __builtin_amdgcn_s_sleep(8); \
    if (threadIdx.x == 1 && dyn[threadIdx.x] == 77) t[blockIdx.x] += 1; \
  }
__global__ __attribute__((amdgpu_num_vgpr(32))) void k_probe32(long long* t, int spin) { PROBE_BODY }
__global__ __attribute__((amdgpu_num_vgpr(64))) void k_probe64(long long* t, int spin) { PROBE_BODY }
__global__ __attribute__((amdgpu_num_vgpr(128))) void k_probe128(long long* t, int spin) { PROBE_BODY }
__global__ __attribute__((amdgpu_num_vgpr(256))) void k_probe256(long long* t, int spin) { PROBE_BODY }
template <int V> struct Pick;
template <> struct Pick<32> { static constexpr auto fn = k_probe32; };
template <> struct Pick<64> { static constexpr auto fn = k_probe64; };
template <> struct Pick<128> { static constexpr auto fn = k_probe128; };
template <> struct Pick<256> { static constexpr auto fn = k_probe256; };
#if 0
template <int V>
__global__ void k_probe(long long* t, int spin) {
  extern __shared__ unsigned char dyn[];
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) t[blockIdx.x] = t0;
  if (spin) {                                       // keep the workgroup resident for ~2 us so that nothing is re-used
    while (wall_clock64() - t0 < 200) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 1 && dyn[threadIdx.x] == 77) t[blockIdx.x] += 1;
  }
}
#endif

template <int V>
int run(int G, int T, int L, long long* d, std::vector<long long>& h) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(Pick<V>::fn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  float best = 1e9, sum = 0;
  for (int rep = 0; rep < 12; ++rep) {
    hipLaunchKernelGGL(Pick<V>::fn, dim3(G), dim3(T), L, 0, d, 1);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(h.data(), d, sizeof(long long) * G, hipMemcpyDeviceToHost));
    const long long lo = *std::min_element(h.begin(), h.begin() + G), hi = *std::max_element(h.begin(), h.begin() + G);
    const float us = (hi - lo) / 100.0f;
    if (rep >= 2) { best = std::min(best, us); sum += us; }
  }
  printf("G=%4d T=%4d LDS=%6d VGPR=%3d : start spread min %.2f  mean %.2f us\n", G, T, L, V, best, sum / 10);
  return 0;
}

int main() {
  long long* d; CK(hipMalloc(&d, sizeof(long long) * 4096));
  std::vector<long long> h(4096);
  for (int G : {128, 256, 1024}) {
    for (int T : {256, 512, 1024}) {
      for (int L : {0, 8192, 34816, 65536}) if (run<32>(G, T, L, d, h)) return 1;
    }
  }
  for (int G : {128, 256}) {
    if (run<64>(G, 512, 34816, d, h)) return 1;
    if (run<128>(G, 512, 34816, d, h)) return 1;
    if (run<128>(G, 1024, 34816, d, h)) return 1;
    if (run<256>(G, 256, 34816, d, h)) return 1;
  }
  return 0;
}
