// start time of every workgroup of one launch against its XCD (HW_REG_XCC_ID) and block index
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k(long long* t, int* xcc) {
  const long long t0 = wall_clock64();
  if (threadIdx.x == 0) { t[blockIdx.x] = t0; xcc[blockIdx.x] = __builtin_amdgcn_s_getreg((4 << 11) | (0 << 6) | 20) & 0xf; }   // HW_REG_XCC_ID = 20, bits [3:0]
  while (wall_clock64() - t0 < 300) __builtin_amdgcn_s_sleep(8);
}
int main() {
  const int G = 256;
  long long* d; int* x; CK(hipMalloc(&d, 8 * G)); CK(hipMalloc(&x, 4 * G));
  std::vector<long long> h(G); std::vector<int> hx(G);
  hipStream_t st; CK(hipStreamCreate(&st));
  hipGraph_t g; hipGraphExec_t ge;
  for (int mode = 0; mode < 2; ++mode) {
    if (mode == 1) {      // back to back inside a replayed graph: look at the LAST launch of a chain of 4
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(k, dim3(G), dim3(512), 0, st, d, x);
      CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    }
    for (int rep = 0; rep < 3; ++rep) {
      if (mode == 0) hipLaunchKernelGGL(k, dim3(G), dim3(512), 0, st, d, x); else CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      CK(hipMemcpy(h.data(), d, 8 * G, hipMemcpyDeviceToHost)); CK(hipMemcpy(hx.data(), x, 4 * G, hipMemcpyDeviceToHost));
      long long lo = h[0]; for (auto v : h) lo = v < lo ? v : lo;
      double sum[8] = {0}, mx[8] = {0}, mn[8]; int cnt[8] = {0}; for (int j = 0; j < 8; ++j) mn[j] = 1e9;
      for (int i = 0; i < G; ++i) { int j = hx[i] & 7; double us = (h[i] - lo) / 100.0; sum[j] += us; cnt[j]++; if (us > mx[j]) mx[j] = us; if (us < mn[j]) mn[j] = us; }
      printf("%s rep %d: per XCD (count, min..max start us):", mode ? "graph " : "eager ", rep);
      for (int j = 0; j < 8; ++j) printf("  x%d:%d %.2f..%.2f", j, cnt[j], mn[j], mx[j]);
      printf("\n   block -> xcc of the first 16 blocks:"); for (int i = 0; i < 16; ++i) printf(" %d", hx[i]); printf("\n");
    }
  }
  return 0;
}
