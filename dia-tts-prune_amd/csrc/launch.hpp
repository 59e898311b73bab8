// One launch path for every kernel of the library.  Normally a plain hipLaunchKernelGGL.  While a recorder is armed
// (dia_engine_time_step, dia_gemm_timed) each launch is bracketed by its own dispatch-level start / stop events
// (hipExtLaunchKernelGGL): the timestamps come from the dispatch packet itself — kernel begin / end, the quantity
// rocprofv3 --kernel-trace reports — not from markers between launches.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <vector>
#include <utility>

struct dia_launch_recorder {
  bool armed = false;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  std::vector<const void*> fn;          // host function of every recorded launch (resolved to its name at collect time)
};
dia_launch_recorder& dia_recorder();          // thread-local
// arm: subsequent launches of this thread are recorded.  collect: synchronises on the last stop event, writes the
// kernel durations (ms, launch order) into out[0..cap) and, when asked, the end-to-end intervals between consecutive
// kernels, returns their count (or a negative DIA_E_*), disarms and releases the events.
void dia_recorder_arm();
int dia_recorder_collect(float* out_ms, int cap, float* out_interval_ms = nullptr);
// kernel instantiation name of the i-th launch of the last collected recording ("k_gemv_small<8, 8, 2, false>"), or ""
const char* dia_recorder_label(int i);

template <auto Kern, typename... Args>
inline void dia_launch(dim3 grid, dim3 block, size_t smem, hipStream_t st, Args... args) {
  dia_launch_recorder& r = dia_recorder();
  if (r.armed) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipExtLaunchKernelGGL(Kern, grid, block, smem, st, e0, e1, 0, args...);
    r.ev.emplace_back(e0, e1);
    r.fn.push_back(reinterpret_cast<const void*>(Kern));
  } else {
    hipLaunchKernelGGL(Kern, grid, block, smem, st, args...);
  }
}
