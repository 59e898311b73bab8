#include "errors.hpp"
#include "launch.hpp"
#include "../../include/dia_hip.h"
#include <string>
#include <cxxabi.h>
#include <cstdlib>
#include <vector>

static thread_local std::string g_err;

int dia_fail(int code, const char* msg) {
  g_err = msg ? msg : "";
  return code;
}

int dia_fail_hip(hipError_t e, const char* where) {
  g_err = std::string(where ? where : "hip") + ": " + hipGetErrorString(e);
  return DIA_E_HIP;
}

int dia_check_launch(const char* kernel) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return DIA_OK;
  return dia_fail_hip(e, kernel);
}

extern "C" const char* dia_last_error(void) { return g_err.c_str(); }
extern "C" int dia_abi_version(void) { return DIA_ABI_VERSION; }
extern "C" int dia_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return dia_fail_hip(e, "hipGetDeviceCount");
  return n;
}

dia_launch_recorder& dia_recorder() {
  static thread_local dia_launch_recorder r;
  return r;
}

void dia_recorder_arm() {
  dia_launch_recorder& r = dia_recorder();
  for (auto& p : r.ev) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
  r.ev.clear();
  r.fn.clear();
  r.armed = true;
}

static thread_local std::vector<std::string> g_last_labels;

const char* dia_recorder_label(int i) {
  return (i >= 0 && i < (int)g_last_labels.size()) ? g_last_labels[i].c_str() : "";
}

int dia_recorder_collect(float* out_ms, int cap, float* out_interval_ms) {
  dia_launch_recorder& r = dia_recorder();
  r.armed = false;
  int rc = (int)r.ev.size();
  if (!r.ev.empty()) {
    hipError_t he = hipEventSynchronize(r.ev.back().second);
    if (he != hipSuccess) rc = dia_fail_hip(he, "dia_recorder_collect: hipEventSynchronize");
  }
  for (size_t i = 0; i < r.ev.size(); ++i) {
    if (rc >= 0 && out_ms && (int)i < cap) {
      float ms = -1.f;
      if (hipEventElapsedTime(&ms, r.ev[i].first, r.ev[i].second) != hipSuccess) { ms = -1.f; (void)hipGetLastError(); }
      out_ms[i] = ms;
      // end of the previous kernel -> end of this one: what a launch costs inside a dependent chain, boundary included
      // (rocprofv3 reports exactly this as the "duration" of a kernel in a replayed graph: begin[n+1] == end[n] there)
      if (out_interval_ms) {
        float iv = ms;
        if (i > 0 && hipEventElapsedTime(&iv, r.ev[i - 1].second, r.ev[i].second) != hipSuccess) { iv = -1.f; (void)hipGetLastError(); }
        out_interval_ms[i] = iv;
      }
    }
  }
  for (size_t i = 0; i < r.ev.size(); ++i) {
    (void)hipEventDestroy(r.ev[i].first); (void)hipEventDestroy(r.ev[i].second);
  }
  r.ev.clear();
  g_last_labels.clear();
  for (const void* f : r.fn) {              // "void (anonymous namespace)::k_gemm16<8, 8, false, false>(...)" -> "k_gemm16<8, 8, false, false>"
    const char* mangled = hipKernelNameRefByPtr(f, nullptr);
    std::string t(mangled ? mangled : "");
    int st = 0;
    char* dem = abi::__cxa_demangle(t.c_str(), nullptr, nullptr, &st);
    if (st == 0 && dem) t = dem;
    free(dem);
    if (t.rfind("void ", 0) == 0) t = t.substr(5);
    const std::string anon = "(anonymous namespace)::";
    for (size_t q; (q = t.find(anon)) != std::string::npos;) t.erase(q, anon.size());
    int depth = 0;                                                    // cut the parameter list: first '(' outside <...>
    for (size_t q = 0; q < t.size(); ++q) {
      if (t[q] == '<') ++depth;
      else if (t[q] == '>') --depth;
      else if (t[q] == '(' && depth == 0) { t.erase(q); break; }
    }
    g_last_labels.push_back(t);
  }
  r.fn.clear();
  return rc;
}
