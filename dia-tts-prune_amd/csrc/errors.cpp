#include "errors.hpp"
#include "../../include/dia_hip.h"
#include <string>

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
