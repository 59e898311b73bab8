// Error plumbing shared by all translation units of libdia_hip.so.
#pragma once
#include <hip/hip_runtime.h>

int dia_fail(int code, const char* msg);
int dia_fail_hip(hipError_t e, const char* where);
// hipGetLastError() after a launch; 0 when clean
int dia_check_launch(const char* kernel);

// one-time per-process kernel attribute setup (large dynamic LDS); called by dia_kernels_init()
int dia_attn_init();
int dia_sample_init();
int dia_gemm_init();
int dia_kernels_init_once();
