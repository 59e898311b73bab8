// Hooks from dia_gemm / the engine into gemm_experiments.hip (EXPERIMENTS=1 builds).  Without DIA_EXPERIMENTS the
// dispatcher never calls them and the C entry points they back return DIA_E_ARG.
#pragma once
#include "../../include/dia_hip.h"

#ifdef DIA_EXPERIMENTS
namespace { struct GemmK; }
// each returns a DIA_* status and sets `handled` when it launched (or failed) instead of the default kernels
int dia_exp_gemm_sparse(const dia_gemm_args* a, void* stream);
int dia_exp_gemm_two_mtiles(const dia_gemm_args* a, void* stream, bool& handled);
int dia_exp_tile_variant(const dia_gemm_args* a, void* stream, int variant);
int dia_exp_init();
#endif
