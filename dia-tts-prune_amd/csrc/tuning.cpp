#include "tuning.hpp"
#include "errors.hpp"
#include "../../include/dia_hip.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

static const char* const k_names[DIA_TUNE_COUNT] = {
    "attn_nz", "attn_gpw", "attn_gpw_cross", "gemm_spw", "gemm_mz_max", "tile_min_blocks", "wo_sk", "wo_pair", "wo_nw", "wo_spw", "act_f32", "wo_diag", "gemm_2t", "gemm_zr", "seg", "seg_nb", "seg_dbg", "seg_sleep", "g2t_wgs", "ckv_merge",
    "mlp_fuse", "tile_v", "blk32_kr", "blk32_ws", "no_g32", "g32_all", "g32m"};
static int g_tune[DIA_TUNE_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};

int dia_tune(int id) { return (id >= 0 && id < DIA_TUNE_COUNT) ? g_tune[id] : -1; }

extern "C" int dia_set_tuning(const char* name, int value) {
  if (!name) return dia_fail(DIA_E_ARG, "dia_set_tuning: null name");
  dia_tuning_init_from_env();       // (an explicit setting must not be overwritten by a later first read of DIA_TUNE)
  for (int i = 0; i < DIA_TUNE_COUNT; ++i)
    if (strcmp(name, k_names[i]) == 0) { g_tune[i] = value < 0 ? -1 : value; return DIA_OK; }
  return dia_fail(DIA_E_ARG, "dia_set_tuning: unknown knob");
}

extern "C" int dia_get_tuning(const char* name) {
  if (!name) return -1;
  dia_tuning_init_from_env();
  for (int i = 0; i < DIA_TUNE_COUNT; ++i)
    if (strcmp(name, k_names[i]) == 0) return g_tune[i];
  return -1;
}

void dia_tuning_init_from_env() {
  static bool done = false;
  if (done) return;
  done = true;
  const char* e = getenv("DIA_TUNE");
  if (!e) return;
  std::string s(e);
  size_t pos = 0;
  while (pos < s.size()) {
    size_t end = s.find(',', pos);
    if (end == std::string::npos) end = s.size();
    const std::string item = s.substr(pos, end - pos);
    const size_t eq = item.find('=');
    // a typo in an A/B sweep would silently measure the default: say so on stderr (the run goes on)
    if (eq == std::string::npos) {
      if (!item.empty()) fprintf(stderr, "dia_hip: DIA_TUNE item '%s' ignored (expected name=value)\n", item.c_str());
    } else if (dia_set_tuning(item.substr(0, eq).c_str(), atoi(item.c_str() + eq + 1)) != DIA_OK) {
      fprintf(stderr, "dia_hip: DIA_TUNE knob '%s' unknown, ignored\n", item.substr(0, eq).c_str());
    }
    pos = end + 1;
  }
}
