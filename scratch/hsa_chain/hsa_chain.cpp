// What does a dependent kernel launch cost when the AQL packets are written by hand, and how much of it is the cache
// maintenance the packet header asks for?  Chains of 146 dispatches on a user-mode HSA queue, barrier bit set, with the
// acquire / release fence scopes of the header varied (HIP's graph replay measures 1.58 us for an empty kernel).
// The ping-pong kernels check whether a launch sees what the previous one wrote under each header.
// Build: hipcc --genco --offload-arch=gfx950 kernels.hip -o kernels.hsaco
//        g++ -O2 -std=c++17 -I/opt/rocm/include hsa_chain.cpp -o hsa_chain -L/opt/rocm/lib -lhsa-runtime64
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <string>
#include <vector>
#define CK(x) do { hsa_status_t s_ = (x); if (s_ != HSA_STATUS_SUCCESS) { const char* m_ = nullptr; hsa_status_string(s_, &m_); printf("%s -> %s\n", #x, m_ ? m_ : "?"); return 1; } } while (0)

static hsa_agent_t g_gpu, g_cpu; static bool g_have_gpu = false, g_have_cpu = false;
static hsa_status_t on_agent(hsa_agent_t a, void*) {
  hsa_device_type_t t; hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
  if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
  if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_amd_memory_pool_t g_dev_pool, g_karg_pool, g_fine_pool, g_ext_pool; static bool g_have_dev = false, g_have_karg = false, g_have_fine = false, g_have_ext = false;
static hsa_status_t on_gpu_pool(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t fl; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
  bool alloc = false; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
  if (alloc && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_have_dev) { g_dev_pool = p; g_have_dev = true; }
  if (alloc && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_EXTENDED_SCOPE_FINE_GRAINED) && !g_have_ext) { g_ext_pool = p; g_have_ext = true; }
  else if (alloc && (fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_FINE_GRAINED) && !g_have_fine) { g_fine_pool = p; g_have_fine = true; }
  return HSA_STATUS_SUCCESS;
}
static hsa_status_t on_cpu_pool(hsa_amd_memory_pool_t p, void*) {
  hsa_amd_segment_t seg; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
  if (seg != HSA_AMD_SEGMENT_GLOBAL) return HSA_STATUS_SUCCESS;
  uint32_t fl; hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &fl);
  if ((fl & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_KERNARG_INIT) && !g_have_karg) { g_karg_pool = p; g_have_karg = true; }
  return HSA_STATUS_SUCCESS;
}

struct Kern { uint64_t object; uint32_t karg_size, lds, scratch; };
static int get_kernel(hsa_executable_t ex, const char* name, Kern& k) {
  hsa_executable_symbol_t sym;
  CK(hsa_executable_get_symbol_by_name(ex, (std::string(name) + ".kd").c_str(), &g_gpu, &sym));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &k.object));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &k.karg_size));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &k.lds));
  CK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &k.scratch));
  return 0;
}

static hsa_queue_t* g_q; static uint64_t g_widx = 0;
static void push(const Kern& k, uint32_t grid_x, uint32_t wg_x, void* kargs, int barrier, int acq, int rel, hsa_signal_t done) {
  const uint64_t idx = g_widx++;
  hsa_kernel_dispatch_packet_t* pk = reinterpret_cast<hsa_kernel_dispatch_packet_t*>(g_q->base_address) + (idx & (g_q->size - 1));
  pk->workgroup_size_x = (uint16_t)wg_x; pk->workgroup_size_y = 1; pk->workgroup_size_z = 1; pk->reserved0 = 0;
  pk->grid_size_x = grid_x * wg_x; pk->grid_size_y = 1; pk->grid_size_z = 1;
  pk->private_segment_size = k.scratch; pk->group_segment_size = k.lds;
  pk->kernel_object = k.object; pk->kernarg_address = kargs; pk->reserved2 = 0; pk->completion_signal = done;
  const uint16_t header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (barrier << HSA_PACKET_HEADER_BARRIER) |
                                     (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
  const uint16_t setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
  __atomic_store_n(reinterpret_cast<uint32_t*>(pk), (uint32_t)header | ((uint32_t)setup << 16), __ATOMIC_RELEASE);
}
static double ring_and_wait(hsa_signal_t done) {
  hsa_queue_store_write_index_screlease(g_q, g_widx);
  const auto t0 = std::chrono::steady_clock::now();
  hsa_signal_store_screlease(g_q->doorbell_signal, (hsa_signal_value_t)(g_widx - 1));
  while (hsa_signal_wait_scacquire(done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_ACTIVE) >= 1) { printf("  (still waiting)\n"); }
  const auto t1 = std::chrono::steady_clock::now();
  hsa_signal_store_screlease(done, 1);
  return std::chrono::duration<double, std::micro>(t1 - t0).count();
}

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "scratch/hsa_chain/kernels.hsaco";
  CK(hsa_init());
  CK(hsa_iterate_agents(on_agent, nullptr));
  if (!g_have_gpu || !g_have_cpu) { printf("no GPU / CPU agent\n"); return 1; }
  CK(hsa_amd_agent_iterate_memory_pools(g_gpu, on_gpu_pool, nullptr));
  CK(hsa_amd_agent_iterate_memory_pools(g_cpu, on_cpu_pool, nullptr));
  if (!g_have_dev || !g_have_karg) { printf("no device / kernarg pool\n"); return 1; }
  CK(hsa_queue_create(g_gpu, 4096, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &g_q));

  std::ifstream f(path, std::ios::binary); std::vector<char> co((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  if (co.empty()) { printf("cannot read %s\n", path); return 1; }
  hsa_code_object_reader_t rd; CK(hsa_code_object_reader_create_from_memory(co.data(), co.size(), &rd));
  hsa_executable_t ex; CK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &ex));
  CK(hsa_executable_load_agent_code_object(ex, g_gpu, rd, nullptr, nullptr));
  CK(hsa_executable_freeze(ex, nullptr));
  Kern k_empty, k_fill, k_pp, k_pp1, k_gv, k_gvw;
  if (get_kernel(ex, "k_empty", k_empty) || get_kernel(ex, "k_fill", k_fill) || get_kernel(ex, "k_pingpong", k_pp) || get_kernel(ex, "k_pingpong_sc1", k_pp1) || get_kernel(ex, "k_gemvlike", k_gv) || get_kernel(ex, "k_gemvlike_wt", k_gvw)) return 1;

  const int N = 146, REPS = 20, TOT = N * REPS;
  // kernel arguments are staged on the host and copied into DEVICE memory before each chain: with the argument blocks in
  // host memory every wave's first scalar load crosses PCIe (15 us for an empty 256 x 512 grid; HIP keeps them on the device too)
  char* kargs; CK(hsa_amd_memory_pool_allocate(g_karg_pool, (size_t)(TOT + 8) * 256, 0, (void**)&kargs));
  CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, kargs));       // the copy below is executed by the GPU
  memset(kargs, 0, (size_t)(TOT + 8) * 256);
  char* kargs_dev; CK(hsa_amd_memory_pool_allocate(g_dev_pool, (size_t)(TOT + 8) * 256, 0, (void**)&kargs_dev));
  unsigned* err; CK(hsa_amd_memory_pool_allocate(g_karg_pool, 4096, 0, (void**)&err));
  CK(hsa_amd_agents_allow_access(1, &g_gpu, nullptr, err));
  unsigned *a0, *a1; CK(hsa_amd_memory_pool_allocate(g_dev_pool, 24576, 0, (void**)&a0)); CK(hsa_amd_memory_pool_allocate(g_dev_pool, 24576, 0, (void**)&a1));
  // the same operand pair in the device's fine-grained and extended-scope fine-grained ("uncached", hipDeviceMallocUncached) pools
  unsigned *f0 = nullptr, *f1 = nullptr, *u0 = nullptr, *u1 = nullptr;
  if (g_have_fine) { CK(hsa_amd_memory_pool_allocate(g_fine_pool, 24576, 0, (void**)&f0)); CK(hsa_amd_memory_pool_allocate(g_fine_pool, 24576, 0, (void**)&f1)); }
  if (g_have_ext) { CK(hsa_amd_memory_pool_allocate(g_ext_pool, 24576, 0, (void**)&u0)); CK(hsa_amd_memory_pool_allocate(g_ext_pool, 24576, 0, (void**)&u1)); }
  printf("device pools: fine-grained %d, extended-scope fine-grained %d\n", (int)g_have_fine, (int)g_have_ext);
  unsigned *c0 = a0, *c1 = a1;
  char* big; CK(hsa_amd_memory_pool_allocate(g_dev_pool, (size_t)512 << 20, 0, (void**)&big));      // 64 x 8 MB of weights: HBM-cold
  hsa_signal_t done; CK(hsa_signal_create(1, 0, nullptr, &done));
  const hsa_signal_t none = {0};

  struct PP { const unsigned* in; unsigned* out; unsigned* err; unsigned step; };
  struct Fill { unsigned* a; unsigned v; };
  const char* scope[3] = {"none", "agent", "system"};
  struct GV { const void* w; const void* a_in; void* a_out; };
  auto chain = [&](const char* what, const Kern& k, uint32_t grid, uint32_t wg, int barrier, int acq, int rel, bool pp, bool gv = false) -> int {
    *err = 0;
    if (pp) {   // operand a0 = 0 everywhere (system-scope fences around the fill)
      Fill* fa = reinterpret_cast<Fill*>(kargs + (size_t)TOT * 256); fa->a = a0; fa->v = 0;
      CK(hsa_memory_copy(kargs_dev + (size_t)TOT * 256, fa, 256));
      push(k_fill, 24, 256, kargs_dev + (size_t)TOT * 256, 1, 2, 2, done);
      ring_and_wait(done);
    }
    for (int i = 0; i < TOT; ++i) {
      void* ka = kargs + (size_t)i * 256;
      if (pp) { PP* p = reinterpret_cast<PP*>(ka); p->in = (i & 1) ? a1 : a0; p->out = (i & 1) ? a0 : a1; p->err = err; p->step = (unsigned)i; }
      else if (gv) { GV* p = reinterpret_cast<GV*>(ka); p->w = big + (size_t)(i % 64) * (8u << 20); p->a_in = (i & 1) ? a1 : a0; p->a_out = (i & 1) ? a0 : a1; }
      else *reinterpret_cast<int**>(ka) = nullptr;
    }
    CK(hsa_memory_copy(kargs_dev, kargs, (size_t)TOT * 256));
    for (int i = 0; i < TOT; ++i) {
      const bool last = i == TOT - 1;
      push(k, grid, wg, kargs_dev + (size_t)i * 256, barrier, last ? 2 : acq, last ? 2 : rel, last ? done : none);
    }
    const double us = ring_and_wait(done);
    printf("%-34s barrier %d  acquire %-6s release %-6s  %6.2f us per launch", what, barrier, scope[acq], scope[rel], us / TOT);
    if (pp) printf("   stale words seen: %u", *err);
    printf("\n");
    return 0;
  };
  for (int rep = 0; rep < 2; ++rep) {     // second pass = warm
    printf("--- pass %d\n", rep);
    for (int b = 1; b >= 0; --b)
      for (int s = 2; s >= 0; --s) chain("empty 1 x 64", k_empty, 1, 64, b, s, s, false);
    for (int s = 2; s >= 0; --s) chain("empty 256 x 512", k_empty, 256, 512, 1, s, s, false);
    chain("empty 256 x 512", k_empty, 256, 512, 1, 1, 0, false);
    chain("empty 256 x 512", k_empty, 256, 512, 1, 0, 1, false);
    for (int s = 2; s >= 0; --s) chain("ping-pong 128 x 256, plain ld/st", k_pp, 128, 256, 1, s, s, true);
    chain("ping-pong 128 x 256, plain ld/st", k_pp, 128, 256, 1, 1, 0, true);
    chain("ping-pong 128 x 256, plain ld/st", k_pp, 128, 256, 1, 0, 1, true);
    for (int s = 2; s >= 0; --s) chain("ping-pong 128 x 256, sc1 ld/st", k_pp1, 128, 256, 1, s, s, true);
    for (int kind = 0; kind < 2; ++kind) {      // operands in fine-grained / uncached device memory, plain loads and stores
      unsigned* p0 = kind ? u0 : f0; unsigned* p1 = kind ? u1 : f1;
      if (!p0) continue;
      a0 = p0; a1 = p1;
      const char* nm = kind ? "ping-pong plain, UNCACHED operand" : "ping-pong plain, FINE-GRAINED op.";
      chain(nm, k_pp, 128, 256, 1, 1, 1, true);
      chain(nm, k_pp, 128, 256, 1, 0, 0, true);
      const char* ng = kind ? "GEMV skeleton, UNCACHED operand" : "GEMV skeleton, FINE-GRAINED operand";
      chain(ng, k_gv, 128, 512, 1, 1, 1, false, true);
      chain(ng, k_gv, 128, 512, 1, 0, 0, false, true);
      a0 = c0; a1 = c1;
    }
    chain("GEMV skeleton, sc1 store + wait", k_gvw, 128, 512, 1, 1, 1, false, true);
    chain("GEMV skeleton, sc1 store + wait", k_gvw, 128, 512, 1, 1, 0, false, true);
    chain("GEMV skeleton, sc1 store + wait", k_gvw, 128, 512, 1, 0, 0, false, true);
    for (int s = 2; s >= 1; --s) chain("GEMV skeleton 128 x 512, 8 MB", k_gv, 128, 512, 1, s, s, false, true);
    chain("GEMV skeleton 128 x 512, 8 MB", k_gv, 128, 512, 1, 0, 1, false, true);
    chain("GEMV skeleton 128 x 512, 8 MB", k_gv, 128, 512, 1, 1, 0, false, true);
    chain("GEMV skeleton 128 x 512, 8 MB", k_gv, 128, 512, 1, 0, 0, false, true);
  }
  hsa_queue_destroy(g_q);
  hsa_shut_down();
  return 0;
}
