// Shared device helpers for the Dia decode-path kernels (gfx950 / CDNA4, wave64).
//
// Data formats used between kernels:
//   * weight tiles   bf16 [strip = n/16][ktile = k/32][lane 0..63][8]   — the B operand of
//                    v_mfma_f32_16x16x32_bf16: lane l holds W[k = 32*ktile + 8*(l>>4) + j][n = 16*strip + (l&15)]
//   * activation planes  bf16 [plane 0..2][mtile = m/16][ktile][lane][8] — the A operand of the same
//                    instruction: lane l holds X[m = 16*mtile + (l&15)][k = 32*ktile + 8*(l>>4) + j].
//                    An fp32 activation v travels as three bf16 planes hi+mid+lo == v (24 significand
//                    bits), so bf16 MFMA products against bf16 weights are exact and accumulate in fp32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef uint16_t bf16_raw;

#define DIA_WAVE 64
#define DIA_NPLANES 3

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t h) { return __uint_as_float(h << 16); }

// v == hi + mid + lo exactly, each bf16 (round-to-nearest-even at every stage; both residuals are
// exact in fp32).  Contraction is off here and in mul_rn(): fusing a preceding multiply into the
// subtraction would split the unrounded product instead of the fp32 value v.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ void split3(float v, __bf16& hi, __bf16& mid, __bf16& lo) {
#pragma clang fp contract(off)
  hi = (__bf16)v;
  const float r = v - (float)hi;
  mid = (__bf16)r;
  const float r2 = r - (float)mid;
  lo = (__bf16)r2;
}

// element offset of the 8-element fragment holding X[m][k0..k0+7] (k0 % 8 == 0) inside one plane
__device__ __forceinline__ long plane_frag_off(int m, int k0, int ktiles) {
  int mt = m >> 4, kt = k0 >> 5, kq = (k0 & 31) >> 3;
  int lane = (m & 15) + 16 * kq;
  return (((long)mt * ktiles + kt) * 64 + lane) * 8;
}

// write 8 consecutive-k activations of row m as three plane fragments (16 B each)
__device__ __forceinline__ void emit_planes8(bf16_raw* P, long plane_stride, int ktiles, int m, int k0,
                                             const float* v) {
  bf16x8 h, mi, lo;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __bf16 a, b, c;
    split3(v[j], a, b, c);
    h[j] = a; mi[j] = b; lo[j] = c;
  }
  long off = plane_frag_off(m, k0, ktiles);
  *reinterpret_cast<bf16x8*>(P + off) = h;
  *reinterpret_cast<bf16x8*>(P + plane_stride + off) = mi;
  *reinterpret_cast<bf16x8*>(P + 2 * plane_stride + off) = lo;
}

// ---- fp32 activation tiles ("act_f32"): the same [mtile][ktile][lane][8] fragment order as one plane, 4-byte elements
// (4 bytes per value instead of the 6 of three planes).  The 5..128-row GEMM holds its A fragments in registers and every
// workgroup pulls the whole activation matrix (16 rows x K): at batch 8 that image is 192 KB per workgroup against 64 KB of
// weights, and it is what the head of each launch waits for.  The consumer splits each value into its three bf16 planes in
// registers (hi + mid + lo == v exactly, as the producer used to), so the arithmetic is unchanged bit for bit.
__device__ __forceinline__ void emit_f32x8(float* P, int ktiles, int m, int k0, const float* v) {
  float* o = P + plane_frag_off(m, k0, ktiles);
  *reinterpret_cast<float4*>(o) = float4{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<float4*>(o + 4) = float4{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ void emit_f32x8_mapped(float* P, int ktiles, int m, int n0, const float* v, const int* cmap) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cmap[n0 + j];
    if (c >= 0) P[plane_frag_off(m, c & ~7, ktiles) + (c & 7)] = v[j];
  }
}
// 8 fp32 values -> the three bf16x8 MFMA fragments their planes would have held
__device__ __forceinline__ void split3x8(const float4 x, const float4 y, bf16x8& h, bf16x8& mi, bf16x8& lo) {
  const float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    __bf16 a, b, c;
    split3(v[j], a, b, c);
    h[j] = a; mi[j] = b; lo[j] = c;
  }
}

// Workgroup barrier for LDS hand-offs that does NOT drain outstanding global loads: __syncthreads()
// makes hipcc wait vmcnt(0), which would serialise the weight stream behind every barrier (guide §5
// 'Pipelining across barriers').  LDS operations are waited for explicitly.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Device-coherent 8-byte accesses (agent-scope relaxed atomics = sc1 global_store/load_dwordx2): written
// through to / read from the point where all XCDs agree, without any cache-wide fence.
__device__ __forceinline__ void st2_agent(float* p, float a, float b) {
  const unsigned long long v = ((unsigned long long)__float_as_uint(b) << 32) | (unsigned long long)__float_as_uint(a);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float2 ld2_agent(const float* p) {
  const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return float2{__uint_as_float((unsigned)(v & 0xffffffffull)), __uint_as_float((unsigned)(v >> 32))};
}

// 16-byte device-coherent accesses: buffer_load / buffer_store_dwordx4 sc1 through the raw-buffer builtins (cache policy
// bit 4 = sc1 on gfx94x/gfx950), which the compiler tracks like any other load — there is no 128-bit atomic to lower to,
// and hand-written global_load asm is invisible to its wait-count insertion.  One 16-byte access replaces two 8-byte
// ones: the write-through / coherent-read paths handle 8-byte requests at 0.5-0.7x the 16-byte rate per byte.
// `base` must be wave-uniform (a kernel argument); offsets are bytes below 2 GiB.
typedef unsigned int dia_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t agent_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ f32x4 ld4_agent(__amdgpu_buffer_rsrc_t r, int byte_off) {
  return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, byte_off, 0, 16));
}
__device__ __forceinline__ void st4_agent(__amdgpu_buffer_rsrc_t r, int byte_off, f32x4 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(dia_u32x4, v), r, byte_off, 0, 16);
}

// compacted consumers: column n0+j goes to plane position cmap[n0+j] (2-byte stores), or nowhere
__device__ __forceinline__ void emit_planes8_mapped(bf16_raw* P, long plane_stride, int ktiles, int m, int n0,
                                                    const float* v, const int* cmap) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cmap[n0 + j];
    if (c >= 0) {
      __bf16 a, b, d;
      split3(v[j], a, b, d);
      const long off = plane_frag_off(m, c & ~7, ktiles) + (c & 7);
      P[off] = *reinterpret_cast<bf16_raw*>(&a);
      P[plane_stride + off] = *reinterpret_cast<bf16_raw*>(&b);
      P[2 * plane_stride + off] = *reinterpret_cast<bf16_raw*>(&d);
    }
  }
}

// sum over each aligned row of 16 lanes, result in every lane of the row — four DPP-modified adds
// (quad xor 1, quad xor 2, half-row mirror, row mirror) instead of four LDS-crossbar bpermutes
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

template <typename T> struct KVElem;
template <> struct KVElem<float> {
  static __device__ __forceinline__ void load8(const float* p, float* o) {
    float4 a = *reinterpret_cast<const float4*>(p);
    float4 b = *reinterpret_cast<const float4*>(p + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
  }
  static __device__ __forceinline__ float round(float v) { return v; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
};
template <> struct KVElem<bf16_raw> {
  static __device__ __forceinline__ void load8(const bf16_raw* p, float* o) {
    uint4 u = *reinterpret_cast<const uint4*>(p);
    o[0] = __uint_as_float(u.x << 16); o[1] = __uint_as_float(u.x & 0xFFFF0000u);
    o[2] = __uint_as_float(u.y << 16); o[3] = __uint_as_float(u.y & 0xFFFF0000u);
    o[4] = __uint_as_float(u.z << 16); o[5] = __uint_as_float(u.z & 0xFFFF0000u);
    o[6] = __uint_as_float(u.w << 16); o[7] = __uint_as_float(u.w & 0xFFFF0000u);
  }
  static __device__ __forceinline__ float round(float v) { return (float)(__bf16)v; }
  static __device__ __forceinline__ void store(bf16_raw* p, float v) {
    __bf16 b = (__bf16)v;
    *p = *reinterpret_cast<bf16_raw*>(&b);
  }
};
