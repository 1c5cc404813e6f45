// Shared device helpers for the gfx950 kernels of libsaspa_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "saspa_hip.h"

typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define SASPA_CHECK_LAUNCH()                      \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)

__device__ __forceinline__ float bf16_bits_to_f32(uint32_t hi16) { return __builtin_bit_cast(float, hi16 << 16); }

// unpack 8 bf16 (one 16-byte chunk) to fp32
__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
  f[0] = __builtin_bit_cast(float, u.x << 16);
  f[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
  f[2] = __builtin_bit_cast(float, u.y << 16);
  f[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
  f[4] = __builtin_bit_cast(float, u.z << 16);
  f[5] = __builtin_bit_cast(float, u.z & 0xffff0000u);
  f[6] = __builtin_bit_cast(float, u.w << 16);
  f[7] = __builtin_bit_cast(float, u.w & 0xffff0000u);
}

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  // one v_cvt_pk_bf16_f32 (RNE, NaN-preserving); the scalar-cast form costs 2 cvt + an sdwa or
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 u;
  u.x = pack2(f[0], f[1]);
  u.y = pack2(f[2], f[3]);
  u.z = pack2(f[4], f[5]);
  u.w = pack2(f[6], f[7]);
  return u;
}

// ---- dtype-generic "vector of V contiguous elements" load/store as fp32 ----
template <typename T> struct Elem;
template <> struct Elem<bf16_t> {
  static constexpr int EPC = 8;  // elements per 16-byte chunk
  __device__ static __forceinline__ void load_chunk(const bf16_t* p, float* f) {
    uint4 u = *reinterpret_cast<const uint4*>(p);
    unpack8(u, f);
  }
  __device__ static __forceinline__ void store_chunk(bf16_t* p, const float* f) {
    *reinterpret_cast<uint4*>(p) = pack8(f);
  }
  __device__ static __forceinline__ void load4(const bf16_t* p, float* f) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    f[0] = __builtin_bit_cast(float, u.x << 16);
    f[1] = __builtin_bit_cast(float, u.x & 0xffff0000u);
    f[2] = __builtin_bit_cast(float, u.y << 16);
    f[3] = __builtin_bit_cast(float, u.y & 0xffff0000u);
  }
  __device__ static __forceinline__ void store4(bf16_t* p, const float* f) {
    uint2 u;
    u.x = pack2(f[0], f[1]);
    u.y = pack2(f[2], f[3]);
    *reinterpret_cast<uint2*>(p) = u;
  }
  __device__ static __forceinline__ float load1(const bf16_t* p) { return (float)*p; }
  __device__ static __forceinline__ void store1(bf16_t* p, float f) { *p = (bf16_t)f; }
};
template <> struct Elem<float> {
  static constexpr int EPC = 4;
  __device__ static __forceinline__ void load_chunk(const float* p, float* f) {
    float4 u = *reinterpret_cast<const float4*>(p);
    f[0] = u.x; f[1] = u.y; f[2] = u.z; f[3] = u.w;
  }
  __device__ static __forceinline__ void store_chunk(float* p, const float* f) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  }
  __device__ static __forceinline__ void load4(const float* p, float* f) { load_chunk(p, f); }
  __device__ static __forceinline__ void store4(float* p, const float* f) { store_chunk(p, f); }
  __device__ static __forceinline__ float load1(const float* p) { return *p; }
  __device__ static __forceinline__ void store1(float* p, float f) { *p = f; }
};

// fp32 storage whose GEMMs run as three bf16 MFMAs per product (x = hi + lo, hi = bf16(x), lo = bf16(x - hi);
// a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi, the a_lo b_lo term of relative size 2^-18 is dropped): SASPA_F32X3.
// Same size / layout as float, so every loader, epilogue and index computation is shared with the exact-fp32 path.
struct f32x3_t { float v; };
template <> struct Elem<f32x3_t> {
  static constexpr int EPC = 4;
  __device__ static __forceinline__ void load_chunk(const f32x3_t* p, float* f) { Elem<float>::load_chunk(reinterpret_cast<const float*>(p), f); }
  __device__ static __forceinline__ void store_chunk(f32x3_t* p, const float* f) { Elem<float>::store_chunk(reinterpret_cast<float*>(p), f); }
  __device__ static __forceinline__ void load4(const f32x3_t* p, float* f) { Elem<float>::load4(reinterpret_cast<const float*>(p), f); }
  __device__ static __forceinline__ void store4(f32x3_t* p, const float* f) { Elem<float>::store4(reinterpret_cast<float*>(p), f); }
  __device__ static __forceinline__ float load1(const f32x3_t* p) { return p->v; }
  __device__ static __forceinline__ void store1(f32x3_t* p, float f) { p->v = f; }
};

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
// SiLU for values that are rounded to bf16 right after: v_exp_f32 + v_rcp_f32 (each within 1 ulp of fp32) instead of expf's
// range handling and the IEEE division sequence (~16 VALU per element -> 5): the GroupNorm apply pass was VALU-bound on it
// (2.4 TB/s on every size, tools/gn_bench.py).  The fp32 parity path keeps silu_f.
__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// activation applied BEFORE the residual add (SiLU, ReLU) / AFTER it (ReLU of ResNet bottlenecks); see saspa_hip.h
__device__ __forceinline__ float act_pre(int act, float x) {
  return act == SASPA_ACT_SILU ? silu_f(x) : (act == SASPA_ACT_RELU ? fmaxf(x, 0.0f) : x);
}
__device__ __forceinline__ float act_post(int act, float x) { return act == SASPA_ACT_ADD_RELU ? fmaxf(x, 0.0f) : x; }

// erf for the bf16 GEGLU epilogue: Abramowitz-Stegun 7.1.26, |error| < 1.5e-7 (far below the
// bf16 output rounding 2^-9), one v_exp + one v_rcp + 6 FMAs instead of libm's ~50-instruction erff.
__device__ __forceinline__ float fast_erf(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
  const float r = fmaf(-poly, e, 1.0f);
  return copysignf(r, x);
}
__device__ __forceinline__ float fast_gelu_mul(float val, float gate) {
  return val * (0.5f * gate * (1.0f + fast_erf(gate * 0.70710678118654752440f)));
}

// The same arithmetic on two values at a time: the multiplies / FMAs of the polynomial are written on <2 x float> so that they become
// v_pk_mul_f32 / v_pk_fma_f32 (two IEEE results per issue slot on gfx950); every element sees exactly the operations of fast_gelu_mul
// in the same order, so the results are bit-identical.  (The transcendentals and the sign transfer stay per element.)
typedef float f32x2_pk __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_pk fast_gelu_mul2(f32x2_pk val, f32x2_pk gate) {
  const f32x2_pk x = gate * 0.70710678118654752440f;
  const f32x2_pk ax = __builtin_elementwise_abs(x);
  const f32x2_pk d = __builtin_elementwise_fma(f32x2_pk{0.3275911f, 0.3275911f}, ax, f32x2_pk{1.0f, 1.0f});
  const f32x2_pk t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
  f32x2_pk poly = __builtin_elementwise_fma(f32x2_pk{1.061405429f, 1.061405429f}, t, f32x2_pk{-1.453152027f, -1.453152027f});
  poly = __builtin_elementwise_fma(poly, t, f32x2_pk{1.421413741f, 1.421413741f});
  poly = __builtin_elementwise_fma(poly, t, f32x2_pk{-0.284496736f, -0.284496736f});
  poly = __builtin_elementwise_fma(poly, t, f32x2_pk{0.254829592f, 0.254829592f});
  poly *= t;
  const f32x2_pk q = -ax * ax * 1.4426950408889634f;
  const f32x2_pk e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
  const f32x2_pk r = __builtin_elementwise_fma(-poly, e, f32x2_pk{1.0f, 1.0f});
  const f32x2_pk er = {copysignf(r[0], x[0]), copysignf(r[1], x[1])};
  return val * (0.5f * gate * (1.0f + er));
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

// Raw buffer loads: 32-bit per-lane byte offset + scalar byte offset against a 128-bit
// descriptor; an offset >= num_records (2 GiB here) returns zeros -- that is how M / N / halo
// padding is produced with no branch and no zero-fill (kInvalid below).
using rsrc_t = __amdgpu_buffer_rsrc_t;
constexpr unsigned kInvalid = 0x80000000u;
__device__ __forceinline__ rsrc_t make_rsrc(const void* base) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
}
typedef __attribute__((address_space(3))) void lds_void_t;

// workgroup barrier that orders LDS accesses only.  __syncthreads() also waits for vmcnt(0): in an epilogue that means for every
// global store issued so far to be acknowledged by L2 -- the stores of one pass then drain before the next pass may stage.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// GroupNorm statistics of one block of <= 128 finished output rows held in LDS as bf16 (`ct`: rows of `cp` elements, the
// STORED values): per unit of `unit` consecutive channels (nunits of them, nunits divides NT) the sum and the sum of squares
// over rows [0, nrows) -> dst[nunits][2].  Fixed summation order (thread (unit, row group) over its interleaved rows, then
// one thread per value over the row groups): deterministic.  scratch: 2 * NT floats of LDS; ends with the caller's barrier.
template <int NT>
__device__ __forceinline__ void gn_tile_stats(const bf16_t* ct, const int cp, const int nrows, const int nunits, const int unit,
                                              float* scratch, float* dst) {
  const int tid = threadIdx.x;
  const int rgs = NT / nunits;
  const int u = tid % nunits, rg = tid / nunits;
  float sm = 0.f, sq = 0.f;
  if (rg < rgs) {
    for (int r = rg; r < nrows; r += rgs) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>(ct + r * cp + u * unit);
      for (int j = 0; j < unit; j += 2) {
        // two bf16 per dword: v_dot2c_f32_bf16 against (1, 1) and against itself -- 2 VALU per pair instead of 6
        const bf16x2_t w = __builtin_bit_cast(bf16x2_t, src[j >> 1]);
        sm = __builtin_amdgcn_fdot2_f32_bf16(w, __builtin_bit_cast(bf16x2_t, 0x3F803F80u), sm, false);
        sq = __builtin_amdgcn_fdot2_f32_bf16(w, w, sq, false);
      }
    }
    scratch[(rg * nunits + u) * 2] = sm;
    scratch[(rg * nunits + u) * 2 + 1] = sq;
  }
  lds_barrier();
  if (tid < nunits * 2) {
    const int uu = tid >> 1, k = tid & 1;
    float a = 0.f;
    for (int g = 0; g < rgs; ++g) a += scratch[(g * nunits + uu) * 2 + k];
    dst[uu * 2 + k] = a;
  }
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
