// GroupNorm (NHWC, optional two-source channel concat, optional SiLU) and LayerNorm.
// HBM-bound: every access is a 16/32-byte vector per lane (8 channels), statistics in fp32
// per thread, combined in fp64 by the finalize step.  See include/saspa_hip.h.
#include "common.h"

namespace {

// ---- stage 1: per-(batch, split, channel) sum / sum of squares -------------------------
// grid = (nsplit, batch, slabs); block = 256 threads laid out as cxw chunk-columns x
// (256/cxw) pixel rows; a "chunk" is 8 consecutive channels.
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const SaspaGroupNormParams p, int cxw, int pix_per_split) {
  __shared__ float red[256 * 16];
  const int tid = threadIdx.x;
  const int cx = tid % cxw, py = tid / cxw;
  const int rows = 256 / cxw;
  const int C = p.c0 + p.c1;
  const int C8 = C >> 3;
  const int chunk = blockIdx.z * cxw + cx;
  const int b = blockIdx.y, split = blockIdx.x;
  float s[8], ss[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.f; ss[j] = 0.f; }
  if (chunk < C8) {
    const int ch = chunk * 8;
    const T* src;
    int ld, cc;
    if (ch < p.c0) { src = reinterpret_cast<const T*>(p.x0); ld = p.ldx0; cc = ch; }
    else { src = reinterpret_cast<const T*>(p.x1); ld = p.ldx1; cc = ch - p.c0; }
    const int pbeg = split * pix_per_split;
    const int pend = min(p.hw, pbeg + pix_per_split);
    for (int px = pbeg + py; px < pend; px += rows) {
      const T* ptr = src + ((long long)b * p.hw + px) * ld + cc;
      float v[8];
      if constexpr (sizeof(T) == 2) {
        Elem<T>::load_chunk(ptr, v);
      } else {
        Elem<T>::load_chunk(ptr, v);
        Elem<T>::load_chunk(ptr + 4, v + 4);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] += v[j]; ss[j] += v[j] * v[j]; }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[tid * 16 + j] = s[j]; red[tid * 16 + 8 + j] = ss[j]; }
  __syncthreads();
  // each (chunk column, value) pair is reduced over the pixel rows by one thread
  for (int wi = tid; wi < cxw * 16; wi += 256) {
    const int col = wi / 16, val = wi % 16;
    float a = 0.f;
    for (int r = 0; r < rows; ++r) a += red[(r * cxw + col) * 16 + val];
    const int chunk2 = blockIdx.z * cxw + col;
    if (chunk2 < C8) {
      const int ch = chunk2 * 8 + (val & 7);
      float* dst = p.partial + (((long long)b * p.nsplit + split) * C + ch) * 2 + (val >> 3);
      *dst = a;
    }
  }
}

// ---- stage 2: one wave per (batch, group): fp64 combine -> scale/shift[b][c] --------------
__global__ __launch_bounds__(256) void gn_finalize_kernel(const SaspaGroupNormParams p) {
  const int lane = threadIdx.x & 63;
  const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int b = blockIdx.y;
  if (g >= p.groups) return;
  const int C = p.c0 + p.c1;
  const int cpg = C / p.groups;
  double s = 0.0, ss = 0.0;
  const int items = cpg * p.nsplit;
  for (int it = lane; it < items; it += 64) {
    const int sp = it / cpg, cj = it - sp * cpg;
    const float* src = p.partial + (((long long)b * p.nsplit + sp) * C + g * cpg + cj) * 2;
    s += (double)src[0];
    ss += (double)src[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    ss += __shfl_xor(ss, o, 64);
  }
  const double n = (double)cpg * (double)p.hw;
  const double mean = s / n;
  double var = ss / n - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
  const float meanf = (float)mean;
  float* sc = p.scale_shift + (long long)b * 2 * C;
  float* sh = sc + C;
  for (int cj = lane; cj < cpg; cj += 64) {
    const int ch = g * cpg + cj;
    const float ga = p.gamma[ch] * rstd;
    sc[ch] = ga;
    sh[ch] = p.beta[ch] - meanf * ga;
  }
}

// ---- apply: y = act(x * scale[b][c] + shift[b][c]) -----------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const SaspaGroupNormParams p) {
  const int C = p.c0 + p.c1;
  const int C8 = C >> 3;
  const long long total = (long long)p.batch * p.hw * C8;
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const long long pix = it / C8;
    const int chunk = (int)(it - pix * C8);
    const int b = (int)(pix / p.hw);
    const int ch = chunk * 8;
    const T* src;
    int ld, cc;
    if (ch < p.c0) { src = reinterpret_cast<const T*>(p.x0); ld = p.ldx0; cc = ch; }
    else { src = reinterpret_cast<const T*>(p.x1); ld = p.ldx1; cc = ch - p.c0; }
    const T* ptr = src + pix * ld + cc;
    float v[8];
    if constexpr (sizeof(T) == 2) {
      Elem<T>::load_chunk(ptr, v);
    } else {
      Elem<T>::load_chunk(ptr, v);
      Elem<T>::load_chunk(ptr + 4, v + 4);
    }
    const float* sc = p.scale_shift + (long long)b * 2 * C + ch;
    const float* sh = sc + C;
    const float4 s0 = *reinterpret_cast<const float4*>(sc), s1 = *reinterpret_cast<const float4*>(sc + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(sh), h1 = *reinterpret_cast<const float4*>(sh + 4);
    const float scv[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float shv[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float y = v[j] * scv[j] + shv[j];
      if (p.act == SASPA_ACT_SILU) y = silu_f(y);
      v[j] = y;
    }
    T* dst = reinterpret_cast<T*>(p.y) + pix * p.ldy + ch;
    if constexpr (sizeof(T) == 2) {
      Elem<T>::store_chunk(dst, v);
    } else {
      Elem<T>::store_chunk(dst, v);
      Elem<T>::store_chunk(dst + 4, v + 4);
    }
  }
}

// ---- LayerNorm: one wave per row, two-pass in registers ---------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* x, int ldx, T* y, int ldy, long long rows, int C,
                                                        const float* gamma, const float* beta, float eps) {
  constexpr int MAXCH = 4;  // C <= 2048
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int C8 = C >> 3;
  const T* xr = x + row * ldx;
  float v[MAXCH][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int chunk = lane + 64 * i;
    if (chunk < C8) {
      if constexpr (sizeof(T) == 2) {
        Elem<T>::load_chunk(xr + chunk * 8, v[i]);
      } else {
        Elem<T>::load_chunk(xr + chunk * 8, v[i]);
        Elem<T>::load_chunk(xr + chunk * 8 + 4, v[i] + 4);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[i][j];
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int chunk = lane + 64 * i;
    if (chunk < C8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; sq += d * d; }
    }
  }
  const float var = wave_sum(sq) / (float)C;
  const float rstd = 1.0f / sqrtf(var + eps);
  T* yr = y + row * ldy;
#pragma unroll
  for (int i = 0; i < MAXCH; ++i) {
    const int chunk = lane + 64 * i;
    if (chunk < C8) {
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (v[i][j] - mean) * rstd * gamma[chunk * 8 + j] + beta[chunk * 8 + j];
      if constexpr (sizeof(T) == 2) {
        Elem<T>::store_chunk(yr + chunk * 8, o);
      } else {
        Elem<T>::store_chunk(yr + chunk * 8, o);
        Elem<T>::store_chunk(yr + chunk * 8 + 4, o + 4);
      }
    }
  }
}

int check_gn(const SaspaGroupNormParams& p) {
  if (!p.x0 || !p.gamma || !p.beta || !p.partial || !p.scale_shift) return SASPA_EINVAL;
  if (p.batch <= 0 || p.hw <= 0 || p.groups <= 0 || p.nsplit <= 0 || p.c0 <= 0 || p.c1 < 0) return SASPA_EINVAL;
  if (p.c1 > 0 && !p.x1) return SASPA_EINVAL;
  if (p.dtype != SASPA_BF16 && p.dtype != SASPA_F32) return SASPA_EINVAL;
  const int C = p.c0 + p.c1;
  if (p.c0 % 8 || p.c1 % 8 || p.ldx0 % 8 || (p.c1 > 0 && p.ldx1 % 8)) return SASPA_EALIGN;
  if (C % p.groups) return SASPA_ERANGE;
  if (!aligned16(p.x0) || (p.x1 && !aligned16(p.x1)) || !aligned16(p.scale_shift)) return SASPA_EALIGN;
  if (p.nsplit > p.hw) return SASPA_ERANGE;
  return 0;
}

}  // namespace

extern "C" int saspa_groupnorm_stats(const SaspaGroupNormParams* pp, void* stream) {
  if (!pp) return SASPA_EINVAL;
  const SaspaGroupNormParams& p = *pp;
  if (int e = check_gn(p)) return e;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int C8 = (p.c0 + p.c1) / 8;
  const int cxw = C8 >= 32 ? 32 : 16;
  const int slabs = (C8 + cxw - 1) / cxw;
  const int pps = (p.hw + p.nsplit - 1) / p.nsplit;
  dim3 grid(p.nsplit, p.batch, slabs);
  if (p.dtype == SASPA_BF16) hipLaunchKernelGGL(gn_partial_kernel<bf16_t>, grid, dim3(256), 0, s, p, cxw, pps);
  else hipLaunchKernelGGL(gn_partial_kernel<float>, grid, dim3(256), 0, s, p, cxw, pps);
  SASPA_CHECK_LAUNCH();
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((p.groups + 3) / 4, p.batch), dim3(256), 0, s, p);
  SASPA_CHECK_LAUNCH();
  return 0;
}

extern "C" int saspa_groupnorm_apply(const SaspaGroupNormParams* pp, void* stream) {
  if (!pp) return SASPA_EINVAL;
  const SaspaGroupNormParams& p = *pp;
  if (int e = check_gn(p)) return e;
  if (!p.y || p.ldy % 8 || !aligned16(p.y)) return SASPA_EALIGN;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const long long total = (long long)p.batch * p.hw * ((p.c0 + p.c1) / 8);
  long long blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (p.dtype == SASPA_BF16) hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, s, p);
  else hipLaunchKernelGGL(gn_apply_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, p);
  SASPA_CHECK_LAUNCH();
  return 0;
}

extern "C" int saspa_layernorm(int dtype, const void* x, int ldx, void* y, int ldy, long long rows, int C,
                               const float* gamma, const float* beta, float eps, void* stream) {
  if (!x || !y || !gamma || !beta || rows <= 0 || C <= 0) return SASPA_EINVAL;
  if (C % 8 || ldx % 8 || ldy % 8 || !aligned16(x) || !aligned16(y)) return SASPA_EALIGN;
  if (C > 2048) return SASPA_ERANGE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned grid = (unsigned)((rows + 3) / 4);
  if (dtype == SASPA_BF16)
    hipLaunchKernelGGL(layernorm_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, rows, C, gamma, beta, eps);
  else if (dtype == SASPA_F32)
    hipLaunchKernelGGL(layernorm_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)x, ldx, (float*)y, ldy, rows, C, gamma, beta, eps);
  else
    return SASPA_EINVAL;
  SASPA_CHECK_LAUNCH();
  return 0;
}
