// GroupNorm (NHWC, optional two-source channel concat, optional SiLU) and LayerNorm.
// HBM-bound: every access is a 16/32-byte vector per lane (8 channels).  GroupNorm is two launches: per-(image, pixel
// split, channel slab, group) fp32 sums, then the apply pass, whose every workgroup first combines the (few KB of) sums
// of ITS image in fp64 -- no finalize launch in between.  See include/saspa_hip.h.
#include <cstdlib>

#include "common.h"
#include "gemm_internal.h"

namespace {

// Thread layout shared by both passes: a block is cxw chunk-columns (a "chunk" = 8 consecutive channels) x rows pixel
// rows, rows = 256 / cxw (threads beyond rows * cxw idle); blockIdx.z picks the slab of cxw chunks.  cxw divides C/8
// whenever C/8 has a divisor in [16, 128] that keeps >= 90 % of the threads busy (40 for 320 channels, 80 for 640 /
// 1280 / 2560, 120 for 960 / 1920, ...), so no slab is ragged on the shapes that matter.
struct GnGeom { int cxw, slabs; };
GnGeom gn_geometry(int C8) {
  if (C8 <= 16) return {C8, 1};
  for (int d = 128; d >= 16; --d)
    if (C8 % d == 0 && (256 / d) * d >= 230) return {d, C8 / d};
  return {32, (C8 + 31) / 32};
}

// ---- stage 1: sum / sum of squares per (image, split, slab, group) --------------------------------
// grid = (nsplit, batch, slabs).  partial[((b * nsplit + split) * slabs + slab) * groups + g] = (sum, sumsq) over the
// block's pixels and the channels of group g that lie in the slab (zero if none).
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const SaspaGroupNormParams p, int cxw, int pix_per_split) {
  __shared__ float red[256 * 16];
  __shared__ float chs[2][128 * 8];
  const int tid = threadIdx.x;
  const int rows = 256 / cxw;
  const bool active = tid < rows * cxw;
  const int cx = tid % cxw, py = tid / cxw;
  const int C = p.c0 + p.c1;
  const int C8 = C >> 3;
  const int chunk = blockIdx.z * cxw + cx;
  const int b = blockIdx.y, split = blockIdx.x;
  float s[8], ss[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[j] = 0.f; ss[j] = 0.f; }
  if (active && chunk < C8) {
    const int ch = chunk * 8;
    const T* src;
    int ld, cc;
    if (ch < p.c0) { src = reinterpret_cast<const T*>(p.x0); ld = p.ldx0; cc = ch; }
    else { src = reinterpret_cast<const T*>(p.x1); ld = p.ldx1; cc = ch - p.c0; }
    const int pbeg = split * pix_per_split;
    const int pend = min(p.hw, pbeg + pix_per_split);
    // four pixels per trip: four independent 16-byte loads in flight per lane (the pass is latency / bandwidth bound)
    int px = pbeg + py;
    for (; px + 3 * rows < pend; px += 4 * rows) {
      float v[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const T* ptr = src + ((long long)b * p.hw + px + u * rows) * ld + cc;
        if constexpr (sizeof(T) == 2) {
          Elem<T>::load_chunk(ptr, v[u]);
        } else {
          Elem<T>::load_chunk(ptr, v[u]);
          Elem<T>::load_chunk(ptr + 4, v[u] + 4);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] += v[u][j]; ss[j] += v[u][j] * v[u][j]; }
    }
    for (; px < pend; px += rows) {
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
  // each (chunk column, value) pair is reduced over the pixel rows by one thread -> per-channel sums of the slab in LDS
  for (int wi = tid; wi < cxw * 16; wi += 256) {
    const int col = wi / 16, val = wi % 16;
    float a = 0.f;
    for (int r = 0; r < rows; ++r) a += red[(r * cxw + col) * 16 + val];
    chs[val >> 3][col * 8 + (val & 7)] = a;
  }
  __syncthreads();
  // channels -> groups (the part of each group inside this slab), one thread per group
  const int cpg = C / p.groups;
  const int ch0 = blockIdx.z * cxw * 8, ch1 = min(C, ch0 + cxw * 8);
  float* dst = p.partial + (((long long)b * p.nsplit + split) * gridDim.z + blockIdx.z) * p.groups * 2;
  for (int g = tid; g < p.groups; g += 256) {
    const int lo = max(g * cpg, ch0), hi = min((g + 1) * cpg, ch1);
    float a = 0.f, q = 0.f;
    for (int c = lo; c < hi; ++c) { a += chs[0][c - ch0]; q += chs[1][c - ch0]; }
    dst[2 * g] = a;
    dst[2 * g + 1] = q;
  }
}

// ---- stage 2: y = act(x * scale[b][c] + shift[b][c]) -----------------------------------------
// grid = (nblk, batch, slabs), the block's pixels are [blockIdx.x * ppb, +ppb).  Prologue: the 32 (groups) x nsplit x slabs
// partial sums of image b -> mean / rstd per group (fp64 combine, 8 threads per group), then each thread derives
// scale / shift of its own 8 channels once; the main loop is a pure stream.
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const SaspaGroupNormParams p, int cxw, int slabs, int ppb) {
  __shared__ float2 stat[256];
  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int C = p.c0 + p.c1;
  const int C8 = C >> 3;
  const int cpg = C / p.groups;
  {
    const int sub = tid & 7;
    const int nparts = p.nsplit * slabs;
    const float* base = p.partial + (long long)b * nparts * p.groups * 2;
    for (int g0 = 0; g0 < p.groups; g0 += 32) {
      const int g = g0 + (tid >> 3);
      double sm = 0.0, sq = 0.0;
      if (g < p.groups && p.stats0) {
        // statistics left by the producers' epilogues (SaspaGemmParams.gn_stats): per (128-row block, unit of `unit` channels)
        // sums of each source; group g = the units of its channel range in source 0 and / or source 1, over the hw / 128 row
        // blocks of image b.  Same 8-loads-in-flight shape and fp64 combine as the partial-sum path below.
        const int nblk = p.hw >> 7;
        const int glo = g * cpg, ghi = glo + cpg;
        for (int src = 0; src < 2; ++src) {
          const float* st = src ? p.stats1 : p.stats0;
          const int cs = src ? p.c1 : p.c0, off = src ? p.c0 : 0;
          if (!st || cs == 0) continue;
          const int lo = max(glo, off) - off, hi = min(ghi, off + cs) - off;
          if (hi <= lo) continue;
          const int u0 = lo / p.unit, nu = (hi - lo) / p.unit, upr = cs / p.unit;
          const int total = nu * nblk;
          for (int e0 = sub; e0 < total; e0 += 64) {
            float2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              const int e = e0 + 8 * u;
              v[u] = make_float2(0.f, 0.f);
              if (e < total) {
                const int blk = e / nu, uu = e - blk * nu;
                v[u] = *reinterpret_cast<const float2*>(st + (((long long)b * nblk + blk) * upr + u0 + uu) * 2);
              }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
              sm += (double)v[u].x;
              sq += (double)v[u].y;
            }
          }
        }
      } else if (g < p.groups) {
        // 8 loads in flight per thread and round (64 partial sums per group and round): one L2 round trip for the usual
        // nsplit * slabs <= 64, instead of one per element
        for (int i0 = sub; i0 < nparts; i0 += 64) {
          float2 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            const int i = i0 + 8 * u;
            v[u] = make_float2(0.f, 0.f);
            if (i < nparts) v[u] = *reinterpret_cast<const float2*>(base + ((long long)i * p.groups + g) * 2);
          }
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            sm += (double)v[u].x;
            sq += (double)v[u].y;
          }
        }
      }
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        sm += __shfl_xor(sm, o, 64);
        sq += __shfl_xor(sq, o, 64);
      }
      if (g < p.groups && sub == 0) {
        const double n = (double)cpg * (double)p.hw;
        const double mean = sm / n;
        double var = sq / n - mean * mean;
        if (var < 0.0) var = 0.0;
        stat[g] = make_float2((float)mean, (float)(1.0 / sqrt(var + (double)p.eps)));
      }
    }
  }
  __syncthreads();
  const int rows = 256 / cxw;
  if (tid >= rows * cxw) return;
  const int cx = tid % cxw, py = tid / cxw;
  const int chunk = blockIdx.z * cxw + cx;
  if (chunk >= C8) return;
  const int ch = chunk * 8;
  float scv[8], shv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float2 st = stat[(ch + j) / cpg];
    scv[j] = p.gamma[ch + j] * st.y;
    shv[j] = p.beta[ch + j] - st.x * scv[j];
  }
  const T* src;
  int ld, cc;
  if (ch < p.c0) { src = reinterpret_cast<const T*>(p.x0); ld = p.ldx0; cc = ch; }
  else { src = reinterpret_cast<const T*>(p.x1); ld = p.ldx1; cc = ch - p.c0; }
  const int pbeg = blockIdx.x * ppb;
  const int pend = min(p.hw, pbeg + ppb);
  const bool silu = p.act == SASPA_ACT_SILU;
  auto one = [&](const int px) __attribute__((always_inline)) {
    const long long pix = (long long)b * p.hw + px;
    const T* ptr = src + pix * ld + cc;
    float v[8];
    if constexpr (sizeof(T) == 2) {
      Elem<T>::load_chunk(ptr, v);
    } else {
      Elem<T>::load_chunk(ptr, v);
      Elem<T>::load_chunk(ptr + 4, v + 4);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float y = v[j] * scv[j] + shv[j];
      if (silu) y = sizeof(T) == 2 ? silu_fast(y) : silu_f(y);
      v[j] = y;
    }
    T* dst = reinterpret_cast<T*>(p.y) + pix * p.ldy + ch;
    if constexpr (sizeof(T) == 2) {
      Elem<T>::store_chunk(dst, v);
    } else {
      Elem<T>::store_chunk(dst, v);
      Elem<T>::store_chunk(dst + 4, v + 4);
    }
  };
  // four pixels per trip: the compiler issues the four loads back to back before the first use
  int px = pbeg + py;
  for (; px + 3 * rows < pend; px += 4 * rows) {
    float v[4][8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const T* ptr = src + ((long long)b * p.hw + px + u * rows) * ld + cc;
      if constexpr (sizeof(T) == 2) {
        Elem<T>::load_chunk(ptr, v[u]);
      } else {
        Elem<T>::load_chunk(ptr, v[u]);
        Elem<T>::load_chunk(ptr + 4, v[u] + 4);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float y = v[u][j] * scv[j] + shv[j];
        if (silu) y = sizeof(T) == 2 ? silu_fast(y) : silu_f(y);
        v[u][j] = y;
      }
      T* dst = reinterpret_cast<T*>(p.y) + ((long long)b * p.hw + px + u * rows) * p.ldy + ch;
      if constexpr (sizeof(T) == 2) {
        Elem<T>::store_chunk(dst, v[u]);
      } else {
        Elem<T>::store_chunk(dst, v[u]);
        Elem<T>::store_chunk(dst + 4, v[u] + 4);
      }
    }
  }
  for (; px < pend; px += rows) one(px);
}

// ---- LayerNorm: LPR lanes per row (64 / LPR rows per wave), NCH chunks of 8 channels per lane, two-pass in registers.
// Narrow rows (320 / 640 channels = 40 / 80 chunks) take 16 / 32 lanes x 3 chunks, so a wave has 4 / 2 rows' loads in flight
// and 83 % of its lanes busy instead of one row and 62 %.
template <typename T, int LPR, int NCH>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* x, int ldx, T* y, int ldy, long long rows, int C,
                                                        const float* gamma, const float* beta, float eps) {
  constexpr int RPW = 64 / LPR;
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPR;
  const long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / LPR;
  const bool live = row < rows;
  const int C8 = C >> 3;
  const T* xr = x + (live ? row : 0) * ldx;
  float v[NCH][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int chunk = sub + LPR * i;
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
  auto row_sum = [](float a) __attribute__((always_inline)) {
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    return a;
  };
  const float mean = row_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int chunk = sub + LPR * i;
    if (chunk < C8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[i][j] - mean; sq += d * d; }
    }
  }
  const float var = row_sum(sq) / (float)C;
  const float rstd = 1.0f / sqrtf(var + eps);
  if (!live) return;
  T* yr = y + row * ldy;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int chunk = sub + LPR * i;
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

template <typename T>
void launch_layernorm(hipStream_t s, const T* x, int ldx, T* y, int ldy, long long rows, int C, const float* gamma, const float* beta,
                      float eps) {
  const int C8 = C >> 3;
  static const bool narrow = !(getenv("SASPA_LN_NARROW") && atoi(getenv("SASPA_LN_NARROW")) == 0);   // A/B knob
  if (narrow && C8 <= 48) {
    hipLaunchKernelGGL((layernorm_kernel<T, 16, 3>), dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, s, x, ldx, y, ldy, rows, C, gamma, beta, eps);
  } else if (narrow && C8 <= 96) {
    hipLaunchKernelGGL((layernorm_kernel<T, 32, 3>), dim3((unsigned)((rows + 7) / 8)), dim3(256), 0, s, x, ldx, y, ldy, rows, C, gamma, beta, eps);
  } else {
    hipLaunchKernelGGL((layernorm_kernel<T, 64, 4>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, ldx, y, ldy, rows, C, gamma, beta, eps);
  }
}

int check_gn(const SaspaGroupNormParams& p) {
  if (!p.x0 || !p.gamma || !p.beta || (!p.partial && !p.stats0)) return SASPA_EINVAL;
  if (p.stats0) {
    // epilogue statistics (ABI 12): whole 128-row blocks per image, groups and sources made of whole units
    const int C_ = p.c0 + p.c1;
    if (p.unit <= 0 || p.groups <= 0 || C_ % p.groups) return SASPA_ERANGE;
    if ((p.hw & 127) || p.c0 % p.unit || p.c1 % p.unit || (C_ / p.groups) % p.unit) return SASPA_ERANGE;
    if (p.c1 > 0 && !p.stats1) return SASPA_EINVAL;
    if ((reinterpret_cast<uintptr_t>(p.stats0) & 7u) || (p.stats1 && (reinterpret_cast<uintptr_t>(p.stats1) & 7u))) return SASPA_EALIGN;
  }
  if (p.batch <= 0 || p.hw <= 0 || p.groups <= 0 || (p.nsplit <= 0 && !p.stats0) || p.c0 <= 0 || p.c1 < 0) return SASPA_EINVAL;
  if (p.c1 > 0 && !p.x1) return SASPA_EINVAL;
  if (p.dtype != SASPA_BF16 && p.dtype != SASPA_F32) return SASPA_EINVAL;
  const int C = p.c0 + p.c1;
  if (p.c0 % 8 || p.c1 % 8 || p.ldx0 % 8 || (p.c1 > 0 && p.ldx1 % 8)) return SASPA_EALIGN;
  if (C % p.groups || p.groups > 256) return SASPA_ERANGE;
  if (!aligned16(p.x0) || (p.x1 && !aligned16(p.x1)) || (reinterpret_cast<uintptr_t>(p.partial) & 7u)) return SASPA_EALIGN;
  if (p.batch > 65535) return SASPA_ERANGE;
  if (p.stats0) return 0;
  if (p.nsplit > p.hw) return SASPA_ERANGE;
  // the workspace contract stays batch * nsplit * C * 2 floats; the per-group sums need slabs * groups <= C of it
  if ((long long)gn_geometry(C / 8).slabs * p.groups > C) return SASPA_ERANGE;
  return 0;
}

// ---- small images: statistics + apply in ONE launch ------------------------------------------------------------------
// At the 8x8 level (hw = 64: no 128-row statistics blocks, so no epilogue statistics either) a GroupNorm was two launches of a
// few microseconds of work each -- 24 pairs per UNet + ControlNet evaluation, all launch latency.  Here one workgroup owns one
// (image, group): its hw x cpg values (64 x 40 = 5 KB at the 8x8 level) are read ONCE into registers, reduced over the workgroup
// (fp32 partial sums per thread, fp64 combine), normalised from the registers and written: one read, one write, one launch.
// items = hw * cpg / 8 chunks of 8 channels, at most GN1_ITEMS per thread.
constexpr int GN1_ITEMS = 4;
template <typename T>
__global__ __launch_bounds__(256) void gn_onepass_kernel(const SaspaGroupNormParams p) {
  __shared__ double red[2][4];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, g = blockIdx.x;
  const int C = p.c0 + p.c1;
  const int cpg = C / p.groups, cp8 = cpg >> 3;
  const int ch0 = g * cpg;
  const T* src;
  int ld, cc;
  if (ch0 < p.c0) { src = reinterpret_cast<const T*>(p.x0); ld = p.ldx0; cc = ch0; }     // a group lies inside ONE source (host checks)
  else { src = reinterpret_cast<const T*>(p.x1); ld = p.ldx1; cc = ch0 - p.c0; }
  const int items = p.hw * cp8;
  float v[GN1_ITEMS][8];
  float sm = 0.f, sq = 0.f;
#pragma unroll
  for (int i = 0; i < GN1_ITEMS; ++i) {
    const int it = tid + i * 256;
    if (it < items) {
      const int px = it / cp8, c8 = it - px * cp8;
      const T* ptr = src + ((long long)b * p.hw + px) * ld + cc + c8 * 8;
      if constexpr (sizeof(T) == 2) {
        Elem<T>::load_chunk(ptr, v[i]);
      } else {
        Elem<T>::load_chunk(ptr, v[i]);
        Elem<T>::load_chunk(ptr + 4, v[i] + 4);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { sm += v[i][j]; sq += v[i][j] * v[i][j]; }
    }
  }
  double dsm = (double)sm, dsq = (double)sq;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dsm += __shfl_xor(dsm, o, 64);
    dsq += __shfl_xor(dsq, o, 64);
  }
  if ((tid & 63) == 0) { red[0][tid >> 6] = dsm; red[1][tid >> 6] = dsq; }
  __syncthreads();
  const double n = (double)cpg * (double)p.hw;
  const double mean_d = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / n;
  double var = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / n - mean_d * mean_d;
  if (var < 0.0) var = 0.0;
  const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)p.eps));
  const bool silu = p.act == SASPA_ACT_SILU;
#pragma unroll
  for (int i = 0; i < GN1_ITEMS; ++i) {
    const int it = tid + i * 256;
    if (it < items) {
      const int px = it / cp8, c8 = it - px * cp8;
      const int ch = ch0 + c8 * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float sc = p.gamma[ch + j] * rstd;
        float y = v[i][j] * sc + (p.beta[ch + j] - mean * sc);          // the arithmetic of gn_apply_kernel
        if (silu) y = sizeof(T) == 2 ? silu_fast(y) : silu_f(y);
        v[i][j] = y;
      }
      T* dst = reinterpret_cast<T*>(p.y) + ((long long)b * p.hw + px) * p.ldy + ch;
      if constexpr (sizeof(T) == 2) {
        Elem<T>::store_chunk(dst, v[i]);
      } else {
        Elem<T>::store_chunk(dst, v[i]);
        Elem<T>::store_chunk(dst + 4, v[i] + 4);
      }
    }
  }
}

// ---- split-K reduce + epilogue + GroupNorm(+SiLU) in one launch (ABI 18) ----------------------------------------------
// One workgroup per (image, group): item = one float4 (4 channels) of one pixel, hw * cpg / 4 items, at most SKGN_ITEMS per thread
// kept in registers between the statistics and the apply.
constexpr int SKGN_ITEMS = 12;
template <typename T>
__global__ __launch_bounds__(256) void splitk_gn_kernel(const SaspaGemmParams gp, const SaspaGroupNormParams p) {
  __shared__ double red[2][4];
  const int tid = threadIdx.x;
  const int b = blockIdx.y, g = blockIdx.x;
  const int N = gp.N;
  const int cpg = N / p.groups, c4 = cpg >> 2;
  const int ch0 = g * cpg;
  const int items = p.hw * c4;
  const long long slab = (long long)gp.M * N;
  const int ks = gp.ksplit;
  float v[SKGN_ITEMS][4];
  float sm = 0.f, sq = 0.f;
  // slab-outer, item-inner: the SKGN_ITEMS loads of one slab are independent and in flight together (an item-outer loop would
  // serialise ks dependent load latencies per item)
  int off[SKGN_ITEMS];
#pragma unroll
  for (int i = 0; i < SKGN_ITEMS; ++i) {
    const int it = tid + i * 256;
    const int px = it / c4, q = it - px * c4;
    off[i] = it < items ? (b * p.hw + px) * N + ch0 + q * 4 : -1;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[i][j] = 0.f;
  }
  for (int s_ = 0; s_ < ks; ++s_) {
    const float* src = gp.workspace + s_ * slab;
    float4 c[SKGN_ITEMS];
#pragma unroll
    for (int i = 0; i < SKGN_ITEMS; ++i)
      if (off[i] >= 0) c[i] = *reinterpret_cast<const float4*>(src + off[i]);
#pragma unroll
    for (int i = 0; i < SKGN_ITEMS; ++i)
      if (off[i] >= 0) { v[i][0] += c[i].x; v[i][1] += c[i].y; v[i][2] += c[i].z; v[i][3] += c[i].w; }
  }
#pragma unroll
  for (int i = 0; i < SKGN_ITEMS; ++i) {
    if (off[i] >= 0) {
      const int it = tid + i * 256;
      const int px = it / c4, q = it - px * c4;
      const int n = ch0 + q * 4;
      float t[4] = {v[i][0], v[i][1], v[i][2], v[i][3]};
      if (gp.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(gp.bias + n);
        t[0] += b4.x; t[1] += b4.y; t[2] += b4.z; t[3] += b4.w;
      }
      if (gp.rowvec) {
        const float4 r4 = *reinterpret_cast<const float4*>(gp.rowvec + (long long)b * gp.ldrv + n);
        t[0] += r4.x; t[1] += r4.y; t[2] += r4.z; t[3] += r4.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float x = t[j] * gp.alpha;
        if constexpr (sizeof(T) == 2) {            // the value the unfused path stores (and its GroupNorm reads back): bf16
          const unsigned w2 = pack2(x, 0.f);
          x = __builtin_bit_cast(float, w2 << 16);
        }
        v[i][j] = x;
        sm += x;
        sq += x * x;
      }
    }
  }
  double dsm = (double)sm, dsq = (double)sq;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    dsm += __shfl_xor(dsm, o, 64);
    dsq += __shfl_xor(dsq, o, 64);
  }
  if ((tid & 63) == 0) { red[0][tid >> 6] = dsm; red[1][tid >> 6] = dsq; }
  __syncthreads();
  const double cnt = (double)cpg * (double)p.hw;
  const double mean_d = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / cnt;
  double var = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / cnt - mean_d * mean_d;
  if (var < 0.0) var = 0.0;
  const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)p.eps));
  const bool silu = p.act == SASPA_ACT_SILU;
#pragma unroll
  for (int i = 0; i < SKGN_ITEMS; ++i) {
    const int it = tid + i * 256;
    if (it < items) {
      const int px = it / c4, q = it - px * c4;
      const int n = ch0 + q * 4;
      const float4 g4 = *reinterpret_cast<const float4*>(p.gamma + n), b4 = *reinterpret_cast<const float4*>(p.beta + n);
      const float gm[4] = {g4.x, g4.y, g4.z, g4.w}, bt[4] = {b4.x, b4.y, b4.z, b4.w};
      float y[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float sc = gm[j] * rstd;
        float o = v[i][j] * sc + (bt[j] - mean * sc);                   // the arithmetic of gn_apply_kernel
        if (silu) o = sizeof(T) == 2 ? silu_fast(o) : silu_f(o);
        y[j] = o;
      }
      Elem<T>::store4(reinterpret_cast<T*>(p.y) + ((long long)b * p.hw + px) * p.ldy + n, y);
    }
  }
}

}  // namespace

// One-launch GroupNorm (ABI 17): eligible when a group lies inside one source, has whole 8-channel chunks and its hw * cpg values
// fit one workgroup's registers (hw * cpg <= 256 * GN1_ITEMS * 8 = 8 192: 8x8 ... 8x12 pixel levels at 40 - 80 channels per group).
extern "C" int saspa_groupnorm_onepass_eligible(const SaspaGroupNormParams* pp) {
  if (!pp) return 0;
  const SaspaGroupNormParams& p = *pp;
  const int C = p.c0 + p.c1;
  if (p.groups <= 0 || C % p.groups) return 0;
  const int cpg = C / p.groups;
  if (cpg % 8 || (p.c1 > 0 && p.c0 % cpg) || (long long)p.hw * cpg > 256LL * GN1_ITEMS * 8) return 0;
  return 1;
}

extern "C" int saspa_groupnorm_onepass(const SaspaGroupNormParams* pp, void* stream) {
  if (!pp) return SASPA_EINVAL;
  const SaspaGroupNormParams& p = *pp;
  if (!p.x0 || !p.gamma || !p.beta || !p.y || p.batch <= 0 || p.hw <= 0 || p.groups <= 0) return SASPA_EINVAL;
  if (p.c1 > 0 && !p.x1) return SASPA_EINVAL;
  if (p.dtype != SASPA_BF16 && p.dtype != SASPA_F32) return SASPA_EINVAL;
  if (p.c0 % 8 || p.c1 % 8 || p.ldx0 % 8 || (p.c1 > 0 && p.ldx1 % 8) || p.ldy % 8) return SASPA_EALIGN;
  if (!aligned16(p.x0) || (p.x1 && !aligned16(p.x1)) || !aligned16(p.y)) return SASPA_EALIGN;
  if (p.batch > 65535 || !saspa_groupnorm_onepass_eligible(pp)) return SASPA_ERANGE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(p.groups, p.batch);
  if (p.dtype == SASPA_BF16) hipLaunchKernelGGL(gn_onepass_kernel<bf16_t>, grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL(gn_onepass_kernel<float>, grid, dim3(256), 0, s, p);
  SASPA_CHECK_LAUNCH();
  return 0;
}

extern "C" int saspa_splitk_groupnorm_eligible(const SaspaGemmParams* gp, const SaspaGroupNormParams* np) {
  if (!gp || !np) return 0;
  const SaspaGemmParams& g = *gp;
  const SaspaGroupNormParams& p = *np;
  if (p.groups <= 0 || g.N <= 0 || g.N % p.groups) return 0;
  const int cpg = g.N / p.groups;
  if (cpg % 4 || (long long)p.hw * cpg > 256LL * SKGN_ITEMS * 4 || (long long)g.M * g.N >= (1LL << 31)) return 0;
  if (g.residual || g.act != SASPA_ACT_NONE || (long long)g.nb1 * g.nb2 > 1 || g.gn_stats) return 0;
  if ((long long)p.batch * p.hw != g.M || p.hw != g.hout * g.wout || p.c0 != g.N || p.c1 != 0) return 0;
  if (g.dtype != SASPA_BF16 && g.dtype != SASPA_F32) return 0;
  return 1;
}

extern "C" int saspa_splitk_groupnorm(const SaspaGemmParams* gp, const SaspaGroupNormParams* np, void* stream) {
  if (!gp || !np) return SASPA_EINVAL;
  const SaspaGemmParams& g = *gp;
  const SaspaGroupNormParams& p = *np;
  if (!g.workspace || g.ksplit < 2 || !p.gamma || !p.beta || !p.y) return SASPA_EINVAL;
  if (!saspa_splitk_groupnorm_eligible(gp, np)) return SASPA_ERANGE;
  if (p.ldy % 4 || (g.rowvec && g.ldrv % 4) || !aligned16(g.workspace) || !aligned16(p.y) || (g.bias && !aligned16(g.bias)) ||
      (g.rowvec && !aligned16(g.rowvec)) || !aligned16(p.gamma) || !aligned16(p.beta))
    return SASPA_EALIGN;
  if (g.rowvec && g.ldrv == 0 && p.batch > 1) {
    // one row vector for every image: blockIdx.y * 0 -- fine
  }
  if (p.batch > 65535) return SASPA_ERANGE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(p.groups, p.batch);
  if (g.dtype == SASPA_BF16) hipLaunchKernelGGL(splitk_gn_kernel<bf16_t>, grid, dim3(256), 0, s, g, p);
  else hipLaunchKernelGGL(splitk_gn_kernel<float>, grid, dim3(256), 0, s, g, p);
  SASPA_CHECK_LAUNCH();
  return 0;
}

int saspa_gn_slabs(int c8) { return gn_geometry(c8).slabs; }

extern "C" int saspa_groupnorm_stats(const SaspaGroupNormParams* pp, void* stream) {
  if (!pp) return SASPA_EINVAL;
  const SaspaGroupNormParams& p = *pp;
  if (p.stats0 || !p.partial) return SASPA_EINVAL;     // with epilogue statistics there is no statistics pass to launch
  if (int e = check_gn(p)) return e;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const GnGeom ge = gn_geometry((p.c0 + p.c1) / 8);
  const int pps = (p.hw + p.nsplit - 1) / p.nsplit;
  dim3 grid(p.nsplit, p.batch, ge.slabs);
  if (p.dtype == SASPA_BF16) hipLaunchKernelGGL(gn_partial_kernel<bf16_t>, grid, dim3(256), 0, s, p, ge.cxw, pps);
  else hipLaunchKernelGGL(gn_partial_kernel<float>, grid, dim3(256), 0, s, p, ge.cxw, pps);
  SASPA_CHECK_LAUNCH();
  return 0;
}

extern "C" int saspa_groupnorm_apply(const SaspaGroupNormParams* pp, void* stream) {
  if (!pp) return SASPA_EINVAL;
  const SaspaGroupNormParams& p = *pp;
  if (int e = check_gn(p)) return e;
  if (!p.y || p.ldy % 8 || !aligned16(p.y)) return SASPA_EALIGN;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const GnGeom ge = gn_geometry((p.c0 + p.c1) / 8);
  const int rows = 256 / ge.cxw;
  // about 1024 workgroups in all, each at least two passes of its pixel rows long (amortises the statistics prologue)
  int nblk = 1024 / (p.batch * ge.slabs);
  const int most = (p.hw + 2 * rows - 1) / (2 * rows);
  if (nblk > most) nblk = most;
  if (nblk < 1) nblk = 1;
  const int ppb = (p.hw + nblk - 1) / nblk;
  nblk = (p.hw + ppb - 1) / ppb;
  dim3 grid(nblk, p.batch, ge.slabs);
  if (p.dtype == SASPA_BF16) hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, grid, dim3(256), 0, s, p, ge.cxw, ge.slabs, ppb);
  else hipLaunchKernelGGL(gn_apply_kernel<float>, grid, dim3(256), 0, s, p, ge.cxw, ge.slabs, ppb);
  SASPA_CHECK_LAUNCH();
  return 0;
}

extern "C" int saspa_layernorm(int dtype, const void* x, int ldx, void* y, int ldy, long long rows, int C,
                               const float* gamma, const float* beta, float eps, void* stream) {
  if (!x || !y || !gamma || !beta || rows <= 0 || C <= 0) return SASPA_EINVAL;
  if (C % 8 || ldx % 8 || ldy % 8 || !aligned16(x) || !aligned16(y)) return SASPA_EALIGN;
  if (C > 2048) return SASPA_ERANGE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (dtype == SASPA_BF16)
    launch_layernorm<bf16_t>(s, (const bf16_t*)x, ldx, (bf16_t*)y, ldy, rows, C, gamma, beta, eps);
  else if (dtype == SASPA_F32)
    launch_layernorm<float>(s, (const float*)x, ldx, (float*)y, ldy, rows, C, gamma, beta, eps);
  else
    return SASPA_EINVAL;
  SASPA_CHECK_LAUNCH();
  return 0;
}
