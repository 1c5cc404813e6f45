// Implicit-GEMM convolution / linear on gfx950 MFMA (bf16: v_mfma_f32_16x16x32_bf16,
// fp32 parity mode: v_mfma_f32_16x16x4_f32).  See include/saspa_hip.h (SaspaGemmParams).
//
// Structure (one workgroup = 4 waves in a 2x2 grid, each wave WM x WN tiles of 16x16):
//   * A operand = im2col view of up to two channel-concatenated NHWC tensors (optionally
//     nearest-x2 upsampled), B operand = weights [N][K]; both K-contiguous, so one
//     16-byte chunk per lane is exactly one MFMA fragment (8 bf16 / 4 fp32 along K).
//   * K-tile = 128 bytes per row for both dtypes (64 bf16 / 32 fp32): the LDS image, its
//     XOR swizzle (chunk ^ (row & 7): conflict-free for ds_read_b128 row reads) and the
//     staging code are shared between the two precisions.
//   * operands are fetched with raw buffer loads (hardware bounds check -> zeros) whose
//     per-lane offsets are fixed per output tile; the K advance lives in scalar registers.
//   * register-staged pipeline, two staging register sets: while tile t is multiplied out of
//     LDS (2 stages), the global loads of tiles t+1 and t+2 are in flight; VGPR -> LDS after
//     the MFMA block, one barrier per K-tile.
//   * the MFMA computes D^T (weights as the A operand) so every lane ends up with 4
//     consecutive output channels of one pixel: 8/16-byte epilogue accesses for bias,
//     time-embedding row vector, residual and the store.
//   * tile shapes: 128x160 when N is a multiple of 160 (the SD-1.5 widths 320/640/1280 and
//     their multiples tile exactly), else 128x128; 128x32 for skinny N; 64x64 when there
//     are too few tiles to fill 256 CUs.
//   * tiles are enumerated so that the 8 XCDs each own a contiguous range (blocks b and
//     b+8 share an XCD): neighbouring tiles (same activation rows) hit the same L2.
//   * split-K (gridDim.y slices of the K range, fp32 partial slabs in a caller workspace +
//     one reduce/epilogue launch) for the deep levels where M is 1-4 K rows but K is 6-23 K.
#include <cstdlib>

#include "common.h"
#include "gemm_internal.h"

#ifdef SASPA_NO_KORDER
constexpr bool KORDER_ON = false;   // A/B build: the K walk exactly as before ABI v4
#else
constexpr bool KORDER_ON = true;
#endif

SaspaDryRun* saspa_dry_state() {      // dry dispatch state of the calling thread (gemm_internal.h)
  static thread_local SaspaDryRun st = {false, 0, 0};
  return &st;
}

namespace {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  __device__ static __forceinline__ void run(const u32x4& wf, const u32x4& xf, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xf),
                                                  acc, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // one 16-byte chunk = 4 consecutive k per lane; lane group g = lane>>4 owns chunk
  // (4*kk + g), so MFMA step j pairs element j of every group: the k order is a fixed
  // permutation shared by both operands (exact fp32 fma chain per output).
  __device__ static __forceinline__ void run(const u32x4& wf, const u32x4& xf, f32x4& acc) {
    // NOTE: bit-cast the WHOLE vector; __builtin_bit_cast(float, vec.x) on an ext-vector
    // element lvalue silently reads element 0 (hipcc / ROCm 7.2).
    const f32x4 a = __builtin_bit_cast(f32x4, wf);
    const f32x4 b = __builtin_bit_cast(f32x4, xf);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b[3], acc, 0, 0, 0);
  }
};

template <> struct Mma<f32x3_t> {
  // per-chunk form (generic loader kernel, fp32 parity of odd shapes): the exact fp32 chain
  __device__ static __forceinline__ void run(const u32x4& wf, const u32x4& xf, f32x4& acc) { Mma<float>::run(wf, xf, acc); }
  // 8 fp32 values of one lane (its two 16-byte chunks of the K-tile) -> 8 bf16 "hi" + 8 bf16 "lo" (round to nearest)
  __device__ static __forceinline__ void split(const u32x4& f0, const u32x4& f1, u32x4& hi, u32x4& lo) {
    const f32x4 a = __builtin_bit_cast(f32x4, f0), b = __builtin_bit_cast(f32x4, f1);
    const float x[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    uint32_t h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t hp = pack2(x[2 * q], x[2 * q + 1]);
      h[q] = hp;
      l[q] = pack2(x[2 * q] - __builtin_bit_cast(float, hp << 16), x[2 * q + 1] - __builtin_bit_cast(float, hp & 0xffff0000u));
    }
    hi = u32x4{h[0], h[1], h[2], h[3]};
    lo = u32x4{l[0], l[1], l[2], l[3]};
  }
  __device__ static __forceinline__ void run3(const u32x4& wh, const u32x4& wl, const u32x4& xh, const u32x4& xl, f32x4& acc) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wl), __builtin_bit_cast(bf16x8, xh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh), __builtin_bit_cast(bf16x8, xl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wh), __builtin_bit_cast(bf16x8, xh), acc, 0, 0, 0);
  }
};
template <typename T> struct is_x3 { static constexpr bool value = false; };
template <> struct is_x3<f32x3_t> { static constexpr bool value = true; };

// ---- shared epilogue of both kernel variants ------------------------------------------
// split-K: raw fp32 slab.  bf16: the tile goes through LDS so that global stores (and the
// residual read) are whole 16-byte chunks of contiguous output rows; GEGLU pairs the value /
// gate halves there.  fp32 parity mode / odd N: direct per-lane path.
// The caller guarantees every wave has finished reading the K-loop's LDS stages.
template <typename T, int WM, int WN, int NWM = 2, int NWN = 2>
__device__ __forceinline__ void gemm_epilogue(const SaspaGemmParams& p, f32x4 (&acc)[WM][WN], u32x4* lds, const int cbm,
                                              const int cbn, const long long ooff) {
  constexpr int BM = 16 * WM * NWM, BN = 16 * WN * NWN, NT = 64 * NWM * NWN;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / NWN, wn = wave % NWN;
  const int frow = lane & 15, fg = lane >> 4;
  const int hw = p.hout * p.wout;
  T* out = reinterpret_cast<T*>(p.out) + ooff;
  const T* res = p.residual ? reinterpret_cast<const T*>(p.residual) + ooff : nullptr;
  const bool geglu = p.act == SASPA_ACT_GEGLU;
  const bool staged = sizeof(T) == 2 && (p.N % 8) == 0 && (p.ldo % 8) == 0 && (!res || (p.ldr % 8) == 0) &&
                      (!geglu || (p.N % BN) == 0);
    if (gridDim.y > 1) {
      // ---- split-K: raw fp32 partial slab, epilogue happens in the reduce launch ----
      float* ws = p.workspace + (long long)blockIdx.y * p.M * p.N;
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int m = cbm * BM + wm * (16 * WM) + i * 16 + frow;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const int n = cbn * BN + wn * (16 * WN) + j * 16 + fg * 4;
          if (n >= p.N) continue;   // N % 4 == 0 is required with split-K
          *reinterpret_cast<float4*>(ws + (long long)m * p.N + n) =
              make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
      }
    } else if (staged) {
    constexpr int CP = BN + 8;                       // LDS row pitch in elements (16-byte pad)
    T* ct = reinterpret_cast<T*>(lds);
    // bias + time-embedding row of the wave's FIRST image, loaded once per column fragment (WN loads in
    // flight together instead of one dependent L2 round trip per accumulator fragment); rows of a later
    // image (tiles that straddle images) take the reload path
    const int m0 = cbm * BM + wm * (16 * WM);
    const int img0 = min(m0, p.M - 1) / hw;
    const int mnext = (img0 + 1) * hw;
    float4 add[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = cbn * BN + wn * (16 * WN) + j * 16 + fg * 4;
      add[j] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < p.N) {   // N % 8 == 0: a 4-vector never straddles N
        if (p.bias) add[j] = *reinterpret_cast<const float4*>(p.bias + n);
        if (p.rowvec) {
          const float4 r4 = *reinterpret_cast<const float4*>(p.rowvec + (long long)img0 * p.ldrv + n);
          add[j].x += r4.x; add[j].y += r4.y; add[j].z += r4.z; add[j].w += r4.w;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      const int mrow = wm * (16 * WM) + i * 16 + frow;
      const int m = cbm * BM + mrow;
      const bool other = p.rowvec && m >= mnext && m < p.M;
      const float* rvd = other ? p.rowvec + (long long)(m / hw) * p.ldrv : nullptr;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int ncol = wn * (16 * WN) + j * 16 + fg * 4;
        const int n = cbn * BN + ncol;
        float v[4] = {acc[i][j][0] + add[j].x, acc[i][j][1] + add[j].y, acc[i][j][2] + add[j].z, acc[i][j][3] + add[j].w};
        if (other && n < p.N) {
          const float4 a4 = *reinterpret_cast<const float4*>(rvd + n);
          const float4 c4 = p.bias ? *reinterpret_cast<const float4*>(p.bias + n) : make_float4(0.f, 0.f, 0.f, 0.f);
          v[0] = acc[i][j][0] + (c4.x + a4.x); v[1] = acc[i][j][1] + (c4.y + a4.y);
          v[2] = acc[i][j][2] + (c4.z + a4.z); v[3] = acc[i][j][3] + (c4.w + a4.w);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= p.alpha;   // SiLU (if any) is applied in the read phase
        Elem<T>::store4(ct + mrow * CP + ncol, v);
      }
    }
    __syncthreads();
    if (!geglu) {
      constexpr int CPR = BN / 8;                    // 16-byte chunks per tile row
      for (int q = tid; q < BM * CPR; q += NT) {
        const int row = q / CPR, ch = q - row * CPR;
        const int m = cbm * BM + row, n = cbn * BN + ch * 8;
        if (m >= p.M || n >= p.N) continue;
        u32x4 c4 = *reinterpret_cast<const u32x4*>(ct + row * CP + ch * 8);
        if (res || p.act != SASPA_ACT_NONE) {
          float a[8];
          unpack8(__builtin_bit_cast(uint4, c4), a);
          if (p.act == SASPA_ACT_SILU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = a[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-a[e]));
          } else if (p.act == SASPA_ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = fmaxf(a[e], 0.0f);
          }
          if (res) {
            float b[8];
            Elem<bf16_t>::load_chunk(reinterpret_cast<const bf16_t*>(res) + (long long)m * p.ldr + n, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += b[e];
          }
          if (p.act == SASPA_ACT_ADD_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = fmaxf(a[e], 0.0f);
          }
          c4 = __builtin_bit_cast(u32x4, pack8(a));
          if (p.gn_stats) *reinterpret_cast<u32x4*>(ct + row * CP + ch * 8) = c4;   // the statistics read the STORED values
        }
        *reinterpret_cast<u32x4*>(out + (long long)m * p.ldo + n) = c4;
      }
      if constexpr (sizeof(T) == 2) if (p.gn_stats) {
        // GroupNorm statistics of this tile's row block (SaspaGemmParams.gn_stats): BM = 128 rows = one block
        static_assert(BM == 128 || BM == 256 || BM <= 64, "row blocks of the statistics are 128 rows");
        __syncthreads();
        const int nunits = BN / p.gn_unit;
        float* scratch = reinterpret_cast<float*>(ct + BM * CP);
#pragma unroll
        for (int hb = 0; hb < (BM + 127) / 128; ++hb) {
          const int r0 = hb * 128;
          const int nrows = min(128, min(BM, p.M - cbm * BM) - r0);
          if (nrows > 0)
            gn_tile_stats<NT>(reinterpret_cast<const bf16_t*>(ct) + r0 * CP, CP, nrows, nunits, p.gn_unit, scratch,
                              p.gn_stats + (((long long)(cbm * BM + r0) / 128) * (p.N / p.gn_unit) + (cbn * BN) / p.gn_unit) * 2);
          if (hb + 1 < (BM + 127) / 128) __syncthreads();
        }
      }
    } else {
      // tile columns [0, BN/2) are values, [BN/2, BN) the matching gates (weights packed so)
      constexpr int HB = BN / 2, CPR = HB / 8;
      for (int q = tid; q < BM * CPR; q += NT) {
        const int row = q / CPR, ch = q - row * CPR;
        const int m = cbm * BM + row, f = cbn * HB + ch * 8;
        if (m >= p.M) continue;
        float a[8], g[8];
        unpack8(*reinterpret_cast<const uint4*>(ct + row * CP + ch * 8), a);
        unpack8(*reinterpret_cast<const uint4*>(ct + row * CP + HB + ch * 8), g);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = fast_gelu_mul(a[e], g[e]);
        *reinterpret_cast<uint4*>(out + (long long)m * p.ldo + f) = pack8(a);
      }
    }
    } else {
  // generic path (fp32 parity mode, odd N): lane holds out[m][n..n+3], m = tile row (lane&15), n = 4*(lane>>4)
#pragma unroll
  for (int i = 0; i < WM; ++i) {
    const int m = cbm * BM + wm * (16 * WM) + i * 16 + frow;
    if (m >= p.M) continue;
    const float* rv = nullptr;
    if (p.rowvec) rv = p.rowvec + (long long)(m / hw) * p.ldrv;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const int n = cbn * BN + wn * (16 * WN) + j * 16 + fg * 4;
      if (n >= p.N) continue;
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (n + 3 < p.N) {
        if (p.bias) {
          const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
          v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
        }
        if (rv) {
          const float4 r4 = *reinterpret_cast<const float4*>(rv + n);
          v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = act_pre(p.act, v[r] * p.alpha);
        if (res) {
          float rr[4];
          Elem<T>::load4(res + (long long)m * p.ldr + n, rr);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] += rr[r];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = act_post(p.act, v[r]);
        Elem<T>::store4(out + (long long)m * p.ldo + n, v);
      } else {
        for (int r = 0; r < 4 && n + r < p.N; ++r) {
          float x = v[r];
          if (p.bias) x += p.bias[n + r];
          if (rv) x += rv[n + r];
          x = act_pre(p.act, x * p.alpha);
          if (res) x += Elem<T>::load1(res + (long long)m * p.ldr + n + r);
          Elem<T>::store1(out + (long long)m * p.ldo + n + r, act_post(p.act, x));
        }
      }
    }
  }
    }
}

// PW: pointwise (1x1, stride 1, no pad, no upsample): the A row of output pixel m is input
// pixel m -- no window arithmetic at all.
// Persistent over tiles: workgroup b walks tiles s(b), s(b)+G, s(b)+2G ... (s = XCD-aware
// remap of the block id, G = gridDim.x).  The global loads of the NEXT tile's first two
// K-tiles are issued before the current tile's epilogue, so workgroups stream continuously
// instead of loading / storing in lock-step bursts (short-K layers are memory-bound).
template <typename T, int WM, int WN, bool PW>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const SaspaGemmParams p, const int ntiles, const int npart8) {
  constexpr int BM = 32 * WM, BN = 32 * WN;
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BK = 8 * EPC;
  constexpr int A_CH = BM * 8 / 256, B_CH = BN * 8 / 256;
  static_assert(A_CH >= 1 && B_CH >= 1, "tile too small for 256 threads");
  constexpr int STAGE = (BM + BN) * 8;  // u32x4 per stage
  __shared__ u32x4 lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  // ---- XCD-aware order (bijective for any grid size): blocks b and b+8 share an XCD and
  //      get neighbouring tiles (same activation rows -> same L2) ----
  const int nbn = (p.N + BN - 1) / BN;
  const int G = gridDim.x;
  int tile;
  {
    const int L = blockIdx.x;
    const int qd = G >> 3, rr = G & 7, xcd = L & 7, idx = L >> 3;
    tile = (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + idx;
  }
  const int z = blockIdx.z;
  const int i1 = z / p.nb2, i2 = z - i1 * p.nb2;

  const T* a0 = reinterpret_cast<const T*>(p.a0) + (i1 * p.sa1 + i2 * p.sa2);
  const T* a1 = reinterpret_cast<const T*>(p.a1);
  const T* w = reinterpret_cast<const T*>(p.w) + (i1 * p.sw1 + i2 * p.sw2);
  const long long ooff = i1 * p.so1 + i2 * p.so2;

  // ---- K range of this block (split-K over gridDim.y) ----
  const int nk_all = (p.K + BK - 1) / BK;
  const int kt_per = (nk_all + gridDim.y - 1) / gridDim.y;
  const int kt0 = blockIdx.y * kt_per;
  const int nk = max(0, min(nk_all, kt0 + kt_per) - kt0);

  // ---- loader state (re-initialised per tile by setup_tile) ----
  // FAST mode (every SD-1.5 layer): a K-tile lies inside ONE tap of ONE source, so the tap,
  // the source and the channel offset are wave-uniform scalars; each lane keeps, per A row,
  // the pixel index of the window centre and a bitmask of the taps that fall inside the
  // image, and per B row a constant byte offset.  A K-tile of loads then costs ~3 VALU
  // instructions per A row and none per B row.  GENERIC mode (channel counts below one
  // K-tile, nearest-x2 upsample, ragged concat) derives tap / source per lane.
  constexpr int SZ = (int)sizeof(T);
  const int kc = tid & 7;
  const int r0 = tid >> 3;  // rows r0 + 32*i
  const int hw = p.hout * p.wout;
  const int ctot = p.c0 + p.c1;
  const int hv = p.upsample ? 2 * p.hin : p.hin;
  const int wv = p.upsample ? 2 * p.win : p.win;
  const bool fast = (ctot % BK) == 0 && (p.c1 == 0 || (p.c0 % BK) == 0) && !p.upsample;
  const rsrc_t rs0 = make_rsrc(a0);
  const rsrc_t rs1 = make_rsrc(p.c1 > 0 ? (const void*)a1 : (const void*)a0);
  const rsrc_t rsw = make_rsrc(w);
  int bm = 0, bn = 0;
  int row_a[A_CH], row_b[A_CH], row_c[A_CH];  // FAST: (centre pixel, tap mask, -); GENERIC: (iy0, ix0, pix0) / PW (m,-,-)
  unsigned offb[B_CH];                        // byte offset of (weight row, chunk kc) or kInvalid
  int k = 0, c = 0, dy = 0, dx = 0;           // GENERIC: per-lane k-state
  int ku = 0, cu = 0, dyu = 0, dxu = 0;       // FAST: wave-uniform k-state (tile start)

  auto setup_tile = [&](int t) __attribute__((always_inline)) {
    if (npart8 > 0) {
      // N-partitioned order (weight-heavy problems, host decides): the XCD of this block (L & 7 -> the x-th eighth of every
      // round of G tiles) owns the x-th eighth of the N tiles for ALL row blocks, so its slice of the weights stays in
      // its L2 while the activations stream through once per XCD
      const int G8 = G >> 3;
      const int k = t / G, r = t - k * G;
      const int x = r / G8;
      const int q = k * G8 + (r - x * G8);
      bm = q / npart8;
      bn = x * npart8 + (q - bm * npart8);
    } else {
      bm = t / nbn;
      bn = t - bm * nbn;
    }
    if (PW) {
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        const int m = bm * BM + r0 + 32 * i;
        row_a[i] = (m < p.M) ? m : -1;
        row_b[i] = (m < p.M) ? 1 : 0;
        row_c[i] = 0;
      }
    } else {
      // first row by division, the following rows (+32 each) by carry propagation
      int m = bm * BM + r0;
      int b = m / hw;
      int rem = m - b * hw;
      int oy = rem / p.wout;
      int ox = rem - oy * p.wout;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        if (fast) {
          // centre pixel of the window and the validity bit of every tap
          row_a[i] = b * p.hin * p.win + (oy * p.stride) * p.win + ox * p.stride;
          int mask = 0;
          if (m < p.M) {
            for (int ty = 0; ty < p.kh; ++ty)
              for (int tx = 0; tx < p.kw; ++tx)
                if ((unsigned)(iy0 + ty) < (unsigned)hv && (unsigned)(ix0 + tx) < (unsigned)wv) mask |= 1 << (ty * p.kw + tx);
          }
          row_b[i] = mask;
          row_c[i] = 0;
        } else if (m < p.M) {
          row_a[i] = iy0;
          row_b[i] = ix0;
          row_c[i] = b * p.hin * p.win;
        } else {
          row_a[i] = row_b[i] = -(1 << 28);
          row_c[i] = 0;
        }
        m += 32;
        ox += 32;
        while (ox >= p.wout) { ox -= p.wout; ++oy; }
        while (oy >= p.hout) { oy -= p.hout; ++b; }
      }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int n = bn * BN + r0 + 32 * i;
      offb[i] = (n < p.N) ? (unsigned)(n * p.ldw * SZ + kc * 16) : kInvalid;
    }
    k = kt0 * BK + kc * EPC;
    const int tap = k / ctot;
    c = k - tap * ctot;
    dy = tap / p.kw;
    dx = tap - dy * p.kw;
    ku = kt0 * BK;
    const int tapu = ku / ctot;
    cu = ku - tapu * ctot;
    dyu = tapu / p.kw;
    dxu = tapu - dyu * p.kw;
  };

  // two register staging sets: while tile t is being multiplied out of LDS, tiles t+1 and
  // t+2 are in flight from global memory (two K-tiles of loads outstanding per workgroup)
  u32x4 ra0[A_CH], rb0[B_CH], ra1[A_CH], rb1[B_CH];

  auto load_tile = [&](u32x4 (&ra)[A_CH], u32x4 (&rb)[B_CH]) __attribute__((always_inline)) {
    if (fast) {
      // ---- wave-uniform tap / source / channel offset ----
      const bool s0 = cu < p.c0;
      const rsrc_t rs = s0 ? rs0 : rs1;
      const int ldsz = (s0 ? p.lda0 : p.lda1) * SZ;
      const unsigned soff = (unsigned)((s0 ? cu : cu - p.c0) * SZ);
      const int tapbit = dyu * p.kw + dxu;
      const int pixoff = PW ? 0 : (dyu - p.pad) * p.win + (dxu - p.pad);
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        const unsigned off = (unsigned)((row_a[i] + pixoff) * ldsz + kc * 16);
        const bool ok = PW ? (row_b[i] != 0) : (((row_b[i] >> tapbit) & 1) != 0);
        ra[i] = buf_load(rs, ok ? off : kInvalid, soff);
      }
      const unsigned soffw = (unsigned)(ku * SZ);
#pragma unroll
      for (int i = 0; i < B_CH; ++i) rb[i] = buf_load(rsw, offb[i], soffw);
      ku += BK;
      cu += BK;
      if (cu >= ctot) {
        cu -= ctot;
        if (++dxu == p.kw) { dxu = 0; ++dyu; }
      }
    } else {
      // ---- per-lane tap / source (small channel counts, upsample, ragged concat) ----
      const bool kvalid = k < p.K;
      const bool s0 = c < p.c0;
      const int ld = s0 ? p.lda0 : p.lda1;
      const int cc = s0 ? c : c - p.c0;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        unsigned off;
        bool inb;
        if (PW) {
          inb = kvalid && row_a[i] >= 0;
          off = (unsigned)((row_a[i] * ld + cc) * SZ);
        } else {
          int iy = row_a[i] + dy, ix = row_b[i] + dx;
          inb = kvalid && (unsigned)iy < (unsigned)hv && (unsigned)ix < (unsigned)wv;
          if (p.upsample) { iy >>= 1; ix >>= 1; }
          off = (unsigned)(((row_c[i] + iy * p.win + ix) * ld + cc) * SZ);
        }
        u32x4 v = buf_load(rs0, (inb && s0) ? off : kInvalid, 0u);
        if (p.c1 > 0) v |= buf_load(rs1, (inb && !s0) ? off : kInvalid, 0u);
        ra[i] = v;
      }
#pragma unroll
      for (int i = 0; i < B_CH; ++i) rb[i] = buf_load(rsw, kvalid ? offb[i] + (unsigned)((k - kc * EPC) * SZ) : kInvalid, 0u);
      k += BK;
      c += BK;
      while (c >= ctot) {
        c -= ctot;
        if (++dx == p.kw) { dx = 0; ++dy; }
      }
    }
  };
  auto store_tile = [&](int stage, const u32x4 (&ra)[A_CH], const u32x4 (&rb)[B_CH]) __attribute__((always_inline)) {
    u32x4* la = lds + stage * STAGE;
    u32x4* lb = la + BM * 8;
    const int sw = kc ^ (r0 & 7);
#pragma unroll
    for (int i = 0; i < A_CH; ++i) la[(r0 + 32 * i) * 8 + sw] = ra[i];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) lb[(r0 + 32 * i) * 8 + sw] = rb[i];
  };

  f32x4 acc[WM][WN];
  const int frow = lane & 15, fg = lane >> 4;
  auto compute = [&](int stage) __attribute__((always_inline)) {
    const u32x4* la = lds + stage * STAGE;
    const u32x4* lb = la + BM * 8;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int chunk = kk * 4 + fg;
      u32x4 xa[WM], wb[WN];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int row = wm * (16 * WM) + i * 16 + frow;
        xa[i] = la[row * 8 + (chunk ^ (row & 7))];
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int row = wn * (16 * WN) + j * 16 + frow;
        wb[j] = lb[row * 8 + (chunk ^ (row & 7))];
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) Mma<T>::run(wb[j], xa[i], acc[i][j]);
    }
  };

  setup_tile(tile);
  if (nk > 0) load_tile(ra0, rb0);
  if (nk > 1) load_tile(ra1, rb1);
  for (;;) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (nk > 0) store_tile(0, ra0, rb0);
    __syncthreads();
    for (int kt = 0; kt < nk; kt += 2) {
      // even tile kt: LDS stage 0; set 0 is free -> prefetch tile kt+2; set 1 (tile kt+1) -> stage 1
      if (kt + 2 < nk) load_tile(ra0, rb0);
      compute(0);
      if (kt + 1 < nk) store_tile(1, ra1, rb1);
      __syncthreads();
      if (kt + 1 >= nk) break;
      // odd tile kt+1: LDS stage 1; set 1 is free -> prefetch tile kt+3; set 0 (tile kt+2) -> stage 0
      if (kt + 3 < nk) load_tile(ra1, rb1);
      compute(1);
      if (kt + 2 < nk) store_tile(0, ra0, rb0);
      __syncthreads();
    }

    // ---- next tile: its first two K-tiles go in flight now, under this tile's epilogue ----
    const int cbm = bm, cbn = bn;
    const int next = tile + G;
    const bool has_next = next < ntiles;
    if (has_next) {
      setup_tile(next);
      if (nk > 0) load_tile(ra0, rb0);
      if (nk > 1) load_tile(ra1, rb1);
    }

    gemm_epilogue<T, WM, WN>(p, acc, lds, cbm, cbn, ooff);
    if (!has_next) break;
    tile = next;
    __syncthreads();   // LDS reads of the staged epilogue are done before the next tile's first store
  }
}

// ---- LDS-DMA variant (FAST layers: a K-tile lies inside one tap of one source) -------------
// Operand tiles go global -> LDS directly (buffer_load ... lds): no staging VGPRs and no
// ds_write pass (register staging costs ~9 KB of ds_write_b128 per wave per K-tile at only
// ~79 B/clk/CU, which alone exceeds the MFMA time of the tile).  The LDS image is lane-linear
// per wave instruction (64 lanes x 16 B = 8 rows x 128 B), so the XOR swizzle is applied on
// the SOURCE side: the lane that lands in slot s of row r fetches logical chunk s ^ (r & 7);
// the fragment reads apply the same XOR.  Out-of-range lanes (M / N edge, halo) use an offset
// beyond num_records and land as zeros.  Two LDS stages: the DMA of tile t+1 is in flight
// during the MFMAs of tile t; one barrier per K-tile.

// UP (nearest-x2 input) is a template parameter: as a run-time flag it put one scalar branch in front of every A-piece
// DMA of every 3x3 conv's K-tile (4 per K-tile on the 128-row tile), and this issue path does not forgive branches.
template <typename T, int WM, int WN, int NWM, int NWN, bool PW, int NSTAGE, bool UP = false>
__global__ __launch_bounds__(64 * NWM * NWN, (NWM * NWN == 4 && NSTAGE > 2) ? 1 : 2) void gemm_dma_kernel(const SaspaGemmParams p,
                                                                                                      const int ntiles_abl, const int npart8) {
  // diagnostic ablation (tools/gemm_ablate.py only; 0 in production): bits 28..30 of the tile count
  const int ntiles = ntiles_abl & 0x0fffffff;
#ifdef SASPA_GEMM_ABLATION
  const int abl = (ntiles_abl >> 28) & 15;     // 1: no MFMA  2: no LDS reads / MFMA  4: no DMA  8: MFMA on stale registers (no LDS reads)
#else
  constexpr int abl = 0;                       // production build: the ablation branches fold away
#endif
  constexpr int BM = 16 * WM * NWM, BN = 16 * WN * NWN, NT = 64 * NWM * NWN;
  constexpr int RPI = NT / 8;                         // tile rows covered by one DMA instruction of the whole workgroup
  constexpr int EPC = Elem<T>::EPC;
  constexpr int BK = 8 * EPC;
  constexpr int A_CH = (BM + RPI - 1) / RPI, B_CH = (BN + RPI - 1) / RPI;
  static_assert(BM % RPI == 0 || NSTAGE == 2, "counted vmcnt: A rows must fill whole DMA instructions");
  // B rows may end inside the last DMA instruction: waves whose 8-row slice lies beyond BN skip it,
  // so the number of DMA instructions per K-tile is wave-dependent (G_LO for the low waves)
  constexpr int B_FULL = BN / RPI;                    // B instructions every wave issues
  constexpr bool B_RAGGED = (BN % RPI) != 0;
  constexpr int STAGE = (BM + BN) * 8;  // u32x4 per stage
  constexpr int SZ = (int)sizeof(T);
  __shared__ u32x4 lds[NSTAGE * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / NWN, wn = wave % NWN;

  const int nbn = (p.N + BN - 1) / BN;
  const int G = gridDim.x;
  int tile;
  {
    const int L = blockIdx.x;
    const int qd = G >> 3, rr = G & 7, xcd = L & 7, idx = L >> 3;
    tile = (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + idx;
  }
  const int z = blockIdx.z;
  const int i1 = z / p.nb2, i2 = z - i1 * p.nb2;
  const T* a0 = reinterpret_cast<const T*>(p.a0) + (i1 * p.sa1 + i2 * p.sa2);
  const T* a1 = reinterpret_cast<const T*>(p.a1);
  const T* w = reinterpret_cast<const T*>(p.w) + (i1 * p.sw1 + i2 * p.sw2);
  const long long ooff = i1 * p.so1 + i2 * p.so2;

  const int nk_all = (p.K + BK - 1) / BK;
  const int kt_per = (nk_all + gridDim.y - 1) / gridDim.y;
  const int kt0 = blockIdx.y * kt_per;
  const int nk = max(0, min(nk_all, kt0 + kt_per) - kt0);

  const int r0 = tid >> 3;                        // tile row r0 + RPI*i lands in LDS row r0 + RPI*i
  const int kcs = (tid & 7) ^ (r0 & 7);           // logical 16-byte chunk fetched by this lane
  const int hw = p.hout * p.wout;
  const int ctot = p.c0 + p.c1;
  const int chunk_major = (KORDER_ON && p.korder == SASPA_KORDER_CHUNK) ? 1 : 0;   // wave-uniform
  const rsrc_t rs0 = make_rsrc(a0);
  const rsrc_t rs1 = make_rsrc(p.c1 > 0 ? (const void*)a1 : (const void*)a0);
  const rsrc_t rsw = make_rsrc(w);
  int bm = 0, bn = 0;
  int pix[A_CH], msk[A_CH], upc[A_CH];
  // !UP: byte offset of (pixel of row i, chunk kcs) in each source, fixed per tile -- the per-K-tile address of an A piece
  // is then one add of the (wave-uniform) tap offset instead of a 64-bit multiply-add per piece and K-tile
  unsigned offa0[A_CH], offa1[A_CH];
  const int hv = UP ? 2 * p.hin : p.hin, wv = UP ? 2 * p.win : p.win;
  unsigned offb[B_CH];
  int ku = 0, cu = 0, dyu = 0, dxu = 0;

  auto setup_tile = [&](int t) __attribute__((always_inline)) {
    if (npart8 > 0) {
      // N-partitioned order (weight-heavy problems, host decides): the XCD of this block (L & 7 -> the x-th eighth of every
      // round of G tiles) owns the x-th eighth of the N tiles for ALL row blocks, so its slice of the weights stays in
      // its L2 while the activations stream through once per XCD
      const int G8 = G >> 3;
      const int k = t / G, r = t - k * G;
      const int x = r / G8;
      const int q = k * G8 + (r - x * G8);
      bm = q / npart8;
      bn = x * npart8 + (q - bm * npart8);
    } else {
      bm = t / nbn;
      bn = t - bm * nbn;
    }
    if (PW) {
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        const int m = bm * BM + r0 + RPI * i;
        pix[i] = m;
        msk[i] = (m < p.M) ? 1 : 0;
        upc[i] = 0;
        offa0[i] = (unsigned)(m * (p.lda0 * SZ) + kcs * 16);
        offa1[i] = (unsigned)(m * (p.lda1 * SZ) + kcs * 16);
      }
    } else {
      int m = bm * BM + r0;
      int b = m / hw;
      int rem = m - b * hw;
      int oy = rem / p.wout;
      int ox = rem - oy * p.wout;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
        if (UP) {
          // nearest x2: the window walks the virtual 2H x 2W grid; keep its top-left corner
          // (packed y | x) and resolve the source pixel ((iy0+dy)>>1, (ix0+dx)>>1) per tap
          pix[i] = b * p.hin * p.win;
          upc[i] = ((iy0 + 1) << 16) | (ix0 + 1);     // +1 keeps both halves non-negative (pad <= 1)
        } else {
          pix[i] = b * p.hin * p.win + (oy * p.stride) * p.win + ox * p.stride;
          offa0[i] = (unsigned)(pix[i] * (p.lda0 * SZ) + kcs * 16);
          offa1[i] = (unsigned)(pix[i] * (p.lda1 * SZ) + kcs * 16);
        }
        int mask = 0;
        if (m < p.M) {
          for (int ty = 0; ty < p.kh; ++ty)
            for (int tx = 0; tx < p.kw; ++tx)
              if ((unsigned)(iy0 + ty) < (unsigned)hv && (unsigned)(ix0 + tx) < (unsigned)wv) mask |= 1 << (ty * p.kw + tx);
        }
        msk[i] = mask;
        m += RPI;
        ox += RPI;
        while (ox >= p.wout) { ox -= p.wout; ++oy; }
        while (oy >= p.hout) { oy -= p.hout; ++b; }
      }
    }
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
      const int n = bn * BN + r0 + RPI * i;
      offb[i] = (n < p.N && r0 + RPI * i < BN) ? (unsigned)(n * p.ldw * SZ + kcs * 16) : kInvalid;
    }
    ku = kt0 * BK;
    if (chunk_major) {
      // chunk-major K: K-tile kt = (channel chunk kt / T, tap kt % T)
      const int ntap = p.kh * p.kw;
      const int chunk = kt0 / ntap, tap = kt0 - chunk * ntap;
      cu = chunk * BK;
      dyu = tap / p.kw;
      dxu = tap - dyu * p.kw;
    } else {
      const int tapu = ku / ctot;
      cu = ku - tapu * ctot;
      dyu = tapu / p.kw;
      dxu = tapu - dyu * p.kw;
    }
  };

  auto dma_tile = [&](int stage) __attribute__((always_inline)) {
    if (abl & 4) return;
    const bool s0 = cu < p.c0;
    const rsrc_t rs = s0 ? rs0 : rs1;
    const int ldsz = (s0 ? p.lda0 : p.lda1) * SZ;
    const int soff = (s0 ? cu : cu - p.c0) * SZ;
    const int tapbit = dyu * p.kw + dxu;
    const int pixoff = PW ? 0 : (dyu - p.pad) * p.win + (dxu - p.pad);
    const unsigned tapoff = (unsigned)(pixoff * ldsz);
    u32x4* la = lds + stage * STAGE;
    u32x4* lb = la + BM * 8;
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      unsigned off;
      if (!PW && UP) {
        const int iy = ((upc[i] >> 16) - 1 + dyu) >> 1, ix = ((upc[i] & 0xffff) - 1 + dxu) >> 1;
        off = (unsigned)((pix[i] + iy * p.win + ix) * ldsz + kcs * 16);
      } else {
        off = (s0 ? offa0[i] : offa1[i]) + tapoff;      // == (pix + pixoff) * ldsz + kcs * 16 mod 2^32
      }
      const bool ok = PW ? (msk[i] != 0) : (((msk[i] >> tapbit) & 1) != 0);
      if (BM % RPI == 0 || RPI * i + 8 * wave < BM)   // compile-time true for whole groups: no branch around the DMA
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(la + (RPI * i + 8 * wave) * 8), 16, (int)(ok ? off : kInvalid),
                                                 soff, 0, 0);
    }
    const int soffw = ku * SZ;
#pragma unroll
    for (int i = 0; i < B_CH; ++i)
      if (RPI * (i + 1) <= BN || RPI * i + 8 * wave < BN)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void_t*)(lb + (RPI * i + 8 * wave) * 8), 16, (int)offb[i], soffw, 0, 0);
    ku += BK;
    if (KORDER_ON) {
      // branch-free mixed-radix step (a scalar branch in this path costs the DMA kernels ~25 %, measured): tap-major
      // counts (c, x, y) with the channel offset fastest, chunk-major (x, y, c) with the tap fastest
      const int cu_t = cu + BK;
      const int wc = (cu_t >= ctot) ? 1 : 0;                        // tap-major: channel wrap carries into x
      const int dx1 = dxu + (chunk_major ? 1 : wc);
      const int wx = (dx1 == p.kw) ? 1 : 0;
      const int dy1 = dyu + wx;
      const int wy = (chunk_major && dy1 == p.kh) ? 1 : 0;          // chunk-major: tap wrap carries into the chunk
      cu = chunk_major ? cu + (wy ? BK : 0) : (wc ? cu_t - ctot : cu_t);
      dxu = wx ? 0 : dx1;
      dyu = wy ? 0 : dy1;
    } else {
      cu += BK;
      if (cu >= ctot) {
        cu -= ctot;
        if (++dxu == p.kw) { dxu = 0; ++dyu; }
      }
    }
  };

  f32x4 acc[WM][WN];
  const int frow = lane & 15, fg = lane >> 4;
  auto compute = [&](int stage) __attribute__((always_inline)) {
    if (abl & 2) return;
    const u32x4* la = lds + stage * STAGE;
    const u32x4* lb = la + BM * 8;
    if constexpr (is_x3<T>::value) {
      // SASPA_F32X3: the lane's two chunks (fg, 4 + fg) are its 8 k values of ONE 16x16x32 bf16 MFMA step -- the same
      // k assignment for both operands, so any fixed assignment is a valid permutation of the sum
      u32x4 xh[WM], xl[WM];
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int row = wm * (16 * WM) + i * 16 + frow;
        Mma<T>::split(la[row * 8 + (fg ^ (row & 7))], la[row * 8 + ((4 + fg) ^ (row & 7))], xh[i], xl[i]);
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int row = wn * (16 * WN) + j * 16 + frow;
        u32x4 wh, wl;
        if (p.w_split) {       // ABI 20: the weights arrive as [hi | lo] per K-tile, chunk fg of either half = this lane's 8 k values
          wh = lb[row * 8 + (fg ^ (row & 7))];
          wl = lb[row * 8 + ((4 + fg) ^ (row & 7))];
        } else {
          Mma<T>::split(lb[row * 8 + (fg ^ (row & 7))], lb[row * 8 + ((4 + fg) ^ (row & 7))], wh, wl);
        }
#pragma unroll
        for (int i = 0; i < WM; ++i) Mma<T>::run3(wh, wl, xh[i], xl[i], acc[i][j]);
      }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int chunk = kk * 4 + fg;
      u32x4 xa[WM], wb[WN];
      if (abl & 8) {
#pragma unroll
        for (int i = 0; i < WM; ++i) xa[i] = u32x4{(unsigned)chunk, (unsigned)i, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
        for (int j = 0; j < WN; ++j) wb[j] = u32x4{(unsigned)j, (unsigned)stage, 0x3f803f80u, 0x3f803f80u};
      } else {
#pragma unroll
      for (int i = 0; i < WM; ++i) {
        const int row = wm * (16 * WM) + i * 16 + frow;
        xa[i] = la[row * 8 + (chunk ^ (row & 7))];
      }
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int row = wn * (16 * WN) + j * 16 + frow;
        wb[j] = lb[row * 8 + (chunk ^ (row & 7))];
      }
      }
      if (abl & 1) {
        // keep the fragment reads alive without issuing MFMAs
#pragma unroll
        for (int i = 0; i < WM; ++i) asm volatile("" ::"v"(xa[i]));
#pragma unroll
        for (int j = 0; j < WN; ++j) asm volatile("" ::"v"(wb[j]));
        continue;
      }
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j) Mma<T>::run(wb[j], xa[i], acc[i][j]);
    }
  };

  setup_tile(tile);
  for (;;) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // NSTAGE-deep ring: tiles kt+1 .. kt+NSTAGE-1 are in flight while tile kt is multiplied.
    // Counted vmcnt + raw s_barrier (a __syncthreads() would drain every DMA).
#pragma unroll
    for (int st = 0; st < NSTAGE - 1; ++st)
      if (st < nk) dma_tile(st);
    for (int kt = 0; kt < nk; ++kt) {
      // tile kt has landed once at most (NSTAGE-2) younger groups are outstanding
      if (NSTAGE > 2 && kt + NSTAGE - 2 < nk) {
        if (B_RAGGED && RPI * B_FULL + 8 * wave < BN) {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * (A_CH + B_FULL + 1)) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * (A_CH + B_FULL)) : "memory");
        }
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                    // everyone's share landed; stage of tile kt-1 is free
      asm volatile("" ::: "memory");
      // (issuing the tail's pieces unconditionally with out-of-range offsets instead of this branch: measured 1.2 % slower)
      if (kt + NSTAGE - 1 < nk) dma_tile((kt + NSTAGE - 1) % NSTAGE);
      // the DMA must be ISSUED before the MFMA block so that it flies under it: without this fence
      // hipcc's scheduler sinks all but one buffer_load..lds below the MFMAs (measured: DMA time
      // and compute time then add up instead of overlapping)
      __builtin_amdgcn_sched_barrier(0);
      compute(kt % NSTAGE);
    }
    __syncthreads();   // every wave is done with the K-loop's LDS stages before the epilogue reuses them
    const int cbm = bm, cbn = bn;
    const int next = tile + G;
    const bool has_next = next < ntiles;
    if (has_next) setup_tile(next);
    gemm_epilogue<T, WM, WN, NWM, NWN>(p, acc, lds, cbm, cbn, ooff);
    if (!has_next) break;
    tile = next;
    __syncthreads();   // staged epilogue reads are done before the next tile's first DMA
  }
}

// split-K reduce + epilogue: out = act(alpha*(sum_s ws[s] + bias + rowvec)) + residual
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const SaspaGemmParams p, int ksplit) {
  const int n4 = p.N >> 2;
  const long long total = (long long)p.M * n4;
  const long long slab = (long long)p.M * p.N;
  const int hw = p.hout * p.wout;
  T* out = reinterpret_cast<T*>(p.out);
  const T* res = reinterpret_cast<const T*>(p.residual);
  for (long long it = (long long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long long)gridDim.x * 256) {
    const int m = (int)(it / n4);
    const int n = (int)(it - (long long)m * n4) * 4;
    float4 a = *reinterpret_cast<const float4*>(p.workspace + (long long)m * p.N + n);
    for (int s = 1; s < ksplit; ++s) {
      const float4 b = *reinterpret_cast<const float4*>(p.workspace + s * slab + (long long)m * p.N + n);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    float v[4] = {a.x, a.y, a.z, a.w};
    if (p.bias) {
      const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
      v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
    }
    if (p.rowvec) {
      const float4 r4 = *reinterpret_cast<const float4*>(p.rowvec + (long long)(m / hw) * p.ldrv + n);
      v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = act_pre(p.act, v[r] * p.alpha);
    if (res) {
      float rr[4];
      Elem<T>::load4(res + (long long)m * p.ldr + n, rr);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] += rr[r];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = act_post(p.act, v[r]);
    Elem<T>::store4(out + (long long)m * p.ldo + n, v);
  }
}

// split-K reduce + epilogue + GroupNorm statistics (SaspaGemmParams.gn_stats): one workgroup per (128-row block, 80-column
// slab) -- 512 - 1024 workgroups on the 16x16 / 32x32 levels, two per CU; four rows per trip (4 x ksplit 16-byte loads in
// flight per lane: the pass is bandwidth bound, the first version with one row per trip ran at half the plain reduce's rate);
// the finished bf16 slab is staged in LDS like a GEMM tile and gn_tile_stats reads it.  bf16 only.
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const SaspaGemmParams p, int ksplit) {
  constexpr int BR = 128, BC = 80, CP = BC + 8, C4 = BC / 4, U = 4;
  __shared__ __attribute__((aligned(16))) bf16_t ct[BR * CP + 4 * 256];
  const int tid = threadIdx.x;
  const int rb = blockIdx.x, cs = blockIdx.y;
  const long long slab = (long long)p.M * p.N;
  const int hw = p.hout * p.wout;
  bf16_t* out = reinterpret_cast<bf16_t*>(p.out);
  const bf16_t* res = reinterpret_cast<const bf16_t*>(p.residual);
  const int nrows = min(BR, p.M - rb * BR);
  const int total = nrows * C4;
  for (int q0 = tid; q0 < total; q0 += U * 256) {
    float4 a[U];
    int row[U], c4[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = min(q0 + u * 256, total - 1);           // clamped duplicates are recomputed, never stored twice (below)
      row[u] = q / C4;
      c4[u] = q - row[u] * C4;
      a[u] = *reinterpret_cast<const float4*>(p.workspace + (long long)(rb * BR + row[u]) * p.N + cs * BC + c4[u] * 4);
    }
    for (int s = 1; s < ksplit; ++s) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float4 b = *reinterpret_cast<const float4*>(p.workspace + s * slab + (long long)(rb * BR + row[u]) * p.N + cs * BC + c4[u] * 4);
        a[u].x += b.x; a[u].y += b.y; a[u].z += b.z; a[u].w += b.w;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (q0 + u * 256 >= total) continue;
      const int m = rb * BR + row[u], n = cs * BC + c4[u] * 4;
      float v[4] = {a[u].x, a[u].y, a[u].z, a[u].w};
      if (p.bias) {
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + n);
        v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w;
      }
      if (p.rowvec) {
        const float4 r4 = *reinterpret_cast<const float4*>(p.rowvec + (long long)(m / hw) * p.ldrv + n);
        v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = act_pre(p.act, v[r] * p.alpha);
      if (res) {
        float rr[4];
        Elem<bf16_t>::load4(res + (long long)m * p.ldr + n, rr);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += rr[r];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = act_post(p.act, v[r]);
      Elem<bf16_t>::store4(out + (long long)m * p.ldo + n, v);
      Elem<bf16_t>::store4(ct + row[u] * CP + c4[u] * 4, v);
    }
  }
  __syncthreads();
  gn_tile_stats<256>(ct, CP, nrows, BC / p.gn_unit, p.gn_unit, reinterpret_cast<float*>(ct + BR * CP),
                     p.gn_stats + ((long long)rb * (p.N / p.gn_unit) + (cs * BC) / p.gn_unit) * 2);
}

template <typename T, int WM, int WN, int NWM = 2, int NWN = 2>
int launch(const SaspaGemmParams& p, hipStream_t s, int ksplit) {
  constexpr int BM = 16 * WM * NWM, BN = 16 * WN * NWN, NT = 64 * NWM * NWN;
  SASPA_DRY_RETURN(SASPA_GEMM_TILED, ksplit);
  const int tiles = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
  // persistent grid: as many workgroups as the chip keeps resident (256 CUs x blocks/CU by LDS / VGPR budget)
  constexpr int kResident = 256 * (NT > 256 ? 1 : ((BM + BN) * 256 > 48 * 1024 ? 2 : 4));
  const int zy = ksplit * p.nb1 * p.nb2;
  int gx = tiles;
  if ((long long)tiles * zy > kResident) gx = max(1, min(tiles, kResident / zy));
  dim3 grid(gx, ksplit, p.nb1 * p.nb2);
  const bool pw = p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0 && !p.upsample;
  constexpr int BK = 128 / (int)sizeof(T);
  const int ctot = p.c0 + p.c1;
  static const bool dma_off = getenv("SASPA_GEMM_DMA") && atoi(getenv("SASPA_GEMM_DMA")) == 0;   // A/B knob
  static const int abl = getenv("SASPA_GEMM_ABLATE") ? (atoi(getenv("SASPA_GEMM_ABLATE")) & 15) : 0;       // diagnostics only
  const int tiles_abl = tiles | (abl << 28);
  const int npart8 = saspa_gemm_npart8(p, BM, BN, gx, tiles);
  const bool fast = (ctot % BK) == 0 && (p.c1 == 0 || (p.c0 % BK) == 0) && !dma_off &&
                    (!p.upsample || (p.pad <= 1 && p.hin < 16000 && p.win < 16000));
  if (p.korder == SASPA_KORDER_CHUNK && !fast) return SASPA_ERANGE;   // only the DMA kernels walk K chunk-major
  if (p.w_split && (!fast || !is_x3<T>::value)) return SASPA_ERANGE;  // pre-split weights: only the SASPA_F32X3 DMA loop reads them
  if constexpr (NT != 256) {
    // 8-wave tiles exist only as DMA kernels; dispatch() guarantees `fast`
    if (!fast) return SASPA_ERANGE;
    if (pw) hipLaunchKernelGGL((gemm_dma_kernel<T, WM, WN, NWM, NWN, true, 3>), grid, dim3(NT), 0, s, p, tiles_abl, npart8);
    else if (p.upsample) hipLaunchKernelGGL((gemm_dma_kernel<T, WM, WN, NWM, NWN, false, 3, true>), grid, dim3(NT), 0, s, p, tiles_abl, npart8);
    else hipLaunchKernelGGL((gemm_dma_kernel<T, WM, WN, NWM, NWN, false, 3>), grid, dim3(NT), 0, s, p, tiles_abl, npart8);
  } else if (fast) {
    // few tiles (<= ~1 workgroup per CU): spend the idle LDS on a 4-deep DMA ring (latency-bound
    // K loops); otherwise 2 stages and 2 workgroups per CU
    static const int force_st = getenv("SASPA_GEMM_STAGES") ? atoi(getenv("SASPA_GEMM_STAGES")) : 0;
    constexpr bool can4 = (BM + BN) * 128 * 4 <= 160 * 1024;
    const bool deep = can4 && (force_st ? force_st == 4 : (long long)tiles * zy <= 320);
    if (deep) {
      if (pw) hipLaunchKernelGGL((gemm_dma_kernel<T, WM, WN, 2, 2, true, can4 ? 4 : 2>), grid, dim3(256), 0, s, p, tiles_abl, npart8);
      else if (p.upsample) hipLaunchKernelGGL((gemm_dma_kernel<T, WM, WN, 2, 2, false, can4 ? 4 : 2, true>), grid, dim3(256), 0, s, p, tiles_abl, npart8);
      else hipLaunchKernelGGL((gemm_dma_kernel<T, WM, WN, 2, 2, false, can4 ? 4 : 2>), grid, dim3(256), 0, s, p, tiles_abl, npart8);
    } else {
      if (pw) hipLaunchKernelGGL((gemm_dma_kernel<T, WM, WN, 2, 2, true, 2>), grid, dim3(256), 0, s, p, tiles_abl, npart8);
      else if (p.upsample) hipLaunchKernelGGL((gemm_dma_kernel<T, WM, WN, 2, 2, false, 2, true>), grid, dim3(256), 0, s, p, tiles_abl, npart8);
      else hipLaunchKernelGGL((gemm_dma_kernel<T, WM, WN, 2, 2, false, 2>), grid, dim3(256), 0, s, p, tiles_abl, npart8);
    }
  } else {
    if (pw) hipLaunchKernelGGL((gemm_kernel<T, WM, WN, true>), grid, dim3(256), 0, s, p, tiles, npart8);
    else hipLaunchKernelGGL((gemm_kernel<T, WM, WN, false>), grid, dim3(256), 0, s, p, tiles, npart8);
  }
  SASPA_CHECK_LAUNCH();
  if (ksplit > 1) return saspa_gemm_splitk_reduce(p, s, ksplit);
  return 0;
}

// K-split of the 8-wave kernel from a cost model instead of "about one workgroup per CU" (round 3): with t tiles and ks slices
// the launch takes ceil(t * ks / 256) ROUNDS of workgroups, each c0 + (K-tiles per slice) * c1 long, plus a reduce pass
// over ks fp32 slabs.  The old rule rounded 256 / t to the nearest integer, which at the tile counts of non-square images
// (88 / 96 / 176 tiles at 512x704 / 512x768) lands just ABOVE a whole number of rounds -- 264 workgroups = two rounds at
// half occupancy.  Constants fitted on tools/conv_variant_sweep.py (profiles/r3_conv_sweep_704.txt): c0 = 20 us per workgroup
// (prologue, epilogue, launch), c1 = 1.34 us per 64-deep K-tile of a 256 x 320 tile (x 0.8 for 256 x 256), reduce at 4 TB/s.
// The wave-specialised kernel runs one workgroup per CU: t tiles take ceil(t / 256) rounds.  352 tiles (the 16x22 level of a
// 512x704 image) would be two rounds at 69 %; the 4-wave tiles (two workgroups per CU) take those in one.  SASPA_GEMM_KSPLIT_MODEL=0
// restores the round-2 rule (any 256 <= t < 512).
static bool ws_round_ok(long long t) {
  static const bool model = !(getenv("SASPA_GEMM_KSPLIT_MODEL") && atoi(getenv("SASPA_GEMM_KSPLIT_MODEL")) == 0);
  if (!model) return true;
  const long long rounds = (t + 255) / 256;
  return t * 100 >= rounds * 256 * 85;
}

// wide (256-row) tiles from which a launch with a twin beside it (SaspaGemmParams.sharing) runs un-split on the 8-wave kernel:
// one constant for dispatch() and saspa_gemm_suggest_ksplit
constexpr long long kTwinWholeRound = 96;

// ncu: CUs the launch may count on -- 256, or 128 when a twin launch shares the chip (SaspaGemmParams.sharing)
static int pp_choose_ksplit(long long t, int ktiles, long long mn, int fn, int ncu = 256) {
  const double c0 = 20.0, c1 = fn == 5 ? 1.34 : 1.07;
  int best = 1;
  double best_cost = 1e30;
  for (int ks = 1; ks <= 8; ++ks) {
    if (ks > 1 && ktiles / ks < 8) break;
    const long long rounds = (t * ks + ncu - 1) / ncu;
    double cost = (double)rounds * (c0 + (double)((ktiles + ks - 1) / ks) * c1);
    if (ks > 1) cost += 5.0 + (4.0 * ks + 4.0) * (double)mn / 4.0e6;
    if (cost < best_cost - 1e-9) { best_cost = cost; best = ks; }
  }
  return best;
}

template <typename T>
int dispatch(const SaspaGemmParams& p, hipStream_t s) {
  const long long nb = (long long)p.nb1 * p.nb2;
  // A-stationary kernel (saspa_gemm_as.hip): pointwise bf16 layers with K = 320 and enough rows to fill the chip; the only
  // kernel that takes a fused LayerNorm / a transposed second output
  if constexpr (sizeof(T) == 2) {
    if (p.variant == SASPA_GEMM_AS) return p.defer_reduce ? SASPA_ERANGE : saspa_gemm_as_launch(p, s);      // (no slabs on this kernel)
    // AUTO: one predicate with the callers that plan around this choice (saspa_gemm_as_auto, saspa_gemm_as.hip)
    if (saspa_gemm_as_auto(&p)) return saspa_gemm_as_launch(p, s);
  }
  if (p.ln_gamma || p.out_t || p.variant == SASPA_GEMM_AS) return SASPA_ERANGE;
  int ksplit = (p.workspace && p.ksplit > 1 && nb == 1 && p.N % 4 == 0) ? p.ksplit : 1;
  // ABI 18 deferred reduce: the caller will hand `workspace` to saspa_splitk_groupnorm, which sums exactly p.ksplit slabs.  A
  // launch that ends up on ONE slice (or on a kernel that writes no slabs) would leave the workspace uninitialised and the
  // consumer would normalise garbage without any error: every branch below that drops the K split is closed to such a call,
  // and a call that cannot be split at all fails here (round-4 advisor finding).
  const bool must_split = p.defer_reduce != 0;
  if (must_split && (ksplit <= 1 || p.act == SASPA_ACT_GEGLU || p.N <= 32 || p.variant == SASPA_GEMM_WS)) return SASPA_ERANGE;
  const bool n160 = (p.N % 160) == 0;
  if (p.gn_stats) {
    // statistics need the LDS-staged bf16 epilogue on 160 / 320-column tiles of 128 / 256 rows (saspa_gemm checked the
    // shape): the 8-wave kernel where AUTO would take it, the 4-wave 128x160 tiles otherwise -- never the wave-specialised
    // kernel (its epilogue waves have no statistics pass) or the 64x64 tiles
    if constexpr (sizeof(T) == 2) {
      if (p.variant == SASPA_GEMM_WS) return SASPA_ERANGE;
      static const int pp_mode_g = getenv("SASPA_GEMM_PP") ? atoi(getenv("SASPA_GEMM_PP")) : 1;
      const bool can = nb == 1 && (p.N % 320) == 0 && saspa_gemm_pp_eligible(p);
      if (p.variant == SASPA_GEMM_WIDE) return can ? saspa_gemm_pp_launch(p, s, ksplit, 5) : SASPA_ERANGE;
      if (can && pp_mode_g != 0 && p.variant == SASPA_GEMM_AUTO && p.K >= 960) {
        const long long t = (long long)((p.M + 255) / 256) * (p.N / 320);
        static const bool model_g = !(getenv("SASPA_GEMM_KSPLIT_MODEL") && atoi(getenv("SASPA_GEMM_KSPLIT_MODEL")) == 0);
        // a twin launch beside this one (sharing): a whole round of the HALF chip is 128 tiles, whatever K (tools/twin_sweep.py)
        if (model_g && p.sharing && ksplit == 1 && t >= kTwinWholeRound) return saspa_gemm_pp_launch(p, s, 1, 5);
        if (model_g && ksplit == 1 && p.K >= 4096 && t >= 128) return saspa_gemm_pp_launch(p, s, 1, 5);
        if (!must_split && t >= (model_g ? 144 : 192) && (!model_g || ksplit == 1)) return saspa_gemm_pp_launch(p, s, 1, 5);
        static const bool wide_ks_g = !(getenv("SASPA_GEMM_WIDE_SPLITK") && atoi(getenv("SASPA_GEMM_WIDE_SPLITK")) == 0);
        if (wide_ks_g && ksplit > 1 && p.K >= 4096 && t >= 24 && t * ksplit >= 128) return saspa_gemm_pp_launch(p, s, ksplit, 5);
      }
      return launch<T, 4, 5>(p, s, ksplit);
    } else {
      return SASPA_ERANGE;
    }
  }
  if constexpr (sizeof(T) == 2) {
    if (p.variant == SASPA_GEMM_WS) return saspa_gemm_ws_launch(p, s);
  } else {
    if (p.variant == SASPA_GEMM_WS) return SASPA_ERANGE;
  }
  if (p.act == SASPA_ACT_GEGLU) {
    if constexpr (sizeof(T) == 2) {
      // level-0 projection (K = 320): 199 vs 214 us on the wave-specialised kernel; K >= 640 is faster on the 4-wave tiles
      static const bool ws_on_g = !(getenv("SASPA_GEMM_WS") && atoi(getenv("SASPA_GEMM_WS")) == 0);
      const long long t = (long long)((p.M + 127) / 128) * (p.N / (n160 ? 160 : 128));
      if (ws_on_g && p.variant == SASPA_GEMM_AUTO && nb == 1 && t >= 256 && p.K <= 384 && saspa_gemm_ws_eligible(p)) return saspa_gemm_ws_launch(p, s);
    }
    if constexpr (sizeof(T) == 2) {
      // long K and at least two waves of 256 x 320 tiles: the wide kernel (116 vs 142 us at (4096, 10240, 1280))
      static const bool wide_g = !(getenv("SASPA_GEMM_WIDE_GEGLU") && atoi(getenv("SASPA_GEMM_WIDE_GEGLU")) == 0);   // A/B knob
      // round 6: K >= 640 (was 1 024).  With the long-interval loop the level-1 projection (16384, 5120, 640) runs 117 us on the wide
      // kernel against 137-139 on the 4-wave tiles (175 vs 198-201 at the 512x704 bucket's 22 528 rows), alone and beside a twin
      // (pair 238 vs 270-275): tools/pointwise_dispatch_sweep.py, tools/twin_sweep.py -> profiles/r6_pointwise_sweep.txt
      static const int wide_g_k = getenv("SASPA_GEMM_WIDE_GEGLU_K") ? atoi(getenv("SASPA_GEMM_WIDE_GEGLU_K")) : 640;
      const bool want = p.variant == SASPA_GEMM_WIDE || (wide_g && p.variant == SASPA_GEMM_AUTO && p.K >= wide_g_k &&
                                                         (long long)((p.M + 255) / 256) * (p.N / 320) >= 384);
      if (want && nb == 1 && saspa_gemm_pp_eligible(p)) return saspa_gemm_pp_launch(p, s, 1, 5);
      if (p.variant == SASPA_GEMM_WIDE) return SASPA_ERANGE;
    }
    return n160 ? launch<T, 4, 5>(p, s, 1) : launch<T, 4, 4>(p, s, 1);
  }
  static const int force_tile = getenv("SASPA_GEMM_TILE") ? atoi(getenv("SASPA_GEMM_TILE")) : 0;   // tuning knob
  if (force_tile == 845) return launch<T, 4, 5, 4, 2>(p, s, ksplit);   // 256x160, 8 waves
  if (force_tile == 45) return launch<T, 4, 5>(p, s, ksplit);
  if (force_tile == 44) return launch<T, 4, 4>(p, s, ksplit);
  if (force_tile == 25) return launch<T, 2, 5>(p, s, ksplit);
  if (force_tile == 24) return launch<T, 2, 4>(p, s, ksplit);
  if (force_tile == 22) return launch<T, 2, 2>(p, s, ksplit);
  if (force_tile == 41) return launch<T, 4, 1>(p, s, ksplit);
  if (p.N <= 32) return launch<T, 4, 1>(p, s, 1);
  if constexpr (sizeof(T) == 2) {
    // long-K layers with enough 256-row tiles to fill the chip: 8-wave 256 x 320/256 kernel (saspa_gemm_pp.hip).
    // SASPA_GEMM_PP: 0 = never (A/B knob), 4 / 5 / 14 / 15 = force a tile / loop flavour (tools/pp_check.py)
    static const int pp_mode = getenv("SASPA_GEMM_PP") ? atoi(getenv("SASPA_GEMM_PP")) : 1;
    const bool can = nb == 1 && saspa_gemm_pp_eligible(p);
    if (p.variant == SASPA_GEMM_WIDE) {
      if (!can) return SASPA_ERANGE;
      return saspa_gemm_pp_launch(p, s, ksplit, (p.N % 320 == 0 || p.N % 256 != 0) ? 5 : 4);
    }
    if (can && pp_mode >= 4) return saspa_gemm_pp_launch(p, s, ksplit, pp_mode);
    // round 6: short K (640 <= K < 960: the level-1 pointwise layers) is admitted where the sweeps of the long-interval loop show
    // the wide kernel ahead -- beside a twin from a whole round of the half chip on (pairs: (16384, 640, 640) + residual 38 vs 50 us,
    // (16384, 1920, 640) 84 vs 100), alone from 144 tiles on ((16384, 1920, 640): 50.6 vs 53.9; at 128 tiles the 4-wave kernel keeps
    // (16384, 640, 640): 25.4 vs 29.7).  SASPA_GEMM_WIDE_KMIN=960 restores the round-5 gate.
    static const int wide_kmin = getenv("SASPA_GEMM_WIDE_KMIN") ? atoi(getenv("SASPA_GEMM_WIDE_KMIN")) : 640;
    if (can && pp_mode == 1 && p.variant == SASPA_GEMM_AUTO && p.K >= wide_kmin) {
      const int fn = (p.N % 320 == 0) ? 5 : (p.N % 256 == 0) ? 4 : 0;
      const long long t0 = fn ? (long long)((p.M + 255) / 256) * (p.N / (64 * fn)) : 0;
      // (alone: from 144 tiles, the un-split threshold below -- (22528, 640, 640) + residual, 176 tiles: 31.9 vs 44.6 us; 128 tiles, the
      // 512x512 case, stay on the 4-wave kernel: 25.4 vs 29.7)
      const bool short_ok = p.K >= 960 || (ksplit == 1 && !must_split && (p.sharing ? t0 >= kTwinWholeRound : t0 >= 144));
      if (fn && short_ok) {
        const long long t = t0;
        static const bool model = !(getenv("SASPA_GEMM_KSPLIT_MODEL") && atoi(getenv("SASPA_GEMM_KSPLIT_MODEL")) == 0);   // A/B knob
        // a twin launch of the same shape shares the chip (SaspaGemmParams.sharing): 96+ wide tiles are a whole round of this
        // launch's half, for every K the wide kernel takes -- pair times of tools/twin_sweep.py (profiles/r4_twin_sweep.txt):
        // (16384, 640, 2560) + residual 112 vs 152 us, conv (16384, 640, 2880) 106 vs 139, conv (16384, 640, 5760) 183 vs 270
        if (model && p.sharing && ksplit == 1 && t >= kTwinWholeRound) return saspa_gemm_pp_launch(p, s, 1, fn);
        // long K, one slice chosen by the cost model (suggest_ksplit) and at least half the CUs busy: still the wide kernel
        if (model && ksplit == 1 && p.K >= 4096 && t >= 128) return saspa_gemm_pp_launch(p, s, 1, fn);
        // 3/4 of a wave of tiles or more: no split-K.  (144 <= t < 192 is the 512x704 / 512x768 buckets' 32x44 / 32x48 level:
        // 176 / 192 workgroups in one round against 704 / 768 4-wave tiles in two rounds of 512 slots -- 74 vs 124 us at
        // (22528, 640, 2560); t = 128, the 512x512 case, ties and stays on the 4-wave tiles)
        // ... unless the tiles are just over a whole number of rounds (264 / 288 tiles of the fused Q|K|V projection of the 16x22 /
        // 16x24 level: two rounds at 52-56 %, 99 us against 81 on the 4-wave tiles): then the finer tiles below
        const bool ragged = model && t > 256 && t * 100 < ((t + 255) / 256) * 256 * 65;
        if (!must_split && !ragged && t >= (model ? 144 : 192) && (!model || ksplit == 1)) return saspa_gemm_pp_launch(p, s, 1, fn);
        // fewer wide tiles than CUs but a long K (the 3x3 convs of the 32x32 / 16x16 levels): the wide kernel on K
        // slices -- measured 1.17-1.45x the 128x160 kernel at M = 16 384 / 4 096 (tools/conv_variant_sweep.py)
        static const bool wide_ks = !(getenv("SASPA_GEMM_WIDE_SPLITK") && atoi(getenv("SASPA_GEMM_WIDE_SPLITK")) == 0);   // A/B knob
        if (wide_ks && ksplit > 1 && p.K >= 4096 && t >= 24 && t * ksplit >= 128) return saspa_gemm_pp_launch(p, s, ksplit, fn);
      }
    }
  } else {
    if (p.variant == SASPA_GEMM_WIDE) return SASPA_ERANGE;
  }
  const int bn = n160 ? 160 : 128;
  if constexpr (sizeof(T) == 2) {
    // exactly one wave of 128-row tiles (256 <= tiles < 512: the 16x16-level linears, where the 2-workgroups-per-CU kernel
    // runs half empty) and no K slices: the wave-specialised kernel (saspa_gemm_ws.hip) -- 19 vs 24 us at (4096, 1280,
    // 1280), 52 vs 72 us at (4096, 1280, 5120); with more tiles it only ties the 4-wave kernel (both sit at the L2 -> LDS
    // fill rate of a 128x160 tile, tools/ws_stamps.py) and, owning the CU's whole LDS, it keeps the other graph branch's
    // kernels off the CU (bench 6.07 vs 6.14 images/s when used everywhere).  SASPA_GEMM_WS=0 turns it off (A/B knob).
    static const bool ws_on = !(getenv("SASPA_GEMM_WS") && atoi(getenv("SASPA_GEMM_WS")) == 0);
    static const int ws_max = getenv("SASPA_GEMM_WS_MAXTILES") ? atoi(getenv("SASPA_GEMM_WS_MAXTILES")) : 512;
    if (ws_on && p.variant == SASPA_GEMM_AUTO && ksplit == 1 && nb == 1 && saspa_gemm_ws_eligible(p)) {
      const long long t = (long long)((p.M + 127) / 128) * ((p.N + bn - 1) / bn);
      if (t >= 256 && t < ws_max && ws_round_ok(t)) return saspa_gemm_ws_launch(p, s);
    }
  }
  const long long tiles = (long long)((p.M + 127) / 128) * ((p.N + bn - 1) / bn) * nb * ksplit;
  if (tiles >= 160 && p.N > 64) return n160 ? launch<T, 4, 5>(p, s, ksplit) : launch<T, 4, 4>(p, s, ksplit);
  return launch<T, 2, 2>(p, s, ksplit);
}

}  // namespace

// Recommended K-split of a problem (1 = none): the caller sizes the fp32 workspace (ksplit*M*N floats) from it.
extern "C" int saspa_gemm_suggest_ksplit(const SaspaGemmParams* pp) {
  if (!pp) return 1;
  const SaspaGemmParams& p = *pp;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || (long long)p.nb1 * p.nb2 > 1 || p.N % 4 || p.act == SASPA_ACT_GEGLU) return 1;
  const int bk = p.dtype == SASPA_BF16 ? 64 : 32;
  const int ktiles = (p.K + bk - 1) / bk;
  static const bool wide_ks = !(getenv("SASPA_GEMM_WIDE_SPLITK") && atoi(getenv("SASPA_GEMM_WIDE_SPLITK")) == 0);       // A/B knob
  {
    // exactly one wave of 128-row tiles on a pointwise layer: the wave-specialised kernel without K slices beats the 8-wave
    // kernel on K slices (feed-forward output projection of the 16x16 level, (4096, 1280, 5120): 52 vs 76 us, and no slabs)
    static const bool ws_on = !(getenv("SASPA_GEMM_WS") && atoi(getenv("SASPA_GEMM_WS")) == 0) &&
                              !(getenv("SASPA_GEMM_WS_NOSPLIT") && atoi(getenv("SASPA_GEMM_WS_NOSPLIT")) == 0);   // A/B knobs
    static const int ws_max = getenv("SASPA_GEMM_WS_MAXTILES") ? atoi(getenv("SASPA_GEMM_WS_MAXTILES")) : 512;
    const int bn_t = (p.N % 160 == 0) ? 160 : 128;
    const long long t128 = (long long)((p.M + 127) / 128) * ((p.N + bn_t - 1) / bn_t);
    // (not with a twin launch beside a long-K layer: there two K slices on the 8-wave kernel win, pair times 124 vs 146 us at
    // (4096, 1280, 5120) + residual, tools/twin_sweep.py)
    if (ws_on && !(p.sharing && p.K >= 4096) && !p.gn_stats && p.dtype == SASPA_BF16 && p.variant == SASPA_GEMM_AUTO && p.kh == 1 && p.kw == 1 &&
        t128 >= 256 && t128 < ws_max && ws_round_ok(t128) && saspa_gemm_ws_eligible(p))
      return 1;
  }
  if (wide_ks && p.dtype == SASPA_BF16 && p.K >= 4096 && saspa_gemm_pp_eligible(p)) {
    const int fn = (p.N % 320 == 0) ? 5 : (p.N % 256 == 0) ? 4 : 0;
    if (fn) {
      const long long t = (long long)((p.M + 255) / 256) * (p.N / (64 * fn));
      static const bool model = !(getenv("SASPA_GEMM_KSPLIT_MODEL") && atoi(getenv("SASPA_GEMM_KSPLIT_MODEL")) == 0);   // A/B knob
      if (!model) {
        if (t >= 192) return 1;                     // the wide kernel fills the chip without slicing K
        if (t >= 24) {
          int ks = (int)((256 + t / 2) / t);        // about one workgroup per CU
          ks = ks < 2 ? 2 : (ks > 8 ? 8 : ks);
          while (ks > 1 && ktiles / ks < 8) --ks;
          if (ks > 1) return ks;
        }
      } else if (t >= 24) {
        // rounds x slice length + reduce: pp_choose_ksplit; a single slice is taken on the wide kernel too when it leaves at
        // least half the CUs busy (dispatch() applies the same rule).  With a twin launch beside it (sharing) the launch
        // counts on half the chip: 128 tiles need no slices, 64 tiles two instead of four
        const int ks = pp_choose_ksplit(t, ktiles, (long long)p.M * p.N, fn, p.sharing ? 128 : 256);
        // one slice only where dispatch() really takes the wide kernel un-split (kTwinWholeRound tiles with a twin, 128 alone);
        // below that a one-slice answer would land on the 4-wave tiles un-split (round-4 advisor finding): fall through
        if (ks > 1 || t >= (p.sharing ? kTwinWholeRound : 128)) return ks;
      }
    }
  }
  // 4-wave tiles: enough K slices to give the 256 CUs about two workgroups each, for the deep levels only
  const int bn = (p.N % 160 == 0) ? 160 : 128;
  const long long tiles = (long long)((p.M + 127) / 128) * ((p.N + bn - 1) / bn);
  if (p.N <= 64 || tiles >= 512 || ktiles < 32) return 1;
  long long ks = (512 + tiles - 1) / tiles;
  if (ks > 8) ks = 8;
  if (ks > ktiles / 8) ks = ktiles / 8;
  return ks < 1 ? 1 : (int)ks;
}

// Tile order of a launch: 0 = M-partitioned (an XCD walks whole rows of N tiles: its activations stay in L2, the weights
// stream through once per row block) or n = nbn / 8 > 0 = N-partitioned (an XCD owns an eighth of the N tiles for every row
// block: its weight slice stays in L2, the activations stream through once per XCD).  Estimated beyond-L2 bytes decide;
// profiles/r2_pmc_per_shape.txt has the measured ones (GEGLU projection at M = 16 384: 684 MB fetched for 28 MB of operands
// with the M-partitioned order).  SASPA_GEMM_NPART=0 turns it off (A/B knob).
int saspa_gemm_npart8(const SaspaGemmParams& p, int BM, int BN, int G, int tiles) {
  static const bool off = getenv("SASPA_GEMM_NPART") && atoi(getenv("SASPA_GEMM_NPART")) == 0;
  const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
  if (off || (long long)p.nb1 * p.nb2 != 1 || (nbn & 7) || (G & 7) || G <= 0 || tiles % G || p.N % BN) return 0;
  const double esz = p.dtype == SASPA_BF16 ? 2.0 : 4.0;
  const double a = (double)p.batch * p.hin * p.win * (p.c0 + p.c1) * esz;     // the input tensor (taps re-read from L2)
  const double w = (double)p.N * p.K * esz;
  static const double l2mb = getenv("SASPA_GEMM_NPART_L2MB") ? atof(getenv("SASPA_GEMM_NPART_L2MB")) : 3.8;   // of the 4 MiB per XCD
  const double l2 = l2mb * (1 << 20);
  const double g8 = G / 8.0;
  const double mpart = a * (nbn > g8 ? nbn / g8 : 1.0) + w * (w <= l2 ? 8.0 : (double)nbm);
  const double npart = 8.0 * a + w * (w / 8.0 <= l2 ? 1.0 : (double)nbm);
  return npart < 0.75 * mpart ? nbn / 8 : 0;
}

int saspa_gemm_splitk_reduce(const SaspaGemmParams& p, hipStream_t s, int ksplit) {
  if (p.defer_reduce) return 0;      // ABI 18: the slabs go to saspa_splitk_groupnorm
  if (p.gn_stats) {        // saspa_gemm checked: bf16, N % 160 == 0, 160 % gn_unit == 0
    hipLaunchKernelGGL(splitk_reduce_stats_kernel, dim3((p.M + 127) / 128, p.N / 80), dim3(256), 0, s, p, ksplit);
    SASPA_CHECK_LAUNCH();
    return 0;
  }
  long long blocks = ((long long)p.M * (p.N / 4) + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (p.dtype == SASPA_BF16) hipLaunchKernelGGL((splitk_reduce_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, s, p, ksplit);
  else hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, p, ksplit);
  SASPA_CHECK_LAUNCH();
  return 0;
}

/* ABI 20: which kernel family saspa_gemm would run `p` on, and on how many K slices -- the dispatch itself, executed dry (nothing is
 * launched).  Returns family | (ksplit << 8) with family = SASPA_GEMM_TILED / WIDE / WS / AS, or the SASPA_E* code saspa_gemm would
 * return.  bench.py attributes every recorded launch to its kernel with this (roofline.by_kernel / roofline.dominant). */
extern "C" int saspa_gemm_which(const SaspaGemmParams* pp) {
  SaspaDryRun* st = saspa_dry_state();
  *st = {true, 0, 1};
  const int rc = saspa_gemm(pp, nullptr);
  const SaspaDryRun d = *st;
  *st = {false, 0, 0};
  if (rc != 0) return rc;
  return d.family | (d.ksplit << 8);
}

extern "C" int saspa_gemm(const SaspaGemmParams* pp, void* stream) {
  if (!pp) return SASPA_EINVAL;
  SaspaGemmParams p = *pp;
  if (p.nb1 <= 0) p.nb1 = 1;
  if (p.nb2 <= 0) p.nb2 = 1;
  if (!p.a0 || !p.w || !p.out) return SASPA_EINVAL;
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.batch <= 0) return SASPA_EINVAL;
  if (p.dtype != SASPA_BF16 && p.dtype != SASPA_F32 && p.dtype != SASPA_F32X3) return SASPA_EINVAL;
  if (p.kh <= 0 || p.kw <= 0 || p.stride <= 0 || p.pad < 0) return SASPA_EINVAL;
  if (p.c1 > 0 && !p.a1) return SASPA_EINVAL;
  if (p.c1 < 0 || p.c0 <= 0) return SASPA_EINVAL;
  const int epc = p.dtype == SASPA_BF16 ? 8 : 4;
  if (p.c0 % epc || p.c1 % epc || p.lda0 % epc || (p.c1 > 0 && p.lda1 % epc) || p.ldw % epc) return SASPA_EALIGN;
  if (p.ldo % 4 || (p.residual && p.ldr % 4)) return SASPA_EALIGN;
  if (!aligned16(p.a0) || !aligned16(p.w) || !aligned16(p.out) || (p.a1 && !aligned16(p.a1)) ||
      (p.residual && !aligned16(p.residual)) || (p.bias && !aligned16(p.bias)) || (p.rowvec && !aligned16(p.rowvec)))
    return SASPA_EALIGN;
  if (p.sa1 % epc || p.sa2 % epc || p.sw1 % epc || p.sw2 % epc || p.so1 % 4 || p.so2 % 4) return SASPA_EALIGN;
  if (p.rowvec && (p.ldrv % 4)) return SASPA_EALIGN;
  if (p.K != p.kh * p.kw * (p.c0 + p.c1)) return SASPA_ERANGE;
  if ((long long)p.batch * p.hout * p.wout != p.M) return SASPA_ERANGE;
  if (p.lda0 < p.c0 || (p.c1 > 0 && p.lda1 < p.c1) || p.ldw < p.K) return SASPA_ERANGE;
  if (p.act != SASPA_ACT_GEGLU && p.ldo < (p.out_t ? p.n_split : p.N)) return SASPA_ERANGE;
  if (p.ln_gamma && !p.ln_beta) return SASPA_EINVAL;
  if (p.out_t) {      // transposed tail columns (ABI 13): geometry is checked here, eligibility of the kernel in dispatch()
    if (p.n_split < 0 || p.n_split > p.N || p.rows_per_batch <= 0 || p.M % p.rows_per_batch || p.ldt < p.rows_per_batch) return SASPA_ERANGE;
    if (p.M / p.rows_per_batch > 1 && p.st < (long long)(p.N - p.n_split) * p.ldt) return SASPA_ERANGE;
    if ((reinterpret_cast<uintptr_t>(p.out_t) & 1u)) return SASPA_EALIGN;
  }
  // the window of every output pixel must come from the declared input extent
  {
    const int hv = p.upsample ? 2 * p.hin : p.hin, wv = p.upsample ? 2 * p.win : p.win;
    if ((p.hout - 1) * p.stride - p.pad >= hv || (p.wout - 1) * p.stride - p.pad >= wv) return SASPA_ERANGE;
  }
  if ((long long)p.batch * p.hin * p.win >= (1ll << 31)) return SASPA_ERANGE;
  {
    // buffer loads address every operand with 32-bit byte offsets below 2 GiB
    const long long esz = p.dtype == SASPA_BF16 ? 2 : 4;
    const long long a0b = (long long)p.batch * p.hin * p.win * p.lda0 * esz;
    const long long a1b = p.c1 > 0 ? (long long)p.batch * p.hin * p.win * p.lda1 * esz : 0;
    const long long wb = (long long)p.N * p.ldw * esz;
    if (a0b >= (1ll << 31) || a1b >= (1ll << 31) || wb >= (1ll << 31)) return SASPA_ERANGE;
    // the tap-validity bitmask of the fast loaders (a K-tile inside one tap) holds 31 taps; larger windows (the 7x7 stem of
    // the filter stage's ResNet, 3 -> 8 padded channels) run on the generic per-lane loader, which has no mask
    const int bk = p.dtype == SASPA_BF16 ? 64 : 32;
    const bool fastpath = ((p.c0 + p.c1) % bk) == 0 && (p.c1 == 0 || (p.c0 % bk) == 0);
    if (p.kh * p.kw > 31 && fastpath) return SASPA_ERANGE;
  }
  if (p.ksplit < 0 || p.ksplit > 64) return SASPA_ERANGE;
  if (p.korder != SASPA_KORDER_TAP && p.korder != SASPA_KORDER_CHUNK) return SASPA_EINVAL;
  if (KORDER_ON && p.korder == SASPA_KORDER_CHUNK) {
    const int bk = p.dtype == SASPA_BF16 ? 64 : 32;
    if (p.c0 % bk || p.c1 % bk) return SASPA_ERANGE;
  }
  if (p.variant < SASPA_GEMM_AUTO || p.variant > SASPA_GEMM_AS) return SASPA_EINVAL;
  if (p.act == SASPA_ACT_GEGLU) {
    // fused GEGLU: bf16 only, whole tiles, weights pre-interleaved per tile (see header)
    const int bn = (p.N % 160) == 0 ? 160 : 128;
    if (p.dtype != SASPA_BF16 || p.N % bn || p.residual || p.ldo % 8 || p.ldo < p.N / 2) return SASPA_ERANGE;
    p.ksplit = 1;
  } else if (p.act != SASPA_ACT_NONE && p.act != SASPA_ACT_SILU && p.act != SASPA_ACT_RELU && p.act != SASPA_ACT_ADD_RELU) {
    return SASPA_EINVAL;
  }
  if (p.ksplit > 1 && p.workspace && !aligned16(p.workspace)) return SASPA_EALIGN;
  if (p.gn_stats) {
    // epilogue GroupNorm statistics (ABI 12): bf16, whole 160-column tiles whose first column is a multiple of the unit
    if (p.dtype != SASPA_BF16 || p.act == SASPA_ACT_GEGLU || (long long)p.nb1 * p.nb2 != 1) return SASPA_ERANGE;
    if (p.gn_unit < 2 || p.gn_unit > 16 || (p.gn_unit & 1) || (80 % p.gn_unit) != 0 || (p.N % 160) != 0) return SASPA_ERANGE;
    if ((p.ldo % 8) != 0 || (p.residual && (p.ldr % 8) != 0)) return SASPA_ERANGE;
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (p.dtype == SASPA_BF16) return dispatch<bf16_t>(p, s);
  if (p.dtype == SASPA_F32X3) return dispatch<f32x3_t>(p, s);
  return dispatch<float>(p, s);
}
