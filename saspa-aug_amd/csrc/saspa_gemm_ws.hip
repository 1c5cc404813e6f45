// Wave-specialised implicit GEMM for the SHORT-K bf16 layers (transformer projections, fused-GEGLU feed-forward, 1x1
// convs, K = 320 .. 1280): one 12-wave workgroup per CU, 128 x 160 (or 128 x 128) output tile.
//
// Why a third kernel: on these layers an output tile is only 5 .. 20 K-tiles long, so the epilogue (bias, activation,
// GEGLU math, staging through LDS, 16-byte stores) is HALF of the 4-wave kernel's time and does not overlap its K loop
// (tools/geglu_ablate.py: barriers-only 100 us + DMA 47 + LDS reads 25 + MFMA 34 = full 211 us for the level-0 GEGLU
// projection -- the parts ADD; two workgroups per CU that happen to be in the same phase contend instead of overlapping).
// Here the two kinds of work are given to different waves of one workgroup and kept in lock-step by the workgroup barrier:
//
//   waves 0..3   "MMA":      per K-tile  s_barrier | fragment reads (ds_read_b128) | 40 MFMA          -- nothing else
//   waves 4..7   "loader":   per K-tile  s_barrier | LDS-DMA issue of the K-tile two steps ahead (9 pieces) | vmcnt(9)
//   waves 8..11  "epilogue": per K-tile  s_barrier | slice of the PREVIOUS output tile: LDS staging -> activation / GEGLU /
//                            residual -> 16-byte global stores
//   (measured with s_memtime stamps, tools/ws_stamps.py: issuing the 9 one-KB pieces costs a wave ~1150 cycles, more than
//   the 640 cycles of the K-tile's MFMAs -- the L2 -> LDS fill of a 128 x 160 tile is what bounds a K-tile, so the pieces
//   get waves of their own and nothing else is put in their way)
//
//   * 3-stage LDS ring that never drains: the K-tile sequence runs across output tiles (persistent workgroup), K-tile g+2
//     is DMAed into the slot K-tile g-1 was read from, one s_barrier per K-tile.
//   * at the end of a tile the MMA waves add bias / time-embedding row, scale, round to bf16 and park the tile in the LDS
//     staging area (one extra barrier so the LE waves are done with the previous tile's staging); the LE waves drain it
//     over the next tile's K-tiles, so the epilogue's VALU / store work sits beside the next tile's MFMAs (different
//     pipes of the same SIMDs).
//   * a loader wave issues exactly 9 DMA pieces per K-tile and nothing else (out-of-range offsets past the end), so
//     "K-tile g+1 has landed" is the constant vmcnt(9); the epilogue waves use ordinary loads / stores, whose waits the
//     compiler places exactly (vmcnt is one in-order counter: mixing the two kinds in a wave would make every residual
//     load wait for the DMA pieces issued before it).
//
// Same operand layout / swizzle / XCD-aware tile order / GEGLU weight packing as gemm_dma_kernel (saspa_gemm.hip).
// Not eligible (dispatch falls back to the 4-wave kernel): fp32, split-K, batched problems, layers
// whose K-tile straddles taps / sources, nearest-x2 input.
#include <cstdlib>

#include "common.h"
#include "gemm_internal.h"

namespace {

typedef bf16_t T;

template <int WN, bool PW, bool GEGLU>
__global__ __launch_bounds__(768) void gemm_ws_kernel(const SaspaGemmParams p, const int ntiles_abl) {
  // diagnostic ablation (SASPA_GEMM_ABLATION builds only): 1 = no fragment reads / MFMA, 2 = no DMA issue, 4 = no epilogue
  const int ntiles = ntiles_abl & 0x0fffffff;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef SASPA_GEMM_ABLATION
  const int abl = (ntiles_abl >> 28) & 7;
  const bool stamp = ((ntiles_abl >> 31) & 1) && blockIdx.x == 0 && p.workspace && lane == 0 && (wave == 0 || wave == 4);   // MMA wave 0, loader wave 4
  unsigned long long* sbuf = reinterpret_cast<unsigned long long*>(p.workspace) + (wave == 4 ? 512 : 0);
  int sidx = 0;
#define STAMP(tag) do { if (stamp && sidx < 500) { sbuf[sidx++] = (__builtin_amdgcn_s_memtime() << 8) | (tag); } } while (0)
#else
#define STAMP(tag) do { } while (0)
  constexpr int abl = 0;
#endif
  constexpr int WM = 4, BM = 128, BN = 32 * WN, BK = 64, NSTAGE = 3, SZ = 2;
  constexpr int STAGE = (BM + BN) * 8;                 // u32x4 per ring slot
  constexpr int CP = BN + 8;                           // staging row pitch (elements)
  constexpr int EPI = BM * CP * SZ / 16;               // u32x4 of the staged output tile
  constexpr int A_CH = 4, B_CH = WN;                   // DMA pieces per LE wave and K-tile: 32 tile rows per instruction of 4 waves
  constexpr int NPIECE = A_CH + B_CH;
  constexpr int NBIAS = 2560;                          // fp32 bias of ALL N columns, resident for the workgroup's lifetime (10 KB)
  constexpr int BIAS4 = (NSTAGE * STAGE + EPI) * 16 + NBIAS * 4 <= 160 * 1024 ? NBIAS / 4 : 0;
  __shared__ u32x4 lds[NSTAGE * STAGE + EPI + BIAS4];
  u32x4* const staging = lds + NSTAGE * STAGE;
  float* const bias_lds = reinterpret_cast<float*>(lds + NSTAGE * STAGE + EPI);

  const bool is_mma = wave < 4;

  const int nbn = (p.N + BN - 1) / BN;
  const int G = gridDim.x;
  int tile0;
  {
    const int L = blockIdx.x;
    const int qd = G >> 3, rr = G & 7, xcd = L & 7, idx = L >> 3;
    tile0 = (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + idx;
  }
  if (tile0 >= ntiles) return;                         // (grid <= ntiles: never taken; keeps the barrier counts honest)
  const int nt_wg = (ntiles - tile0 + G - 1) / G;      // output tiles of this workgroup: tile0, tile0 + G, ...
  const int nk = (p.K + BK - 1) / BK;
  const int total = nt_wg * nk;                        // K-tile steps of this workgroup
  const int hw = p.hout * p.wout;
  constexpr bool geglu = GEGLU;
  // epilogue iterations (16-byte output chunks per LE thread and tile), spread over the nk steps of the next tile
  constexpr int cpr = GEGLU ? BN / 16 : BN / 8;
  constexpr int iters = BM * cpr / 256;
  const int ips = (iters + nk - 1) / nk;               // per step (uniform: masked beyond `iters`)
  // bias resident in LDS (no per-image row vector, N small enough): the MMA waves' tile-end epilogue then costs no global
  // round trip (it was ~800 exposed cycles per tile: tools/ws_stamps.py)
  const bool bias_in_lds = BIAS4 > 0 && p.bias && !p.rowvec && p.N <= NBIAS;
  if (bias_in_lds && wave >= 8) {
    for (int i = tid - 512; i < p.N; i += 256) bias_lds[i] = p.bias[i];
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // visible before B(0); the first reader is a tile later
  }

  if (is_mma) {
    // =============================== MMA waves ===============================
    const int wm = wave >> 1, wn = wave & 1;
    const int frow = lane & 15, fg = lane >> 4;
    f32x4 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    int ti = 0, kt = 0;
    int cbm = 0, cbn = 0, mnext = 0;
    bool straddle = false;
    float4 add[WN];
#pragma unroll
    for (int j = 0; j < WN; ++j) add[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int g = 0; g < total; ++g) {
      STAMP(1);
      __builtin_amdgcn_s_barrier();                    // B(g): K-tile g has landed (LE waited for it), staging of the previous tile is filled
      asm volatile("" ::: "memory");
      STAMP(2);
      if (kt == nk - 1) {
        // last K-tile of output tile ti: its bias / row-vector values are fetched now, under this step's MFMAs
        const int t = tile0 + ti * G;
        cbm = t / nbn;
        cbn = t - cbm * nbn;
        const int m0 = cbm * BM + wm * (16 * WM);
        const int img0 = min(m0, p.M - 1) / hw;
        mnext = (img0 + 1) * hw;
        straddle = p.rowvec && (m0 + 16 * WM > mnext);
#pragma unroll
        for (int j = 0; j < WN; ++j) {
          const int n = cbn * BN + wn * (16 * WN) + j * 16 + fg * 4;
          add[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (n < p.N) {
            if (bias_in_lds) add[j] = *reinterpret_cast<const float4*>(bias_lds + n);
            else if (p.bias) add[j] = *reinterpret_cast<const float4*>(p.bias + n);
            if (p.rowvec) {
              const float4 r4 = *reinterpret_cast<const float4*>(p.rowvec + (long long)img0 * p.ldrv + n);
              add[j].x += r4.x; add[j].y += r4.y; add[j].z += r4.z; add[j].w += r4.w;
            }
          }
        }
      }
      const u32x4* la = lds + (g % NSTAGE) * STAGE;
      const u32x4* lb = la + BM * 8;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if (abl & 1) break;
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
          for (int j = 0; j < WN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wb[j]), __builtin_bit_cast(bf16x8, xa[i]),
                                                                acc[i][j], 0, 0, 0);
      }
      if (++kt == nk) {
        // ---- end of output tile ti: park it (bias + row vector, alpha, bf16) in the staging area ----
        kt = 0;
        ++ti;
        STAMP(3);
        __builtin_amdgcn_s_barrier();                  // Bx: the LE waves have read the previous tile's staging completely
        asm volatile("" ::: "memory");
        STAMP(4);
        T* ct = reinterpret_cast<T*>(staging);
        if (!straddle && p.alpha == 1.0f) {            // every layer but the ControlNet zero convs: x * 1.0f is the identity
#pragma unroll
          for (int i = 0; i < WM; ++i) {
            const int mrow = wm * (16 * WM) + i * 16 + frow;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
              const int ncol = wn * (16 * WN) + j * 16 + fg * 4;
              float v[4] = {acc[i][j][0] + add[j].x, acc[i][j][1] + add[j].y, acc[i][j][2] + add[j].z, acc[i][j][3] + add[j].w};
              Elem<T>::store4(ct + mrow * CP + ncol, v);
              acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
          }
        } else if (!straddle) {
#pragma unroll
          for (int i = 0; i < WM; ++i) {
            const int mrow = wm * (16 * WM) + i * 16 + frow;
#pragma unroll
            for (int j = 0; j < WN; ++j) {
              const int ncol = wn * (16 * WN) + j * 16 + fg * 4;
              float v[4] = {(acc[i][j][0] + add[j].x) * p.alpha, (acc[i][j][1] + add[j].y) * p.alpha,
                            (acc[i][j][2] + add[j].z) * p.alpha, (acc[i][j][3] + add[j].w) * p.alpha};
              Elem<T>::store4(ct + mrow * CP + ncol, v);
              acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
          }
        } else {
          // rare: the wave's rows belong to more than one image and a per-image row vector is added (H*W < 64 rows)
#pragma unroll
          for (int i = 0; i < WM; ++i) {
            const int mrow = wm * (16 * WM) + i * 16 + frow;
            const int m = cbm * BM + mrow;
            const bool other = m >= mnext && m < p.M;
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
              for (int r = 0; r < 4; ++r) v[r] *= p.alpha;
              Elem<T>::store4(ct + mrow * CP + ncol, v);
              acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staging writes done before the next B(g) releases the LE waves
        STAMP(5);
      }
    }
    __builtin_amdgcn_s_barrier();                      // B(end): last tile's staging is filled
    return;
  }

  if (wave < 8) {
    // ================================ loader waves ================================
    const int ltid = tid - 256;
    const int lw = wave - 4;
    const int r0 = ltid >> 3;                            // tile row r0 + 32*i lands in LDS row r0 + 32*i
    const int kcs = (ltid & 7) ^ (r0 & 7);               // logical 16-byte chunk fetched by this lane (source-side swizzle)
    const int ctot = p.c0 + p.c1;
    const T* a0 = reinterpret_cast<const T*>(p.a0);
    const T* a1 = reinterpret_cast<const T*>(p.a1);
    const T* w = reinterpret_cast<const T*>(p.w);
    const rsrc_t rs0 = make_rsrc(a0);
    const rsrc_t rs1 = make_rsrc(p.c1 > 0 ? (const void*)a1 : (const void*)a0);
    const rsrc_t rsw = make_rsrc(w);
    const int chunk_major = p.korder == SASPA_KORDER_CHUNK ? 1 : 0;
    int msk[A_CH];
    unsigned offa0[A_CH], offa1[A_CH], offb[B_CH];
    int ku = 0, cu = 0, dyu = 0, dxu = 0;
    int iti = 0, ikt = 0;                                // (tile, K-tile) the NEXT DMA issue belongs to

    auto setup_tile = [&](int t) __attribute__((always_inline)) {
      const int bm = t / nbn, bn = t - bm * nbn;
      if (PW) {
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
          const int m = bm * BM + r0 + 32 * i;
          msk[i] = (m < p.M) ? 1 : 0;
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
          const int pix = b * p.hin * p.win + (oy * p.stride) * p.win + ox * p.stride;
          offa0[i] = (unsigned)(pix * (p.lda0 * SZ) + kcs * 16);
          offa1[i] = (unsigned)(pix * (p.lda1 * SZ) + kcs * 16);
          int mask = 0;
          if (m < p.M) {
            for (int ty = 0; ty < p.kh; ++ty)
              for (int tx = 0; tx < p.kw; ++tx)
                if ((unsigned)(iy0 + ty) < (unsigned)p.hin && (unsigned)(ix0 + tx) < (unsigned)p.win) mask |= 1 << (ty * p.kw + tx);
          }
          msk[i] = mask;
          m += 32;
          ox += 32;
          while (ox >= p.wout) { ox -= p.wout; ++oy; }
          while (oy >= p.hout) { oy -= p.hout; ++b; }
        }
      }
#pragma unroll
      for (int i = 0; i < B_CH; ++i) {
        const int n = bn * BN + r0 + 32 * i;
        offb[i] = (n < p.N) ? (unsigned)(n * p.ldw * SZ + kcs * 16) : kInvalid;
      }
      ku = 0;
      cu = 0;
      dyu = 0;
      dxu = 0;
    };

    // one K-tile (NPIECE pieces per wave, always) into ring slot `slot`; `live` = false past the end: zeros, nothing reads them
    auto dma_step = [&](int slot, bool live) __attribute__((always_inline)) {
      if (abl & 2) return;
      const bool s0 = cu < p.c0;
      const rsrc_t rs = s0 ? rs0 : rs1;
      const int ldsz = (s0 ? p.lda0 : p.lda1) * SZ;
      const int soff = (s0 ? cu : cu - p.c0) * SZ;
      const int tapbit = dyu * p.kw + dxu;
      const int pixoff = PW ? 0 : (dyu - p.pad) * p.win + (dxu - p.pad);
      const unsigned tapoff = (unsigned)(pixoff * ldsz);
      u32x4* la = lds + slot * STAGE;
      u32x4* lb = la + BM * 8;
#pragma unroll
      for (int i = 0; i < A_CH; ++i) {
        const unsigned off = (s0 ? offa0[i] : offa1[i]) + tapoff;
        const bool ok = live && (PW ? (msk[i] != 0) : (((msk[i] >> tapbit) & 1) != 0));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(la + (32 * i + 8 * lw) * 8), 16, (int)(ok ? off : kInvalid), soff, 0, 0);
      }
      const int soffw = ku * SZ;
#pragma unroll
      for (int i = 0; i < B_CH; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void_t*)(lb + (32 * i + 8 * lw) * 8), 16, (int)(live ? offb[i] : kInvalid), soffw, 0, 0);
      // advance the wave-uniform K state (branch-free mixed radix, as in gemm_dma_kernel)
      ku += BK;
      const int cu_t = cu + BK;
      const int wc = (cu_t >= ctot) ? 1 : 0;
      const int dx1 = dxu + (chunk_major ? 1 : wc);
      const int wx = (dx1 == p.kw) ? 1 : 0;
      const int dy1 = dyu + wx;
      const int wy = (chunk_major && dy1 == p.kh) ? 1 : 0;
      cu = chunk_major ? cu + (wy ? BK : 0) : (wc ? cu_t - ctot : cu_t);
      dxu = wx ? 0 : dx1;
      dyu = wy ? 0 : dy1;
    };
    auto issue_next = [&](int g) __attribute__((always_inline)) {     // DMA of K-tile step g (tile iti, K-tile ikt)
      const bool live = g < total;
      if (live && ikt == 0) setup_tile(tile0 + iti * G);
      dma_step(g % NSTAGE, live);
      if (++ikt == nk) { ikt = 0; ++iti; }
    };

    // ---- prologue: K-tiles 0 and 1 in flight, K-tile 0 landed ----
    issue_next(0);
    issue_next(1);
    if (WN == 5) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    int kt = 0;
    for (int g = 0; g < total; ++g) {
      STAMP(11);
      __builtin_amdgcn_s_barrier();                      // B(g)
      asm volatile("" ::: "memory");
      STAMP(12);
      issue_next(g + 2);                                 // slot (g+2) % 3 == (g-1) % 3: read in step g-1, free since B(g)
      STAMP(13);
      // K-tile g+1 (issued in step g-1) has landed once all but this step's NPIECE pieces are done (these waves issue
      // nothing else, and exactly NPIECE pieces per step: out-of-range offsets past the end)
      if (WN == 5) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      STAMP(15);
      if (++kt == nk) {
        kt = 0;
        __builtin_amdgcn_s_barrier();                    // Bx
        asm volatile("" ::: "memory");
      }
    }
    __builtin_amdgcn_s_barrier();                        // B(end)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the zero-filled tail DMAs must not outlive the kernel's LDS
    return;
  }

  // ================================ epilogue waves ================================
  // drain the staged tile over the K-tiles of the NEXT tile: activation / GEGLU / residual, 16-byte global stores.  Plain
  // loads and stores (no LDS-DMA in these waves), so the compiler's own waits are exact.
  const int etid = tid - 512;
  T* const out = reinterpret_cast<T*>(p.out);
  const T* const res = reinterpret_cast<const T*>(p.residual);
  T* const ct = reinterpret_cast<T*>(staging);
  auto epilogue_slice = [&](int cbm, int cbn, int it0) __attribute__((always_inline)) {
    if (abl & 4) return;
    for (int j = 0; j < ips; ++j) {
      const int it = it0 + j;
      if (it >= iters) break;
      const int q = etid + 256 * it;
      const int row = q / cpr, ch = q - row * cpr;
      const int m = cbm * BM + row;
      if (m >= p.M) continue;
      if (!geglu) {
        const int n = cbn * BN + ch * 8;
        if (n >= p.N) continue;
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
            float bq[8];
            Elem<bf16_t>::load_chunk(res + (long long)m * p.ldr + n, bq);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += bq[e];
          }
          if (p.act == SASPA_ACT_ADD_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = fmaxf(a[e], 0.0f);
          }
          c4 = __builtin_bit_cast(u32x4, pack8(a));
        }
        *reinterpret_cast<u32x4*>(out + (long long)m * p.ldo + n) = c4;
      } else {
        constexpr int HB = BN / 2;
        const int f = cbn * HB + ch * 8;
        float a[8], gt[8];
        unpack8(*reinterpret_cast<const uint4*>(ct + row * CP + ch * 8), a);
        unpack8(*reinterpret_cast<const uint4*>(ct + row * CP + HB + ch * 8), gt);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = fast_gelu_mul(a[e], gt[e]);
        *reinterpret_cast<uint4*>(out + (long long)m * p.ldo + f) = pack8(a);
      }
    }
  };
  int eti = -1, ekt = 0;                                 // staged tile being drained (index into this workgroup's tiles), its slice
  int ti = 0, kt = 0;
  for (int g = 0; g < total; ++g) {
    __builtin_amdgcn_s_barrier();                        // B(g)
    asm volatile("" ::: "memory");
    if (eti >= 0) {
      const int t = tile0 + eti * G;
      const int cbm = t / nbn;
      epilogue_slice(cbm, t - cbm * nbn, ekt * ips);
    }
    ++ekt;
    if (++kt == nk) {
      kt = 0;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                      // Bx: staging of tile eti fully read -> the MMA waves may park tile ti
      asm volatile("" ::: "memory");
      eti = ti;
      ekt = 0;
      ++ti;
    }
  }
  __builtin_amdgcn_s_barrier();                          // B(end)
  asm volatile("" ::: "memory");
  {
    const int t = tile0 + eti * G;
    const int cbm = t / nbn, cbn = t - cbm * nbn;
    for (int e = 0; e * ips < iters; ++e) epilogue_slice(cbm, cbn, e * ips);
  }
}

}  // namespace

bool saspa_gemm_ws_eligible(const SaspaGemmParams& p) {
  const int ctot = p.c0 + p.c1;
  if (p.dtype != SASPA_BF16 || (long long)p.nb1 * p.nb2 != 1 || p.upsample) return false;
  if ((ctot % 64) != 0 || (p.c1 > 0 && (p.c0 % 64) != 0)) return false;
  if ((p.N % 8) != 0 || (p.ldo % 8) != 0 || (p.residual && (p.ldr % 8) != 0)) return false;
  if (p.act == SASPA_ACT_GEGLU && (p.N % ((p.N % 160) == 0 ? 160 : 128)) != 0) return false;
  if (p.kh * p.kw > 31) return false;
  return true;
}

int saspa_gemm_ws_launch(const SaspaGemmParams& p, hipStream_t s) {
  if (!saspa_gemm_ws_eligible(p)) return SASPA_ERANGE;
  SASPA_DRY_RETURN(SASPA_GEMM_WS, 1);
  const bool n160 = (p.N % 160) == 0;
  const int bn = n160 ? 160 : 128;
  const int tiles = ((p.N + bn - 1) / bn) * ((p.M + 127) / 128);
  const int gx = saspa_balanced_grid(tiles, 256);
  static const int abl = getenv("SASPA_GEMM_ABLATE") ? (atoi(getenv("SASPA_GEMM_ABLATE")) & 15) : 0;   // diagnostics only (8 = stamps)
  const bool pw = p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0;
  const bool gg = p.act == SASPA_ACT_GEGLU;
  const int ta = tiles | (abl << 28);
#define WS_LAUNCH(WN_, PW_, GG_) hipLaunchKernelGGL((gemm_ws_kernel<WN_, PW_, GG_>), dim3(gx), dim3(768), 0, s, p, ta)
  if (n160) {
    if (pw) { if (gg) WS_LAUNCH(5, true, true); else WS_LAUNCH(5, true, false); }
    else { if (gg) WS_LAUNCH(5, false, true); else WS_LAUNCH(5, false, false); }
  } else {
    if (pw) { if (gg) WS_LAUNCH(4, true, true); else WS_LAUNCH(4, true, false); }
    else { if (gg) WS_LAUNCH(4, false, true); else WS_LAUNCH(4, false, false); }
  }
#undef WS_LAUNCH
  SASPA_CHECK_LAUNCH();
  return 0;
}
