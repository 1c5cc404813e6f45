// Ping-pong implicit-GEMM for the long-K bf16 layers (3x3 convs of UNet / ControlNet / VAE, the
// wide projections): one workgroup of 8 waves per CU owns a 256 x (64*FN) output tile.
//
// Why a second kernel: the 4-wave 128x160 kernel of saspa_gemm.hip fetches 1 byte of operand per
// ~71 MFMA flops; at the ~90 GB/s a CU sustains through the LDS-DMA path that caps it near
// 0.9 PFLOP/s whatever the schedule (profiles/r1_*: DMA time and MFMA time add).  A 256x256
// (256x320) tile needs 128 (142) flops per fetched byte, and the two wave groups below keep the
// MFMA pipe busy while the other group issues its LDS reads and DMA.
//
// Structure (cdna_hip_programming.md, "256^2 8-phase template", re-derived for an M-sliced phase
// split and the im2col A operand):
//   * waves 2 (M) x 4 (N); a wave owns 128 x (16*FN) outputs = 8 x FN accumulator fragments
//     (D^T form: weights are the MFMA A operand, so a lane holds 4 consecutive channels).
//   * K-tile = 64 bf16 (128-byte LDS rows, chunk ^= row & 7 swizzle applied on the DMA source
//     side and on the fragment reads); 2 LDS buffers of (256 + BN) rows.
//   * a K-tile is 4 phases; phase P multiplies rows [32P, 32P+32) of each wave's 128 rows against
//     the wave's whole B slice (B fragments are read once per K-tile, in phase 0):
//         ds_read fragments | issue LDS-DMA pieces | counted vmcnt | s_barrier | MFMAs | s_barrier
//   * the wm = 1 group runs one barrier behind the wm = 0 group: while one group is in its MFMA
//     block the other does its reads / DMA issue on the same SIMDs (one wave of each per SIMD).
//   * slots are restaged as soon as they are free: A slice P of tile t is overwritten with slice P
//     of tile t+2 one phase after its last read (each group stages only its own A rows, so its
//     own barrier orders read -> overwrite); the B fragment reads of phase 0 are retired by a
//     counted lgkmcnt BEFORE that phase's first barrier, so B pieces follow from phase 1 on.
//     Two pieces per wave per phase (one A, one B; the 5th B piece of BN = 320 rides with slice 3).
//   * RAW: one counted vmcnt per K-tile (phase 3, ahead of its first barrier, leaving the six
//     pieces of phases 1-3 in flight): everything older -- the whole next tile -- has landed
//     before the phase that first reads it.  Nothing is drained to vmcnt(0) inside the loop.
//   * tail: tiles beyond the K range are "staged" with out-of-range offsets (the DMA writes zeros
//     into slots nobody reads again), which keeps the per-wave vmcnt arithmetic uniform.
//   * round 5: the default loop merges the phases in pairs (LOOP == 2 below: two 40-MFMA intervals per K-tile, the second
//     slice's A fragments read inside the block); the four-phase loop described above is LOOP == 0, kept as an A/B arm.
//   * epilogue (round 4): each wave group finishes its own 128-row half, 64 rows at a time -- bias / time-embedding rows
//     from LDS slots filled by DMA at tile set-up, bf16 staging, LDS-only barrier, coalesced 16-byte store pass with the
//     residual loads of a batch in flight together, GroupNorm statistics accumulated over both passes; split-K writes
//     fp32 slabs.  (profiles/r4_conv_epilogue.txt: what the previous form cost and why.)
#include <cstdlib>

#include "common.h"
#include "gemm_internal.h"

#ifdef SASPA_NO_KORDER
constexpr bool KORDER_ON = false;   // A/B build: the K walk exactly as before ABI v4
#else
constexpr bool KORDER_ON = true;
#endif

#ifndef SASPA_PP_CT_ABL
#define SASPA_PP_CT_ABL 0
#endif

namespace {

typedef bf16_t T;

__device__ __forceinline__ void mma(const u32x4& wf, const u32x4& xf, f32x4& acc) {
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf), __builtin_bit_cast(bf16x8, xf), acc, 0, 0, 0);
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LOOP: 0 = two-barrier ping-pong, four 20-MFMA phases per K-tile; 1 = one barrier per phase, asymmetric programs (round 3,
// slower, A/B arm); 2 = two-barrier ping-pong with TWO 40-MFMA phases per K-tile (round 5: half the pipe hand-offs)
template <int FN, bool PW, bool UP, int LOOP>
__global__ __launch_bounds__(512) void gemm_pp_kernel(const SaspaGemmParams p, const int ntiles_abl, const int npart8) {
  constexpr bool ONEBAR = LOOP == 1;
  // diagnostic ablation (tools/gemm_ablate.py, `make ABLATION=1` only): bits 28..31 of the tile count
  //   1: no MFMA   2: no fragment reads   4: no DMA issue   8: no barriers;  bits 24..26 (SASPA_GEMM_EPI_ABLATE): epilogue
  const int ntiles = ntiles_abl & 0x00ffffff;
#ifdef SASPA_GEMM_ABLATION
  const int abl = (ntiles_abl >> 28) & 15;
  const bool stamp = (ntiles_abl >> 27) & 1;           // block 0 leaves (shader clocks, 100 MHz ticks) of its K loop in out[0..15]
  unsigned long long st0 = 0, sr0 = 0, st1 = 0, sr1 = 0, sr_setup = 0, sr_issue = 0, sr_e0 = 0, sr_e1 = 0;
  const unsigned long long sr_entry = __builtin_amdgcn_s_memrealtime();
  const int eabl = (ntiles_abl >> 24) & 7;             // epilogue ablation: 1 = no global stores, 2 = no epilogue at all
#else
  // compile-time ablation of the K loop (tools/pp_ct_ablate.py builds one library per value; 0 in the shipped library): the
  // same bits as above without a run-time test inside the unrolled loop, which changes what it measures
  constexpr int abl = SASPA_PP_CT_ABL, eabl = 0;
#endif
  constexpr int BM = 256, BN = 64 * FN, BK = 64, SZ = 2;
  constexpr int STAGE = (BM + BN) * 8;                 // u32x4 per LDS buffer
  constexpr int CP = BN + 8;                           // epilogue row pitch (elements)
  constexpr int EPI = 128 * CP * SZ / 16;              // u32x4 for one 128-row half of the output tile
  constexpr int NLDS = 2 * STAGE > EPI ? 2 * STAGE : EPI;
  // bias [BN] and time-embedding rows [half][image of the half][BN] of the tile live behind the staging buffers, staged by
  // LDS-DMA when the tile is set up (a half spans at most 4 images when H*W >= 43; smaller images take a per-fragment
  // global-load path).  The epilogue reads them back from LDS -- one read per column fragment where the half lies in one image
  // (loading them in the epilogue cost one dependent L2 round trip per fragment: 5 - 10 us of the ~20 us epilogue that
  // tools/pp_clock.py measured in round 4).  They are part of the SAME __shared__ object as the staging buffers on purpose:
  // with DMA into a second LDS object in the kernel the compiler put s_waitcnt vmcnt(0) in front of every fragment read of
  // the K loop (+ 10 - 25 % per launch).
  constexpr int NIMG = 4;
  constexpr int ADDV_F = (1 + 2 * NIMG) * BN;          // floats
  __shared__ u32x4 lds[NLDS + ADDV_F / 4];
  float* const addvb = reinterpret_cast<float*>(lds + NLDS);
  float* const addvr = addvb + BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int frow = lane & 15, fg = lane >> 4;
  const int lr = lane >> 3;                            // row inside an 8-row DMA piece
  const int kcs = (lane & 7) ^ lr;                     // logical 16-byte chunk this lane fetches (source-side swizzle)

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

  const int hw = p.hout * p.wout;
  const int ctot = p.c0 + p.c1;
  const int chunk_major = (KORDER_ON && p.korder == SASPA_KORDER_CHUNK) ? 1 : 0;   // wave-uniform
  const int hv = UP ? 2 * p.hin : p.hin, wv = UP ? 2 * p.win : p.win;
  // A descriptors: for the 3x3 / pad 1 window the base is moved back by one row + one pixel so that the
  // tap offset (dy*win + dx) * pitch is a non-negative SCALAR offset (no per-lane arithmetic per tap)
  const int back = (PW || UP) ? 0 : p.win + 1;
  const T* a0s = a0 - (long long)back * p.lda0;
  const T* a1s = p.c1 > 0 ? a1 - (long long)back * p.lda1 : a0s;
  // descriptors are rebuilt from (pointer, record count) scalars whenever the staged tile changes source:
  // zero records = every access out of range -> the DMA writes zeros and touches no memory
  auto make_desc = [](const T* base, bool live) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), (short)0, live ? 0x7fffffff : 0, 0x00020000);
  };
  const rsrc_t rsn = make_desc(w, false);

  // ---- per-lane loader state: one A row per phase slice (4 slices), one weight row ----
  //   PW        va = byte offset of (row m, chunk kcs) in the current source
  //   3x3       va = same for the window's centre pixel; vb = pixel index | (8 halo-tap "outside" bits << 24)
  //   upsample  va = first pixel of the image, vb = 9-bit tap validity, vc = packed top-left corner (generic path)
  // Rows beyond M are aliased to row 0: they are computed and never stored.
  int bm = 0, bn = 0;
  int va[4], vb[PW ? 1 : 4], vc[UP ? 4 : 1];
  unsigned offb0 = 0;                                  // weight row (8*wave + lr) of the tile, chunk kcs; piece i adds 64 rows (scalar)
  int nrows = 0;                                       // valid weight rows of the tile (wave-uniform)
  int ku = 0, cu = 0, dyu = 0, dxu = 0, staged = 0;    // K state of the NEXT tile to stage (wave-uniform)
  int src = -1;                                        // source tensor va[] is currently scaled for
  const int arow0 = wm * 128 + (wave & 3) * 8 + lr;    // A row of slice 0 handled by this lane; + 32 per slice

  auto setup_tile = [&](int t) __attribute__((always_inline)) {
    if (npart8 > 0) {      // N-partitioned order (saspa_gemm.hip: saspa_gemm_npart8)
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
    if (!PW) {
      int m = bm * BM + arow0;
      int b = m / hw;
      int rem = m - b * hw;
      int oy = rem / p.wout;
      int ox = rem - oy * p.wout;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const bool live = m < p.M;
        const int eb = live ? b : 0, ey = live ? oy : 0, ex = live ? ox : 0;
        const int iy0 = ey * p.stride - p.pad, ix0 = ex * p.stride - p.pad;
        if (UP) {
          va[s] = eb * p.hin * p.win;
          vc[UP ? s : 0] = ((iy0 + 1) << 16) | (ix0 + 1);
          int mask = 0;
          for (int ty = 0; ty < 3; ++ty)
            for (int tx = 0; tx < 3; ++tx)
              if ((unsigned)(iy0 + ty) < (unsigned)hv && (unsigned)(ix0 + tx) < (unsigned)wv) mask |= 1 << (ty * 3 + tx);
          vb[PW ? 0 : s] = mask;
        } else {
          // "outside" bit of the 8 non-centre taps (the centre tap of a live row is always inside)
          int outside = 0;
#pragma unroll
          for (int tb = 0; tb < 9; ++tb) {
            if (tb == 4) continue;
            const int ty = tb / 3, tx = tb - ty * 3;
            if (!((unsigned)(iy0 + ty) < (unsigned)hv && (unsigned)(ix0 + tx) < (unsigned)wv)) outside |= 1 << (tb < 4 ? tb : tb - 1);
          }
          vb[PW ? 0 : s] = (eb * p.hin * p.win + (ey * p.stride) * p.win + ex * p.stride) | (outside << 24);
        }
        m += 32;
        ox += 32;
        while (ox >= p.wout) { ox -= p.wout; ++oy; }
        while (oy >= p.hout) { oy -= p.hout; ++b; }
      }
    }
    offb0 = (unsigned)((bn * BN + wave * 8 + lr) * p.ldw * SZ + kcs * 16);
    nrows = p.N - bn * BN;                             // N % 8 == 0: validity is uniform over an 8-row piece
    ku = kt0 * BK;
    if (chunk_major) {
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
    staged = 0;
    src = -1;
  };

  // wave-uniform description of the tile being staged
  bool sv = false;
  rsrc_t rs = rsn, rswv = rsn;
  int ldsz = 0, soff = 0, soffw = 0, sdy = 0, sdx = 0, bsh = 0, tapbit = 0;
  unsigned amask = 0;
  auto begin_stage = [&]() __attribute__((always_inline)) {
    sv = staged < nk;
    const bool s0 = cu < p.c0;
    ldsz = (s0 ? p.lda0 : p.lda1) * SZ;
    rs = make_desc(s0 ? a0s : a1s, sv);
    rswv = make_desc(w, sv);
    soff = (s0 ? cu : cu - p.c0) * SZ;
    soffw = ku * SZ;
    if (!UP) {
      if (!ONEBAR && src != (s0 ? 0 : 1)) {
        // (re)scale the per-lane offsets for this source's row pitch: once per tap and source, not per tile.  (The
        // one-barrier 3x3 loop has no register to spare for va[]: it re-derives the offset per piece, 2 VALU.)
        src = s0 ? 0 : 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          int px;
          if (PW) {
            const int m = bm * BM + arow0 + 32 * s;
            px = m < p.M ? m : 0;
          } else {
            px = vb[PW ? 0 : s] & 0xffffff;
          }
          va[s] = (int)(__umul24((unsigned)px, (unsigned)ldsz) + (unsigned)(kcs * 16));
        }
      }
      if (!PW) {
        const int tb = dyu * 3 + dxu;
        soff += (dyu * p.win + dxu) * ldsz;
        bsh = 7 - (tb < 4 ? tb : tb - 1);
        amask = tb == 4 ? 0u : 0x80000000u;
      }
    } else {
      tapbit = dyu * 3 + dxu;
      sdy = dyu;
      sdx = dxu;
    }
    // advance to the following tile
    ++staged;
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
  // LDS element (u32x4) index of the READ buffer (cur) and of the other one; swapped every K-tile
  int cur = 0, oth = STAGE;
  // fragment read offsets of this lane inside a buffer (kk = 0 / 1: swizzled chunk differs in bit 2)
  const int swz0 = fg ^ (frow & 7);
  int ra0 = (wm * 128 + frow) * 8 + swz0, ra1 = ra0 ^ 4;
  int rb0 = (BM + wn * (16 * FN) + frow) * 8 + swz0, rb1 = rb0 ^ 4;
  const int dma_a = (wm * 128 + (wave & 3) * 8) * 8;   // + 32 rows per slice
  const int dma_b = (BM + wave * 8) * 8;               // + 64 rows per piece

  auto stage_a = [&](const int s, const int buf) __attribute__((always_inline)) {
    unsigned off;
    if (UP) {
      const int iy = ((vc[UP ? s : 0] >> 16) - 1 + sdy) >> 1, ix = ((vc[UP ? s : 0] & 0xffff) - 1 + sdx) >> 1;
      const int px = va[s] + iy * p.win + ix;
      const bool ok = ((vb[PW ? 0 : s] >> tapbit) & 1) != 0;
      off = ok ? __umul24((unsigned)px, (unsigned)ldsz) + (unsigned)(kcs * 16) : kInvalid;
    } else if (PW) {
      if (ONEBAR) {                                    // no register to spare for va[]: row -> offset per piece (3 VALU)
        const int m = bm * BM + arow0 + 32 * s;
        off = __umul24((unsigned)(m < p.M ? m : 0), (unsigned)ldsz) + (unsigned)(kcs * 16);
      } else {
        off = (unsigned)va[s];
      }
    } else {
      // bit 31 set (out of range -> zeros) iff this tap falls outside the image for this row
      const unsigned base = ONEBAR ? __umul24((unsigned)vb[PW ? 0 : s] & 0xffffffu, (unsigned)ldsz) + (unsigned)(kcs * 16) : (unsigned)va[s];
      off = (((unsigned)vb[PW ? 0 : s] << bsh) & amask) | base;
    }
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(lds + buf + dma_a + s * 32 * 8), 16, (int)off, soff, 0, 0);
  };
  auto stage_b = [&](const int i, const int buf) __attribute__((always_inline)) {
    // rows beyond N (wave-uniform per 8-row piece): a scalar offset beyond num_records -> zeros
    const bool ok = (wave + 8 * i) * 8 < nrows;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rswv, (lds_void_t*)(lds + buf + dma_b + i * 64 * 8), 16, (int)(ok ? offb0 : kInvalid),
                                             soffw + i * 64 * p.ldw * SZ, 0, 0);
  };
  // B pieces that travel with A slice s: piece s for s < 3, pieces 3 .. FN-1 with slice 3
  auto stage_slice = [&](const int s, const int buf) __attribute__((always_inline)) {
    stage_a(s, buf);
    if (s < 3) {
      stage_b(s, buf);
    } else {
#pragma unroll
      for (int i = 3; i < FN; ++i) stage_b(i, buf);
    }
  };

  f32x4 acc[8][FN];
  u32x4 wb[FN][2], xa[2][2];
  int dbgv[4] = {0, 0, 0, 0};                          // ablation bit 32 only
#if defined(SASPA_GEMM_ABLATION) || SASPA_PP_CT_ABL
#pragma unroll
  for (int j = 0; j < FN; ++j) wb[j][0] = wb[j][1] = u32x4{(unsigned)j, 0u, 0x3f803f80u, 0x3f803f80u};
#pragma unroll
  for (int i = 0; i < 2; ++i) xa[i][0] = xa[i][1] = u32x4{(unsigned)i, 1u, 0x3f803f80u, 0x3f803f80u};
#endif

  // one phase of the tile in `cur`: fragment reads of slice P (+ all B fragments in phase 0), DMA issue of
  // slice SS into buffer sbuf, then barrier | MFMAs | barrier
  auto phase = [&](const int P, const int SS, const int sbuf, const bool sync) __attribute__((always_inline)) {
    if (!(abl & 2)) {
      if (P == 0) {
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          wb[j][0] = lds[rb0 + j * 16 * 8];
          wb[j][1] = lds[rb1 + j * 16 * 8];
        }
        __builtin_amdgcn_sched_barrier(0);             // B reads are issued (and counted) before the A reads
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        xa[i][0] = lds[ra0 + (P * 32 + i * 16) * 8];
        xa[i][1] = lds[ra1 + (P * 32 + i * 16) * 8];
      }
    }
    if (!(abl & 4)) stage_slice(SS, sbuf);
    // phase 3: everything but this K-tile's slices 0..2 (six pieces) has landed, i.e. the whole next tile
    if (sync && P == 3) wait_vm<6>();
    // phase 0: the B fragment reads (issued first) are complete before this wave passes the barrier, so the
    // other group may restage the B slot in ITS next phase
    if (sync && P == 0) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
    if (sync && !(abl & 8)) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
    if (abl & 1) {
#pragma unroll
      for (int i = 0; i < 2; ++i) { asm volatile("" ::"v"(xa[i][0])); asm volatile("" ::"v"(xa[i][1])); }
#pragma unroll
      for (int j = 0; j < FN; ++j) { asm volatile("" ::"v"(wb[j][0])); asm volatile("" ::"v"(wb[j][1])); }
    } else {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j) mma(wb[j][kk], xa[i][kk], acc[P * 2 + i][j]);
    }
    __builtin_amdgcn_s_setprio(0);
    if (sync && !(abl & 8)) __builtin_amdgcn_s_barrier();
  };

  setup_tile(tile);
#ifdef SASPA_GEMM_ABLATION
  if (stamp) sr_setup = __builtin_amdgcn_s_memrealtime();
#endif
  for (;;) {
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (gridDim.y == 1) {
      // bias and time-embedding rows of this tile -> LDS by DMA (no register, no wait: they are the oldest vector-memory
      // operations of the tile, so the first counted vmcnt of the K loop covers them).  256-byte pieces, one wave each:
      // piece row 0 = bias, row 1 + half * NIMG + il = row vector of image (first image of the half) + il.  A descriptor
      // with zero records turns a piece into zeros (no bias / no row vector / image beyond the batch / column beyond N).
      // (The previous tile's epilogue ended with a barrier after its last read of these slots.)
      constexpr int PCS = BN / 64;
      const int nimg = p.rowvec ? NIMG : 1;
      for (int pc = wave; pc < PCS * (1 + 2 * NIMG); pc += 8) {
        const int rowi = pc / PCS, part = pc - rowi * PCS;
        const int voff = (bn * BN + part * 64 + lane) * 4;
        if (rowi == 0) {
          const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), (short)0, p.bias ? p.N * 4 : 0, 0x00020000);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_t*)(addvb + part * 64), 4, voff, 0, 0, 0);
        } else {
          const int hi = rowi - 1, h = hi / NIMG, il = hi - h * NIMG;
          const int img = min(bm * BM + h * 128, p.M - 1) / hw + il;
          const bool live = p.rowvec && il < nimg && (long long)img * hw < p.M;
          const rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.rowvec) + (live ? (long long)img * p.ldrv : 0), (short)0,
                                                              live ? p.N * 4 : 0, 0x00020000);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_void_t*)(addvr + hi * BN + part * 64), 4, voff, 0, 0, 0);
        }
      }
    }

    if (nk > 0) {
      if (ONEBAR) {
        // ---- ONE barrier per phase, asymmetric programs (round 3) ----
        // The ping-pong loop below spends two s_barriers per phase to hand the matrix pipe from one wave group to the
        // other (8 per K-tile: 52-57 % MFMA busy against 93 % with the barriers ablated).  Here the two
        // groups run DIFFERENT programs between the same single barrier per phase:
        //     wm = 0:  barrier | fragment reads of slice P | DMA issue | MFMAs of slice P
        //     wm = 1:  barrier | MFMAs of slice P (fragments read in the previous phase) | reads of slice P+1 | DMA issue
        // so right after a barrier the wm = 1 wave of every SIMD owns the matrix pipe while its wm = 0 partner waits for
        // LDS, and they swap roles half a phase later without any hand-off.  Hazards (slots and DMA schedule exactly as
        // in the ping-pong loop): a slot is restaged one phase after its slice's phase -- wm = 0 read it in that phase
        // (retired before its MFMAs, i.e. before the next barrier), wm = 1 a phase earlier; tile t+1 must be complete
        // before wm = 1 reads its slice 0 / B fragments in phase 3 of tile t: counted vmcnt fences at the END of a phase,
        // in front of the barrier after which the data is read (table below).
        // DMA schedule of this loop (round 3): ALL weight pieces of tile t+2 leave in phases 1 - 2 of tile t (their slots
        // are free after phase 0), the four activation slices in phases 1, 2, 3 and 0 -- every piece is 4 - 6 phases in
        // flight, against 2 - 3 for the pieces that completed a tile in the first version of this loop (which lost 13 -
        // 30 % to the ping-pong loop: the loop was waiting for DMA, profiles/r3_pp_ab_v1.txt).
        constexpr int NB1 = (FN + 1) / 2;              // weight pieces issued in phase 1; the rest in phase 2
        auto stage_p1 = [&](const int buf) __attribute__((always_inline)) {
          stage_a(0, buf);
#pragma unroll
          for (int i = 0; i < NB1; ++i) stage_b(i, buf);
        };
        auto stage_p2 = [&](const int buf) __attribute__((always_inline)) {
          stage_a(1, buf);
#pragma unroll
          for (int i = NB1; i < FN; ++i) stage_b(i, buf);
        };
        begin_stage();
#pragma unroll
        for (int s = 0; s < 4; ++s) stage_slice(s, cur);
        begin_stage();
        stage_p1(oth);
        stage_p2(oth);
        stage_a(2, oth);
        wait_vm<3 + FN>();                             // tile 0 has landed; tile 1 (all but slice 3) stays in flight
        __builtin_amdgcn_s_barrier();
#ifdef SASPA_GEMM_ABLATION
        if (stamp) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }
#endif
        auto read_b = [&]() __attribute__((always_inline)) {
          const int o1 = rb0 ^ 4;
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            wb[j][0] = lds[rb0 + j * 16 * 8];
            wb[j][1] = lds[o1 + j * 16 * 8];
          }
        };
        auto read_a = [&](const int P) __attribute__((always_inline)) {
          const int o1 = ra0 ^ 4;
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            xa[i][0] = lds[ra0 + (P * 32 + i * 16) * 8];
            xa[i][1] = lds[o1 + (P * 32 + i * 16) * 8];
          }
        };
        auto mfma_block = [&](const int P) __attribute__((always_inline)) {
          if (abl & 1) {
#pragma unroll
            for (int i = 0; i < 2; ++i) { asm volatile("" ::"v"(xa[i][0])); asm volatile("" ::"v"(xa[i][1])); }
#pragma unroll
            for (int j = 0; j < FN; ++j) { asm volatile("" ::"v"(wb[j][0])); asm volatile("" ::"v"(wb[j][1])); }
            return;
          }
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
              for (int j = 0; j < FN; ++j) mma(wb[j][kk], xa[i][kk], acc[P * 2 + i][j]);
          __builtin_amdgcn_s_setprio(0);
        };
        auto swap_bufs = [&]() __attribute__((always_inline)) {
          const int d = oth - cur;
          ra0 += d; rb0 += d;
          cur += d;
          oth -= d;
        };
        // one loop body for both groups (wave-uniform scalar branches): two copies of the loop spilled inside it
        // `wm` (an SGPR integer) is tested at every site: a bool carried across the loop was kept in a VGPR and spilled
        // wm = 1 ("lead"): MFMAs first, reads of the NEXT slice afterwards
        if (wm != 0) { read_b(); read_a(0); }
        // counted waits (issue order is program order; c0..c3 = pieces a wave issues in phases 0..3 = 1, 1 + NB1,
        // 1 + FN - NB1, 1): a fence sits at the END of a phase, in front of the barrier after which the wm = 1 group reads
        //   end of phase 2: B + slices 0, 1 of tile t+1 (issued up to phase 2 of tile t-1) -> all but c3+c0+c1+c2 = 4 + FN
        //   end of phase 0: slice 2 of tile t (issued in phase 3 of tile t-2)             -> all but 2c0+c1+c2+c3 = 5 + FN
        //   end of phase 1: slice 3 of tile t (issued in phase 0 of tile t-1)             -> all but c0+2c1+c2+c3 = 5 + FN + NB1
        // ablation hooks (diagnostics build only; plain statements otherwise)
#ifdef SASPA_GEMM_ABLATION
#define BAR() do { if (!(abl & 8)) __builtin_amdgcn_s_barrier(); } while (0)
#define RD(...) { if (!(abl & 2)) { __VA_ARGS__ } }
#define ST(...) { if (!(abl & 4)) { __VA_ARGS__ } }
#else
#define BAR() __builtin_amdgcn_s_barrier()
#define RD(...) { __VA_ARGS__ }
#define ST(...) { __VA_ARGS__ }
#endif
        for (int t = 0; t < nk; ++t) {
          // ---- phase 0 ----
          BAR();
          if (wm == 0) { RD(read_b(); read_a(0);) ST(stage_a(3, oth);) }
          mfma_block(0);
          __builtin_amdgcn_sched_barrier(0);
          if (wm != 0) { RD(read_a(1);) ST(stage_a(3, oth);) }
          wait_vm<5 + FN>();
          begin_stage();
          // ---- phase 1 ----
          BAR();
          if (wm == 0) { RD(read_a(1);) ST(stage_p1(cur);) }
          mfma_block(1);
          __builtin_amdgcn_sched_barrier(0);
          if (wm != 0) { RD(read_a(2);) ST(stage_p1(cur);) }
          wait_vm<5 + FN + NB1>();
          // ---- phase 2 ----
          BAR();
          if (wm == 0) { RD(read_a(2);) ST(stage_p2(cur);) }
          mfma_block(2);
          __builtin_amdgcn_sched_barrier(0);
          if (wm != 0) { RD(read_a(3);) ST(stage_p2(cur);) }
          wait_vm<4 + FN>();
          // ---- phase 3 ----
          BAR();
          if (wm == 0) { RD(read_a(3);) ST(stage_a(2, cur);) }
          mfma_block(3);
          __builtin_amdgcn_sched_barrier(0);
          if (wm != 0) { ST(stage_a(2, cur);) }
          swap_bufs();
          if (wm != 0) { RD(read_b(); read_a(0);) }        // tile t+1 (zeros past the K range): complete since this phase's barrier
        }
#undef BAR
#undef RD
#undef ST
#ifdef SASPA_GEMM_ABLATION
        if (stamp) { st1 = __builtin_amdgcn_s_memtime(); sr1 = __builtin_amdgcn_s_memrealtime(); }
#endif
      } else if (LOOP == 2) {
        // ---- long ping-pong (round 5): the loop below with its phases merged in pairs ----
        // Fewer, longer intervals: a K-tile is cut into TWO intervals of 40 MFMAs (Q0: slices 0, 1; Q1: slices 2, 3) instead
        // of four of 20.  (Built on the rounds 3 - 4 reading that a hand-off costs ~100 cycles; measured properly this round it is
        // ~12 -- tools/micro/barrier_bench.hip -- and what the merge really buys is fewer per-interval waits on the partner's DMA
        // issue: +2 ... +8 % per launch, profiles/r5_pp_long_ab.txt.)  No more fragment registers than
        // before: the MFMAs of a slice run row-fragment-major (i outer), and the A fragments of the interval's SECOND slice are
        // read inside the block into the registers its first slice has just released (10 MFMAs = 160 cycles of cover each).
        // Per accumulator the K order is unchanged (kk = 0, then 1): bit-identical to the other loops.
        // DMA schedule: Q1 of tile t sends A slices 0, 1 and ALL weight pieces of tile t+2 into `cur` (the weight slot is free
        // once both groups have read tile t's B fragments: retired before Q0's first barrier; slices 0, 1 were read in Q0),
        // Q0 of tile t sends A slices 2, 3 of tile t+1 into `oth` (read in Q1 of tile t-1).  Every piece is two intervals in
        // flight: one counted wait per interval, vmcnt(4 + FN), ahead of its first barrier, covers what the NEXT interval reads.
        auto stage_q1 = [&](const int buf) __attribute__((always_inline)) {
          stage_a(0, buf);
          stage_a(1, buf);
#pragma unroll
          for (int i = 0; i < FN; ++i) stage_b(i, buf);
        };
        auto read_a2 = [&](const int i, const int S) __attribute__((always_inline)) {
          xa[i][0] = lds[ra0 + (S * 32 + i * 16) * 8];
          xa[i][1] = lds[ra1 + (S * 32 + i * 16) * 8];
        };
        auto mma_row = [&](const int i, const int R) __attribute__((always_inline)) {
          if (abl & 1) {
            asm volatile("" ::"v"(xa[i][0])); asm volatile("" ::"v"(xa[i][1]));
#pragma unroll
            for (int j = 0; j < FN; ++j) { asm volatile("" ::"v"(wb[j][0])); asm volatile("" ::"v"(wb[j][1])); }
            return;
          }
#pragma unroll
          for (int kk = 0; kk < 2; ++kk)
#pragma unroll
            for (int j = 0; j < FN; ++j) mma(wb[j][kk], xa[i][kk], acc[R][j]);
        };
        // ablation bit 32 (compile time, timing library only): per wave, the shader cycles spent between the END of an MFMA
        // block and the release of the barrier behind it, summed per interval kind -- i.e. how long the matrix pipe's owner
        // waits for its partner's reads / DMA issue to finish (plus the barrier's own latency); waves 0 and 4 of workgroup 0
        // leave {sum Q0, sum Q1, loop cycles, intervals} in the first 32 bytes of the output
        int wsum[2] = {0, 0};
        long long tloop0 = 0;
        if (abl & 32) tloop0 = __builtin_amdgcn_s_memtime();
        auto qphase = [&](const int Q) __attribute__((always_inline)) {
          if (!(abl & 8)) __builtin_amdgcn_s_barrier();
          if (!(abl & 64)) __builtin_amdgcn_s_setprio(1);          // (bit 64: A/B of the block's priority; no measurable effect)
          mma_row(0, 4 * Q + 0);
          __builtin_amdgcn_sched_barrier(0);
          if (!(abl & 2)) read_a2(0, 2 * Q + 1);
          __builtin_amdgcn_sched_barrier(0);
          mma_row(1, 4 * Q + 1);
          __builtin_amdgcn_sched_barrier(0);
          if (!(abl & 2)) read_a2(1, 2 * Q + 1);
          __builtin_amdgcn_sched_barrier(0);
          mma_row(0, 4 * Q + 2);
          mma_row(1, 4 * Q + 3);
          if (!(abl & 64)) __builtin_amdgcn_s_setprio(0);
          long long tb = 0;
          if (abl & 32) { __builtin_amdgcn_sched_barrier(0); tb = __builtin_amdgcn_s_memtime(); }
          if (!(abl & 8)) __builtin_amdgcn_s_barrier();
          if (abl & 32) {
            const long long te = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            wsum[Q] += (int)(te - tb);
            __builtin_amdgcn_sched_barrier(0);
          }
        };
        begin_stage();
#pragma unroll
        for (int s = 0; s < 4; ++s) stage_slice(s, cur);
        begin_stage();
        stage_q1(oth);
        wait_vm<2 + FN>();                             // tile 0 has landed; tile 1 (slices 0, 1 + weights) stays in flight
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();     // the wm = 1 group runs one barrier behind
#ifdef SASPA_GEMM_ABLATION
        if (stamp) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }
#endif
        for (int t = 0; t < nk; ++t) {
          // ---- Q0: B fragments + slice 0 of tile t; slices 2, 3 of tile t+1 -> oth ----
          if (!(abl & 2)) {
#pragma unroll
            for (int j = 0; j < FN; ++j) {
              wb[j][0] = lds[rb0 + j * 16 * 8];
              wb[j][1] = lds[rb1 + j * 16 * 8];
            }
            __builtin_amdgcn_sched_barrier(0);         // B reads are issued (and counted) before the A reads
            read_a2(0, 0);
            read_a2(1, 0);
          }
          if (!(abl & 4)) { stage_a(2, oth); stage_a(3, oth); }
          wait_vm<4 + FN>();                           // slices 2, 3 of tile t (Q1 reads them)
          asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");   // B reads retired: the other group may restage the slot in its Q1
          qphase(0);
          begin_stage();
          // ---- Q1: slice 2 of tile t; slices 0, 1 + weights of tile t+2 -> cur ----
          if (!(abl & 2)) { read_a2(0, 2); read_a2(1, 2); }
          if (!(abl & 4)) stage_q1(cur);
          wait_vm<4 + FN>();                           // slices 0, 1 + weights of tile t+1 (the next Q0 reads them)
          qphase(1);
          const int d = oth - cur;
          ra0 += d; ra1 += d; rb0 += d; rb1 += d;
          cur += d;
          oth -= d;
        }
#ifdef SASPA_GEMM_ABLATION
        if (stamp) { st1 = __builtin_amdgcn_s_memtime(); sr1 = __builtin_amdgcn_s_memrealtime(); }
#endif
        if (wm == 0) __builtin_amdgcn_s_barrier();     // barrier counts of the two groups match again
        if ((abl & 32) && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (wave & 3) == 0 && lane == 0) {
          dbgv[0] = wsum[0]; dbgv[1] = wsum[1]; dbgv[2] = (int)(__builtin_amdgcn_s_memtime() - tloop0); dbgv[3] = 2 * nk;
        }
      } else {
      // ---- prologue: tile 0 complete, slices 0..2 of tile 1 in flight ----
      begin_stage();
#pragma unroll
      for (int s = 0; s < 4; ++s) stage_slice(s, cur);
      begin_stage();
#pragma unroll
      for (int s = 0; s < 3; ++s) stage_slice(s, oth);
#ifdef SASPA_GEMM_ABLATION
      if (stamp) sr_issue = __builtin_amdgcn_s_memrealtime();
#endif
      wait_vm<6>();
      __builtin_amdgcn_s_barrier();
      if (wm == 1) __builtin_amdgcn_s_barrier();       // the wm = 1 group runs one barrier behind
#ifdef SASPA_GEMM_ABLATION
      if (stamp) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }
#endif
      for (int t = 0; t < nk; ++t) {
        // tile t is multiplied out of `cur`.  Phase 0 completes tile t+1 (slice 3 -> oth); phases 1..3 put
        // slices 0..2 of tile t+2 into the slots of `cur` that phases 0..2 have just released.
        phase(0, 3, oth, true);
        begin_stage();
        phase(1, 0, cur, true);
        phase(2, 1, cur, true);
        phase(3, 2, cur, true);
        const int d = oth - cur;
        ra0 += d; ra1 += d; rb0 += d; rb1 += d;
        cur += d;
        oth -= d;
      }
#ifdef SASPA_GEMM_ABLATION
      if (stamp) { st1 = __builtin_amdgcn_s_memtime(); sr1 = __builtin_amdgcn_s_memrealtime(); }
#endif
      if (wm == 0) __builtin_amdgcn_s_barrier();       // barrier counts of the two groups match again
      }
      wait_vm<0>();                                    // tail DMAs (zeros) must not land in the epilogue's LDS
      if (cur != 0) {                                  // next output tile starts from buffer 0 again
        ra0 -= STAGE; ra1 -= STAGE; rb0 -= STAGE; rb1 -= STAGE;
        cur = 0;
        oth = STAGE;
      }
    }
    __syncthreads();

    const int cbm = bm, cbn = bn;
    const int next = tile + G;
    const bool has_next = next < ntiles;

    // ---- epilogue ----
    // lane geometry re-derived from the thread id behind an optimisation barrier: the copies computed before the K loop would
    // otherwise be kept alive across it -- the loop has no register to spare, so they were spilled and came back as ~50
    // one-at-a-time scratch reloads (each followed by s_waitcnt vmcnt(0)) in front of the epilogue's LDS writes
    int etid = threadIdx.x;
    asm volatile("" : "+v"(etid));
    const int elane = etid & 63;
    const int ewave = __builtin_amdgcn_readfirstlane(etid >> 6);
    const int ewm = ewave >> 2, ewn = ewave & 3, efrow = elane & 15, efg = elane >> 4;
    T* out = reinterpret_cast<T*>(p.out) + ooff;
    const T* res = p.residual ? reinterpret_cast<const T*>(p.residual) + ooff : nullptr;
#ifdef SASPA_GEMM_ABLATION
    if (stamp) sr_e0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (eabl == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else if (gridDim.y > 1) {
      float* ws = p.workspace + (long long)blockIdx.y * p.M * p.N;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = cbm * BM + ewm * 128 + i * 16 + efrow;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int n = cbn * BN + ewn * (16 * FN) + j * 16 + efg * 4;
          if (n >= p.N) continue;
          *reinterpret_cast<float4*>(ws + (long long)m * p.N + n) =
              make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
      }
    } else {
      // Two wave groups (ewm = 0 / 1), each finishing ITS 128-row half of the tile, 64 rows at a time:
      //     stage 64 rows as bf16 in the group's LDS block | barrier | coalesced store pass by the group's 256 threads | barrier
      // All 8 waves run the same program.  (Until round 4 the halves took turns -- `if (wm == h)` staged 128 rows, then all 8
      // waves stored them -- with the other half's 160 accumulator registers waiting: the store pass had no registers for a
      // batch of residual loads, its spill reloads waited on vmcnt(0) = on the stores just issued, and bias / row vector came
      // from global memory once per fragment: tools/pp_clock.py measured 22 - 40 us for 160 KB per workgroup.)
      // The barriers order LDS accesses only (lds_barrier: lgkmcnt(0) + s_barrier).  __syncthreads() also waits for vmcnt(0),
      // i.e. for every global store of the pass to be acknowledged by L2: each pass then drained (4 - 6 us for the chip's
      // 21 MB) before the next one could stage; now the stores drain under the following pass and the kernel's tail.
      T* ct = reinterpret_cast<T*>(lds);
      T* ctg = ct + ewm * (64 * CP);                     // this group's staging block
      const int gtid = etid & 255;
      const int m0 = cbm * BM + ewm * 128;
      const int img0 = min(m0, p.M - 1) / hw;
      const int mnext = (img0 + 1) * hw;                 // first row of the next image
      // bias + time-embedding row from addv: one LDS read per column fragment where the half lies in one image, one per
      // accumulator fragment where it spans several (H*W < 128) -- wave-uniform, no global load either way
      const bool multi = p.rowvec && m0 + 128 > mnext && mnext < p.M;
      const float* avb = addvb + ewn * (16 * FN) + efg * 4;
      const float* av = addvr + ewm * NIMG * BN + ewn * (16 * FN) + efg * 4;
      float4 add[FN];
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        add[j] = *reinterpret_cast<const float4*>(avb + j * 16);
        const float4 r4 = *reinterpret_cast<const float4*>(av + j * 16);
        add[j].x += r4.x; add[j].y += r4.y; add[j].z += r4.z; add[j].w += r4.w;   // bias + row vector, then acc + that
      }
      // GroupNorm statistics (SaspaGemmParams.gn_stats: one (sum, sum of squares) per 128-row block and unit): thread
      // (unit gu, row group grg) of the group keeps its partial sums over both passes
      const int gunit = p.gn_stats ? p.gn_unit : BN;
      const int nunits = BN / gunit;
      const int rgs = 256 / nunits;
      const int gu = gtid % nunits, grg = gtid / nunits;
      float gsm = 0.f, gsq = 0.f;
      constexpr int CPR = BN / 8;
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
        if (!multi) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < FN; ++j) {
              const f32x4& c = acc[qt * 4 + i][j];
              float v[4] = {c[0] + add[j].x, c[1] + add[j].y, c[2] + add[j].z, c[3] + add[j].w};
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] *= p.alpha;
              Elem<T>::store4(ctg + (i * 16 + efrow) * CP + ewn * (16 * FN) + j * 16 + efg * 4, v);
            }
          }
        } else if (hw >= 43) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = m0 + (qt * 4 + i) * 16 + efrow;
            const int il = (m >= mnext ? 1 : 0) + (m >= mnext + hw ? 1 : 0) + (m >= mnext + 2 * hw ? 1 : 0);   // image of this row
            const float* avr = av + il * BN;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
              float4 ad = *reinterpret_cast<const float4*>(avb + j * 16);
              const float4 r4 = *reinterpret_cast<const float4*>(avr + j * 16);
              ad.x += r4.x; ad.y += r4.y; ad.z += r4.z; ad.w += r4.w;
              const f32x4& c = acc[qt * 4 + i][j];
              float v[4] = {c[0] + ad.x, c[1] + ad.y, c[2] + ad.z, c[3] + ad.w};
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] *= p.alpha;
              Elem<T>::store4(ctg + (i * 16 + efrow) * CP + ewn * (16 * FN) + j * 16 + efg * 4, v);
            }
          }
        } else {
          // images of fewer than 43 pixels: a half spans more than the NIMG staged rows -- the row vector comes from global
          // memory per accumulator fragment (tiny problems only; the dispatcher never sends a production shape here)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int m = m0 + (qt * 4 + i) * 16 + efrow;
            const float* rvd = p.rowvec + (long long)(min(m, p.M - 1) / hw) * p.ldrv;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
              const int n = cbn * BN + ewn * (16 * FN) + j * 16 + efg * 4;
              float4 ad = *reinterpret_cast<const float4*>(avb + j * 16);
              if (n < p.N) {
                const float4 r4 = *reinterpret_cast<const float4*>(rvd + n);
                ad.x += r4.x; ad.y += r4.y; ad.z += r4.z; ad.w += r4.w;
              }
              const f32x4& c = acc[qt * 4 + i][j];
              float v[4] = {c[0] + ad.x, c[1] + ad.y, c[2] + ad.z, c[3] + ad.w};
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] *= p.alpha;
              Elem<T>::store4(ctg + (i * 16 + efrow) * CP + ewn * (16 * FN) + j * 16 + efg * 4, v);
            }
          }
        }
        lds_barrier();
        const int mq = m0 + qt * 64;                     // first output row of the staged block
        if (p.act == SASPA_ACT_GEGLU) {
          // weights are packed per 160 columns (weights.pack_geglu): [80 values | their 80 gates]; a 320-wide tile holds
          // two such groups -> 160 output features per tile row
          if constexpr (FN == 5) {
            constexpr int CPG = 20;
            for (int q = gtid; q < 64 * CPG; q += 256) {
              const int row = q / CPG, ch = q - row * CPG;
              const int m = mq + row;
              if (m >= p.M) continue;
              const int sub = ch / 10, c10 = ch - sub * 10;
              float a[8], g[8];
              unpack8(*reinterpret_cast<const uint4*>(ctg + row * CP + sub * 160 + c10 * 8), a);
              unpack8(*reinterpret_cast<const uint4*>(ctg + row * CP + sub * 160 + 80 + c10 * 8), g);
#pragma unroll
              for (int e = 0; e < 8; ++e) a[e] = fast_gelu_mul(a[e], g[e]);
              *reinterpret_cast<uint4*>(out + (long long)m * p.ldo + cbn * 160 + ch * 8) = pack8(a);
            }
          }
        } else {
          // one 16-byte chunk: activation / residual on the staged bf16 values, then the coalesced store
          auto finish = [&](u32x4 c4, const uint4 r4, const int row, const int ch, const int m, const int n) __attribute__((always_inline)) {
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
                unpack8(r4, b);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] += b[e];
              }
              if (p.act == SASPA_ACT_ADD_RELU) {
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] = fmaxf(a[e], 0.0f);
              }
              c4 = __builtin_bit_cast(u32x4, pack8(a));
              if (p.gn_stats) *reinterpret_cast<u32x4*>(ctg + row * CP + ch * 8) = c4;   // the statistics read the STORED values
            }
            if (eabl != 1) *reinterpret_cast<u32x4*>(out + (long long)m * p.ldo + n) = c4;
          };
          constexpr int NIT = 64 * CPR / 256;            // chunks per thread and pass: 10 (BN = 320) / 8 (BN = 256)
          static_assert(NIT * 256 == 64 * CPR && NIT % 2 == 0, "store pass: whole rounds of 256 chunks");
          if (mq + 64 <= p.M && cbn * BN + BN <= p.N) {
            // whole block inside the output (wave-uniform): two batches of NIT / 2 chunks, every residual load of a batch in
            // flight before the first is used (the rolled loop below waits for one L2 round trip per chunk)
            constexpr int NB = NIT / 2;
#pragma unroll
            for (int b0 = 0; b0 < NIT; b0 += NB) {
              uint4 r4[NB];
              int row[NB], ch[NB];
#pragma unroll
              for (int k = 0; k < NB; ++k) {
                const int q = gtid + (b0 + k) * 256;
                row[k] = q / CPR;
                ch[k] = q - row[k] * CPR;
                r4[k] = make_uint4(0u, 0u, 0u, 0u);
              }
              if (res) {
#pragma unroll
                for (int k = 0; k < NB; ++k)
                  r4[k] = *reinterpret_cast<const uint4*>(res + (long long)(mq + row[k]) * p.ldr + cbn * BN + ch[k] * 8);
              }
#pragma unroll
              for (int k = 0; k < NB; ++k) {
                const u32x4 c4 = *reinterpret_cast<const u32x4*>(ctg + row[k] * CP + ch[k] * 8);
                finish(c4, r4[k], row[k], ch[k], mq + row[k], cbn * BN + ch[k] * 8);
              }
            }
          } else {
            for (int q = gtid; q < 64 * CPR; q += 256) {
              const int row = q / CPR, ch = q - row * CPR;
              const int m = mq + row, n = cbn * BN + ch * 8;
              if (m >= p.M || n >= p.N) continue;
              const u32x4 c4 = *reinterpret_cast<const u32x4*>(ctg + row * CP + ch * 8);
              uint4 r4 = make_uint4(0u, 0u, 0u, 0u);
              if (res) r4 = *reinterpret_cast<const uint4*>(res + (long long)m * p.ldr + n);
              finish(c4, r4, row, ch, m, n);
            }
          }
        }
#ifdef SASPA_GEMM_ABLATION
        if (stamp && qt == 0) sr_e1 = __builtin_amdgcn_s_memrealtime();
#endif
        if (p.gn_stats) {
          lds_barrier();                               // values a residual / activation changed were written back above
          const int nrows = min(64, p.M - mq);
          if (grg < rgs) {
            for (int r = grg; r < nrows; r += rgs) {
              const uint32_t* src = reinterpret_cast<const uint32_t*>(ctg + r * CP + gu * gunit);
              for (int j = 0; j < gunit; j += 2) {
                // two bf16 per dword: v_dot2c_f32_bf16 against (1, 1) and against itself (as gn_tile_stats)
                const bf16x2_t w2 = __builtin_bit_cast(bf16x2_t, src[j >> 1]);
                gsm = __builtin_amdgcn_fdot2_f32_bf16(w2, __builtin_bit_cast(bf16x2_t, 0x3F803F80u), gsm, false);
                gsq = __builtin_amdgcn_fdot2_f32_bf16(w2, w2, gsq, false);
              }
            }
          }
        }
        lds_barrier();
      }
      if (p.gn_stats) {
        // fixed summation order (rows of a row group in order, then the row groups in order): deterministic
        float* scr = reinterpret_cast<float*>(ct + 128 * CP) + ewm * 512;
        if (grg < rgs) {
          scr[(grg * nunits + gu) * 2] = gsm;
          scr[(grg * nunits + gu) * 2 + 1] = gsq;
        }
        lds_barrier();
        if (gtid < nunits * 2 && m0 < p.M) {
          const int uu = gtid >> 1, k = gtid & 1;
          float a = 0.f;
          for (int g = 0; g < rgs; ++g) a += scr[(g * nunits + uu) * 2 + k];
          p.gn_stats[((long long)(cbm * 2 + ewm) * (p.N / p.gn_unit) + (cbn * BN) / p.gn_unit + uu) * 2 + k] = a;
        }
        lds_barrier();
      }
    }
#ifdef SASPA_GEMM_ABLATION
    if (stamp && etid == 0) {
      // diagnostics: the caller's `out` allocation extends 16 x 8 bytes per workgroup beyond M rows
      const unsigned long long sr_issued = __builtin_amdgcn_s_memrealtime();
      __threadfence();
      unsigned long long* o = reinterpret_cast<unsigned long long*>(reinterpret_cast<T*>(p.out) + (long long)p.M * p.ldo) + 16 * blockIdx.x;
      o[0] = sr0;
      o[1] = sr1;
      o[2] = ((st1 - st0) << 20) | ((sr0 - sr_entry) & 0xfffff);   // loop clocks | entry -> loop start ticks
      o[3] = __builtin_amdgcn_s_memrealtime();                     // this wave's stores are visible
      o[4] = sr_entry;
      o[5] = sr_setup;                                             // per-lane im2col state ready
      o[6] = sr_issue;                                             // prologue DMAs issued
      o[7] = sr_e0;                                                // K loop drained, epilogue starts
      o[8] = sr_e1;                                                // half 0: stores issued
      o[9] = sr_issued;                                            // half 1: stores issued
    }
#endif
    if (!has_next) break;
    tile = next;
    setup_tile(tile);
  }
  if (abl & 32) {
    __syncthreads();
    if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && (wave & 3) == 0 && lane == 0) {
      int* dbg = reinterpret_cast<int*>(p.out) + (wave >> 2) * 4;
      dbg[0] = dbgv[0]; dbg[1] = dbgv[1]; dbg[2] = dbgv[2]; dbg[3] = dbgv[3];
    }
  }
}

template <int FN, int ONEBAR>
int launch_pp(const SaspaGemmParams& p, hipStream_t s, int ksplit) {
  constexpr int BM = 256, BN = 64 * FN;
  const int tiles = ((p.N + BN - 1) / BN) * ((p.M + BM - 1) / BM);
  const int zy = ksplit * p.nb1 * p.nb2;
  // (balancing the launches of the paired encoders for HALF the chip -- cap 128 under SaspaGemmParams.sharing -- measured
  // 0 / -1.2 / 0 % at 512x512 / 512x704 / 512x768: the dispatcher's own interleaving of the two queues does better)
  const int gx = saspa_balanced_grid(tiles, 256 / zy);
  dim3 grid(gx, ksplit, p.nb1 * p.nb2);
  const bool pw = p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0 && !p.upsample;
  static const int abl = getenv("SASPA_GEMM_ABLATE") ? (atoi(getenv("SASPA_GEMM_ABLATE")) & 15) : 0;   // diagnostics only
  static const int stamp = getenv("SASPA_GEMM_STAMP") ? (atoi(getenv("SASPA_GEMM_STAMP")) & 1) : 0;
  static const int eabl = getenv("SASPA_GEMM_EPI_ABLATE") ? (atoi(getenv("SASPA_GEMM_EPI_ABLATE")) & 7) : 0;
  const int ta = tiles | (abl << 28) | (stamp << 27) | (eabl << 24);
  const int npart8 = saspa_gemm_npart8(p, BM, BN, gx, tiles);
  if (pw) hipLaunchKernelGGL((gemm_pp_kernel<FN, true, false, ONEBAR>), grid, dim3(512), 0, s, p, ta, npart8);
  else if (p.upsample) hipLaunchKernelGGL((gemm_pp_kernel<FN, false, true, ONEBAR>), grid, dim3(512), 0, s, p, ta, npart8);
  else hipLaunchKernelGGL((gemm_pp_kernel<FN, false, false, ONEBAR>), grid, dim3(512), 0, s, p, ta, npart8);
  SASPA_CHECK_LAUNCH();
  if (ksplit > 1) return saspa_gemm_splitk_reduce(p, s, ksplit);
  return 0;
}

}  // namespace

bool saspa_gemm_pp_eligible(const SaspaGemmParams& p) {
  const int ctot = p.c0 + p.c1;
  if (p.dtype != SASPA_BF16) return false;
  // fused GEGLU: whole 320-column tiles of the per-160 packing, no residual, no K slices (the caller passes ksplit 1)
  if (p.act == SASPA_ACT_GEGLU && ((p.N % 320) != 0 || p.residual || p.alpha != 1.0f)) return false;
  if ((ctot % 64) != 0 || (p.c1 > 0 && (p.c0 % 64) != 0)) return false;        // a K-tile lies in one tap of one source
  if ((p.N % 8) != 0 || (p.ldo % 8) != 0 || (p.residual && (p.ldr % 8) != 0)) return false;
  const bool pw = p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0 && !p.upsample;
  if (!pw) {
    // 3x3 / pad 1 window (any stride, optional nearest x2): 8 halo taps + an always-inside centre tap
    if (p.kh != 3 || p.kw != 3 || p.pad != 1) return false;
    const int hv = p.upsample ? 2 * p.hin : p.hin, wv = p.upsample ? 2 * p.win : p.win;
    if ((p.hout - 1) * p.stride > hv - 1 || (p.wout - 1) * p.stride > wv - 1) return false;
    if (p.upsample && (p.hin >= 16000 || p.win >= 16000)) return false;
  }
  // per-lane offsets are 24-bit pixel index x 24-bit pitch products
  if ((long long)p.batch * p.hin * p.win >= (1ll << 24) || p.M >= (1 << 24)) return false;
  if ((long long)p.lda0 * 2 >= (1ll << 24) || (long long)p.lda1 * 2 >= (1ll << 24)) return false;
  return true;
}

int saspa_gemm_pp_launch(const SaspaGemmParams& p, hipStream_t s, int ksplit, int fn) {
  if (!saspa_gemm_pp_eligible(p)) return SASPA_ERANGE;
  SASPA_DRY_RETURN(SASPA_GEMM_WIDE, ksplit);
  // K loop flavour (read per launch; all are bit-identical: same MFMA order per accumulator):
  //   SASPA_GEMM_PP_LOOP=2 (default since round 5)  two-barrier ping-pong, TWO 40-MFMA intervals per K-tile: +2 ... +8 % on every
  //                         shape against =0 (tools/pp_ab.py, profiles/r5_pp_long_ab.txt)
  //   =0  two-barrier ping-pong, four 20-MFMA phases per K-tile (rounds 2 - 4)
  //   =1  one barrier per phase, asymmetric programs (round 3: 12 - 30 % SLOWER, profiles/r3_pp_ab_v*.txt).  Only in the
  //       diagnostics library (`make ABLATION=1`): its 340 bytes of scratch per lane are not something the shipped library should
  //       ever ask the runtime for; the shipped library reads =1 as the default
  if (fn == 4 || fn == 5) {
    const char* e = getenv("SASPA_GEMM_PP_LOOP");
    const int loop = e ? atoi(e) : 2;
#ifdef SASPA_GEMM_ABLATION
    if (loop == 1) fn += 10;
#endif
    if (loop != 0 && loop != 1) fn += 20;
#ifndef SASPA_GEMM_ABLATION
    if (loop == 1) fn += 20;
#endif
  }
  if (fn == 5) return launch_pp<5, 0>(p, s, ksplit);
  if (fn == 4) return launch_pp<4, 0>(p, s, ksplit);
#ifdef SASPA_GEMM_ABLATION
  if (fn == 15) return launch_pp<5, 1>(p, s, ksplit);
  if (fn == 14) return launch_pp<4, 1>(p, s, ksplit);
#endif
  if (fn == 25) return launch_pp<5, 2>(p, s, ksplit);
  if (fn == 24) return launch_pp<4, 2>(p, s, ksplit);
  return SASPA_ERANGE;
}
