// Halo-tiled 3x3 / stride 1 / pad 1 convolution with the consuming GroupNorm (+ SiLU) applied to the input tile IN LDS
// (ResnetBlock2D: norm1 -> SiLU -> conv1, norm2 -> SiLU -> conv2): the normalised tensor is never written to HBM and no
// GroupNorm apply pass is launched.
//
// Why a third conv kernel: saspa_gemm_pp.hip stages, for every one of the nine taps, the 256 x 64 im2col slice of the input
// again (9 x 32 KB of LDS-DMA per 64-channel chunk and output tile).  Here the input pixels an output tile needs -- its 256
// pixels plus a one-pixel halo, ~400 rows -- go to LDS ONCE per 32-channel chunk (25 KB instead of 9 x 16 KB), the
// GroupNorm scale / shift and SiLU are applied there once (1x the activation work; in the im2col form it would be 9x), and
// the nine taps read shifted windows of that tile.  The weights keep streaming through the two-buffer ring of the 8-wave
// kernel, and the K loop keeps its structure (two wave groups one barrier apart, four phases of 20 MFMAs per 64-deep
// K-tile, counted vmcnt, raw s_barrier): cdna_hip_programming.md "256^2 8-phase template".
//
// Layout / schedule:
//   * output tile 256 pixels (consecutive m = (b, oy, ox), inside ONE image: H*W % 256 == 0) x BN = 64*FN channels;
//     waves 2 (M) x 4 (N) as in saspa_gemm_pp.hip, D^T form (weights = MFMA A operand).
//   * halo tile: the pixels are addressed in the PADDED image (row pitch W + 2, one pad row above and below); an output
//     pixel's window is rows  base + dy * (W + 2) + dx  of the linear range [q_first - (W + 2) - 1, q_last + (W + 2) + 1]
//     (<= 512 rows), so a tap is ONE scalar offset for the whole tile, whatever the tile's alignment to image rows.
//     A row is 32 channels = 64 bytes; 16-byte chunk c of row r sits at chunk c ^ (((r >> 2) & 1) << 1): conflict-free
//     ds_read_b128 for every window shift (searched exhaustively, tools/halo_swizzle.py); applied on the DMA source side.
//   * K order of the packed weights: ((chunk32 * 9 + tap) * 32 + c) (SASPA_KORDER_CHUNK32).  A 64-deep K-tile of weights
//     = two (chunk32, tap) slots; a PERIOD = two chunks A, B = 18 slots = 9 K-tiles.  Chunk A lives in halo buffer 0 (read by
//     K-tiles 0..4 of the period), chunk B in buffer 1 (K-tiles 4..8); a buffer is reloaded while the other one is read:
//         K-tile 0 phases 1-3, K-tile 1 phase 0 : DMA the next chunk B into buffer 1 (4 pieces of 16 rows per wave) + its gamma | beta
//         K-tile 2 phase 0                      : wave 0 derives the chunk's scale / shift (32 channels) into LDS
//         K-tile 2 phases 1-3, K-tile 3 phase 0 : every lane normalises the 16 bytes IT brought in (one piece per phase)
//         K-tiles 5, 6, 7, 8                    : the same for the next period's chunk A into buffer 0
//     (each in the part of a phase where the other wave group owns the matrix pipe).
//   * counted vmcnt once per K-tile (phase 3): 3 weight pieces of phases 1-3 + the halo pieces issued in them stay in flight.
//   * epilogue: saspa_gemm_pp.hip's (bias / time-embedding row from LDS slots filled by DMA at tile set-up, bf16 staging,
//     coalesced stores with the residual loads in flight, GroupNorm statistics of the output, fp32 slabs under split-K).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "gemm_internal.h"

namespace {

typedef bf16_t T;

struct HaloGeom {
  int wp;          // W + 2
  int nper;        // channel-chunk pairs = (c0 + c1) / 64
  int per_slice;   // pairs per K slice
  int gn_slabs;    // channel slabs of the statistics pass (partial-sum form)
  int ntiles;
};

// MFMA through inline asm with the accumulator pinned in the ACCUMULATOR file ("+a").  With the builtin the compiler keeps the
// accumulators in the same file as everything else, re-homes them between MFMAs and, at 256 registers, spills accumulator
// fragments to scratch inside the K loop -- every reload sits behind s_waitcnt vmcnt(0), i.e. drains the weight ring (measured:
// 0.9 us per reload group; 115 -> 154 us per launch when a schedule edit added nine of them).  hipcc gives a 2-waves-per-SIMD
// kernel that uses AGPRs 128 AGPRs + 128 VGPRs, so the first FOUR 16-column fragments of a wave's row block live in AGPRs
// (8 x 4 x 4 = 128) and the fifth (FN = 5) stays with the builtin in VGPRs (32) beside the operand fragments (36).
// Hazards the compiler does not pad for an asm statement (guide 5.7 item 2): a VALU-written operand -> MFMA needs two wait
// states (`first` opens the phase's block with s_nop 1: the operands are ds_read results, but a compiler copy may sit in
// front); an accumulator is read again as C no sooner than 19 MFMAs later; its first VALU reader is the epilogue.
__device__ __forceinline__ void mma_a(const u32x4& wf, const u32x4& xf, f32x4& acc, const bool first) {
  const f32x4 a = __builtin_bit_cast(f32x4, wf), b = __builtin_bit_cast(f32x4, xf);
  if (first) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void mma_v(const u32x4& wf, const u32x4& xf, f32x4& acc) {
  // (asm too: with AGPRs in the kernel the builtin is selected in its AGPR form and copied in and out of the full file)
  const f32x4 a = __builtin_bit_cast(f32x4, wf), b = __builtin_bit_cast(f32x4, xf);
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int V> using ic = std::integral_constant<int, V>;

// REG: W % 16 == 0 -- a 16-pixel fragment never straddles image rows, so the row wraps between a wave's fragments are
// wave-uniform (scalar arithmetic) instead of a per-lane nibble table
template <int FN, bool GN, bool REG>
__global__ __launch_bounds__(512) void conv_halo_kernel(const SaspaGemmParams p, const SaspaConvGnParams g, const HaloGeom geo) {
  constexpr int BM = 256, BN = 64 * FN, SZ = 2;
  constexpr int HROWS = 480;
  constexpr int HB = HROWS * 4;                        // u32x4 per halo buffer (64-byte rows)
  constexpr int BB = BN * 8;                           // u32x4 per weight buffer (128-byte rows)
  constexpr int OFF_H0 = 0, OFF_H1 = HB, OFF_B0 = 2 * HB;
  constexpr int NLDS = 2 * HB + 2 * BB;
  constexpr int CP = BN + 8;                           // epilogue row pitch (elements)
  constexpr int EPI = 128 * CP * SZ / 16;
  static_assert(EPI + 256 <= NLDS, "epilogue staging + statistics scratch fit the K-loop buffers");
  constexpr int ADDV_F = 2 * BN;                       // bias | time-embedding row of the tile's image
  constexpr int GNF = 128 + 2 * 64 + 2 * 64;           // gstat float2[64] | gb[2][gamma 32 | beta 32] | ss[2][scale 32 | shift 32]
  constexpr int HRELF = 512 * 8;                       // per thread: DMA byte offset of its row in halo piece j, for either source's pitch
  // ONE __shared__ object (a second one makes the compiler guard every LDS read of the loop with vmcnt(0): saspa_gemm_pp.hip)
  __shared__ u32x4 lds[NLDS + ADDV_F / 4 + GNF / 4 + HRELF / 4];
  float* const addvb = reinterpret_cast<float*>(lds + NLDS);
  float* const addvr = addvb + BN;
  float* const gstat = addvr + BN;                     // (mean, rstd) per group
  float* const gbb = gstat + 128;
  float* const ssb = gbb + 128;
  unsigned* const hoffb = reinterpret_cast<unsigned*>(ssb + 128);
  // lane id re-derived at the point of use (2 VALU) behind an optimisation barrier: every lane-derived constant the K loop
  // needs a few times per period (halo piece offsets, weight fragment offsets) would otherwise be hoisted out of the loop and
  // held in -- or spilled from -- registers the accumulators need
  auto lane_now = []() __attribute__((always_inline)) {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int frow = lane & 15, fg = lane >> 4;
  const int lr = lane >> 3;                            // weight DMA: row inside an 8-row piece
  const int kcs = (lane & 7) ^ lr;                     //             logical 16-byte chunk this lane fetches
  // halo DMA: logical 16-byte chunk (8 channels) a lane fetches = (lane & 3) ^ (((lane >> 4) & 1) << 1), the same for its 4 pieces

  const int nbn = (p.N + BN - 1) / BN;
  const int G = gridDim.x;
  int tile;
  {
    const int L = blockIdx.x;
    const int qd = G >> 3, rr = G & 7, xcd = L & 7, idx = L >> 3;
    tile = (xcd < rr ? xcd * (qd + 1) : rr * (qd + 1) + (xcd - rr) * qd) + idx;
  }
  const T* a0 = reinterpret_cast<const T*>(p.a0);
  const T* a1 = reinterpret_cast<const T*>(p.a1);
  const T* w = reinterpret_cast<const T*>(p.w);

  const int Wd = p.wout, Hd = p.hout;
  const int hw = Hd * Wd;
  const int wp = geo.wp;
  const int wp64 = wp * 64;
  const int ctot = p.c0 + p.c1;
  // K slice of this workgroup, in periods (pairs of 32-channel chunks)
  const int pr0 = blockIdx.y * geo.per_slice;
  const int npr = max(0, min(geo.nper, pr0 + geo.per_slice) - pr0);
  const int nk = 9 * npr;
  const int kt0 = 9 * pr0;

  auto make_desc = [](const void* base, bool live) __attribute__((always_inline)) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), (short)0, live ? 0x7fffffff : 0, 0x00020000);
  };

  // ---- per-tile state ----
  int bm = 0, bn = 0, img = 0;
  // halo row of fragment i's window origin, as a byte offset: X0 + i * 1024 + 128 * (image rows fragment i lies below fragment 0,
  // 4 bits per fragment in xw) -- two registers instead of eight (the K loop has none to spare)
  int X0 = 0;
  unsigned xw = 0;
  int ad[4] = {0, 0, 0, 0};                            // LDS byte addresses of the NEXT phase's four pixel fragments
  unsigned hoffn = kInvalid;                           // DMA offset of the next phase's halo piece (if it has one)
  unsigned hvalid = 0;                                 // bit j: this lane's row of halo piece j is a pixel of the image (not padding)
  unsigned offb0 = 0;
  int nrows = 0;
  int stg = 0;                                         // K-tiles of weights staged so far
  rsrc_t rswv = make_desc(w, false);
  int soffw = 0;

  auto setup_tile = [&](int t) __attribute__((always_inline)) {
    bm = t / nbn;
    bn = t - bm * nbn;
    const int m0 = bm * BM;
    img = m0 / hw;
    const int rem0 = m0 - img * hw;
    const int oy0 = rem0 / Wd, ox0 = rem0 - oy0 * Wd;
    const int qf = (oy0 + 1) * wp + ox0 + 1;
    const int oyl = (rem0 + BM - 1) / Wd, oxl = rem0 + BM - 1 - oyl * Wd;
    const int Lh = (oyl + 1) * wp + oxl + 1 - qf + 2 * wp + 3;
    const int qlo = qf - wp - 1;
    {
      // byte offset (pixel * pitch + logical chunk * 16, or "out of range" for padding) of this lane's row in halo piece j,
      // for source 0's and source 1's pixel pitch -> LDS table [j][source][thread]: the K loop fetches it with one ds_read
      const int hlc = (lane & 3) ^ (((lane >> 4) & 1) << 1);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = (wave + 8 * j) * 16 + (lane >> 2);
        const int q = qlo + r;
        const int prow = q / wp, pcol = q - prow * wp;
        const int iy = prow - 1, ix = pcol - 1;
        const bool ok = r < Lh && (unsigned)iy < (unsigned)Hd && (unsigned)ix < (unsigned)Wd;
        const unsigned rel = (unsigned)(iy * Wd + ix);
        if (j == 0) hvalid = 0;
        hvalid |= (ok && (wave + 8 * j) * 16 < HROWS) ? (1u << j) : 0u;
        hoffb[(j * 2 + 0) * 512 + tid] = ok ? rel * (unsigned)(p.lda0 * SZ) + (unsigned)(hlc * 16) : kInvalid;
        hoffb[(j * 2 + 1) * 512 + tid] = ok ? rel * (unsigned)(p.lda1 * SZ) + (unsigned)(hlc * 16) : kInvalid;
      }
    }
    {
      const int rem = rem0 + wm * 128 + frow;
      int oy = rem / Wd, ox = rem - oy * Wd;
      X0 = ((oy + 1) * wp + ox + 1 - qf) * 64 + fg * 16;
      xw = 0;
      int wr = 0;
      if (REG) ox = (rem0 + wm * 128) % Wd;            // the fragment's first pixel: the same row arithmetic for every lane -> scalar
#pragma unroll
      for (int i = 1; i < 8; ++i) {
        ox += 16;
        while (ox >= Wd) { ox -= Wd; ++wr; }
        xw |= (unsigned)wr << (4 * i);                 // <= 15 row wraps inside 128 pixels: W >= 9 (checked by _eligible)
      }
      if (REG) xw = (unsigned)__builtin_amdgcn_readfirstlane((int)xw);
    }
    offb0 = (unsigned)((bn * BN + wave * 8 + lr) * p.ldw * SZ + kcs * 16);
    nrows = p.N - bn * BN;
    stg = 0;
  };

  // weights of K-tile (kt0 + stg): descriptor + scalar offset; zero records past the slice (the DMA writes zeros, touches nothing)
  auto begin_stage = [&]() __attribute__((always_inline)) {
    rswv = make_desc(w, stg < nk);
    soffw = (kt0 + stg) * 64 * SZ;
    ++stg;
  };
  const rsrc_t rsnull = make_desc(w, false);
  auto stage_b = [&](const int i, const int buf) __attribute__((always_inline)) {
    // rows beyond N (wave-uniform per 8-row piece): the zero-record descriptor (a scalar select; a per-lane "invalid" offset
    // would be one more loop-invariant register per piece)
    const bool ok = (wave + 8 * i) * 8 < nrows;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(ok ? rswv : rsnull, (lds_void_t*)(lds + buf + (wave * 8 + i * 64) * 8), 16, (int)offb0,
                                             soffw + i * 64 * p.ldw * SZ, 0, 0);
  };
  // halo piece j of 32-channel chunk c32 (global index over [source 0 | source 1]) -> halo buffer hb
  auto stage_h = [&](const int j, const int c32, const int hb, const bool direct = false) __attribute__((always_inline)) {
    const int cc = c32 * 32;
    const bool live = cc < ctot;
    const bool s0 = cc < p.c0;
    const int ld = s0 ? p.lda0 : p.lda1;
    const T* base = (s0 ? a0 + (long long)img * hw * p.lda0 + cc : a1 + (long long)img * hw * p.lda1 + (cc - p.c0));
    const rsrc_t rs = make_desc(base, live);
    if ((wave + 8 * j) * 16 >= HROWS) return;          // pieces 30, 31 lie beyond the buffer (the strict vmcnt count does not depend on them)
    // the offset was fetched from the LDS table between the previous phase's MFMAs (the prologue reads it here)
    const unsigned off = direct ? hoffb[(j * 2 + (s0 ? 0 : 1)) * 512 + wave * 64 + lane_now()] : hoffn;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(lds + hb + (wave + 8 * j) * 64), 16, (int)off, 0, 0, 0);
  };
  // gamma | beta of chunk c32 (host-packed [C / 32][64] floats) -> gb[sb]: one 256-byte piece, issued by EVERY wave (same
  // bytes, same place: keeps the vmcnt arithmetic of all waves identical)
  auto stage_gb = [&](const int c32, const int sb) __attribute__((always_inline)) {
    const rsrc_t rs = make_desc(g.gamma_beta32, c32 * 32 < ctot);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(gbb + sb * 64), 4, lane_now() * 4, c32 * 256, 0, 0);
  };
  const int cpg = GN ? ctot / g.groups : 1;
  auto make_ss = [&](const int c32, const int sb) __attribute__((always_inline)) {
    if (wave == 0) {
      const int ln = lane_now();
      const int ch = c32 * 32 + ln;
      if (ln < 32 && ch < ctot) {
        const int gi = ch / cpg;
        const float mean = gstat[2 * gi], rstd = gstat[2 * gi + 1];
        const float scv = gbb[sb * 64 + ln] * rstd;                     // the arithmetic of gn_apply_kernel
        const float shv = gbb[sb * 64 + 32 + ln] - mean * scv;
        ssb[sb * 64 + ln] = scv;
        ssb[sb * 64 + 32 + ln] = shv;
      }
    }
  };
  const bool gsilu = GN && g.act == SASPA_ACT_SILU;
  // LDS accesses of the normalisation through inline asm: written as plain loads, the compiler puts s_waitcnt vmcnt(0) in front
  // of the scale / shift reads (it cannot tell them from the gamma | beta DMA target next to them), which drains the weight
  // ring eight times per period.  The explicit lgkmcnt + sched_barrier pair is rule 18 of the guide (an asm ds_read's consumer
  // must not be hoisted above the wait).
  // (the asm result is a FLOAT vector on purpose: with an integer 4-vector as "=v" output hipcc 7.2 reads element 0 for elements
  // 0 and 1 -- checked in isolation --, with f32x4 the sub-registers are right)
  auto lds_rd = [](const unsigned byte_addr) __attribute__((always_inline)) {
    f32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(byte_addr));
    return v;
  };
  constexpr unsigned SSB_BYTES = (unsigned)(NLDS * 16 + ADDV_F * 4 + 128 * 4 + 128 * 4);      // byte offset of ss[] inside lds
  // half a piece per phase: 4 of the 8 channels of this lane's 16 bytes of halo piece hs >> 1 (~150 cycles of VALU, which fits
  // beside the other wave group's 20 MFMAs; a whole piece per phase did not).  Split in two so that its LDS reads are the FIRST
  // instructions of a phase's read section and their latency runs under the fragment reads and the DMA issue:
  //   tr_issue : ds_read of the 8 data bytes, 4 scales, 4 shifts            (top of the read section)
  //   tr_finish: lgkmcnt(0), affine + SiLU, ds_write                       (after the DMA issue; the write is retired by the
  //              lgkmcnt(0) every phase executes ahead of its MFMAs -- two barriers before any other wave reads the row)
  unsigned long long trd = 0;
  f32x4 trs = {0.f, 0.f, 0.f, 0.f}, trt = {0.f, 0.f, 0.f, 0.f};
  unsigned trda = 0;
  auto tr_issue = [&](const int hs, const int hb, const int sb) __attribute__((always_inline)) {
    const int j = hs >> 1, h = hs & 1;
    const int ln = lane_now();
    const int hlc = (ln & 3) ^ (((ln >> 4) & 1) << 1);
    const unsigned lds_base = (unsigned)(size_t)(lds_void_t*)lds;
    trda = lds_base + (unsigned)((hb + (wave + 8 * j) * 64 + ln) * 16 + h * 8);
    const unsigned sa = lds_base + SSB_BYTES + (unsigned)(sb * 256 + hlc * 32 + h * 16);
    asm volatile("ds_read_b64 %0, %1" : "=v"(trd) : "v"(trda));      // (a 64-bit SCALAR: 2-vectors as asm outputs have the sub-register bug too)
    trs = lds_rd(sa);
    trt = lds_rd(sa + 128);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto tr_finish = [&](const int hs) __attribute__((always_inline)) {
    const int j = hs >> 1;
    if ((hvalid >> j) & 1u) {                          // padding stays zero: the conv pads the NORMALISED tensor
      const unsigned d0 = (unsigned)trd, d1 = (unsigned)(trd >> 32);
      float v[4] = {__builtin_bit_cast(float, d0 << 16), __builtin_bit_cast(float, d0 & 0xffff0000u),
                    __builtin_bit_cast(float, d1 << 16), __builtin_bit_cast(float, d1 & 0xffff0000u)};
      if (gsilu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = silu_fast(v[e] * trs[e] + trt[e]);
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = v[e] * trs[e] + trt[e];
      }
      const unsigned long long o = (unsigned long long)pack2(v[0], v[1]) | ((unsigned long long)pack2(v[2], v[3]) << 32);
      asm volatile("ds_write_b64 %0, %1" ::"v"(trda), "v"(o) : "memory");
    }
  };

  int cur = OFF_B0, oth = OFF_B0 + BB;

  f32x4 acc[8][4];                                     // AGPRs (asm MFMA)
  f32x4 accv[8];                                       // fifth column fragment (FN = 5): VGPRs (builtin MFMA)
  u32x4 wb[FN], xa[4];

  // LDS addresses of the four pixel fragments phase (TT, P) multiplies: window origin + tap offset, chunk ^= 2 on rows with bit 2
  // set.  Computed one phase AHEAD, between the MFMAs of the previous phase (the read section of a phase is what the other
  // wave group's matrix block has to cover: address arithmetic in front of the ds_reads made it 30 % longer)
  auto addr4 = [&](auto tc, auto pc) __attribute__((always_inline)) {
    constexpr int TT = decltype(tc)::value, P = decltype(pc)::value;
    constexpr int KK = P >> 1, I0 = 4 * (P & 1);
    constexpr int S = 2 * TT + KK;                     // slot of the period
    constexpr int hb = S < 9 ? OFF_H0 : OFF_H1;
    constexpr int tap = S < 9 ? S : S - 9;
    constexpr int dy = tap / 3, dx = tap - dy * 3;
    int toff = dy * wp64 + dx * 64;
    // opaque to LICM: the 8 x 9 (fragment, tap) addresses are loop invariants the compiler would otherwise precompute, keep in
    // 72 registers and spill (230 scratch reloads inside the loop, each behind vmcnt(0))
    asm volatile("" : "+s"(toff));
    int x0v = X0;
    if constexpr (REG) {
      int xws = (int)xw;
      asm volatile("" : "+v"(x0v), "+s"(xws));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = x0v + ((((xws >> (4 * (I0 + i))) & 15) << 7) + toff + (I0 + i) * 1024);
        ad[i] = (x ^ ((x >> 3) & 32)) + hb * 16;
      }
    } else {
      unsigned xwv = xw;
      asm volatile("" : "+v"(xwv), "+v"(x0v));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int x = x0v + (int)(((xwv >> (4 * (I0 + i))) & 15u) << 7) + (toff + (I0 + i) * 1024);
        ad[i] = (x ^ ((x >> 3) & 32)) + hb * 16;
      }
    }
  };
  // halo piece schedule (piece index or -1): chunk 2 pr + 1 -> buffer 1 at (0,1) (0,2) (1,0) (1,1) -- its last reader was phase
  // (8,3) --; chunk 2 pr + 2 -> buffer 0 at (4,3) (5,0) (5,1) (5,2) -- last reader (4,1).  Both have landed at the phase-3 wait
  // of K-tiles 1 / 5, where the chunk's scale / shift are derived; the eight half-piece normalisation steps follow, one per phase.
  auto halo_piece = [](const int tt, const int p) constexpr {
    return (tt == 0 && (p == 1 || p == 2)) ? p - 1 : (tt == 1 && p <= 1) ? 2 + p : (tt == 4 && p == 3) ? 0 : (tt == 5 && p <= 2) ? 1 + p : -1;
  };
  auto halo_is_b = [](const int tt) constexpr { return tt <= 1; };

  // one phase of K-tile TT (0..8 inside the period starting at chunk pair `pr`) out of weight buffer `cur`.  Phase P multiplies
  // the kk = P >> 1 half of the K-tile (ONE (chunk32, tap) slot) against four of the wave's eight pixel fragments: the weight
  // fragments of one half (20 registers) are live at a time, re-read for the second half in phase 2.
  auto phase = [&](auto tc, auto pc, const int pr) __attribute__((always_inline)) {
    constexpr int TT = decltype(tc)::value, P = decltype(pc)::value;
    constexpr int KK = P >> 1, I0 = 4 * (P & 1);
    constexpr int TN = P == 3 ? (TT + 1) % 9 : TT, PN = (P + 1) & 3;          // the following phase
    // ---- read section (the other wave group owns the matrix pipe) ----
    // normalisation steps: step q of a chunk is ISSUED (its LDS reads) in phase q of the window and FINISHED (VALU + write) in
    // phase q + 1, so nothing in a read section waits for it.  Windows: chunk B K-tiles 2, 3 (+ (4,0)), chunk A' K-tiles 6, 7 (+ (8,0))
    constexpr int PH = TT * 4 + P;
    constexpr int qB = PH - 8, qA = PH - 24;          // phase index inside the window
    constexpr bool issB = GN && qB >= 0 && qB < 8, finB = GN && qB >= 1 && qB <= 8;
    constexpr bool issA = GN && qA >= 0 && qA < 8, finA = GN && qA >= 1 && qA <= 8;
    if constexpr ((P & 1) == 0) {
      const int ln = lane_now();
      const int rb = cur + (wn * (16 * FN) + (ln & 15)) * 8 + (((ln >> 4) ^ (ln & 7)) ^ (KK ? 4 : 0));
#pragma unroll
      for (int j = 0; j < FN; ++j) wb[j] = lds[rb + j * 16 * 8];
      __builtin_amdgcn_sched_barrier(0);               // weight reads are issued (and counted) before the halo reads
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) xa[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(lds) + ad[i]);
    // weight pieces of K-tile t+2 follow the LAST fragment read of buffer `cur` (phase 2): pieces 0, 1 in phase 3, one piece in
    // each of phases 0.. of the next K-tile (where the buffer is `oth`)
    const int c32b = 2 * (pr0 + pr) + 1;
    // (past the slice this is the next slice's first chunk, or zero records past the last channel: loaded, normalised, never read)
    const int c32a = 2 * (pr0 + pr) + 2;
    constexpr int lj = halo_piece(TT, P);
    if constexpr (lj >= 0 && P != 3) stage_h(lj, halo_is_b(TT) ? c32b : c32a, halo_is_b(TT) ? OFF_H1 : OFF_H0);
    if constexpr (GN && TT == 0 && P == 1) stage_gb(c32b, 1);
    if constexpr (GN && TT == 5 && P == 0) stage_gb(c32a, 0);
    if constexpr (P == 3) {
      begin_stage();
      stage_b(0, cur);
      stage_b(1, cur);
      stage_b(2, cur);
      // (4,3): the first piece of chunk A' goes out BEHIND the weight pieces and stays in flight across this phase's wait
      if constexpr (lj >= 0) stage_h(lj, c32a, OFF_H0);
    } else if constexpr (P == 0) {
#pragma unroll
      for (int i = 3; i < FN; ++i) stage_b(i, oth);    // >= 3 phases in flight before the phase-3 wait
    }
    if constexpr (finB) tr_finish(qB - 1);
    if constexpr (finA) tr_finish(qA - 1);
    if constexpr (issB) tr_issue(qB, OFF_H1, 1);
    if constexpr (issA) tr_issue(qA, OFF_H0, 0);       // (past the slice: the next slice's chunk or zeros; nobody reads it)
    // phase 3: everything but the three weight pieces just issued has landed -- K-tile t+1 (read from the next phase on) and, in
    // K-tiles 1 / 6, the halo pieces and gamma | beta of the chunk being reloaded
    if constexpr (P == 3) wait_vm<(lj >= 0 ? 4 : 3)>();
    if constexpr (GN && P == 3 && TT == 1) make_ss(c32b, 1);                      // gamma | beta of the chunk have just landed
    if constexpr (GN && P == 3 && TT == 5) make_ss(c32a, 0);
    constexpr bool wrote = GN && P == 3 && (TT == 1 || TT == 5);
    if constexpr (wrote) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // scale / shift or normalised rows are in LDS before the barrier
    else if constexpr (P == 2) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory"); // weight fragment reads retired: the slot may be restaged
    __builtin_amdgcn_s_barrier();
    // ---- matrix section; between the MFMAs: the next phase's addresses, the next halo piece's offset, and (GN) this phase's
    // share of the normalisation of the chunk that landed ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) mma_a(wb[j], xa[i], acc[I0 + i][j], i == 0 && j == 0);
      if constexpr (FN == 5) mma_v(wb[4], xa[i], accv[I0 + i]);
    }
    __builtin_amdgcn_s_setprio(0);
    if constexpr (GN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // incl. the normalisation reads issued in this phase (consumed in the next)
    // everything below issues while the 20 MFMAs drain (the matrix pipe runs 16 cycles per instruction, its issue takes 8): no
    // register of the fragments is live any more, so the normalisation's temporaries cost the loop nothing
    __builtin_amdgcn_sched_barrier(0);
    addr4(ic<TN>{}, ic<PN>{});
    constexpr int nj = halo_piece(TN, PN);
    if constexpr (nj >= 0) {
      if ((wave + 8 * nj) * 16 < HROWS) {
        // source 0 or 1 of the chunk the next piece belongs to (wave-uniform)
        const int cn = (halo_is_b(TN) ? c32b : c32a) * 32;
        hoffn = hoffb[(nj * 2 + (cn < p.c0 ? 0 : 1)) * 512 + wave * 64 + lane_now()];
      }
    }
    __builtin_amdgcn_s_barrier();
  };
  auto ktile = [&](auto tc, const int pr) __attribute__((always_inline)) {
    phase(tc, ic<0>{}, pr);
    phase(tc, ic<1>{}, pr);
    phase(tc, ic<2>{}, pr);
    phase(tc, ic<3>{}, pr);
    const int t = cur; cur = oth; oth = t;
  };

  setup_tile(tile);
  for (;;) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      accv[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    if (gridDim.y == 1) {
      // bias and the time-embedding row of the tile's image -> LDS by DMA (256-byte pieces, one wave each)
      constexpr int PCS = BN / 64;
      for (int pc = wave; pc < PCS * 2; pc += 8) {
        const int rowi = pc / PCS, part = pc - rowi * PCS;
        const int voff = (bn * BN + part * 64 + lane) * 4;
        if (rowi == 0) {
          const rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.bias), (short)0, p.bias ? p.N * 4 : 0, 0x00020000);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_t*)(addvb + part * 64), 4, voff, 0, 0, 0);
        } else {
          const rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.rowvec) + (p.rowvec ? (long long)img * p.ldrv : 0), (short)0,
                                                              p.rowvec ? p.N * 4 : 0, 0x00020000);
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_void_t*)(addvr + part * 64), 4, voff, 0, 0, 0);
        }
      }
    }

    if (nk > 0) {
      // ---- prologue: chunk A of the first period -> halo buffer 0 (+ gamma | beta), weights of K-tile 0 and pieces 0..2 of K-tile 1
#pragma unroll
      for (int j = 0; j < 4; ++j) stage_h(j, 2 * pr0, OFF_H0, true);
      if (GN) stage_gb(2 * pr0, 0);
      begin_stage();
#pragma unroll
      for (int i = 0; i < FN; ++i) stage_b(i, cur);
      begin_stage();
      stage_b(0, oth);
      stage_b(1, oth);
      stage_b(2, oth);
      if (GN) {
        // (mean, rstd) per group of the tile's image: the fp64 combine of gn_apply_kernel's prologue, same order -> same bits
        if (tid < 256) {
          const int sub = tid & 7;
          for (int g0 = 0; g0 < g.groups; g0 += 32) {
            const int gi = g0 + (tid >> 3);
            double sm = 0.0, sq = 0.0;
            if (gi < g.groups && g.stats0) {
              const int nblk = hw >> 7;
              const int glo = gi * cpg, ghi = glo + cpg;
              for (int src = 0; src < 2; ++src) {
                const float* st = src ? g.stats1 : g.stats0;
                const int cs = src ? p.c1 : p.c0, off = src ? p.c0 : 0;
                if (!st || cs == 0) continue;
                const int lo = max(glo, off) - off, hi = min(ghi, off + cs) - off;
                if (hi <= lo) continue;
                const int u0 = lo / g.unit, nu = (hi - lo) / g.unit, upr = cs / g.unit;
                const int total = nu * nblk;
                for (int e0 = sub; e0 < total; e0 += 64) {
                  float2 v[8];
#pragma unroll
                  for (int u = 0; u < 8; ++u) {
                    const int e = e0 + 8 * u;
                    v[u] = make_float2(0.f, 0.f);
                    if (e < total) {
                      const int blk = e / nu, uu = e - blk * nu;
                      v[u] = *reinterpret_cast<const float2*>(st + (((long long)img * nblk + blk) * upr + u0 + uu) * 2);
                    }
                  }
#pragma unroll
                  for (int u = 0; u < 8; ++u) {
                    sm += (double)v[u].x;
                    sq += (double)v[u].y;
                  }
                }
              }
            } else if (gi < g.groups) {
              const int nparts = g.nsplit * geo.gn_slabs;
              const float* base = g.partial + (long long)img * nparts * g.groups * 2;
              for (int i0 = sub; i0 < nparts; i0 += 64) {
                float2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                  const int i = i0 + 8 * u;
                  v[u] = make_float2(0.f, 0.f);
                  if (i < nparts) v[u] = *reinterpret_cast<const float2*>(base + ((long long)i * g.groups + gi) * 2);
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
            if (gi < g.groups && sub == 0) {
              const double n = (double)cpg * (double)hw;
              const double mean = sm / n;
              double var = sq / n - mean * mean;
              if (var < 0.0) var = 0.0;
              gstat[2 * gi] = (float)mean;
              gstat[2 * gi + 1] = (float)(1.0 / sqrt(var + (double)g.eps));
            }
          }
        }
      }
      wait_vm<3>();                                    // halo chunk, gamma | beta, bias / row vector and K-tile 0 have landed
      if (GN) {
        lds_barrier();
        make_ss(2 * pr0, 0);
        lds_barrier();
#pragma unroll
        for (int hs = 0; hs < 8; ++hs) {
          tr_issue(hs, OFF_H0, 0);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_sched_barrier(0);
          tr_finish(hs);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      addr4(ic<0>{}, ic<0>{});
      __builtin_amdgcn_s_barrier();
      if (wm == 1) __builtin_amdgcn_s_barrier();       // the wm = 1 group runs one barrier behind
      for (int pr = 0; pr < npr; ++pr) {
        ktile(ic<0>{}, pr);
        ktile(ic<1>{}, pr);
        ktile(ic<2>{}, pr);
        ktile(ic<3>{}, pr);
        ktile(ic<4>{}, pr);
        ktile(ic<5>{}, pr);
        ktile(ic<6>{}, pr);
        ktile(ic<7>{}, pr);
        ktile(ic<8>{}, pr);
      }
      if (wm == 0) __builtin_amdgcn_s_barrier();       // barrier counts of the two groups match again
      wait_vm<0>();                                    // tail DMAs (zeros) must not land in the epilogue's LDS
      cur = OFF_B0;
      oth = OFF_B0 + BB;
    }
    __syncthreads();

    const int cbm = bm, cbn = bn;
    const int next = tile + G;
    const bool has_next = next < geo.ntiles;

    // ---- epilogue (saspa_gemm_pp.hip's, one image per tile) ----
    int etid = threadIdx.x;
    asm volatile("" : "+v"(etid));
    const int elane = etid & 63;
    const int ewave = __builtin_amdgcn_readfirstlane(etid >> 6);
    const int ewm = ewave >> 2, ewn = ewave & 3, efrow = elane & 15, efg = elane >> 4;
    T* out = reinterpret_cast<T*>(p.out);
    const T* res = p.residual ? reinterpret_cast<const T*>(p.residual) : nullptr;
    if (gridDim.y > 1) {
      float* ws = p.workspace + (long long)blockIdx.y * p.M * p.N;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int m = cbm * BM + ewm * 128 + i * 16 + efrow;
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int n = cbn * BN + ewn * (16 * FN) + j * 16 + efg * 4;
          if (n >= p.N) continue;
          const f32x4 c = j < 4 ? acc[i][j < 4 ? j : 0] : accv[i];
          *reinterpret_cast<float4*>(ws + (long long)m * p.N + n) = make_float4(c[0], c[1], c[2], c[3]);
        }
      }
    } else {
      T* ct = reinterpret_cast<T*>(lds);
      T* ctg = ct + ewm * (64 * CP);
      const int gtid = etid & 255;
      const int m0 = cbm * BM + ewm * 128;
      const float* avb = addvb + ewn * (16 * FN) + efg * 4;
      const float* av = addvr + ewn * (16 * FN) + efg * 4;
      float4 add[FN];
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        add[j] = *reinterpret_cast<const float4*>(avb + j * 16);
        const float4 r4 = *reinterpret_cast<const float4*>(av + j * 16);
        add[j].x += r4.x; add[j].y += r4.y; add[j].z += r4.z; add[j].w += r4.w;
      }
      const int gunit = p.gn_stats ? p.gn_unit : BN;
      const int nunits = BN / gunit;
      const int rgs = 256 / nunits;
      const int gu = gtid % nunits, grg = gtid / nunits;
      float gsm = 0.f, gsq = 0.f;
      constexpr int CPR = BN / 8;
#pragma unroll
      for (int qt = 0; qt < 2; ++qt) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
          for (int j = 0; j < FN; ++j) {
            const f32x4 c = j < 4 ? acc[qt * 4 + i][j < 4 ? j : 0] : accv[qt * 4 + i];
            float v[4] = {c[0] + add[j].x, c[1] + add[j].y, c[2] + add[j].z, c[3] + add[j].w};
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= p.alpha;
            Elem<T>::store4(ctg + (i * 16 + efrow) * CP + ewn * (16 * FN) + j * 16 + efg * 4, v);
          }
        }
        lds_barrier();
        const int mq = m0 + qt * 64;
        auto finish = [&](u32x4 c4, const uint4 r4, const int row, const int ch, const int m, const int n) __attribute__((always_inline)) {
          if (res || p.act != SASPA_ACT_NONE) {
            float a[8];
            unpack8(__builtin_bit_cast(uint4, c4), a);
            if (p.act == SASPA_ACT_SILU) {
#pragma unroll
              for (int e = 0; e < 8; ++e) a[e] = a[e] * __builtin_amdgcn_rcpf(1.0f + __expf(-a[e]));
            }
            if (res) {
              float b[8];
              unpack8(r4, b);
#pragma unroll
              for (int e = 0; e < 8; ++e) a[e] += b[e];
            }
            c4 = __builtin_bit_cast(u32x4, pack8(a));
            if (p.gn_stats) *reinterpret_cast<u32x4*>(ctg + row * CP + ch * 8) = c4;   // the statistics read the STORED values
          }
          *reinterpret_cast<u32x4*>(out + (long long)m * p.ldo + n) = c4;
        };
        constexpr int NIT = 64 * CPR / 256;
        static_assert(NIT * 256 == 64 * CPR && NIT % 2 == 0, "store pass: whole rounds of 256 chunks");
        if (cbn * BN + BN <= p.N) {
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
            if (n >= p.N) continue;
            const u32x4 c4 = *reinterpret_cast<const u32x4*>(ctg + row * CP + ch * 8);
            uint4 r4 = make_uint4(0u, 0u, 0u, 0u);
            if (res) r4 = *reinterpret_cast<const uint4*>(res + (long long)m * p.ldr + n);
            finish(c4, r4, row, ch, m, n);
          }
        }
        if (p.gn_stats) {
          lds_barrier();
          if (grg < rgs) {
            for (int r = grg; r < 64; r += rgs) {
              const uint32_t* src = reinterpret_cast<const uint32_t*>(ctg + r * CP + gu * gunit);
              for (int j = 0; j < gunit; j += 2) {
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
        float* scr = reinterpret_cast<float*>(ct + 128 * CP) + ewm * 512;
        if (grg < rgs) {
          scr[(grg * nunits + gu) * 2] = gsm;
          scr[(grg * nunits + gu) * 2 + 1] = gsq;
        }
        lds_barrier();
        if (gtid < nunits * 2) {
          const int uu = gtid >> 1, k = gtid & 1;
          float a = 0.f;
          for (int gq = 0; gq < rgs; ++gq) a += scr[(gq * nunits + uu) * 2 + k];
          p.gn_stats[((long long)(cbm * 2 + ewm) * (p.N / p.gn_unit) + (cbn * BN) / p.gn_unit + uu) * 2 + k] = a;
        }
        lds_barrier();
      }
    }
    if (!has_next) break;
    tile = next;
    setup_tile(tile);
  }
}

int halo_rows_max(int H, int W) {
  // longest linear range of padded pixels a 256-pixel tile (aligned to 256 inside the image) can need
  int best = 0;
  const int wp = W + 2;
  for (int rem0 = 0; rem0 < H * W; rem0 += 256) {
    const int oy0 = rem0 / W, ox0 = rem0 % W, oyl = (rem0 + 255) / W, oxl = (rem0 + 255) % W;
    const int L = (oyl + 1) * wp + oxl + 1 - ((oy0 + 1) * wp + ox0 + 1) + 2 * wp + 3;
    if (L > best) best = L;
  }
  return best;
}

}  // namespace

extern "C" int saspa_conv3x3_halo_eligible(const SaspaGemmParams* pp, const SaspaConvGnParams* gp) {
  if (!pp) return 0;
  const SaspaGemmParams& p = *pp;
  if (p.dtype != SASPA_BF16 || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.pad != 1 || p.upsample) return 0;
  if (p.hin != p.hout || p.win != p.wout || p.batch <= 0) return 0;
  const int ctot = p.c0 + p.c1;
  if (ctot % 64 || p.c0 % 32 || p.c1 % 32 || p.c0 <= 0) return 0;
  if (p.N % 320 && p.N % 256) return 0;
  if (p.korder != SASPA_KORDER_CHUNK32 || p.K != 9 * ctot || p.ldw < p.K || p.ldw % 8) return 0;
  const int hw = p.hout * p.wout;
  if (hw % 256 || hw > 16000 || (long long)p.batch * hw != p.M) return 0;
  if (halo_rows_max(p.hout, p.wout) > 480 || p.wout < 9) return 0;
  if (p.act != SASPA_ACT_NONE && p.act != SASPA_ACT_SILU) return 0;
  if ((long long)p.nb1 * p.nb2 > 1 || p.ln_gamma || p.out_t) return 0;
  if ((p.ldo % 8) || (p.residual && (p.ldr % 8)) || p.lda0 % 8 || (p.c1 > 0 && p.lda1 % 8)) return 0;
  if ((long long)hw * p.lda0 * 2 >= (1ll << 31) || (long long)hw * p.lda1 * 2 >= (1ll << 31)) return 0;      // 32-bit per-lane byte offsets
  if ((long long)p.lda0 * 2 >= (1ll << 24) || (long long)p.lda1 * 2 >= (1ll << 24)) return 0;
  if (p.gn_stats && (p.N % 160 || 80 % p.gn_unit || p.gn_unit % 2 || p.gn_unit > 16 || p.gn_unit <= 0)) return 0;
  if (gp) {
    const SaspaConvGnParams& g = *gp;
    if (!g.gamma_beta32 || g.groups <= 0 || g.groups > 64 || ctot % g.groups) return 0;
    if (g.act != SASPA_ACT_NONE && g.act != SASPA_ACT_SILU) return 0;
    if (g.stats0) {
      const int cpg = ctot / g.groups;
      if (g.unit <= 0 || hw % 128 || p.c0 % g.unit || p.c1 % g.unit || cpg % g.unit || (p.c1 > 0 && !g.stats1)) return 0;
    } else if (!g.partial || g.nsplit <= 0) {
      return 0;
    }
  }
  return 1;
}

extern "C" int saspa_conv3x3_halo(const SaspaGemmParams* pp, const SaspaConvGnParams* gp, void* stream) {
  if (!pp) return SASPA_EINVAL;
  const SaspaGemmParams& p = *pp;
  if (!p.a0 || !p.w || !p.out || (p.c1 > 0 && !p.a1)) return SASPA_EINVAL;
  if (!aligned16(p.a0) || !aligned16(p.w) || !aligned16(p.out) || (p.a1 && !aligned16(p.a1)) || (p.residual && !aligned16(p.residual)) ||
      (p.bias && !aligned16(p.bias)) || (p.rowvec && !aligned16(p.rowvec)) || (gp && gp->gamma_beta32 && !aligned16(gp->gamma_beta32)))
    return SASPA_EALIGN;
  if (!saspa_conv3x3_halo_eligible(pp, gp)) return SASPA_ERANGE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int fn = (p.N % 320 == 0) ? 5 : 4;
  const int BN = 64 * fn;
  const int tiles = ((p.N + BN - 1) / BN) * (p.M / 256);
  HaloGeom geo;
  geo.wp = p.wout + 2;
  geo.nper = (p.c0 + p.c1) / 64;
  int ks = (p.workspace && p.ksplit > 1 && p.N % 4 == 0) ? p.ksplit : 1;
  if (ks > geo.nper) ks = geo.nper;
  geo.per_slice = (geo.nper + ks - 1) / ks;
  ks = (geo.nper + geo.per_slice - 1) / geo.per_slice;           // no empty slices: the reduce sums exactly `ks` slabs
  // ABI 19 contract of `defer_reduce`, as for saspa_gemm: the caller's saspa_splitk_groupnorm will sum EXACTLY p.ksplit slabs.  A
  // launch that would end on one slice (it would write `out` directly) or on fewer slices than asked (uninitialised slabs would be
  // summed) is refused; the caller sizes its request with saspa_conv3x3_halo_ksplit
  if (p.defer_reduce && (ks <= 1 || ks != p.ksplit)) return SASPA_ERANGE;
  geo.gn_slabs = gp ? saspa_gn_slabs((p.c0 + p.c1) / 8) : 1;
  geo.ntiles = tiles;
  const int gx = saspa_balanced_grid(tiles, 256 / ks);
  dim3 grid(gx, ks, 1);
  SaspaConvGnParams g0 = {};
  const SaspaConvGnParams& g = gp ? *gp : g0;
  const bool reg = (p.wout % 16) == 0;
#define HALO_LAUNCH(F, GNF, R) hipLaunchKernelGGL((conv_halo_kernel<F, GNF, R>), grid, dim3(512), 0, s, p, g, geo)
  if (fn == 5) {
    if (gp) { if (reg) HALO_LAUNCH(5, true, true); else HALO_LAUNCH(5, true, false); }
    else { if (reg) HALO_LAUNCH(5, false, true); else HALO_LAUNCH(5, false, false); }
  } else {
    if (gp) { if (reg) HALO_LAUNCH(4, true, true); else HALO_LAUNCH(4, true, false); }
    else { if (reg) HALO_LAUNCH(4, false, true); else HALO_LAUNCH(4, false, false); }
  }
#undef HALO_LAUNCH
  SASPA_CHECK_LAUNCH();
  if (ks > 1) {
    SaspaGemmParams q = p;
    q.ksplit = ks;
    return saspa_gemm_splitk_reduce(q, s, ks);
  }
  return 0;
}

/* K slices the launch will really use for a requested factor (the caller sizes the workspace with the request; the reduce reads
 * exactly this many slabs) */
extern "C" int saspa_conv3x3_halo_ksplit(const SaspaGemmParams* pp, int want) {
  if (!pp || want <= 1) return 1;
  const int nper = (pp->c0 + pp->c1) / 64;
  if (want > nper) want = nper;
  const int per = (nper + want - 1) / want;
  return (nper + per - 1) / per;
}
