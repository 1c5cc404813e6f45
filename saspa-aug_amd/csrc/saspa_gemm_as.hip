// A-stationary bf16 GEMM for the pointwise layers of the level-0 transformer blocks (K = 320 channels, M = batch x 4096+
// tokens): to_q | to_k | to_v, the cross-attention to_q, the GEGLU projection, to_out + residual, proj_in.
//
// Why another kernel.  With K = 320 a 128x160 or 256x320 output tile is 5 K-tiles long: the tiled kernels spend their time in
// prologues, epilogues and in re-reading the same 128 / 256 rows of A for every column tile (the GEGLU projection has 16 / 8
// of them), and a LayerNorm in front costs a separate 2 x M x K byte pass.  Here a wave keeps its 32 rows of A -- all 320
// channels, 20 MFMA B-operand fragments = 80 VGPRs -- in registers for the whole launch:
//   * A is read from HBM exactly once, by the lanes that use it, in MFMA operand layout (no LDS round trip);
//   * LayerNorm (SaspaGemmParams.ln_gamma) is applied to those registers in place: a row lives in two lanes, so the statistics
//     are 160 in-lane adds and one cross-lane swap;
//   * W streams through a 3-slot LDS ring in slices of 64 output columns (LDS-DMA, 40 KB + 64 biases per slice), shared by the 8 waves of
//     the workgroup: one barrier per slice, the next-but-one slice in flight;
//   * the product is taken transposed (W K^T-style: D = W_slice X^T, v_mfma_f32_32x32x16_bf16), so a lane ends up with 4
//     consecutive output columns of ONE row per accumulator quad: bias is the accumulators' initial value, the tile goes
//     through a wave-private LDS tile to 16-byte row-major stores, and the V^T operand of the attention kernel is a
//     2-byte-per-lane store with 32 consecutive tokens per channel (SaspaGemmParams.out_t) -- Q | K | V^T leave one launch;
//   * the epilogue of slice t-1 and the DMA of slice t+2 are dealt between the MFMA groups of slice t (one wave, two streams).
// Roofline of a slice per workgroup: 320 MFMAs (8 waves x 40) = 2 560 cycles of every SIMD's matrix pipe.  Measured with in-kernel
// stamps (tools/as_stamps.py): a step takes 5 150 cycles on the plain layers and 6 470 with the GELU -- the MFMA groups with the
// previous slice's epilogue and the next-but-one slice's DMA pieces dealt between them take 4 200-4 500 (each vector-memory
// instruction costs the wave 100-200 issue cycles, ten of them per step), the DMA wait 300-400, the barrier skew the rest:
// 40-50 % of the matrix-pipe bound, i.e. the level of the 8-wave conv kernel, on layers where the tiled kernels reach 20-25 %.
#include <cstdlib>

#include "common.h"
#include "gemm_internal.h"

namespace {

constexpr int AS_K = 320;
constexpr int AS_KS = AS_K / 16;          // MFMA K-steps
constexpr int AS_BM = 256;                // rows per workgroup: 8 waves x 32
constexpr int AS_BN = 64;                 // W rows (output columns) per ring slot
constexpr int AS_PITCH = 40;              // 16-byte chunks per W row in LDS; chunk kc of row n sits at kc ^ ((n >> 1) & 7): conflict-free
                                          // ds_read_b128 (rows 2 i and 2 i + 1 differ by 8 chunks = half the banks, the XOR spreads
                                          // the pairs) AND every aligned lane quad of the DMA reads one aligned 64-byte segment
constexpr int AS_NDMA = 5;                // wave-wide LDS-DMA instructions per wave per slice: 8 * 5 * 64 = 2 560 chunks = 64 rows x 40
constexpr int AS_STAGE = 8 * AS_NDMA * 64 + 64;   // chunks per ring slot: the slice + one more wave-wide DMA for its 64 biases (wave 7)
constexpr int AS_RING = 3;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// vmcnt <= n with n a run-time (wave-uniform) value: the instruction takes an immediate; 63 is the counter's maximum
__device__ __forceinline__ void wait_vm_dyn(int n_any) {
  // the count is wave-uniform by construction; say so, or the switch becomes a tree of exec-mask branches on a VGPR (measured:
  // ~750 cycles per step on tools/as_stamps.py)
  const int n = __builtin_amdgcn_readfirstlane(n_any);
#define SASPA_W1(i) case i: wait_vm<i>(); break;
#define SASPA_W8(i) SASPA_W1(i) SASPA_W1(i + 1) SASPA_W1(i + 2) SASPA_W1(i + 3) SASPA_W1(i + 4) SASPA_W1(i + 5) SASPA_W1(i + 6) SASPA_W1(i + 7)
  switch (n < 63 ? n : 63) {
    SASPA_W8(0) SASPA_W8(8) SASPA_W8(16) SASPA_W8(24) SASPA_W8(32) SASPA_W8(40) SASPA_W8(48) SASPA_W8(56)
    default: wait_vm<0>(); break;
  }
#undef SASPA_W8
#undef SASPA_W1
}

// bf16 x 8 -> fp32 x 8 of a register the compiler must treat as new each time: the three LayerNorm passes re-unpack the
// packed rows (2 VALU per pair) instead of keeping 160 unpacked floats alive across them (268 spilled VGPRs)
__device__ __forceinline__ void unpack_opaque(const u32x4& a, float* v) {
  u32x4 t = a;
  asm volatile("" : "+v"(t));
  unpack8(__builtin_bit_cast(uint4, t), v);
}

// Output staging: a lane's accumulator quads are 4 columns of ONE row each, 32 different rows per store instruction -- 64
// separate 8-byte writes, which cost more than the slice's MFMAs (tools/as_ablate.py).  Each wave transposes its 32 x 64 (GEGLU:
// 32 x 32) tile through a private LDS tile (row pitch 136 / 72 bytes: 34 / 18 banks, conflict-free 8-byte writes) and stores
// 16 bytes per lane with 8 (4) consecutive lanes on one row: whole 128 (64) byte segments.  Wave-private, so no barrier.
constexpr int AS_STG_PITCH = 136;
constexpr int AS_STG_WAVE = 32 * AS_STG_PITCH;                 // 4 352 bytes per wave
constexpr int AS_STG_CHUNKS = 8 * AS_STG_WAVE / 16;            // 2 176 chunks: ring 125 952 B + staging 34 816 B = 160 768 B of 163 840

constexpr int AS_BIAS_SLOT = 8 * AS_NDMA * 64;        // wave 7's extra DMA instruction of a slice carries its 64 biases

// EPI: 0 = plain (bias, optional transposed columns), 1 = fused GEGLU (W packed per `gb`-column tile: values then gates);
// RES: a residual is added (plain only).  Its loads are older than the slice's own stores but younger than the previous
// slice's: waiting for them (in-order vmcnt) waits for those stores too, so the no-residual layers get their own instance.
// GB: the GEGLU packing tile (160 / 128 raw columns: values then gates; 0 for plain) -- compile-time, the row mapping divides by it.
template <int EPI, bool RES, int GB>
__global__ __launch_bounds__(512, 1) void gemm_as_kernel(const SaspaGemmParams p, const int abl_arg) {
  // diagnostics (tools/as_ablate.py builds one library per value: a run-time switch would change the loop it measures):
  // 1 no W fragment reads, 2 no MFMA, 4 no epilogue, 8 no DMA, 16 no barrier, 32 stores out of range (issued, no bytes move)
#ifdef SASPA_AS_ABLATE
  constexpr int abl = SASPA_AS_ABLATE;
#else
  constexpr int abl = 0;
#endif
  // diagnostics (-DSASPA_AS_STAMPS, tools/as_stamps.py): s_memtime of workgroup 0's waves 0 and 7 at four points of every step,
  // kept in the last 2.5 KB of LDS (a global store per stamp would be a vector-memory operation the vmcnt arithmetic does not
  // count) and copied to SaspaGemmParams.workspace (unused by this kernel otherwise) at the end: [wave 0 | wave 7][step][4]
#ifdef SASPA_AS_STAMPS
  __shared__ unsigned long long stamp_lds[2 * 40 * 4];
  const bool stamping = blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 448) && p.workspace;
  unsigned long long* stamps = stamp_lds + (threadIdx.x ? 160 : 0);
#define AS_STAMP(step_, k_) do { if (stamping && (step_) < 40) stamps[(step_) * 4 + (k_)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AS_STAMP(step_, k_) do { } while (0)
#endif
  __shared__ u32x4 lds[AS_RING * AS_STAGE + AS_STG_CHUNKS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  // ---- work partition (round 5): the (row block, column slice) steps of the whole problem, block-major, are dealt to the
  // workgroups in CONTIGUOUS shares of ceil(total / grid) steps.  A share that starts or ends inside a row block is a "run" of
  // that block's slices: slices are independent outputs, so nothing has to be combined.  With 256 row blocks (512x512) every
  // workgroup gets exactly one whole block -- the launch of rounds 3 - 4; with 352 (512x704: 1.375 rounds, which used to
  // disqualify the kernel) every workgroup gets 1.375 blocks' worth in two or three runs, each reloading its rows.
  const int nslices_all = p.N / AS_BN;
  const long long total_steps = (long long)((p.M + AS_BM - 1) / AS_BM) * nslices_all;
  const long long share = (total_steps + gridDim.x - 1) / gridDim.x;
  long long pos = (long long)blockIdx.x * share;
  const long long pend = pos + share < total_steps ? pos + share : total_steps;
  while (pos < pend) {
  const int blk = (int)(pos / nslices_all);
  const int s_first = (int)(pos - (long long)blk * nslices_all);
  const int nslices = (int)((pend - pos) < (long long)(nslices_all - s_first) ? (pend - pos) : (long long)(nslices_all - s_first));   // slices of this run
  const long long row = (long long)blk * AS_BM + wave * 32 + m;
  const bool live = row < p.M;

  // ---- this lane's half of row `row`: channels 16 s + 8 h .. + 8, s = 0 .. 19 ----
  const rsrc_t rsa = make_rsrc(p.a0);
  const unsigned aoff = live ? (unsigned)(row * p.lda0 * 2 + h * 16) : kInvalid;
  u32x4 af[AS_KS];
#pragma unroll
  for (int s = 0; s < AS_KS; ++s) af[s] = buf_load(rsa, aoff, s * 32);

  // ---- W slices: LDS chunk q = 64 j + lane of a slice <- W row q / 40 of the slice, 16-byte chunk (q % 40) ^ ((row >> 1) & 7) ----
  const rsrc_t rsw = make_rsrc(p.w);
  const rsrc_t rsb = make_rsrc(p.bias);
  const bool has_bias = p.bias != nullptr;
  // A run that covers its whole block walks the slices in a rotated order (its own starting slice; workgroups of one XCD --
  // blockIdx = xcd + 8 k -- get consecutive starts), so that the 256 workgroups do not all ask the L2 for the same 40 KB in the
  // same microsecond.  (Measured neutral on its own -- the DMA alone runs at 10.8 TB/s either way, tools/as_ablate.py -- kept:
  // it costs nothing.)  Partial runs start where their share starts, which staggers them by itself.
  const int rot = nslices == nslices_all ? (int)((blockIdx.x >> 3) % (unsigned)nslices) : 0;
  auto slice_of = [&](int i) __attribute__((always_inline)) -> int {      // i-th slice of this run
    const int u = i + rot;
    return s_first + (u >= nslices ? u - nslices : u);
  };
  int dn[AS_NDMA], dkc[AS_NDMA];
#pragma unroll
  for (int i = 0; i < AS_NDMA; ++i) {
    const int q = (wave * AS_NDMA + i) * 64 + lane;      // LDS chunk this lane fills
    const int n = q / AS_PITCH, kcp = q - n * AS_PITCH;
    dn[i] = n;
    dkc[i] = kcp ^ ((n >> 1) & 7);                       // the W chunk that lives there
  }
  const int ndma = AS_NDMA + (wave == 7 ? 1 : 0);        // vector-memory instructions dma_slice issues on this wave
  constexpr int gb = GB, half = GB > 0 ? GB / 2 : 1;
  // W row (in the packed matrix) of slice row n of slice t
  auto wrow = [&](int t, int n) __attribute__((always_inline)) -> int {
    if (EPI == 0) return t * AS_BN + n;
    const int f = t * 32 + (n & 31);                 // feature: value row in block 0, its gate in block 1
    const int tile = f / half;
    return tile * gb + (f - tile * half) + ((n >> 5) ? half : 0);
  };
  // ALWAYS `ndma` wave-wide DMA instructions per slice (the loop's vmcnt arithmetic counts them): a slice past the end loads
  // zeros.  Piece i < AS_NDMA: 64 chunks of W; piece AS_NDMA (wave 7 only): the slice's 64 biases (fp32, 16 lanes x 16 bytes).
  auto dma_piece = [&](int step, int i) __attribute__((always_inline)) {
    const bool real = step < nslices;
    const int t = real ? slice_of(step) : 0;
    if (abl & 8) return;
    if (i < AS_NDMA) {
      const unsigned off = real ? (unsigned)(wrow(t, dn[i]) * p.ldw * 2 + dkc[i] * 16) : kInvalid;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void_t*)(lds + (step % AS_RING) * AS_STAGE + (wave * AS_NDMA + i) * 64), 16, (int)off, 0, 0, 0);
    } else if (wave == 7) {
      const unsigned off = (real && has_bias && lane < 16) ? (unsigned)(wrow(t, 4 * lane) * 4) : kInvalid;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb, (lds_void_t*)(lds + (step % AS_RING) * AS_STAGE + AS_BIAS_SLOT), 16, (int)off, 0, 0, 0);
    }
  };
  auto dma_slice = [&](int step) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i <= AS_NDMA; ++i) dma_piece(step, i);
  };
  dma_slice(0);
  dma_slice(1);

  // ---- LayerNorm of the row, in registers (the arithmetic of layernorm_kernel: two-pass, fp32, bf16 result) ----
  if (p.ln_gamma) {
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < AS_KS; ++s) {
      float v[8];
      unpack_opaque(af[s], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[j];
    }
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum / (float)AS_K;
    float sq = 0.f;
#pragma unroll
    for (int s = 0; s < AS_KS; ++s) {
      float v[8];
      unpack_opaque(af[s], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[j] - mean; sq += d * d; }
    }
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = 1.0f / sqrtf(sq / (float)AS_K + p.ln_eps);
#pragma unroll
    for (int s = 0; s < AS_KS; ++s) {
      const int k0 = 16 * s + 8 * h;
      float v[8];
      unpack_opaque(af[s], v);
      const float4 g0 = *reinterpret_cast<const float4*>(p.ln_gamma + k0), g1 = *reinterpret_cast<const float4*>(p.ln_gamma + k0 + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(p.ln_beta + k0), b1 = *reinterpret_cast<const float4*>(p.ln_beta + k0 + 4);
      const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (v[j] - mean) * rstd * g[j] + bb[j];
      af[s] = __builtin_bit_cast(u32x4, pack8(v));
      if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // four K-steps of gamma / beta in flight, not all twenty
    }
  }

  const rsrc_t rsr = make_rsrc(p.residual);
  const rsrc_t rso = make_rsrc(p.out);
  const rsrc_t rst = make_rsrc(p.out_t);
  const bool transposed_tail = EPI == 0 && p.out_t != nullptr;
  // fragment (K-step s, column block nb) of this lane: row n = 32 nb + m, chunk 2 s + h, stored at chunk ^ ((n >> 1) & 7);
  // 32 nb does not change (n >> 1) & 7, so one per-lane XOR key serves both blocks
  const unsigned char* fbase = reinterpret_cast<const unsigned char*>(lds) + m * AS_PITCH * 16;
  const int fkey = (m >> 1) & 7;
  // (2 s + h) ^ fkey only touches the low three bits: four per-lane byte offsets (s & 3), the rest is an immediate
  int foff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) foff[j] = (((2 * j) | h) ^ fkey) << 4;
  // stores per wave of slice u (the vmcnt arithmetic below): 2 GEGLU, 4 plain (16 bytes per lane), 32 transposed (2 bytes)
  auto nstores = [&](int u) __attribute__((always_inline)) -> int {
    return EPI == 1 ? 2 : ((transposed_tail && u * AS_BN >= p.n_split) ? 32 : 4);
  };
  unsigned trow = kInvalid;
  if (transposed_tail && live) {
    const long long bi = row / p.rows_per_batch;
    trow = (unsigned)((bi * p.st + (row - bi * p.rows_per_batch)) * 2);
  }
  // store layout of the staged tile: instruction i covers rows RPI * i + lane / LPR, 16-byte chunk lane % LPR of the row
  constexpr int LPR = EPI == 1 ? 4 : 8, RPI = 64 / LPR, NST = 32 / RPI, SPITCH = EPI == 1 ? 72 : AS_STG_PITCH;
  unsigned char* stg = reinterpret_cast<unsigned char*>(lds + AS_RING * AS_STAGE) + wave * AS_STG_WAVE;
  const long long row0 = (long long)blk * AS_BM + wave * 32;
  unsigned so_off[NST], sr_off[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const long long r = row0 + RPI * i + lane / LPR;
    so_off[i] = (r < p.M && !(abl & 32)) ? (unsigned)(r * p.ldo * 2 + (lane % LPR) * 16) : kInvalid;   // abl 32: stores issued, all out of range
    sr_off[i] = (RES && r < p.M) ? (unsigned)(r * p.ldr * 2 + (lane % LPR) * 16) : kInvalid;
  }
  const unsigned char* stg_rd = stg + (lane / LPR) * SPITCH + (lane % LPR) * 16;
  unsigned char* stg_wr = stg + m * SPITCH + h * 8;

  // ---- a step: the slice's MFMAs with the PREVIOUS slice's epilogue dealt into the gaps between them ----
  // One wave carries two instruction streams (as the attention kernel does): the matrix pipe runs slice t while the VALU / LDS /
  // store work of slice t-1 (GELU, packing, the staging round trip, the stores) is issued between its MFMA groups -- with the
  // epilogue as a separate phase after the MFMAs the loop measured the SUM of the two (tools/as_ablate.py), because the
  // per-slice barrier puts all eight waves in the same phase.  What slice t-1 leaves behind for that: its 32 x 64 tile packed
  // to bf16 (16 registers; GEGLU: the two fp32 accumulator blocks, 32 registers).
  // mma(step, acc, between): bias (delivered with the slice) as the accumulators' initial value, then 40 MFMAs in groups of 4
  // (two K-steps x two column blocks) with the W fragments of the next TWO groups in flight; between(gi) runs after group gi's
  // MFMAs; the sched_barrier pins the deal and keeps the compiler from hoisting all 40 fragment reads (160 VGPRs) to the top.
  constexpr int NG = AS_KS / 2;
  auto mma = [&](int step, f32x16 (&acc)[2], auto&& between) __attribute__((always_inline)) {
    const unsigned char* fs = fbase + (step % AS_RING) * (AS_STAGE * 16);
    const unsigned char* bs = reinterpret_cast<const unsigned char*>(lds) + ((step % AS_RING) * AS_STAGE + AS_BIAS_SLOT) * 16;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bs + (nb * 8 + 2 * g + h) * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[nb][4 * g + j] = bv[j];
      }
    auto frag = [&](int s, int nb) __attribute__((always_inline)) -> u32x4 {
      if (abl & 1) return af[(s + nb) % AS_KS];
      return *reinterpret_cast<const u32x4*>(fs + foff[s & 3] + (nb * 32 * AS_PITCH * 16 + ((2 * s) & ~7) * 16));
    };
    u32x4 wf[3][4];
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int j = 0; j < 4; ++j) wf[gi][j] = frag(2 * gi + (j >> 1), j & 1);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi + 2 < NG) {
#pragma unroll
        for (int j = 0; j < 4; ++j) wf[(gi + 2) % 3][j] = frag(2 * (gi + 2) + (j >> 1), j & 1);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int sx = 2 * gi + (j >> 1), nb = j & 1;
        if (abl & 2) asm volatile("" ::"v"(wf[gi % 3][j]), "v"(af[sx]));
        else acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wf[gi % 3][j]), __builtin_bit_cast(bf16x8, af[sx]), acc[nb], 0, 0, 0);
      }
      between(gi);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // What a finished slice hands to the next step
  struct Done {
    u32x2 pk[8];          // plain: quad (nb, g) of the tile as 4 bf16 (columns 32 nb + 8 g + 4 h ..+4 of row m)
    f32x16 sv[2];         // GEGLU: value block, gate block
  };
  auto finish = [&](f32x16 (&acc)[2], Done& d) __attribute__((always_inline)) {
    if (abl & 4) {               // (no epilogue: keep the MFMAs alive)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) asm volatile("" ::"v"(acc[nb]));
      return;
    }
    if (EPI == 1) {
      d.sv[0] = acc[0];
      d.sv[1] = acc[1];
    } else {
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) d.pk[nb * 4 + g] = u32x2{pack2(acc[nb][4 * g], acc[nb][4 * g + 1]), pack2(acc[nb][4 * g + 2], acc[nb][4 * g + 3])};
    }
  };
  // epilogue_part(gi, t, d, rv): the share of slice t's epilogue that goes behind MFMA group gi of the next slice (all ten
  // shares together issue exactly nstores(t) vector-memory stores).  Plain: groups 0-3 write the eight quads to the staging
  // tile, groups 5-8 read a 16-byte row chunk back, add the residual and store (two roundings with a residual, as Linear -> add
  // in the reference and as the tiled kernels).  Transposed tail (V^T): four 2-byte stores per quad, groups 0-7.  GEGLU: one
  // quad of value * gelu(gate) per even group 0-6, the two stores behind groups 8 and 9.
  auto epilogue_part = [&](int gi, int t, const Done& d, const u32x4 (&rv)[NST]) __attribute__((always_inline)) {
    if (abl & 4) return;
    if (EPI == 1) {
      if (gi < 8 && (gi & 1) == 0) {
        const int g = gi >> 1;
        const u32x2 o = {pack2(fast_gelu_mul(d.sv[0][4 * g], d.sv[1][4 * g]), fast_gelu_mul(d.sv[0][4 * g + 1], d.sv[1][4 * g + 1])),
                         pack2(fast_gelu_mul(d.sv[0][4 * g + 2], d.sv[1][4 * g + 2]), fast_gelu_mul(d.sv[0][4 * g + 3], d.sv[1][4 * g + 3]))};
        *reinterpret_cast<u32x2*>(stg_wr + g * 16) = o;
      }
    } else if (transposed_tail && t * AS_BN >= p.n_split) {
      if (gi < 8) {
        const int nb = gi >> 2, g = gi & 3;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = t * AS_BN - p.n_split + nb * 32 + 8 * g + 4 * h + j;
          const unsigned w32 = (j & 2) ? d.pk[gi].y : d.pk[gi].x;
          const unsigned short bits = (unsigned short)((j & 1) ? (w32 >> 16) : (w32 & 0xffffu));
          __builtin_amdgcn_raw_buffer_store_b16(bits, rst, (int)(trow == kInvalid ? kInvalid : trow + (unsigned)(c * p.ldt * 2)), 0, 0);
        }
      }
      return;
    } else if (gi < 4) {
#pragma unroll
      for (int q = 2 * gi; q < 2 * gi + 2; ++q) *reinterpret_cast<u32x2*>(stg_wr + (q >> 2) * 64 + (q & 3) * 16) = d.pk[q];
    }
    const int i = EPI == 1 ? gi - 8 : gi - 5;
    if (i >= 0 && i < NST) {
      const int cbytes = EPI == 1 ? t * 32 * 2 : t * AS_BN * 2;
      const u32x2 lo = *reinterpret_cast<const u32x2*>(stg_rd + i * RPI * SPITCH), hi = *reinterpret_cast<const u32x2*>(stg_rd + i * RPI * SPITCH + 8);
      u32x4 o = {lo.x, lo.y, hi.x, hi.y};
      if (RES) {
        float a[8], r8[8];
        unpack8(__builtin_bit_cast(uint4, o), a);
        unpack8(__builtin_bit_cast(uint4, rv[i]), r8);
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += r8[j];
        o = __builtin_bit_cast(u32x4, pack8(a));
      }
      __builtin_amdgcn_raw_buffer_store_b128(o, rso, (int)(so_off[i] == kInvalid ? kInvalid : so_off[i] + cbytes), 0, 0);
    }
  };

  // ---- vmcnt bookkeeping ----
  // This wave's share of a slice is in LDS once at most the vector-memory operations it issued AFTER that slice's DMA are
  // outstanding (vector memory returns in order).  Every such operation below is issued unconditionally (out-of-range lanes /
  // slices use the invalid offset), so a running count is exact: `issued` counts them, mark[k] is its value right after the
  // DMA into ring slot k.  The prologue's wait leaves at most slice 1's DMA (the last thing issued) in flight.
  // (With a LayerNorm its gamma / beta loads -- younger than both DMAs -- were already waited for, which implies the same.)
  wait_vm_dyn(ndma);
  int issued = 0;
  // marks of the slices of this step, the next one and the one after (shifted at the end of every step: no dynamic indexing)
  int mark0 = -64, mark1 = 0, mark2 = 0;
  auto wait_slice = [&](int step) __attribute__((always_inline)) {
#ifndef SASPA_AS_NOWAIT      // (diagnostics: how long is the wait itself?  results are garbage without it)
    if (step > 0) wait_vm_dyn(issued - mark0);
#endif
  };
  auto load_res = [&](int t, u32x4 (&rv)[NST]) __attribute__((always_inline)) {
    if (RES) {
#pragma unroll
      for (int i = 0; i < NST; ++i) rv[i] = buf_load(rsr, sr_off[i] == kInvalid ? kInvalid : sr_off[i] + t * AS_BN * 2, 0);
      issued += NST;
    }
  };

  // One barrier per step, every wave the same program.  (Built and dropped: two wave groups half a step apart -- waves 0-3 in a
  // slice's MFMAs while their SIMD partners 4-7 run the previous slice's epilogue, two barriers per step.  Bit-identical, and
  // slower: the extra barrier and the half-empty phases cost more than the overlap returned; profiles/r3_as_bench.txt.)
  // the second-dispatched half of the workgroup loses every VALU arbitration against its SIMD partners by age (it arrived
  // 1 000-1 700 cycles late at every barrier, tools/as_stamps.py): static priority for it, no per-phase flips
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
  f32x16 acc[2];
  Done done;
  u32x4 rv[NST] = {};
  // step 0: nothing to finish yet
  if (!(abl & 16)) __builtin_amdgcn_s_barrier();      // publishes slice 0 (the wait above covered this wave's share)
  dma_slice(2);
  issued += ndma;
  mark2 = issued;
  mma(0, acc, [](int) {});
  finish(acc, done);
  mark0 = mark1;
  mark1 = mark2;
  for (int step = 1; step < nslices; ++step) {
    const int tp = slice_of(step - 1);
    AS_STAMP(step, 0);
    wait_slice(step);
    AS_STAMP(step, 1);
    // publishes this step's slice to the other waves; also: every wave is done reading the previous one, whose ring slot the
    // DMA below overwrites
    if (!(abl & 16)) __builtin_amdgcn_s_barrier();
    AS_STAMP(step, 2);
    load_res(tp, rv);                   // first: consumed behind MFMA groups 5-8, the compiler's own wait leaves the DMA in flight
    AS_STAMP(step, 3);
    // the slice after next: one DMA piece behind each of the first MFMA groups (a piece costs 100-200 issue cycles: six of them
    // at the top of the step kept the matrix pipe idle for 600-1 200 cycles).  The residual loads above stay older than every
    // piece; the previous slice's stores are issued from group 5 on (after the last piece), except the transposed tail's
    mma(step, acc, [&](int gi) __attribute__((always_inline)) {
      if (gi <= AS_NDMA) dma_piece(step + 2, gi);
      epilogue_part(gi, tp, done, rv);
    });
    // bookkeeping in issue order: a transposed slice's 2-byte stores (four per group from group 0) interleave with the pieces,
    // so those of the groups before the last piece (group ndma - 1) are OLDER than it; every other kind of slice stores from
    // group 5 on, after the last piece
    {
      const bool tr = EPI == 0 && transposed_tail && tp * AS_BN >= p.n_split;
      const int older = tr ? 4 * (ndma - 1) : 0;
      issued += ndma + older;
      mark2 = issued;
      issued += nstores(tp) - older;
    }
    finish(acc, done);
    mark0 = mark1;
    mark1 = mark2;
  }
  {
    const int tl = slice_of(nslices - 1);
    load_res(tl, rv);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) epilogue_part(gi, tl, done, rv);
  }
  pos += nslices;
  if (pos < pend) {
    // another run follows: its DMA reuses the ring, its epilogue the staging tile and the bias slots -- drain this run's
    // vector-memory operations (tail DMAs of slices past the end write zeros into the ring) and let every wave finish reading
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  }   // runs of this workgroup
#ifdef SASPA_AS_STAMPS
  if (stamping)
    for (int i = 0; i < 160; ++i) (reinterpret_cast<unsigned long long*>(p.workspace) + (threadIdx.x ? 4096 : 0))[i] = stamps[i];
#endif
}

}  // namespace

bool saspa_gemm_as_ok(const SaspaGemmParams& p) {
  static const bool on = !(getenv("SASPA_GEMM_AS") && atoi(getenv("SASPA_GEMM_AS")) == 0);      // A/B knob
  static const int min_blocks = getenv("SASPA_GEMM_AS_MINBLOCKS") ? atoi(getenv("SASPA_GEMM_AS_MINBLOCKS")) : 192;
  if (!on || p.dtype != SASPA_BF16) return false;
  if (p.kh != 1 || p.kw != 1 || p.stride != 1 || p.pad != 0 || p.upsample || p.c1 != 0 || p.a1) return false;
  if (p.c0 != AS_K || p.K != AS_K || p.lda0 % 8 || p.ldw % 8 || p.ldw < AS_K) return false;
  if ((long long)p.nb1 * p.nb2 > 1 || (p.ksplit > 1 && p.workspace) || p.gn_stats || p.rowvec || p.alpha != 1.0f) return false;
  if (p.act != SASPA_ACT_NONE && p.act != SASPA_ACT_GEGLU) return false;
  if (p.N % AS_BN || p.N <= 0 || (p.M + AS_BM - 1) / AS_BM < min_blocks) return false;
  if ((long long)p.M * (p.lda0 > p.ldo ? p.lda0 : p.ldo) * 2 >= 0x7fffffffLL) return false;   // 32-bit buffer offsets
  if (p.act == SASPA_ACT_GEGLU) {
    const int gb = (p.N % 160 == 0) ? 160 : 128;
    if (p.N % gb || p.residual || p.out_t || p.ldo % 8) return false;
  } else {
    if (p.ldo % 8 || (p.residual && (p.ldr % 8 || (long long)p.M * p.ldr * 2 >= 0x7fffffffLL))) return false;
    if (p.out_t && (p.n_split % AS_BN || p.n_split < 0 || p.n_split > p.N || p.rows_per_batch <= 0 || p.rows_per_batch % 32 ||
                    p.M % p.rows_per_batch || p.ldt < p.rows_per_batch ||
                    ((long long)(p.M / p.rows_per_batch - 1) * p.st + (long long)(p.N - p.n_split) * p.ldt) * 2 >= 0x7fffffffLL))   // 32-bit buffer offsets
      return false;
  }
  if (!aligned16(p.a0) || !aligned16(p.w) || !aligned16(p.out) || (p.bias && !aligned16(p.bias)) || (p.residual && !aligned16(p.residual)) || (p.ln_gamma && (!p.ln_beta || (reinterpret_cast<uintptr_t>(p.ln_gamma) & 15u) ||
                                                                                       (reinterpret_cast<uintptr_t>(p.ln_beta) & 15u))))
    return false;
  return true;
}

int saspa_gemm_as_launch(const SaspaGemmParams& p, hipStream_t s) {
  if (!saspa_gemm_as_ok(p)) return SASPA_ERANGE;
  SASPA_DRY_RETURN(SASPA_GEMM_AS, 1);
  // one workgroup per CU, the steps dealt evenly (see the kernel); SASPA_GEMM_BALANCE=0: one workgroup per row block as before
  const long long nblk = (p.M + AS_BM - 1) / AS_BM, steps = nblk * (p.N / AS_BN);
  static const bool balance_off = getenv("SASPA_GEMM_BALANCE") && atoi(getenv("SASPA_GEMM_BALANCE")) == 0;
  // few slices per block (N = 320: five): a share of 6.9 steps would reload and re-normalise its rows twice for seven slices
  // -- measured slower than letting the hardware deal whole blocks (LayerNorm + N = 320 at 352 blocks: 60 us against 53.5)
  const bool whole_blocks = balance_off || p.N / AS_BN < 8 || nblk % 256 == 0;
  const dim3 grid((unsigned)(whole_blocks ? nblk : (steps < 256 ? steps : 256)));
  const int abl = 0;
  if (p.act == SASPA_ACT_GEGLU) {
    if (p.N % 160 == 0) hipLaunchKernelGGL((gemm_as_kernel<1, false, 160>), grid, dim3(512), 0, s, p, abl);
    else hipLaunchKernelGGL((gemm_as_kernel<1, false, 128>), grid, dim3(512), 0, s, p, abl);
  } else if (p.residual) {
    hipLaunchKernelGGL((gemm_as_kernel<0, true, 0>), grid, dim3(512), 0, s, p, abl);
  } else {
    hipLaunchKernelGGL((gemm_as_kernel<0, false, 0>), grid, dim3(512), 0, s, p, abl);
  }
  SASPA_CHECK_LAUNCH();
  return 0;
}

// 0: cannot run; 1: can; 2: can, and the work fills the chip evenly -- the sizes where the kernel beats the wave-specialised one
// on the GEGLU projection too (tools/as_bench.py).  Until round 5 that meant row blocks filling whole rounds of the 256 CUs
// (>= 85 % of the last round: 352 blocks at 512x704 did not); with the steps dealt evenly every eligible size does.
// SASPA_GEMM_BALANCE=0 restores the old launch and the old answer.
// THE predicate "variant AUTO runs this problem on the A-stationary kernel" (round 6: shared by dispatch() in saspa_gemm.hip and
// by callers that decide on it before the launch -- ops.conv drops the epilogue GroupNorm statistics only where this says yes;
// saspa_gemm_as_eligible() alone said 2 for every size since the balanced launch, while dispatch() kept its measured rule).
// A fused LayerNorm / transposed tail exists on this kernel only; otherwise it is taken where it measured faster than the tiled /
// wave-specialised kernels (tools/as_bench.py): whole rounds of 256-row blocks, or -- with a ragged last round (352 blocks at
// 512x704) -- the layers with a residual or >= 640 columns, not the GEGLU projection.
extern "C" int saspa_gemm_as_auto(const SaspaGemmParams* pp) {
  if (!pp || pp->variant != SASPA_GEMM_AUTO || pp->defer_reduce || !saspa_gemm_as_ok(*pp)) return 0;
  const SaspaGemmParams& p = *pp;
  const long long blocks = (p.M + 255) / 256;
  const bool whole = blocks * 100 >= ((blocks + 255) / 256) * 256 * 85;
  return (p.ln_gamma || p.out_t || whole || (p.act != SASPA_ACT_GEGLU && (p.residual || p.N >= 640))) ? 1 : 0;
}

extern "C" int saspa_gemm_as_eligible(const SaspaGemmParams* p) {
  if (!p || !saspa_gemm_as_ok(*p)) return 0;
  static const bool balance_off = getenv("SASPA_GEMM_BALANCE") && atoi(getenv("SASPA_GEMM_BALANCE")) == 0;
  if (!balance_off) return 2;
  const long long blocks = (p->M + AS_BM - 1) / AS_BM;
  return blocks * 100 >= ((blocks + 255) / 256) * 256 * 85 ? 2 : 1;
}
