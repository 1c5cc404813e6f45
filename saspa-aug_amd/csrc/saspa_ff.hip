// The feed-forward half of a level-0 transformer block in ONE launch (C = 320 channels, inner width F, GEGLU):
//     y = x + W2 ( v * gelu(g) ) + b2,   [v ; g] = W1 LayerNorm(x) + b1
// i.e. BasicTransformerBlock.norm3 -> ff.net.0 (GEGLU: proj + gate) -> ff.net.2 (Linear) -> residual add of diffusers
// (the reference reaches it through pipe(), run_aug/run_aug.py:278).  As two launches (A-stationary LayerNorm + GEGLU projection,
// then the output projection) the [M, F] hidden state makes an HBM round trip: 168 MB written + 168 MB read at M = 65 536,
// F = 1 280 -- three quarters of the pair's bytes; here x is read once (+ once more as the residual) and y written once.
//
// Built like saspa_xattn.hip on the A-stationary GEMM: a wave keeps its 32 token rows as MFMA B-operand fragments (80 VGPRs) for
// the whole launch, every product is taken transposed (features on the accumulator rows, tokens on the lanes), so the GEGLU result
// of a slice IS the B operand of the next product after packing to bf16.  New here: the wave also keeps the OUTPUT accumulators
// Y^T [320 x 32] (160 registers) for the whole launch, which is why a workgroup is four waves on one wave per SIMD
// (__launch_bounds__(256, 1): 512 registers per wave) -- round 5 sized this fusion on two waves per SIMD and found no room.
// Per slice of 32 hidden features (F / 32 slices, a 2-slot LDS ring filled by LDS-DMA one slice ahead):
//   A. [v ; g]^T = W1_slice LN(x)^T + b1: 64 rows of W1 (32 value rows, then their 32 gate rows: weights.pack_ff_block) in the row
//      layout of the A-stationary kernel (40 16-byte chunks per row, chunk kc of row n at kc ^ ((n >> 1) & 7)): 40 MFMAs;
//   B. h = v * gelu(g) on the 16 accumulator registers, packed to two K-step fragments;
//   C. Y^T += W2_slice h^T: the 320 x 32 slice of W2 arrives as ready-made MFMA A-operand fragments [10 blocks][2 K-steps][64 lanes]
//      [8 bf16] with its K columns in the order stage B's registers come out (a fragment = one wave-wide DMA instruction = 1 KB
//      contiguous in LDS: linear, conflict-free reads): 20 MFMAs.
// Epilogue: Y^T through the (now free) ring as wave-private [32 tokens][320 channels] tiles, + b2 + residual, 16-byte row stores.
// STATUS (round 6): parity-green (tests/test_ff_block_gpu.py) and measured SLOWER than the pair it replaces -- 250 vs 214 us at
// M = 65 536, a tie at 90 112 / 98 304 (profiles/r6_ff_block.txt): a 128-row workgroup streams all of W1 and W2 through LDS-DMA per 128
// rows (3x the weight bytes per row of the 256-row A-stationary launch) and, alone on its SIMD, a wave overlaps neither its DMA
// issue nor the gate's VALU work with its own MFMAs.  Not used by the pipelines; kept as a tested entry point and a measured answer.
// Roofline per slice and CU: 240 MFMAs = 1 920 cycles per SIMD; 4 waves x 60 KB of fragment reads = 1 875 cycles of LDS bandwidth;
// algorithmic HBM bytes per token row: 640 (x) + 640 (residual) + 640 (y) against 640 + 2 x 2 x F + 640 + 640 for the pair.
#include <cstdlib>

#include "common.h"
#include "gemm_internal.h"

namespace {

constexpr int FF_K = 320;
constexpr int FF_KS = FF_K / 16;            // MFMA K-steps of the first projection
constexpr int FF_NW = 4;                    // waves per workgroup (one per SIMD)
constexpr int FF_BM = 32 * FF_NW;           // token rows per workgroup
constexpr int FF_F = 32;                    // hidden features per slice
constexpr int FF_PITCH = 40;                // 16-byte chunks per W1 row in LDS
constexpr int FF_W1_CHUNKS = 64 * FF_PITCH;             // 2 560: 64 rows (32 values + 32 gates)
constexpr int FF_BIAS_SLOT = FF_W1_CHUNKS;               // 64 chunks (16 used: 64 fp32 biases)
constexpr int FF_W1_STAGE = FF_W1_CHUNKS + 64;           // 2 624 chunks = 41 984 bytes per W1 ring slot
constexpr int FF_W2_STAGE = 20 * 64;                     // 1 280 chunks = 20 480 bytes per W2 ring slot (20 fragments)
constexpr int FF_W1_RING = 2, FF_W2_RING = 2;            // 83 968 + 40 960 = 124 928 bytes
constexpr int FF_LDS = FF_W1_RING * FF_W1_STAGE + FF_W2_RING * FF_W2_STAGE;
constexpr int FF_NP1 = 41, FF_NP2 = 20;                  // wave-wide DMA instructions per slice: W1 rows + biases / W2 fragments
constexpr int FF_STG_PITCH = 656;                        // bytes per token row of the epilogue tile (41 chunks: odd)
constexpr int FF_STG_WAVE = 32 * FF_STG_PITCH;           // 20 992 bytes per wave (4 waves: 83 968 <= the rings' 124 928)

__device__ __forceinline__ void ff_unpack_opaque(const u32x4& a, float* v) {
  u32x4 t = a;
  asm volatile("" : "+v"(t));
  unpack8(__builtin_bit_cast(uint4, t), v);
}

__device__ __forceinline__ f32x16 ffmfma(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int abl>
__global__ __launch_bounds__(64 * FF_NW, 1) void ff_block_kernel(const SaspaFfBlockParams p) {
  // abl (diagnostics, SASPA_FF_ABLATE; results are garbage, only the timing counts): 1 no DMA in the loop, 2 no gate arithmetic,
  // 4 no stage C, 8 no stage A MFMAs, 16 no per-slice barrier / vmcnt wait
  __shared__ u32x4 lds[FF_LDS];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 31, h = lane >> 5;
  const long long row = (long long)blockIdx.x * FF_BM + wave * 32 + m;      // < M: the host checks M % 128 == 0
  const int nslices = p.F / FF_F;

  // ---- this lane's half of its row: channels 16 s + 8 h .. + 8 ----
  const rsrc_t rsa = make_rsrc(p.x);
  const unsigned aoff = (unsigned)(row * p.ldx * 2 + h * 16);
  u32x4 af[FF_KS];
#pragma unroll
  for (int s = 0; s < FF_KS; ++s) af[s] = buf_load(rsa, aoff, s * 32);

  // ---- two rings filled by LDS-DMA (a piece = one wave-wide instruction = 1 KB of LDS; wave w issues pieces w, w + 4, ...):
  //      W1 slices (64 rows + 64 biases: 41 pieces) in slot `slice & 1`, W2 slices (20 fragments) in slot `slice & 1` ----
  const rsrc_t rsw1 = make_rsrc(p.w1);
  const rsrc_t rsb1 = make_rsrc(p.b1);
  const rsrc_t rsw2 = make_rsrc(p.w2f);
  u32x4* const w2ring = lds + FF_W1_RING * FF_W1_STAGE;
  // per-lane parts of the source offsets, once: piece k of this wave fills LDS chunks (wave + 4 k) * 64 + lane of the W1 area; the
  // slice term goes into the instruction's scalar offset, so a piece costs no VALU in the loop
  unsigned w1off[11];
#pragma unroll
  for (int k = 0; k < 11; ++k) {
    const int q = wave + FF_NW * k;
    const int c = q * 64 + lane;                                            // LDS chunk of the W1 area this lane fills
    const int n = (int)(((unsigned)c * 52429u) >> 21);                      // c / 40 for c < 2 560
    const int kcp = c - n * FF_PITCH;
    const int dkc = kcp ^ ((n >> 1) & 7);                                   // the W1 chunk that lives there
    w1off[k] = q < 40 ? (unsigned)(n * p.ldw1 * 2 + dkc * 16) : (lane < 16 ? (unsigned)(16 * lane) : kInvalid);
  }
  const unsigned w2off = (unsigned)(lane * 16);
  const int w1_slice_bytes = 64 * p.ldw1 * 2;
  auto dma_w1 = [&](int sl, int k) __attribute__((always_inline)) {         // this wave's k-th W1 piece of slice sl (k = 0 .. 10)
    const int q = wave + FF_NW * k;
    if (q >= FF_NP1 || sl >= nslices || ((abl & 1) && sl >= 2)) return;
    u32x4* slot = lds + (sl & 1) * FF_W1_STAGE;
    if (q < 40) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw1, (lds_void_t*)(slot + q * 64), 16, (int)w1off[k], sl * w1_slice_bytes, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb1, (lds_void_t*)(slot + FF_BIAS_SLOT), 16, (int)w1off[k], sl * 256, 0, 0);
  };
  auto dma_w2 = [&](int sl, int k) __attribute__((always_inline)) {         // this wave's k-th W2 fragment of slice sl (k = 0 .. 4)
    const int j = wave + FF_NW * k;
    if (j >= FF_NP2 || sl >= nslices || ((abl & 1) && sl >= 1)) return;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw2, (lds_void_t*)(w2ring + (sl & 1) * FF_W2_STAGE + j * 64), 16, (int)w2off, (sl * 20 + j) * 1024, 0, 0);
  };
#pragma unroll
  for (int k = 0; k < 11; ++k) dma_w1(0, k);
#pragma unroll
  for (int k = 0; k < 5; ++k) dma_w2(0, k);
#pragma unroll
  for (int k = 0; k < 11; ++k) dma_w1(1, k);

  // ---- LayerNorm of the row, in registers (the arithmetic of layernorm_kernel / gemm_as_kernel / xattn_block_kernel) ----
  if (p.ln_gamma != nullptr) {
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < FF_KS; ++s) {
      float v[8];
      ff_unpack_opaque(af[s], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) sum += v[j];
    }
    sum += __shfl_xor(sum, 32, 64);
    const float mean = sum / (float)FF_K;
    float sq = 0.f;
#pragma unroll
    for (int s = 0; s < FF_KS; ++s) {
      float v[8];
      ff_unpack_opaque(af[s], v);
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[j] - mean; sq += d * d; }
    }
    sq += __shfl_xor(sq, 32, 64);
    const float rstd = 1.0f / sqrtf(sq / (float)FF_K + p.ln_eps);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s = 0; s < FF_KS; ++s) {
      const int k0 = 16 * s + 8 * h;
      float v[8];
      ff_unpack_opaque(af[s], v);
      const f32x4 g0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.ln_gamma + k0)), g1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.ln_gamma + k0 + 4));
      const f32x4 b0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.ln_beta + k0)), b1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.ln_beta + k0 + 4));
      const float g[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
      const float bb[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (v[j] - mean) * rstd * g[j] + bb[j];
      af[s] = __builtin_bit_cast(u32x4, pack8(v));
      if ((s & 3) == 3) {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("" ::: "memory");
      }
    }
  }

  // W1 fragment (K-step s, row block nb) of this lane: row n = 32 nb + m, chunk 2 s + h, stored at chunk ^ ((n >> 1) & 7)
  const unsigned char* fbase = reinterpret_cast<const unsigned char*>(lds) + m * FF_PITCH * 16;
  const int fkey = (m >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) foff[j] = (((2 * j) | h) ^ fkey) << 4;
  constexpr int NG = FF_KS / 2;

  // stage A of slice `sl`: acc = [v ; g]^T = W1_slice LN(x)^T + b1 -- 40 MFMAs in groups of 4 (the W1 fragments of the next two
  // groups in flight), `between(gi)` dealt in behind group gi
  auto stage_a = [&](int sl, f32x16 (&acc)[2], auto&& between) __attribute__((always_inline)) {
    const unsigned char* fs = fbase + (sl & 1) * (FF_W1_STAGE * 16);
    const unsigned char* bs = reinterpret_cast<const unsigned char*>(lds) + ((sl & 1) * FF_W1_STAGE + FF_BIAS_SLOT) * 16;
#pragma unroll
    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *reinterpret_cast<const f32x4*>(bs + (nb * 8 + 2 * g + h) * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[nb][4 * g + j] = bv[j];
      }
    auto frag = [&](int s, int nb) __attribute__((always_inline)) -> u32x4 {
      return *reinterpret_cast<const u32x4*>(fs + foff[s & 3] + (nb * 32 * FF_PITCH * 16 + ((2 * s) & ~7) * 16));
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
        if (abl & 8) asm volatile("" ::"v"(wf[gi % 3][j]));
        else acc[nb] = ffmfma(wf[gi % 3][j], af[sx], acc[nb]);
      }
      between(gi);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // stage C of slice `sl`: Y^T += W2_slice h^T, 20 MFMAs: K-step 0 of blocks 0..9, then K-step 1 (consecutive MFMAs hit different
  // accumulators); fragment (nb, s) = 1 KB at (2 nb + s) * 64 chunks of the slot, three fragments in flight
  f32x16 y[10];
#pragma unroll
  for (int nb = 0; nb < 10; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) y[nb][r] = 0.f;
  auto stage_c = [&](int sl, const u32x4& hf0, const u32x4& hf1) __attribute__((always_inline)) {
    const u32x4* w2 = w2ring + (sl & 1) * FF_W2_STAGE + lane;
    u32x4 vf[4];
    vf[0] = w2[0 * 64];
    vf[1] = w2[2 * 64];
    vf[2] = w2[4 * 64];
#pragma unroll
    for (int i = 0; i < 20; ++i) {
      const int sx = i / 10, nb = i % 10;
      if (i + 3 < 20) {
        const int i3 = i + 3;
        vf[i3 & 3] = w2[(2 * (i3 % 10) + i3 / 10) * 64];
      }
      y[nb] = ffmfma(vf[i & 3], sx ? hf1 : hf0, y[nb]);
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  auto gate = [&](float v, float g) __attribute__((always_inline)) -> float { return (abl & 2) ? v + g : fast_gelu_mul(v, g); };

  // ---- software pipeline: iteration t runs stage A of slice t + 1 on the matrix pipe WITH the GEGLU of slice t on the VALU between
  //      its MFMA groups (the gate costs ~270 VALU per wave and slice: as a phase of its own it was half of the step), then stage C
  //      of slice t.  DMA: W1(t + 2) goes into the W1 slot stage A(t) released an iteration ago, W2(t + 1) into the W2 slot stage
  //      C(t - 1) released; everything a wave issued in an iteration is waited for (vmcnt 0) at the start of the next ----
  f32x16 acur[2], anext[2];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  stage_a(0, acur, [&](int) __attribute__((always_inline)) {});
  for (int t = 0; t < nslices; ++t) {
    if (!(abl & 16)) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of W1(t + 1) and W2(t) have landed
      __builtin_amdgcn_s_barrier();                        // ... everyone's have; W1 slot t & 1 and W2 slot (t + 1) & 1 are free
    }
    float hv[16];
    if (t + 1 < nslices) {
      stage_a(t + 1, anext, [&](int gi) __attribute__((always_inline)) {
        // DMA first (the longest latency: W1(t + 2), then W2(t + 1): 16 pieces behind the first nine groups -- one piece per four
        // MFMAs measured slower, 258 vs 251 us: what hurts is a piece's flight time, not the issue rate), then this group's share
        // of the gate: 16 values over the 10 groups
        if (gi < 6) {
          dma_w1(t + 2, 2 * gi);
          if (2 * gi + 1 < 11) dma_w1(t + 2, 2 * gi + 1);
        } else if (gi < 9) {
          dma_w2(t + 1, 2 * (gi - 6));
          if (2 * (gi - 6) + 1 < 5) dma_w2(t + 1, 2 * (gi - 6) + 1);
        }
        const int r0 = 2 * ((8 * gi) / NG), r1 = 2 * ((8 * (gi + 1)) / NG);
#pragma unroll
        for (int r = r0; r < r1; ++r) hv[r] = gate(acur[0][r], acur[1][r]);
      });
    } else {
#pragma unroll
      for (int r = 0; r < 16; ++r) hv[r] = gate(acur[0][r], acur[1][r]);
    }
    const u32x4 hf0 = {pack2(hv[0], hv[1]), pack2(hv[2], hv[3]), pack2(hv[4], hv[5]), pack2(hv[6], hv[7])};
    const u32x4 hf1 = {pack2(hv[8], hv[9]), pack2(hv[10], hv[11]), pack2(hv[12], hv[13]), pack2(hv[14], hv[15])};
    if (!(abl & 4)) stage_c(t, hf0, hf1);
    acur[0] = anext[0];
    acur[1] = anext[1];
  }

  // ---- epilogue: Y^T -> wave-private [32 tokens][320 channels] tile in the (free) ring -> + b2 + residual -> 16-byte row stores ----
  __builtin_amdgcn_s_barrier();                            // every wave is done reading the ring
  unsigned char* stg = reinterpret_cast<unsigned char*>(lds) + wave * FF_STG_WAVE;
  {
    unsigned char* wr = stg + m * FF_STG_PITCH + h * 8;
#pragma unroll
    for (int nb = 0; nb < 10; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<u32x2*>(wr + nb * 64 + g * 16) = u32x2{pack2(y[nb][4 * g], y[nb][4 * g + 1]), pack2(y[nb][4 * g + 2], y[nb][4 * g + 3])};
  }
  // (wave-private tile: the wave's own LDS writes are ordered before its reads by the lgkmcnt wait the compiler inserts)
  const rsrc_t rsr = make_rsrc(p.residual);
  const rsrc_t rso = make_rsrc(p.out);
  const long long row0 = (long long)blockIdx.x * FF_BM + wave * 32;
#pragma unroll 4
  for (int it = 0; it < 20; ++it) {
    const int q = it * 64 + lane;                          // chunk q of the tile: row q / 40, 16-byte chunk q % 40
    const int r = (int)(((unsigned)q * 52429u) >> 21);
    const int ch = q - r * 40;
    const u32x4 t4 = *reinterpret_cast<const u32x4*>(stg + r * FF_STG_PITCH + ch * 16);
    const u32x4 rv = buf_load(rsr, (unsigned)((row0 + r) * p.ldr * 2 + ch * 16), 0);
    float a[8], r8[8];
    unpack8(__builtin_bit_cast(uint4, t4), a);
    unpack8(__builtin_bit_cast(uint4, rv), r8);
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(p.b2 + ch * 8), c1 = *reinterpret_cast<const f32x4*>(p.b2 + ch * 8 + 4);
    const float bb[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
    // Linear (bias in fp32, one rounding to bf16) then the residual add (a second rounding): the two roundings of the two-launch path
    float o8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o8[j] = a[j] + bb[j];
    u32x4 o4 = __builtin_bit_cast(u32x4, pack8(o8));
    unpack8(__builtin_bit_cast(uint4, o4), o8);
#pragma unroll
    for (int j = 0; j < 8; ++j) o8[j] += r8[j];
    o4 = __builtin_bit_cast(u32x4, pack8(o8));
    __builtin_amdgcn_raw_buffer_store_b128(o4, rso, (int)((row0 + r) * p.ldo * 2 + ch * 16), 0, 0);
  }
}


// ---- wave-specialised form (SASPA_FF_WS=1): EIGHT waves, two per SIMD.  Waves 0-3 ("A") keep the token rows and run LayerNorm, stage A
// and the gate; waves 4-7 ("C", wave 4 + i shares its 32 rows and its SIMD with wave i) keep the output accumulators and run stage C
// and the epilogue.  h of a slice goes from A to C through a 2 KB LDS buffer per pair (lane L reads what lane L wrote: the B-operand
// layout is per lane), one workgroup barrier per slice; iteration t = A(t) || C(t - 1).  Two waves per SIMD again: the gate's VALU
// work and either wave's DMA issue run beside the other wave's MFMAs.  Rows / accumulators split over two waves: <= 256 registers each.
constexpr int FF_HBUF = 4 * 2 * 128;                      // chunks: 4 pairs x 2 buffers x 2 KB

template <int abl>
__global__ __launch_bounds__(512, 1) void ff_block_ws_kernel(const SaspaFfBlockParams p) {
  __shared__ u32x4 lds[FF_LDS + FF_HBUF];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool role_c = wave >= 4;
  const int pair = wave & 3;
  const int m = lane & 31, h = lane >> 5;
  const long long row = (long long)blockIdx.x * FF_BM + pair * 32 + m;
  const int nslices = p.F / FF_F;
  u32x4* const w2ring = lds + FF_W1_RING * FF_W1_STAGE;
  u32x4* const hbuf = lds + FF_LDS + pair * 256;          // this pair's two buffers (128 chunks each)

  // ---- DMA: piece q of a W1 slice belongs to wave q % 8 (k = q / 8: 0 .. 5), fragment j of a W2 slice to wave j % 8 (k = 0 .. 2) ----
  const rsrc_t rsw1 = make_rsrc(p.w1);
  const rsrc_t rsb1 = make_rsrc(p.b1);
  const rsrc_t rsw2 = make_rsrc(p.w2f);
  unsigned w1off[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int q = wave + 8 * k;
    const int c = q * 64 + lane;
    const int n = (int)(((unsigned)c * 52429u) >> 21);
    const int kcp = c - n * FF_PITCH;
    const int dkc = kcp ^ ((n >> 1) & 7);
    w1off[k] = q < 40 ? (unsigned)(n * p.ldw1 * 2 + dkc * 16) : (lane < 16 ? (unsigned)(16 * lane) : kInvalid);
  }
  const unsigned w2off = (unsigned)(lane * 16);
  const int w1_slice_bytes = 64 * p.ldw1 * 2;
  auto dma_w1 = [&](int sl, int k) __attribute__((always_inline)) {
    const int q = wave + 8 * k;
    if (q >= FF_NP1 || sl >= nslices || ((abl & 1) && sl >= 2)) return;
    u32x4* slot = lds + (sl & 1) * FF_W1_STAGE;
    if (q < 40) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw1, (lds_void_t*)(slot + q * 64), 16, (int)w1off[k], sl * w1_slice_bytes, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsb1, (lds_void_t*)(slot + FF_BIAS_SLOT), 16, (int)w1off[k], sl * 256, 0, 0);
  };
  auto dma_w2 = [&](int sl, int k) __attribute__((always_inline)) {
    const int j = wave + 8 * k;
    if (j >= FF_NP2 || sl >= nslices || ((abl & 1) && sl >= 2)) return;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw2, (lds_void_t*)(w2ring + (sl & 1) * FF_W2_STAGE + j * 64), 16, (int)w2off, (sl * 20 + j) * 1024, 0, 0);
  };
  // the pieces this wave issues in iteration t: W1(t + 1) (stage A(t + 1) runs in iteration t + 1) and W2(t - 1) (stage C(t - 1) runs
  // in iteration t + 1: the gate sits between them)
  auto dma_iter = [&](int t, int part) __attribute__((always_inline)) {       // part 0 .. 2: two W1 pieces and one W2 fragment each
    dma_w1(t + 1, 2 * part);
    dma_w1(t + 1, 2 * part + 1);
    if (t >= 1) dma_w2(t - 1, part);
  };
#pragma unroll
  for (int k = 0; k < 6; ++k) dma_w1(0, k);

  if (!role_c) {
    // =============================== A: rows, LayerNorm, stage A, gate ===============================
    const rsrc_t rsa = make_rsrc(p.x);
    const unsigned aoff = (unsigned)(row * p.ldx * 2 + h * 16);
    u32x4 af[FF_KS];
#pragma unroll
    for (int s = 0; s < FF_KS; ++s) af[s] = buf_load(rsa, aoff, s * 32);
    if (p.ln_gamma != nullptr) {
      float sum = 0.f;
#pragma unroll
      for (int s = 0; s < FF_KS; ++s) {
        float v[8];
        ff_unpack_opaque(af[s], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += v[j];
      }
      sum += __shfl_xor(sum, 32, 64);
      const float mean = sum / (float)FF_K;
      float sq = 0.f;
#pragma unroll
      for (int s = 0; s < FF_KS; ++s) {
        float v[8];
        ff_unpack_opaque(af[s], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[j] - mean; sq += d * d; }
      }
      sq += __shfl_xor(sq, 32, 64);
      const float rstd = 1.0f / sqrtf(sq / (float)FF_K + p.ln_eps);
      asm volatile("" ::: "memory");
#pragma unroll
      for (int s = 0; s < FF_KS; ++s) {
        const int k0 = 16 * s + 8 * h;
        float v[8];
        ff_unpack_opaque(af[s], v);
        const f32x4 g0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.ln_gamma + k0)), g1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.ln_gamma + k0 + 4));
        const f32x4 b0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.ln_beta + k0)), b1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p.ln_beta + k0 + 4));
        const float g[8] = {g0[0], g0[1], g0[2], g0[3], g1[0], g1[1], g1[2], g1[3]};
        const float bb[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (v[j] - mean) * rstd * g[j] + bb[j];
        af[s] = __builtin_bit_cast(u32x4, pack8(v));
        if ((s & 3) == 3) {
          __builtin_amdgcn_sched_barrier(0);
          asm volatile("" ::: "memory");
        }
      }
    }
    const unsigned char* fbase = reinterpret_cast<const unsigned char*>(lds) + m * FF_PITCH * 16;
    const int fkey = (m >> 1) & 7;
    int foff[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) foff[j] = (((2 * j) | h) ^ fkey) << 4;
    constexpr int NG = FF_KS / 2;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // W1(0) is in place
    // iteration t: the MFMAs of stage A(t) with the gate of slice t - 1 on the VALU between their groups (the gate is ~270 VALU per
    // slice: after the MFMAs it made this wave the critical path, 184 us per launch); h(t - 1) is handed over at the end
    f32x16 acur[2], anext[2];
    for (int t = 0; t <= nslices; ++t) {
      float hv[16];
      if (t < nslices) {
        const unsigned char* fs = fbase + (t & 1) * (FF_W1_STAGE * 16);
        const unsigned char* bs = reinterpret_cast<const unsigned char*>(lds) + ((t & 1) * FF_W1_STAGE + FF_BIAS_SLOT) * 16;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(bs + (nb * 8 + 2 * g + h) * 16);
#pragma unroll
            for (int j = 0; j < 4; ++j) anext[nb][4 * g + j] = bv[j];
          }
        auto frag = [&](int s, int nb) __attribute__((always_inline)) -> u32x4 {
          return *reinterpret_cast<const u32x4*>(fs + foff[s & 3] + (nb * 32 * FF_PITCH * 16 + ((2 * s) & ~7) * 16));
        };
        u32x4 wf[3][4];
#pragma unroll
        for (int gi = 0; gi < 2; ++gi)
#pragma unroll
          for (int j = 0; j < 4; ++j) wf[gi][j] = frag(2 * gi + (j >> 1), j & 1);
#pragma unroll
        for (int gi = 0; gi < NG; ++gi) {
          if (gi + 2 < NG && !(abl & 16)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) wf[(gi + 2) % 3][j] = frag(2 * (gi + 2) + (j >> 1), j & 1);
          }
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int sx = 2 * gi + (j >> 1), nb = j & 1;
            if (abl & 8) asm volatile("" ::"v"(wf[gi % 3][j]));
            else anext[nb] = ffmfma(wf[(abl & 16) ? (gi & 1) : gi % 3][j], af[sx], anext[nb]);
          }
          if (gi < 3) dma_iter(t, gi);                     // this wave's share of W1(t + 1) / W2(t - 1), early in the iteration
          if (t >= 1) {
            const int r0 = 2 * ((8 * gi) / NG), r1 = 2 * ((8 * (gi + 1)) / NG);
#pragma unroll
            for (int r = r0; r < r1; r += 2) {
              if (abl & 2) { hv[r] = acur[0][r] + acur[1][r]; hv[r + 1] = acur[0][r + 1] + acur[1][r + 1]; }
              else {
                const f32x2_pk o = fast_gelu_mul2(f32x2_pk{acur[0][r], acur[0][r + 1]}, f32x2_pk{acur[1][r], acur[1][r + 1]});
                hv[r] = o[0];
                hv[r + 1] = o[1];
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int part = 0; part < 3; ++part) dma_iter(t, part);   // W2(nslices - 1)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2_pk o = fast_gelu_mul2(f32x2_pk{acur[0][r], acur[0][r + 1]}, f32x2_pk{acur[1][r], acur[1][r + 1]});
          hv[r] = o[0];
          hv[r + 1] = o[1];
        }
      }
      if (t >= 1) {
        u32x4* hb = hbuf + ((t - 1) & 1) * 128 + lane;
        hb[0] = u32x4{pack2(hv[0], hv[1]), pack2(hv[2], hv[3]), pack2(hv[4], hv[5]), pack2(hv[6], hv[7])};
        hb[64] = u32x4{pack2(hv[8], hv[9]), pack2(hv[10], hv[11]), pack2(hv[12], hv[13]), pack2(hv[14], hv[15])};
      }
      acur[0] = anext[0];
      acur[1] = anext[1];
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                        // h(t - 1), W1(t + 1), W2(t - 1) are in place for everyone
    }
    __builtin_amdgcn_s_barrier();                          // (matches the C waves' last iteration)
    __builtin_amdgcn_s_barrier();                          // (matches the C waves' epilogue barrier)
    return;
  }

  // =============================== C: output accumulators, stage C, epilogue ===============================
  f32x16 y[10];
#pragma unroll
  for (int nb = 0; nb < 10; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) y[nb][r] = 0.f;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                            // W1(0) is in place (the A waves' first barrier)
  for (int t = 0; t <= nslices + 1; ++t) {
    if (t <= nslices) {
#pragma unroll
      for (int part = 0; part < 3; ++part) dma_iter(t, part);
    }
    if (t >= 2 && !(abl & 4)) {
      const int sl = t - 2;
      const u32x4* hb = hbuf + (sl & 1) * 128 + lane;
      const u32x4 hf0 = hb[0], hf1 = hb[64];
      const u32x4* w2 = w2ring + (sl & 1) * FF_W2_STAGE + lane;
      u32x4 vf[4];
      vf[0] = w2[0 * 64];
      vf[1] = w2[2 * 64];
      vf[2] = w2[4 * 64];
#pragma unroll
      for (int i = 0; i < 20; ++i) {
        const int sx = i / 10, nb = i % 10;
        if (i + 3 < 20 && !(abl & 16)) {
          const int i3 = i + 3;
          vf[i3 & 3] = w2[(2 * (i3 % 10) + i3 / 10) * 64];
        }
        y[nb] = ffmfma(vf[(abl & 16) ? i % 3 : (i & 3)], sx ? hf1 : hf0, y[nb]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // ---- epilogue (as the four-wave kernel): Y^T -> [32 tokens][320 channels] tile in the freed rings -> + b2 + residual -> stores ----
  __builtin_amdgcn_s_barrier();
  unsigned char* stg = reinterpret_cast<unsigned char*>(lds) + pair * FF_STG_WAVE;
  {
    unsigned char* wr = stg + m * FF_STG_PITCH + h * 8;
#pragma unroll
    for (int nb = 0; nb < 10; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        *reinterpret_cast<u32x2*>(wr + nb * 64 + g * 16) = u32x2{pack2(y[nb][4 * g], y[nb][4 * g + 1]), pack2(y[nb][4 * g + 2], y[nb][4 * g + 3])};
  }
  const rsrc_t rsr = make_rsrc(p.residual);
  const rsrc_t rso = make_rsrc(p.out);
  const long long row0 = (long long)blockIdx.x * FF_BM + pair * 32;
#pragma unroll 4
  for (int it = 0; it < 20; ++it) {
    const int q = it * 64 + lane;
    const int r = (int)(((unsigned)q * 52429u) >> 21);
    const int ch = q - r * 40;
    const u32x4 t4 = *reinterpret_cast<const u32x4*>(stg + r * FF_STG_PITCH + ch * 16);
    const u32x4 rv = buf_load(rsr, (unsigned)((row0 + r) * p.ldr * 2 + ch * 16), 0);
    float a[8], r8[8];
    unpack8(__builtin_bit_cast(uint4, t4), a);
    unpack8(__builtin_bit_cast(uint4, rv), r8);
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(p.b2 + ch * 8), c1 = *reinterpret_cast<const f32x4*>(p.b2 + ch * 8 + 4);
    const float bb[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
    float o8[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o8[j] = a[j] + bb[j];
    u32x4 o4 = __builtin_bit_cast(u32x4, pack8(o8));
    unpack8(__builtin_bit_cast(uint4, o4), o8);
#pragma unroll
    for (int j = 0; j < 8; ++j) o8[j] += r8[j];
    o4 = __builtin_bit_cast(u32x4, pack8(o8));
    __builtin_amdgcn_raw_buffer_store_b128(o4, rso, (int)((row0 + r) * p.ldo * 2 + ch * 16), 0, 0);
  }
}

}  // namespace

extern "C" int saspa_ff_block_eligible(const SaspaFfBlockParams* pp) {
  if (!pp) return 0;
  const SaspaFfBlockParams& p = *pp;
  if (p.M <= 0 || p.M % FF_BM || p.F <= 0 || p.F % FF_F || p.F > 8192) return 0;
  if (p.ldx < FF_K || p.ldr < FF_K || p.ldo < FF_K || p.ldw1 < FF_K) return 0;
  if (p.ldx % 8 || p.ldr % 8 || p.ldo % 8 || p.ldw1 % 8) return 0;
  const long long ld = p.ldx > p.ldo ? (p.ldx > p.ldr ? p.ldx : p.ldr) : (p.ldo > p.ldr ? p.ldo : p.ldr);
  if ((long long)p.M * ld * 2 >= 0x7fffffffLL) return 0;                                   // 32-bit buffer offsets
  if ((long long)2 * p.F * p.ldw1 * 2 >= 0x7fffffffLL) return 0;
  return 1;
}

extern "C" int saspa_ff_block(const SaspaFfBlockParams* pp, void* stream) {
  if (!pp) return SASPA_EINVAL;
  const SaspaFfBlockParams& p = *pp;
  if (!p.x || !p.residual || !p.out || !p.w1 || !p.b1 || !p.w2f || !p.b2) return SASPA_EINVAL;
  if ((p.ln_gamma == nullptr) != (p.ln_beta == nullptr)) return SASPA_EINVAL;
  if (!aligned16(p.x) || !aligned16(p.residual) || !aligned16(p.out) || !aligned16(p.w1) || !aligned16(p.b1) || !aligned16(p.w2f) ||
      !aligned16(p.b2) || (p.ln_gamma && (!aligned16(p.ln_gamma) || !aligned16(p.ln_beta))))
    return SASPA_EALIGN;
  if (!saspa_ff_block_eligible(pp)) return SASPA_ERANGE;
  const dim3 grid((unsigned)(p.M / FF_BM)), block(64 * FF_NW);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // the wave-specialised form is the one that ships (it is the faster: tools/ff_bench.py); SASPA_FF_WS=0 selects the four-wave form,
  // kept as the structurally simpler cross-check of the same arithmetic (bit-identical results).  Read per launch: ff_bench flips it.
  const char* we = getenv("SASPA_FF_WS");
  const bool ws = !(we && atoi(we) == 0);
#ifdef SASPA_FF_ABLATION
  // diagnostics build only (-DSASPA_FF_ABLATION, tools/ff_bench.py <flags>): one instantiation per ablation, read per launch
  const char* ae = getenv("SASPA_FF_ABLATE");
  const int av = ae ? atoi(ae) : 0;
  if (ws) {
    switch (av) {
#define SASPA_FFW(A_) case A_: hipLaunchKernelGGL(ff_block_ws_kernel<A_>, grid, dim3(512), 0, s, p); SASPA_CHECK_LAUNCH(); return 0;
      SASPA_FFW(1) SASPA_FFW(2) SASPA_FFW(4) SASPA_FFW(8) SASPA_FFW(3) SASPA_FFW(16) SASPA_FFW(17) SASPA_FFW(19)
#undef SASPA_FFW
      default: break;
    }
  } else {
    switch (av) {
#define SASPA_FFA(A_) case A_: hipLaunchKernelGGL(ff_block_kernel<A_>, grid, block, 0, s, p); SASPA_CHECK_LAUNCH(); return 0;
      SASPA_FFA(1) SASPA_FFA(2) SASPA_FFA(4) SASPA_FFA(8) SASPA_FFA(16) SASPA_FFA(3) SASPA_FFA(12) SASPA_FFA(15) SASPA_FFA(31)
#undef SASPA_FFA
      default: break;
    }
  }
#endif
  if (ws) hipLaunchKernelGGL(ff_block_ws_kernel<0>, grid, dim3(512), 0, s, p);
  else hipLaunchKernelGGL(ff_block_kernel<0>, grid, block, 0, s, p);
  SASPA_CHECK_LAUNCH();
  return 0;
}
