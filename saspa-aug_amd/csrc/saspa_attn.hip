// Fused flash attention (bf16, fp32 accumulate) for gfx950 and the row-softmax used
// by the unfused (fp32 parity / VAE 512-wide head) path.  See include/saspa_hip.h.
//
// Formulation (per wave: 32 queries, per workgroup: 4 waves = 128 queries, KV tile = 64 or 128 keys):
//   S^T = K Q^T     v_mfma_f32_32x32x16_bf16, A = K rows (LDS), B = Q rows (registers).
//                   The accumulator then has the QUERY on the lane (column) and 16 of the
//                   32 keys in registers, so the softmax max / sum are in-register plus one
//                   exchange with lane^32, and the O rescale is a per-lane scalar.
//   O^T += V^T P^T  P^T (bf16-packed accumulator registers 8s..8s+7) is directly the B
//                   operand of the next MFMA; its k order inside a 16-key step is
//                   key = 8*(j>>2) + 4*(lane>>5) + (j&3), so the A operand (V^T rows, keys
//                   contiguous) is read as two 8-byte pieces at key offsets 4h and 8+4h.
//   V^T comes from the value projection computed with swapped GEMM operands
//   (vt[b] = Wv x_b^T), so no transpose pass exists anywhere.
// LDS images: K tile rows padded to an odd number of 16-byte slots (conflict-free
// ds_read_b128 across 32 distinct rows); V^T rows = 64 keys + 8 bytes pad (conflict-free
// ds_read_b64: row stride 136 B = 8*17).
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

// ---- V ROW-MAJOR (SASPA_ATTN_V_ROWMAJOR, ABI 14) ------------------------------------------------------------------
// The PV product wants, per lane, 8 consecutive keys of ONE channel (the A operand of O^T += V^T P^T) -- which is why V was
// only ever materialised transposed, by a separate projection launch.  gfx950's ds_read_b64_tr_b16 delivers exactly that
// from a ROW-MAJOR [key][d] LDS tile: per group of 16 lanes it reads a block of 4 rows x 16 columns and hands lane i column
// i of the 4 rows.  So V can stay what the fused Q | K | V projection writes ([tokens][C] row-major) and the V^T launches of
// the 32x32 / 16x16 levels disappear.  Lane 4 q + p of a group supplies the address of (row q, columns 4 p .. 4 p + 3).
// LDS row pitch: a 32-lane half touches 4 rows x 16 banks (two groups 32 bytes apart x four 8-byte pieces), so the rows must
// sit 16 or 48 banks apart (mod 64): 64 B for 32 columns, 192 B for 64 / 96, 320 B for 128 / 160.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x2 lds_read_tr16(const unsigned char* ptr) {
  return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ptr)));
}
constexpr int rm_pitch(int dv) {
  int b = dv * 2;
  while ((b / 4) % 64 != 16 && (b / 4) % 64 != 48) b += 64;
  return b;
}

template <int KS, int NB, bool ONES, int KT, bool RM = false>
__global__ __launch_bounds__(256) void flash_attn_kernel(const SaspaAttnParams p) {
  // KT = keys per K/V tile (64 or 128): a larger tile halves the barriers / waits / staging bursts per key
  constexpr int NKB = KT / 32;               // 32-key blocks of S^T per tile
  constexpr int VCH = KT / 8;                // 16-byte chunks per V^T row
  // ONES: D < 32*NB, so V^T row D is a spare MFMA row; it is filled with ones and the PV
  // MFMA then accumulates the softmax denominator there (no VALU row-sum in the loop).
  constexpr int KSLOTS = (2 * KS) | 1;       // 16-byte slots per K row (odd)
  constexpr int KCH = 2 * KS;                // chunks per K row that are written
  constexpr int DV = NB * 32;
  constexpr int VROW = KT * 2 + 8;           // bytes; (VROW/8) odd -> conflict-free ds_read_b64
  constexpr int RROW = rm_pitch(DV);         // RM: bytes per key row of the row-major V tile
  constexpr int DCH = DV / 8;                // RM: 16-byte chunks per key row
  constexpr int K_BYTES = KT * KSLOTS * 16;
  constexpr int V_BYTES = RM ? KT * RROW : DV * VROW;
  constexpr int NCH_K = (KT * KCH + 255) / 256;
  constexpr int NCH_V = RM ? (KT * DCH + 255) / 256 : (DV * VCH + 255) / 256;
  __shared__ __attribute__((aligned(16))) unsigned char smem[K_BYTES + V_BYTES];
  unsigned char* ksm = smem;
  unsigned char* vsm = smem + K_BYTES;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware block order: consecutive workgroup ids go round-robin over the 8 XCDs (each with its own L2), so the
  // plain (query block fastest) order puts every (batch, head)'s K / V^T into all 8 L2s -- 8x the HBM-side fetch (PMC:
  // 713 MB per level-0 launch against 168 MB of operands).  Remap: ids congruent mod 8 (= one XCD) walk the query blocks
  // of ONE (batch, head); the 8 pairs of a group share the 8 XCDs.  A tail group of fewer than 8 pairs is spread evenly.
  const int gx = gridDim.x, nbh = gridDim.y * gridDim.z;
  const int lin = blockIdx.x + gx * (blockIdx.y + gridDim.y * blockIdx.z);
  const int grp = lin / (8 * gx), rr = lin - grp * 8 * gx;
  const int gsz = min(8, nbh - grp * 8);
  const int bh = grp * 8 + rr % gsz, xq = rr / gsz;
  const int head = bh % (int)gridDim.y, b = bh / (int)gridDim.y;
  const int D = p.D, D8 = D >> 3;
  const int q0 = xq * 128 + wave * 32;
  const int qi = q0 + r;  // this lane's query

  const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + b * p.sqb + head * D;
  const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.k) + b * p.skb + head * D;
  const bf16_t* VT = RM ? reinterpret_cast<const bf16_t*>(p.vt) + b * p.svb + head * D          // V[b][key][head*D + d]
                        : reinterpret_cast<const bf16_t*>(p.vt) + b * p.svb + (long long)head * D * p.ldvt;
  bf16_t* O = reinterpret_cast<bf16_t*>(p.o) + b * p.sob + head * D;

  const u32x4 zero4 = {0u, 0u, 0u, 0u};

  // ---- Q fragments (B operand), resident for the whole kernel ----
  u32x4 qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int d = 16 * s + 8 * h;
    qf[s] = (qi < p.nq && d < D) ? *reinterpret_cast<const u32x4*>(Q + (long long)qi * p.ldq + d) : zero4;
  }

  // ---- staging: bounds-checked buffer loads (an offset >= num_records returns zeros), per-lane
  //      offsets fixed for the whole kernel, the tile advance in a scalar offset ----
  const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Kp), (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(VT), (short)0, 0x7fffffff, 0x00020000);
  constexpr unsigned kInv = 0x80000000u;
  unsigned koff[NCH_K], voff[NCH_V];
  int k_key[NCH_K], k_lds[NCH_K], v_lds[NCH_V], v_d[NCH_V], v_kc[NCH_V];
#pragma unroll
  for (int i = 0; i < NCH_K; ++i) {
    const int q = tid + 256 * i;
    const int key = q / KCH, ch = q - key * KCH;
    k_key[i] = key;
    k_lds[i] = (q < KT * KCH) ? (key * KSLOTS + ch) * 16 : -1;
    koff[i] = (q < KT * KCH && ch < D8) ? (unsigned)(key * p.ldk * 2 + ch * 16) : kInv;   // pad chunks read as zeros
  }
  if constexpr (RM) {
    // row-major V tile: chunk ch (8 channels) of key row `key`; chunks >= D/8 never change: the ones column (denominator,
    // when ONES: channel D = first element of chunk D/8) / zeros
#pragma unroll
    for (int i = 0; i < NCH_V; ++i) {
      const int q = tid + 256 * i;
      const int key = q / DCH, ch = q - key * DCH;
      v_d[i] = ch;
      v_kc[i] = key;                         // (RM: the tile row = key)
      v_lds[i] = (q < KT * DCH && ch < D8) ? key * RROW + ch * 16 : -1;
      voff[i] = (q < KT * DCH && ch < D8) ? (unsigned)(key * p.ldvt * 2 + ch * 16) : kInv;
    }
    for (int q = tid; q < KT * DCH; q += 256) {
      const int key = q / DCH, ch = q - key * DCH;
      if (ch >= D8) *reinterpret_cast<u32x4*>(vsm + key * RROW + ch * 16) = u32x4{(ONES && ch == D8) ? 0x00003F80u : 0u, 0u, 0u, 0u};
    }
  } else {
#pragma unroll
  for (int i = 0; i < NCH_V; ++i) {
    const int q = tid + 256 * i;
    const int d = q / VCH, kc = q - d * VCH;
    v_d[i] = d;
    v_kc[i] = kc;
    v_lds[i] = (q < DV * VCH && d < D) ? d * VROW + kc * 16 : -1;     // rows >= D are written once, below
    voff[i] = (q < DV * VCH && d < D) ? (unsigned)(d * p.ldvt * 2 + kc * 16) : kInv;
  }
  // rows D .. DV-1 of the V^T tile never change: ones (denominator row, when ONES) / zeros
  for (int q = tid; q < DV * VCH; q += 256) {
    const int d = q / VCH, kc = q - d * VCH;
    if (d >= D) {
      const unsigned fill = (ONES && d == D) ? 0x3F803F80u : 0u;
      u32x2* dst = reinterpret_cast<u32x2*>(vsm + d * VROW + kc * 16);
      dst[0] = u32x2{fill, fill};
      dst[1] = u32x2{fill, fill};
    }
  }
  }
  u32x4 kreg[NCH_K], vreg[NCH_V];
  // RM: this lane's address inside a 4-key x 16-channel block of the transposed read (group g = lane >> 4: channels
  // 16 (g & 1) .., keys 4 (g >> 1) ..; lane 4 q + p of the group: key q, channels 4 p ..)
  const int tr_base = (4 * (lane >> 5) + ((lane & 15) >> 2)) * RROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  auto load_tile = [&](int key0) __attribute__((always_inline)) {
    const bool tail = key0 + KT > p.nk;                  // wave-uniform
    const unsigned sk = (unsigned)(key0 * p.ldk * 2), sv = (unsigned)(key0 * 2);
#pragma unroll
    for (int i = 0; i < NCH_K; ++i) {
      unsigned o = koff[i];
      if (tail && key0 + k_key[i] >= p.nk) o = kInv;
      kreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsk, (int)o, (int)sk, 0));
    }
#pragma unroll
    for (int i = 0; i < NCH_V; ++i) {
      unsigned o = voff[i];
      if (RM) {
        if (tail && key0 + v_kc[i] >= p.nk) o = kInv;                    // key rows >= nk: zeros
        vreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsv, (int)o, (int)(key0 * p.ldvt * 2), 0));
      } else {
        if (tail && key0 + v_kc[i] * 8 >= p.nk) o = kInv;
        vreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsv, (int)o, (int)sv, 0));
      }
    }
  };
  auto store_tile = [&](int key0) __attribute__((always_inline)) {
    const bool tail = key0 + KT > p.nk;   // wave-uniform: only the last tile can hold keys >= nk
#pragma unroll
    for (int i = 0; i < NCH_K; ++i)
      if (k_lds[i] >= 0) *reinterpret_cast<u32x4*>(ksm + k_lds[i]) = kreg[i];
#pragma unroll
    for (int i = 0; i < NCH_V; ++i) {
      if (RM) {
        if (v_lds[i] >= 0) *reinterpret_cast<u32x4*>(vsm + v_lds[i]) = vreg[i];
        continue;
      }
      if (v_lds[i] >= 0) {
        u32x4 v = vreg[i];
        if (tail) {
          // zero the keys >= nk (pad columns of vt are not initialised by the producer)
          const int nvalid = p.nk - (key0 + v_kc[i] * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned keep = ((2 * e < nvalid) ? 0x0000ffffu : 0u) | ((2 * e + 1 < nvalid) ? 0xffff0000u : 0u);
            v[e] &= keep;
          }
        }
        u32x2* dst = reinterpret_cast<u32x2*>(vsm + v_lds[i]);
        dst[0] = u32x2{v.x, v.y};
        dst[1] = u32x2{v.z, v.w};
      }
    }
  };

  f32x16 acc_o[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc_o[nb][i] = 0.f;
  const float c = p.scale * 1.4426950408889634f;   // scores are kept raw; exp2(s*c - m) folds the scale
  float m_run = -1e30f;                             // running max, already multiplied by c
  float l_run = 0.f;                                // VALU row-sum (only when !ONES)

  const int ntiles = (p.nk + KT - 1) / KT;
  load_tile(0);
  for (int t = 0; t < ntiles; ++t) {
    const int key0 = t * KT;
    __syncthreads();            // previous tile fully consumed
    store_tile(key0);
    __syncthreads();
    if (t + 1 < ntiles) load_tile(key0 + KT);   // in flight during the MFMA / softmax block below

    // ---- S^T = K Q^T for two 32-key blocks ----
    f32x16 acc_s[NKB];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const u32x4 kf = *reinterpret_cast<const u32x4*>(ksm + ((kb * 32 + r) * KSLOTS + 2 * s + h) * 16);
        acc_s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[s]),
                                                            s == 0 ? zero16 : acc_s[kb], 0, 0, 0);   // C = inline 0
      }
    }
    // ---- masks only on the tiles that need them (wave-uniform) ----
    if (key0 + KT > p.nk || p.causal) {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = key0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (key >= p.nk || (p.causal && key > qi)) acc_s[kb][i] = -INFINITY;
        }
    }
    // ---- online softmax, query on the lane ----
    float mx = acc_s[0][0];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, acc_s[kb][i]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx * c);
    if (__any(m_new > m_run)) {   // some query's max moved: rescale everything at the old max once
      const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
      m_run = m_new;
      if (!ONES) l_run *= alpha;
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc_o[nb][i] *= alpha;
    }
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(acc_s[kb][i], c, -m_run));
        acc_s[kb][i] = pv;
        if (!ONES) psum += pv;
      }
    if (!ONES) l_run += psum;

    // ---- O^T += V^T P^T  (row D of V^T is ones when ONES: accumulates the denominator) ----
#pragma unroll
    for (int ks = 0; ks < 2 * NKB; ++ks) {
      const int kb = ks >> 1, half = ks & 1;
      u32x4 pf;
      pf.x = pack2(acc_s[kb][8 * half + 0], acc_s[kb][8 * half + 1]);
      pf.y = pack2(acc_s[kb][8 * half + 2], acc_s[kb][8 * half + 3]);
      pf.z = pack2(acc_s[kb][8 * half + 4], acc_s[kb][8 * half + 5]);
      pf.w = pack2(acc_s[kb][8 * half + 6], acc_s[kb][8 * half + 7]);
      const int koff = (kb * 32 + 16 * half + 4 * h) * 2;  // bytes
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        u32x2 lo, hi;
        if (RM) {
          // keys 16 ks + 4 h + (0..3) and + 8, channel 32 nb + r: two transposed reads of the row-major tile
          const unsigned char* blk = vsm + tr_base + (kb * 32 + 16 * half) * RROW + nb * 64;
          lo = lds_read_tr16(blk);
          hi = lds_read_tr16(blk + 8 * RROW);
        } else {
          const unsigned char* vrow = vsm + (nb * 32 + r) * VROW + koff;
          lo = *reinterpret_cast<const u32x2*>(vrow);
          hi = *reinterpret_cast<const u32x2*>(vrow + 16);
        }
        const u32x4 vf = {lo.x, lo.y, hi.x, hi.y};
        acc_o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf),
                                                            __builtin_bit_cast(bf16x8, pf), acc_o[nb], 0, 0, 0);
      }
    }
  }

  // ---- normalise and store: lane holds O[qi][d], d = 32*nb + 8*g + 4*h + (0..3) ----
  float l_tot;
  if (ONES) {
    // the denominator sits in accumulator row D: block D/32, register 4*((D%32)/8), half h = 0
    float lsel = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (32 * nb + 8 * g == D) lsel = acc_o[nb][4 * g];
    l_tot = __shfl(lsel, r, 64);
  } else {
    l_tot = l_run + __shfl_xor(l_run, 32, 64);
  }
  const float inv = 1.0f / l_tot;
  if (qi < p.nq) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * nb + 8 * g + 4 * h;
        if (d < D) {
          float v[4] = {acc_o[nb][4 * g + 0] * inv, acc_o[nb][4 * g + 1] * inv, acc_o[nb][4 * g + 2] * inv,
                        acc_o[nb][4 * g + 3] * inv};
          Elem<bf16_t>::store4(O + (long long)qi * p.ldo + d, v);
        }
      }
  }
}

// ---- v2: the loop for PRESCALED queries (SaspaAttnParams.flags & SASPA_ATTN_QPRESCALED) -------------------------
// The caller has folded scale * log2(e) into the query projection's weights, so K Q^T already is the log2-domain
// logit.  That removes the two VALU instructions per score the loop above spends besides the exponential:
//   * the running reference level m rides the QK^T MFMA as its C operand (a 16-register splat of -m that only
//     changes when m does): the accumulator comes out as s - m, and p = exp2(acc) needs no subtract / FMA;
//   * softmax is exact for ANY reference level (it cancels in O / l), it only has to keep exp2 in range.  So m is
//     not the running maximum but "the maximum seen at the last update + BIAS": p stays <= 2^-BIAS until a logit
//     exceeds that maximum by BIAS + 1, which the loop detects without a max chain -- p >= 2 <=> bit 14 of its
//     bf16 pattern, so OR-ing the packed P registers (v_or3_b32, 16 per 128-key tile) and testing 0x40004000
//     is an exact "some p >= 2.0" (inf / NaN included).  Only then (and on tile 0, which sets the level) the
//     slow path takes the true row maximum, moves m, rescales O once and re-exponentiates the tile.
//     bf16 / fp32 keep 8 exponent bits, so p ~ 2^-8 loses no relative precision; l and O are fp32.
//   * V^T tile keys are stored permuted inside each 16-key group ([0-3 | 8-11 | 4-7 | 12-15]) so that the A
//     operand of the PV MFMA (k order 8*(j>>2) + 4*(lane>>5) + (j&3)) is ONE ds_read_b128 per lane instead of
//     a ds_read2_b64; rows are KT*2 + 16 bytes (an odd number of 16-byte slots: conflict-free).
//   * DB: two LDS tile buffers, one barrier per tile (tile t+1 is written while tile t is multiplied).
template <int KS, int NB, bool ONES, int KT, bool DB>
__global__ __launch_bounds__(256) void flash_attn_v2_kernel(const SaspaAttnParams p) {
  constexpr int NKB = KT / 32;
  constexpr int VCH = KT / 8;
  constexpr int KSLOTS = (2 * KS) | 1;
  constexpr int KCH = 2 * KS;
  constexpr int DV = NB * 32;
  constexpr int VROW = KT * 2 + 16;
  constexpr int K_BYTES = KT * KSLOTS * 16;
  constexpr int V_BYTES = DV * VROW;
  constexpr int BUF = K_BYTES + V_BYTES;
  constexpr int NCH_K = (KT * KCH + 255) / 256;
  constexpr int NCH_V = (DV * VCH + 255) / 256;
  constexpr float BIAS = 8.0f;
  __shared__ __attribute__((aligned(16))) unsigned char smem[(DB ? 2 : 1) * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int gx = gridDim.x, nbh = gridDim.y * gridDim.z;
  const int lin = blockIdx.x + gx * (blockIdx.y + gridDim.y * blockIdx.z);
  const int grp = lin / (8 * gx), rr = lin - grp * 8 * gx;
  const int gsz = min(8, nbh - grp * 8);
  const int bh = grp * 8 + rr % gsz, xq = rr / gsz;
  const int head = bh % (int)gridDim.y, b = bh / (int)gridDim.y;
  const int D = p.D, D8 = D >> 3;
  const int q0 = xq * 128 + wave * 32;
  const int qi = q0 + r;

  const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + b * p.sqb + head * D;
  const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.k) + b * p.skb + head * D;
  const bf16_t* VT = reinterpret_cast<const bf16_t*>(p.vt) + b * p.svb + (long long)head * D * p.ldvt;
  bf16_t* O = reinterpret_cast<bf16_t*>(p.o) + b * p.sob + head * D;

  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int d = 16 * s + 8 * h;
    qf[s] = (qi < p.nq && d < D) ? *reinterpret_cast<const u32x4*>(Q + (long long)qi * p.ldq + d) : zero4;
  }

  const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Kp), (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(VT), (short)0, 0x7fffffff, 0x00020000);
  constexpr unsigned kInv = 0x80000000u;
  unsigned koff[NCH_K], voff[NCH_V];
  int k_key[NCH_K], k_lds[NCH_K], v_lds[NCH_V], v_kc[NCH_V];
#pragma unroll
  for (int i = 0; i < NCH_K; ++i) {
    const int q = tid + 256 * i;
    const int key = q / KCH, ch = q - key * KCH;
    k_key[i] = key;
    k_lds[i] = (q < KT * KCH) ? (key * KSLOTS + ch) * 16 : -1;
    koff[i] = (q < KT * KCH && ch < D8) ? (unsigned)(key * p.ldk * 2 + ch * 16) : kInv;
  }
#pragma unroll
  for (int i = 0; i < NCH_V; ++i) {
    const int q = tid + 256 * i;
    const int d = q / VCH, kc = q - d * VCH;
    v_kc[i] = kc;
    // chunk kc = keys 8kc .. 8kc+7 of 16-key group kc >> 1: its two 4-key pieces go to 8-byte slots (kc & 1) and 2 + (kc & 1)
    v_lds[i] = (q < DV * VCH && d < D) ? d * VROW + (kc >> 1) * 32 + (kc & 1) * 8 : -1;
    voff[i] = (q < DV * VCH && d < D) ? (unsigned)(d * p.ldvt * 2 + kc * 16) : kInv;
  }
  // rows D .. DV-1 of the V^T tile never change: ones (denominator row, when ONES) / zeros -- in every buffer
  for (int q = tid; q < DV * VCH; q += 256) {
    const int d = q / VCH, kc = q - d * VCH;
    if (d >= D) {
      const unsigned fill = (ONES && d == D) ? 0x3F803F80u : 0u;
#pragma unroll
      for (int bf = 0; bf < (DB ? 2 : 1); ++bf)
        *reinterpret_cast<u32x4*>(smem + bf * BUF + K_BYTES + d * VROW + kc * 16) = u32x4{fill, fill, fill, fill};
    }
  }
  u32x4 kreg[NCH_K], vreg[NCH_V];

  auto load_tile = [&](int key0) __attribute__((always_inline)) {
    const bool tail = key0 + KT > p.nk;
    const unsigned sk = (unsigned)(key0 * p.ldk * 2), sv = (unsigned)(key0 * 2);
#pragma unroll
    for (int i = 0; i < NCH_K; ++i) {
      unsigned o = koff[i];
      if (tail && key0 + k_key[i] >= p.nk) o = kInv;
      kreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsk, (int)o, (int)sk, 0));
    }
#pragma unroll
    for (int i = 0; i < NCH_V; ++i) {
      unsigned o = voff[i];
      if (tail && key0 + v_kc[i] * 8 >= p.nk) o = kInv;
      vreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsv, (int)o, (int)sv, 0));
    }
  };
  auto store_tile = [&](int key0, unsigned char* buf) __attribute__((always_inline)) {
    const bool tail = key0 + KT > p.nk;
    unsigned char* ksm = buf;
    unsigned char* vsm = buf + K_BYTES;
#pragma unroll
    for (int i = 0; i < NCH_K; ++i)
      if (k_lds[i] >= 0) *reinterpret_cast<u32x4*>(ksm + k_lds[i]) = kreg[i];
#pragma unroll
    for (int i = 0; i < NCH_V; ++i) {
      if (v_lds[i] >= 0) {
        u32x4 v = vreg[i];
        if (tail) {
          const int nvalid = p.nk - (key0 + v_kc[i] * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned keep = ((2 * e < nvalid) ? 0x0000ffffu : 0u) | ((2 * e + 1 < nvalid) ? 0xffff0000u : 0u);
            v[e] &= keep;
          }
        }
        u32x2* dst = reinterpret_cast<u32x2*>(vsm + v_lds[i]);
        dst[0] = u32x2{v.x, v.y};
        dst[2] = u32x2{v.z, v.w};
      }
    }
  };

  f32x16 acc_o[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc_o[nb][i] = 0.f;
  f32x16 negm;                                      // -m (reference level of this lane's query), splat
#pragma unroll
  for (int i = 0; i < 16; ++i) negm[i] = 0.f;
  float l_run = 0.f;                                // VALU row-sum (only when !ONES)

  const int ntiles = (p.nk + KT - 1) / KT;
  load_tile(0);
  if (DB) {
    store_tile(0, smem);
    if (ntiles > 1) load_tile(KT);
    __syncthreads();
  }
  for (int t = 0; t < ntiles; ++t) {
    const int key0 = t * KT;
    const unsigned char* buf;
    if (DB) {
      buf = smem + (t & 1) * BUF;
      if (t + 1 < ntiles) {
        store_tile(key0 + KT, smem + ((t + 1) & 1) * BUF);   // read last in iteration t-1: every wave has passed its barrier
        if (t + 2 < ntiles) load_tile(key0 + 2 * KT);
      }
    } else {
      buf = smem;
      __syncthreads();
      store_tile(key0, smem);
      __syncthreads();
      if (t + 1 < ntiles) load_tile(key0 + KT);
    }
    const unsigned char* ksm = buf;
    const unsigned char* vsm = buf + K_BYTES;

    // ---- S'^T = K Q^T - m: the reference level is the MFMA's C operand ----
    f32x16 acc_s[NKB];
    u32x4 kf[2][KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) kf[0][s] = *reinterpret_cast<const u32x4*>(ksm + (r * KSLOTS + 2 * s + h) * 16);
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (kb + 1 < NKB) {
#pragma unroll
        for (int s = 0; s < KS; ++s)
          kf[(kb + 1) & 1][s] = *reinterpret_cast<const u32x4*>(ksm + (((kb + 1) * 32 + r) * KSLOTS + 2 * s + h) * 16);
      }
#pragma unroll
      for (int s = 0; s < KS; ++s)
        acc_s[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[kb & 1][s]), __builtin_bit_cast(bf16x8, qf[s]),
                                                            s == 0 ? negm : acc_s[kb], 0, 0, 0);
    }
    if (key0 + KT > p.nk || p.causal) {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = key0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (key >= p.nk || (p.causal && key > qi)) acc_s[kb][i] = -INFINITY;
        }
    }
    // ---- P = exp2(S'), packed; OR of the bf16 patterns tells whether any p reached 2.0 ----
    unsigned pk[NKB][8];
    unsigned orv = 0u;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float p0 = __builtin_amdgcn_exp2f(acc_s[kb][2 * j]), p1 = __builtin_amdgcn_exp2f(acc_s[kb][2 * j + 1]);
        if (!ONES) psum += p0 + p1;
        pk[kb][j] = pack2(p0, p1);
        orv |= pk[kb][j];
      }
    if (__any(t == 0 || (orv & 0x40004000u) != 0u)) {
      // slow path (wave-uniform): true row maximum of this tile relative to the current level
      float mx = acc_s[0][0];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, acc_s[kb][i]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float delta = t == 0 ? mx + BIAS : fmaxf(mx + BIAS, 0.f);   // the level only ever rises after tile 0
#pragma unroll
      for (int i = 0; i < 16; ++i) negm[i] += -delta;
      if (t != 0) {
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        if (!ONES) l_run *= alpha;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc_o[nb][i] *= alpha;
      }
      psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float p0 = __builtin_amdgcn_exp2f(acc_s[kb][2 * j] - delta), p1 = __builtin_amdgcn_exp2f(acc_s[kb][2 * j + 1] - delta);
          if (!ONES) psum += p0 + p1;
          pk[kb][j] = pack2(p0, p1);
        }
    }
    if (!ONES) l_run += psum;

    // ---- O^T += V^T P^T ----
#pragma unroll
    for (int ks = 0; ks < 2 * NKB; ++ks) {
      const int kb = ks >> 1, half = ks & 1;
      const u32x4 pf = {pk[kb][4 * half + 0], pk[kb][4 * half + 1], pk[kb][4 * half + 2], pk[kb][4 * half + 3]};
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const u32x4 vf = *reinterpret_cast<const u32x4*>(vsm + (nb * 32 + r) * VROW + ks * 32 + 16 * h);
        acc_o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf), acc_o[nb], 0, 0, 0);
      }
    }
    if (DB) __syncthreads();      // tile t consumed by everyone; tile t+1 visible to everyone
  }

  float l_tot;
  if (ONES) {
    float lsel = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (32 * nb + 8 * g == D) lsel = acc_o[nb][4 * g];
    l_tot = __shfl(lsel, r, 64);
  } else {
    l_tot = l_run + __shfl_xor(l_run, 32, 64);
  }
  const float inv = 1.0f / l_tot;
  if (qi < p.nq) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * nb + 8 * g + 4 * h;
        if (d < D) {
          float v[4] = {acc_o[nb][4 * g + 0] * inv, acc_o[nb][4 * g + 1] * inv, acc_o[nb][4 * g + 2] * inv,
                        acc_o[nb][4 * g + 3] * inv};
          Elem<bf16_t>::store4(O + (long long)qi * p.ldo + d, v);
        }
      }
  }
}

// ---- v3: v2's arithmetic, software-pipelined inside the wave ------------------------------------------------------
// v2 runs QK^T (matrix pipe only), the exponentials (VALU only) and PV (matrix pipe only) one after the other and
// leaves the overlap to whichever other wave shares the SIMD; measured, the two waves of a SIMD overlap poorly
// (46 % MFMA busy with 40 % of the VALU work removed).  Here ONE wave carries two independent instruction streams
// through every 64-key tile step t:
//     matrix pipe:  O += V(t-1) P(t-1)        and        S(t+1) = K(t+1) Q - m
//     VALU:         P(t) = exp2(S(t)), packed, OR-checked
// (S and P ping-pong between two register sets, the loop is unrolled by two so every index is static), dealt by
// hand into one MFMA | LDS read | a few VALU issue pattern and pinned with sched_barrier.  If the check moves the
// reference level (rare), S(t+1) -- computed against the old level -- takes the same delta.  K / V^T tiles live in a
// 4-slot LDS ring written two steps ahead (global -> registers one step earlier still): one barrier per tile, no tile
// is overwritten before the step after its last read.  64-key tiles keep the whole wave below 256 registers, so
// two workgroups still share a CU.
// ABL (diagnostics, `make ABLATION=1` + SASPA_ATTN_ABLATE only; 0 in the shipped library): 1 no exponentials, 2 no MFMAs,
// 4 no LDS fragment reads, 8 no staging (global loads / LDS stores / barriers), 16 stamps (s_memtime / s_memrealtime of
// the tile loop of every workgroup's thread 0 behind the output tensor: tools/attn_ablate.py)
#ifndef SASPA_ATTN_PF
#define SASPA_ATTN_PF 2
#endif
// R (ring slots): 4 = one workgroup barrier per 64-key step; 6 = ONE BARRIER PER TWO STEPS (round 6): at every even step t, after the
// barrier, tiles t+3 and t+4 go from the two staging register sets into slots (t+3) % 6 and (t+4) % 6 and tiles t+5 / t+6 are
// requested.  Behind barrier(t) every wave is in step t or t+1, i.e. reads V(t-1), K(t+1), V(t), K(t+2): four live tiles + two
// being written = six slots; a tile stored in step t is first read (K(t+3)) in step t+2, behind barrier(t+2).  The eight waves
// re-align half as often: the step measured 1 717 cycles against ~1 100 of issue work, much of the rest barrier skew.
template <int KS, int NB, bool ONES, int NW = 8, int ABL = 0, bool RM = false, int R = 4>
__global__ __launch_bounds__(64 * NW, 2) void flash_attn_v3_kernel(const SaspaAttnParams p) {
  static_assert(R == 4 || R == 6, "ring of 4 (barrier per step) or 6 slots (barrier per two steps)");
  // RM: V row-major, read through ds_read_b64_tr_b16 (see lds_read_tr16 at the top of the file)
  // NW waves of 32 queries per workgroup: 8 waves halve the K / V^T bytes every query block pulls through L1 / LDS
  // (the staging is the largest single cost of the loop: tools/attn_ablate.py, profiles/r3_attn_ablation.txt)
  constexpr int NT = 64 * NW;
  constexpr int KT = 64, NKB = 2, VCH = KT / 8;
  constexpr int KSLOTS = (2 * KS) | 1;
  constexpr int KCH = 2 * KS;
  constexpr int DV = NB * 32;
  constexpr int VROW = KT * 2 + 16;            // 144 B = 9 slots of 16 B (odd): conflict-free ds_read_b128
  constexpr int RROW = rm_pitch(DV), DCH = DV / 8;
  constexpr int K_BYTES = KT * KSLOTS * 16;
  constexpr int V_BYTES = RM ? KT * RROW : DV * VROW;
  constexpr int BUF = K_BYTES + V_BYTES;
  constexpr int NCH_K = (KT * KCH + NT - 1) / NT;
  constexpr int NCH_V = RM ? (KT * DCH + NT - 1) / NT : (DV * VCH + NT - 1) / NT;
  constexpr float BIAS = 8.0f;
  __shared__ __attribute__((aligned(16))) unsigned char smem[R * BUF];
  auto slot = [](int t) __attribute__((always_inline)) -> int { return R == 4 ? (t & 3) : (int)((unsigned)t % 6u); };

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int gx = gridDim.x, nbh = gridDim.y * gridDim.z;
  const int lin = blockIdx.x + gx * (blockIdx.y + gridDim.y * blockIdx.z);
  const int grp = lin / (8 * gx), rr = lin - grp * 8 * gx;
  const int gsz = min(8, nbh - grp * 8);
  const int bh = grp * 8 + rr % gsz, xq = rr / gsz;
  const int head = bh % (int)gridDim.y, b = bh / (int)gridDim.y;
  const int D = p.D, D8 = D >> 3;
  const int q0 = xq * (32 * NW) + wave * 32;
  const int qi = q0 + r;

  const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + b * p.sqb + head * D;
  const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.k) + b * p.skb + head * D;
  const bf16_t* VT = RM ? reinterpret_cast<const bf16_t*>(p.vt) + b * p.svb + head * D
                        : reinterpret_cast<const bf16_t*>(p.vt) + b * p.svb + (long long)head * D * p.ldvt;
  bf16_t* O = reinterpret_cast<bf16_t*>(p.o) + b * p.sob + head * D;

  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int d = 16 * s + 8 * h;
    qf[s] = (qi < p.nq && d < D) ? *reinterpret_cast<const u32x4*>(Q + (long long)qi * p.ldq + d) : zero4;
  }

  const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Kp), (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(VT), (short)0, 0x7fffffff, 0x00020000);
  constexpr unsigned kInv = 0x80000000u;
  unsigned koff[NCH_K], voff[NCH_V];
  int k_key[NCH_K], k_lds[NCH_K], v_lds[NCH_V], v_kc[NCH_V];
#pragma unroll
  for (int i = 0; i < NCH_K; ++i) {
    const int q = tid + NT * i;
    const int key = q / KCH, ch = q - key * KCH;
    k_key[i] = key;
    k_lds[i] = (q < KT * KCH) ? (key * KSLOTS + ch) * 16 : -1;
    koff[i] = (q < KT * KCH && ch < D8) ? (unsigned)(key * p.ldk * 2 + ch * 16) : kInv;
  }
  if constexpr (RM) {
#pragma unroll
    for (int i = 0; i < NCH_V; ++i) {
      const int q = tid + NT * i;
      const int key = q / DCH, ch = q - key * DCH;
      v_kc[i] = key;                         // (RM: the tile row = key)
      v_lds[i] = (q < KT * DCH && ch < D8) ? key * RROW + ch * 16 : -1;
      voff[i] = (q < KT * DCH && ch < D8) ? (unsigned)(key * p.ldvt * 2 + ch * 16) : kInv;
    }
    for (int q = tid; q < KT * DCH; q += NT) {
      const int key = q / DCH, ch = q - key * DCH;
      if (ch >= D8) {
#pragma unroll
        for (int bf = 0; bf < R; ++bf)
          *reinterpret_cast<u32x4*>(smem + bf * BUF + K_BYTES + key * RROW + ch * 16) = u32x4{(ONES && ch == D8) ? 0x00003F80u : 0u, 0u, 0u, 0u};
      }
    }
  } else {
#pragma unroll
  for (int i = 0; i < NCH_V; ++i) {
    const int q = tid + NT * i;
    const int d = q / VCH, kc = q - d * VCH;
    v_kc[i] = kc;
    v_lds[i] = (q < DV * VCH && d < D) ? d * VROW + (kc >> 1) * 32 + (kc & 1) * 8 : -1;   // key permutation: see v2
    voff[i] = (q < DV * VCH && d < D) ? (unsigned)(d * p.ldvt * 2 + kc * 16) : kInv;
  }
  for (int q = tid; q < DV * VCH; q += NT) {
    const int d = q / VCH, kc = q - d * VCH;
    if (d >= D) {
      const unsigned fill = (ONES && d == D) ? 0x3F803F80u : 0u;
#pragma unroll
      for (int bf = 0; bf < R; ++bf)
        *reinterpret_cast<u32x4*>(smem + bf * BUF + K_BYTES + d * VROW + kc * 16) = u32x4{fill, fill, fill, fill};
    }
  }
  }
  const int tr_base = (4 * (lane >> 5) + ((lane & 15) >> 2)) * RROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
  // two staging register sets: tile t+4 is requested at step t and written to LDS at step t+2 (one step of
  // flight time is less than the L2 latency under load)
  u32x4 kregA[NCH_K], vregA[NCH_V], kregB[NCH_K], vregB[NCH_V];
  auto load_tile = [&](int key0, u32x4 (&kreg)[NCH_K], u32x4 (&vreg)[NCH_V]) __attribute__((always_inline)) {
    const bool tail = key0 + KT > p.nk;
    const unsigned sk = (unsigned)(key0 * p.ldk * 2), sv = (unsigned)(key0 * 2);
#pragma unroll
    for (int i = 0; i < NCH_K; ++i) {
      unsigned o = koff[i];
      if (tail && key0 + k_key[i] >= p.nk) o = kInv;
      kreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsk, (int)o, (int)sk, 0));
    }
#pragma unroll
    for (int i = 0; i < NCH_V; ++i) {
      unsigned o = voff[i];
      if (RM) {
        if (tail && key0 + v_kc[i] >= p.nk) o = kInv;                    // key rows >= nk: zeros
        vreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsv, (int)o, (int)(key0 * p.ldvt * 2), 0));
      } else {
        if (tail && key0 + v_kc[i] * 8 >= p.nk) o = kInv;
        vreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsv, (int)o, (int)sv, 0));
      }
    }
  };
  auto store_tile = [&](int key0, unsigned char* buf, const u32x4 (&kreg)[NCH_K], const u32x4 (&vreg)[NCH_V]) __attribute__((always_inline)) {
    const bool tail = key0 + KT > p.nk;
    unsigned char* ksm = buf;
    unsigned char* vsm = buf + K_BYTES;
#pragma unroll
    for (int i = 0; i < NCH_K; ++i)
      if (k_lds[i] >= 0) *reinterpret_cast<u32x4*>(ksm + k_lds[i]) = kreg[i];
#pragma unroll
    for (int i = 0; i < NCH_V; ++i) {
      if (RM) {
        if (v_lds[i] >= 0) *reinterpret_cast<u32x4*>(vsm + v_lds[i]) = vreg[i];
        continue;
      }
      if (v_lds[i] >= 0) {
        u32x4 v = vreg[i];
        if (tail) {
          const int nvalid = p.nk - (key0 + v_kc[i] * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned keep = ((2 * e < nvalid) ? 0x0000ffffu : 0u) | ((2 * e + 1 < nvalid) ? 0xffff0000u : 0u);
            v[e] &= keep;
          }
        }
        u32x2* dst = reinterpret_cast<u32x2*>(vsm + v_lds[i]);
        dst[0] = u32x2{v.x, v.y};
        dst[2] = u32x2{v.z, v.w};
      }
    }
  };

  f32x16 acc_o[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc_o[nb][i] = 0.f;
  f32x16 negm;
#pragma unroll
  for (int i = 0; i < 16; ++i) negm[i] = 0.f;
  float l_run = 0.f;

  const int ntiles = (p.nk + KT - 1) / KT;

  // S(t) = K(t) Q - m from ring slot t & 3, with the tail mask
  auto qk_tile = [&](int t, f32x16 (&S)[NKB]) __attribute__((always_inline)) {
    const unsigned char* ksm = smem + slot(t) * BUF;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const u32x4 kf = *reinterpret_cast<const u32x4*>(ksm + ((kb * 32 + r) * KSLOTS + 2 * s + h) * 16);
        S[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[s]),
                                                        s == 0 ? negm : S[kb], 0, 0, 0);
      }
  };
  auto mask_tail = [&](int t, f32x16 (&S)[NKB]) __attribute__((always_inline)) {
    const int key0 = t * KT;
    if (key0 + KT > p.nk) {
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = key0 + kb * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (key >= p.nk) S[kb][i] = -INFINITY;
        }
    }
  };
  auto pv_tile = [&](int t, const unsigned (&P)[NKB][8]) __attribute__((always_inline)) {
    const unsigned char* vsm = smem + slot(t) * BUF + K_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2 * NKB; ++ks) {
      const int kb = ks >> 1, half = ks & 1;
      const u32x4 pf = {P[kb][4 * half + 0], P[kb][4 * half + 1], P[kb][4 * half + 2], P[kb][4 * half + 3]};
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        u32x4 vf;
        if (RM) {
          const unsigned char* blk = vsm + tr_base + ks * 16 * RROW + nb * 64;
          const u32x2 lo = lds_read_tr16(blk), hi = lds_read_tr16(blk + 8 * RROW);
          vf = u32x4{lo.x, lo.y, hi.x, hi.y};
        } else {
          vf = *reinterpret_cast<const u32x4*>(vsm + (nb * 32 + r) * VROW + ks * 32 + 16 * h);
        }
        acc_o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf), acc_o[nb], 0, 0, 0);
      }
    }
  };

  // one step: exp(t) on the VALU beside PV(t-1) + QK(t+1) on the matrix pipe.  has_prev / has_next are compile-time
  // (first / last steps are peeled) so that the two streams sit in ONE basic block the scheduler can interleave.
  auto step = [&](auto hp, auto hn, int t, f32x16 (&Sc)[NKB], f32x16 (&Sn)[NKB], const unsigned (&Pp)[NKB][8],
                  unsigned (&Pc)[NKB][8], u32x4 (&kreg)[NCH_K], u32x4 (&vreg)[NCH_V]) __attribute__((always_inline)) {
    constexpr bool has_prev = decltype(hp)::value, has_next = decltype(hn)::value;
    if constexpr (R == 4) {
      if (t > 0 && !(ABL & 8)) __syncthreads();        // tile t+1 is in LDS for everyone; everyone is done with tile t-2
      if (t + 2 < ntiles && !(ABL & 8)) {
        store_tile((t + 2) * KT, smem + ((t + 2) & 3) * BUF, kreg, vreg);
        if (t + 4 < ntiles) load_tile((t + 4) * KT, kreg, vreg);
      }
    } else if ((t & 1) == 0 && !(ABL & 8)) {
      // even steps only (see the template comment): set B carries the odd tiles, set A the even ones
      if (t > 0) __syncthreads();
      if (t + 3 < ntiles) store_tile((t + 3) * KT, smem + slot(t + 3) * BUF, kregB, vregB);
      if (t + 4 < ntiles) store_tile((t + 4) * KT, smem + slot(t + 4) * BUF, kregA, vregA);
      if (t + 5 < ntiles) load_tile((t + 5) * KT, kregB, vregB);
      if (t + 6 < ntiles) load_tile((t + 6) * KT, kregA, vregA);
    }
    // only the last tile can hold keys >= nk, and the last step is a peeled one: the steady step stays branch-free
    if constexpr (has_prev && !has_next) mask_tail(t, Sc);
    unsigned orv = 0u;
    float psum = 0.f;
    auto exp_unit = [&](int u) __attribute__((always_inline)) {      // two scores -> one packed register
      const int kb = u >> 3, j = u & 7;
      const float p0 = (ABL & 1) ? Sc[kb][2 * j] : __builtin_amdgcn_exp2f(Sc[kb][2 * j]);
      const float p1 = (ABL & 1) ? Sc[kb][2 * j + 1] : __builtin_amdgcn_exp2f(Sc[kb][2 * j + 1]);
      if (!ONES) psum += p0 + p1;
      Pc[kb][j] = pack2(p0, p1);
      orv |= (ABL & 1) ? 0u : Pc[kb][j];
    };
    if constexpr (has_prev && has_next) {
      // The steady step, dealt by hand into ONE issue order (sched_barrier(0) pins it; the sched_group_barrier form left
      // the MFMAs clustered at both ends): per MFMA slot [MFMA i | LDS read of the operand of MFMA i+2 | its share of the
      // 8*NKB exponential units].  MFMA i: PV of tile t-1 for i < NPV (ks = i / NB, nb = i % NB), then QK of tile t+1.
      constexpr int NPV = 2 * NKB * NB, NM = NPV + NKB * KS, NU = 8 * NKB;
      const unsigned char* vsm = smem + slot(t - 1) * BUF + K_BYTES;
      const unsigned char* ksm = smem + slot(t + 1) * BUF;
      auto frag = [&](int i) __attribute__((always_inline)) -> u32x4 {
        if (ABL & 4) return u32x4{(unsigned)i, 0x3f803f80u, (unsigned)lane, 0x3f803f80u};
        if (i < NPV) {
          if (RM) {
            const unsigned char* blk = vsm + tr_base + (i / NB) * 16 * RROW + (i % NB) * 64;
            const u32x2 lo = lds_read_tr16(blk), hi = lds_read_tr16(blk + 8 * RROW);
            return u32x4{lo.x, lo.y, hi.x, hi.y};
          }
          return *reinterpret_cast<const u32x4*>(vsm + ((i % NB) * 32 + r) * VROW + (i / NB) * 32 + 16 * h);
        }
        const int q = i - NPV, kb = q / KS, sx = q - kb * KS;
        return *reinterpret_cast<const u32x4*>(ksm + ((kb * 32 + r) * KSLOTS + 2 * sx + h) * 16);
      };
      // operand fragments are read PF MFMA slots ahead of their use (SASPA_ATTN_PF, default 2: round 3's choice)
      constexpr int PF = SASPA_ATTN_PF;
      u32x4 fr[PF + 1];
#pragma unroll
      for (int i = 0; i < PF; ++i) fr[i] = frag(i);
      int u = 0;
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        if (i + PF < NM) fr[(i + PF) % (PF + 1)] = frag(i + PF);
        if (i < NPV) {
          const int ks = i / NB, nb = i % NB, kb = ks >> 1, half = ks & 1;
          const u32x4 pf = {Pp[kb][4 * half + 0], Pp[kb][4 * half + 1], Pp[kb][4 * half + 2], Pp[kb][4 * half + 3]};
          if (ABL & 2) { asm volatile("" ::"v"(fr[i % (PF + 1)]), "v"(pf)); }
          else acc_o[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[i % (PF + 1)]), __builtin_bit_cast(bf16x8, pf), acc_o[nb], 0, 0, 0);
        } else {
          const int q = i - NPV, kb = q / KS, sx = q - kb * KS;
          if (ABL & 2) { asm volatile("" ::"v"(fr[i % (PF + 1)])); if (sx == 0) Sn[kb] = negm; }
          else Sn[kb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[i % (PF + 1)]), __builtin_bit_cast(bf16x8, qf[sx]),
                                                                sx == 0 ? negm : Sn[kb], 0, 0, 0);
        }
        const int nu = NU / NM + (i < NU % NM ? 1 : 0);
#pragma unroll
        for (int k = 0; k < nu; ++k) exp_unit(u++);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      if constexpr (has_prev) pv_tile(t - 1, Pp);
      if constexpr (has_next) qk_tile(t + 1, Sn);
#pragma unroll
      for (int u = 0; u < 8 * NKB; ++u) exp_unit(u);
    }
    if (__any(t == 0 || (orv & 0x40004000u) != 0u)) {
      float mx = Sc[0][0];
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, Sc[kb][i]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float delta = t == 0 ? mx + BIAS : fmaxf(mx + BIAS, 0.f);
#pragma unroll
      for (int i = 0; i < 16; ++i) negm[i] += -delta;
      if (t != 0) {
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        if (!ONES) l_run *= alpha;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc_o[nb][i] *= alpha;
      }
      if (has_next) {                                // S(t+1) was taken against the old level
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
          for (int i = 0; i < 16; ++i) Sn[kb][i] -= delta;
      }
      psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float p0 = __builtin_amdgcn_exp2f(Sc[kb][2 * j] - delta), p1 = __builtin_amdgcn_exp2f(Sc[kb][2 * j + 1] - delta);
          if (!ONES) psum += p0 + p1;
          Pc[kb][j] = pack2(p0, p1);
        }
    }
    if (!ONES) l_run += psum;
  };

  // prologue: tiles 0 and 1 in LDS, tile 2 in registers, S(0)
  load_tile(0, kregA, vregA);
  if (ntiles > 1) load_tile(KT, kregB, vregB);
  store_tile(0, smem, kregA, vregA);
  if (ntiles > 2) load_tile(2 * KT, kregA, vregA);     // set A: even tiles (stored by the even steps)
  if (ntiles > 1) store_tile(KT, smem + BUF, kregB, vregB);
  if (ntiles > 3) load_tile(3 * KT, kregB, vregB);     // set B: odd tiles
  if constexpr (R == 6) {                              // tile 2 too: step 1 reads K(2) without a barrier in between
    if (ntiles > 2) store_tile(2 * KT, smem + 2 * BUF, kregA, vregA);
    if (ntiles > 4) load_tile(4 * KT, kregA, vregA);
  }
  __syncthreads();
  f32x16 S0[NKB], S1[NKB];
  unsigned P0[NKB][8], P1[NKB][8];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int j = 0; j < 8; ++j) P0[kb][j] = P1[kb][j] = 0u;
  qk_tile(0, S0);
  mask_tail(0, S0);
  using T_ = std::true_type;
  using F_ = std::false_type;
#ifdef SASPA_ATTN_YOUNG_PRIO
  // A/B (round 5): static priority for the second-dispatched half of the workgroup, which loses every issue arbitration against
  // its SIMD partners by age (MI355X_MICROARCH.md, "Two waves per SIMD", item 4)
  if (wave >= NW / 2) __builtin_amdgcn_s_setprio(SASPA_ATTN_YOUNG_PRIO);
#endif
  unsigned long long st0 = 0, sr0 = 0;
  if (ABL & 16) { st0 = __builtin_amdgcn_s_memtime(); sr0 = __builtin_amdgcn_s_memrealtime(); }
  if (ntiles > 1) step(F_{}, T_{}, 0, S0, S1, P1, P0, kregA, vregA);
  else step(F_{}, F_{}, 0, S0, S1, P1, P0, kregA, vregA);
  for (int t = 1; t + 1 < ntiles; t += 2) {          // steady steps 1 .. ntiles-2
    step(T_{}, T_{}, t, S1, S0, P0, P1, kregB, vregB);
    if (t + 2 < ntiles) step(T_{}, T_{}, t + 1, S0, S1, P1, P0, kregA, vregA);
  }
  if (ntiles > 1) {
    const int last = ntiles - 1;
    if (last & 1) step(T_{}, F_{}, last, S1, S0, P0, P1, kregB, vregB);
    else step(T_{}, F_{}, last, S0, S1, P1, P0, kregA, vregA);
  }
  if ((ntiles - 1) & 1) pv_tile(ntiles - 1, P1);
  else pv_tile(ntiles - 1, P0);
  if ((ABL & 16) && tid == 0) {
    // diagnostics: the tool's output allocation extends 16 bytes per workgroup beyond batch * sob elements
    unsigned long long* dbg = reinterpret_cast<unsigned long long*>(reinterpret_cast<bf16_t*>(p.o) + (long long)p.batch * p.sob) + 2 * lin;
    dbg[0] = __builtin_amdgcn_s_memtime() - st0;
    dbg[1] = __builtin_amdgcn_s_memrealtime() - sr0;
  }

  float l_tot;
  if (ONES) {
    float lsel = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (32 * nb + 8 * g == D) lsel = acc_o[nb][4 * g];
    l_tot = __shfl(lsel, r, 64);
  } else {
    l_tot = l_run + __shfl_xor(l_run, 32, 64);
  }
  const float inv = 1.0f / l_tot;
  if (qi < p.nq) {
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = 32 * nb + 8 * g + 4 * h;
        if (d < D) {
          float v[4] = {acc_o[nb][4 * g + 0] * inv, acc_o[nb][4 * g + 1] * inv, acc_o[nb][4 * g + 2] * inv,
                        acc_o[nb][4 * g + 3] * inv};
          Elem<bf16_t>::store4(O + (long long)qi * p.ldo + d, v);
        }
      }
  }
}

// ---- v4 (round 6): v3's pipeline with SIXTY-FOUR queries per wave and 32-key half steps ------------------------------------
// v3's ablations add up linearly -- removing the LDS fragment reads saves 89 us of 508, the staging (global -> LDS, barriers) 126 --
// because an in-order wave pays for everything it issues.  Both are per-QUERY-BLOCK costs: a K or V^T fragment read from LDS feeds ONE
// MFMA (32 queries), a staged tile serves 256 queries.  Here a wave owns two query blocks (q0 + r and q0 + 32 + r on lane r), so every
// fragment feeds TWO MFMAs and a workgroup of eight waves shares each staged tile among 512 queries; a step covers 32 keys (one key
// block) so that S / P of two query blocks are the registers v3 spends on two key blocks of one:
//     matrix pipe:  O[qb] += V(ht-1) P[qb](ht-1)  (2 NB fragments x 2)   and   S[qb](ht+1) = K(ht+1) Q[qb]  (KS fragments x 2)
//     VALU:         P[qb](ht) = exp2(S[qb](ht)), packed, OR-checked
// 14 MFMAs and 32 exponentials per step as in v3, but 7 LDS fragment reads instead of 14, one barrier per TWO steps (tiles stay 64 keys)
// and half the staging per query.  The registers come from two places: one staging set instead of two (a tile requested at its
// predecessor's store has the same two steps of flight), and NO accumulator-init vector for the reference level -- the level rides
// in the head dimension's padding: K's first pad element is 1, Q's is -m (bf16-exact, per query block), so S = K Q - m leaves the
// MFMA with a literal-zero C operand.  That needs D < 16 KS (d = 40: dims 40 .. 47 are padding) and the ones row of V^T (D < 32 NB).
template <int KS, int NB, int NW = 8>
__global__ __launch_bounds__(64 * NW, 2) void flash_attn_v4_kernel(const SaspaAttnParams p) {
  constexpr int NT = 64 * NW;
  constexpr int KT = 64, VCH = KT / 8;
  constexpr int KSLOTS = (2 * KS) | 1;
  constexpr int KCH = 2 * KS;
  constexpr int DV = NB * 32;
  constexpr int VROW = KT * 2 + 16;
  constexpr int K_BYTES = KT * KSLOTS * 16;
  constexpr int V_BYTES = DV * VROW;
  constexpr int BUF = K_BYTES + V_BYTES;
  constexpr int NCH_K = (KT * KCH + NT - 1) / NT;
  constexpr int NCH_V = (DV * VCH + NT - 1) / NT;
  constexpr float BIAS = 8.0f;
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int gx = gridDim.x, nbh = gridDim.y * gridDim.z;
  const int lin = blockIdx.x + gx * (blockIdx.y + gridDim.y * blockIdx.z);
  const int grp = lin / (8 * gx), rr = lin - grp * 8 * gx;
  const int gsz = min(8, nbh - grp * 8);
  const int bh = grp * 8 + rr % gsz, xq = rr / gsz;
  const int head = bh % (int)gridDim.y, b = bh / (int)gridDim.y;
  constexpr int D = 16 * KS - 8, D8 = D >> 3;          // exactly one pad chunk: d = 40 for KS = 3 (the launcher checks p.D)
  constexpr int sp = KS - 1, hp = 1;                   // the first pad element of the head dimension: fragment sp, lane half hp
  const int q0 = xq * (64 * NW) + wave * 64;

  const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q) + b * p.sqb + head * D;
  const bf16_t* Kp = reinterpret_cast<const bf16_t*>(p.k) + b * p.skb + head * D;
  const bf16_t* VT = reinterpret_cast<const bf16_t*>(p.vt) + b * p.svb + (long long)head * D * p.ldvt;
  bf16_t* O = reinterpret_cast<bf16_t*>(p.o) + b * p.sob + head * D;

  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  u32x4 qf[2][KS];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const int d = 16 * s + 8 * h, qi = q0 + 32 * qb + r;
      qf[qb][s] = (qi < p.nq && d < D) ? *reinterpret_cast<const u32x4*>(Q + (long long)qi * p.ldq + d) : zero4;
    }

  const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Kp), (short)0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsv = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(VT), (short)0, 0x7fffffff, 0x00020000);
  constexpr unsigned kInv = 0x80000000u;
  unsigned koff[NCH_K], voff[NCH_V];
  int k_key[NCH_K], k_lds[NCH_K], v_lds[NCH_V], v_kc[NCH_V];
  bool k_one[NCH_K];
#pragma unroll
  for (int i = 0; i < NCH_K; ++i) {
    const int q = tid + NT * i;
    const int key = q / KCH, ch = q - key * KCH;
    k_key[i] = key;
    k_lds[i] = (q < KT * KCH) ? (key * KSLOTS + ch) * 16 : -1;
    koff[i] = (q < KT * KCH && ch < D8) ? (unsigned)(key * p.ldk * 2 + ch * 16) : kInv;
    k_one[i] = ch == D8;                               // the chunk whose first element is the constant 1 that multiplies Q's -m
  }
#pragma unroll
  for (int i = 0; i < NCH_V; ++i) {
    const int q = tid + NT * i;
    const int d = q / VCH, kc = q - d * VCH;
    v_kc[i] = kc;
    v_lds[i] = (q < DV * VCH && d < D) ? d * VROW + (kc >> 1) * 32 + (kc & 1) * 8 : -1;   // key permutation: see v2
    voff[i] = (q < DV * VCH && d < D) ? (unsigned)(d * p.ldvt * 2 + kc * 16) : kInv;
  }
  for (int q = tid; q < DV * VCH; q += NT) {
    const int d = q / VCH, kc = q - d * VCH;
    if (d >= D) {
      const unsigned fill = d == D ? 0x3F803F80u : 0u;   // the ones row: O^T row D accumulates the softmax denominator
#pragma unroll
      for (int bf = 0; bf < 4; ++bf)
        *reinterpret_cast<u32x4*>(smem + bf * BUF + K_BYTES + d * VROW + kc * 16) = u32x4{fill, fill, fill, fill};
    }
  }
  u32x4 kreg[NCH_K], vreg[NCH_V];                        // ONE staging set: tile t+3 is requested when tile t+2 is stored
  auto load_tile = [&](int key0) __attribute__((always_inline)) {
    const bool tail = key0 + KT > p.nk;
    const unsigned sk = (unsigned)(key0 * p.ldk * 2), sv = (unsigned)(key0 * 2);
#pragma unroll
    for (int i = 0; i < NCH_K; ++i) {
      unsigned o = koff[i];
      if (tail && key0 + k_key[i] >= p.nk) o = kInv;
      kreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsk, (int)o, (int)sk, 0));
    }
#pragma unroll
    for (int i = 0; i < NCH_V; ++i) {
      unsigned o = voff[i];
      if (tail && key0 + v_kc[i] * 8 >= p.nk) o = kInv;
      vreg[i] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsv, (int)o, (int)sv, 0));
    }
  };
  auto store_tile = [&](int key0, unsigned char* buf) __attribute__((always_inline)) {
    const bool tail = key0 + KT > p.nk;
    unsigned char* ksm = buf;
    unsigned char* vsm = buf + K_BYTES;
#pragma unroll
    for (int i = 0; i < NCH_K; ++i)
      if (k_lds[i] >= 0) *reinterpret_cast<u32x4*>(ksm + k_lds[i]) = k_one[i] ? u32x4{0x00003F80u, 0u, 0u, 0u} : kreg[i];
#pragma unroll
    for (int i = 0; i < NCH_V; ++i) {
      if (v_lds[i] >= 0) {
        u32x4 v = vreg[i];
        if (tail) {
          const int nvalid = p.nk - (key0 + v_kc[i] * 8);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const unsigned keep = ((2 * e < nvalid) ? 0x0000ffffu : 0u) | ((2 * e + 1 < nvalid) ? 0xffff0000u : 0u);
            v[e] &= keep;
          }
        }
        u32x2* dst = reinterpret_cast<u32x2*>(vsm + v_lds[i]);
        dst[0] = u32x2{v.x, v.y};
        dst[2] = u32x2{v.z, v.w};
      }
    }
  };

  f32x16 acc_o[2][NB];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc_o[qb][nb][i] = 0.f;
  float mlev[2] = {0.f, 0.f};                            // the reference levels (bf16-exact), -mlev sits in qf[qb][sp].x of the hp lanes
  f32x16 zero16;
#pragma unroll
  for (int i = 0; i < 16; ++i) zero16[i] = 0.f;

  const int ntiles = (p.nk + KT - 1) / KT;
  const int nh = 2 * ntiles;                             // half steps: ht = 2 * tile + half

  auto kfrag = [&](int ht, int s) __attribute__((always_inline)) -> u32x4 {
    return *reinterpret_cast<const u32x4*>(smem + ((ht >> 1) & 3) * BUF + (((ht & 1) * 32 + r) * KSLOTS + 2 * s + h) * 16);
  };
  auto vfrag = [&](int ht, int j, int nb) __attribute__((always_inline)) -> u32x4 {
    return *reinterpret_cast<const u32x4*>(smem + ((ht >> 1) & 3) * BUF + K_BYTES + (nb * 32 + r) * VROW + (2 * (ht & 1) + j) * 32 + 16 * h);
  };
  auto qk_half = [&](int ht, f32x16 (&S)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      const u32x4 kf = kfrag(ht, s);
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
        S[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf), __builtin_bit_cast(bf16x8, qf[qb][s]),
                                                        s == 0 ? zero16 : S[qb], 0, 0, 0);
    }
  };
  auto mask_tail = [&](int ht, f32x16 (&S)[2]) __attribute__((always_inline)) {
    const int key0 = ht * 32;
    if (key0 + 32 > p.nk) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int key = key0 + (i & 3) + 8 * (i >> 2) + 4 * h;
          if (key >= p.nk) S[qb][i] = -INFINITY;
        }
    }
  };
  auto pv_half = [&](int ht, const unsigned (&P)[2][8]) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const u32x4 vf = vfrag(ht, j, nb);
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
          const u32x4 pf = {P[qb][4 * j + 0], P[qb][4 * j + 1], P[qb][4 * j + 2], P[qb][4 * j + 3]};
          acc_o[qb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, pf), acc_o[qb][nb], 0, 0, 0);
        }
      }
  };

  auto step = [&](auto hpv, auto hnx, auto evn, int ht, f32x16 (&Sc)[2], f32x16 (&Sn)[2], const unsigned (&Pp)[2][8],
                  unsigned (&Pc)[2][8]) __attribute__((always_inline)) {
    constexpr bool has_prev = decltype(hpv)::value, has_next = decltype(hnx)::value, even = decltype(evn)::value;
    if constexpr (even) {
      // a new tile: tile t+1 is in LDS for everyone and everyone is done with tile t-2 (its V half 1 was read in step (t-1, 0))
      const int t = ht >> 1;
      if (t > 0) __syncthreads();
      if (t + 2 < ntiles) {
        store_tile((t + 2) * KT, smem + ((t + 2) & 3) * BUF);
        if (t + 3 < ntiles) load_tile((t + 3) * KT);
      }
    }
    unsigned orv = 0u;
    auto exp_unit = [&](int u) __attribute__((always_inline)) {
      const int qb = u >> 3, j = u & 7;
      const float p0 = __builtin_amdgcn_exp2f(Sc[qb][2 * j]), p1 = __builtin_amdgcn_exp2f(Sc[qb][2 * j + 1]);
      Pc[qb][j] = pack2(p0, p1);
      orv |= Pc[qb][j];
    };
    if constexpr (has_prev && has_next) {
      // the steady step: per fragment [LDS read of fragment f + PF | the TWO MFMAs of fragment f | its share of the 16 exponential units]
      constexpr int NPV = 2 * NB, NF = NPV + KS, NU = 16, PF = 2;
      auto frag = [&](int f) __attribute__((always_inline)) -> u32x4 { return f < NPV ? vfrag(ht - 1, f / NB, f % NB) : kfrag(ht + 1, f - NPV); };
      u32x4 fr[PF + 1];
#pragma unroll
      for (int f = 0; f < PF; ++f) fr[f] = frag(f);
      int u = 0;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        if (f + PF < NF) fr[(f + PF) % (PF + 1)] = frag(f + PF);
        const u32x4 a = fr[f % (PF + 1)];
        if (f < NPV) {
          const int j = f / NB, nb = f % NB;
#pragma unroll
          for (int qb = 0; qb < 2; ++qb) {
            const u32x4 pf = {Pp[qb][4 * j + 0], Pp[qb][4 * j + 1], Pp[qb][4 * j + 2], Pp[qb][4 * j + 3]};
            acc_o[qb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, pf), acc_o[qb][nb], 0, 0, 0);
          }
        } else {
          const int sx = f - NPV;
#pragma unroll
          for (int qb = 0; qb < 2; ++qb)
            Sn[qb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, qf[qb][sx]),
                                                             sx == 0 ? zero16 : Sn[qb], 0, 0, 0);
        }
        const int nu = NU / NF + (f < NU % NF ? 1 : 0);
#pragma unroll
        for (int k = 0; k < nu; ++k) exp_unit(u++);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else {
      if constexpr (has_prev) pv_half(ht - 1, Pp);
      if constexpr (has_next) qk_half(ht + 1, Sn);
#pragma unroll
      for (int u = 0; u < 16; ++u) exp_unit(u);
    }
    if constexpr (has_next) mask_tail(ht + 1, Sn);       // (a uniform test: only the last tile can hold keys >= nk)
    if (__any(ht == 0 || (orv & 0x40004000u) != 0u)) {
#pragma unroll
      for (int qb = 0; qb < 2; ++qb) {
        float mx = Sc[qb][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, Sc[qb][i]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float want = ht == 0 ? mx + BIAS : fmaxf(mx + BIAS, 0.f);
        // the new level must be a bf16 number (it travels as an element of Q): round, then move by the EXACT difference
        const float mnew = __builtin_bit_cast(float, pack2(0.f, mlev[qb] + want) & 0xffff0000u);
        const float delta = mnew - mlev[qb];
        mlev[qb] = mnew;
        if (h == hp) qf[qb][sp].x = pack2(-mnew, 0.f);
        if (ht != 0) {
          const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc_o[qb][nb][i] *= alpha;
        }
        if (has_next) {                                  // S(ht+1) was taken against the old level
#pragma unroll
          for (int i = 0; i < 16; ++i) Sn[qb][i] -= delta;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) Pc[qb][j] = pack2(__builtin_amdgcn_exp2f(Sc[qb][2 * j] - delta), __builtin_amdgcn_exp2f(Sc[qb][2 * j + 1] - delta));
      }
    }
  };

  // prologue: tiles 0 and 1 in LDS, tile 2 in the staging registers, S(0)
  load_tile(0);
  store_tile(0, smem);
  if (ntiles > 1) {
    load_tile(KT);
    store_tile(KT, smem + BUF);
  }
  if (ntiles > 2) load_tile(2 * KT);
  __syncthreads();
  f32x16 S0[2], S1[2];
  unsigned P0[2][8], P1[2][8];
#pragma unroll
  for (int qb = 0; qb < 2; ++qb)
#pragma unroll
    for (int j = 0; j < 8; ++j) P0[qb][j] = P1[qb][j] = 0u;
  qk_half(0, S0);
  mask_tail(0, S0);
  using T_ = std::true_type;
  using F_ = std::false_type;
  step(F_{}, T_{}, T_{}, 0, S0, S1, P1, P0);             // nh >= 2: the first step always has a successor
  for (int ht = 1; ht + 1 < nh; ht += 2) {               // steady steps 1 .. nh - 2 (odd, then even: one tile per trip)
    step(T_{}, T_{}, F_{}, ht, S1, S0, P0, P1);
    step(T_{}, T_{}, T_{}, ht + 1, S0, S1, P1, P0);
  }
  step(T_{}, F_{}, F_{}, nh - 1, S1, S0, P0, P1);        // nh is even: the last step is an odd one
  pv_half(nh - 1, P1);

#pragma unroll
  for (int qb = 0; qb < 2; ++qb) {
    float lsel = 0.f;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (32 * nb + 8 * g == D) lsel = acc_o[qb][nb][4 * g];
    const float inv = 1.0f / __shfl(lsel, r, 64);
    const int qi = q0 + 32 * qb + r;
    if (qi < p.nq) {
#pragma unroll
      for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int d = 32 * nb + 8 * g + 4 * h;
          if (d < D) {
            float v[4] = {acc_o[qb][nb][4 * g + 0] * inv, acc_o[qb][nb][4 * g + 1] * inv, acc_o[qb][nb][4 * g + 2] * inv,
                          acc_o[qb][nb][4 * g + 3] * inv};
            Elem<bf16_t>::store4(O + (long long)qi * p.ldo + d, v);
          }
        }
    }
  }
}

template <int KS, int NB>
int launch_attn(const SaspaAttnParams& p0, hipStream_t s) {
  SaspaAttnParams p = p0;
  dim3 grid((p.nq + 127) / 128, p.heads, p.batch);
  // 128-key tiles for the long self-attention sequences at head dims <= 96 (register budget);
  // 64-key tiles otherwise (cross-attention has 77 keys, d = 160 needs the registers for O^T)
  constexpr bool BIG_OK = KS <= 6;
  const bool big = BIG_OK && p.nk >= 512;
  // A/B knob (diagnostics; read per launch so that tools/attn_bench.py can flip it inside one process): SASPA_ATTN_MODE
  //   unset: the dispatch rule below.  0: prescaled queries through the v1 loop (c = 1).  1 / 2: v2 with one LDS buffer /
  //   with two buffers and one barrier per tile.  3 / 4: the software-pipelined v3 loop (8 waves = 256 queries per workgroup).
  const char* me = getenv("SASPA_ATTN_MODE");
  int mode = me ? atoi(me) : -1;
  const bool rm = (p.flags & SASPA_ATTN_V_ROWMAJOR) != 0;
  if (p.flags & SASPA_ATTN_QPRESCALED) {
    // v2 / v3 keep S' and the packed P live together, which fits two waves per SIMD up to d = 96 (v3 at d = 96 only with
    // the spare ones row); wider heads (SD-1.5's 16x16 / 8x8 levels, d = 160) and short key sequences (cross-attention:
    // 77 keys = two tiles, the first of which always takes the slow path; 40 vs 44 us at level 0) run the v1 loop.
    constexpr bool V2_OK = KS <= 6;
    constexpr bool V3_OK = V2_OK && (KS <= 5 || 32 * NB > 16 * KS);
    const dim3 grid8((p.nq + 255) / 256, p.heads, p.batch);
    if (mode < 0) {
      // measured (tools/attn_bench.py, profiles/r3_attn_bench.txt): the 8-wave v3 loop wins once its 256-query workgroups
      // still give every CU two of them; below that v2's 128-query workgroups balance better
      const long long wg8 = (long long)grid8.x * grid8.y * grid8.z;
      mode = (V3_OK && !p.causal && wg8 >= 512) ? 4 : 2;
    }
    if ((mode == 3 || mode == 4) && (!V3_OK || p.causal)) mode = 2;
    if (rm && mode != 3 && mode != 4) mode = 0;       // row-major V exists on the v1 and v3 loops
    if (mode == 0 || !V2_OK || !big) {
      p.scale = 0.6931471805599453f;       // the v1 loop multiplies by log2(e): net factor 1
    } else if constexpr (V2_OK) {
      constexpr int KTB = KS <= 3 ? 128 : 64;
#define SASPA_V2(ONES_, DB_) hipLaunchKernelGGL((flash_attn_v2_kernel<KS, NB, ONES_, KTB, DB_>), grid, dim3(256), 0, s, p)
      if constexpr (V3_OK) {
        if (mode >= 3) {
#ifdef SASPA_GEMM_ABLATION
          if constexpr (KS == 3 && NB == 2) {
            const char* ae = getenv("SASPA_ATTN_ABLATE");
            const int abl = ae ? atoi(ae) : 0;
#define SASPA_V3A(A_) case A_: hipLaunchKernelGGL((flash_attn_v3_kernel<KS, NB, true, 8, A_>), grid8, dim3(512), 0, s, p); SASPA_CHECK_LAUNCH(); return 0;
            switch (abl) {
              SASPA_V3A(1) SASPA_V3A(2) SASPA_V3A(4) SASPA_V3A(8) SASPA_V3A(16) SASPA_V3A(14) SASPA_V3A(13) SASPA_V3A(11) SASPA_V3A(7)
              default: break;
            }
#undef SASPA_V3A
          }
#endif
          // six ring slots / one barrier per two steps: measured (tools/attn_bench.py, ring4 / rm4 columns, profiles/r6_attn_ring.txt)
          // -0.5 % at (16, 8, 4096, 4096, 40) and -1.5 % at 5 632 keys on the V^T path with the ones row, but +3.7 % at d = 64 without
          // the ones row and +7.7 % on the row-major-V path (236 against 202 registers) -- the barrier is not what the step waits
          // for.  Taken where it wins; SASPA_ATTN_RING=4: the four-slot ring everywhere (A/B knob, read per launch)
          // v4 (64 queries per wave, 32-key half steps, the level in the head dimension's padding): d = 40 with V^T;
          // SASPA_ATTN_V4=0 keeps v3, =2 takes v4 for every eligible launch (A/B / test knob, read per launch)
          if constexpr (KS == 3 && NB == 2) {
            const char* v4e = getenv("SASPA_ATTN_V4");
            const long long wg16 = (long long)((p.nq + 511) / 512) * p.heads * p.batch;
            const int v4 = v4e ? atoi(v4e) : 1;          // 0: never, 1: from two workgroups per CU on, 2: whenever the shape allows (tests)
            if (!rm && p.D == 40 && v4 != 0 && (wg16 >= 512 || v4 == 2)) {
              hipLaunchKernelGGL((flash_attn_v4_kernel<KS, NB, 8>), dim3((p.nq + 511) / 512, p.heads, p.batch), dim3(512), 0, s, p);
              SASPA_CHECK_LAUNCH();
              return 0;
            }
          }
          const char* re = getenv("SASPA_ATTN_RING");
          if constexpr (KS == 3 && NB == 2) {
            if (!rm && p.D < 32 * NB && !(re && atoi(re) == 4)) {
              hipLaunchKernelGGL((flash_attn_v3_kernel<KS, NB, true, 8, 0, false, 6>), grid8, dim3(512), 0, s, p);
              SASPA_CHECK_LAUNCH();
              return 0;
            }
          }
          if (rm) {
            if (p.D < 32 * NB) hipLaunchKernelGGL((flash_attn_v3_kernel<KS, NB, true, 8, 0, true>), grid8, dim3(512), 0, s, p);
            else hipLaunchKernelGGL((flash_attn_v3_kernel<KS, NB, false, 8, 0, true>), grid8, dim3(512), 0, s, p);
          } else {
            if (p.D < 32 * NB) hipLaunchKernelGGL((flash_attn_v3_kernel<KS, NB, true, 8>), grid8, dim3(512), 0, s, p);
            else hipLaunchKernelGGL((flash_attn_v3_kernel<KS, NB, false, 8>), grid8, dim3(512), 0, s, p);
          }
          SASPA_CHECK_LAUNCH();
          return 0;
        }
      }
      const bool db = mode >= 2;
      if (p.D < 32 * NB) { if (db) SASPA_V2(true, true); else SASPA_V2(true, false); }
      else { if (db) SASPA_V2(false, true); else SASPA_V2(false, false); }
#undef SASPA_V2
      SASPA_CHECK_LAUNCH();
      return 0;
    }
  }
  if (rm) {        // (64-key tiles: the row-major tile of 128 keys x 320 bytes would not leave two workgroups per CU at d = 160)
    if (p.D < 32 * NB) hipLaunchKernelGGL((flash_attn_kernel<KS, NB, true, 64, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((flash_attn_kernel<KS, NB, false, 64, true>), grid, dim3(256), 0, s, p);
  } else if (p.D < 32 * NB) {
    if (big) hipLaunchKernelGGL((flash_attn_kernel<KS, NB, true, BIG_OK ? 128 : 64>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((flash_attn_kernel<KS, NB, true, 64>), grid, dim3(256), 0, s, p);
  } else {
    if (big) hipLaunchKernelGGL((flash_attn_kernel<KS, NB, false, BIG_OK ? 128 : 64>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((flash_attn_kernel<KS, NB, false, 64>), grid, dim3(256), 0, s, p);
  }
  SASPA_CHECK_LAUNCH();
  return 0;
}

// ---- row softmax (unfused path) ----
template <typename T>
__global__ __launch_bounds__(256) void softmax_rows_kernel(T* x, long long rows, int n, int ld, float scale, int causal,
                                                          int rows_per_mat) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  T* xr = x + row * ld;
  int nvis = n;
  if (causal) {
    const int rloc = (int)(row % rows_per_mat);
    nvis = min(n, rloc + 1);
  }
  float mx = -INFINITY;
  for (int j = lane; j < nvis; j += 64) mx = fmaxf(mx, Elem<T>::load1(xr + j) * scale);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int j = lane; j < nvis; j += 64) sum += expf(Elem<T>::load1(xr + j) * scale - mx);
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int j = lane; j < ld; j += 64) {
    float v = 0.f;
    if (j < nvis) v = expf(Elem<T>::load1(xr + j) * scale - mx) * inv;
    Elem<T>::store1(xr + j, v);
  }
}

}  // namespace

extern "C" int saspa_flash_attn_bf16(const SaspaAttnParams* pp, void* stream) {
  if (!pp) return SASPA_EINVAL;
  const SaspaAttnParams& p = *pp;
  if (!p.q || !p.k || !p.vt || !p.o) return SASPA_EINVAL;
  if (p.batch <= 0 || p.heads <= 0 || p.nq <= 0 || p.nk <= 0 || p.D <= 0) return SASPA_EINVAL;
  if (p.D % 8 || p.D > 160) return SASPA_ERANGE;
  if (p.ldq % 8 || p.ldk % 8 || p.ldvt % 8 || p.ldo % 4 || p.sqb % 8 || p.skb % 8 || p.svb % 8 || p.sob % 4) return SASPA_EALIGN;
  if (!aligned16(p.q) || !aligned16(p.k) || !aligned16(p.vt) || !aligned16(p.o)) return SASPA_EALIGN;
  if (p.flags & ~(SASPA_ATTN_QPRESCALED | SASPA_ATTN_V_ROWMAJOR)) return SASPA_EINVAL;
  if (p.flags & SASPA_ATTN_V_ROWMAJOR) {       // vt is V[b][key][heads * D] (row pitch ldvt)
    if (p.ldvt < p.heads * p.D || (long long)p.nk * p.ldvt * 2 >= (1ll << 31)) return SASPA_ERANGE;
  } else {
    if (p.ldvt < ((p.nk + 7) / 8) * 8 || (long long)p.D * p.ldvt * 2 >= (1ll << 31)) return SASPA_ERANGE;
  }
  if (p.ldq < p.heads * p.D || p.ldk < p.heads * p.D || p.ldo < p.heads * p.D) return SASPA_ERANGE;
  if ((long long)p.nk * p.ldk * 2 >= (1ll << 31)) return SASPA_ERANGE;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const int D = p.D;
  if (D <= 16) return launch_attn<1, 1>(p, s);
  if (D <= 32) return launch_attn<2, 1>(p, s);
  if (D <= 48) return launch_attn<3, 2>(p, s);
  if (D <= 64) return launch_attn<4, 2>(p, s);
  if (D <= 80) return launch_attn<5, 3>(p, s);
  if (D <= 96) return launch_attn<6, 3>(p, s);
  if (D <= 128) return launch_attn<8, 4>(p, s);
  return launch_attn<10, 5>(p, s);
}

extern "C" int saspa_softmax_rows(int dtype, void* x, long long rows, int n, int ld, float scale, int causal,
                                  int rows_per_mat, void* stream) {
  if (!x || rows <= 0 || n <= 0 || ld < n) return SASPA_EINVAL;
  if (causal && rows_per_mat <= 0) return SASPA_EINVAL;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  const unsigned grid = (unsigned)((rows + 3) / 4);
  if (dtype == SASPA_BF16)
    hipLaunchKernelGGL(softmax_rows_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (bf16_t*)x, rows, n, ld, scale, causal, rows_per_mat);
  else if (dtype == SASPA_F32)
    hipLaunchKernelGGL(softmax_rows_kernel<float>, dim3(grid), dim3(256), 0, s, (float*)x, rows, n, ld, scale, causal, rows_per_mat);
  else
    return SASPA_EINVAL;
  SASPA_CHECK_LAUNCH();
  return 0;
}
