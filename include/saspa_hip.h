/*
 * saspa_hip.h -- C ABI of libsaspa_hip.so: the MI355X (gfx950) kernels behind the
 * SaSPA augmentation-generation hot path.
 *
 * The reference (EyalMichaeli/SaSPA-Aug) has NO FFI / native layer: the path sits
 * behind one Python call, `pipe(**pipe_args)` (run_aug/run_aug.py:278), whose
 * arithmetic runs inside diffusers/torch CUDA kernels, and behind
 * `cv2.Canny` (all_utils/utils.py:83).  This header is therefore the boundary a
 * maintainer binds *instead of* those library calls; each entry point names the
 * upstream module whose arithmetic it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; no allocation, no
 *     host sync, no hidden stream: work is enqueued on `stream` (a hipStream_t
 *     passed as void*), so calls are graph-capturable;
 *   - activations are channels-last: [batch][H*W][C], pixel pitch given in
 *     elements; dtype selects bf16 (SASPA_BF16) or fp32 (SASPA_F32) storage,
 *     accumulation is always fp32;
 *   - weights are [N][K] with K contiguous (conv: K = (ky*kw+kx)*Cin + c);
 *   - bias / norm parameters / per-batch vectors are always fp32;
 *   - return value 0 on success, negative SASPA_E* on a rejected argument
 *     (checked on the host BEFORE any launch), positive = hipError_t of the launch.
 */
#ifndef SASPA_HIP_H
#define SASPA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SASPA_BF16 0
#define SASPA_F32 1
/* saspa_gemm only: fp32 storage (every pointer as for SASPA_F32), products formed as three bf16 MFMAs (x = hi + lo with
 * hi = bf16(x), lo = bf16(x - hi); a b ~ a_hi b_hi + a_hi b_lo + a_lo b_hi, fp32 accumulation): ~2^-17 relative per
 * product instead of 2^-24, at several times the rate of the fp32 MFMA.  For the fp32-upcast SDXL VAE
 * (run_aug/run_aug.py:224 `pipe.upcast_vae()`), whose range -- not its last 7 bits -- is why the reference leaves fp16. */
#define SASPA_F32X3 2

#define SASPA_EINVAL (-1) /* null pointer / non-positive size */
#define SASPA_EALIGN (-2) /* channel count, pitch or pointer not 16-byte compatible */
#define SASPA_ERANGE (-3) /* shape outside what the kernel supports */

#define SASPA_ACT_NONE 0
#define SASPA_ACT_SILU 1
/* saspa_gemm only, bf16 only: fused GEGLU epilogue.  N = 2F columns are computed, F are
 * stored: out[m][f] = val * gelu_erf(gate).  The weight rows (and bias) must be packed per
 * output tile of BN columns (BN = 160 if N % 160 == 0 else 128; N % BN == 0 required):
 * tile t holds features [t*BN/2, (t+1)*BN/2) -- their value rows first, then their gate rows. */
#define SASPA_ACT_GEGLU 3
/* filter stage (CLIP-RN50 / WSDAN_CAL ResNet convs with the BatchNorm folded into weights + bias): */
#define SASPA_ACT_RELU 5     /* out = relu(alpha*(acc + bias + rowvec)) + residual */
#define SASPA_ACT_ADD_RELU 6 /* out = relu(alpha*(acc + bias + rowvec) + residual): Bottleneck "out += identity; relu" */

#define SASPA_GEMM_AUTO 0
#define SASPA_GEMM_TILED 1 /* 4-wave 128x160 / 128x128 / 64x64 tiles, two workgroups per CU */
#define SASPA_GEMM_WIDE 2  /* 8-wave 256x320 / 256x256 tile, one workgroup per CU */
#define SASPA_GEMM_WS 3    /* wave-specialised 128x160 tile (12 waves: 4 MMA + 4 loader + 4 epilogue): short-K bf16 layers */
#define SASPA_GEMM_AS 4    /* A-stationary: 256 rows x K = 320 held in registers, W streamed through LDS (level-0 pointwise layers) */

#define SASPA_KORDER_TAP 0
#define SASPA_KORDER_CHUNK 1
#define SASPA_KORDER_CHUNK32 2   /* saspa_conv3x3_halo only (ABI 19): K = ((chunk32 * 9 + ky*3+kx) * 32 + c_in_chunk), chunks of 32 channels */

/* ---- implicit-GEMM convolution / linear ----------------------------------
 * out[m][n] = act( alpha * (sum_k A[m][k] * W[n][k] + bias[n] + rowvec[b(m)][n]) ) + residual[m][n]
 * where m = (b, oy, ox) and A is the im2col view of up to two channel-concatenated
 * NHWC sources (skip-concat of the UNet up blocks), optionally nearest-x2 upsampled.
 * Replaces: ResnetBlock2D.conv1/conv2/conv_shortcut, Downsample2D.conv, Upsample2D
 * (interpolate+conv), Transformer2DModel.proj_in/proj_out, Attention.to_q/k/v/to_out,
 * GEGLU.proj, FeedForward.net.2, time_emb_proj, ControlNet zero convs and
 * cond-embedding, VAE decoder convs, CLIP linears (a7.1, a7.5, a7.6, a7.8).
 * A linear layer is the kh=kw=1 case with batch=1, hin=hout=M, win=wout=1.
 * Batched use (nb1*nb2 independent problems, element strides per operand) serves
 * the unfused attention path (scores = Q K^T, out = P V) and the V^T projection. */
typedef struct SaspaGemmParams {
  int dtype;                 /* SASPA_BF16 | SASPA_F32 (A, W, residual, out) */
  const void* a0;            /* source 0, NHWC */
  const void* a1;            /* source 1 (channels appended after source 0) or NULL */
  int c0, c1;                /* channels taken from each source (multiples of 8) */
  int lda0, lda1;            /* pixel pitch of each source, elements */
  int batch, hin, win;       /* stored input extent */
  int hout, wout;            /* output extent */
  int kh, kw, stride, pad;   /* kernel window */
  int upsample;              /* 1: conv sees the input nearest-upsampled x2 */
  const void* w;             /* [N][K] */
  int ldw;                   /* row pitch of w, elements */
  int M, N, K;               /* M = batch*hout*wout, K = kh*kw*(c0+c1) */
  const float* bias;         /* [N] or NULL */
  const float* rowvec;       /* [batch or 1][N] or NULL (time-embedding projection) */
  int ldrv;                  /* batch pitch of rowvec (0: same vector for every batch) */
  const void* residual;      /* [M][N] or NULL */
  int ldr;
  float alpha;
  int act;                   /* SASPA_ACT_* */
  void* out;                 /* [M][N] */
  int ldo;
  /* batching: problem index z = i1*nb2 + i2 */
  int nb1, nb2;
  long long sa1, sa2;        /* element strides of a0 */
  long long sw1, sw2;        /* element strides of w */
  long long so1, so2;        /* element strides of out (and residual) */
  /* split-K (optional): ksplit > 1 with a workspace of ksplit*M*N floats makes the library
   * slice the K range over ksplit workgroups per tile (fp32 partial slabs + one reduce /
   * epilogue launch).  Ignored when workspace is NULL, for batched problems or N % 4 != 0. */
  int ksplit;
  float* workspace;
  /* kernel variant: SASPA_GEMM_AUTO lets the library choose per shape; the other values pin one
   * (tests, tuning).  SASPA_GEMM_WIDE = one 8-wave workgroup per CU on a 256 x 256/320 tile
   * (long-K bf16 layers); it returns SASPA_ERANGE for a problem it cannot run (fp32, fused GEGLU
   * unless N % 320 == 0, channel counts that are not multiples of 64, windows other than 1x1 or 3x3/pad 1). */
  int variant;
  /* K order of the packed weights (and of the K walk), ABI v4.  SASPA_KORDER_TAP: K = (ky*kw+kx)*C + c (tap-major, the
   * default).  SASPA_KORDER_CHUNK: the C = c0+c1 channels are cut into chunks of one K-tile (64 bf16 / 32 fp32 elements)
   * and K = ((chunk*kh*kw) + ky*kw+kx)*tile + c_in_chunk: the kh*kw taps of one channel chunk are consecutive K-tiles,
   * so a workgroup re-reads the same input rows back to back (L2 hits) instead of once per pass over all channels.
   * Requires c0 (and c1, if used) to be multiples of the K-tile; SASPA_ERANGE otherwise. */
  int korder;
  /* GroupNorm statistics out of the epilogue (ABI 12; removes the consumer's statistics pass, saspa_groupnorm_stats).
   * gn_stats != NULL: besides `out`, the launch leaves, for every block rb of 128 consecutive output rows and every unit u of
   * gn_unit consecutive output channels, gn_stats[(rb * (N / gn_unit) + u) * 2 + {0, 1}] = (sum, sum of squares) of the
   * STORED values (after bias / row vector / activation / residual, as rounded to the output dtype) over the block's rows < M.
   * A unit is the finest channel granule every consuming GroupNorm's groups are made of (block_out[0] / groups = 10 for
   * SD-1.5 / SDXL), so one buffer serves a plain consumer, a consumer over a channel concat and both consumers of a skip
   * tensor (SaspaGroupNormParams.stats0 / stats1).  bf16 only, N % 160 == 0, 80 % gn_unit == 0, gn_unit even and <= 16, every
   * tile's first column a multiple of gn_unit (the library picks 160 / 320-column tiles), no fused GEGLU, unbatched;
   * SASPA_ERANGE otherwise.  Deterministic (fixed summation order; no atomics). */
  float* gn_stats;
  int gn_unit;
  /* LayerNorm fused into the operand load, and a transposed second output (ABI 13): the A-stationary kernel for pointwise
   * layers with K = 320 (the level-0 transformer blocks: each workgroup keeps 256 rows of A in REGISTERS across every column
   * tile, so the rows can be normalised once, in place, before the first product).
   * ln_gamma != NULL: A is replaced by LayerNorm(A) over its K channels (two-pass mean / variance in fp32, eps = ln_eps,
   * affine ln_gamma / ln_beta [K], rounded to bf16 -- the arithmetic of saspa_layernorm) before the product; removes the
   * LayerNorm launch and its 2 x M x K bytes.  Only problems saspa_gemm_as_eligible() accepts; SASPA_ERANGE otherwise.
   * out_t != NULL: output columns n >= n_split are written TRANSPOSED to out_t instead of out: element (row, n) goes to
   * out_t[(row / rows_per_batch) * st + (n - n_split) * ldt + row % rows_per_batch] -- the V^T operand of saspa_flash_attn
   * out of the same launch as Q | K (BasicTransformerBlock.attn1: one read of the normalised tokens instead of two
   * launches).  n_split % 64 == 0, rows_per_batch % 32 == 0, same eligibility. */
  const float* ln_gamma;
  const float* ln_beta;
  float ln_eps;
  void* out_t;
  int ldt;
  long long st;
  int n_split;
  int rows_per_batch;
  /* Concurrency hint (ABI 15).  0: the launch has the chip to itself (the default).  1: a TWIN launch of the same shape
   * runs beside it on another stream -- the two encoder branches of a sampling step (UNet encoder || ControlNet encoder walk
   * identical layer shapes side by side) -- so AUTO sizes tile choice and split-K for HALF the CUs: the convs / projections
   * of the 32x32 level (128 tiles of 256 x 320) run un-split on the 8-wave kernel next to their twin instead of on two K
   * slices with fp32 slabs and a reduce launch, the 16x16 level takes two slices instead of four.  Only saspa_gemm_suggest_ksplit
   * and the AUTO dispatch read it; the result differs from sharing = 0 by the summation order of K only. */
  int sharing;
  /* ABI 18.  1: when the launch runs on K slices (ksplit > 1 with a workspace), saspa_gemm leaves the fp32 partial slabs in
   * `workspace` and does NOT run its reduce / epilogue launch -- the caller hands them to saspa_splitk_groupnorm, which sums
   * them, adds bias / row vector and applies the consuming GroupNorm in the same launch (`out` is then not written).
   * ABI 19: a contract, not a hint -- a launch that would end on ONE K slice or on a kernel without slabs (fused GEGLU, N <= 32,
   * the weight-stationary and A-stationary kernels) returns SASPA_ERANGE instead of writing `out` behind the caller's back. */
  int defer_reduce;
  /* ABI 20, SASPA_F32X3 only.  1: `w` holds the weights PRE-SPLIT: every K-tile of 32 packed-K values of a row (128 bytes, where the fp32
   * values would sit) is [32 bf16 hi | 32 bf16 lo], hi = bf16(w), lo = bf16(w - hi), in the order the kernel's lanes consume a tile:
   * 16-byte chunk c (0..3) of either half carries tile positions 4c..4c+3 and 16+4c..16+4c+3 (weights.presplit_x3).  The launch then
   * splits only the activation operand in registers (half the VALU work of the K loop, which is what bounds the SASPA_F32X3 GEMMs);
   * same products, same order: bit-identical to the in-kernel split.  Needs the LDS-DMA loader ((c0 + c1) % 32 == 0, c0 % 32 == 0 with a
   * second source); SASPA_ERANGE otherwise. */
  int w_split;
} SaspaGemmParams;
/* Non-zero if the A-stationary kernel can run the problem (bf16 pointwise layer, K = c0 = 320, N % 64 == 0, at least 192
 * blocks of 256 rows, no row vector / split-K / GroupNorm statistics / batching, alpha = 1, activation none or fused GEGLU,
 * 16-byte aligned operands, ldo % 8 == 0): the only kernel that takes ln_gamma / out_t.  2 = the row blocks also fill whole
 * rounds of the 256 CUs (the sizes where it beats the other kernels on every layer shape), 1 = they do not. */
int saspa_gemm_as_eligible(const SaspaGemmParams* p);
/* ABI 20: 1 if saspa_gemm with variant AUTO will run THIS problem on the A-stationary kernel (the predicate dispatch itself uses):
 * what a caller must ask before it plans around that choice -- e.g. dropping SaspaGemmParams.gn_stats, which that kernel's
 * epilogue does not produce.  (saspa_gemm_as_eligible says whether the kernel CAN run it.) */
int saspa_gemm_as_auto(const SaspaGemmParams* p);
/* ABI 20: which kernel family saspa_gemm would run `p` on and on how many K slices: the dispatch executed DRY (same validation, nothing
 * launched, `stream` not needed).  Returns family | (ksplit << 8), family = SASPA_GEMM_TILED / WIDE / WS / AS, or the SASPA_E* code
 * saspa_gemm would return.  Measurement aid: bench.py attributes every recorded launch to its kernel (roofline.dominant). */
int saspa_gemm_which(const SaspaGemmParams* p);
int saspa_gemm(const SaspaGemmParams* p, void* stream);
/* The library's recommended K-split factor for a problem (1 = none; every field but ksplit / workspace filled in):
 * the caller allocates ksplit*M*N floats, sets p->ksplit / p->workspace and calls saspa_gemm.  Long-K layers with
 * too few tiles to fill the chip get K slices (the 8-wave kernel on the 32x32 / 16x16 levels' 3x3 convs, the 4-wave
 * tiles on the 8x8 level). */
int saspa_gemm_suggest_ksplit(const SaspaGemmParams* p);

/* ---- fused flash attention (bf16) -----------------------------------------
 * O[b][i][h*D+d] = sum_j softmax_j(scale * Q[b][i][h,:] . K[b][j][h,:]) * V[b][j][h,d]
 * V is consumed TRANSPOSED: vt[b][(h*D+d)][j] (produced by saspa_gemm with the
 * operands swapped), keys j >= nk are masked; causal=1 additionally masks j > i
 * (CLIP text tower).  Replaces AttnProcessor2_0 / F.scaled_dot_product_attention in
 * BasicTransformerBlock.attn1/attn2 and CLIPAttention (a7.1, a7.5, a7.6). */
typedef struct SaspaAttnParams {
  const void* q; int ldq; long long sqb;   /* [B][nq][heads*D], row pitch, batch stride (elements) */
  const void* k; int ldk; long long skb;   /* [B][nk][heads*D] */
  const void* vt; int ldvt; long long svb; /* [B][heads*D][ldvt >= nk] */
  void* o; int ldo; long long sob;
  int batch, heads, D;                     /* D multiple of 8, <= 160 */
  int nq, nk;
  float scale;
  int causal;
  int flags;                               /* SASPA_ATTN_* bits (ABI 11) */
} SaspaAttnParams;
/* The queries already carry scale * log2(e) (folded into the to_q weights when they are packed): K Q^T is the
 * log2-domain logit, `scale` is ignored, and the kernel runs the loop with no per-score multiply / subtract / max
 * (saspa_attn.hip, "v2").  Same result up to the rounding point of the scale. */
#define SASPA_ATTN_QPRESCALED 1
/* ABI 14: `vt` holds V ROW-MAJOR -- V[b][key][heads * D], row pitch ldvt >= heads * D, batch stride svb -- i.e. the columns of
 * a fused Q | K | V projection as they are; the kernel transposes on the way from LDS to the MFMA (ds_read_b64_tr_b16), so
 * no transposed value projection is launched (BasicTransformerBlock.attn1 at the 32x32 / 16x16 levels). */
#define SASPA_ATTN_V_ROWMAJOR 2
int saspa_flash_attn_bf16(const SaspaAttnParams* p, void* stream);

/* row softmax in place over a [rows][ld] matrix (unfused attention path: fp32
 * parity mode and the 512-wide single-head VAE attention): x = softmax(scale*x) over
 * the first n columns; causal: row r of each `rows_per_mat` block only sees cols <= r. */
int saspa_softmax_rows(int dtype, void* x, long long rows, int n, int ld, float scale,
                       int causal, int rows_per_mat, void* stream);

/* ---- GroupNorm (+SiLU): NHWC, two launches -------------------------------
 * stats: per (image, pixel split, channel slab, group) sums of x and x^2 into `partial`;
 * apply: every workgroup first combines the sums of ITS image in fp64 into mean / rstd per group (no
 * finalize launch in between, ABI 10), then streams y = act((x - mean) * rstd * gamma + beta).
 * Replaces ResnetBlock2D.norm1/norm2 (+SiLU), Transformer2DModel.norm, conv_norm_out,
 * VAE GroupNorms (a7.5, a7.6, a7.8).
 * x may be the channel concat of two sources (x1 != NULL). */
typedef struct SaspaGroupNormParams {
  int dtype;
  const void* x0; const void* x1;
  int c0, c1, ldx0, ldx1;
  int batch, hw, groups;
  float eps;
  const float* gamma; const float* beta;  /* [C] */
  float* partial;        /* workspace: batch * nsplit * C * 2 floats (holds the per-(image, split, slab, group) sums that
                            saspa_groupnorm_stats leaves for saspa_groupnorm_apply; same p for both calls) */
  int nsplit;            /* pixel splits of the statistics pass, 1 .. hw */
  float* scale_shift;    /* unused since ABI 10 (the apply pass derives scale / shift itself); may be NULL */
  int act;
  void* y; int ldy;      /* apply output [batch*hw][C] */
  /* ABI 12: statistics left by the producers' epilogues (SaspaGemmParams.gn_stats) instead of a saspa_groupnorm_stats
   * launch: stats0 belongs to x0, stats1 to x1 (NULL without x1); each is [batch * hw / 128][c_i / unit][2] floats.
   * Needs hw % 128 == 0, c0 % unit == 0 and (C / groups) % unit == 0; `partial` / `nsplit` are then unused. */
  const float* stats0; const float* stats1;
  int unit;
} SaspaGroupNormParams;
int saspa_groupnorm_stats(const SaspaGroupNormParams* p, void* stream);
int saspa_groupnorm_apply(const SaspaGroupNormParams* p, void* stream);
/* ABI 17: statistics + apply in ONE launch for small images (one workgroup per (image, group): the group's hw * C / groups
 * values are read once into registers, reduced, normalised and written) -- the 8x8 level of the UNet / ControlNet, where a
 * GroupNorm was two launches of launch latency.  `partial` / `nsplit` / `stats*` are not read.  _eligible: non-zero when every
 * group lies inside one source, has whole 8-channel chunks and hw * (C / groups) <= 8 192. */
int saspa_groupnorm_onepass_eligible(const SaspaGroupNormParams* p);
int saspa_groupnorm_onepass(const SaspaGroupNormParams* p, void* stream);
/* ABI 18: split-K reduce + epilogue + GroupNorm(+SiLU) in ONE launch for small images: ResnetBlock2D.conv1 -> norm2 -> SiLU at
 * the 8x8 / 16x16 levels, where conv1 runs on K slices and its reduce launch, the GroupNorm statistics and the apply pass were
 * three launches of launch latency around a tensor nobody else reads.  g: the conv's parameters as given to saspa_gemm with
 * defer_reduce = 1 (M, N, hout * wout rows per image, workspace, ksplit, bias, rowvec / ldrv, alpha; no activation, no residual);
 * n: the GroupNorm (batch, hw, groups, eps, gamma, beta, act, y / ldy; one source of N channels).  One workgroup per (image,
 * group) sums the slabs of its hw x N / groups values, rounds them to the storage dtype (the rounding point of the unfused path),
 * takes the statistics, normalises from registers and writes y.  _eligible: non-zero if (N / groups) % 4 == 0 and
 * hw * N / groups <= 12 288. */
int saspa_splitk_groupnorm_eligible(const SaspaGemmParams* g, const SaspaGroupNormParams* n);
int saspa_splitk_groupnorm(const SaspaGemmParams* g, const SaspaGroupNormParams* n, void* stream);

/* ---- halo-tiled 3x3 conv with the CONSUMING GroupNorm (+SiLU) applied to its input tile in LDS (ABI 19) -----------------
 * out = conv3x3(act(GroupNorm(cat(a0, a1)))) (+ bias + rowvec, * alpha, + residual, epilogue statistics: as saspa_gemm) in ONE
 * launch: ResnetBlock2D's  norm1 -> SiLU -> conv1  and  norm2 -> SiLU -> conv2  (diffusers ResnetBlock2D.forward; SURVEY 8a
 * a7.5 / a7.6).  Replaces saspa_groupnorm_apply + saspa_gemm for these layers: the normalised tensor is never written.
 * p: the conv exactly as for saspa_gemm (bf16, 3x3 / stride 1 / pad 1, no upsampling, c0 / c1 multiples of 32 with c0 + c1 a
 * multiple of 64, N % 320 == 0 or N % 256 == 0, hout * wout % 256 == 0, weights packed SASPA_KORDER_CHUNK32; ksplit /
 * workspace / defer_reduce / gn_stats as for saspa_gemm).  g: the GroupNorm over the conv's INPUT channels, NULL for a plain
 * conv of an already normalised tensor; its statistics come from the producers' epilogues (stats0 / stats1 / unit, see
 * SaspaGroupNormParams) or from saspa_groupnorm_stats (partial / nsplit).  Same arithmetic and rounding points as the two
 * launches it replaces (scale / shift in fp32, SiLU by v_exp + v_rcp, rounding to bf16 before the product).
 * _eligible: non-zero if the launch can run; _ksplit: the K slices a requested factor really uses (slices are whole pairs of
 * 32-channel chunks) -- size the workspace and set p->ksplit from it. */
typedef struct SaspaConvGnParams {
  const float* gamma_beta32;   /* [(c0 + c1) / 32][64]: per 32-channel chunk gamma[32] | beta[32] */
  int groups;
  float eps;
  int act;                     /* SASPA_ACT_NONE | SASPA_ACT_SILU, applied after the affine */
  const float* stats0; const float* stats1; int unit;
  const float* partial; int nsplit;
} SaspaConvGnParams;
int saspa_conv3x3_halo_eligible(const SaspaGemmParams* p, const SaspaConvGnParams* g);
int saspa_conv3x3_halo_ksplit(const SaspaGemmParams* p, int want);
int saspa_conv3x3_halo(const SaspaGemmParams* p, const SaspaConvGnParams* g, void* stream);

/* LayerNorm over the last dim (BasicTransformerBlock.norm1/2/3, CLIP LNs). */
int saspa_layernorm(int dtype, const void* x, int ldx, void* y, int ldy, long long rows, int C,
                    const float* gamma, const float* beta, float eps, void* stream);

/* ---- elementwise ----------------------------------------------------------*/
/* GEGLU: y[m][f] = x[m][f] * gelu_erf(x[m][F+f])   (diffusers GEGLU) */
int saspa_geglu(int dtype, const void* x, int ldx, void* y, int ldy, long long rows, int F, void* stream);
/* act: 1 SiLU, 2 quick-GELU x*sigmoid(1.702x) (CLIP MLP), 4 erf-GELU (BERT / Q-Former FFN), 5 ReLU; in/out [rows][C] */
int saspa_activation(int dtype, int act, const void* x, int ldx, void* y, int ldy, long long rows, int C,
                     void* stream);
/* CLIP embeddings: out[i][:] = tok[ids[i]][:] + pos[i % npos][:] */
int saspa_embed_tokens(int dtype, const int* ids, int n, int npos, const void* tok, const void* pos, int C,
                       void* out, void* stream);
/* ContextCLIPTextEmbeddings of BLIP-Diffusion (a8): ids [nseq][ntok] prompt tokens, ctx
 * [nseq][nctx][C] subject tokens spliced in at position ctx_begin (reference: ctx_begin_pos = 2, 16 tokens,
 * ntok = 77 - 16); out [nseq][ntok+nctx][C] = spliced token embeddings + pos[0 .. ntok+nctx). */
int saspa_embed_tokens_ctx(int dtype, const int* ids, int nseq, int ntok, const void* ctx, int nctx, int ctx_begin,
                           const void* tok, const void* pos, int C, void* out, void* stream);
/* classifier-free guidance + one linear-multistep update (PNDMScheduler.step_plms with skip_prk_steps, the
 * scheduler BlipDiffusionControlNetPipeline keeps, run_aug/run_aug.py:217), fused:
 *   e = eu + g (ec - eu);  if store_slot >= 0: hist[store_slot] = e
 *   m = w_cur e + sum_k w_hist[k] hist[k]   (hist slots as they were BEFORE this call's store)
 *   x' = coef_sample * s + coef_model * m,  s = sample if non-NULL else x;  x' goes to both CFG halves of x.
 * hist: 4 slots of nimg*hw*ldc elements; the host owns which slot holds which past output. */
int saspa_cfg_plms_step(int dtype, const void* eps, void* x, void* hist, const void* sample, int nimg, long long hw,
                        int C, int ldc, float guidance, int store_slot, float w_cur, const float* w_hist,
                        float coef_sample, float coef_model, void* stream);
/* classifier-free guidance + DDIM step (eta=0), fused; writes x_prev into both
 * CFG halves of the model-input buffer.  eps: [2*nimg][hw][ldc] (uncond first),
 * x: [2*nimg][hw][ldc]; channels c < C are live.
 *   e = eu + g*(ec-eu); x0 = (x - sqrt(1-a_t) e)/sqrt(a_t); x' = sqrt(a_p) x0 + sqrt(1-a_p) e
 * Replaces the CFG lines + DDIMScheduler.step (a7.4, a7.7). */
int saspa_cfg_ddim_step(int dtype, const void* eps, void* x, int nimg, long long hw, int C, int ldc,
                        float guidance, float sqrt_a_t, float sqrt_1m_a_t, float sqrt_a_prev,
                        float sqrt_1m_a_prev, void* stream);
/* DDIM step (eta=0) without classifier-free guidance: eps, x [nimg][hw][ldc], x updated in place.  The
 * SDXL-Turbo operating point of the reference (guidance_scale 0 -> one conditional evaluation per step,
 * run_aug/run_aug.py:564-571; SURVEY 8a: a9). */
int saspa_ddim_step(int dtype, const void* eps, void* x, int nimg, long long hw, int C, int ldc, float sqrt_a_t,
                    float sqrt_1m_a_t, float sqrt_a_prev, float sqrt_1m_a_prev, void* stream);
/* ---- device-side step state: ONE captured hipGraph of a sampling step serves every timestep -------------
 * The per-step values a captured launch sequence cannot take as kernel arguments live in device tables indexed by a
 * device-resident step counter: saspa_gather_row_f32 copies row *index of a [rows][row_elems] fp32 table into the
 * buffer the time-embedding row vectors point into (dst[0..n)), saspa_ddim_step_dev is the (CFG +) DDIM update with
 * its four coefficients read from row *index of coefs[steps][4] (cfg = 1: classifier-free guidance pair as
 * saspa_cfg_ddim_step, 0: plain as saspa_ddim_step), saspa_index_add bumps the counter at the end of the step. */
int saspa_gather_row_f32(const float* table, long long row_elems, const int* index, float* dst, long long n, void* stream);
int saspa_ddim_step_dev(int dtype, const void* eps, void* x, int nimg, long long hw, int C, int ldc, int cfg, float guidance,
                        const float* coefs, const int* index, void* stream);
int saspa_index_add(int* index, int delta, void* stream);
/* saspa_cfg_plms_step with its per-evaluation parameters read from row *index of table[evaluations][10] =
 * (store_slot, w_cur, w_hist[4], coef_sample, coef_model, save_sample, use_saved): the evaluation with save_sample copies
 * x into `saved` ([nimg][hw][ldc]) before the update, the one with use_saved reads it instead of x. */
int saspa_cfg_plms_step_dev(int dtype, const void* eps, void* x, void* hist, void* saved, int nimg, long long hw, int C, int ldc,
                            float guidance, const float* table, const int* index, void* stream);
/* y = x * s  (latents / scaling_factor before the VAE) */
int saspa_scale(int dtype, const void* x, void* y, long long n, float s, void* stream);
/* u8 RGB [n][H*W][3] -> [n][H*W][8] activations in [0,1], pad channels zero
 * (VaeImageProcessor.preprocess with do_normalize=False; SURVEY 8a: a7.2) */
int saspa_u8_to_act(int dtype, const uint8_t* src, void* dst, long long npix, void* stream);
/* VAE output [npix][ldx] (3 live channels) -> u8 RGB: round(255*clamp(x/2+0.5,0,1))
 * (VaeImageProcessor.postprocess; SURVEY 8a: a7.10) */
int saspa_act_to_u8(int dtype, const void* x, int ldx, uint8_t* dst, long long npix, void* stream);

/* ---- Canny edge extractor (integer exact vs cv2.Canny, aperture 3, L1) ------
 * src: u8 [n][H][W][3]; dst: u8 [n][H][W][3] in {0,255} (HWC3 replicated);
 * work: 8*n*H*W bytes of scratch.  The hysteresis bitmaps of one image (2*H*ceil(W/32)*4 bytes)
 * live in one CU's LDS when they fit 160 KiB (up to e.g. 512x1280); larger images (1024x1024)
 * keep them in the tail of `work` (W >= 64 required there, else SASPA_ERANGE).
 * Replaces all_utils/utils.py:81-99 CannyDetector/preprocess_canny (a6). */
int saspa_canny(const uint8_t* src, uint8_t* dst, uint8_t* work, int n, int H, int W, int low, int high,
                void* stream);

/* ---- image pre-processing of the CLIP image front-ends (integer exact) ----------
 * One separable pass of Pillow's antialiased resampling for 8-bit channels (ImagingResample, the arithmetic behind
 * PIL.Image.resize that transformers' CLIPImageProcessor -- the feature_extractor of StableDiffusionSafetyChecker --
 * and BlipImageProcessor call; SURVEY 8a: a7.9, a8):
 *   dst[(o*out_len + t)*inner + i] = clip8(((1 << 21) + sum_{k < bounds[t][1]} src[(o*in_len + bounds[t][0] + k)*inner + i]
 *                                           * coeffs[t][k]) >> 22)
 * bounds [out_len][2] = (first input sample, sample count <= ksize), coeffs [out_len][ksize] = the filter weights in
 * 22-bit fixed point -- both DEVICE int32 tables prepared by the host (they depend only on the sizes).  A horizontal
 * pass over [n][H][W][3] is outer = n*H, inner = 3; a vertical pass is outer = n, inner = W*3.  A crop is folded in by
 * passing only the table rows of the wanted outputs. */
int saspa_resample_u8(const uint8_t* src, uint8_t* dst, long long outer, int in_len, int out_len, int inner,
                      const int* bounds, const int* coeffs, int ksize, void* stream);
/* u8 RGB [npix][3] -> [npix][8] activations ((x/255) - mean[c]) / std[c], pad channels zero
 * (CLIPImageProcessor rescale + normalize; SURVEY 8a: a7.9) */
int saspa_u8_to_act_norm(int dtype, const uint8_t* src, void* dst, long long npix, float mean0, float mean1, float mean2,
                         float std0, float std1, float std2, void* stream);

/* StableDiffusionSafetyChecker decision + black-out on the device (no host round trip; SURVEY 8a: a7.9):
 * dots [nimg][ldd] = image embedding . unit(special-care | concept embeddings) (special-care columns first),
 * gram [nimg][ldg] = e e^T (diagonal = |e|^2), both fp32 GEMM outputs; special_w / concept_w = the fp64 thresholds.
 * Per image, in fp64 and upstream's order: cos = dot/|e|; a special-care score cos - w >= threshold sets the 0.01
 * adjustment; the image is flagged when any concept score cos - w + adjustment >= threshold, where `threshold` is
 * the smallest double t with Python's round(t, 3) > 0 (the host computes it once).  flags[i] = 0/1; the bytes of a
 * flagged image (bytes_per_image, multiple of 16) are zeroed -- upstream's black image. */
int saspa_safety_decide(const float* dots, int ldd, const float* gram, int ldg, int nimg, const double* special_w,
                        int n_special, const double* concept_w, int n_concepts, double threshold, uint8_t* images,
                        long long bytes_per_image, int* flags, void* stream);

/* CFG + one UniPCMultistepScheduler step (run_aug/run_aug.py:218-219 `sampler="unipcmultistep"`): eps, x:
 * [2*nimg][hw][8]; state: [3][nimg][hw][8] = last sample | newest x0-prediction | the one before (zero-initialised by the
 * caller).  The 12-float row (saspa_aug_amd/scheduler.py UniPCMultistepScheduler.plan) comes from `row` (host pointer) or,
 * when `table` is non-NULL, from row *index of a device table (hipGraph replays):
 *   x0 = (x - r1 * e) * r0;  if r2: x = r3*last + r4*m0 + r5*m1 + r6*x0;  m1, m0, last = m0, x0, x;  x = r7*x + r8*m0 + r9*m1 */
int saspa_cfg_unipc_step(int dtype, const void* eps, void* x, void* state, int nimg, long long hw, int C, int ldc, float guidance,
                         const float* row, const float* table, const int* index, void* stream);
/* the same update without classifier-free guidance (sd_xl-turbo at guidance_scale 0 with sampler = "unipcmultistep",
 * run_aug/run_aug.py:223-226, :567-571): eps / x are [nimg][hw][ldc], nothing is duplicated.  ABI 11. */
int saspa_unipc_step(int dtype, const void* eps, void* x, void* state, int nimg, long long hw, int C, int ldc,
                     const float* row, const float* table, const int* index, void* stream);

/* SDEdit / img2img start latents (StableDiffusionControlNetImg2ImgPipeline.prepare_latents): per 8-channel
 * latent pixel, moments = AutoencoderKL.encode's quant_conv output (mean | logvar), e1 / e2 = the two generator draws:
 * out = sa * ((mean + exp(0.5 * clamp(logvar, -30, 20)) * e1) * scaling) + s1m * e2, pad channels zero. */
int saspa_vae_sample_noise(int dtype, const void* moments, const void* e1, const void* e2, void* out, long long npix,
                           float scaling, float sa, float s1m, void* stream);

/* ---- filter stage (all_utils/utils.py:306-323, :357-375) ----------------------------------------------
 * 2-D pooling over channels-last activations [batch][hin][win][C] -> [batch][hout][wout][C], hout = (hin + 2*pad - k) /
 * stride + 1: mode 0 = max (nn.MaxPool2d(3, 2, 1) of the ResNet stem: padding never wins), mode 1 = average over the k*k
 * window (nn.AvgPool2d(k): CLIP's anti-aliased strides, the attention pool's mean token; pad must be 0).  C % 8 == 0. */
int saspa_pool2d(int dtype, int mode, const void* x, int ldx, void* y, int ldy, int batch, int hin, int win, int C, int k,
                 int stride, int pad, void* stream);
/* Bilinear-attention-pooling tail of WSDAN_CAL (fgvc/models/cal.py:73-77): per row of [rows][C],
 * y = sign(x) * sqrt(|x| + eps); out = scale * y / max(||y||_2, 1e-12).  fp32 in / out. */
int saspa_signsqrt_l2norm(const float* x, long long ldx, float* y, long long ldy, int rows, long long C, float eps,
                          float scale, void* stream);

/* ---- cv2.resize, 8-bit RGB [n][h][w][3] -> [n][dh][dw][3] (all_utils/utils.py:58-79 `resize_image`) -------
 * saspa_resize_taps_u8: OpenCV's separable fixed-point filters; xofs / yofs = first tap index per destination sample,
 * xw / yw = ntaps 11-bit weights per destination sample.  mode 0 = INTER_LANCZOS4 (ntaps 8), mode 1 = the 8-bit bilinear
 * code INTER_AREA falls back to when a side is up-scaled (ntaps 2).  Out-of-image taps are clamped (replicate).
 * saspa_resize_area_u8: INTER_AREA down-scaling.  isx, isy > 0: integer scale factors (resizeAreaFast_), tables unused;
 * otherwise CSR-style tables from computeResizeAreaTab: destination sample d owns entries [start[d], start[d+1]) of
 * (source index, float weight); accumulation order and float roundings follow resizeArea_<uchar, float>. */
int saspa_resize_taps_u8(const uint8_t* src, uint8_t* dst, int n, int h, int w, int dh, int dw, const int* xofs,
                         const short* xw, const int* yofs, const short* yw, int ntaps, int mode, void* stream);
int saspa_resize_area_u8(const uint8_t* src, uint8_t* dst, int n, int h, int w, int dh, int dw, const int* xstart,
                         const int* xsi, const float* xalpha, const int* ystart, const int* ysi, const float* ybeta,
                         int isx, int isy, void* stream);

/* library self-description */
/* ---- fp8 (OCP e4m3) W8A8 linears (BASELINE.json configs[4] "fp8 MFMA") ------------------------------------
 * saspa_layernorm_quant_fp8: LayerNorm over the last dim of bf16 x [rows][C] (C <= 2048), then per-row quantisation:
 *   scale[r] = max|y[r][:]| / 448, q[r][c] = e4m3(y[r][c] / scale[r])      (BasicTransformerBlock.norm2 / norm3 feeding
 *   attn2.to_q / ff.net.0.proj on the fp8 path; the normalised bf16 tensor is never written).
 * saspa_gemm_fp8: out[m][n] = act(sa[m] * sw[n] * sum_k a[m][k] * w[n][k] + bias[n]) (+ residual[m][n]), a / w e4m3 bytes
 *   (row pitches lda / ldw in bytes = elements, multiples of 16), K and N multiples of 128, sa per row (the quantiser's
 *   scales), sw per output channel (weights.quantize_fp8), out / residual bf16; act = SASPA_ACT_NONE or SASPA_ACT_GEGLU
 *   (rows of w regrouped per 128-column tile: 64 values then their 64 gates; out has N / 2 columns).  128x128 tiles on
 *   v_mfma_f32_16x16x128_f8f6f4, fp32 accumulation. */
typedef struct SaspaGemmF8Params {
  const void* a; int lda;
  const void* w; int ldw;
  int M, N, K;
  const float* sa; const float* sw; const float* bias;
  const void* residual; int ldr;
  int act;
  void* out; int ldo;
  /* ABI 20 (configs[4] breadth: self-attention Q|K|V and the feed-forward OUTPUT projection on fp8 tiles too) */
  int sa_broadcast;        /* 1: `sa` is ONE fp32 scale for every row (an activation tensor quantised under a tensor-wide scale) */
  int out_fp8;             /* GEGLU only. 1: `out` receives e4m3 BYTES, out[m][n] = sat(gelu-gated value / *out_scale), ldo in bytes
                            * (% 16): BasicTransformerBlock.ff.net.0 -> ff.net.2 without a bf16 round trip or a quantisation pass */
  const float* out_scale;  /* device scalar read by the launch (required with out_fp8); a power of two keeps e4m3 rounding scale-free */
  float* amax;             /* GEGLU only, optional device scalar: atomic max of |gated value| over the launch (calibration of out_scale) */
} SaspaGemmF8Params;
int saspa_gemm_fp8(const SaspaGemmF8Params* p, void* stream);
int saspa_layernorm_quant_fp8(const void* x, int ldx, void* q, int ldq, float* scale, long long rows, int C,
                              const float* gamma, const float* beta, float eps, void* stream);

/* ---- HED annotator head (run_aug/run_aug.py:311-312, :438-439 -> controlnet_aux HEDdetector.__call__) -------
 * The network's conv stack runs as saspa_gemm / saspa_pool2d launches; this is what follows it: side output k
 * ([n][mh][mw] fp32 samples at element pitch ld -- channel 0 of a channel-padded NHWC tensor) is resized to H x W like
 * cv2.resize(INTER_LINEAR) on float32 (xofs [W] first source column, xw [W][2] weights; yofs [H][2] the two clamped source
 * rows, yw [H][2] weights: host tables), the nmaps results are averaged in float32 in order, then
 * u8 = trunc(clip(255 / (1 + exp(-mean)), 0, 255)) in fp64, written to all three channels of dst [n][H][W][3]. */
typedef struct SaspaHedFuseParams {
  int nmaps, n, H, W;
  const float* map[5]; int mh[5], mw[5], ld[5];
  const int* xofs[5]; const float* xw[5];
  const int* yofs[5]; const float* yw[5];
  uint8_t* dst;
} SaspaHedFuseParams;
int saspa_hed_fuse(const SaspaHedFuseParams* p, void* stream);

/* ---- cross-attention half of a level-0 transformer block in one launch (ABI 16) --------------------------------------
 * out = residual + to_out( softmax( to_q(LayerNorm(x)) K^T ) V ) + bias for C = 320 channels, 8 heads of 40 and nk <= 96 keys:
 * BasicTransformerBlock.norm2 -> attn2.to_q -> scaled_dot_product_attention over the text tokens -> attn2.to_out.0 -> residual
 * (diffusers attention.py / attention_processor.py; the reference reaches it through pipe(), run_aug/run_aug.py:278).
 * Replaces three launches (LayerNorm + to_q, flash attention, to_out + residual) that each stream the [M, 320] token matrix.
 * Operands in the layouts the kernel's MFMA chain consumes (saspa_aug_amd/weights.py: pack_xattn_*; built once):
 *   w    [640][ldw >= 320] bf16: rows 0..319 = to_q rows (softmax scale * log2 e folded in) ordered [heads 0-7: channels 0-31 |
 *        heads 0-7: channels 32-39]; rows 320..639 = to_out with its K columns in the order the attention stage emits them.
 *   bias [640] fp32: 320 zeros, then the to_out bias.
 *   kf / vf: per sample (byte strides kf_stride >= 73 728, vf_stride >= 98 304) the text K and V^T of this block cut into
 *        MFMA A-operand fragments [head][key block 3][K-step 3][lane 64][8 bf16] and [head][channel block 2][K-step 6][lane 64][8 bf16]
 *        (zero-padded; V^T row 40 of every head = 1.0: the softmax denominator rides the same MFMAs).  Time-invariant.
 * M % 256 == 0, rows_per_sample % 256 == 0 (a workgroup's 256 rows share one sample's keys), all pitches % 8 == 0,
 * 16-byte aligned operands, 32-bit byte offsets (M * pitch * 2 < 2 GiB); SASPA_ERANGE / SASPA_EALIGN otherwise. */
typedef struct SaspaXattnBlockParams {
  const void* x;             /* [M][ldx] bf16: the block's hidden states (LayerNorm input) */
  int ldx;
  const void* residual;      /* [M][ldr] bf16: added to the result (the reference adds x itself) */
  int ldr;
  long long M;
  int rows_per_sample;       /* tokens per sample (H/8 * W/8) */
  const float* ln_gamma;     /* [320] */
  const float* ln_beta;
  float ln_eps;
  const void* w;             /* [640][ldw] bf16, see above */
  int ldw;
  const float* bias;         /* [640] */
  const void* kf;
  long long kf_stride;       /* bytes per sample */
  const void* vf;
  long long vf_stride;
  int nk;                    /* valid keys (77) */
  void* out;                 /* [M][ldo] bf16 */
  int ldo;
} SaspaXattnBlockParams;
int saspa_xattn_block(const SaspaXattnBlockParams* p, void* stream);

/* ---- measurement aid (ABI 20): effective shader clock under load ------------------------------------------------------
 * One sleeping wave measures s_memtime (shader clocks) against s_memrealtime (constant 100 MHz) over iters x s_sleep 127
 * (about 4 us each at 2 GHz; 1 <= iters <= 100000) and writes out2[0] = shader clocks, out2[1] = 100 MHz ticks:
 * clock [MHz] = 100 * out2[0] / out2[1].  bench.py launches it on a side stream while the timed region runs, so that the
 * record of a run carries the clock the box's power management granted it (MI355X_MICROARCH.md: MFMA-dense loops are
 * power-limited and boxes differ by up to 12 %).  No counterpart in the reference (it reports no clocks). */
int saspa_clock_probe(unsigned long long* out2, int iters, void* stream);

/* ---- feed-forward half of a level-0 transformer block in one launch (ABI 20) ---------------------------------------------
 * out = residual + W2 ( v * gelu(g) ) + b2 with [v ; g] = W1 LayerNorm(x) + b1, for C = 320 channels and an inner width F (1 280):
 * BasicTransformerBlock.norm3 -> ff.net.0 (GEGLU) -> ff.net.2 -> residual add (diffusers attention.py; the reference reaches it
 * through pipe(), run_aug/run_aug.py:278).  Replaces two launches (LayerNorm + GEGLU projection, output projection + residual)
 * whose [M, F] hidden state made an HBM round trip.  Operands in the layouts the kernel's MFMA chain consumes (weights.pack_ff_block):
 *   w1  [2 F][ldw1 >= 320] bf16: per slice t of 32 hidden features rows 64 t .. 64 t + 31 = the VALUE rows of features 32 t .. 32 t + 31
 *       (ff.net.0.proj.weight rows f), rows 64 t + 32 .. 64 t + 63 = their GATE rows (rows F + f);  b1 [2 F] fp32 in the same order.
 *   w2f [F / 32][10][2][64][8] bf16: ff.net.2.weight as MFMA A-operand fragments -- fragment (t, nb, s), lane (m, h = lane >> 5),
 *       element e = W2[32 nb + m][32 t + 16 s + (e & 3) + 8 (e >> 2) + 4 h];  b2 [320] fp32.
 * M % 128 == 0, F % 32 == 0, pitches % 8 == 0, 16-byte aligned operands, 32-bit byte offsets; ln_gamma / ln_beta NULL = no LayerNorm.
 * Two forms of the same arithmetic (bit-identical results): the wave-specialised one (8 waves: 4 keep the token rows and run LayerNorm /
 * the first projection / the gate, 4 keep the output accumulators and run the second projection / the epilogue) is the default;
 * SASPA_FF_WS=0 in the environment (read per launch) selects the four-wave form (slower; kept as a cross-check).
 * saspa_ff_block_eligible: host-side check of the geometry (1 = can run). */
typedef struct SaspaFfBlockParams {
  const void* x;             /* [M][ldx] bf16: the block's hidden states (LayerNorm input) */
  int ldx;
  const void* residual;      /* [M][ldr] bf16: added to the result (the reference adds x itself) */
  int ldr;
  long long M;
  int F;                     /* inner width (hidden features after the gate): 4 x 320 */
  const float* ln_gamma;     /* [320] or NULL */
  const float* ln_beta;
  float ln_eps;
  const void* w1;
  int ldw1;
  const float* b1;
  const void* w2f;
  const float* b2;
  void* out;                 /* [M][ldo] bf16 */
  int ldo;
} SaspaFfBlockParams;
int saspa_ff_block(const SaspaFfBlockParams* p, void* stream);
int saspa_ff_block_eligible(const SaspaFfBlockParams* p);

int saspa_abi_version(void);
const char* saspa_build_arch(void);

#ifdef __cplusplus
}
#endif
#endif /* SASPA_HIP_H */
