/* dupl_hip.h -- C ABI of libdupl_hip.so: the gfx950 (MI355X / CDNA4) kernels behind DuPL's per-step
 * hot path (BASELINE.json:north_star).
 *
 * The reference (Wu0409/DuPL) has no native layer: every op below is, in the reference, a stock ATen
 * call issued from Python.  Each entry point therefore cites the reference *Python* site it replaces
 * (paths relative to the reference checkout).  The binding a maintainer adds is a ctypes stub
 * (INTEGRATION.md); dupl_amd/_lib.py is exactly that stub.
 *
 * Conventions: raw device pointers, explicit dims / leading dimensions in ELEMENTS,
 * fp32 contiguous data unless noted, no allocation, no hidden synchronisation, work is enqueued on
 * `stream` (a hipStream_t passed as void*; NULL = the null stream), re-entrant per stream.
 * Return value: 0 on success, <0 on error (-1 bad argument, -2 launch failure).
 */
#ifndef DUPL_HIP_H
#define DUPL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dupl_stream_t;

/* library info: returns the ABI version (2: descriptors carry struct_size, tuning knobs travel in the descriptors, one entry point
 * per operation; 3: the loss-sum buffers of dupl_ptc_reduce / dupl_seg_loss_fwd are DUPL_LOSS_SUMS_FLOATS floats, the determinism
 * switch is per call; 4: dupl_build_digest, dupl_adamw's grad_scale, the fused loss total (dupl_loss_total) -- see INTEGRATION.md).
 * Infrastructure, no reference counterpart. */
int dupl_abi_version(void);
/* build identity: the sha256 (64 hex digits + NUL into out[cap], cap >= 65) of the kernel sources this library was compiled from
 * (every .hip / .h under csrc/ and include/dupl_hip.h), baked in at build time (dupl_amd/build.py).  The Python stub refuses a library whose
 * digest differs from the sources it finds next to it; bench.py tags its roofline line with this value.  Infrastructure. */
int dupl_build_digest(char* out, int32_t cap);
/* DETERMINISM is a per-call argument since ABI 3 (the library holds no mode: re-entrant per stream and per caller).  Every entry
 * point that otherwise accumulates with fp32 atomics takes `deterministic` (a descriptor field or a trailing argument): != 0 =
 * that accumulation runs in a fixed order, so that two identical steps give bit-identical gradients (torch.use_deterministic_
 * algorithms / cudnn.deterministic of train_final_voc.py:95-102; slower).  These are: dupl_gemm_f32 / dupl_gemm_f16x3 (split-K
 * and stream-K weight / data gradients), dupl_split_prepare (fused bias column sums: refused), dupl_layernorm_bwd (dgamma / dbeta),
 * dupl_colsum, dupl_seg_loss_bwd (bilinear scatter).  The loss scalars (dupl_ptc_reduce, dupl_seg_loss_fwd) are order-independent
 * in every mode.  dupl_amd.set_deterministic (Python) is the caller-side switch that fills the argument. */

/* ---------------------------------------------------------------------------------------------
 * GEMM on exact-fp32 MFMA (v_mfma_f32_32x32x2_f32), fp32 in / fp32 accumulate / fp32 out.
 *   C[z][m][n] = epilogue( alpha * sum_k A[z](m,k) * B[z](k,n) )
 * Replaces: nn.Linear / F.linear (vit.py:92-102,115-122,136), q@k^T and attn@v (vit.py:129,135),
 * F.conv2d 1x1 classifiers (model_dupl.py:82-83,89,95), the patch-embed conv as im2row GEMM
 * (vit.py:176-183), LargeFOV convs as im2col GEMMs (conv_head.py:32-41), the PTC Gram matrix
 * (losses.py:12) and every dgrad / wgrad autograd derives from them.
 * flags: */
#define DUPL_LOSS_SUMS_FLOATS 136 /* size of the `sums` buffers of dupl_ptc_reduce / dupl_seg_loss_fwd (results in [0..3]) */
#define DUPL_GEMM_A_MCONTIG 1   /* A stored [K][lda] (m contiguous) instead of [M][lda] (k contiguous) */
#define DUPL_GEMM_B_NCONTIG 2   /* B stored [K][ldb] (n contiguous) instead of [N][ldb] (k contiguous) */
#define DUPL_GEMM_GELU 4        /* v = gelu_erf(v) after bias */
#define DUPL_GEMM_ACCUM 8       /* C += v instead of C = v */
#define DUPL_GEMM_MUL_DGELU 16  /* v *= gelu'(aux[m][n])   (MLP backward) */
#define DUPL_GEMM_RELU 32       /* v = max(v,0) */
#define DUPL_GEMM_MUL_RELUMASK 64 /* v *= (aux[m][n] > 0)   (ReLU backward; aux = post-ReLU output) */
#define DUPL_GEMM_ABS 128       /* v = |v|  (PTC cosine matrix) */
#define DUPL_GEMM_STORE_PRE 256 /* also WRITE the pre-activation (alpha*acc+bias) to aux[m][n] (training forward of fc1) */
typedef struct dupl_gemm_desc {
    uint32_t struct_size;  /* sizeof(dupl_gemm_desc) of the caller's header; a mismatch is refused */
    int32_t tile_rows;     /* launch tuning, per call (0 = heuristic): row tile 64 / 128 */
    int32_t tile_cols;     /* column tile of the 64-row kernels, 64 / 128 */
    int32_t group;         /* row tiles per group of the block -> C-tile order inside an XCD band (0 = 16; 4096 = plain row-major) */
    const float* A; const float* B; float* C;
    const float* bias;     /* [N] or NULL, added before the activation */
    const float* res;      /* [M][ldr] or NULL, added after the activation */
    const float* aux;      /* [M][ldaux] or NULL */
    int32_t M, N, K;
    int32_t lda, ldb, ldc, ldr, ldaux;
    int32_t batch;         /* grid.z; z -> (z / zdiv, z % zdiv) */
    int32_t zdiv;
    int64_t sA0, sA1, sB0, sB1, sC0, sC1, sR0, sR1, sX0, sX1, sBias0, sBias1; /* element strides */
    float alpha;
    int32_t flags;
    int32_t deterministic; /* != 0: no split-K (fp32 atomics): one block owns the whole reduction of its tile */
    int32_t reserved0;
} dupl_gemm_desc;
/* the GEMM described above (vit.py:92-136, model_dupl.py:82-95, conv_head.py:32-41, losses.py:12 and their autograd) */
int dupl_gemm_f32(const dupl_gemm_desc* d, dupl_stream_t stream);
/* ---------------------------------------------------------------------------------------------
 * fp32-equivalent GEMM on the f16 matrix cores by operand splitting (csrc/gemm_split.hip):
 *   x = hi + lo/2048 with hi = fp16(x), lo = fp16((x - hi) * 2048);  a*b ~= hi_a hi_b + (hi_a lo_b + lo_a hi_b)/2048,
 * fp32 accumulation, 3 v_mfma_f32_32x32x16_f16 per 32x32x16 block.  Same reference sites as dupl_gemm_f32 for the
 * k-contiguous x k-contiguous case (every nn.Linear forward: vit.py:92-102,115-122,136).  Operands are passed as two
 * fp16 planes each ([rows][ld] halfs); the result can be written as fp32 and / or as planes (the next GEMM's A operand).
 *   C = act(alpha * A . B^T + bias) (+ res);   flags: DUPL_GEMM_GELU | DUPL_GEMM_RELU | DUPL_GEMM_STORE_PRE (aux =
 *   pre-activation) | DUPL_GEMM_MUL_DGELU | DUPL_GEMM_MUL_RELUMASK (aux read) | DUPL_GEMM_ACCUM (C += alpha * A . B^T, split-K
 *   with fp32 atomics: the weight gradients).  The backward GEMMs (autograd of vit.py:92-136) reach this layout through
 *   transposed operand planes (dupl_split_prepare): dgrad = dy . (W^T)^T, wgrad = dy^T . (x^T)^T.
 * K % 32 == 0; lda / ldb in halfs, multiples of 8; plane pointers 16-byte aligned. */
typedef struct dupl_gemm16_desc {
    uint32_t struct_size;                 /* sizeof(dupl_gemm16_desc) of the header the CALLER was built against: a mismatch is
                                             refused (DUPL_ERR_ARG) instead of reading fields the caller never wrote */
    int32_t deterministic;                /* != 0: no split-K / stream-K (fp32 atomics) under DUPL_GEMM_ACCUM: one block per tile over all of K */
    const void* A_hi; const void* A_lo;   /* [M][lda] fp16   (a_layout 1: [K][lda], the M rows contiguous) */
    const void* B_hi; const void* B_lo;   /* [N][ldb] fp16   (b_layout 1: [K][ldb], the N rows contiguous) */
    float* C;                             /* [M][ldc] fp32 or NULL */
    void* C_hi; void* C_lo;               /* [M][ldo] fp16 planes of the result, or both NULL */
    const float* bias;                    /* [N] or NULL */
    const float* res;                     /* [M][ldr] or NULL, added after the activation */
    float* aux;                           /* [M][ldaux], written when DUPL_GEMM_STORE_PRE */
    int32_t M, N, K;
    int32_t lda, ldb, ldc, ldo, ldr, ldaux;
    int32_t flags;
    int32_t c_rows;                       /* > 0: the fp32 outputs (C, and aux under DUPL_GEMM_STORE_PRE) are written for rows < c_rows only --
                                             the planes for all M rows (shared ms-CAM / training pass: only the training rows are
                                             back-propagated); 0 = all rows */
    const float* alpha_dev;               /* device scalar multiplied into A.B^T before the epilogue (inverse operand scales of
                                             scaled gradient planes, dupl_split_prepare), or NULL (= 1) */
    int32_t fmt;                          /* operand plane format of A and B: 0 = lo planes scaled by 2048 (two accumulator sets), 1 = planes of
                                             x * 2^s with unscaled lo (dupl_split_f16x2b: one accumulator set, 256 x 256 tiles) */
    int32_t out_exp;                      /* C_hi / C_lo: 0 = format 0; s > 0 = format 1 planes of C * 2^s */
    float post_scale;                     /* multiplied into A.B^T together with alpha (2^-(sA + sB) for format 1 operands); 0 = 1 */
    void* amax_out;                       /* NULL, or the amax word of a scale slot (dupl_split_prepare, amax_mode 1): the kernel
                                             raises it (atomic max on the bits) to max |C| over the M x N result, so that the
                                             split of C needs no pass of its own.  Not with DUPL_GEMM_ACCUM or c_rows. */
    int32_t a_layout, b_layout;           /* 0 = k-contiguous ([rows][K], the Linear forward's layout); 1 = K-MAJOR ([K][rows], rows
                                             contiguous): how the backward's operands lie in memory -- dgrad dx = dy . W reads the forward's
                                             W planes as a k-major B, wgrad dW += dy^T . x reads dy and x planes as k-major A and B --
                                             so no transposed planes are built.  Needs fmt 1; M (a_layout 1) / N (b_layout 1) % 8 == 0. */
    int32_t ka_valid, kb_valid;           /* k-major operands: k-rows that exist in memory (0 = K).  Rows beyond are read as the last
                                             valid row; the OTHER operand must hold zeros there (K itself is a multiple of 32, >= 96). */
    /* launch tuning, per call (0 = the library's heuristic): nothing about kernel selection is process-global any more */
    int32_t tile;                         /* block tile.  Format 0 operands: 3: 128x64 on 4 waves, 5: 128x128 on 8 waves, 6 / 7: 256x128
                                             ring kernel on 8 / 4 waves, 10: its persistent form, 11: the stream-K form of that for
                                             DUPL_GEMM_ACCUM.  Format 1 (one accumulator set): 8: 256x256 on 8 waves, 12: 256x128, 14:
                                             persistent 256x128 */
    int32_t concurrency;                  /* how many streams issue split GEMMs at the same time (2 while the two students of
                                             siamese_network run on their own streams, model_dupl.py:157-213): tile heuristic input */
    int32_t persist_blocks;               /* blocks of the persistent kernels, a multiple of 8 (0: 256 alone, 192 at concurrency 2) */
    int32_t group;                        /* row tiles per group of the block -> tile order */
    int32_t sk_slices;                    /* stream-K forms of the k-major kernels (DUPL_GEMM_ACCUM, not deterministic): n > 0 = every tile's k axis
                                             in n aligned slices, one (tile, slice) unit per block, units dealt slice-major to the XCDs (operands
                                             shared in L2; an n that does not fit a launch's shape -- more units than blocks, a slice under 3 k-steps -- is
                                             replaced by the library's choice); 0 = the library picks n from the grid; < 0 = equal runs of (tile, k-step) pairs */
    int32_t reserved1;
} dupl_gemm16_desc;
int dupl_gemm_f16x3(const dupl_gemm16_desc* d, dupl_stream_t stream);
/* n (<= DUPL_GEMM16_GROUP_MAX) independent weight gradients C_i += alpha_i A_i^T . B_i (every descriptor: fmt 1, a_layout = b_layout = 1,
 * flags = DUPL_GEMM_ACCUM) as ONE launch, a whole 256 x 128 tile per block over all of K: the four dW of a transformer block
 * (autograd of vit.py:92-136: 18 .. 72 tiles each) fill the chip together, nothing is split along K and nothing meets in atomics --
 * bit-reproducible with and without dupl_set_deterministic.  descs: HOST array (copied into the launch). */
#define DUPL_GEMM16_GROUP_MAX 8
int dupl_gemm_f16x3_group(const dupl_gemm16_desc* descs, int32_t n, dupl_stream_t stream);
/* the operand split of the GEMM above: n fp32 values (n % 4 == 0) -> hi / lo fp16 planes (no reference counterpart) */
int dupl_split_f16x2(const float* x, void* hi, void* lo, int64_t n, dupl_stream_t stream);
/* the same into format 1 planes of x * 2^scale_exp (dupl_gemm16_desc.fmt) */
int dupl_split_f16x2b(const float* x, void* hi, void* lo, int64_t n, int32_t scale_exp, dupl_stream_t stream);
/* operand preparation for the backward split GEMMs (csrc/split_prep.hip): x [R][ld] fp32 (C columns) -> row-major planes
 * hi / lo [R][C] and / or transposed planes hiT / loT [C][Rp] (Rp >= R, multiple of 8; rows R.. are zeros).
 * slot != NULL (gradients, far below fp16's normal range): the tensor is scaled by the power of two that brings its
 * max-abs into [2^(target_exp-1), 2^target_exp) (15 for GEMM operands); slot = 4 floats of device memory {scale, 1 / scale,
 * amax word, -}; pass slot + 1 as the GEMM's alpha_dev.  next_bits (optional): the amax word of the slot the next scaled call
 * on this stream will use -- it is zeroed by this call (a ring of slots then needs no memset).
 * amax_mode: where the amax word of `slot` comes from -- 0 = this call computes it (the word must be zero on entry); 1 = the
 * kernel that produced x has left max |x| there already (dupl_gemm16_desc.amax_out, dupl_layernorm_bwd): no amax pass; 2 = the
 * word may hold a stale value: it is cleared (memset node), then computed.
 * colsum_accum != NULL: colsum_accum[c] += sum_r x[r][c] (fp32 atomics, unscaled values) -- the bias gradient of a Linear
 * (autograd of vit.py:92-136's `+ bias`) from the pass that reads dy anyway; refused in deterministic mode (dupl_colsum there).
 * fmt 1: the planes are written in format 1 (unscaled lo, dupl_gemm16_desc.fmt) -- the single-accumulator k-major backward GEMMs.
 * rows_zero_to > R: the row-major planes have rows_zero_to rows and rows R .. rows_zero_to - 1 are written as zeros (the k-major
 * A operand of a weight gradient, whose contraction index is padded to a multiple of 32).
 * No reference counterpart (the reference's autograd calls ATen GEMMs on fp32 operands). */
typedef struct dupl_split_desc {
    uint32_t struct_size;                 /* sizeof(dupl_split_desc), checked */
    int32_t ld, R, C;
    const float* x;
    float* slot; void* next_bits;
    void* hi; void* lo; void* hiT; void* loT;
    int32_t Rp, target_exp;
    float* colsum_accum;
    int32_t amax_mode, fmt, rows_zero_to;
    int32_t deterministic;                /* != 0: colsum_accum (fp32 atomics) is refused -- the caller uses dupl_colsum(..., deterministic) */
} dupl_split_desc;
int dupl_split_prepare(const dupl_split_desc* d, dupl_stream_t stream);
/* several UNSCALED matrices (saved activations, weights: the x^T / W^T operands of one transformer block's backward) in ONE
 * launch: items[i] = the x / ld / R / C / hi / lo / hiT / loT / Rp fields of dupl_split_desc with slot = NULL.  items: host array, n <= DUPL_SPLIT_MULTI_MAX.
 * The short operand-preparation kernels run chip-exclusive between the persistent GEMMs (which take every CU's LDS and
 * registers), so a launch saved is its whole duration saved. */
#define DUPL_SPLIT_MULTI_MAX 16
typedef struct dupl_split_item {
    const float* x; void* hi; void* lo; void* hiT; void* loT;
    int32_t ld, R, C, Rp;
} dupl_split_item;
int dupl_split_prepare_multi(const dupl_split_item* items, int32_t n, dupl_stream_t stream);
/* ---------------------------------------------------------------------------------------------
 * Range guard of the f16x3 operand planes (csrc/range.hip; no reference counterpart: the reference's fp32 has range 3.4e38,
 * the split format |x| <= 65504).  For n tensors of a parameter buffer (table_dev: device array of descriptors; a vector is
 * rows = 1) writes out[2 e] = max |x| and out[2 e + 1] = the largest row L2 norm (device floats; a NaN gives +inf).  The host
 * turns them into rigorous bounds on every tensor that is written as planes (engine.RangeGuard) and routes the Linear /
 * attention whose operands could leave fp16's range to the exact-f32 kernels. */
typedef struct dupl_bound_desc {
    int64_t offset;        /* in floats from `base` */
    int32_t rows, cols;
} dupl_bound_desc;
int dupl_param_bounds(const float* base, const dupl_bound_desc* table_dev, int32_t n, float* out, dupl_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * LayerNorm over the last dim D (D % 4 == 0, D <= 2048), one wavefront per row.
 * Replaces nn.LayerNorm(eps=1e-6) (vit.py:146,152,256; applied at :157,:159,:323).
 * fwd: y = (x-mean)*rstd*gamma+beta; mean/rstd [rows] saved when non-NULL.
 * bwd: dx = [dres +] rstd*(g - mean(g) - xhat*mean(g*xhat)), g = dy*gamma;
 *      dgamma += sum_rows dy*xhat, dbeta += sum_rows dy  (atomic accumulate: zero them first). */
int dupl_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                       float* mean, float* rstd, int64_t rows, int32_t D, float eps, dupl_stream_t s);
/* the same LayerNorm writing its output (also / only) as f16x3 operand planes y_hi / y_lo [rows][D] fp16 for
 * dupl_gemm_f16x3 (y may be NULL when only the planes are wanted; y_hi / y_lo both NULL = dupl_layernorm_fwd).
 * f32_rows > 0: the fp32 copy y (and mean / rstd) is written for the first f32_rows rows only and has that many rows (the
 * shared ms-CAM / training pass back-propagates only those); plane_exp > 0: the planes in format 1 (y * 2^plane_exp, unscaled
 * lo: dupl_gemm16_desc.fmt) */
int dupl_layernorm_fwd16(const float* x, const float* gamma, const float* beta, float* y, void* y_hi, void* y_lo, float* mean,
                         float* rstd, int64_t rows, int32_t D, float eps, int64_t f32_rows, int32_t plane_exp, dupl_stream_t s);
/* LayerNorm backward (autograd of vit.py:157,159,323), optionally fused with the residual-stream gradient add (dres).
 * amax_out != NULL: max |dx| is raised into *amax_out (the amax word of a scale slot, see dupl_split_prepare amax_mode 1).
 * partials != NULL (two-stage dgamma / dbeta): every wave writes its partial sums to partials [partial_rows][2 D] (partial_rows >=
 * dupl_layernorm_bwd_blocks(rows, rows_per_wave)) and a second kernel adds them up -- in a fixed order under
 * dupl_set_deterministic(1): the bit-reproducible form without a second pass over dy and x (not faster than the atomics: 27 vs
 * 22 us at 3140 x 768); partials == NULL: one kernel, fp32 atomics.  rows_per_wave: 0 = default (4); a block = 4 waves.
 * dy_clear: NULL, or dy itself = hand dy back ZERO-FILLED (dy is the accumulation target of a stream-K data gradient, which wants
 * zeros for its next use: the rows are cleared by the kernel that has just read them instead of a fill launch). */
int dupl_layernorm_bwd_blocks(int64_t rows, int32_t rows_per_wave);   /* a count, not a status */
int dupl_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean,
                       const float* rstd, const float* dres, float* dx, float* dgamma, float* dbeta,
                       int64_t rows, int32_t D, void* amax_out, float* partials, int64_t partial_rows, int32_t rows_per_wave,
                       float* dy_clear, int32_t deterministic, dupl_stream_t s);

/* column sums: out[n] (+)= sum_m x[m][n]: the bias gradients autograd derives for nn.Linear (vit.py:92-102,115-122)
 * and the patch-embed conv (vit.py:176-183).  accumulate!=0 adds to out (atomic). */
int dupl_colsum(const float* x, float* out, int64_t M, int32_t N, int32_t ldx, int32_t accumulate,
                int32_t deterministic, dupl_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * Fused multi-head self-attention core (flash style, exact-fp32 MFMA, online softmax).
 * Replaces Attention.forward's q@k^T*scale -> softmax -> @v (vit.py:123-135) without
 * materialising the (B,H,N,N) probabilities (the reference returns them; vit.py:320 drops them).
 * qkv: [B*N][3*H*hd] exactly as nn.Linear(dim,3*dim) writes it (q|k|v, head-major inside each);
 * out: [B*N][H*hd] (= the (attn@v).transpose(1,2).reshape(B,N,C) layout); lse: [B][H][N] or NULL.
 * hd in {32, 64}. */
int dupl_attention_fwd(const float* qkv, float* out, float* lse, int32_t B, int32_t N, int32_t H,
                       int32_t hd, float scale, dupl_stream_t s);
/* The same attention forward as fp32-equivalent f16x3 split products (csrc/attn_split.hip; head dim 64): q, k, v are read
 * from the fp16 hi / lo planes of the qkv GEMM output ([B*N][3*H*hd] halfs each), S = q k^T and O = P v are
 * hi hi + (hi lo + lo hi) / 2048 on v_mfma_f32_32x32x16_f16 with fp32 accumulation, softmax in fp32.  V is read in place
 * (k-major operand, transposing LDS reads): no scratch.
 * out (fp32) and / or out_hi / out_lo (planes, the A operand of the projection GEMM); lse optional.
 * B_f32 (0 = B): the fp32 output and lse are written for the first B_f32 images only (out: [B_f32*N][H*hd], lse: [B_f32][H][N]),
 * the planes for all B; out_exp > 0: the output planes in format 1 (out * 2^out_exp, unscaled lo). */
int dupl_attention_fwd16(const void* qkv_hi, const void* qkv_lo, float* out, void* out_hi, void* out_lo, float* lse, int32_t B,
                         int32_t N, int32_t H, int32_t hd, float scale, int32_t B_f32, int32_t out_exp, dupl_stream_t s);
/* the same for up to DUPL_ATTN_SEGS_MAX batches that live in ONE token buffer (the merged ms-CAM / training pass of a step: all scales'
 * rows concatenated), as ONE launch, longest batch first: segment i = B images of N tokens starting at token row row0 of the qkv planes
 * (and of out_hi / out_lo); out / lse: this batch's own fp32 output ([B_f32*N][H*hd]) and lse ([B_f32][H][N]) or NULL (planes only).
 * segs: HOST array (copied into the launch). */
#define DUPL_ATTN_SEGS_MAX 4
typedef struct dupl_attn_seg {
    int64_t row0;
    int32_t B, N, B_f32, reserved0;
    float* out;
    float* lse;
} dupl_attn_seg;
int dupl_attention_fwd16_segs(const void* qkv_hi, const void* qkv_lo, void* out_hi, void* out_lo, const dupl_attn_seg* segs, int32_t n,
                              int32_t H, int32_t hd, float scale, int32_t out_exp, dupl_stream_t s);
/* backward (what autograd derives for vit.py:123-135): dqkv [B*N][3*H*hd] fully written; delta: workspace [B][H][N]. */
int dupl_attention_bwd(const float* qkv, const float* out, const float* dout, const float* lse,
                       float* delta, float* dqkv, int32_t B, int32_t N, int32_t H, int32_t hd,
                       float scale, dupl_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * Token plumbing (vit.py:176-184, 289-306; model_dupl.py:64-67, 88-95). */
/* im2row for the 16x16/s16 patch conv PatchEmbed.proj (vit.py:176-183): x (B,3,H,W) -> rows [B*h*w][3*P*P] (c,py,px
 * order = conv weight layout); trailing H % P rows / W % P columns are ignored like the strided conv does */
int dupl_patch_im2row(const float* x, float* rows, int32_t B, int32_t H, int32_t W, int32_t P, dupl_stream_t s);
/* inverse scatter is never needed (inputs get no grad). */
/* bicubic (A=-0.75, align_corners=False) resize of the (g x g) pos-embed grid to (h x w); pos_embed (1,1+g*g,D)
 * -> out [1+h*w][D] with the cls row copied (vit.py:294-297). */
int dupl_pos_embed_resize(const float* pos_embed, float* out, int32_t g, int32_t h, int32_t w, int32_t D, dupl_stream_t s);
/* tokens[b][0] = cls + pos[0]; tokens[b][1+i] = patch[b][i] + pos[1+i]  (vit.py:300-304) */
int dupl_assemble_tokens(const float* patch, const float* cls, const float* pos, float* tokens,
                         int32_t B, int32_t n, int32_t D, dupl_stream_t s);
/* backward of the above (autograd of vit.py:300-304) wrt patch rows and cls token: dpatch[b][i] = dtok[b][1+i];
 * dcls += sum_b dtok[b][0] */
int dupl_assemble_tokens_bwd(const float* dtok, float* dpatch, float* dcls, int32_t B, int32_t n, int32_t D, dupl_stream_t s);
/* F.adaptive_max_pool2d(x, (1,1)) of model_dupl.py:87-92 on token-major activations: max over the n patch tokens
 * (skipping the cls row): tokens [B][1+n][D] -> out [B][D], idx [B][D] (first maximum, as torch) */
int dupl_gmp_fwd(const float* tokens, float* out, int32_t* idx, int32_t B, int32_t n, int32_t D, dupl_stream_t s);
/* its adjoint (autograd of model_dupl.py:87-92): dtokens[b][1+idx][d] += dout[b][d] */
int dupl_gmp_bwd(const float* dout, const int32_t* idx, float* dtokens, int32_t B, int32_t n, int32_t D, dupl_stream_t s);
/* tokens [B][1+n][D] (skip cls) <-> NCHW (B,D,h,w): network.to_2D (model_dupl.py:64-67) and its adjoint
 * (adjoint ACCUMULATES into dtokens rows 1..n). */
int dupl_tokens_to_nchw(const float* tokens, float* out, int32_t B, int32_t n, int32_t D, int32_t skip_cls, dupl_stream_t s);
/* adjoint of network.to_2D (autograd of model_dupl.py:64-67) */
int dupl_nchw_to_tokens_add(const float* dnchw, float* dtokens, int32_t B, int32_t n, int32_t D, int32_t skip_cls, dupl_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * Multi-scale CAM (cam_helper.py:164-204 / camutils.py:87-127), CAM->label (cam_helper.py:8-55). */
/* F.interpolate(mode="bilinear") (B,C,Hi,Wi)->(B,C,Ho,Wo); flip_cat!=0 writes 2B images [x ; flip_w(x)]
 * (the torch.cat([inputs, inputs.flip(-1)]) of cam_helper.py:169,184). */
int dupl_resize_bilinear(const float* in, float* out, int32_t B, int32_t C, int32_t Hi, int32_t Wi,
                         int32_t Ho, int32_t Wo, int32_t flip_cat, int32_t align_corners, dupl_stream_t s);
/* Fused ms-CAM: lows[i] = token-major CAM logits of scale i, [2B][row_off + hs*ws][ldc] (first B images = original
 * input, last B = w-flipped input; row_off = 1 skips the cls row).
 *   cam[b][c][y][x] = sum_i relu(max(up_i(low_i[b])(y,x), up_i(low_i[B+b])(y, W-1-x)))       (cam_helper.py:173-196)
 * and mm[b*C+c] = {min, max} over the plane.  `lows`, `hs`, `ws` are HOST arrays of nscale (<= 4) entries.
 * impl (test / tuning, per call): 0 = the library's choice (the LDS-staged band kernel where it applies), 1 = the per-pixel kernel
 * (same bits); band_blocks: blocks the band kernel aims at (bands = blocks / planes, at least 8 rows each), 0 = 768. */
int dupl_cam_fuse(const float* const* lows, const int32_t* hs, const int32_t* ws, int32_t nscale, int32_t row_off,
                  int32_t ldc, float* cam, float* mm, int32_t B, int32_t C, int32_t H, int32_t W, int32_t impl,
                  int32_t band_blocks, dupl_stream_t s);
/* per plane, in place: cam = (cam - min) / ((max - min) + 1e-5)  == `cam + maxpool(-cam); cam /= maxpool(cam) + 1e-5`
 * (cam_helper.py:197-199).  mm [planes][2]; have_minmax = 0 recomputes it first. */
int dupl_cam_minmax_normalise(float* cam, float* mm, int32_t planes, int32_t HW, int32_t have_minmax, dupl_stream_t s);
/* cam_to_label / cam_to_label_dynamic_cls (cam_helper.py:8-55).  cam (b,C,h,w); cls_label (b,C); img_box (b,4) int32
 * [y0,y1,x0,x1] or NULL (then only bkg_thre applies, no box paste); high_thre (b,) floats; label out (b,h,w) int64;
 * valid_cam out (b,C,h,w) or NULL. */
int dupl_cam_to_label(const float* cam, const float* cls_label, const int32_t* img_box, const float* high_thre,
                      float bkg_thre, float low_thre, int32_t ignore_mid, int32_t ignore_index,
                      int64_t* label, float* valid_cam, int32_t b, int32_t C, int32_t h, int32_t w, dupl_stream_t s);
/* denormalize_img / denormalize_img2 (imutils.py:17-31): out = float(uint8_trunc(x*std+mean))/255, IEEE mul+add (no fma);
 * mean_std: HOST pointer to {mean[3], std[3]} or NULL for the reference's defaults (ImageNet, 0-255 scale) */
int dupl_denormalize_img(const float* x, float* out, int32_t B, int32_t HW, const float* mean_std, dupl_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * PAR (model/PAR.py:26-91) and the refine wrappers (cam_helper.py:338-440).
 * A "job" is one PAR.forward call of the reference: (image job_img[j], K = job_K[j] mask channels). */
/* aff [B][8*ndil][h][w] = softmax_k(-mean_c((|I_k-I|/(std_k+1e-8)/0.3)^2)) + pos_term[k]   (PAR.py:66-85).
 * imgs (B,3,h,w); dilations HOST int array; pos_term DEVICE [8*ndil] = 0.01*softmax(pos) (constant). */
int dupl_par_affinity(const float* imgs, float* aff, const int32_t* dilations, int32_t ndil, const float* pos_term,
                      int32_t B, int32_t h, int32_t w, dupl_stream_t s);
/* one propagation iteration for all jobs: out[j][k] = sum_n aff[job_img[j]][n] * in[j][k][nbr_n]   (PAR.py:87-89).
 * in/out [njobs][Kmax][h][w]; job_img, job_K DEVICE int arrays. */
int dupl_par_propagate(const float* aff, const float* in, float* out, const int32_t* job_img, const int32_t* job_K,
                       const int32_t* dilations, int32_t ndil, int32_t njobs, int32_t Kmax, int32_t h, int32_t w,
                       dupl_stream_t s);
/* refine pre (cam_helper.py:358-367,406-415): per job, channel 0 = background threshold (thr[j], or the down-sampled
 * thr_map[img] (b,1,H,W) when non-NULL), channel k>0 = cams[img][keys[j][k]-1] (cams (b,C,H,W) already multiplied by the
 * image labels); bilinear (H,W) -> (h,w) = (H // down_scale, W // down_scale), align_corners False; softmax over the K
 * channels -> masks [njobs][Kmax][h][w]. keys DEVICE [njobs][Kmax]. */
int dupl_refine_pre(const float* cams, const float* thr_map, const float* thr, const int32_t* job_img,
                    const int32_t* job_K, const int32_t* keys, int32_t njobs, int32_t Kmax, float* masks, int32_t C,
                    int32_t H, int32_t W, int32_t h, int32_t w, dupl_stream_t s);
/* refine post (cam_helper.py:434-440 + box paste :376-379): bilinear (h,w) -> (H,W) -> first argmax -> keys -> float label
 * [njobs][H][W], ignore_index outside box[job_img[j]] (box DEVICE (b,4) int32). */
int dupl_refine_post(const float* masks, const int32_t* job_img, const int32_t* job_K, const int32_t* keys,
                     int32_t njobs, int32_t Kmax, const int32_t* box, float ignore_index, float* label, int32_t h,
                     int32_t w, int32_t H, int32_t W, dupl_stream_t s);
/* merge (cam_helper.py:381-383): out = lab_h; out[lab_h==0] = ignore; out[lab_h+lab_l==0] = 0 */
int dupl_refine_merge(const float* lab_h, const float* lab_l, float* out, float ignore_index, int64_t n, dupl_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * Losses (model/losses.py; train_final_voc.py:210-216,247-254,345-352). */
/* PTC (losses.py:6-21 + cam_helper.py:323-335 fused: the int64 (b,hw,hw) mask is never built).
 * cos (b,hw,hw) = SIGNED xhat^T xhat from dupl_gemm_f32; pairs are classified from label (b,hw) int64, or -- the
 * reference API get_masked_ptc_loss(inputs, mask) -- from an explicit mask (b,hw,hw) int64 (1 pos / 0 neg / else ignored)
 * when mask != NULL.
 * sums: DUPL_LOSS_SUMS_FLOATS (136) floats, zero-filled by the caller; after the launch sums[0..3] = {sum_pos |cos|, n_pos,
 * sum_neg |cos|, n_neg}.  The rest is the reduction's own state (ABI 3): the blocks accumulate in 64-bit fixed point, so the
 * four results do not depend on the order the blocks retire in -- bit-reproducible in every mode.
 * finish (ABI 4): 1 = the last block to retire also writes the loss VALUE 0.5 (1 - s0 / (s1 + 1)) + 0.5 s2 / (s3 + 1) (losses.py:17-21,
 * every operation rounded as the torch expression rounds it) to sums[6]; 0 = sums only. */
int dupl_ptc_reduce(const float* cos, const int64_t* label, const int64_t* mask, int32_t ignore_index, float* sums,
                    int32_t b, int32_t hw, int32_t finish, dupl_stream_t s);
/* backward of get_masked_ptc_loss (autograd of losses.py:6-21), in place: cos_signed -> d loss/d cos_signed = sign(cos) * (pos ? -0.5*g/(n_pos+1) : neg ? 0.5*g/(n_neg+1) : 0), g = gscale[0] */
int dupl_ptc_bwd_mask(float* cos_signed, const int64_t* label, const int64_t* mask, int32_t ignore_index,
                      const float* sums, const float* gscale, int32_t b, int32_t hw, dupl_stream_t s);
/* F.normalize(p=2, dim=channel, eps) of get_masked_ptc_loss (losses.py:11) and its adjoint, on token-major rows: row r of image i at x + i*img_stride + r*ldx;
 * xhat [rows][c] dense, norm [rows]. */
int dupl_l2norm_rows_fwd(const float* x, float* xhat, float* norm, int64_t rows, int32_t c, int64_t ldx,
                         int32_t rows_per_img, int64_t img_stride, float eps, dupl_stream_t s);
/* adjoint of the F.normalize above (autograd of losses.py:11) */
int dupl_l2norm_rows_bwd(const float* dxhat, const float* xhat, const float* norm, float* dx, int64_t rows, int32_t c,
                         int64_t ldx, int32_t rows_per_img, int64_t img_stride, float eps, int32_t accumulate,
                         dupl_stream_t s);
/* fused bilinear upsample (align_corners False) + CE, bg/fg balanced (get_seg_loss, losses.py:24-39 on
 * F.interpolate(segs, (H,W)), train_final_voc.py:345-352).  logits token-major [b][h*w][C1]; label (b,H,W) float32
 * or int64 (is_i64).  sums: DUPL_LOSS_SUMS_FLOATS zero-filled floats, sums[0..3] = {ce_bg, n_bg, ce_fg, n_fg} after the
 * launch (order-independent fixed-point reduction, see dupl_ptc_reduce). */
int dupl_seg_loss_fwd(const float* logits, const void* label, int32_t is_i64, int32_t ignore_index, float* sums,
                      int32_t b, int32_t C1, int32_t h, int32_t w, int32_t H, int32_t W, int32_t flip, int32_t finish,
                      dupl_stream_t s);
/* ^ finish (ABI 4): the last block also writes the loss value to sums[6]: 2 = get_seg_loss' 0.5 (s0 / (s1 + 1e-6) + s2 / (s3 + 1e-6))
 * (losses.py:33-39), 3 = the plain mean (s0 + s2) / max(s1 + s3, 1) of the consistency loss (train_final_voc.py:430-436), 0 = sums only */
/* same fused upsample + CE, but the per-pixel value ce_map (b,H,W) (0 where label == ignore): the detached
 * ce_criterion(segs, refined_label) maps the GMM noise filter is fitted on (train_final_voc.py:360-361).
 * flip != 0 reads the low-res logits w-flipped (torch.flip(segs_aug, dims=[3]), :407-408). */
int dupl_seg_ce_map(const float* logits, const void* label, int32_t is_i64, int32_t ignore_index, float* ce_map,
                    int32_t b, int32_t C1, int32_t h, int32_t w, int32_t H, int32_t W, int32_t flip, dupl_stream_t s);
/* consistency-regularisation targets (train_final_voc.py:416-426): pseudo = argmax of the up-sampled logits where the
 * OTHER student's refined label == ignore and max-softmax > conf_thr, else ignore; count[0] += kept pixels. */
int dupl_seg_pseudo_label(const float* logits, const float* other_label, int32_t ignore_index, float conf_thr,
                          int64_t* out_label, float* count, int32_t b, int32_t C1, int32_t h, int32_t w, int32_t H,
                          int32_t W, dupl_stream_t s);
/* The phase-C label-noise filter on the device (train_final_voc.py:358-394), one workgroup per image: fits the
 * reference's sklearn GaussianMixture(n_components=2, max_iter=em_iters, tol=em_tol, reg_covar, random_state=seed)
 * (k-means++ / Lloyd initialisation included) on the ce_map values of the pixels with label != 0, != ignore_index and
 * ce > min_ce -- only when more than min_count of them exist -- and, when |mean0 - mean1| > valid_thre, sets
 * label = ignore_index where P(high-mean component | ce) > gamma and label != 0.  label (B,HW) float32 in/out.
 * u0,u1,u2 = the three uniforms numpy's RandomState(seed) yields (random_sample(), uniform(size=2)): sklearn's
 * k-means++ consumes exactly those.  xs_scratch (B*HW floats) and lab_scratch (B*HW bytes) are work space.
 * stats [B][DUPL_GMM_STATS] = {n selected, filtered?, mean0, mean1, cov0, cov1, weight0, weight1, EM iterations,
 * Lloyd iterations, mean log-likelihood, k-means centre0, centre1, #pixels relabelled, first seed index, second}. */
#define DUPL_GMM_STATS 16
/* (train_final_voc.py:363-394, see the comment above).  seeding selects the k-means++ variant: 0 = sklearn >= 1.2 (the three
 * uniforms above), 1 = sklearn 1.0.2, the version the reference pins (requirements.txt:4): first centre = RandomState.randint(n),
 * numpy's masked rejection on 32-bit MT19937 words, the trial uniforms from the words after it; mt_raw_host = the first 48 32-bit
 * outputs of RandomState(seed) (a HOST array, copied into the launch; NULL with seeding 0). */
int dupl_gmm_noise_filter(const float* ce_map, float* label, float* xs_scratch, uint8_t* lab_scratch, float* stats,
                          int32_t B, int32_t HW, int32_t ignore_index, float min_ce, int32_t min_count,
                          float valid_thre, float gamma, float reg_covar, float em_tol, int32_t em_iters, double u0,
                          double u1, double u2, int32_t seeding, const uint32_t* mt_raw_host, dupl_stream_t s);
/* label[i] = value where mask[i] != 0 (noise-mask write-back, train_final_voc.py:381,393) */
int dupl_mask_fill(float* label, const uint8_t* mask, float value, int64_t n, dupl_stream_t s);
/* dlogits (token-major, zero first) += gscale[0] * d loss / d logits (wave-reduced atomics when H/h, W/w are multiples
 * of 16; per-lane atomics otherwise).  balanced = 1: get_seg_loss' 0.5*(bg mean + fg mean); 0: plain mean over the
 * valid pixels (sum CE / count: the consistency loss, train_final_voc.py:430-434). */
int dupl_seg_loss_bwd(const float* logits, const void* label, int32_t is_i64, int32_t ignore_index, const float* sums,
                      const float* gscale, float* dlogits, int32_t b, int32_t C1, int32_t h, int32_t w, int32_t H, int32_t W,
                      int32_t flip, int32_t balanced, int32_t deterministic, dupl_stream_t s);
/* nn.CosineSimilarity(dim=-1) over the n tokens of every (image, channel) (train_final_voc.py:247-254); a, b
 * token-major (element (i, t, c) at + i*img_stride + t*ld + c).  out [B][c]; stats [B][c][3] = {dot, |a|^2, |b|^2}. */
int dupl_cos_sim_fwd(const float* a, const float* b, float* out, float* stats, int32_t B, int32_t n, int32_t c,
                     int64_t ld, int64_t img_stride, float eps, dupl_stream_t s);
/* autograd of train_final_voc.py:251-252 -- gradient wrt b only (a is detached): db (+)= g[0]*gmul * d cos / d b */
int dupl_cos_sim_bwd(const float* a, const float* b, const float* stats, const float* g, float gmul, float* db, int32_t B,
                     int32_t n, int32_t c, int64_t ld, int64_t img_stride, float eps, int32_t accumulate, dupl_stream_t s);
/* The loss assembly of an iteration (train_final_voc.py:210-216,247-254,451-456: cls = l1 + l2 + l3 + l4, sim = (1 + c1) + (1 + c2),
 * loss = 1.0 cls + w_ptc ptc + w_seg seg + 0.1 sim [+ 0.05 reg]) as ONE launch over device scalars (ABI 4):
 *   v_i = add[i] + *terms[i]  (add[i] == 0: the term as it is);   G_g = the v_i with group[i] == g, summed left to right in list order;
 *   total = ((weight[0] G_0 + weight[1] G_1) + weight[2] G_2) + ...   -- every operation one fp32 rounding, like the torch expression.
 * Forward (total != NULL, gterm == NULL): total[0], gsums[g] = G_g (gsums may be NULL).   Backward (total == NULL): gterm[i] = g[0] *
 * weight[group[i]].  terms / add / group / weight are HOST arrays (n_terms, n_groups <= DUPL_LOSS_TERMS_MAX); *terms[i], total, gsums, g,
 * gterm device memory. */
#define DUPL_LOSS_TERMS_MAX 16
int dupl_loss_total(const float* const* terms, const float* add, const int32_t* group, int32_t n_terms, const float* weight,
                    int32_t n_groups, float* total, float* gsums, const float* g, float* gterm, dupl_stream_t s);
/* the .mean() of train_final_voc.py:251-252: loss[0] += mul * sum(x[0..n)) */
int dupl_mean_accum(const float* x, float* loss, int64_t n, float mul, dupl_stream_t s);
/* F.multilabel_soft_margin_loss (train_final_voc.py:210-216; mean over classes then batch): loss[0] += value (if loss != NULL);
 * dlogits (if != NULL) = gscale[0] * d loss / d logits */
int dupl_multilabel_soft_margin(const float* logits, const float* target, float* loss, float* dlogits, const float* gscale,
                                int32_t b, int32_t C, dupl_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * LargeFOV helpers (conv_head.py:32-41): 3x3 dilated conv (zero padding = dilation) as im2col + GEMM with
 * token-major activations.  Column order is (c, tap) so conv weight (Cout,Cin,3,3) is the GEMM B operand in place. */
/* conv6 / conv7 of LargeFOV.forward (conv_head.py:34-39): x: pixel p, channel c of image b at
 * x + b*img_stride + p*ld + c  ->  col [B*h*w][9*Cin] */
int dupl_im2col_dil3(const float* x, float* col, int32_t B, int32_t h, int32_t w, int32_t Cin, int32_t dil, int64_t ld,
                     int64_t img_stride, dupl_stream_t s);
/* adjoint (autograd of conv_head.py:34-39): dx (+)= gather of dcol, zeroed where relu_of (same layout as dx, post-ReLU activations) <= 0 if non-NULL */
int dupl_col2im_dil3(const float* dcol, float* dx, int32_t B, int32_t h, int32_t w, int32_t Cin, int32_t dil, int64_t ld,
                     int64_t img_stride, int32_t accumulate, const float* relu_of, dupl_stream_t s);

/* ---------------------------------------------------------------------------------------------
 * Optimiser (utils/optimizer.py:38-68 -> torch.optim.AdamW) and small element-wise helpers. */
/* PolyWarmupAdamW.step -> torch.optim.AdamW.step (utils/optimizer.py:38-68) fused over one flat fp32 segment (16-byte
 * aligned); bc1 = 1-beta1^t, bc2_sqrt = sqrt(1-beta2^t) (host, double).  p_hi / p_lo (both or neither; 8-byte aligned): the updated
 * parameters are also written as the f16x3 operand planes of the next forward -- format 0 (plane_exp 0) or format 1 (p * 2^plane_exp,
 * unscaled lo), bit-identical to dupl_split_f16x2 / dupl_split_f16x2b of the updated segment.
 * grad_scale (> 0; ABI 4): the gradient is taken as g * grad_scale -- one rounding, written back to g -- i.e. DDP's division by
 * the world size (train_final_voc.py:153-155) folded into the update of a range whose all-reduce has completed; 1 = g is read only */
int dupl_adamw(float* p, float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
               float eps, float wd, float bc1, float bc2_sqrt, void* p_hi, void* p_lo, int32_t plane_exp, float grad_scale,
               dupl_stream_t s);
/* optimizer.zero_grad() (train_final_voc.py:470) and buffer initialisation: p[0..n) = v */
int dupl_fill(float* p, float v, int64_t n, dupl_stream_t s);
/* y += a*x: the aux-branch gradient joining the residual stream (autograd of vit.py:316-326) */
int dupl_axpy(float* y, const float* x, float a, int64_t n, dupl_stream_t s);
/* y *= a: DDP's division of the all-reduced gradients by the world size (train_final_voc.py:155) */
int dupl_scale(float* y, float a, int64_t n, dupl_stream_t s);

/* ------------------------------------------------------------------ validation / evaluation */
/* torch.argmax(F.interpolate(logits, (H,W), 'bilinear', align_corners=False), dim=1) fused: logits (B,C,h,w) NCHW ->
 * out (B,H,W) int64 (utils/train_helper.py:143-145, the Seg_k prediction of validate_siamase). */
int dupl_upsample_argmax(const float* logits, int64_t* out, int32_t B, int32_t C, int32_t h, int32_t w, int32_t H, int32_t W,
                         dupl_stream_t s);
/* one scale of tools/eval_seg_voc.py:58-72 / tools/eval_seg_coco_ddp.py:80-119: segs (2,C,h,w) = the logits of
 * [x; flip(x)]; v = up(segs[0]) + flip(up(segs[1])) at (H,W); acc (C,H,W) = v (mode 0), max(acc, v) (mode 1: VOC's max
 * over scales at label size), acc + v (mode 2: COCO's sum over scales at the scale-1 logit size). */
int dupl_msc_seg_accum(const float* segs, float* acc, int32_t C, int32_t h, int32_t w, int32_t H, int32_t W, int32_t mode,
                       dupl_stream_t s);
/* torch.argmax(seg, dim=1) of tools/eval_seg_voc.py:77-78: x (B,C,HW) -> out (B,HW) int64, first maximum wins */
int dupl_argmax_channels(const float* x, int64_t* out, int32_t B, int32_t C, int64_t HW, dupl_stream_t s);
/* utils/evaluate.py:9-16 (_fast_hist) accumulated on the device: hist[t*nc + p] += 1 over the n pixels with
 * 0 <= gt < nc (predictions outside [0,nc) are skipped as well).  hist: nc*nc int64, zeroed by the caller. */
int dupl_confusion_accum(const int64_t* gt, const int64_t* pred, int64_t n, int32_t num_classes, int64_t* hist,
                         dupl_stream_t s);
/* utils/evaluate.py:4-6 (sklearn f1_score of one multi-hot row): sum[0] += 2TP/(2TP+FP+FN) of (logits > 0) vs label,
 * per row of (B,C). */
int dupl_multilabel_f1_accum(const float* logits, const float* label, int32_t B, int32_t C, float* sum, dupl_stream_t s);

/* Attention backward as fp32-equivalent f16x3 split products (csrc/attn_split_bwd.hip; head dim 64, N <= 2048): autograd of
 * vit.py:123-135.  qkv_hi / qkv_lo: planes of the qkv GEMM output saved by the forward; out / dout: fp32 attention output
 * and its gradient ([B*N][H*hd]); do_hi / do_lo + do_slot: dout as planes scaled with target_exp 4 (dupl_split_prepare;
 * do_slot = its {scale, 1/scale,..} record); lse from the forward; delta: B*H*N floats of scratch (the only one: K, Q and dO are
 * read in place as k-major operands where a product contracts over their rows); dqkv [B*N][3*H*hd] fp32 receives dq | dk | dv.  amax_out != NULL: max |dqkv| is raised into
 * *amax_out (the amax word of a scale slot, see dupl_split_prepare amax_mode 1). */
int dupl_attention_bwd16(const void* qkv_hi, const void* qkv_lo, const float* out, const float* dout, const void* do_hi,
                         const void* do_lo, const float* do_slot, const float* lse, float* delta, float* dqkv, int32_t B, int32_t N,
                         int32_t H, int32_t hd, float scale, void* amax_out, dupl_stream_t stream);

/* ------------------------------------------------------------------ per-step strong augmentation
 * utils/imutils.py:305-317 augment_data_strong / utils/randomaug.py RandAugment on the device: planar uint8 images
 * (3,H,W), Pillow's exact 8-bit arithmetic (see csrc/augment.hip).  The op sequence per image is drawn on the host. */
/* transforms.ToPILImage of a float tensor in [0,1] (imutils.py:306,311): out = (uint8)(x * 255) */
int dupl_aug_to_u8(const float* x, uint8_t* out, int64_t n, dupl_stream_t s);
/* mode 0: AutoContrast (randomaug.py:62-63), 1: Equalize (randomaug.py:70-71), in place; hist_scratch 768 uint32, lut_scratch 768 bytes */
int dupl_aug_lut_op(uint8_t* img, int32_t H, int32_t W, int32_t mode, uint32_t* hist_scratch, uint8_t* lut_scratch,
                    dupl_stream_t s);
/* Posterize (randomaug.py:92-95): PIL.ImageOps.posterize(img, bits), in place over n bytes */
int dupl_aug_posterize(uint8_t* img, int64_t n, int32_t bits, dupl_stream_t s);
/* Color (mode 0, randomaug.py:103-105), Contrast (1, :98-100), Brightness (2, :108-110): PIL.ImageEnhance.*(img)
 * .enhance(factor), in place; sum_scratch: 1 uint64 */
int dupl_aug_enhance(uint8_t* img, int32_t H, int32_t W, int32_t mode, float factor, uint64_t* sum_scratch, dupl_stream_t s);
/* Sharpness (randomaug.py:113-115): PIL.ImageEnhance.Sharpness(img).enhance(factor); out != in */
int dupl_aug_sharpness(const uint8_t* in, uint8_t* out, int32_t H, int32_t W, float factor, dupl_stream_t s);
/* transforms.ToTensor + Normalize(ImageNet mean/std) + torch.flip(dims=[2]) (imutils.py:307-315): out (3,H,W) float32 */
int dupl_aug_finish(const uint8_t* img, float* out, int32_t H, int32_t W, dupl_stream_t s);

/* ------------------------------------------------------------------ loader-side input pipeline (SURVEY 8f-3 ii)
 * The geometric part of VOC12ClsDataset / CocoClsDataset.__getitem__ (datasets/voc.py:134-186) on interleaved
 * uint8 images (H,W,3) as decoded, bit-exact with Pillow / numpy (see csrc/loader.hip).  The random draws and the
 * fixed-point coefficient tables of Pillow's resize are made on the host (dupl_amd/datasets/transforms.py). */
/* horizontal pass of PIL.Image.resize(..., BILINEAR) inside transforms._img_rescaling (transforms.py:63-76):
 * in (h,w,3) -> out (h,w2,3); coef (w2,ksize) int32 22-bit fixed point, bounds (w2,2) = (first input column, taps) */
int dupl_loader_resample_h(const uint8_t* in, uint8_t* out, const int32_t* coef, const int32_t* bounds, int32_t ksize,
                           int32_t h, int32_t w, int32_t w2, dupl_stream_t s);
/* vertical pass of the same resize fused with transforms.random_fliplr (transforms.py:103-116) and transforms.random_crop
 * (transforms.py:147-204; mean_rgb = 0): tmp (h,w2,3) -> out (crop,crop,3); the rescaled (h2,w2) image sits at
 * (h_pad,w_pad) of the zero canvas, the crop window starts at (h_start,w_start); coef (h2,ksize), bounds (h2,2) */
int dupl_loader_resample_v_crop(const uint8_t* tmp, uint8_t* out, const int32_t* coef, const int32_t* bounds, int32_t ksize,
                                int32_t w2, int32_t h2, int32_t flip, int32_t h_pad, int32_t w_pad, int32_t h_start,
                                int32_t w_start, int32_t crop, dupl_stream_t s);
/* in (H,W,3) uint8 -> out (3,H,W) float32.  mode 0: T.ToTensor + T.Normalize(ImageNet) of the train items
 * (datasets/voc.py:96-99,165); mode 1: transforms.normalize_img of the val items (transforms.py:45-52, voc.py:248) */
int dupl_loader_normalize(const uint8_t* in, float* out, int32_t H, int32_t W, int32_t mode, dupl_stream_t s);

/* Photometric half of the train transform: global_view1 of VOC12ClsDataset / CocoClsDataset (datasets/voc.py:101-114,
 * 145-146) = torchvision RandomApply([ColorJitter]) + RandomGrayscale + transforms.GaussianBlur (transforms.py:11-29), all
 * of which run in Pillow's 8-bit arithmetic; here in place on the interleaved uint8 crop (H,W,3), bit-exact with Pillow
 * (see csrc/photometric.hip).  The draws (which ops, order, factors) are made on the host in torchvision's order. */
/* ColorJitter brightness (mode 2) / contrast (1) / saturation (0): PIL.ImageEnhance.{Brightness,Contrast,Color}(img)
 * .enhance(factor); sum_scratch: 1 uint64 (contrast only) */
int dupl_photo_enhance(uint8_t* img, int32_t H, int32_t W, int32_t mode, float factor, uint64_t* sum_scratch,
                       dupl_stream_t s);
/* ColorJitter hue: img.convert("HSV"), h += shift (uint8 wrap, shift = uint8(hue_factor * 255)), convert("RGB") */
int dupl_photo_hue(uint8_t* img, int64_t n_px, int32_t shift, dupl_stream_t s);
/* RandomGrayscale: img.convert("L") replicated to 3 channels */
int dupl_photo_grayscale(uint8_t* img, int64_t n_px, dupl_stream_t s);
/* transforms.GaussianBlur: img.filter(PIL.ImageFilter.GaussianBlur(radius)); tmp: H*W*3 bytes of scratch; result in img */
int dupl_photo_gaussian_blur(uint8_t* img, uint8_t* tmp, int32_t H, int32_t W, float radius, dupl_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* DUPL_HIP_H */
