/*
 * mdm_hip.h -- C ABI of libmdm_hip.so: the MI355X (gfx950) kernels under the
 * masked-diffusion train step and reverse sampler.
 *
 * The reference (hytae1993/masked-diffusion-model) has no FFI: its hot path is
 * Python calling ATen ops.  Each entry point below replaces the ATen call sites
 * named in its comment (file:line relative to /root/reference/code).  The Python
 * host in masked-diffusion-model_amd/mdm binds these with ctypes and keeps the
 * reference's Scheduler / Sampler / Trainer / model(x,t).sample surface.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory unless the
 *     name ends in _host.  The caller owns all buffers (incl. workspaces).
 *   - `stream` is a hipStream_t passed as void*; all work is stream-ordered,
 *     nothing synchronises, nothing allocates -> every call is graph-capturable.
 *   - return 0 on success, negative on error; mdm_last_error() gives the
 *     thread-local message.  Nothing throws across the ABI.
 *   - dtype: MDM_F32 = 0 (exact-fp32 VALU contraction path, used for parity),
 *            MDM_BF16 = 1 (bf16 storage, MFMA 16x16x32 contractions, fp32 accumulate).
 *   - activations are NHWC with the channel count a multiple of 8; images enter
 *     and leave as NCHW fp32 exactly like the reference's tensors.
 */
#ifndef MDM_HIP_H
#define MDM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MDM_F32 0
#define MDM_BF16 1

const char* mdm_last_error(void);
int mdm_version(void);
/* number of visible HIP devices, or negative */
int mdm_device_count(void);

/* ------------------------------------------------------------------------- *
 * Contractions.  One descriptor drives the three MFMA/VALU contraction
 * kernels; it replaces F.conv2d (unet6.py:232-235) forward, its autograd
 * dgrad/wgrad, F.linear (unet6.py:170-171), the 1x1 convs and the two einsums
 * of AttentionBlock.qkv (unet6.py:316-324) plus their backward products.
 *
 * layout 0 (NT): D[m][n] = sum_k A[m][k] * B[n][k]      A rows gathered if conv
 * layout 1 (NN): D[m][n] = sum_k A[m][k] * B[k][n]      A rows gathered if conv
 * layout 2 (TN): D[m][n] = sum_k A[k][m] * B[k][n]      B rows gathered if conv
 *
 * conv != 0 turns the gathered operand into an implicit im2col over NHWC
 * sources: row index = (img, oy, ox) over OH x OW, reduction index k = tap*Ck+c
 * (layouts 0/1) or row index k = pixel with the tap taken from the grid
 * (layout 2, weight gradient).  Two sources give the channel concat of
 * unet6.py:501 without materialising it; ups=1 reads the source through a
 * nearest x2 upsample (unet6.py:472); stride/pad_t/pad_l cover SamePad2d +
 * stride-2 (unet6.py:257-272, 438-440); transposed=1 is the data-gradient form.
 * ------------------------------------------------------------------------- */
typedef struct mdm_gemm_desc {
    int32_t dtype;              /* MDM_F32 | MDM_BF16: element type of A, B, sources, resid and (unless out_f32) D */
    int32_t layout;             /* 0 NT, 1 NN, 2 TN */
    int32_t M, N, K;
    int32_t batch;              /* >= 1; blockIdx.z when conv == 0 */
    int64_t sA, sB, sD, sR;     /* batch strides in elements */
    const void* A; int32_t lda;
    int32_t f32_split;          /* dtype MDM_F32, layout 0: 1 = the register-staged fp32 kernel may multiply as bf16 hi / lo pairs (see B_split) */
    const void* B; int32_t ldb; int32_t _p1;
    /* implicit-GEMM gather */
    int32_t conv;
    int32_t OH, OW;             /* row space of the gathered operand */
    int32_t IH, IW;             /* logical source extent (after the virtual upsample) */
    int32_t KH, KW, stride, pad_t, pad_l, transposed, ups;
    int32_t C0, C1;             /* channels taken from src0 / src1 */
    int32_t Ck;                 /* reduction channels per tap (layouts 0/1) */
    const void* src0; const void* src1;
    int32_t ld0, ld1;           /* pixel pitch of each source in elements */
    int64_t wtap;               /* weight offset per tap in elements (layouts 0/1) */
    /* epilogue */
    void* D0; void* D1;         /* columns [0,N0) -> D0, [N0,N) -> D1 */
    int32_t ldd0, ldd1, N0;
    int32_t out_f32;            /* D is fp32 regardless of dtype */
    float alpha;
    int32_t acc0, acc1;         /* D += instead of D = */
    const float* bias;          /* [N] or NULL */
    const float* rowvec;        /* [M / rows_per_img][rv_ld] fp32 or NULL (time-embedding add, unet6.py:359) */
    int32_t rv_ld, rows_per_img;
    const void* resid;          /* [M][ldr] or NULL (residual add, unet6.py:333,362) */
    int32_t ldr;
    int32_t splitk;             /* reduction split over grid z (0 = choose; conv layouts 0/1 never split) */
    int64_t dtap;               /* layout 2 + conv: D0 offset per tap */
    /* split-K workspace (optional).  With it, every split writes its fp32 partial tile with plain
     * stores into ws[split][tap|batch][M][N] and a second kernel sums the splits into D0 -- float
     * atomics from dozens of workgroups into one tile serialise at the memory side AND sum in arrival
     * order.  Without it (or when D0 is not a dense [tap|batch][M][N] block) the reduction is not
     * split: there is no atomic fallback, results do not depend on scheduling. */
    void* ws; int64_t ws_bytes;
    /* layout 2, bf16: if set, dbias[m] += sum_k A[k][m] (the bias gradient of a convolution is the column
     * sum of dY, which the weight-gradient kernel already holds as MFMA fragments: one extra MFMA against
     * a ones-fragment per k-step in the tap-0 / first-column workgroups, fp32 atomics at the end). */
    float* dbias;
    /* Fused GroupNorm backward (unet6.py:291-293 backward) in the epilogue of the DATA GRADIENT of the conv that
     * consumes the normalised tensor, on the whole-image tiles of the 4x4 / 8x8 maps (mdm_gemm_can_fuse_gn_bwd says
     * whether this descriptor qualifies): the contraction result is d(z), z = silu?(GroupNorm(x)); instead of storing
     * it, the epilogue reads x, stats (mean, rstd per image and group), gamma, beta and writes dx into D0 (acc0 honoured),
     * adds dgamma / dbeta, and optionally the column sums of dx (gnb_sum_img[n * gnb_sum_ld + c] = sum_p dx,
     * gnb_sum_all[c] += sum_{n,p} dx) like mdm_groupnorm_bwd_sums.  gnb_add (with acc0 == 0): a tensor laid out like D0
     * whose values are added to dx (the gradient arriving over the residual branch, unet6.py:362).  NULL gnb_x = plain epilogue. */
    const void* gnb_x; const float* gnb_stats; const float* gnb_gamma; const float* gnb_beta;
    float* gnb_dgamma; float* gnb_dbeta; float* gnb_sum_img; float* gnb_sum_all;
    int32_t gnb_G, gnb_silu, gnb_sum_ld, _p3;
    const void* gnb_add;
    /* Fused GroupNorm FORWARD in the epilogue of the conv that PRODUCES the tensor, same tiles (mdm_gemm_can_fuse_gn_fwd):
     * after the usual epilogue has stored y (bf16), the workgroup normalises the values it just rounded:
     * gnf_out = silu?(GroupNorm_G(y) * gamma + beta) (bf16, same shape as y), gnf_stats[n][g] = (mean, rstd).
     * NULL gnf_out = plain epilogue. */
    void* gnf_out; const float* gnf_gamma; const float* gnf_beta; float* gnf_stats;
    int32_t gnf_G, gnf_silu; float gnf_eps; int32_t _p4;
    /* dtype MDM_F32, optional: the filters of B once more, written by mdm_split_shadow (same offsets and leading dimension as B).
     * With it, a 3x3 stride-1 forward convolution that qualifies for the halo kernel multiplies on the bf16 matrix pipe with fp32
     * storage and fp32 accumulation: every operand x = hi + lo (hi = bf16(x), lo = bf16(x - hi)), a product = hi*hi + hi*lo + lo*hi
     * -- relative error <= ~2^-16 per product instead of 2^-24, at a fifth of the matrix time of the exact path.  Descriptors that do
     * not qualify ignore it and run the exact fp32 path on B.  NULL = exact fp32 everywhere (the parity path).
     * (NaN operands stay NaN; an infinite operand becomes NaN -- Inf - bf16(Inf) is NaN -- where the exact path would give +-Inf.) */
    const void* B_split;
} mdm_gemm_desc;

int mdm_gemm(const mdm_gemm_desc* desc_host, void* stream);
/* Two INDEPENDENT contractions (neither reads what the other writes), in one launch where the pair is a 3x3 halo convolution
 * `a` and a 1x1 convolution `b` of the shapes a ResidualBlock produces (conv1 next to the skip projection, unet6.py:350-362;
 * conv2's data gradient next to the skip projection's): the workgroups of `b` follow those of `a` in the same grid.
 * Every other pair runs as mdm_gemm(a) then mdm_gemm(b): the results are the same either way. */
int mdm_gemm_pair(const mdm_gemm_desc* a_host, const mdm_gemm_desc* b_host, void* stream);
/* What mdm_gemm would choose for this descriptor given unlimited workspace: the split count and the
 * workspace bytes it needs (0 when it would not use partial slabs).  No launch. */
int mdm_gemm_plan(const mdm_gemm_desc* desc_host, int* splitk_out, int64_t* ws_bytes_out);
/* 1 if mdm_gemm would run this descriptor (a bf16 3x3 data gradient through the transposed filters) on the whole-image
 * halo tiles AND groups of G channels are whole inside a 64-channel tile, i.e. the gnb_* epilogue may be used; else 0. */
int mdm_gemm_can_fuse_gn_bwd(const mdm_gemm_desc* desc_host, int G);
/* the same question for the gnf_* epilogue of a bf16 3x3 forward convolution */
int mdm_gemm_can_fuse_gn_fwd(const mdm_gemm_desc* desc_host, int G);

/* ------------------------------------------------------------------------- *
 * A GROUP of weight gradients in ONE launch (autograd's conv2d weight gradient, unet6.py:232-235).
 * The weight gradient of a convolution only READS dY and the layer input; both stay in memory until the end of
 * the backward pass, so the host may collect the (layout 2, conv) descriptors of a whole stretch of the backward
 * and run them together: every contraction is cut into work items (output tile x filter tap x k-range), the items of
 * all layers are sorted longest first and launched as one flat grid, followed by ONE launch that sums the split-K
 * partial slabs of the layers that were split.  The group fills the chip, so a layer needs only the k-splits that
 * balance asks for (descriptor.splitk > 0 is honoured, 0 = the per-layer rule of mdm_gemm).
 *   mdm_wgrad_group_accepts   1 if the descriptor can be a member (bf16, layout 2, conv, dense fp32 [tap][M][N] output,
 *                             geometry of the linear-gather kernel), else 0 (run it through mdm_gemm instead).
 *   mdm_wgrad_group_create    resolves the n HOST descriptors (operands must already be the final device pointers;
 *                             a descriptor with splitk > 1 needs its own ws slice), builds the item table and copies
 *                             both into dev_buf (device memory owned by the caller, kept alive as long as the handle).
 *                             *need_bytes_out = bytes dev_buf must hold; with dev_buf == NULL or too small nothing is
 *                             built (*handle_out = NULL, return 0): call once to size, once to build.  Synchronous
 *                             (hipMemcpy): call it outside stream capture.
 *   mdm_wgrad_group_launch    the launches of the group, stream-ordered and capturable.  D0 = (acc0 ? D0 : 0) + dW.
 *   mdm_wgrad_group_destroy   frees the host-side handle (not dev_buf).
 * Nine-tap form (round 3).  3x3 stride-1 members (maps 8 / 16 / 32 wide, overwrite form) whose group gives every CU a share
 * of >= MDM_TAPS_MIN_SHARE (default 48) 64-pixel slabs run all nine taps in one pass over dY and the input (wgrad_taps_body):
 * the group is then ONE persistent launch of one workgroup per CU walking its column of the item table -- equal contiguous
 * shares of the (tile, slab) space, the remaining per-tap items in the same queues -- plus the split-K sum of per-tap layers
 * and tile_parts_reduce_kernel, which adds the fp32 partial slots of the tiles a share boundary cut.  The slots live in
 * dev_buf too (they are part of *need_bytes_out: ~80 MB for the whole cfg2 backward), descriptor.splitk / ws of such a member
 * are ignored.  Same results as the per-tap form up to fp32 summation order.  Environment: MDM_WGRAD_TAPS=0 keeps every layer
 * on the per-tap kernels; MDM_TAPS_DEBUG=1 prints the schedule of each group; MDM_WGRAD_RESERVE_CUS=r builds the persistent nine-tap
 * launch for CUs - r workgroups (data-parallel runs: room for the collective's kernels beside it).
 * ------------------------------------------------------------------------- */
int mdm_wgrad_group_accepts(const mdm_gemm_desc* desc_host);
int mdm_wgrad_group_create(const mdm_gemm_desc* descs_host, int n, void* dev_buf, int64_t dev_bytes,
                           int64_t* need_bytes_out, void** handle_out);
int mdm_wgrad_group_launch(void* handle, void* stream);
int mdm_wgrad_group_destroy(void* handle);

/* ------------------------------------------------------------------------- *
 * A CHAIN of small-map convolutions as one persistent launch.  The trunk of unet6 (4x4 and 8x8 maps: ResidualBlocks unet6.py:336-362,
 * AttentionBlock projections :296-333, the level loop :478-506) is ~130 launches of 7-16 us per step at 32 images per GPU, most of
 * each being launch boundary, prologue and drain.  A chain takes a run of consecutive mdm_gemm / mdm_gemm_pair calls whose tiles are
 * 64 output pixels = whole images (the 64 x 32 halo tiles and 64 x 64 1x1 tiles mdm_gemm gives them) and runs them from ONE grid of
 * resident workgroups: phase p's block for a set of images starts as soon as the blocks of phase p-1 that cover THOSE images have
 * finished (per-image arrival counters, agent-scope release / acquire), with no grid-wide barrier and no dependence on where a
 * workgroup is placed.  The arithmetic is that of the separate launches, bit for bit.
 *   descs_host: the descriptors of all phases back to back; roles[p] in {1, 2} = how many of them phase p takes (2 = the pair
 *   mdm_gemm_pair would fuse: a 3x3 convolution + a 1x1 projection).  Every phase must pass mdm_chain_accepts.  dev_buf / need_bytes_out
 *   as in mdm_wgrad_group_create (size query with dev_buf = NULL); the buffer holds the descriptor copies, the phase table and the
 *   counters, which mdm_chain_launch clears (one small launch) before the persistent one.  mdm_chain_status: 0 unless a bounded wait
 *   inside some launch timed out (the launch then still ended, with wrong results).
 * ------------------------------------------------------------------------- */
int mdm_chain_accepts(const mdm_gemm_desc* a_host, const mdm_gemm_desc* b_host /* NULL: a single convolution */);
int mdm_chain_create(const mdm_gemm_desc* descs_host, const int* roles, int n_phases, void* dev_buf, int64_t dev_bytes,
                     int64_t* need_bytes_out, void** handle_out);
int mdm_chain_launch(void* handle, void* stream);
int mdm_chain_status(void* handle, unsigned* err_out);
int mdm_chain_destroy(void* handle);

/* ------------------------------------------------------------------------- *
 * GroupNorm(32, eps) [+ SiLU]  (unet6.py:291-293, 358, 360, 330, 505)
 * x = concat(src0[C0], src1[C1]) along channels, NHWC, P = H*W pixels per image.
 * stats: [N][G][2] fp32 (mean, rstd).  One kernel per call, every sum in a fixed order (no float atomics); `ws` is
 * not used by the forward (may be NULL).
 * ------------------------------------------------------------------------- */
int mdm_groupnorm_fwd(int dtype, const void* src0, int C0, const void* src1, int C1,
                      int N, int P, int G, float eps, const float* gamma, const float* beta,
                      int silu, void* y, float* stats, float* ws, void* stream);
/* dx -> dst0/dst1 (channel split like the sources), acc flags add into them;
 * dgamma/dbeta are ACCUMULATED: bf16, one fp32 atomic per channel per image; fp32 (dtype MDM_F32), NO atomics -- per-image partial
 * sums go to `ws` (REQUIRED there, >= 3*N*C floats) and a second launch adds them in image order, so the fp32 path gives the same bits
 * on every run.  (bf16: `ws` may be NULL.) */
int mdm_groupnorm_bwd(int dtype, const void* src0, int C0, const void* src1, int C1,
                      int N, int P, int G, const float* gamma, const float* beta, int silu,
                      const void* dy, const float* stats, void* dst0, int acc0, void* dst1, int acc1,
                      float* dgamma, float* dbeta, float* ws, void* stream);

/* floats the three backward entry points write into `ws` for an [N][P][C] problem of this dtype (0: `ws` may be NULL).  Replaces the
 * sizes quoted in prose above: the entry points cannot check a size they are not given. */
int64_t mdm_groupnorm_bwd_ws_floats(int dtype, int N, int C);

/* Same, and the column sums of the dx it writes (dx = the complete gradient of a conv output: acc0 == 0, one
 * source): sum_img[n*sum_ld + c] = sum_p dx (the time-embedding gradient, unet6.py:359) and
 * sum_all[c] += sum_{n,p} dx (that conv's bias gradient).  Either may be NULL. */
int mdm_groupnorm_bwd_sums(int dtype, const void* src0, int C0, const void* src1, int C1,
                           int N, int P, int G, const float* gamma, const float* beta, int silu,
                           const void* dy, const float* stats, void* dst0, int acc0, void* dst1, int acc1,
                           float* dgamma, float* dbeta, float* sum_img, int sum_ld, float* sum_all, float* ws,
                           void* stream);
/* The general form: dst = add + dx, where add0 / add1 (NULL = none) are tensors laid out like dst0 / dst1 -- dst itself
 * gives the accumulating form above; another tensor adds the gradient that arrives over a residual branch
 * (x + block(x), unet6.py:333, 362) WITHOUT modifying that tensor, which a later (grouped) weight gradient still reads. */
int mdm_groupnorm_bwd_add(int dtype, const void* src0, int C0, const void* src1, int C1,
                          int N, int P, int G, const float* gamma, const float* beta, int silu,
                          const void* dy, const float* stats, void* dst0, const void* add0, void* dst1, const void* add1,
                          float* dgamma, float* dbeta, float* sum_img, int sum_ld, float* sum_all, float* ws,
                          const void* add0b, void* stream);
/* add0b (NULL = none): a SECOND tensor laid out like dst0 that is added too -- an activation with two forward consumers and a
 * residual join receives dst0 (accumulate) + the residual branch's dY + dx in one pass. */

/* ------------------------------------------------------------------------- *
 * Fused single-head attention of AttentionBlock.qkv (unet6.py:316-324): o = softmax(q k^T * scale) v over L tokens,
 * qkv = [N][L][3C] (q | k | v) as project_in writes it (NHWC), o = [N][L][C]; bf16, C in {32, 64, 128, 256},
 * L % 16 == 0 (mdm_attn_supported).  One forward launch (online softmax over 64-key tiles, the scores never reach
 * memory), lse[N][L] = log-sum-exp of the scaled scores per query (fp32) is kept for the backward.
 * Backward: two launches -- dQ (also writes delta[N][L] = sum_d dO*O) and dK/dV -- recomputing P from q, k, lse;
 * dqkv = [N][L][3C] is written completely (dq | dk | dv), nothing is accumulated.
 * Replaces both einsums, the softmax and `.contiguous()` of unet6.py:319-324 and their autograd backward.
 * ------------------------------------------------------------------------- */
int mdm_attn_supported(int dtype, int L, int C);
/* Exact-fp32 forward of the same attention for short sequences (L in {16, 32, 48, 64}, C in {64, 128, 256}: the 8x8 / 4x4 attention
 * blocks on the fp32 path): one launch on v_mfma_f32_16x16x4_f32 instead of two batched contractions and a softmax launch.  It also
 * writes the probabilities S[N][L][L] (fp32) -- what the unfused fp32 backward (softmax_bwd + three contractions) reads. */
int mdm_attn_f32_small_supported(int L, int C);
int mdm_attn_f32_small_fwd(const float* qkv, float* o, float* S, int N, int L, int C, float scale, void* stream);
int mdm_attn_fwd(int dtype, const void* qkv, void* o, float* lse, int N, int L, int C, float scale, void* stream);
int mdm_attn_bwd(int dtype, const void* qkv, const void* o, const void* d_o, const float* lse, float* delta, void* dqkv,
                 int N, int L, int C, float scale, void* stream);

/* The same attention with MANY SMALL heads on separate q, k, v, o tensors [N][L][C], head h = channels [h D, (h+1) D),
 * D = C / heads in {8, 16, 32}, any L, bf16 or fp32: diffusers' attention blocks inside UNet2DModel (attention_head_dim = 8;
 * reference utils/model.py:24-32 builds them).  lse / delta are [N][heads][L] fp32.  dq, dk, dv are written completely. */
int mdm_attn_mh_fwd(int dtype, const void* q, const void* k, const void* v, void* o, float* lse, int N, int L, int C,
                    int heads, float scale, void* stream);
int mdm_attn_mh_bwd(int dtype, const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                    float* delta, void* dq, void* dk, void* dv, int N, int L, int C, int heads, float scale, void* stream);

/* row softmax of S[rows][L] in place (unet6.py:320-322), and its backward
 * dS = P * (dP - sum_j dP*P) written over dP. */
int mdm_softmax_fwd(int dtype, void* S, int rows, int L, void* stream);
int mdm_softmax_bwd(int dtype, const void* P, void* dP, int rows, int L, void* stream);

/* y[N][dim] = [sin(t*f) | cos(t*f)], f_i = exp(-i*ln(1e4)/(dim/2-1))  (unet6.py:18-34); fp32 */
int mdm_timestep_embedding(const float* t, int N, int dim, float* y, void* stream);
/* general form: flip_sin_to_cos puts the cosines first, the exponent is -i ln(1e4) / (dim/2 - freq_shift)
 * (diffusers `Timesteps(flip_sin_to_cos=True, downscale_freq_shift=0)` as UNet2DModel builds it; (0, 1) = the call above) */
int mdm_timestep_embedding2(const float* t, int N, int dim, int flip_sin_to_cos, float freq_shift, float* y, void* stream);
/* fp32 SiLU on small [n] vectors (unet6.py:397, 359) and its backward (dx = dy * silu'(x), acc adds) */
int mdm_silu_fwd(const float* x, float* y, int64_t n, void* stream);
int mdm_silu_bwd(const float* x, const float* dy, float* dx, int acc, int64_t n, void* stream);

/* The linear layers of the time-embedding path (unet6.py:395-399 `embed`, unet6.py:350 the per-block projections, all in one
 * [sum Cout][temb] matrix) and their data gradients: fp32 contractions whose row count M is the batch.  M <= 128,
 * N % 16 == 0, K % (64 * splits) == 0 (mdm_skinny_supported); larger batches go through mdm_gemm.
 *   fwd: y[M][N] = x[M][K] W[N][K]^T + bias; act_out (may be NULL) = silu(y).  With x == NULL the input is the sinusoidal
 *        embedding of t[M] (dimension K, as mdm_timestep_embedding2), also stored to emb_out[M][K] when that is not NULL.
 *   bwd: dx[M][N] = dy[M][K] W[K][N] (* silu'(pre[M][N]) when pre != NULL); with splits > 1 the reduction is cut into
 *        `splits` ranges whose fp32 partial results go to slabs[splits][M][N] (pre and dx unused) for mdm_silu_bwd_sum.
 *   mdm_silu_bwd_sum: dx[n] = (sum_s slabs[s][n]) * silu'(pre[n]) (pre may be NULL: plain sum). */
int mdm_skinny_supported(int M, int N, int K, int splits);
int mdm_skinny_linear_fwd(const float* x, int ldx, const float* t, int flip_sin_to_cos, float freq_shift, float* emb_out,
                          const float* W, int ldw, const float* bias, int M, int N, int K, float* y, int ldy, float* act_out,
                          void* stream);
int mdm_skinny_linear_bwd(const float* dy, int lddy, const float* W, int ldw, int M, int N, int K, int splits, const float* pre,
                          float* dx, float* slabs, void* stream);
int mdm_silu_bwd_sum(const float* pre, const float* slabs, int nslab, int64_t n, float* dx, void* stream);

/* column sums of dY[N][P][C] (NHWC): per_img[n*ld + c] (= or +=) sum_p, and dbias[c] += sum_{n,p}.
 * Either output may be NULL.  Backward of the bias add (unet6.py:233) and of the
 * time-embedding broadcast add (unet6.py:359). */
int mdm_colsum(int dtype, const void* dY, int N, int P, int C, float* per_img, int ld, int acc_img,
               float* dbias, void* stream);

/* 2x2 sum-pool of g[N][2H][2W][C] into dst[N][H][W][C] (backward of nn.Upsample(2,'nearest'), unet6.py:472) */
int mdm_sumpool2(int dtype, const void* g, void* dst, int acc, int N, int H, int W, int C, void* stream);

/* dst += src over n elements of `dtype` (n % 8 == 0): joins two gradient contributions of one activation */
int mdm_add(int dtype, void* dst, const void* src, int64_t n, void* stream);
/* dst = x + y (y == NULL: dst = x) */
int mdm_add3(int dtype, void* dst, const void* x, const void* y, int64_t n, void* stream);

/* layout converters: NCHW fp32 <-> NHWC dtype with the channel count padded to Cp (pad = 0) */
int mdm_nchw_to_nhwc(int dtype, const float* x, void* y, int N, int C, int H, int W, int Cp, void* stream);
int mdm_nhwc_to_nchw(int dtype, const void* x, float* y, int N, int C, int H, int W, int Cp, void* stream);

/* ------------------------------------------------------------------------- *
 * Scheduler kernels (NCHW fp32, like the reference tensors)
 * ------------------------------------------------------------------------- */
/* On-device draw of the per-sample step (trainer_masked_mean_shift.py:109-112,
 * scheduler.py:88-100, 780-794): idx ~ U{0..n_used-1}; t = used[idx]; amount = table[t-1];
 * weight = wtab[idx] (or 1); out2 = table2[t-1] for a second schedule table (the shift ratio; both may be NULL);
 * zero_out[0] = zero_out[1] = 0 (may be NULL: the two-word loss accumulator mdm_loss_fwd_bwd adds to -- this is the first
 * launch of a step).
 * Philox4x32-10 keyed by rng[0]=seed, rng[1]=offset (device memory). */
int mdm_draw_timesteps(const uint64_t* rng, const int32_t* used, int n_used, const double* table,
                       const float* wtab, int N, float* t_out, double* amount_out, float* weight_out,
                       int32_t* idx_out, const double* table2, double* out2, int64_t* zero_out, void* stream);

/* Thresholding mask + fill + degrade (scheduler.py:286-323 == 438-477 == 572-598).
 *   u: [N][Cm*HW] uniforms (replay mode) or NULL (device Philox, stream id `rng_stream`);
 *   mask_in: existing [N][C][HW] keep-mask to use instead of drawing (degrade_with_mask), or NULL;
 *   amount: [N] double ratios (thresholding) ; Cm = 1 ('1-channel') or C ('3-channel');
 *   fill_mode: 0 constant `fill_const`, 1 degraded_area image-wise, 2 degraded_area channel-wise,
 *              3 non_degraded_area (NaN -> 0);
 * outputs (any may be NULL): x_t, mask [N][C][HW] fp32, mean_pixel [N][C]. */
int mdm_degrade(const float* x0, const float* u, const float* mask_in, const double* amount, int amount_stride,
                const uint64_t* rng, int rng_stream, int N, int C, int HW, int Cm, int fill_mode, float fill_const,
                float* x_t, float* mask, float* mean_pixel, void* stream);

/* 'indexing' mode on device (scheduler.py:279-284): keep-mask [N][C][HW] with EXACTLY count[n] zeroed pixels
 * per image (the count smallest of HW Philox keys; same distribution as randperm(HW)[:count], different
 * stream).  The replay mode ships the host's randperm-built mask through mdm_degrade(mask_in=...). */
int mdm_index_mask(const double* count, int count_stride, const uint64_t* rng, int rng_stream, int N, int C, int HW,
                   float* mask, void* stream);

/* Shift (scheduler.py:612-732) + perturb_shift (:757-766):  s = z * ratio,  x_in = x_t + s.
 *   z: [N][zc][zhw] draws (replay) or NULL (device Philox normal(mean,1) / uniform(-1,1) per `kind`);
 *   kind: 0 non_shift, 1 '1-d_constant' (zc=1,zhw=1,uniform), 2 '3-d_constant' (zc=3,zhw=1,uniform),
 *         3 'noise_reduction' (zc=1,zhw=HW,normal), 4 'noise_with_perturbation' (zc=3,zhw=HW,normal),
 *         5 'noise_std_reduction' (zc=3,zhw=HW; s = N(noise_mean, ratio_n); a replayed z IS the shift);
 *   ratio: [N] double; per_column != 0 reproduces the reference's N==W broadcast (ratio indexed by column w);
 *   outputs: s [N][C][HW] (may be NULL), x_in fp32 NCHW (may be NULL), x_in_nhwc (dtype, Cp channels; may be NULL: only the
 *   C real channels of every pixel are written -- the caller provides the pad channels zeroed once, they never change). */
int mdm_shift(const float* x_t, const float* z, const double* ratio, const uint64_t* rng, int rng_stream,
              int kind, float noise_mean, int per_column, int N, int C, int H, int W,
              float* s, float* x_in, int dtype, void* x_in_nhwc, int Cp, void* stream);
/* x_nhwc[pix][C .. Cp) = 0 for npix pixels: what a caller that did not get its buffer from a zeroing allocator runs ONCE
 * before the first mdm_shift on it (mdm_shift leaves the pad channels alone; the first convolution multiplies them by zero
 * weights, so garbage there -- a NaN -- would poison its output and its weight gradient). */
int mdm_zero_pad_channels(int dtype, void* x_nhwc, int64_t npix, int C, int Cp, void* stream);

/* Fused x0-space loss + its gradient (trainer_masked_mean_shift.py:142-159, trainer_masked.py:126-140):
 *   r = (x_in + pred) - s - x0 ;  loss += sum(w_n * r^2) / numel ;  dpred = 2 w_n r / numel * gscale
 * pred, dpred: NHWC dtype with Cp channels (pad channels of dpred are written 0). s, w may be NULL.
 * loss_q40: TWO int64 words the caller zeroes before a step (mdm_draw_timesteps does).  [0] += the loss in Q23.40 fixed
 * point (loss = (double)loss_q40[0] * 2^-40): the workgroups' partial sums meet through INTEGER atomics, which commute, so
 * the value is bit-identical from run to run (a float atomic sums in arrival order).  [1] counts partials that were not
 * finite or >= 2^22: non-zero means the loss is not representable (report NaN). */
int mdm_loss_fwd_bwd(int dtype, const void* pred, const float* x_in, const float* s, const float* x0,
                     const float* w, int N, int C, int H, int W, int Cp, float gscale,
                     void* dpred, int64_t* loss_q40, void* stream);

/* Reverse-step pieces (sampler.py:146-152, 199-216):
 *   x0_hat = (x_in + pred) - s                                              (mdm_sampler_x0)
 *   x_t   <- x_t + (d_next - d_t)  [momentum]  |  x_t <- d_next  [base]     (mdm_sampler_update) */
int mdm_sampler_x0(int dtype, const void* pred_nhwc, int Cp, const float* x_in, const float* s,
                   int N, int C, int H, int W, float* pred_nchw, float* shifted0, float* x0_hat, void* stream);
/* rng = {seed, offset} (uint64 x2, device): ++offset in stream order.  Every step / graph replay starts with it (the
 * kernels key Philox by (seed, offset, stream id, element)). */
int mdm_rng_advance(uint64_t* rng, void* stream);
/* Device-side per-step parameters (so that one reverse step is one hipGraph replayed T times): with step = *step_ctr,
 * i = T-1-step, t = timesteps[i], t_next = t-1 (t when i == 0) writes time_out[n] = t, ratio_out[n] = ratio_tab[t-1]
 * (optional), amt_t[n] = amount_tab[t-1], amt_next[n] = amount_tab[t_next-1], then ++*step_ctr and ++rng[1]
 * (sampler.py:137-170; the Philox offset bump of the host path). */
int mdm_sampler_step_params(const int32_t* timesteps, int T, int32_t* step_ctr, const double* ratio_tab,
                            const double* amount_tab, int n, float* time_out, double* ratio_out, double* amt_t,
                            double* amt_next, uint64_t* rng, void* stream);
int mdm_sampler_update(const float* d_t, const float* d_next, float* x_t, float* diff, int momentum,
                       int64_t n, void* stream);

/* ------------------------------------------------------------------------- *
 * Evaluation caller (the step after the sampler; tester.py:57-223, sampler.py:487-526): the similarity matrix of
 * generated images against data images is ONE fp32 contraction (mdm_gemm, layout 0) over unit-length rows.
 *   mdm_normalize01  y = (x - min) / (max - min) per image [N][E], NaN -> 0       (utils/datautils.py:211-222)
 *   mdm_unit_rows    y[r] = x[r] / max(||x[r]||, eps) over [R][D]                  (F.cosine_similarity's normalisation)
 *   mdm_col_argmax   per column of S[M][B]: largest value and the first row holding it (score.max(dim=0), tester.py:198)
 * ------------------------------------------------------------------------- */
int mdm_normalize01(const float* x, float* y, int N, int E, void* stream);
int mdm_unit_rows(const float* x, float* y, int R, int D, float eps, void* stream);
int mdm_col_argmax(const float* S, int M, int B, float* val, int64_t* idx, void* stream);

/* ------------------------------------------------------------------------- *
 * Optimizer: global-norm clip + AdamW + EMA + bf16 weight shadow in one pass over
 * flat fp32 buffers (trainer_masked_mean_shift.py:163-172, main_train_masked.py:134-141).
 *   hp (device, 8 floats): lr, beta1, beta2, eps, weight_decay, bias_corr1, bias_corr2, ema_decay
 *   mdm_sqnorm stores sum(g^2) to *out (two-stage, fixed summation order: bit-identical on every rank).
 *   clip_coef = min(1, max_norm / (sqrt(*sqnorm) * gscale_inv + 1e-6)); g is multiplied by gmul first
 *   (gmul = 1/world for the DP mean).  ema / shadow may be NULL.
 * ------------------------------------------------------------------------- */
int mdm_sqnorm(const float* g, int64_t n, float* out, void* stream);
int mdm_adamw_ema(float* p, const float* g, float* m, float* v, float* ema, void* shadow_bf16,
                  int64_t n, const float* hp, const float* sqnorm, float max_norm, float gmul, void* stream);
int mdm_cast_bf16(const float* src, void* dst, int64_t n, void* stream);
/* Transposed bf16 shadow of the conv filters for the data-gradient pass, from the bf16 shadow Pb the optimizer kernel writes:
 * for every 64x64 tile listed in `tiles` (device, int64 x5 per tile: element offset of one tap's [Cout][Cin] matrix, Cout,
 * Cin, row0, col0; Cout % 8 == Cin % 8 == 0) write PT[off + c*Cout + r] = Pb[off + r*Cin + c]. */
int mdm_transpose_shadow_bf16(const void* Pb, void* PT, const int64_t* tiles, int ntiles, void* stream);
/* The B_split shadow of fp32 filters (mdm_gemm_desc.B_split): for each of the `nseg` segments (off, len) of the device table `segs`
 * (element offsets into P / Ps; off % 4 == 0, len % 32 == 0; a segment is a run of filter rows whose length is a multiple of 32),
 * every 32-element block b of the segment becomes 128 bytes at the same offset of Ps: 16-byte chunk g (g = 0..3) = the bf16 hi halves
 * of elements {4g..4g+3, 16+4g..16+4g+3} of the block -- 32-bit word k of the chunk = (element 4g+k in the low half, element 16+4g+k in
 * the high half) --, chunk 4+g = their bf16 lo halves in the same order (hi = bf16(x), lo = bf16(x - hi), both round-to-nearest-even):
 * the arrangement the split kernels make of their input rows in LDS.  Replaces nothing upstream: it is
 * how the fp32 `F.conv2d` of unet6.py:232-235 reaches the bf16 MFMA pipe at ~fp32 accuracy. */
int mdm_split_shadow(const float* P, float* Ps, const int64_t* segs, int nseg, void* stream);
int mdm_fill_f32(float* p, float v, int64_t n, void* stream);
/* base[off .. off + len) = v for nseg segments segs[i] = {off, len} (device, int64 pairs; len % 4 == 0, len <= 4096, off % 4 == 0):
 * the per-step zeroing of the ACCUMULATED gradient slots only (biases, GroupNorm scales: `optimizer.zero_grad()` at
 * trainer_masked_shift.py:137 -- the weight gradients overwrite their slots, mdm_wgrad_group_*) */
int mdm_fill_segments_f32(float* base, const int64_t* segs, int nseg, float v, void* stream);

/* ------------------------------------------------------------------------- *
 * hipGraph capture of a launch sequence issued on `stream` (the whole train step
 * or one reverse step is ~10^3 short kernels: replay removes the host launch cost).
 * ------------------------------------------------------------------------- */
int mdm_graph_begin(void* stream);
int mdm_graph_end(void* stream, void** graph_exec_out);
int mdm_graph_launch(void* graph_exec, void* stream);
int mdm_graph_destroy(void* graph_exec);

/* HIP events on a caller-supplied stream (bench.py times kernels on the launch stream) */
int mdm_event_create(void** ev);
int mdm_event_record(void* ev, void* stream);
int mdm_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_out);   /* synchronises on ev_stop */
int mdm_event_destroy(void* ev);
int mdm_stream_sync(void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MDM_HIP_H */
