/*
 * wm2f.h -- C ABI of libwm2f.so: hand-written HIP (gfx950 / MI355X) kernels for the
 * Mask2Former hot path of marco-conciatori-public/weed_instance_segmentation.
 *
 * The reference has no FFI for this path: it calls the Python package `transformers`
 * (train.py:196, metrics.py:56, inference.py:27).  Each entry point below therefore
 * cites the Python function of that dependency whose arithmetic it replaces
 * ("HF:n" = transformers/models/mask2former/modeling_mask2former.py line n,
 *  "TORCHF:n" = torch/nn/functional.py line n, transformers 5.15.0 / torch 2.10).
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer into caller-owned memory unless marked "host";
 *     tensors are contiguous, row-major, in the shape written next to them;
 *   - nothing is allocated, freed or retained; there is no global mutable state (host or device) and no
 *     environment variable is read: behaviour is a function of the arguments alone;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), no hidden syncs;
 *   - return 0 on success, a negative WM2F_E* code otherwise; wm2f_last_error() gives the
 *     thread-local message of the last failing call on this thread;
 *   - dtype: WM2F_F32 = 0 (fp32 storage + fp32 arithmetic); WM2F_BF16 = 1 where an entry point says so (bf16 storage of the
 *     named tensors, fp32 arithmetic).
 */
#ifndef WM2F_H
#define WM2F_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WM2F_VERSION 100 /* 0.1.0 */

#define WM2F_F32 0
#define WM2F_BF16 1

#define WM2F_OK 0
#define WM2F_EINVAL (-1)      /* bad argument / unsupported shape */
#define WM2F_EUNSUPPORTED (-2) /* dtype or configuration not built */
#define WM2F_ELAUNCH (-3)     /* hipLaunch / runtime error */

#define WM2F_MAX_LEVELS 8

int wm2f_version(void);
const char* wm2f_last_error(void);

/* ---- K1: multi-scale deformable attention core ------------------------------------------
 * Replaces multi_scale_deformable_attention(), HF:798-837.
 *   value   (B, S, heads, D)        S = sum_l H_l*W_l, levels stored back to back
 *   loc     (B, Q, heads, L, P, 2)  normalised (x, y); pixel = loc * (W_l, H_l) - 0.5
 *   attn_w  (B, Q, heads, L, P)
 *   out     (B, Q, heads*D)
 *   level_hw  host int32 [L][2] = (H_l, W_l)
 * Bilinear, zero padding, align_corners = False.  D in {8, 16, 32, 64}.
 */
int wm2f_msdeform_fwd(const void* value, const void* loc, const void* attn_w, void* out,
                      const int32_t* level_hw, int B, int S, int Q, int heads, int D, int L, int P,
                      int dtype, void* stream);

/* Backward of the above.  grad_value (B,S,heads,D) must be ZEROED by the caller (it is
 * accumulated with float atomics); grad_loc, grad_attn_w are overwritten. */
int wm2f_msdeform_bwd(const void* value, const void* loc, const void* attn_w, const void* grad_out,
                      void* grad_value, void* grad_loc, void* grad_attn_w,
                      const int32_t* level_hw, int B, int S, int Q, int heads, int D, int L, int P,
                      int dtype, void* stream);

/* K1 backward with run-to-run identical results (same arguments and outputs as wm2f_msdeform_bwd plus a workspace of
 * wm2f_msdeform_bwd_det_workspace(level_hw, B, S, heads, D, L) bytes; 0 = not built for these levels).  No float atomics:
 * every tile stores its LDS window into its own slab of the workspace and a second kernel sums, per grad_value element,
 * the windows covering it in a fixed tile order; the rare points outside a window are added as integers into an int64
 * fixed-point image (unit 2^-44 of the largest |grad_out| of the image and head; overflow-free for S < 2^17).
 * grad_loc / grad_attn_w never had a scatter.  `grad_value` need not be cleared by the caller.
 * Built for the LDS-window backward only (head_dim 32, 4 points, Q == S, levels that fit its windows); otherwise
 * WM2F_EUNSUPPORTED. */
int64_t wm2f_msdeform_bwd_det_workspace(const int32_t* level_hw, int B, int S, int heads, int D, int L);
int wm2f_msdeform_bwd_det(const void* value, const void* loc, const void* attn_w, const void* grad_out,
                          void* grad_value, void* grad_loc, void* grad_attn_w, void* workspace,
                          const int32_t* level_hw, int B, int S, int Q, int heads, int D, int L, int P, int dtype,
                          void* stream);

/* K1 for TRAINING on the merged projection's rows (replaces, as one differentiable op of (value, rows), the prologue of
 * transformers modeling_mask2former.py:983-1002 -- offsets / (W, H) + reference points, softmax over the L * P logits -- the
 * core :798-837 and the autograd of both):
 *   value      (B, S, heads, 32)       fp32 (the arithmetic is fp32, as the dependency's grid_sample is under autocast)
 *   rows       (B, Q, heads * L*P*3)   [offsets (heads, L, P, 2) in pixels | logits (heads, L*P)] per token, as
 *                                      cat(sampling_offsets, attention_weights) writes them
 *   out        (B, Q, heads * 32)
 *   grad_out   as out;  grad_rows as rows (every element written);  grad_value as value (need not be cleared)
 * dtype = WM2F_F32: rows, out, grad_out, grad_rows are fp32; WM2F_BF16: all four are bf16 (what a bf16-autocast Linear writes
 * and reads; bf16 -> fp32 is exact, the outputs are rounded to nearest even once).  Reference points are the tokens' pixel
 * centres (:1127-1156 with valid ratios of 1).  Forward: the streaming kernel (3 levels 1 : 2 : 4 coarse first or a pyramid
 * it takes, P = 4, Q == S); backward: the LDS-window kernels; heads even.  Otherwise WM2F_EUNSUPPORTED (compose the op from
 * wm2f_msdeform_fwd / _bwd). */
int wm2f_msdeform_rows_fwd(const void* value, const void* rows, void* out, const int32_t* level_hw, int B, int S, int Q,
                           int heads, int D, int L, int P, int dtype, void* stream);
int wm2f_msdeform_rows_bwd(const void* value, const void* rows, const void* grad_out, void* grad_value, void* grad_rows,
                           const int32_t* level_hw, int B, int S, int Q, int heads, int D, int L, int P, int dtype,
                           void* stream);

/* K1 with the module prologue fused (HF:983-1002): softmax over the L*P logits and
 * loc = ref + offset / (W_l, H_l) are computed in-kernel.
 *   offsets (B, Q, heads, L, P, 2)  raw sampling_offsets output
 *   logits  (B, Q, heads, L*P)      raw attention_weights output (pre-softmax)
 *   ref     (Q, L, 2)               reference points, shared by the batch (valid ratios are 1)
 */
int wm2f_msdeform_fused_fwd(const void* value, const void* offsets, const void* logits, const void* ref,
                            void* out, const int32_t* level_hw, int B, int S, int Q, int heads, int D,
                            int L, int P, int dtype, void* stream);

/* Fused variant reading the two projections from ONE packed row per token, as a single merged
 * Linear writes them:  packed (B, Q, heads*L*P*3) = [offsets (heads,L,P,2) | logits (heads,L*P)].
 * LDS-window kernel only (D = 32, P = 4, Q == S, L <= 4); returns WM2F_EUNSUPPORTED otherwise. */
int wm2f_msdeform_fused_packed_fwd(const void* value, const void* packed, void* out,
                                   const int32_t* level_hw, int B, int S, int Q, int heads, int D, int L,
                                   int P, int dtype, int margin, void* stream);

/* The same for rows in the kernel's RECORD order -- a free choice of the row order of the merged Linear that writes them
 * (ops.k1_lane_order), made for the kernel's loads.  Per token and head 36 floats = 144 bytes, in 16-byte pieces; lane j of
 * a query's quad owns sampling-point slot j of every level:
 *   rec[ 4 j + {0,1,2,3}]      = offsets[.., h, level 0, j, {x,y}], offsets[.., h, level 1, j, {x,y}]
 *   rec[16 + 4 j + {0,1,2,3}]  = offsets[.., h, level 2, j, {x,y}], logits[.., h, 0*4 + j], logits[.., h, 1*4 + j]
 *   rec[32 + j]                = logits[.., h, 2*4 + j]
 * A lane fetches two aligned 16-byte pieces and one dword per query (3 loads inside one 144-byte record) instead of 6 loads
 * scattered over the token's 1152-byte row.  Streaming kernel only (D = 32, P = 4, L = 3, Q == S, levels coarse first with
 * sides 1:2:4); any other shape returns WM2F_EUNSUPPORTED (use the [offsets | logits] form).  `head_major` is a bit set:
 *   bit 0  lanes is (heads, B, Q, 36) instead of (B, Q, heads, 36): a head's records of consecutive tokens are contiguous
 *          (what wm2f_token_linear_fwd with out_group = 36 writes);
 *   bit 1  `value` is stored (heads, B, S, 32) instead of (B, S, heads, 32) (out_group = 32);
 *   bit 2  SLAB order: the tiles are walked heads-outermost, so that the workgroups of an XCD share ONE (image, head) slab
 *          of `value` (2.75 MB at config 2) in its 4 MiB L2 and the window halos (3.6 x the slab at L2 level) come from HBM
 *          once.  Meant for bit 0 = 1 (with token-major records every 1152-byte row would be fetched once per head).
 *          Measured in the model: HBM traffic 1.29 x -> 1.005 x the algorithmic bytes, 156 -> 136 us.
 * The result does not depend on any of the bits (tests: bit-identical). */
int wm2f_msdeform_fused_lanes_fwd(const void* value, const void* lanes, void* out, const int32_t* level_hw, int B, int S,
                                  int Q, int heads, int D, int L, int P, int dtype, int head_major, void* stream);

/* Same two operations with the kernel choice exposed, so that the two fall-back kernels can be held to the golden vectors
 * on shapes `auto` gives to the first:
 *   fused   0: a = loc, b = attn_w, ref unused      1: a = offsets, b = logits, ref as above
 *   variant 0: auto (tries 4, 2, 1 in that order)
 *           1: direct gather (any D)
 *           2: LDS-window kernel (D = 32, P = 4, Q == S, L <= 4)
 *           4: streaming quad kernel (3 levels coarse first whose sides about double, P = 4, D = 32, Q == S; persistent
 *              workgroups + loader waves), one workgroup per CU
 *   margin  window margin in pixels for the LDS-window kernel; sampling points farther than that
 *           from their reference point take a slow path (results never depend on it).
 * Any other variant returns WM2F_EUNSUPPORTED: superseded kernels (phased quads, half-head form), measured negatives
 * (flags instead of barriers, strip order, the round-1 loader schedule), timing ablations and stamped builds live in the
 * separate profiling library (include/wm2f_prof.h), never in libwm2f.so.  wm2f_msdeform_fwd / _fused_fwd are variant 0,
 * margin 4. */
int wm2f_msdeform_fwd_v(const void* value, const void* a, const void* b, const void* ref, void* out,
                        const int32_t* level_hw, int B, int S, int Q, int heads, int D, int L, int P,
                        int dtype, int fused, int variant, int margin, void* stream);

/* ---- K3 in bf16 (BASELINE configs 3-5: bf16 autocast) ---------------------------------------------
 * Same line as wm2f_mask_einsum_fwd (HF:2046) with bf16 operands, fp32 accumulation on the bf16 matrix cores and
 * fp32 output; HBM-bound (478 MB per call at config 2).
 *   emb (B, Q, C) bf16, Q <= 112 per call;  pix_pixel_major (B, HW, C) bf16 -- the pixel features PIXEL-MAJOR (the
 *   MFMA operands want the contraction index contiguous), made once per forward by wm2f_nchw_to_pixel_major_bf16
 *   from the (B, C, HW) map;  out (B, Q, HW) fp32;  C % 32 == 0, C <= 512. */
int wm2f_mask_einsum_bf16_fwd(const void* emb, const void* pix_pixel_major, void* out, int B, int Q, int C,
                              int HW, void* stream);

/* K3 backward under bf16 autocast (replaces grad.to(bfloat16) + two batched library GEMMs): grad_out is fp32 (the
 * logits are), emb / pix and both gradients bf16 (round-to-nearest-even), accumulation fp32; g_emb summed in a fixed
 * order.  Either output may be NULL.  `pix` is the NCHW tensor (B, C, HW), not the pixel-major copy.
 * `workspace`: wm2f_mask_einsum_bf16_bwd_workspace(B, Q, C, HW) bytes.
 *   C % 64 == 0, Q % 4 == 0, Q <= 112, HW % 8 == 0; other shapes: WM2F_EUNSUPPORTED */
int64_t wm2f_mask_einsum_bf16_bwd_workspace(int B, int Q, int C, int HW);
int wm2f_mask_einsum_bf16_bwd(const void* emb, const void* pix, const void* grad_out, void* g_emb, void* g_pix,
                              void* workspace, int B, int Q, int C, int HW, void* stream);
int wm2f_nchw_to_pixel_major_bf16(const void* src, void* dst, int B, int C, int HW, void* stream);

/* ---- K3: mask einsum --------------------------------------------------------------------
 * Replaces torch.einsum("bqc,bchw->bqhw"), HF:2046.
 *   emb (B, Q, C)   pix (B, C, HW)   out (B, Q, HW)      C % 16 == 0
 */
int wm2f_mask_einsum_fwd(const void* emb, const void* pix, void* out, int B, int Q, int C, int HW,
                         int dtype, void* stream);

/* K3 with the attention-mask epilogue fused (the `out_attn_mask` of SURVEY section 8b): HF:2046 followed by
 * HF:2051-2053 (sigmoid, `< 0.5`) and the row flag of HF:1912-1914, for predictions that only feed the next layer's
 * mask -- the logits are never written.  There is no resize in it: `pix` is the mask-feature map ALREADY resized to
 * the level's resolution (wm2f_resize_bilinear once per forward; resize and einsum commute), so HW = Hn * Wn.
 *   emb (B, Q, C)   pix (B, C, HW)   mask (B, Q, HW) uint8, 1 = blocked   row_open (B, Q) int32 (cleared by the call)
 *   C % 16 == 0, HW % 4 == 0 */
int wm2f_mask_einsum_attn_mask_fwd(const void* emb, const void* pix, void* mask, void* row_open, int B, int Q, int C,
                                   int HW, int dtype, void* stream);

/* K3 backward (HF:2046 under autograd; replaces the two batched library GEMMs autograd derives from the einsum):
 *   g_pix[b][c][p] = sum_q emb[b][q][c] * grad_out[b][q][p]      g_emb[b][q][c] = sum_p grad_out[b][q][p] * pix[b][c][p]
 * Either output may be NULL (not wanted).  g_emb is summed in a fixed order (pixel ranges to a workspace, then in range
 * order): no atomics, run-to-run identical.  `workspace`: wm2f_mask_einsum_bwd_workspace(B, Q, C, HW) bytes.
 *   emb (B, Q, C)   pix (B, C, HW)   grad_out (B, Q, HW)   g_emb (B, Q, C)   g_pix (B, C, HW)   all fp32
 *   C % 64 == 0, Q % 4 == 0, HW % 4 == 0; other shapes: WM2F_EUNSUPPORTED */
int64_t wm2f_mask_einsum_bwd_workspace(int B, int Q, int C, int HW);
int wm2f_mask_einsum_bwd(const void* emb, const void* pix, const void* grad_out, void* g_emb, void* g_pix,
                         void* workspace, int B, int Q, int C, int HW, int dtype, void* stream);

/* ---- attention-mask build ----------------------------------------------------------------
 * Replaces HF:2048-2054 (bilinear resize, sigmoid, < 0.5) WITHOUT the x num_heads replication,
 * and the row fix-up of HF:1912-1914.
 *   logits   (B, Q, H, W) fp32
 *   mask     (B, Q, Hn*Wn) uint8, 1 = blocked
 *   row_open (B, Q) int32, 1 if at least one key of the row is open (0 => attend everywhere)
 */
int wm2f_attn_mask_build(const void* logits, void* mask, void* row_open, int B, int Q, int H, int W,
                         int Hn, int Wn, void* stream);

/* ---- K2: masked cross-attention ------------------------------------------------------------
 * Replaces the attention arithmetic of nn.MultiheadAttention as called at HF:1644-1650
 * (TORCHF:6578-6600): softmax(bias + (q/sqrt(D)) k^T) v, bias = -inf where mask == 1; rows with
 * row_open == 0 ignore the mask (HF:1912-1914).  Projections stay outside.
 *   q (B, Q, heads*D) ALREADY scaled by 1/sqrt(D);  k, v (B, N, heads*D)
 *   mask (B, Q, N) uint8 shared by all heads, or NULL;  row_open (B, Q) int32 or NULL
 *   out (B, Q, heads*D);  lse (B, heads, Q) fp32 log-sum-exp per row (for backward) or NULL
 *   workspace: device scratch of at least wm2f_masked_xattn_workspace(...) bytes
 * D in {16, 32, 64}.
 */
int64_t wm2f_masked_xattn_workspace(int B, int heads, int Q, int N, int D);
int wm2f_masked_xattn_fwd(const void* q, const void* k, const void* v, const void* mask,
                          const void* row_open, void* out, void* lse, void* workspace,
                          int B, int heads, int Q, int N, int D, int dtype, void* stream);

/* The same with bf16 operands, for the bf16-autocast configurations (BASELINE configs 2-4), where the dependency's two
 * attention products are bf16 matrix products around an fp32 softmax (TORCHF:6578-6600 under autocast): q, k, v bf16 as the
 * in_proj Linears emit them (no cast pass, half the K / V bytes), both products on the bf16 matrix cores with fp32
 * accumulation, softmax / (m, l) / O in fp32, P rounded to bf16 for the second product; out (B, Q, heads*D) and lse fp32.
 * Same mask / row_open / workspace (wm2f_masked_xattn_workspace) as the fp32 form.  Full-tile form only: D = 32,
 * N % 16 == 0, 16-byte aligned operands -- anything else returns WM2F_EUNSUPPORTED (cast to fp32, call the form above). */
int wm2f_masked_xattn_bf16_fwd(const void* q, const void* k, const void* v, const void* mask, const void* row_open,
                               void* out, void* lse, void* workspace, int B, int heads, int Q, int N, int D, void* stream);

/* Backward of the above (P is recomputed from q, k and lse).  grad_q is the gradient w.r.t. the
 * PRE-SCALED q; grad_q / grad_k / grad_v are overwritten.
 *   out, lse: the forward's results; grad_out (B, Q, heads*D)
 *   workspace: at least wm2f_masked_xattn_bwd_workspace(...) bytes */
int64_t wm2f_masked_xattn_bwd_workspace(int B, int heads, int Q, int N, int D);
int wm2f_masked_xattn_bwd(const void* q, const void* k, const void* v, const void* mask,
                          const void* row_open, const void* out, const void* lse, const void* grad_out,
                          void* grad_q, void* grad_k, void* grad_v, void* workspace,
                          int B, int heads, int Q, int N, int D, int dtype, void* stream);

/* Backward of wm2f_masked_xattn_bf16_fwd: q, k, v bf16 as saved by the forward, out / lse / grad_out fp32; grad_q fp32,
 * grad_k / grad_v bf16 (the operands' dtype).  The five products per (key tile, query tile) on the bf16 matrix cores, p and dS
 * in fp32 from fp32 accumulators.  Workspace: wm2f_masked_xattn_bwd_workspace.  D = 32, N % 16 == 0, Q <= 112 (one query
 * chunk), 16-byte aligned operands; anything else returns WM2F_EUNSUPPORTED (cast to fp32, wm2f_masked_xattn_bwd). */
int wm2f_masked_xattn_bf16_bwd(const void* q, const void* k, const void* v, const void* mask, const void* row_open,
                               const void* out, const void* lse, const void* grad_out, void* grad_q, void* grad_k,
                               void* grad_v, void* workspace, int B, int heads, int Q, int N, int D, void* stream);

/* ---- K4: Hungarian-matcher cost matrices ------------------------------------------------------
 * Replaces Mask2FormerHungarianMatcher.forward up to (not including) the scipy solver,
 * HF:444-472 with sample_point HF:245-274 and the pair-wise losses HF:328-374, batched over
 * NL prediction levels and B images so that ONE device->host copy feeds every solver call.
 *   mask_logits  (NL, B, Q, h, w) fp32, NL <= 16   class_logits (NL, B, Q, C1) fp32
 *   tgt_masks    concatenation over images of (T_i, Ht, Wt); tgt_dtype 0 = fp32, 1 = uint8
 *   tgt_offset   host int32 [B+1]: image i owns targets [tgt_offset[i], tgt_offset[i+1])
 *   tgt_classes  int64 [sum T_i]
 *   points       (NL, B, P, 2) fp32 in [0,1] as (x, y)
 *   cost         (NL, B, Q, Tmax) fp32; columns >= T_i are left untouched
 *   workspace    scratch, at least wm2f_matcher_workspace(...) bytes (sampled targets, the grouped points, band offsets)
 * cost = w_mask*BCE_pair + w_class*(-softmax(class)[:, tgt]) + w_dice*dice_pair, clamped to
 * +-1e10 with NaN -> 0.
 * The points may come in any order (the cost is a sum over them): with P >= 1024 and maps up to ~2000 wide the call groups
 * each (level, image)'s points by band of map rows (stable, so results are reproducible) and samples the predictions from LDS
 * bands instead of gathering from memory.
 */
int64_t wm2f_matcher_workspace(int NL, int B, int Q, int P, int Tsum);
int wm2f_matcher_cost(const void* mask_logits, const void* class_logits, const void* tgt_masks,
                      int tgt_dtype, const int32_t* tgt_offset, const void* tgt_classes,
                      const void* points, void* cost, void* workspace, int NL, int B, int Q, int C1,
                      int h, int w, int Ht, int Wt, int P, int Tmax, float w_class, float w_mask,
                      float w_dice, void* stream);
/* The same with the levels NOT stacked: mask_levels = HOST array of NL (<= 16) DEVICE pointers, each (B, Q, h, w) fp32
 * (the mask predictor's outputs where they are; a stacked copy is 10 x 210 MB at config 2). */
int wm2f_matcher_cost_levels(const void* const* mask_levels, const void* class_logits, const void* tgt_masks,
                             int tgt_dtype, const int32_t* tgt_offset, const void* tgt_classes,
                             const void* points, void* cost, void* workspace, int NL, int B, int Q, int C1, int h,
                             int w, int Ht, int Wt, int P, int Tmax, float w_class, float w_mask, float w_dice,
                             void* stream);

/* ---- linear sum assignment on the device (HF:474: scipy.optimize.linear_sum_assignment on a host copy of each cost matrix) ----
 *   cost    (problems, Q, Tmax) fp32 -- wm2f_matcher_cost's output, problems = levels x B, image = problem % B
 *   counts  (B) int32, DEVICE: targets of each image (its valid columns)
 *   rows / cols (problems, Tcap) int32: the min(Q, T) matched (query, target) pairs of each problem, sorted by query -- exactly
 *           what scipy returns for cost[:, :T] (entries beyond min(Q, T) are not written; Tcap >= max over images of min(Q, T_b)).
 * scipy's own shortest-augmenting-path algorithm with its arithmetic (float64), scan order, tie rule and output order, one wave
 * per problem: bit-identical indices, ties included (tests).  Sides up to 1024; larger returns WM2F_EUNSUPPORTED. */
int wm2f_lsa_batched(const void* cost, const void* counts, void* rows, void* cols, int problems, int B, int Q, int Tmax, int Tcap,
                     void* stream);

/* ---- point sampling (shared by the loss, HF:245-274) ----------------------------------------
 *   feat (N, H, W) fp32 or uint8 (feat_dtype 0 / 1); pts (M, P, 2); out (M, P) fp32.
 *   map_index: int32 [M] -- row m samples feat[map_index[m]] (matched prediction / target maps are
 *   sampled in place instead of being gathered into a copy); NULL means M == N, identity.
 * Backward (fp32 feat only): grad_feat (N,H,W) must be zeroed by the caller (float atomics). */
int wm2f_point_sample_fwd(const void* feat, int feat_dtype, const void* pts, const void* map_index,
                          void* out, int M, int H, int W, int P, void* stream);
int wm2f_point_sample_bwd(const void* grad_out, const void* pts, const void* map_index, void* grad_feat,
                          int M, int H, int W, int P, void* stream);

/* ---- fused HBM-bound passes around the stock GEMMs / convolutions (inference) -------------------
 * wm2f_bias_act:      y = act(x + bias[c] (+ residual)), x / y / residual (N, C, H*W) fp32, H*W % 4 == 0;
 *                     y may alias x.  relu = 1 applies max(., 0).
 * wm2f_add_layernorm: out = LayerNorm(x + residual) * gamma + beta over rows of C = 256 (HF:1076-1078,
 *                     :1086-1088); residual may be NULL; if out_plus_pos != NULL it receives
 *                     out + pos[row % pos_rows] (the next layer's `hidden + pos`, HF:972). */
/* out[b][i] = a[b][i] + p[i], b < B, i < n (fp32, n % 4 == 0, 16-byte aligned): a level's positional embedding added to its tokens
 * for every image of the batch -- the keys of the masked cross-attention, `with_pos_embed` at HF:1644-1650 (inference). */
int wm2f_add_broadcast(const void* a, const void* p, void* out, int B, int64_t n, void* stream);
int wm2f_bias_act(const void* x, const void* bias, const void* residual, void* y, int N, int C, int HW,
                  int relu, void* stream);
int wm2f_add_layernorm(const void* x, const void* residual, const void* gamma, const void* beta,
                       const void* pos, void* out, void* out_plus_pos, int64_t rows, int C,
                       int64_t pos_rows, float eps, void* stream);
/* wm2f_add_layernorm_train_fwd / _bwd: the same residual add + LayerNorm for the TRAIN step (HF:1076-1078, :1086-1088, and the
 *                     next layer's `hidden + pos`, HF:972), C = 256, with the tensors its consumers read written in the same pass:
 *                       y (rows, 256) fp32 = LayerNorm(x + residual) * gamma + beta;   x fp32 or bf16 (x_dtype), residual fp32 or NULL
 *                       y_bf16      NULL, or y rounded to bf16 (the next Linear's operand under bf16 autocast)
 *                       y_plus_pos  NULL, or y + pos[row % pos_rows] in yp_dtype (fp32 / bf16)
 *                       stats (rows, 2) fp32 = (mean, rstd) for the backward;
 *                       clamp > 0: y is limited to [-clamp, clamp], NaN left as it is -- the overflow guard of HF:1090-1093
 *                       (clamp to finfo.max - 1000 when a value is not finite) without the host synchronisation its `if` costs:
 *                       on finite values the clamp is the identity, so applying it always is the same function.
 *                     Backward: the gradients of the three outputs (any of them NULL) are summed in registers;
 *                       grad_sum (rows, 256) fp32 = d loss / d (x + residual) (NULL: not written), grad_x the same in x's dtype
 *                       (NULL: not written), grad_gamma / grad_beta (256) fp32 -- per-workgroup partial sums over fixed row
 *                       ranges in `workspace` (wm2f_add_layernorm_train_workspace(rows) bytes), added in workgroup order:
 *                       deterministic.  Replaces torch's two LayerNorm-backward kernels, the gradient-accumulation adds and the
 *                       casts around them; every operand byte moves once (HBM-bound). */
int64_t wm2f_add_layernorm_train_workspace(int64_t rows);
int wm2f_add_layernorm_train_fwd(const void* x, int x_dtype, const void* residual, const void* gamma, const void* beta,
                                 const void* pos, void* y, void* y_bf16, void* y_plus_pos, int yp_dtype, void* stats,
                                 int64_t rows, int C, int64_t pos_rows, float eps, float clamp, void* stream);
int wm2f_add_layernorm_train_bwd(const void* x, int x_dtype, const void* residual, const void* gamma, const void* stats,
                                 const void* grad_y, const void* grad_y_bf16, const void* grad_y_plus_pos, int gyp_dtype,
                                 void* grad_sum, void* grad_x, void* grad_gamma, void* grad_beta, void* workspace,
                                 int64_t rows, int C, void* stream);
/* wm2f_token_linear_fwd: out (M, N) = epilogue(x (M, K) . W (N, K)^T + bias[N]) on the fp32 matrix cores, for the narrow Linears
 *                        of the pixel decoder's encoder layers (HF:978 value_proj, :983-991 sampling_offsets | attention_weights
 *                        merged, :1012 output_proj, :1086 fc2).  N = 256 or 288, K % 64 == 0, all fp32 row-major (W as
 *                        nn.Linear stores it).  Epilogue, in this order:  relu != 0: max(., 0);  ln_gamma / ln_beta != NULL:
 *                        LayerNorm over the N features of (value + residual[M, N]) (residual may be NULL) -- HF:1076-1078,
 *                        :1086-1088;  out_plus_pos != NULL: additionally out + pos[row % pos_rows] (the next layer's
 *                        `hidden + pos`, HF:972).  x / out below 2 GiB each.
 *                        out_group = G > 0 (G % 4 == 0, N % G == 0, no LayerNorm): out is written (N / G, M, G), feature-group
 *                        major, instead of (M, N): with G = 36 the merged projection writes K1's operand rows head-major
 *                        (wm2f_msdeform_fused_lanes_fwd, head_major = 1). */
int wm2f_token_linear_fwd(const void* x, const void* w, const void* bias, const void* residual, const void* ln_gamma,
                          const void* ln_beta, const void* pos, void* out, void* out_plus_pos, int64_t M, int K, int N, int relu,
                          int64_t pos_rows, float eps, int out_group, void* stream);

/* wm2f_token_wgrad_bf16: the weight / bias gradient of such a Linear (the backward autograd derives for nn.Linear; train
 *                        step of HF:1036-1103 under bf16 autocast):  dw (N, K) fp32 = dy (M, N)^T . x (M, K),
 *                        db (N) fp32 = column sums of dy (NULL: skipped); dy, x bf16 row-major, fp32 accumulation on the bf16
 *                        matrix cores.  Split over the M = batch x tokens contraction with one fp32 partial tile per
 *                        workgroup in `workspace` (wm2f_token_wgrad_workspace(M, N, K) bytes), added in split order by a
 *                        second kernel: deterministic, no atomics.  HBM-bound (each operand read once).  N % 8 == 0,
 *                        K % 8 == 0, operands below 2 GiB, pointers 16-byte aligned. */
int64_t wm2f_token_wgrad_workspace(int64_t M, int N, int K);
int wm2f_token_wgrad_bf16(const void* dy, const void* x, void* dw, void* db, void* workspace, int64_t M, int N, int K,
                          void* stream);
/* The same with fp32 operands (the fp32 train step) on the fp32 matrix cores: exact fp32 products, MFMA-bound
 * (2 M N K flop at 157 TFLOP/s).  N % 4 == 0, K % 4 == 0; same workspace function. */
int wm2f_token_wgrad_f32(const void* dy, const void* x, void* dw, void* db, void* workspace, int64_t M, int N, int K,
                         void* stream);
/* wm2f_tokens_to_nchw: out (B, C, HW) = tokens (B, S, C) rows [start, start + HW) transposed per image -- the
 *                      `hidden[:, start:start+hw].transpose(1, 2).reshape(B, C, h, w)` of HF:1384-1391. */
int wm2f_tokens_to_nchw(const void* tokens, void* out, int B, int S, int C, int start, int HW, void* stream);
/* wm2f_group_norm_tokens: tokens[b][start + p][c] = GroupNorm_G(x + bias)[b][c][p] * gamma + beta for x (B, C, HW) fp32,
 *                         bias (C) or NULL, tokens (B, S, C) -- one level's input projection of the pixel decoder
 *                         (HF:1341-1357: Conv2d 1x1 [+ bias] + GroupNorm, flatten(2).transpose(1, 2), cat over levels)
 *                         written into its rows of the token buffer.  stats_ws: 2 * B * G doubles of scratch. */
int wm2f_group_norm_tokens(const void* x, const void* bias, const void* gamma, const void* beta, void* tokens,
                           void* stats_ws, int B, int C, int G, int HW, int S, int start, float eps, void* stream);
/* wm2f_resize_bilinear: y (NC, Ho, Wo) = torch.nn.functional.interpolate(x (NC, H, W), size=(Ho, Wo), mode="bilinear",
 *                       align_corners=False) in fp32, Wo % 4 == 0 -- the resize of HF:2048-2050, applied to the mask
 *                       FEATURES once per level instead of to every layer's logits (resize and einsum commute). */
int wm2f_resize_bilinear(const void* x, void* y, int NC, int H, int W, int Ho, int Wo, void* stream);
/* wm2f_resize_pyramid: the same resize to (H/2, W/2), (H/4, W/4) and (H/8, W/8) in one pass over x (NC, H, W), H and W
 *                      divisible by 8: y2 (NC, H/2, W/2), y4 (NC, H/4, W/4), y8 (NC, H/8, W/8), each bit for bit what
 *                      wm2f_resize_bilinear gives for that size (at these ratios every output is the mean of a 2 x 2 block). */
int wm2f_resize_pyramid(const void* x, void* y2, void* y4, void* y8, int NC, int H, int W, void* stream);
/* wm2f_bias_relu_maxpool: y (N, C, H/2, W/2) = MaxPool2d(kernel 3, stride 2, padding 1)(ReLU(x + bias[c])), x (N, C, H, W)
 *                         fp32, H even, W % 8 == 0 -- the stem of transformers' ResNet embeddings
 *                         (modeling_resnet.py ResNetEmbeddings: convolution, normalization folded, ReLU, pooler). */
int wm2f_bias_relu_maxpool(const void* x, const void* bias, void* y, int N, int C, int H, int W, void* stream);
/* wm2f_group_norm_act: y = act(GroupNorm_G(x) * gamma + beta (+ bilinear_upsample(up -> H x W))) on (B, C, H, W) fp32,
 *                      W % 4 == 0 -- the GroupNorm tails of the FPN step, HF:1395-1405 (adapter: with `up` =
 *                      the coarser map (B, C, Hs, Ws), align_corners = False, no ReLU; output layer: up = NULL,
 *                      relu = 1).  stats_ws: 2 * B * G doubles of scratch.  y may alias x. */
int wm2f_group_norm_act(const void* x, const void* gamma, const void* beta, const void* up, void* y, void* stats_ws,
                        int B, int C, int G, int H, int W, int Hs, int Ws, float eps, int relu, void* stream);

/* Importance sampling of the mask losses, HF:688-704 (sample_points_using_uncertainty): the points of the k largest scores of
 * each row -- gather(coords, topk(uncertainty, k)[1]) -- by radix selection in LDS instead of the sort a stock top-k of
 * thousands is.
 *   score (rows, n) fp32; pts (rows, n, 2) fp32; out (rows, out_row_points, 2) fp32: entries [0, k) of each row are written,
 *   in INDEX order (the losses sum over points, so only the set matters); equal scores at the threshold: lowest indices first;
 *   NaN ranks highest (as torch.topk).  n <= 38400 (the row's keys live in LDS), else WM2F_EUNSUPPORTED. */
int wm2f_select_top_points(const void* score, const void* pts, void* out, int rows, int n, int k, int out_row_points,
                           void* stream);

/* ---- point-sampled mask loss, batched over the prediction levels --------
 * Replaces, for all levels of a step in one launch each, the per-level tensor work of Mask2FormerLoss.loss_masks
 * (HF:580-640) with sample_points_using_uncertainty (HF:671-724), sample_point (HF:245-274),
 * sigmoid_cross_entropy_loss (HF:308-324) and dice_loss (HF:278-305).
 *   level_maps / level_grads: HOST array of n_levels (<= 16) DEVICE pointers, each (N, H, W) fp32 -- the level
 *   tensors are used where they are, not stacked.   pts (n_levels, M, P, 2) in [0,1] (x, y); index (n_levels, M)
 *   int32 DEVICE: which map of its level row m samples.
 * wm2f_point_sample_levels_fwd: out (n_levels, M, P); neg_abs != 0 stores -|value| (the uncertainty, HF:688-690).
 * wm2f_point_sample_levels_bwd: atomically adds grad_out * bilinear weights into the (zero-initialised) level_grads.
 * wm2f_point_sample_levels_bwd_unique: the same when every (level, index) pair is distinct (the matched rows of a
 *                               one-to-one assignment): each indexed map is accumulated band by band in LDS and
 *                               OVERWRITTEN with plain stores (no global atomics; maps that no row indexes are not
 *                               touched -- clear those).  W <= 16384.
 * wm2f_mask_loss_rows_fwd:      logits, labels (R, P) -> bce_mean (R), dice (R), sums (R, 4) kept for the backward.
 * wm2f_mask_loss_rows_bwd:      grad (R, P) = g_bce[r] * d bce_mean[r] + g_dice[r] * d dice[r]. */
int wm2f_point_sample_levels_fwd(const void* const* level_maps, int n_levels, const void* pts, const int32_t* index,
                                 void* out, int M, int H, int W, int P, int neg_abs, void* stream);
int wm2f_point_sample_levels_bwd(const void* grad_out, const void* pts, const int32_t* index,
                                 void* const* level_grads, int n_levels, int M, int H, int W, int P, void* stream);
int wm2f_point_sample_levels_bwd_unique(const void* grad_out, const void* pts, const int32_t* index,
                                        void* const* level_grads, int n_levels, int M, int H, int W, int P, void* stream);
int wm2f_mask_loss_rows_fwd(const void* logits, const void* labels, void* sums, void* bce_mean, void* dice,
                            int R, int P, void* stream);
int wm2f_mask_loss_rows_bwd(const void* logits, const void* labels, const void* sums, const void* g_bce,
                            const void* g_dice, void* grad, int R, int P, void* stream);

/* ---- instance post-processing on device ----------------------------------
 * Replaces the tensor work of Mask2FormerImageProcessor.post_process_instance_segmentation,
 * transformers 5.15.0 models/mask2former/image_processing_mask2former.py:627-746 (callers: reference
 * models/metrics.py:58-63, models/mask2former/inference.py:30).  The dependency resizes the logits to a fixed
 * gh x gw = 384 x 384 grid (bilinear, :680-682); these kernels evaluate that resize on the fly.
 *   mask_logits (B, Q, h, w) fp32; qidx (B, K) int32 DEVICE: the source query of each selected (query, class)
 *   pair (:698-702).
 * wm2f_instance_scores:        sum_sig[b,k] = sum sigmoid(l) over grid pixels with l > 0, cnt[b,k] = their number
 *                              (:703-708: mask score = sum_sig / (cnt + 1e-6)).
 * wm2f_instance_any:           any_out[b,k] = does the `nearest`-resized (Ho x Wo) mask have a set pixel (:715-717,
 *                              :724); evaluated only where cand[b,k] != 0.
 * wm2f_instance_segmentation:  segmentation (B, Ho, Wo) fp32: id r of the LAST kept instance covering the pixel,
 *                              -1 elsewhere (:712-735).  kept_q (B, K) int32: source query of kept instance r,
 *                              n_kept (B) int32, both DEVICE.
 * wm2f_instance_maps:          the kept binary masks of ONE image, (n, Ho, Wo) fp32 0/1 (return_binary_maps, :741-743);
 *                              image_logits = that image's (Q, h, w) slab. */
int wm2f_instance_scores(const void* mask_logits, const int32_t* qidx, void* sum_sig, void* cnt, int B, int Q,
                         int K, int h, int w, int gh, int gw, void* stream);
int wm2f_instance_any(const void* mask_logits, const int32_t* qidx, const uint8_t* cand, int32_t* any_out, int B,
                      int Q, int K, int h, int w, int gh, int gw, int Ho, int Wo, void* stream);
int wm2f_instance_segmentation(const void* mask_logits, const int32_t* kept_q, const int32_t* n_kept,
                               void* segmentation, int B, int Q, int K, int h, int w, int gh, int gw, int Ho,
                               int Wo, void* stream);
int wm2f_instance_maps(const void* image_logits, const int32_t* kept_q, int n, void* maps, int h, int w, int gh,
                       int gw, int Ho, int Wo, void* stream);

/* ---- label expansion on device ------------------------------------------
 * The tensor work of convert_segmentation_map_to_binary_masks (image_processing_mask2former.py:227-259,
 * image_processing_pil_mask2former.py:81-114), so that a sample can travel as its (H, W) instance-id map instead of
 * the float (T, H, W) mask stack the reference stores (datasets/dataset_utils.py:56-70).
 *   label_map (n_pixels) int32 DEVICE, n_pixels % 4 == 0; ids (T) int32 DEVICE (ascending unique ids, ignore
 *   index removed); masks (T, n_pixels) uint8: masks[t][i] = (label_map[i] == ids[t]). */
int wm2f_labelmap_to_masks(const int32_t* label_map, const int32_t* ids, uint8_t* masks, int64_t n_pixels, int T,
                           void* stream);

#ifdef __cplusplus
}
#endif
#endif /* WM2F_H */
