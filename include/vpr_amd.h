/*
 * vpr_amd.h — C ABI of the MI355X (gfx950) visual-place-recognition + geopose hot path.
 *
 * This is the drop-in boundary.  The reference
 * (anshium/visual-place-recognition-and-geopose-estimation) is pure Python and has no FFI of
 * its own; its seam is the nn.Module.forward contract of the validation scripts.  Each entry
 * point below names the reference interface (file:line, relative to the reference root) whose
 * device-side arithmetic it replaces.  All pointers are DEVICE pointers unless stated, all
 * functions are asynchronous on `stream` (a hipStream_t passed as void*), allocate nothing
 * (safe under hipGraph capture) and never throw: they return a vpr_status (0 = ok, negative =
 * error).  Workspace is caller-provided; size it with the matching *_workspace_bytes() call.
 * State the library keeps between calls — all of it, none of it data-dependent:
 *   (1) per DEVICE (keyed by hipGetDevice() at every launch): which kernels have received their
 *       > 64 KB dynamic-LDS opt-in (hipFuncSetAttribute is a per-device setting) and the cached
 *       CU count.  A process may drive any number of GPUs, from any threads;
 *   (2) per PROCESS: the A/B tuning switches (VPR_KNN_VARIANT, ...), read from the environment
 *       ONCE when the library is loaded — a later setenv() changes nothing — and settable only
 *       through vpr_tuning_set() (benchmark scripts and tests; every value gives identical results).
 *
 * dtype conventions: "bf16" = uint16_t holding the upper 16 bits of an IEEE fp32
 * (round-to-nearest-even); "f32" = float.
 */
#ifndef VPR_AMD_H
#define VPR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VPR_AMD_ABI_VERSION 3

typedef enum vpr_status {
  VPR_OK = 0,
  VPR_ERR_INVALID_ARG = -1,   /* null pointer, non-positive dim, unsupported shape */
  VPR_ERR_UNSUPPORTED = -2,   /* shape outside what the kernels are built for       */
  VPR_ERR_WORKSPACE = -3,     /* workspace too small                                */
  VPR_ERR_LAUNCH = -4         /* hipLaunch / hipMemsetAsync reported an error       */
} vpr_status;

/* Human-readable text for a status code (static storage). */
const char* vpr_status_string(int status);
/* ABI version of the loaded library (== VPR_AMD_ABI_VERSION it was built with). */
int vpr_abi_version(void);
/* Tuning switches (process-wide A/B knobs; names = the VPR_* environment variables read at load: VPR_KNN_VARIANT,
 * VPR_KNN_GEMM_MIN_B, VPR_KNN_GEMM_KSPLIT, VPR_KNN_FP8_GEMM256, VPR_GEMM_NT_STAGES, VPR_GEMM_GROUP_VARIANT,
 * VPR_ATTN_VARIANT, VPR_LN_ROWS, VPR_POSE_KS, VPR_SKINNY_NW, VPR_SKINNY_MBW, VPR_SALAD_VARIANT, VPR_POSE_VARIANT,
 * VPR_LNHEAD_VARIANT, VPR_GEMM256_DEPTH, VPR_HEAD_TRAIN_VARIANT, VPR_GEMM256_STAGGER).  vpr_tuning_set: unset != 0 restores "not set".  vpr_tuning_get: 0 and *value, 1 if the
 * switch is not set, VPR_ERR_INVALID_ARG for an unknown name.  Not for production code paths: no call may be in
 * flight on another thread while a switch changes. */
int vpr_tuning_set(const char* name, int value, int unset);
int vpr_tuning_get(const char* name, int* value);

/* ------------------------------------------------------------------------------------------
 * SALAD optimal-transport aggregation.
 * Replaces: the aggregator inside `feature_extractor(x)` —
 *   dinov2salad/dinov2salad_validation.py:49-51 (call site), :65 (torch.hub "serizba/salad"),
 *   output width 8448 pinned at :44.  Algorithm = SURVEY.md §8a-2 (arXiv:2311.15937).
 *
 * tokens      [B, tokens_per_image, C] bf16, row-major; row 0 of every image is the cls token,
 *             rows 1..n are the n patch tokens (DINOv2 layout; n = tokens_per_image-1 = 256).
 * weights     bf16 [out,in] row-major (nn.Linear / 1x1-conv layout), biases f32:
 *   w1_sc [2*hidden, C]  rows 0..hidden-1   = score.0.weight,  rows hidden.. = cluster_features.0.weight
 *   b1_sc [2*hidden]
 *   w2_s  [m, hidden], b2_s [m]          (score.3)
 *   w2_c  [l, hidden], b2_c [l]          (cluster_features.3)
 *   w1_t  [hidden, C], b1_t [hidden]     (token_features.0)
 *   w2_t  [t, hidden], b2_t [t]          (token_features.2)
 * dustbin     host float (aggregator.dust_bin)
 * out_f32     [B, t + l*m] f32 L2-normalised descriptor: [token(t) | V flattened l-major (l*m + m_idx)]
 * out_bf16    same, rounded to bf16 (may be NULL) — the kNN query/gallery format.
 * Built for n = 256, m = 64, l = 128, t = 256, hidden = 512, C % 64 == 0, sinkhorn_iters >= 1.
 * ------------------------------------------------------------------------------------------ */
typedef struct vpr_salad_weights {
  const uint16_t* w1_sc; const float* b1_sc;
  const uint16_t* w2_s;  const float* b2_s;
  const uint16_t* w2_c;  const float* b2_c;
  const uint16_t* w1_t;  const float* b1_t;
  const uint16_t* w2_t;  const float* b2_t;
  /* optional (may be NULL): w2_s / w2_c re-laid in MFMA fragment order by vpr_salad_pack_w2_fragments — the fused MLP
   * kernel then fetches its second-layer operands as contiguous 1 KB wave loads instead of 16 rows x 64 B each */
  const uint16_t* w2_s_frag;
  const uint16_t* w2_c_frag;
} vpr_salad_weights;

/* w2 [n_out, hidden] bf16 row-major -> out (same element count), fragment order of the fused kernel's second GEMM:
 * out[((((s * (n_out/16) + ob) * 8 + ks) * 64 + lane) * 8 + e] = w2[ob*16 + (lane & 15)][s*256 + ks*32 + 8*(lane >> 4) + e]
 * (s = 256-column slab of the hidden layer, ob = block of 16 outputs, ks = 32-deep k-step, lane = wavefront lane).
 * n_out % 16 == 0, hidden % 256 == 0.  Done once per weight set (a 200 KB copy). */
int vpr_salad_pack_w2_fragments(const uint16_t* w2, int n_out, int hidden, uint16_t* out, void* stream);

/* WORKSPACE CONTRACT (round 3): the first 4096 bytes of the SALAD workspace are reserved for the arrival counters of the
 * four-workgroups-per-image form of the aggregation kernel (an A/B option, VPR_SALAD_VARIANT=3; the default form does not
 * touch them): zero before the first call on a buffer, left zero by every call — memset once after allocation; calls of
 * different shapes may share a workspace. */
size_t vpr_salad_workspace_bytes(int B, int n, int C, int m, int l, int t, int hidden);

int vpr_salad_aggregate(const uint16_t* tokens, int B, int tokens_per_image, int C,
                        const vpr_salad_weights* w, float dustbin,
                        int m, int l, int t, int hidden, int sinkhorn_iters,
                        float* out_f32, uint16_t* out_bf16,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Same aggregation with the patch tokens and the cls tokens in separate contiguous arrays (the layout
 * the backbone computes in): patch_tokens [B, n, C] bf16, cls_tokens [B, C] bf16. */
int vpr_salad_aggregate_split(const uint16_t* patch_tokens, const uint16_t* cls_tokens, int B,
                              int patches_per_image, int C,
                              const vpr_salad_weights* w, float dustbin,
                              int m, int l, int t, int hidden, int sinkhorn_iters,
                              float* out_f32, uint16_t* out_bf16,
                              void* workspace, size_t workspace_bytes, void* stream);

/* The aggregation as its three stages, for callers that overlap them (same kernels, same workspace layout, same result as
 * the one-call forms above; `workspace` must be the same buffer in all three):
 *   stage_token      token MLP on the B cls rows                          -> workspace   (reads cls tokens only)
 *   stage_mlps       score + cluster MLPs on the B*n patch rows           -> workspace   (reads patch tokens only)
 *   stage_aggregate  Sinkhorn, aggregation, normalisations                -> out_f32 / out_bf16
 * stage_token and stage_mlps touch disjoint workspace regions and may run concurrently on two streams (the token MLP is a
 * 10 us weight stream, the MLPs a 40 us all-CU GEMM); stage_aggregate must be ordered after both.
 * patch row r of image b at patch_tokens + b*patch_img_stride + r*C, cls token of image b at cls_tokens + b*cls_stride
 * (elements; strides % 8 == 0).  With hidden = 512 the MLP stage is ONE kernel: the second layers are fused into the
 * layer-1 tile epilogue and leave two partial-sum slabs the aggregation stage adds (no hidden activations in HBM). */
int vpr_salad_stage_token(const uint16_t* cls_tokens, long long cls_stride, int B, int n, int C,
                          const vpr_salad_weights* w, int m, int l, int t, int hidden,
                          void* workspace, size_t workspace_bytes, void* stream);
int vpr_salad_stage_mlps(const uint16_t* patch_tokens, long long patch_img_stride, int B, int n, int C,
                         const vpr_salad_weights* w, int m, int l, int t, int hidden,
                         void* workspace, size_t workspace_bytes, void* stream);
int vpr_salad_stage_aggregate(int B, int n, int C, float dustbin, int m, int l, int t, int hidden,
                              int sinkhorn_iters, float* out_f32, uint16_t* out_bf16,
                              void* workspace, size_t workspace_bytes, void* stream);

/* The same aggregation at the reference's own precision: f32 tokens, f32 weights, f32-accurate arithmetic end to end
 * (dinov2salad/dinov2salad_validation.py:65-66,80-81 run the extractor in fp32: `.cuda()`, no cast).  Every linear
 * layer runs on the bf16 matrix pipe with both operands split into three bf16 planes (x = h + m + l exactly) and the six
 * products above 2^-24 of the leading one kept — exact products, f32 accumulation: an f32 GEMM; hidden activations stay
 * f32 (no bf16 rounding point).  ~6x the matrix work of vpr_salad_aggregate: a precision mode for parity runs.
 * patch row r of image b at patch + b*patch_img_stride + r*C (elements: covers [B, n, C] and the patch rows inside a
 * hub-layout [B, 1+n, C] tensor), cls token of image b at cls + b*cls_stride.  Weights: the shapes of vpr_salad_weights,
 * all f32.  Same output contract.  Pointers 16-byte aligned, strides % 4 == 0. */
typedef struct vpr_salad_weights_f32 {
  const float* w1_sc; const float* b1_sc;
  const float* w2_s;  const float* b2_s;
  const float* w2_c;  const float* b2_c;
  const float* w1_t;  const float* b1_t;
  const float* w2_t;  const float* b2_t;
} vpr_salad_weights_f32;

size_t vpr_salad_f32_workspace_bytes(int B, int n, int C, int m, int l, int t, int hidden);

int vpr_salad_aggregate_f32(const float* patch, long long patch_img_stride, const float* cls, long long cls_stride,
                            int B, int n, int C, const vpr_salad_weights_f32* w, float dustbin,
                            int m, int l, int t, int hidden, int sinkhorn_iters,
                            float* out_f32, uint16_t* out_bf16,
                            void* workspace, size_t workspace_bytes, void* stream);

/* Sinkhorn + aggregation stage alone (scores/features already computed) — exposed so tests can
 * pin it against closed-form known answers.
 * scores [B, n, m] f32 (token-major), feats [B, n, l] f32, tokfeat [B, t] f32 (un-normalised). */
int vpr_salad_sinkhorn_aggregate(const float* scores, const float* feats, const float* tokfeat,
                                 int B, int n, int m, int l, int t, float dustbin,
                                 int sinkhorn_iters, float* out_f32, uint16_t* out_bf16,
                                 void* stream);

/* Generic C[M,N] = act(A[M,K] * W[N,K]^T + bias) on MFMA (bf16 in, f32 accumulate).
 * The SALAD MLPs are built from it; exposed for parity tests of the dense stage.
 * A row r lives at A + (r / a_group_rows) * a_group_stride + (r % a_group_rows) * lda (elements);
 * pass a_group_rows = 0 for a plain lda matrix.  out_is_bf16 selects the C element type.
 * Requires K % 64 == 0. */
int vpr_gemm_nt_bf16(const uint16_t* A, int lda, int a_group_rows, long long a_group_stride,
                     const uint16_t* W, int ldw, const float* bias, int relu,
                     void* C, int ldc, int out_is_bf16, int M, int N, int K, void* stream);

/* Same operation on 256 x 256 output tiles with LDS-DMA kept in flight across barriers (the
 * large-M / large-N form: SALAD layer 1 is exactly 256 such tiles at B = 64).  Additional
 * requirements: K >= 128, ldc % 4 == 0, C 16-byte aligned. */
int vpr_gemm256_nt_bf16(const uint16_t* A, int lda, int a_group_rows, long long a_group_stride,
                        const uint16_t* W, int ldw, const float* bias, int relu,
                        void* C, int ldc, int out_is_bf16, int M, int N, int K, void* stream);

/* ------------------------------------------------------------------------------------------
 * kNN: cosine top-k of L2-normalised descriptors against a gallery shard.
 * Replaces: nothing in the reference tree (north-star addition between the
 * SALAD descriptor, dinov2salad_validation.py:51, and the post-processing at :84).
 *
 * q        [B, D] bf16, gallery [N, D] bf16 (row-major, D % 64 == 0)
 * out_val  [B, k] f32 scores, descending; out_idx [B, k] int32 = index_base + local row.
 * Ordering contract: key = (score desc, index asc) where score = fp32 rounding of the exact
 * (fp64-accumulated) dot product of the bf16 operands — independent of tile order, so shards
 * merged with vpr_topk_merge give the same answer as one unsharded call.
 * If N < k the tail is filled with (-inf, -1).   1 <= k <= 64.
 * ------------------------------------------------------------------------------------------ */
size_t vpr_knn_workspace_bytes(int B, int N, int D, int k);

int vpr_knn_topk(const uint16_t* q, const uint16_t* gallery, int B, int N, int D, int k,
                 int index_base, float* out_val, int32_t* out_idx,
                 void* workspace, size_t workspace_bytes, void* stream);

/* Stage entry points of vpr_knn_topk (same workspace layout), so the dominant kernel can be
 * timed by itself (bench.py roofline) and tested by itself.
 *   scores:  S[b, n] = <q_b, g_n>  (bf16 MFMA, f32 accumulate)  -> workspace
 *            B <= 64: the HBM-streaming kernel (one gallery pass, 208- or 256-row tiles by shard size); B > 64 (the
 *            all-gathered batch of a multi-GPU job): an MFMA GEMM, 128 x 128 tiles or — 256-row query tiles at least 3/4
 *            full and >= 256 tiles — the 256 x 256-tile kernel.  vpr_knn_scores_kernel_name() tells which.
 *   select:  per-query candidate selection + exact rescoring + final ordering + certificate (see "Checked forms") */
int vpr_knn_scores(const uint16_t* q, const uint16_t* gallery, int B, int N, int D,
                   void* workspace, size_t workspace_bytes, void* stream);
int vpr_knn_select(const uint16_t* q, const uint16_t* gallery, int B, int N, int D, int k,
                   int index_base, float* out_val, int32_t* out_idx,
                   void* workspace, size_t workspace_bytes, void* stream);
/* Name (as a kernel trace shows it, without the argument list) of the kernel vpr_knn_topk[_fp8] launches for the
 * score stage of a B-query batch against an N-row shard of 8448-d descriptors: lets bench.py tie its roofline line to
 * profiles/ by name.  <= 64 queries: knn_scores_kernel<fp8, tile rows, 2, flags> (208-row tiles, K split over up to
 * 16 slabs below 106k rows; 256-row tiles above; above 131k rows — several tiles per workgroup — flags 52: score
 * tiles leave as whole row segments through LDS with nt stores); more queries: gemm_nt / gemm256 (split-K for small
 * shards). */
const char* vpr_knn_scores_kernel_name(int is_fp8, int B, int N);
/* Device pointer + leading dimension of the score matrix inside a workspace (for tests). */
float* vpr_knn_scores_ptr(void* workspace, int B, int N, int D, int k, int* ld_out);

/* fp8 variant (BASELINE config 5): operands are OCP e4m3 bytes with one f32 scale per row
 * (value = scale * fp8).  score = fp32 rounding of (sum_i q_i*g_i, exact in f64) * q_scale * g_scale
 * evaluated in f64 in that order; same ordering contract, same workspace size.  D % 128 == 0. */
int vpr_knn_topk_fp8(const uint8_t* q, const float* q_scale, const uint8_t* gallery,
                     const float* gallery_scale, int B, int N, int D, int k, int index_base,
                     float* out_val, int32_t* out_idx, void* workspace, size_t workspace_bytes,
                     void* stream);

/* Checked forms: the exactness contract above, certified per query on the device.
 * The search is approximate-then-exact: MFMA scores pick KP = max(2k, k+8) candidates per query, those are
 * rescored exactly (f64) and ordered.  That is the exact top-k iff no row outside the candidate set has an exact
 * score >= the k-th best exact score e_k.  Every such row has an MFMA score <= a_min (the smallest MFMA score
 * kept) and |MFMA score - exact score| <= eps = 1.1 * D * 2^-24 * |q| * gallery_norm_bound (f32 accumulation of
 * exact products in any order), so  e_k - eps > a_min  proves it.  Otherwise the kernel widens the candidate set to
 * every first-level candidate with MFMA score >= e_k - eps and rescoring them too (more than KP - k gallery rows
 * within eps of the k-th score, e.g. near-duplicate frames).  status[b]:
 *   0  certified at once;   1  certified after widening;
 *   2  NOT certified (more than KP such rows inside one 8192-row chunk, > 256 candidates, or a shard so large /
 *      rows so wide that the search ran the multi-level path): out_* hold the best effort; re-run those queries
 *      with vpr_knn_topk_exhaustive.  `uncertified` (device int32, may be NULL) is incremented once per status-2
 *      query, so a pipeline can check one word at the end instead of synchronising per batch.
 * gallery_norm_bound: upper bound of the L2 norms of the (dequantised) gallery rows; the unchecked entry points
 * assume L2-normalised descriptors (1.002 bf16, 1.0625 e4m3).  status may be NULL. */
int vpr_knn_topk_checked(const uint16_t* q, const uint16_t* gallery, int B, int N, int D, int k,
                         int index_base, float* out_val, int32_t* out_idx, void* workspace, size_t workspace_bytes,
                         float gallery_norm_bound, int32_t* status, int32_t* uncertified, void* stream);
int vpr_knn_select_checked(const uint16_t* q, const uint16_t* gallery, int B, int N, int D, int k,
                           int index_base, float* out_val, int32_t* out_idx, void* workspace, size_t workspace_bytes,
                           float gallery_norm_bound, int32_t* status, int32_t* uncertified, void* stream);
int vpr_knn_topk_fp8_checked(const uint8_t* q, const float* q_scale, const uint8_t* gallery,
                             const float* gallery_scale, int B, int N, int D, int k, int index_base,
                             float* out_val, int32_t* out_idx, void* workspace, size_t workspace_bytes,
                             float gallery_norm_bound, int32_t* status, int32_t* uncertified, void* stream);

/* vpr_knn_topk[_fp8]_checked as two calls — the HBM-bound score stage and the latency-bound tail (level-0 select,
 * exact rescoring, ordering, certificate) — for callers that record events or enqueue other work between them
 * (bench.py times the score kernel this way inside the end-to-end step).  Same kernels and results as the one-call
 * form; between the two calls the workspace holds the scores in a private layout (K-slice slabs for small shards and
 * for gathered batches; use vpr_knn_scores for a plain score matrix).  bf16: is_fp8 = 0, scales NULL. */
int vpr_knn_topk_scores_stage(const void* q, const float* q_scale, const void* gallery, const float* gallery_scale,
                              int is_fp8, int B, int N, int D, int k, void* workspace, size_t workspace_bytes,
                              void* stream);
int vpr_knn_topk_select_stage(const void* q, const float* q_scale, const void* gallery, const float* gallery_scale,
                              int is_fp8, int B, int N, int D, int k, int index_base, float* out_val,
                              int32_t* out_idx, void* workspace, size_t workspace_bytes, float gallery_norm_bound,
                              int32_t* status, int32_t* uncertified, void* stream);

/* Exhaustive form (the fallback for status-2 queries): every score computed exactly (f64 accumulation, one wave
 * per gallery row), then the same selection — exact by construction, ~B gallery passes of f64 work: meant for a
 * handful of queries.  q / gallery: bf16 (is_fp8 = 0, scales NULL) or e4m3 bytes with per-row scales.
 * Rows of at most 17408 bytes.  Same workspace as vpr_knn_topk. */
int vpr_knn_topk_exhaustive(const void* q, const float* q_scale, const void* gallery, const float* gallery_scale,
                            int is_fp8, int B, int N, int D, int k, int index_base, float* out_val,
                            int32_t* out_idx, void* workspace, size_t workspace_bytes, void* stream);

/* Per-row symmetric quantisation f32 -> e4m3: scale = max|x|/448 (1 for a zero row),
 * q = fp8_rne(x / scale).  x [rows, D] f32 (D % 4 == 0), q [rows, D] bytes, scale [rows] f32. */
int vpr_quantize_fp8_rows(const float* x, long long rows, int D, uint8_t* q, float* scale, void* stream);

/* Merge per-shard top-k lists (after an all-gather): vals/idxs [shards, B, k] -> [B, k],
 * same ordering contract.  Entries with idx < 0 are padding. */
int vpr_topk_merge(const float* vals, const int32_t* idxs, int shards, int B, int k,
                   float* out_val, int32_t* out_idx, void* stream);

/* ------------------------------------------------------------------------------------------
 * Pose head: out = W2 * relu(W1 * x + b1) + b2   (f32 end to end, f32-input MFMA),
 * optionally followed by F.normalize(p=2, eps=1e-6) of the output pair
 * [sincos_offset, sincos_offset+1].
 * Replaces: DINOv2RegressionModel.regressor  dinov2salad/dinov2salad_validation.py:43-47,52
 *           Swin-Base MLP head              swin_transformer/val_and_test_swin_2.py:168-177
 *           sin/cos heads  angle_prediction/swin/swin_angle_finetuning_sin_cos.py:56-62,
 *                          angle_prediction/swin/swin_angle_finetuning_gemini.py:101-106
 * With hidden = 0 the head is the single Linear(D, n_out): out = W2 * x + b2 (W1,b1 ignored)
 *           swin_transformer/swin_validation.py:41,46.
 * x [B, D] f32; W1 [hidden, D]; b1 [hidden]; W2 [n_out, hidden or D]; b2 [n_out]; out [B, n_out].
 * A fused (lat, lon, sin, cos) head is n_out = 4 with row-concatenated W2 (block structure is
 * the caller's business) and sincos_offset = 2.  sincos_offset < 0 disables the normalise.
 * Requires D % 16 == 0, hidden % 32 == 0 (or 0), 1 <= n_out <= 8.
 * ------------------------------------------------------------------------------------------ */
size_t vpr_pose_head_workspace_bytes(int B, int D, int hidden, int n_out);

int vpr_pose_head(const float* x, const float* W1, const float* b1,
                  const float* W2, const float* b2, float* out,
                  int B, int D, int hidden, int n_out, int sincos_offset,
                  void* workspace, size_t workspace_bytes, void* stream);

/* The same MLP head with the first layer on the bf16 matrix pipe at f32 accuracy: W1 packed once by
 * vpr_pose_head_pack_w1 into two bf16 planes (hi = bf16(w), lo = bf16(w - hi): the same 4 bytes per weight),
 * x split the same way on the fly, (x_hi + x_lo)(w_hi + w_lo) = four exact-product MFMAs with f32 accumulation
 * (<= 2^-16 relative per product; the f32 MFMA of vpr_pose_head runs at 1/16 of the bf16 rate and costs as
 * much time as the weight stream).  Split-K slabs summed in a fixed order: bitwise reproducible.
 * Requires D % 32 == 0, hidden % 16 == 0, hidden > 0, 1 <= n_out <= 8; W1_hi / W1_lo [hidden, D]. */
int vpr_pose_head_pack_w1(const float* W1, long long count, uint16_t* hi, uint16_t* lo, void* stream);
size_t vpr_pose_head_split_workspace_bytes(int B, int D, int hidden);
int vpr_pose_head_split(const float* x, const uint16_t* W1_hi, const uint16_t* W1_lo, const float* b1,
                        const float* W2, const float* b2, float* out, int B, int D, int hidden,
                        int n_out, int sincos_offset, void* workspace, size_t workspace_bytes, void* stream);

/* Single-launch form of the same head (round 3).  W1's planes in MFMA FRAGMENT order (vpr_pose_head_pack_w1_frag:
 * plane[((nb * (D/32) + s) * 64 + lane) * 8 + e] = W1[nb*16 + (lane & 15)][s*32 + 8*(lane >> 4) + e] — a wave's weight
 * load is then one contiguous 1 KB), and the split-K slabs finished inside the kernel by arrival counters: the workgroup
 * that arrives last at a (hidden tile, batch tile) adds that tile's slabs in slice order, applies bias / ReLU / its
 * columns of W2; the last of those per batch tile adds the second-layer partials in tile order, b2, the pair normalise.
 * Same arithmetic and summation orders as vpr_pose_head_split: bitwise reproducible run to run.
 * WORKSPACE CONTRACT: the first vpr_pose_head_fused_counter_bytes() bytes (4096, whatever the shape) hold the arrival
 * counters; they must be ZERO before the first call on a buffer and are left zero by every call (memset once after
 * allocation, never again); calls of different shapes may share a workspace.  Shapes with more than 1024 (hidden tile,
 * batch tile) pairs return VPR_ERR_UNSUPPORTED (use vpr_pose_head_split).
 * Requires D % 32 == 0, hidden % 16 == 0, 1 <= n_out <= 8, 16-byte aligned pointers. */
size_t vpr_pose_head_fused_workspace_bytes(int B, int D, int hidden);
size_t vpr_pose_head_fused_counter_bytes(int B, int D, int hidden);
int vpr_pose_head_pack_w1_frag(const float* W1, int hidden, int D, uint16_t* hi, uint16_t* lo, void* stream);
int vpr_pose_head_fused(const float* x, const uint16_t* W1_hi_frag, const uint16_t* W1_lo_frag, const float* b1,
                        const float* W2, const float* b2, float* out, int B, int D, int hidden,
                        int n_out, int sincos_offset, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * One optimizer step of the two-layer regression head on CACHED descriptors (head-only fine-tuning):
 *   preds = W2 relu(W1 x + b1) + b2 ; loss = mean((preds - y)^2) over the B * n_out elements ; backward ; AdamW.
 * Replaces, per batch, dinov2salad/dinov2salad_finetuning.py:119-125 (`preds = model(inputs)`, `loss_fn(preds, targets)`,
 * `optimizer.zero_grad()`, `loss.backward()`, `optimizer.step()`) with `optimizer = torch.optim.AdamW(lr=1e-5)` (:95) and
 * `loss_fn = nn.MSELoss()` (:96), for a frozen extractor whose descriptors were computed once.
 * X [rows, x_stride] f32 descriptors, Y [rows, y_stride] f32 (standardised) targets; idx [B] int32 device = the rows of this
 * batch (NULL: rows 0..B-1; the caller guarantees 0 <= idx < rows).  W1 [hidden, D], b1 [hidden], W2 [n_out, hidden],
 * b2 [n_out]: f32, updated IN PLACE.  m, v: AdamW moments, vpr_head_train_state_floats() floats each, laid out
 * [W1 | b1 | W2 | b2], zero before step 1, updated in place.  step = 1, 2, ... (number of this update: bias corrections).
 * Hyper-parameters are doubles, as Python holds them; every derived scalar (1 - lr*wd, lr / (1 - beta1^step), ...) is formed in
 * double and rounded to f32 once, and the update follows torch.optim.AdamW's order of operations.
 * loss_kind: VPR_LOSS_MSE (nn.MSELoss, the script above) or VPR_LOSS_HUBER with huber_delta > 0 (nn.HuberLoss(delta): the loss of
 * dinov2salad_finetuning_2.py:154 / swin_transformer/swin_attempt_2.py:158 — 0.5 d^2 for |d| < delta, delta (|d| - 0.5 delta) beyond;
 * mean over the B * n_out elements); huber_delta is ignored for MSE.
 * loss_out (device, may be NULL) receives this batch's loss (before the update).  All arithmetic f32 with fixed summation
 * orders: bitwise reproducible.  The gradient of W1 is never written to memory (formed in registers, consumed by the update).
 * Requires 1 <= B <= 64, D % 16 == 0, hidden % 32 == 0, 1 <= n_out <= 8, x_stride % 4 == 0, 16-byte aligned X / W1 / m / v /
 * workspace; workspace contents are scratch.  Status codes as everywhere (VPR_ERR_UNSUPPORTED for other shapes).
 * ------------------------------------------------------------------------------------------ */
#define VPR_LOSS_MSE 0
#define VPR_LOSS_HUBER 1
size_t vpr_head_train_workspace_bytes(int B, int D, int hidden, int n_out);
long long vpr_head_train_state_floats(int D, int hidden, int n_out);
int vpr_head_train_step(const float* X, long long x_stride, const int* idx, const float* Y, long long y_stride,
                        int B, int D, int hidden, int n_out, float* W1, float* b1, float* W2, float* b2,
                        float* m, float* v, int step, double lr, double beta1, double beta2, double eps,
                        double weight_decay, int loss_kind, double huber_delta, float* loss_out, void* workspace,
                        size_t workspace_bytes, void* stream);
/* A whole pass in one call: batches order[0:bs], order[bs:2bs], ... of the n rows listed in `order` (device int32; the last
 * batch may be short, as DataLoader's default keeps it: dinov2salad_finetuning.py:89), steps first_step, first_step + 1, ...;
 * losses [ceil(n / batch_size)] (device, may be NULL) receives the batch losses (their mean is the reference's epoch figure,
 * :126-128).  3 * ceil(n / batch_size) launches are enqueued; nothing waits for the GPU.  The workspace must hold
 * vpr_head_train_workspace_bytes(min(batch_size, n), ...) bytes.  Arguments are validated before the first launch. */
int vpr_head_train_epoch(const float* X, long long x_stride, const int* order, int n, int batch_size,
                         const float* Y, long long y_stride, int D, int hidden, int n_out,
                         float* W1, float* b1, float* W2, float* b2, float* m, float* v, int first_step,
                         double lr, double beta1, double beta2, double eps, double weight_decay,
                         int loss_kind, double huber_delta, float* losses, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Swin pooler + linear head: pooled = mean_t LayerNorm(x[b,t,:]) ; out = Wh * pooled + bh
 * Replaces: `outputs.pooler_output` + `self.regressor`  swin_transformer/swin_validation.py:43-46
 *           (HF SwinModel: layernorm -> AdaptiveAvgPool1d(1) over tokens) and the normalised
 *           sin/cos variant angle_prediction/swin/swin_angle_finetuning_sin_cos.py:58-62.
 * x [B, T, H] (bf16 if x_is_bf16 else f32); gamma, beta [H] f32; eps as in the model config.
 * pooled_out [B, H] f32 (may be NULL); Wh [n_out, H], bh [n_out], out [B, n_out] (Wh may be
 * NULL to skip the head, e.g. when an MLP head follows through vpr_pose_head).
 * H in {512, 768, 1024, 1536}; 0 <= n_out <= 8.
 * ------------------------------------------------------------------------------------------ */
int vpr_ln_meanpool_head(const void* x, int x_is_bf16, int B, int T, int H,
                         const float* gamma, const float* beta, float eps,
                         float* pooled_out,
                         const float* Wh, const float* bh, int n_out, int sincos_offset,
                         float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Image preprocessing on the GPU: PIL-exact antialiased resize + ToTensor + Normalize.
 * Replaces the per-item host transforms: Resize((224,224)) -> ToTensor -> Normalize
 *   dinov2salad/dinov2salad_validation.py:18-22, dinov2salad_finetuning.py:45-50, and the HF
 *   image processor at swin_transformer/swin_validation.py:30.
 * in [B,H,W,3] u8 (RGB, HWC).  kx [OW,ksize_x] / ky [OH,ksize_y] are Pillow's 22-bit fixed-point
 * coefficients, xbounds/ybounds [out,2] = (first input index, tap count) — all device int32,
 * produced on the host by vpr_amd.preprocess.  mean3/std3 are HOST float[3].
 * out [B,3,OH,OW] f32 or bf16; out_u8 (optional) [B,OH,OW,3] receives the resized bytes.
 * ------------------------------------------------------------------------------------------ */
size_t vpr_preprocess_workspace_bytes(int B, int H, int OW);
int vpr_preprocess_resize_normalize(const uint8_t* in, int B, int H, int W, int OH, int OW,
                                    const int32_t* kx, const int32_t* xbounds, int ksize_x,
                                    const int32_t* ky, const int32_t* ybounds, int ksize_y,
                                    const float* mean3, const float* std3,
                                    void* out, int out_is_bf16, uint8_t* out_u8,
                                    void* workspace, size_t workspace_bytes, void* stream);

/* Row LayerNorm on bf16 activations (backbone helper; PyTorch semantics, f32 statistics):
 * y[r,:] = (x[r,:] - mean) / sqrt(var + eps) * gamma + beta.  x,y [M,C] bf16; gamma/beta [C] bf16 or f32.
 * C % 8 == 0, C <= 2048. */
int vpr_layernorm_bf16(const uint16_t* x, const void* gamma, const void* beta, int params_are_bf16,
                       float eps, uint16_t* y, long long M, int C, void* stream);

/* Residual add fused with the LayerNorm that follows it: sum_out = bf16(x + res); y = LayerNorm(sum_out).
 * Same shapes / constraints as vpr_layernorm_bf16; sum_out may alias x. */
int vpr_add_layernorm_bf16(const uint16_t* x, const uint16_t* res, uint16_t* sum_out, const void* gamma,
                           const void* beta, int params_are_bf16, float eps, uint16_t* y, long long M,
                           int C, void* stream);

/* LayerNorm of (x + pre_bias): pre_bias [C] f32 is a per-column offset kept OUTSIDE the bf16 residual
 * stream (the running sum of the proj / fc2 biases when the residual add lives in the GEMM, beta = 1).
 * y = LayerNorm(f32(x) + pre_bias); x is not modified.  Constraints as vpr_layernorm_bf16; pre_bias 16-B aligned. */
int vpr_bias_layernorm_bf16(const uint16_t* x, const float* pre_bias, const void* gamma, const void* beta,
                            int params_are_bf16, float eps, uint16_t* y, long long M, int C, void* stream);

/* Linear layer on a few rows (backbone helper: the cls-token rows of the split row layout):
 * mode 0: out = in W^T + bias; mode 1: out = gelu_tanh(in W^T + bias); mode 2: out += in W^T (bf16 read-modify-write,
 * bias unused); mode 3: relu(in W^T + bias); mode 4: exact (erf) GELU, the nn.GELU() of DINOv2's Mlp.  in [M, K] bf16 (ldi), W [N, K] bf16 (ldw), bias [N] bf16 or f32, out [M, N] bf16 (ldo).
 * K % 32 == 0, ldi/ldw % 8 == 0; f32 accumulation in a fixed order (deterministic). */
int vpr_skinny_linear_bf16(const uint16_t* in, int ldi, const uint16_t* W, int ldw, const void* bias,
                           int bias_is_bf16, int mode, uint16_t* out, int ldo, int M, int N, int K, void* stream);

/* vpr_bias_layernorm_bf16 over all M rows (-> y) AND, in the same launch, a linear layer on the LayerNorm of rows
 * [cls_row0, cls_row0 + n_cls):  out[n_cls, N] = act(LN(x[cls rows] + pre_bias) W^T + b), act = identity or tanh-GELU,
 * computed as  rstd * (x W'^T + cprime - mean * colsum) + bprime  on the raw rows, with the static per-layer operands
 * W_scaled = W diag(gamma) [N, C] bf16, colsum[n] = sum_k W'[n,k], cprime = W' pre_bias, bprime = b + W beta (f32 [N]).
 * row_stats [C/16, n_cls, 2] f32: (mean, centred sum of squares) of (cls rows + pre_bias) per 16-column block, as
 * vpr_skinny_linear_stats_bf16 leaves them.  Saves the separate 64-row launch after each LayerNorm of the split
 * row layout.  gamma/beta bf16, pre_bias f32 or NULL, C % 32 == 0, C <= 2048. */
int vpr_bias_layernorm_cls_linear_bf16(const uint16_t* x, const float* pre_bias, const uint16_t* gamma,
                                       const uint16_t* beta, float eps, uint16_t* y, long long M, int C,
                                       long long cls_row0, int n_cls, const float* row_stats,
                                       const uint16_t* W_scaled, int ldw, const float* colsum,
                                       const float* cprime, const float* bprime, int gelu,
                                       uint16_t* out, int ldo, int N, void* stream);

/* vpr_skinny_linear_bf16 that also leaves, for every output row and 16-column block, the (mean, centred sum of
 * squares) of bf16(out) + stats_bias in row_stats [N/16, M, 2] f32 (N % 16 == 0; stats_bias [N] f32 or NULL):
 * the LayerNorm statistics of the rows it just wrote, for vpr_bias_layernorm_cls_linear_bf16. */
int vpr_skinny_linear_stats_bf16(const uint16_t* in, int ldi, const uint16_t* W, int ldw, const void* bias,
                                 int bias_is_bf16, int mode, uint16_t* out, int ldo, int M, int N, int K,
                                 const float* stats_bias, float* row_stats, void* stream);

/* Patch extraction for the ViT patch embedding (backbone helper; replaces the stride-P conv's im2col):
 * images [B, Cin, H, W] bf16 -> out [B * (lead_rows + (H/P)*(W/P)), kpad] bf16,
 * out[b*(lead+n) + lead + py*(W/P) + px][c*P*P + i*P + j] = images[b][c][py*P+i][px*P+j]; the lead_rows rows
 * of every image and columns >= Cin*P*P are zero.  H % P == W % P == 0, W % 8 == 0, kpad % 8 == 0. */
int vpr_patchify_bf16(const uint16_t* images, int B, int Cin, int H, int W, int patch, int kpad,
                      int lead_rows, uint16_t* out, void* stream);

/* Multi-head self-attention for short ViT sequences (backbone helper): softmax(q k^T * scale) v, non-causal.
 * qkv [B, T, 3, H, 64] bf16 (the fused projection output), out [B, T, H*64] bf16.  T <= 288, head_dim == 64. */
int vpr_attention_qkv_bf16(const uint16_t* qkv, uint16_t* out, int B, int T, int H, int head_dim,
                           float scale, void* stream);

/* Same attention on the "body rows first, tail rows last" token layout the backbone uses so that its
 * GEMMs see M = B * body_tokens (an exact number of 256-row tiles at 256 patches per image):
 * token t of image b is row b*body_tokens + t for t < body_tokens, else row
 * tail_row0 + b*(T - body_tokens) + (t - body_tokens), in qkv [rows, 3, H, 64] and out [rows, H*64]. */
int vpr_attention_qkv_split_bf16(const uint16_t* qkv, uint16_t* out, int B, int T, int body_tokens,
                                 long long tail_row0, int H, int head_dim, float scale, void* stream);

/* Utility: f32 -> bf16 (RNE) row copy, used to build galleries from f32 descriptors. */
int vpr_f32_to_bf16(const float* src, uint16_t* dst, long long count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* VPR_AMD_H */
