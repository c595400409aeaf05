/*
 * mmx_relevancy.h -- C-ABI of libmmx_hip.so, the MI355X (gfx950) relevancy-propagation engine.
 *
 * The reference (hila-chefer/Transformer-MM-Explainability) has NO FFI boundary: its hot path is
 * Python calling stock ATen kernels.  The entry points below are what a ctypes binding of that path
 * binds instead of those ATen call sequences; each cites the reference site it replaces
 * (paths relative to the reference checkout).  INTEGRATION.md shows the reference-side stub.
 *
 * Conventions
 *   - every pointer named *_dev / documented "device" is a HIP device pointer owned by the caller;
 *     pointer TABLES (const void* const*) are HOST arrays of device pointers
 *   - plain C types only: no torch / hip types in the signatures; `stream` is a hipStream_t passed
 *     as void* (NULL = default stream)
 *   - all launches are stream-ordered, never synchronise, never allocate (scratch is caller-provided;
 *     sizes from the *_workspace_bytes queries)
 *   - return value: 0 = ok, MMX_E* < 0 on error (no exceptions cross the boundary);
 *     mmx_last_error() gives a thread-local message
 *   - relevancy matrices R are always fp32 row-major; captured attention / gradient buffers are
 *     `dtype` (MMX_F32 / MMX_F16 / MMX_BF16), row-major [B, H, Nq, Nk] with index b*H + h
 *     (== CLIP's [B*H, N, N], CLIP/clip/auxilary.py:194)
 */
#ifndef MMX_RELEVANCY_H
#define MMX_RELEVANCY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Within a version entry points are only ever ADDED (the *_ex forms); a binding checks ==.
 * 2 (round 5 / 6): mmx_lxmert_schedule_ex and mmx_lxmert_schedule_v2 removed, mmx_lxmert_schedule gained `text_len_dev`,
 *    `workspace_dev`, `workspace_bytes` (a caller built against version 1 would pass `stream` where a pointer is expected), and
 *    several mmx_set_option keys now return MMX_EINVAL -- an incompatible change, hence the bump.  A version-1 binding fails its
 *    mmx_abi_version() check instead of misreading arguments. */
#define MMX_ABI_VERSION 2
#define MMX_MAX_LAYERS 48

enum mmx_dtype { MMX_F32 = 0, MMX_F16 = 1, MMX_BF16 = 2 };
/* OR-ed into the `slab_dtype` argument of mmx_attn_capture_fwd_ex / _bwd_ex: run the attention products on the bf16 matrix
 * cores (v_mfma_f32_16x16x32_bf16: operands rounded to bf16 as they are read, fp32 accumulation, softmax / dS arithmetic in
 * fp32) instead of the exact-fp32 MFMA.  BASELINE config 5 (a bf16 CLIP body, CLIP/clip/model.py:381-402). */
#define MMX_ATTN_MMA_BF16 0x100
/* Backward only: `do_dev` holds bf16 and dq / dk / dv are written as bf16 (the gradient stream between the bf16 GEMMs of a
 * bf16 body needs no conversion passes); strides stay in elements, 16-byte aligned.  With MMX_ATTN_MMA_BF16: any shape the
 * streaming kernels serve.  WITHOUT it (round 4): the exact-fp32 arithmetic of the whole-head kernels around bf16 gradient
 * I/O -- fp32 slabs, Nk <= 128, Nq <= 256, head_dim % 4 == 0 and <= 64, 8-byte aligned gradient rows; MMX_ENOTSUP otherwise.
 * No entry point reads or writes outside the buffers it is handed (16-bit slabs: the vector loads of the streaming kernels
 * fall back to element loads for the last few elements of a slab). */
#define MMX_ATTN_IO_BF16 0x200

enum mmx_status {
    MMX_OK = 0,
    MMX_EINVAL = -22,       /* bad argument (null pointer, non-positive size, unsupported dtype) */
    MMX_ENOTSUP = -95,      /* shape outside what the kernels support (message says which limit) */
    MMX_EWORKSPACE = -105,  /* workspace too small */
    MMX_EHIP = -1000        /* -1000 - hipError_t */
};

/* flags for mmx_mm_attention_rules / chain schedules */
#define MMX_MM_NORMALIZE 1u        /* apply handle_residual to R_ss / R_qq (apply_normalization=True) */
#define MMX_MM_SELF_IN_RULE10 2u   /* apply_self_in_rule_10=True; if clear the addition is cam_sq itself */
#define MMX_MM_NAN_TO_ZERO 4u      /* DETR: R_sq_addition[isnan] = 0 (DETR/modules/ExplanationGenerator.py:42) */

/* scale placement of the attention-capture op */
#define MMX_SCALE_Q_FIRST 0   /* q*scale, then q.k^T  (CLIP/clip/auxilary.py:153, DETR/modules/layers.py:745) */
#define MMX_SCALE_SCORES 1    /* (q.k^T)/sqrt(d)      (lxmert_lrp.py:399, BERT_ours.py:325) */

int mmx_abi_version(void);
const char* mmx_last_error(void);
/* tuning knobs (process-wide; results stay within the parity tolerance for every setting):
 *   "self_chain_algo"    0 auto: one workgroup per (sample, layer group); with more than one group the last arriver of a sample
 *                        multiplies the group products (re-associated at the group boundaries).  fp32 slabs run the kernels with
 *                        barrier-free stream waves -- csrc/relevancy_chain_groups.hip (several groups: every A_bar of the group
 *                        resident in LDS) / csrc/relevancy_chain_cols.hip (one group, N >= 40: ring of A_bar images) --, everything
 *                        else the fused kernel of relevancy_kernels.hip | 1: the fused kernel everywhere (same bits) |
 *                        4: same as 0 | 5: relevancy_chain_cols.hip for every
 *                        fp32 shape (strict layer order, bit-identical to "self_chain_groups" = 1)
 *   "self_chain_cols_c" / "self_chain_cols_nb"   0 auto (1 / as many as fit, <= 6) | workgroups per sample that split the COLUMNS of R
 *                        (each reduces the full A_bar; measured slower than layer groups, profiles/r05_chain_cols_probe.txt) / LDS
 *                        images of A_bar in the ring, of relevancy_chain_cols.hip
 *   "self_chain_pipe"    4 (default) fused chain kernel (algo 1 / one group / 16-bit slabs), fp32 slabs, N >= 40: the stream waves run a software pipeline of raw buffer loads (the
 *                        next batch of 8 16-byte loads in flight across the head reduction, the LDS write and the per-layer barrier)
 *                        with up to 4 KB contiguous per (head, array) and wave | 2 / 1: at most 2 / 1 KB contiguous |
 *                        0 the plain chunk loop of rounds 1-2.  Same arithmetic, bit-identical results
 *   "self_chain_nt"      1 (default) the pipelined stream waves load the read-once slabs with the nt (streaming) cache policy (a
 *                        probability slab shared by the batch keeps the default policy) | 0: default policy everywhere.  Same results
 *   "self_chain_groups"  0 auto (the fewest groups that put a workgroup on ~70 % of the CUs, at most 4, never more workgroups than
 *                        CUs; 1 below 1 MB per sample) | 1..8 layer groups per sample (1 = strict sequential order)
 *   "self_chain_rows"    0 (default) N > 128: avg_heads_kernel + the tiled product, two launches per layer with A_bar through memory |
 *                        1: no second right-hand side: a layer of the chain is ONE launch -- the head reduction of a 16-row block into
 *                        LDS, then that block row of A_bar . R on the exact-fp32 MFMA (csrc/relevancy_chain_rows.hip; same results to
 *                        summation order, measured 0.87x the speed of the default at 577 tokens: profiles/r06_chain_rows_probe.txt)
 *   "attn_head"          1 (default) register-resident whole-head attention kernels (Nk <= 128, Nq <= 256) | 0 never
 *   "attn_head_tile_skip" 1 (default) the whole-head kernels skip the products of 16-key tiles whose probabilities are exact zeros for a
 *                        whole 16-row strip (causal / padding masks; same bits, dP stays dense) | 0 never (A / B runs)
 *   "attn_stream"        1 (default) long-sequence streaming attention kernels | 0 only the general tiled kernels (any head_dim, any
 *                        alignment: what every shape the other families turn down runs on)
 *   "attn_bf16_v3"       non-zero (default): third-generation bf16 backward of the shared-forward row-relevancy mode | 0: second generation
 *   "attn_bf16_v2"       1 (default) second-generation bf16 backward | 0: the streaming kernels' bf16 path
 *   "attn_fwd_split"     1 (default) streaming forward on a small grid (< 160 workgroups of 64 rows, fp32 slabs):
 *                        16-row workgroups whose waves split the keys | 0 always the 64-row kernel
 *   "debug_flags"        profiling only (phase skipping), 0 in production; one meaning per bit for every chain kernel the dispatcher
 *                        may pick: 1 return before the hand-off / combine | 4 matrix waves skip the MFMAs | 8 layer-group kernel:
 *                        ticket without combine | 16 column kernel: no block rotation
 * These are A / B switches for tests and probes (every remaining value is the default for some shape or the reference arm of an
 * equivalence test); round 5 removed the kernel families that were nobody's default (self_chain_algo 2, self_chain_big, attn_small,
 * linear_stream, bmm_tile, the 4-wave / fourth-generation bf16 variants, the one-workgroup bi-modal schedule).
 * Unknown keys / out-of-range values return MMX_EINVAL. */
int mmx_set_option(const char* key, int value);

/* ---------------------------------------------------------------------------------------------
 * Rule 5: A_bar[b] = mean_h( clamp(grad[b,h] * attn[b,h], min=0) )              out: [B, Nq, Nk] fp32
 * replaces avg_heads (DETR/modules/ExplanationGenerator.py:19-24, lxmert/.../ExplanationGenerator.py:18-23,
 * ViT notebook cell 7:2-7) and the inline reshape/mul/clamp/mean of CLIP_explainability.ipynb cell 6:26-31.
 * Unbatched callers pass B=1 and H = (product of all leading dims).
 */
int mmx_avg_heads(const void* attn_dev, const void* grad_dev, void* out_dev,
                  int B, int H, int Nq, int Nk, int dtype, void* stream);
/* Same with an explicit batch stride (in elements) for the attention slab: H*Nq*Nk (or -1) = per-sample slabs, 0 = ONE
 * forward pass shared by all B samples of `grad` (shared-forward mode, see mmx_relevancy_self_chain_ex). */
int mmx_avg_heads_ex(const void* attn_dev, const void* grad_dev, void* out_dev,
                     int B, int H, int Nq, int Nk, int dtype, int64_t attn_batch_stride, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The fused self-attention chain (rules 5+6 [+7]) -- ONE launch for all layers:
 *     R <- R_init (identity if NULL);  for l in 0..n_layers-1:  R <- R + A_bar_l . R
 *     optional second right-hand side (rule 7): R_sq <- R_sq + A_bar_l . R_sq      (R_sq: [B, N, M])
 * replaces the per-layer loops of CLIP_explainability.ipynb cell 6:22-32 / 6:45-55, CLIP/example.py:22-30,
 * ViT notebook cell 7:28-33, VisualBERT/.../ExplanationGenerator.py:86-93,
 * DETR/modules/ExplanationGenerator.py:110-118 (encoder) -- `start_layer` is applied by the caller by
 * passing the pointer sub-range.
 *   attn_layers/grad_layers: HOST arrays of n_layers device pointers, each [B, H, N, N] `dtype`
 *   R_init_dev: NULL or [B, N, N] fp32;  R_out_dev: [B, N, N] fp32
 *   Rsq_init_dev/Rsq_out_dev: NULL or [B, N, M] fp32 (M = 0 when unused)
 *   workspace_dev: NULL allowed when mmx_self_chain_workspace_bytes() == 0 for the shape (fused path)
 */
size_t mmx_self_chain_workspace_bytes(int n_layers, int B, int H, int N, int M, int dtype);
int mmx_relevancy_self_chain(const void* const* attn_layers, const void* const* grad_layers, int n_layers,
                             int B, int H, int N, int dtype,
                             const void* R_init_dev, void* R_out_dev,
                             const void* Rsq_init_dev, void* Rsq_out_dev, int M,
                             void* workspace_dev, size_t workspace_bytes, void* stream);

/* Same, with an explicit batch stride (in elements) of the ATTENTION slabs: H*N*N (or -1) for per-sample slabs, 0 when
 * one forward pass is shared by the whole batch (the reference's CLIP `interpret` repeats ONE image B times,
 * CLIP_explainability.ipynb cell 6:3, so the image tower's probabilities are identical for every sample while the
 * gradients differ).  Gradient slabs are always per sample. */
int mmx_relevancy_self_chain_ex(const void* const* attn_layers, const void* const* grad_layers, int n_layers,
                                int B, int H, int N, int dtype, int64_t attn_batch_stride,
                                const void* R_init_dev, void* R_out_dev,
                                const void* Rsq_init_dev, void* Rsq_out_dev, int M,
                                void* workspace_dev, size_t workspace_bytes, void* stream);

/* Same, with flags (ABI 2, added in round 6).
 *   MMX_CHAIN_CAUSAL  the probability slabs come out of CAUSALLY MASKED attention (CLIP's text tower: CLIP/clip/model.py:334-340
 *                     `build_attention_mask`, applied in CLIP/clip/auxilary.py:232-236): every entry above the diagonal is an exact 0,
 *                     hence clamp(grad * attn, 0) is 0 there for any finite gradient, and the fp32 chain kernels do not READ the
 *                     4-element chunks that lie entirely above the diagonal (of either slab) -- about half the bytes of the launch.
 *                     Results are bit-identical to flags = 0 on such slabs.  The caller vouches for the mask: the kernels do not
 *                     look at the skipped entries (a non-finite gradient above the diagonal, which the reference would turn into a
 *                     NaN, goes unseen).  Ignored by the paths that have no use for it (16-bit slabs, N > 128, a second right-hand side). */
#define MMX_CHAIN_CAUSAL 1u
int mmx_relevancy_self_chain_flags(const void* const* attn_layers, const void* const* grad_layers, int n_layers,
                                   int B, int H, int N, int dtype, int64_t attn_batch_stride,
                                   const void* R_init_dev, void* R_out_dev,
                                   const void* Rsq_init_dev, void* Rsq_out_dev, int M, unsigned flags,
                                   void* workspace_dev, size_t workspace_bytes, void* stream);

/* The same chain in the reference's HALF-PRECISION mode.  On a GPU the reference runs the CLIP model after
 * `convert_weights` (CLIP/clip/model.py:381-402, 440) and creates R in the dtype of the fp16 attention probabilities
 * (CLIP_explainability.ipynb cell 6:20,43), so `grad * cam`, `.clamp(min=0).mean(dim=1)`, `torch.bmm(cam, R)` and
 * `R + ...` each produce an fp16 tensor (fp32 arithmetic inside the op, ONE rounding of its result).  This entry point
 * applies exactly those roundings (round to nearest even, overflow to inf like torch) around the fp32 sums of the kernels.
 *   slabs: fp32 or fp16 (`dtype`), R_out_dev: [B, N, N] fp32 storage holding fp16-representable values (R starts as I)
 *   workspace: mmx_self_chain_workspace_bytes(n_layers, B, H, N, 0, dtype) bytes (0 for N <= 128: one fused launch) */
int mmx_relevancy_self_chain_half(const void* const* attn_layers, const void* const* grad_layers, int n_layers,
                                  int B, int H, int N, int dtype, int64_t attn_batch_stride, void* R_out_dev,
                                  void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Batched fp32 matmul on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32):
 *     C[b] = (accumulate ? Cin[b] : 0) + op(A[b]) . B[b],   op = transpose if trans_a
 * A: [batch, M, K] (or [batch, K, M] if trans_a), B: [batch, K, N], C/Cin: [batch, M, N]; a batch
 * stride of 0 broadcasts an operand.  Cin may alias C only if it does not alias A or B.
 * replaces torch.matmul / torch.bmm in rules 6, 7, 10, 11 (DETR/.../ExplanationGenerator.py:27-43,
 * lxmert/.../ExplanationGenerator.py:26-42) and in compute_rollout_attention.
 */
int mmx_bmm_f32(const void* A_dev, const void* B_dev, const void* Cin_dev, void* C_dev,
                int batch, int M, int N, int K, int trans_a,
                int64_t stride_a, int64_t stride_b, int64_t stride_c,
                int nan_to_zero, void* stream);

/* y = x . Wt + bias on the same kernel (32 x 32 tiles at these sizes):  x [M, K], Wt [K, N] (the TRANSPOSED nn.Linear weight,
 * row-major), bias [N] or NULL, y [M, N], fp32 contiguous.  For the handful of one-sample projections of a shared forward where
 * the library's heuristic picks a 256-row tile for ~100 rows (DETR decoder, 100 queries, 256 -> 256: 30 us in the library, 9 us
 * here -- profiles/r03_detr_probe.txt); everything else stays on the library GEMMs, which are faster from 256 x 512 outputs up. */
int mmx_linear_f32(const void* x_dev, const void* wt_dev, const void* bias_dev, void* out_dev, int M, int N, int K, void* stream);

/* One row per sample <-> a dense [B, N, E] fp32 tensor (the top block of a CLIP tower carries a gradient on ONE token per sample,
 * the class / EOT token: CLIP/clip/model.py:235, 360):  mmx_rows_to_dense: out[b, n, :] = (n == rows[b]) ? vals[b, :] : 0;
 * mmx_rows_add: dense[b, rows[b], :] += vals[b, :].  rows: int64 [B] (0 <= rows[b] < N), vals [B, E], E % 4 == 0, 16-byte aligned. */
int mmx_rows_to_dense(const void* vals_dev, const void* rows_dev, void* out_dev, int B, int N, int E, void* stream);
int mmx_rows_add(void* dense_dev, const void* rows_dev, const void* vals_dev, int B, int N, int E, void* stream);

/* The chain on VECTORS (rows-only DETR rules): when a caller returns single rows of R_q_i (`aggregated[:, target_index, :]`,
 * DETR/modules/ExplanationGenerator.py:180-182) the encoder product R_ii = (I + A_6) ... (I + A_1) (`:110-118`) is needed only
 * as  R_ii . 1  (the row sums `handle_residual` divides by, `:26-31`) and as  v . R_ii : mat-vecs with the head-averaged maps.
 *   mmx_chain_matvec:  out[b] = base[b] + A[b] . y[b]      mmx_chain_vecmat:  out[b] = base[b] + x[b] . A[b]
 * A [B, N, N], vectors [B, N], fp32; `base` separate from the multiplied vector so the caller can carry the deviation from
 * the start vector (no cancellation at the end); out may not alias y / x; vecmat needs mmx_chain_vecmat_workspace_bytes. */
int mmx_chain_matvec(const void* A_dev, const void* y_dev, const void* base_dev, void* out_dev, int B, int N, void* stream);
size_t mmx_chain_vecmat_workspace_bytes(int B, int N);
int mmx_chain_vecmat(const void* A_dev, const void* x_dev, const void* base_dev, void* out_dev, int B, int N,
                     void* workspace_dev, size_t workspace_bytes, void* stream);

/* One layer of the row-vector chain without materialising A_bar (ViT nb cell 7:27-33 carried as ONE row, top layer down):
 *   out[b] = base[b] + x[b] . mean_h clamp(grad[b, h] * attn[b, h], 0),   x / base / out [B, N] fp32, slabs [B, H, N, N] of `dtype`
 * (attn_batch_stride: 0 = one forward shared by the batch, < 0 or H*N*N = per sample).  Two launches (partials per 4 slab rows, then
 * their sum in workgroup order: deterministic) instead of mmx_avg_heads + mmx_chain_vecmat's three; out may not alias x;
 * workspace 16-byte aligned, mmx_avg_heads_vecmat_workspace_bytes. */
size_t mmx_avg_heads_vecmat_workspace_bytes(int B, int N);
int mmx_avg_heads_vecmat(const void* attn_dev, const void* grad_dev, const void* x_dev, const void* base_dev, void* out_dev,
                         int B, int H, int N, int dtype, int64_t attn_batch_stride, void* workspace_dev,
                         size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * eq. 8-9: out = (R - I) / rowsum(R - I) + I  (0/0 rows -> NaN like the reference).
 * replaces handle_residual (DETR/.../ExplanationGenerator.py:46-53, lxmert/.../ExplanationGenerator.py:45-54).
 * The reference's `assert diag(R-I) >= 0` is reported through *diag_min_dev (device float, may be NULL):
 * it receives min_i (R[i,i]-1) so the host wrapper can raise AssertionError like the reference.
 */
int mmx_handle_residual(const void* R_dev, void* out_dev, int batch, int N, void* diag_min_dev, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Rules 10 (+11):  R_sq_add = Rn_ss^T . (cam_sq . Rn_qq)   [, R_ss_add = cam_sq . R_qs]
 * replaces apply_mm_attention_rules -- DETR 5-arg form (DETR/.../ExplanationGenerator.py:33-43, flags
 * MMX_MM_NAN_TO_ZERO) and LXMERT 6-arg form (lxmert/.../ExplanationGenerator.py:32-42, R_qs/R_ss_add given).
 *   R_ss [Ns,Ns], R_qq [Nq,Nq], cam_sq [Ns,Nq], R_qs [Nq,Ns] or NULL, outputs R_sq_add [Ns,Nq], R_ss_add [Ns,Ns] or NULL
 *   workspace: mmx_mm_rules_workspace_bytes(Ns, Nq) bytes
 */
size_t mmx_mm_rules_workspace_bytes(int Ns, int Nq);
int mmx_mm_attention_rules(const void* R_ss_dev, const void* R_qq_dev, const void* R_qs_dev,
                           const void* cam_sq_dev, void* R_sq_add_dev, void* R_ss_add_dev,
                           int Ns, int Nq, unsigned flags, void* diag_min_dev,
                           void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * The whole bi-modal (two-stream, LXMERT-style) schedule in ONE launch -- rules 5, 6, 7, 10, 11 and eq. 8-9:
 *   n_lang language + n_vis vision self-attention layers, then n_x cross layers (language cross, image cross, language
 *   self, image self; the last cross layer runs its language half only), finally R_tt[0,0] = 0.
 * replaces the rule schedule of GeneratorOurs.generate_ours (lxmert/lxmert/src/ExplanationGenerator.py:131-211 with the
 * helpers :18-54, 61-129).  Two phases in one launch: the rule-5 head averages of every (sample, block) are spread over the whole
 * chip (16-byte loads, write-through A_bar blocks into `workspace`), the last-arriving workgroup of a sample runs the 38 rule
 * applications from the L2-resident A_bar blocks on the exact-fp32 MFMA with every relevancy matrix in LDS; needs T, I <= 48.
 *   all tables: HOST arrays of device pointers to fp32 [B, H, Nq, Nk] slabs: lang/x_lang_self [.,.,T,T],
 *   vis/x_img_self [.,.,I,I], x_lang_cross [.,.,T,I], x_img_cross [.,.,I,T]; the image tables need n_x-1 entries.
 *   flags: MMX_MM_NORMALIZE | MMX_MM_SELF_IN_RULE10 (NaNs propagate like the reference's LXMERT variant).
 *   outputs fp32: R_tt [B,T,T], R_ti [B,T,I], optional R_ii [B,I,I], R_it [B,I,T]; diag_min_dev as in
 *   mmx_handle_residual (min over all normalisations), may be NULL.
 */
/* text_len_dev: for a batch that is PADDED to T question tokens, a device array of B ints, the real number of question tokens of
 * every sample (1..T; NULL = all T).  The slabs keep their padded [.., T, ..] shape (padded keys carry zero probability under the
 * attention mask); sample b's rules run on its leading text_len[b] tokens only, and rows / columns of R_tt / R_ti / R_it beyond
 * them are written as zero.  One padded batch replaces the grouping of samples by question length
 * (lxmert/lxmert/perturbation.py:45-83 tokenises one question per call, so the reference never pads).
 * workspace: mmx_lxmert_schedule_workspace_bytes(...) bytes, caller-owned, stream-ordered.  One kernel launch (+ one reset launch
 * for the tickets / diag word).  (Rounds 1-3 had a one-workgroup-per-sample form under this name and an `_ex` / `_v2` pair; round 5
 * keeps the one kernel that serves every shape.) */
size_t mmx_lxmert_schedule_workspace_bytes(int n_lang, int n_vis, int n_x, int B, int T, int I);
int mmx_lxmert_schedule(const void* const* lang_attn, const void* const* lang_grad, int n_lang,
                           const void* const* vis_attn, const void* const* vis_grad, int n_vis,
                           const void* const* x_lang_cross_attn, const void* const* x_lang_cross_grad,
                           const void* const* x_img_cross_attn, const void* const* x_img_cross_grad,
                           const void* const* x_lang_self_attn, const void* const* x_lang_self_grad,
                           const void* const* x_img_self_attn, const void* const* x_img_self_grad, int n_x,
                           int B, int H, int T, int I, unsigned flags, const void* text_len_dev,
                           void* R_tt_dev, void* R_ti_dev, void* R_ii_dev, void* R_it_dev,
                           void* diag_min_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Attention rollout: prod_{i >= start}( (A_i + I) [/ rowsum] ), left-multiplied.
 * replaces compute_rollout_attention (DETR/.../ExplanationGenerator.py:5-16, lxmert/...:5-15 with
 * normalize=1; VisualBERT/.../ExplanationGenerator.py:5-17 batched with normalize=0).
 *   layers: HOST array of n_layers device pointers, each [B, N, N] fp32 (caller applies start_layer)
 *   out: [B, N, N] fp32; workspace: mmx_rollout_workspace_bytes(B, N)
 */
size_t mmx_rollout_workspace_bytes(int B, int N);
int mmx_rollout_chain(const void* const* layers, int n_layers, int B, int N, int normalize,
                      void* out_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * On-device post-processing of relevancy maps (one workgroup per map, no host round trip):
 *  mmx_heatmap_bilinear_minmax: [B, g, g] patch maps -> bilinear upsample to [B, S, S] (torch interpolate semantics,
 *    align_corners=False) + min-max normalisation; replaces CLIP_explainability.ipynb cell 7:14-18 and
 *    Transformer_MM_explainability_ViT.ipynb cell 8:25-28.  g <= 64.
 *  mmx_otsu_masks: [K, n] maps -> min-max to [0,255], truncate to 8 bit, Otsu threshold (OpenCV getThreshVal_Otsu_8u),
 *    masks [K, n] fp32 in {0, 255}; thresholds_dev: optional int32 [K]; replaces DETR/mask_generator.py:116-121
 *    (D2H copy + cv2.threshold per kept query).
 */
int mmx_heatmap_bilinear_minmax(const void* in_dev, void* out_dev, int B, int g, int S, void* stream);
int mmx_otsu_masks(const void* cam_dev, void* masks_dev, void* thresholds_dev, int K, int n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Attention-capture op (replaces the save_attn / save_attn_gradients Python hooks):
 * forward : P = softmax(scale.Q.K^T + mask) written straight into the caller's capture slab, O = P.V
 * backward: dP = dO.V^T written straight into the caller's gradient slab (this IS the hooked
 *           attn gradient), dS = P*(dP - rowsum(dP*P)), dQ, dK, dV
 * replaces CLIP/clip/auxilary.py:225-252, DETR/modules/layers.py:753-762, lxmert_lrp.py:398-414,
 * BERT_ours.py:323-343 (+ the autograd of those ops).
 * Layouts (element strides, last dim contiguous): q(b,h,n,:) = q_dev + b*q_sb + h*q_sh + n*q_sn, same for
 * k, v, o and their gradients.  probs/dprobs: [B, H, Nq, Nk] contiguous, `dtype`-typed capture slabs are
 * fp32 in this ABI version.  mask_dev: NULL or additive fp32 mask, element (b,i,j) at
 * mask_dev + b*mask_sb + i*mask_sq + j (mask_sb = 0 / mask_sq = 0 broadcast).
 * `scale` is the multiplier d^-0.5 in MMX_SCALE_Q_FIRST mode and the divisor sqrt(d) in MMX_SCALE_SCORES mode
 * (the reference multiplies resp. divides; keeping the operation keeps the rounding).
 * Backward takes `probs_sb`, the batch stride of P in elements: H*Nq*Nk for a per-sample slab, 0 when ONE forward's
 * P is shared by every sample of the batch (shared-forward / batched-backward, see clip_explainability.py); q/k/v
 * batch strides may be 0 likewise.  dprobs is always per sample.
 * need_dqkv = 0 skips dS/dQ/dK/dV (lowest layer: only the captured gradient is wanted).
 */
int mmx_attn_capture_fwd(const void* q_dev, const void* k_dev, const void* v_dev,
                         int64_t q_sb, int64_t q_sh, int64_t q_sn,
                         int64_t k_sb, int64_t k_sh, int64_t k_sn,
                         int64_t v_sb, int64_t v_sh, int64_t v_sn,
                         const void* mask_dev, int64_t mask_sb, int64_t mask_sq,
                         void* probs_dev, void* o_dev, int64_t o_sb, int64_t o_sh, int64_t o_sn,
                         int B, int H, int Nq, int Nk, int D, float scale, int scale_mode, void* stream);

size_t mmx_attn_capture_bwd_workspace_bytes(int B, int H, int Nq);   /* rowsum(dP*P): [B, H, Nq] fp32 */
int mmx_attn_capture_bwd(const void* q_dev, const void* k_dev, const void* v_dev,
                         int64_t q_sb, int64_t q_sh, int64_t q_sn,
                         int64_t k_sb, int64_t k_sh, int64_t k_sn,
                         int64_t v_sb, int64_t v_sh, int64_t v_sn,
                         const void* probs_dev, int64_t probs_sb,
                         const void* do_dev, int64_t o_sb, int64_t o_sh, int64_t o_sn,
                         void* dprobs_dev,
                         void* dq_dev, void* dk_dev, void* dv_dev,
                         int64_t dq_sb, int64_t dq_sh, int64_t dq_sn,
                         int64_t dk_sb, int64_t dk_sh, int64_t dk_sn,
                         int64_t dv_sb, int64_t dv_sh, int64_t dv_sn,
                         int B, int H, int Nq, int Nk, int D, float scale, int scale_mode,
                         int need_dqkv, void* workspace_dev, size_t workspace_bytes, void* stream);

/* The same two entry points with an element type for the capture slabs (`probs`, `dprobs`): MMX_F32, MMX_F16 or
 * MMX_BF16 (the GPU notebooks of the reference run the models in fp16; BASELINE config 5 asks for bf16 capture).
 * P and dL/dP are rounded to nearest even when they are stored; the forward's O is computed from the unrounded P,
 * the backward reads the stored (rounded) P.  q/k/v/O and their gradients stay fp32.  Half-precision slabs are written
 * by the long-sequence streaming kernels only: head_dim % 4 == 0 and 16-byte aligned q/k/v views, else MMX_ENOTSUP.
 * The rule kernels (mmx_avg_heads*, mmx_relevancy_self_chain*) read all three slab types and accumulate in fp32.
 * `fwd_o_dev` (backward, optional, may be NULL): the forward's O with its (batch, head, token) strides; batch stride 0
 * when one forward is shared.  With it the streaming backward gets rowsum(dP * P) = rowsum(dO * O) from one row-wise
 * dot product instead of a first sweep over all keys (a third of its MFMA work). */
int mmx_attn_capture_fwd_ex(const void* q_dev, const void* k_dev, const void* v_dev,
                            int64_t q_sb, int64_t q_sh, int64_t q_sn,
                            int64_t k_sb, int64_t k_sh, int64_t k_sn,
                            int64_t v_sb, int64_t v_sh, int64_t v_sn,
                            const void* mask_dev, int64_t mask_sb, int64_t mask_sq,
                            void* probs_dev, int slab_dtype,
                            void* o_dev, int64_t o_sb, int64_t o_sh, int64_t o_sn,
                            int B, int H, int Nq, int Nk, int D, float scale, int scale_mode, void* stream);
int mmx_attn_capture_bwd_ex(const void* q_dev, const void* k_dev, const void* v_dev,
                            int64_t q_sb, int64_t q_sh, int64_t q_sn,
                            int64_t k_sb, int64_t k_sh, int64_t k_sn,
                            int64_t v_sb, int64_t v_sh, int64_t v_sn,
                            const void* probs_dev, int64_t probs_sb, int slab_dtype,
                            const void* do_dev, int64_t o_sb, int64_t o_sh, int64_t o_sn,
                            const void* fwd_o_dev, int64_t fo_sb, int64_t fo_sh, int64_t fo_sn,
                            void* dprobs_dev,
                            void* dq_dev, void* dk_dev, void* dv_dev,
                            int64_t dq_sb, int64_t dq_sh, int64_t dq_sn,
                            int64_t dk_sb, int64_t dk_sh, int64_t dk_sn,
                            int64_t dv_sb, int64_t dv_sh, int64_t dv_sn,
                            int B, int H, int Nq, int Nk, int D, float scale, int scale_mode,
                            int need_dqkv, void* workspace_dev, size_t workspace_bytes, void* stream);

/* Row-relevancy mode of the backward (BASELINE config 5; CLIP `interpret`, CLIP/clip/... notebook cell 7:27-37, returns
 * only `R[:, 0, 1:]` of the image tower): row 0 of  R_final = (I + A_L) ... (I + A_start)  is  e_0^T (I + A_L) ... , i.e.
 * a ROW vector carried from the top layer down -- the order the backward visits the layers anyway:
 *     rel_out[b] = rel_in[b] + rel_in[b] . A_bar_l[b],   A_bar_l = mean_h clamp(dP * P, 0).
 * The query-side kernel reduces that product from the dP / P values it already holds (deterministic: per-workgroup
 * partial rows summed in a fixed order), so for this layer dP is neither written nor re-read and no N x N A_bar or R
 * exists.  Same arguments as mmx_attn_capture_bwd_ex plus the two rows (`rel_in_dev`, `rel_out_dev`: fp32 [B, Nq],
 * may not alias); `dprobs_dev` may be NULL; `slab_dtype` must carry MMX_ATTN_MMA_BF16; Nq == Nk. */
size_t mmx_attn_capture_bwd_rowrel_workspace_bytes(int B, int H, int Nq, int Nk);
int mmx_attn_capture_bwd_rowrel(const void* q_dev, const void* k_dev, const void* v_dev,
                                int64_t q_sb, int64_t q_sh, int64_t q_sn,
                                int64_t k_sb, int64_t k_sh, int64_t k_sn,
                                int64_t v_sb, int64_t v_sh, int64_t v_sn,
                                const void* probs_dev, int64_t probs_sb, int slab_dtype,
                                const void* do_dev, int64_t o_sb, int64_t o_sh, int64_t o_sn,
                                const void* fwd_o_dev, int64_t fo_sb, int64_t fo_sh, int64_t fo_sn,
                                void* dprobs_dev,
                                void* dq_dev, void* dk_dev, void* dv_dev,
                                int64_t dq_sb, int64_t dq_sh, int64_t dq_sn,
                                int64_t dk_sb, int64_t dk_sh, int64_t dk_sn,
                                int64_t dv_sb, int64_t dv_sh, int64_t dv_sn,
                                int B, int H, int Nq, int Nk, int D, float scale, int scale_mode,
                                int need_dqkv, const void* rel_in_dev, void* rel_out_dev,
                                void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * K2-DETR: the decoder half of DETR's rule schedule for ROWS of R_q_i -- rules 5, 6, 7 and 10 with
 * eq. 8-9 and the NaN policy of DETR/modules/ExplanationGenerator.py:19-53 (rule functions), :120-140 (handle_co_attn_*) --
 * in four launches for all decoder layers.  For sample k and explained query t_k = targets[k]:
 *     s[k][:] = sum_l clean_l ? u_l . N(R_qq^(l))^T . C_l : 0,     u_l = e_t^T (I + B_L) ... (I + B_(l+1)),
 * B_l / C_l = mean_h clamp(grad * attn, 0) of decoder layer l's self- / cross-attention, R_qq^(l) = (I + B_l) ... (I + B_1),
 * N = handle_residual; clean_l = no NaN in N(R_qq^(l)) nor in C_l (the reference zeroes the NaNs of the rule-10 addition,
 * :42).  The caller finishes row t_k of R_q_i as s . N(R_ii) (mmx_chain_vecmat over the encoder maps).
 * Layer tables are HOST arrays of device pointers: self_* [K | 1, H, Q, Q], cross_* [K | 1, H, Q, Ni] fp32 contiguous;
 * *_attn_bstride = batch stride of the probability slabs in elements, 0 when ONE forward is shared by the K samples; the
 * gradient slabs are always per sample.  targets_dev: int64 [K].  s_out_dev: fp32 [K, Ni].  diag_min_dev (may be NULL):
 * min over samples and layers of diag(R_qq^(l) - I), the value handle_residual asserts to be >= 0.  Q <= 128.
 */
size_t mmx_detr_decoder_rows_workspace_bytes(int n_layers, int K, int Q, int Ni);
int mmx_detr_decoder_rows(const void* const* self_attn, const void* const* self_grad, const void* const* cross_attn,
                          const void* const* cross_grad, int n_layers, int K, int H, int Q, int Ni,
                          int64_t self_attn_bstride, int64_t cross_attn_bstride, const void* targets_dev,
                          void* s_out_dev, void* diag_min_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * LRP relevance through the attention core: the two `einsum` relprops inside the reference's
 * MultiheadAttention.relprop (DETR/modules/layers.py:770-781; einsum = RelPropSimple, layers.py:54-66, each result halved;
 * softmax / dropout relprops are the identity, layers.py:170-186), which the reference evaluates by re-running the einsums
 * and calling torch.autograd.grad.  With S = safe_divide(cam_O, O) (layers.py:11-14) and Z = (scale q).k^T:
 *   cam_P = P * (S.V^T) / 2   (-> cam_probs_dev, what the reference stores with save_attn_cam, layers.py:776)
 *   cam_V = V * (P^T.S) / 2,   S1 = safe_divide(cam_P, Z),   cam_Q = (scale q) * (S1.k) / 2,   cam_K = k * (S1^T.(scale q)) / 2
 * q / k / v / o / cam_o and the three outputs use the (batch, head, token) element strides of the capture op; probs and
 * cam_probs are [B, H, Nq, Nk] fp32 contiguous.  MMX_SCALE_Q_FIRST: q is multiplied by `scale` first (DETR, layers.py:741);
 * MMX_SCALE_SCORES: Z is the raw q.k^T product (BERT-style modules divide afterwards).  head_dim <= 64.  Two launches
 * (query side, then key side) on `stream`; no workspace.
 */
int mmx_attn_relprop(const void* q_dev, const void* k_dev, const void* v_dev, const void* o_dev, const void* cam_o_dev,
                     int64_t q_sb, int64_t q_sh, int64_t q_sn, int64_t k_sb, int64_t k_sh, int64_t k_sn,
                     int64_t v_sb, int64_t v_sh, int64_t v_sn, int64_t o_sb, int64_t o_sh, int64_t o_sn,
                     int64_t co_sb, int64_t co_sh, int64_t co_sn,
                     const void* probs_dev, void* cam_probs_dev, void* cam_q_dev, void* cam_k_dev, void* cam_v_dev,
                     int64_t cq_sb, int64_t cq_sh, int64_t cq_sn, int64_t ck_sb, int64_t ck_sh, int64_t ck_sn,
                     int64_t cv_sb, int64_t cv_sh, int64_t cv_sn,
                     int B, int H, int Nq, int Nk, int D, float scale, int scale_mode, void* stream);

/* The same core in halves (BertSelfAttention.relprop, VisualBERT/mmf/models/transformers/backends/BERT_ours.py:345-395, applies
 * Add.relprop of [scores / sqrt(d), attention_mask] -- a rule with whole-tensor sums -- between the two matmul relprops):
 *   MMX_LRP_VALUES: cam_P (-> cam_probs_dev) and cam_V from cam_O;  q / k are still read (tiling), cam_q / cam_k may be NULL.
 *   MMX_LRP_SCORES: cam_Q and cam_K from S1 = safe_divide(cam_scores, Z);  cam_scores_dev [B, H, Nq, Nk] fp32 = the relevance of
 *                   the pre-softmax scores the caller derived from cam_P;  v / o / cam_o / probs / cam_probs / cam_v may be NULL.
 * MMX_LRP_VALUES | MMX_LRP_SCORES with cam_scores_dev == NULL is mmx_attn_relprop (LxmertAttention.relprop,
 * lxmert/lxmert/src/lxmert_lrp.py:422-461: its attention_mask slot is never assigned, the Add rule is skipped). */
#define MMX_LRP_VALUES 1
#define MMX_LRP_SCORES 2
int mmx_attn_relprop_phase(const void* q_dev, const void* k_dev, const void* v_dev, const void* o_dev, const void* cam_o_dev,
                           int64_t q_sb, int64_t q_sh, int64_t q_sn, int64_t k_sb, int64_t k_sh, int64_t k_sn,
                           int64_t v_sb, int64_t v_sh, int64_t v_sn, int64_t o_sb, int64_t o_sh, int64_t o_sn,
                           int64_t co_sb, int64_t co_sh, int64_t co_sn,
                           const void* probs_dev, void* cam_probs_dev, void* cam_q_dev, void* cam_k_dev, void* cam_v_dev,
                           int64_t cq_sb, int64_t cq_sh, int64_t cq_sn, int64_t ck_sb, int64_t ck_sh, int64_t ck_sn,
                           int64_t cv_sb, int64_t cv_sh, int64_t cv_sn,
                           int B, int H, int Nq, int Nk, int D, float scale, int scale_mode,
                           const void* cam_scores_dev, int phase, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused elementwise halves of the alpha-beta LRP layer rules (csrc/lrp_kernels.hip).  All tensors
 * fp32, contiguous, device pointers; stream-ordered; deterministic two-stage sums.  workspace: mmx_lrp_workspace_bytes().
 *   mmx_lrp_split_signs    out [rows, 2n] = [clamp(x, min=0) | clamp(x, max=0)]                (Linear.relprop's px / nx,
 *                          DETR/modules/layers.py:411-414; the same split of W is cached by the caller)
 *   mmx_lrp_safe_divide    out = safe_divide(a, b)                                              (layers.py:11-14)
 *   mmx_lrp_linear_combine out [rows, n] = xx[:, :n] * y[:, :n] + xx[:, n:] * y[:, n:]           (layers.py:424-430, alpha = 1);
 *                          r_dev != NULL: then out *= safe_divide(sum(r), sum(out))             (layers.py:432, DETR flavour)
 *   mmx_lrp_add_relprop    Add.relprop (layers.py:197-222) on [batch, per] views: the three sums per batch item (batch = 1:
 *                          whole-tensor sums, the reference's one-sample pass)
 *   mmx_lrp_clone_relprop  Clone.relprop (layers.py:257-267): out = x * sum_i safe_divide(r_i, x), r_list = HOST array of
 *                          n_r <= 8 device pointers
 */
size_t mmx_lrp_workspace_bytes(void);
int mmx_lrp_split_signs(const void* x_dev, void* out_dev, int64_t rows, int n, void* stream);
int mmx_lrp_safe_divide(const void* a_dev, const void* b_dev, void* out_dev, int64_t n, void* stream);
int mmx_lrp_linear_combine(const void* xx_dev, const void* y_dev, void* out_dev, int64_t rows, int n,
                           const void* r_dev, int64_t r_numel, void* workspace_dev, void* stream);
int mmx_lrp_add_relprop(const void* r_dev, const void* a_dev, const void* b_dev, void* ra_dev, void* rb_dev,
                        int batch, int64_t per, void* workspace_dev, void* stream);
int mmx_lrp_clone_relprop(const void* const* r_list, int n_r, const void* x_dev, void* out_dev, int64_t n, void* stream);
/* MultiheadAttention.relprop's closing q / k rescale (DETR/modules/layers.py:791-799), two launches, in place on cam_k / cam_q:
 * if all_zero(cam_v after its projection rule) and not all_zero(cam_v before it), cam_k *= safe_divide(|ks| / (|ks| + |qs|) *
 * sum(cam_o), ks) and cam_q likewise (ks, qs = their sums).  n_*: element counts. */
int mmx_lrp_mha_rescale(const void* v_pre_dev, int64_t n_vpre, const void* v_post_dev, int64_t n_vpost, void* cam_k_dev,
                        int64_t n_k, void* cam_q_dev, int64_t n_q, const void* cam_o_dev, int64_t n_o,
                        void* workspace_dev, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Fused QuickGELU of the CLIP body's MLP, y = x * sigmoid(1.702 x) (CLIP/clip/model.py:162-164): one HBM pass forward,
 * one backward (dx from x and dy; nothing saved but x) instead of PyTorch's 3 + 5 elementwise kernels.
 * fp32, contiguous, 16-byte aligned, n elements.
 */
int mmx_quick_gelu_fwd(const void* x_dev, void* y_dev, int64_t n, void* stream);
/* x * sigmoid(1.702 x) with a bf16 result (n % 4 == 0): the activation of a bf16 body only feeds the next GEMM */
int mmx_quick_gelu_fwd_bf16(const void* x_dev, void* y_dev, int64_t n, void* stream);
int mmx_quick_gelu_bwd(const void* x_dev, const void* dy_dev, void* dx_dev, int64_t n, void* stream);
/* Same with x ([x_n] elements) broadcast over the leading dimension of dy / dx ([n] elements, n % x_n == 0, x_n % 4 == 0):
 * the shared-forward backward has ONE activation tensor for B upstream gradients. */
/* bf16 gradient stream (BASELINE config 5's bf16 body): dy / dx bf16, x fp32 (broadcast over the batch as above; x_n % 8 == 0);
 * LayerNorm backward + residual with a bf16 upstream gradient, written as fp32 (`dx_dev`, the next residual) and / or bf16
 * (`dx_bf16_dev`, the next GEMM's operand) -- either may be NULL. */
int mmx_quick_gelu_bwd_bcast_bf16(const void* x_dev, const void* dy_dev, void* dx_dev, int64_t n, int64_t x_n, void* stream);
int mmx_layernorm_bwd_add_bf16(const void* dy_dev, const void* x_dev, const void* mean_dev, const void* rstd_dev,
                               const void* gamma_dev, const void* d_res_dev, void* dx_dev, void* dx_bf16_dev,
                               int64_t rows, int x_rows, int E, void* stream);
int mmx_quick_gelu_bwd_bcast(const void* x_dev, const void* dy_dev, void* dx_dev, int64_t n, int64_t x_n, void* stream);

/* LayerNorm input gradient + residual add with forward statistics shared by the batch (shared-forward backward of the
 * CLIP image tower / ViT): dx[r] = d_res[r] + LN'(dy[r]; x[r % x_rows], mean, rstd, gamma).  dy, d_res (may be NULL),
 * dx: [rows, E]; x: [x_rows, E]; mean, rstd: [x_rows]; gamma: [E]; fp32 contiguous, E % 4 == 0. */
int mmx_layernorm_bwd_add(const void* dy_dev, const void* x_dev, const void* mean_dev, const void* rstd_dev,
                          const void* gamma_dev, const void* d_res_dev, void* dx_dev, int64_t rows, int x_rows, int E,
                          void* stream);

/* Residual add fused with the LayerNorm that follows it in a pre-LN block (CLIP/clip/model.py:195-197: x = x + attn(...);
 * ... ln_2(x)): sum = x + y (written to sum_dev: the next residual), h = LayerNorm(sum) * gamma + beta, and the row
 * statistics mean / rstd that mmx_layernorm_bwd_add consumes.  y_dev == NULL: plain LayerNorm of x (sum_dev unused).
 * x, y, sum, h: [rows, E]; mean, rstd: [rows]; gamma, beta: [E]; fp32 contiguous, E % 4 == 0, E <= 4096. */
int mmx_add_layernorm_fwd(const void* x_dev, const void* y_dev, const void* gamma_dev, const void* beta_dev,
                          void* sum_dev, void* h_dev, void* mean_dev, void* rstd_dev, int64_t rows, int E, float eps,
                          void* stream);
/* The same with h_dtype = MMX_F32 | MMX_BF16: a bf16 body (CLIP/clip/model.py:381-402 converts the weights to half precision) feeds
 * h to a half-precision GEMM only, so the kernel writes it as bf16 (round to nearest even) instead of a conversion pass. */
int mmx_add_layernorm_fwd_ex(const void* x_dev, const void* y_dev, const void* gamma_dev, const void* beta_dev,
                             void* sum_dev, void* h_dev, void* mean_dev, void* rstd_dev, int64_t rows, int E,
                             float eps, int h_dtype, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Kernel timing helper for bench.py: runs `fn`-independent HIP-event timing on `stream` is done in
 * Python via these thin wrappers so that events live on the SAME stream the kernels are launched on.
 */
int mmx_event_create(void** event_out);
int mmx_event_destroy(void* event);
int mmx_event_record(void* event, void* stream);
int mmx_event_elapsed_ms(void* start, void* stop, float* ms_out); /* synchronises on `stop` */

#ifdef __cplusplus
}
#endif
#endif /* MMX_RELEVANCY_H */
