/*
 * mergerec_hip.h -- C ABI of libmergerec_hip.so (MI355X / gfx950 only).
 *
 * The reference (DIALLab-SKKU/MergeRec) is pure Python and has no FFI of its own: every device op
 * on its merged-model inference path is a stock torch / transformers call made from Python.  Each
 * entry point below replaces the torch op sequence of one reference function; the citation after
 * "replaces:" is the reference file:line (relative to the upstream repo root) whose arithmetic the
 * kernel reproduces.  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - return 0 on success, a negative MR_E* code otherwise; no C++ exception crosses the ABI.
 *   - every pointer is a caller-owned DEVICE pointer (hipMalloc'd, e.g. a torch tensor's
 *     data_ptr()), contiguous, 16-byte aligned unless stated; the library never allocates or frees
 *     caller-visible memory.  Scratch is passed in (`ws`, size from the matching *_ws_bytes()).
 *   - asynchronous on `stream` (a hipStream_t passed as void*); no hidden synchronisation.
 *   - re-entrant, no global mutable state; one host thread per GPU.
 *   - "packed tokens": the encoder works on the T = sum(len_b) non-masked tokens of a batch, rows
 *     [cu_seqlens[b], cu_seqlens[b+1]) of a (T, d) fp32 matrix belong to sequence b.
 *   - "arena": all parameters of one model live in one flat fp32 buffer; each tensor starts on a
 *     64-float boundary (reference key order is kept; pads are zero).  Task vectors use the same
 *     layout, so the merge is one elementwise pass and the encoder reads weights in place.
 */
#ifndef MERGEREC_HIP_H
#define MERGEREC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MR_OK 0
#define MR_EINVAL (-1)   /* bad argument (null pointer, negative size, unsupported shape) */
#define MR_EALIGN (-2)   /* pointer or leading dimension not aligned as required */
#define MR_ELAUNCH (-3)  /* hipLaunch / runtime error (message via mr_last_hip_error) */
#define MR_EWS (-4)      /* workspace too small */
#define MR_EUNSUPPORTED (-5)

#define MR_ACT_NONE 0
#define MR_ACT_GELU_ERF 1
#define MR_ACT_TANH 2 /* mr_gemm_nt_bias_act_f32 only: the RoBERTa pooler head tanh(W h_cls + b) (encoder/_base.py:46-47, pooling_method='pooler') */
/* `products` of the split-precision entry points: 3 = two bf16 pieces per operand (hi*hi + hi*lo + lo*hi, ~2^-16 per product), 6 = three
 * bf16 pieces (six products, ~2^-24), MR_PRODUCTS_F16X3 = two FP16 pieces per operand and the same three products (~2^-21 per product at
 * the cost of bf16x3; operands must stay inside fp16's range: |x| < 65504, weights |w| < 255.9 -- see mr_split_weights_kblock_f16_f32). */
#define MR_PRODUCTS_BF16X3 3
#define MR_PRODUCTS_BF16X6 6
#define MR_PRODUCTS_F16X3 35

#define MR_EMBED_ROBERTA 0   /* LN((word + type) + pos)            -- transformers RobertaEmbeddings */
#define MR_EMBED_RECFORMER 1 /* LN(((word + pos) + type) + itempos) -- recformer/models.py:130-133   */

typedef void* mr_stream_t; /* hipStream_t */

int mr_version(void);
const char* mr_strerror(int code);
/* text of the last HIP runtime error seen by the calling thread ("" if none) */
const char* mr_last_hip_error(void);

/* ---- merger ------------------------------------------------------------------------------- */

/* tv[p] = theta[p] - base[p], p in [0, n).
 * replaces: rec_retrieval/merger/algorithms/task_vector.py:8-10 (get_task_vectors, one row). */
int mr_task_vector_f32(const float* theta, const float* base, int64_t n, float* tv, mr_stream_t stream);

/* Fixed-weight merges as ONE streaming pass, a running sum in model order with every operation rounded on its own (no FMA):
 *   base != NULL ("task_vector"): out = (...((base + w_0 (m_0 - base)) + w_1 (m_1 - base)) ...)
 *   base == NULL ("linear")     : out = (...((0 + w_0 m_0) + w_1 m_1) ...)
 * models: (N, P) with row stride `stride` (fine-tuned PARAMETERS, not task vectors); weights: N floats on the device.
 * replaces: rec_retrieval/merger/merger.py:46-93 with algorithms/task_vector.py:13-34 and algorithms/linear.py:8-27. */
int mr_merge_running_f32(const float* base, const float* models, int64_t stride, const float* weights, int N, int64_t P,
                         float* out, mr_stream_t stream);

/* out[p] = base[p] + sum_{i<N} round(alpha[s(p)*N + i] * tv[i*tv_stride + p]),  p in [p_begin, p_begin+p_count)
 * with the sum taken sequentially i = 0..N-1 from the first product and NO fused multiply-add, i.e.
 * bit-for-bit torch's `base + (alpha[:, None] * T).sum(0)` on CPU.
 * s(p) is the segment containing p: seg_off[s] <= p < seg_off[s+1] (int64, S+1 entries, every entry a
 * multiple of 4, seg_off[0] <= p_begin, seg_off[S] >= p_begin+p_count); seg_off == NULL means S == 1.
 * alpha is a DEVICE array of S*N floats (the effective coefficients gw*per+gb of each segment's group).
 * p_begin and p_count are multiples of 4 (they let one rank merge only its slice of the arena).
 * replaces: merger/weight_learning/module/task_wise.py:36-48 (S == 1) and
 *           merger/weight_learning/module/layer_wise.py:64-83 (S > 1; group table :13-33). */
int mr_merge_nway_f32(const float* base, const float* tv, int64_t tv_stride, const float* alpha,
                      const int64_t* seg_off, int N, int S, int64_t p_begin, int64_t p_count, float* out,
                      mr_stream_t stream);

/* Rows of ONE table inside the arena, merged alone: out[table_off + r*d + c] = base[..] + sum_i alpha[i] * tv[i][..] for the T rows r = idx[t]
 * (int32, 0 <= r < rows; duplicates allowed; other values are skipped), element for element the operations of mr_merge_nway_f32 in its order
 * (bit-identical rows).  alpha: the N coefficients of the segment that holds the table.  The alpha-learning step
 * (weight_learning/module/_base.py:78-81 under merge_train.py) reads only its batch's word-embedding rows of the merged model: this writes those
 * ~600 rows instead of streaming the 38.6 M-element table.  d % 4 == 0, table_off % 4 == 0. */
int mr_merge_rows_f32(const float* base, const float* tv, int64_t tv_stride, const float* alpha, int N, const int32_t* idx, int T,
                      int rows, int d, int64_t table_off, float* out, mr_stream_t stream);

/* dalpha[s*N + i] = sum_{p in segment s} tv[i*tv_stride + p] * g[p]   (the backward of the merge w.r.t.
 * the effective coefficients).  Deterministic two-stage reduction (fixed chunking, fixed order).
 * replaces: autograd backward through task_wise.py:43-47 / layer_wise.py:75-81 (merge_train.py path). */
size_t mr_merge_bwd_alpha_ws_bytes(int N, int S, int64_t P);
int mr_merge_bwd_alpha_f32(const float* tv, int64_t tv_stride, const float* g, const int64_t* seg_off, int N, int S,
                           int64_t P, float* dalpha, void* ws, size_t ws_bytes, mr_stream_t stream);

/* ---- task-vector pre-processing at init (TIES / Localize-and-Stitch) ------------------------ */

/* T = the k-th largest |x[i]| over i < n (1 <= k <= n), by exact radix select: *thr_bits = float bits of T and
 * *need_eq = how many elements with |x| == T belong to the top-k (both written to DEVICE memory; no host sync).
 * ws: mr_select_ws_bytes(n) bytes, 16-byte aligned.
 * replaces: the threshold implied by torch.topk(update.abs(), k) in merger/algorithms/ties.py:21 and
 *           merger/algorithms/localize_and_stitch.py:40. */
size_t mr_select_ws_bytes(int64_t n);
int mr_abs_kth_largest_f32(const float* x, int64_t n, int64_t k, uint32_t* thr_bits, int64_t* need_eq, void* ws,
                           size_t ws_bytes, mr_stream_t stream);

/* y[i] = x[i] if x[i] is among the k largest magnitudes (|x| > T, or |x| == T and fewer than need_eq equal elements
 * precede i: ties resolved towards LOWER indices; torch.topk leaves that choice unspecified), else 0;
 * mask_or_null[i] = 1 / 0 likewise.  thr_bits / need_eq come from mr_abs_kth_largest_f32 (device memory).
 * replaces: merger/algorithms/ties.py:21-23 (sparse_update) and localize_and_stitch.py:40-42 (masks). */
int mr_abs_topk_mask_f32(const float* x, int64_t n, const uint32_t* thr_bits, const int64_t* need_eq, float* y,
                         uint8_t* mask_or_null, void* ws, size_t ws_bytes, mr_stream_t stream);

/* In place on sparse (N rows of length P, row stride `stride`): sign election by summed positive / negative mass,
 * keep the entries that agree with the elected sign, divide by their count (0/0 -> 0); sums run sequentially over
 * the task index exactly like torch's dim-0 sum.
 * replaces: merger/algorithms/ties.py:31-72 (_compute_final_sign + disjoint mean of get_ties_vectors). */
int mr_ties_combine_f32(float* sparse, int64_t stride, int N, int64_t P, mr_stream_t stream);

/* out[i, p] = (mask[i, p] / max(sum_j mask[j, p], 1)) * tv[i, p]   (two separate roundings, as torch computes it).
 * replaces: merger/algorithms/localize_and_stitch.py:43-49. */
int mr_lns_combine_f32(const float* tv, const uint8_t* mask, int64_t stride, int N, int64_t P, float* out, mr_stream_t stream);

/* *out (device float) = the k-th largest element of x[0..n): by value (is_signed != 0) or by magnitude (the |x| value).
 * Order statistics for PCB's quantile clamps: sorted_x[j] (ascending) is the (n - j)-th largest.
 * replaces: torch.sort(...)[index] in merger/algorithms/pcb.py:15-25 (_clamp). */
int mr_kth_largest_value_f32(const float* x, int64_t n, int64_t k, int is_signed, float* out, void* ws, size_t ws_bytes,
                             mr_stream_t stream);

/* PCB, per task row `row` of tv (N rows of length P, COMPACT layout -- pads would shift the quantiles):
 * clamped = sign(tau) * clamp(|tau|, lo, hi); task_pcb = exp(N * ((clamp - lo)/(hi - lo))^2) * tanh(tau * sum_j tau_j);
 * q_lo_hi = device [lo, hi].   replaces: merger/algorithms/pcb.py:44-52. */
int mr_pcb_stage1_f32(const float* tv, int64_t stride, int N, int row, int64_t P, const float* q_lo_hi, float* clamped,
                      float* task_pcb, mr_stream_t stream);

/* out_i = clamped_i * scale_i / max(sum_j scale_j, 1e-12) / N with scale_i = (clamp(task_pcb_i, q2[2i], q2[2i+1]) - q2[2i]) /
 * (q2[2i+1] - q2[2i]).   replaces: merger/algorithms/pcb.py:54-58. */
int mr_pcb_stage2_f32(const float* clamped, const float* task_pcb, int64_t stride, int N, int64_t P, const float* q2, float* out,
                      mr_stream_t stream);

/* ---- encoder: token packing + embeddings --------------------------------------------------- */

/* From the reference's padded batch tensors (int64 (B, L), row-major) build packed per-token index
 * arrays (int32, T entries each): word id, position id = cumsum(ids != pad) * (ids != pad) + pad
 * (computed over the WHOLE row, masked positions included), token type, item position.
 * Token t of row b is the t-th position of that row with attention_mask != 0; cu_seqlens (int32,
 * B+1, device) must hold the exclusive prefix sums of the per-row mask counts.
 * token_type_ids / item_position_ids / tok_tt / tok_ip may be NULL (RoBERTa).
 * replaces: recformer/models.py:64-75 (create_position_ids_from_input_ids; same formula in
 *           transformers RobertaEmbeddings) and the padding contract of
 *           datamodule/collator/recommender/recommender.py:27-32, utils/recformer_utils.py:71-113. */
int mr_pack_tokens(const int64_t* input_ids, const int64_t* attention_mask, const int64_t* token_type_ids,
                   const int64_t* item_position_ids, int B, int L, int pad_id, const int32_t* cu_seqlens,
                   int32_t* tok_word, int32_t* tok_pos, int32_t* tok_tt, int32_t* tok_ip, mr_stream_t stream);

/* mr_pack_tokens with the input checks of the host layer folded in (no device -> host sync per batch): violations are OR-ed into
 * *err_bits (int32, device, caller-zeroed) as MR_IN_* bits and read back by the caller at its next synchronisation point.
 * Checked: ids in [0, vocab); position 0 attended (CLS pooling reads it, encoder/_base.py:45); token_type_ids / item_position_ids
 * of attended tokens inside their tables; global_attention_mask (may be NULL) == 1 at position 0 and 0 elsewhere (the only
 * pattern the reference's collators emit, utils/recformer_utils.py:51,59); per-row mask count == cu_seqlens span.
 * err_bits == NULL: no checks (== mr_pack_tokens).  With vocab / n_type / n_ip > 0 the indices WRITTEN to tok_word / tok_tt / tok_ip are
 * clamped into their tables (the flag reports the original value): every consumer -- the embedding gather, the training graph's row
 * gathers and scatter-adds, mr_merge_rows_f32 -- then addresses a real row, so bad ids never fault.
 * replaces: the index errors torch raises inside nn.Embedding for the same inputs (recformer/models.py:121-131). */
#define MR_IN_BAD_ID 1
#define MR_IN_NO_CLS 2
#define MR_IN_BAD_TOKEN_TYPE 4
#define MR_IN_BAD_ITEM_POS 8
#define MR_IN_GLOBAL_PATTERN 16
#define MR_IN_LEN_MISMATCH 32
int mr_pack_tokens_checked(const int64_t* input_ids, const int64_t* attention_mask, const int64_t* token_type_ids,
                           const int64_t* item_position_ids, const int64_t* global_attention_mask, int B, int L, int pad_id,
                           int vocab, int n_type, int n_ip, const int32_t* cu_seqlens, int32_t* tok_word, int32_t* tok_pos,
                           int32_t* tok_tt, int32_t* tok_ip, int32_t* err_bits, mr_stream_t stream);

/* out[t, :] = LayerNorm(sum of gathered rows) * gamma + beta, one wavefront per token.
 * mode MR_EMBED_ROBERTA:   (word[tok_word] + type[tok_tt or 0]) + pos[tok_pos]
 * mode MR_EMBED_RECFORMER: ((word[tok_word] + pos[tok_pos]) + type[tok_tt]) + itempos[tok_ip]
 * d % 4 == 0, d <= 2048.  n_word / n_pos / n_type / n_ip are the tables' row counts: indices are clamped
 * into range on the device so a corrupt id can never fault the GPU (callers validate ids up front).
 * replaces: transformers RobertaEmbeddings.forward (reached from module/models/encoder/_base.py:37) and
 *           recformer/models.py:104-136 (RecformerEmbeddings.forward). */
int mr_embed_gather_ln_f32(const int32_t* tok_word, const int32_t* tok_pos, const int32_t* tok_tt,
                           const int32_t* tok_ip, const float* word, const float* pos, const float* type,
                           const float* itempos, int n_word, int n_pos, int n_type, int n_ip, const float* gamma,
                           const float* beta, float eps, int T, int d, int mode, float* out, mr_stream_t stream);

/* ---- encoder: dense layers ------------------------------------------------------------------ */

/* C[m, s*seg_n + n] = act( sum_k A[m,k] * W_s[n,k] + bias_s[n] ) (+ R[m, s*seg_n + n] if R != NULL)
 * for m < M, s < nseg (<= 3), n < seg_n.  A is (M, K) with leading dimension lda, every W_s is
 * (seg_n, K) row-major contiguous (torch nn.Linear layout), bias_s may be NULL.  The k-sum of each
 * output element is ONE fp32 fused-multiply-add chain in ascending k starting from 0 (exactly what
 * v_mfma_f32_32x32x2_f32 computes), so results are bit-identical to oracle/oracle_c.c gemm_nt_ref.
 * K % 16 == 0; lda % 4 == 0 (ldc, ldr unrestricted); nseg > 1 requires seg_n % 128 == 0.
 * replaces: torch.nn.functional.linear inside transformers RobertaSelfAttention / RobertaSelfOutput /
 *           RobertaIntermediate / RobertaOutput and the Longformer equivalents (reached from
 *           module/models/encoder/_base.py:37 and recformer/models.py:340-348); with bias == NULL it is
 *           `user_encoding @ self.item_embeddings.T` of module/recommender/module.py:137. */
int mr_gemm_nt_bias_act_f32(const float* A, int64_t lda, const float* w0, const float* w1, const float* w2,
                            const float* b0, const float* b1, const float* b2, int nseg, int M, int seg_n, int K,
                            int act, const float* R, int64_t ldr, float* C, int64_t ldc, mr_stream_t stream);

/* Split-precision variant of mr_gemm_nt_bias_act_f32 on the bf16 matrix cores (same math, fp32-grade accuracy,
 * 2.67x fewer matrix-pipe cycles): every fp32 value x is x = hi + mid + lo with three bf16 pieces and a*b is the
 * six products hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid accumulated in fp32 (relative error ~2^-24 per
 * product; NOT bit-identical to the fp32 FMA chain -- the scoring GEMM keeps the exact kernel).
 * Weights come pre-split in K-BLOCKED form (mr_split_weights_kblock_f32 after each merge): w_hi / w_mid / w_lo are
 * bf16 arenas in which the (seg_n, K) matrix of segment s occupies the same element range [off_s, off_s + seg_n*K)
 * as in the fp32 arena, but element (n, k) sits at off_s + ((k / 16) * seg_n + n) * 16 + k % 16 -- the 128 x 16
 * tile a workgroup needs per k-step is then one contiguous 4 KB chunk per piece.  off_s % 8 == 0.
 * Activations A (fp32, row-major) are split on the fly.
 * products: 6 (the decomposition above) or 3 (two pieces per operand: hi*hi + hi*lo + lo*hi, ~2^-16 per product,
 * half the matrix work; w_lo is not read).
 * replaces: the same torch.nn.functional.linear calls as mr_gemm_nt_bias_act_f32. */
int mr_gemm_nt_bf16x6_f32(const float* A, int64_t lda, const uint16_t* w_hi, const uint16_t* w_mid,
                          const uint16_t* w_lo, int64_t off0, int64_t off1, int64_t off2, const float* b0,
                          const float* b1, const float* b2, int nseg, int M, int seg_n, int K, int act, const float* R,
                          int64_t ldr, float* C, int64_t ldc, int products, mr_stream_t stream);

/* hi[i] = bf16(x[i]); mid[i] = bf16(x[i] - hi[i]); lo[i] = bf16(x[i] - hi[i] - mid[i])  (round-to-nearest-even;
 * the subtractions are exact in fp32).  n % 4 == 0.  Run once per merge over the parameter arena. */
int mr_split_bf16x3_f32(const float* x, int64_t n, uint16_t* hi, uint16_t* mid, uint16_t* lo, mr_stream_t stream);

/* Split every weight matrix listed in `table` (device int64, 3 per matrix: arena offset, N, K; K % 16 == 0, offset % 8
 * == 0) from the fp32 arena into the three bf16 piece arenas in the k-blocked layout mr_gemm_nt_bf16x6_f32 reads.
 * unit_prefix (device int64, n_mat + 1) holds the exclusive prefix sums of N*K/4; total_units = unit_prefix[n_mat].
 * lo may be NULL: only the hi / mid pieces are written (all the "products = 3" GEMMs read). */
int mr_split_weights_kblock_f32(const float* arena, const int64_t* table, const int64_t* unit_prefix, int n_mat,
                                int64_t total_units, uint16_t* hi, uint16_t* mid, uint16_t* lo, mr_stream_t stream);
/* The FP16 pieces of the same matrices for products = MR_PRODUCTS_F16X3: hi = fp16(256 w), lo = fp16(256 w - hi) in the same k-blocked
 * layout (the 2^8 scale is exact, keeps the low piece of ordinary 0.01..0.1 weights a normal fp16 number, and is undone in the GEMM
 * epilogue).  overflow (device int32, may be NULL; set to 1, never cleared): a scaled weight left fp16's range (|w| >= 255.9) or is NaN --
 * the caller must then use the bf16 pieces.  Same reference lines as mr_split_weights_kblock_f32 (no counterpart upstream: the pieces are
 * a derived form of the merged parameters of merger/weight_learning/utils.py:29-40). */
int mr_split_weights_kblock_f16_f32(const float* arena, const int64_t* table, const int64_t* unit_prefix, int n_mat, int64_t total_units,
                                    uint16_t* hi, uint16_t* lo, int32_t* overflow, mr_stream_t stream);

/* out[t,:] = LayerNorm(x[t,:]) * gamma + beta   (x already holds dense(...) + residual).
 * replaces: the LayerNorm of transformers RobertaSelfOutput / RobertaOutput (post-LN blocks). */
int mr_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps, int T, int d,
                     float* out, int64_t ldo, mr_stream_t stream);

/* ---- encoder: attention --------------------------------------------------------------------- */

/* Multi-head self-attention over packed tokens.  qkv is (T, 3*H*dh): [q | k | v] per token, head h at
 * columns h*dh.  ctx is (T, H*dh).  dh == 64.  Keys of a sequence are its own tokens only.
 * seq_order (int32, B entries, may be NULL): a permutation of the sequence ids, heaviest first -- scheduling hint only.
 * window < 0  : full attention   softmax(q k^T * scale) v                     (RoBERTa / BLaIR)
 * window >= 0 : Longformer local attention with the first token of every sequence global:
 *               query i >= 1 sees key j iff j == 0 or |i - j| <= window; row 0 is NOT written
 *               (it is produced by mr_attn_global_row_f32 from the *_global projections).
 * replaces: transformers RobertaSelfAttention.forward (eager/sdpa) and LongformerSelfAttention.forward
 *           (sliding_chunks + global key column), reached from recformer/models.py:340-348. */
int mr_attn_f32(const float* qkv, const int32_t* cu_seqlens, const int32_t* seq_order, int B, int H, int dh, int max_len,
                float scale, int window, float* ctx, mr_stream_t stream);

/* mr_attn_f32 with both products (S = Q K^T and O = P V) evaluated in split precision on the bf16 matrix cores:
 * products = 3 (two bf16 pieces per operand) or 6 (three pieces, fp32-grade), fp32 accumulation and an fp32 online softmax.
 * Same arguments, masking rules and output as mr_attn_f32. */
int mr_attn_split_f32(const float* qkv, const int32_t* cu_seqlens, const int32_t* seq_order, int B, int H, int dh, int max_len,
                      float scale, int window, int products, float* ctx, mr_stream_t stream);

/* Work-list form of mr_attn_split_f32 (same arguments otherwise, same masking rules, bit-identical output): the grid is a list of the
 * (sequence, query block) pairs that exist instead of a (max_len / 128, H, B) box, so ragged batches launch no empty workgroups; a
 * workgroup covers q_rows = mr_attn_split_q_rows(window, products) query rows (256 for full attention with products = 3: every wave owns
 * two 32-row query tiles and K / V are staged once per 256 queries; 128 otherwise).
 *   work (device int32, n_slots * 8 entries): entry [slot * 8 + x] = sequence | (query_block << 24), or -1 (padding).  Workgroup id
 *   8 * (slot * H + head) + x reads entry (slot, x): the hardware deals workgroup ids to the 8 XCDs round-robin, so column x is XCD x's
 *   queue and the query blocks of one (sequence, head) -- which read the same K / V rows -- share one L2.
 * mr_attn_work_plan (HOST memory in and out, no GPU work) builds that list from the B sequence lengths: sequences by decreasing length,
 * dealt over the 8 queues in snake order.  It returns n_slots; with work == NULL or capacity < n_slots * 8 nothing is written (size query).
 * An entry holds a BLOCK index, so a list only means something for the block height it was planned with: every work-list entry point takes
 * the (B, q_rows) the list was planned for and returns MR_EINVAL unless q_rows is its kernel's block height (mr_attn_split_q_rows(window,
 * products) here, 128 for mr_attn_work_f32 / mr_attn_bwd_work_f32); entries naming a sequence >= B are skipped by the kernel (cu_seqlens
 * has B + 1 entries and is never read past them).
 * replaces: the same reference code as mr_attn_f32 (transformers RobertaSelfAttention / LongformerSelfAttention). */
int mr_attn_split_q_rows(int window, int products);
/* mr_attn_f32 (exact fp32) on the same kind of work list (q_rows = 128); drop_p > 0: the training-graph dropout of mr_attn_train_f32. */
int mr_attn_work_f32(const float* qkv, const int32_t* cu_seqlens, const int32_t* work, int64_t n_slots, int B, int q_rows, int H, int dh, float scale,
                     int window, float drop_p, uint32_t drop_key, float* ctx, mr_stream_t stream);
int64_t mr_attn_work_plan(const int64_t* lens_host, int B, int q_rows, int32_t* work_host, int64_t capacity);
int mr_attn_split_work_f32(const float* qkv, const int32_t* cu_seqlens, const int32_t* work, int64_t n_slots, int B, int q_rows, int H, int dh,
                           float scale, int window, int products, float* ctx, mr_stream_t stream);

/* Global-token row of Longformer attention: for each sequence b, ctx[cu[b], :] =
 * softmax(qg_b kg^T * scale) vg over all tokens of b, where qg is (B, H*dh) (the global query of
 * each sequence's first token) and kvg is (T, 2*H*dh) = [k_global | v_global] per token.
 * compact != 0: ctx is (B, H*dh) and row b is written instead.  The same kernel serves the LAST layer of both encoder kinds (only
 * [:, 0] is pooled, encoder/_base.py:45): qg = the query projection of the CLS rows, kvg = [k | v] of all tokens.
 * replaces: LongformerSelfAttention._compute_global_attn_output_from_hidden. */
int mr_attn_global_row_f32(const float* qg, const float* kvg, const int32_t* cu_seqlens, int B, int H, int dh, int max_len, float scale,
                           float* ctx, int compact, mr_stream_t stream);

/* out[b,:] = x[cu_seqlens[b], :]  (CLS pooling), then if normalize: out / max(||out||_2, 1e-12).
 * cu_seqlens == NULL: out[b,:] = x[b,:] (rows already one per sequence: the CLS-only last layer).
 * replaces: module/models/encoder/_base.py:44-45 (pool 'cls') + module/recommender/module.py:74-77. */
int mr_cls_pool_normalize_f32(const float* x, int64_t ldx, const int32_t* cu_seqlens, int B, int d, int normalize,
                              float* out, mr_stream_t stream);

/* pooling_method = "mean" (encoder/_base.py:42-43): out[b] = mean over the PADDED batch width of the last hidden state, pad positions
 * included, as ``outputs.last_hidden_state.mean(dim=1)`` computes it on the reference's padded (B, L, d) tensor:
 * (sum of sequence b's packed token rows + (pad_len[b] - len_b) * xpad[b]) / pad_len[b], optionally L2-normalised (module.py:74-77).
 * xpad (B, d): the hidden state shared by every pad position of sequence b (a pad token attends to the sequence's valid keys and is no key
 * itself, so one extra query row per sequence reproduces them); pad_len (int32, B): the padded width of the batch the sequence came in. */
int mr_mean_pool_f32(const float* x, int64_t ldx, const int32_t* cu_seqlens, const float* xpad, const int32_t* pad_len, int B, int d,
                     int normalize, float* out, mr_stream_t stream);

/* Gather rows x[row_idx[i], :] -> out[i, :] (the last layer's dense blocks run on CLS rows only; teacher rows
 * S_ds[sequence_id] of module/distiller/sequence/module.py:66).  16-byte vectors when d, ldx, ldo are multiples of 4 and the
 * pointers 16-byte aligned, one dword per lane otherwise. */
int mr_gather_rows_f32(const float* x, int64_t ldx, const int32_t* row_idx, int n, int d, float* out, int64_t ldo,
                       mr_stream_t stream);

/* ---- training graph: dropout ------------------------------------------------------------------ */

/* The reference optimises alpha (merge_train.py:178-196) and fine-tunes under lightning.Trainer.fit, i.e. in train() mode with HF's
 * hidden_dropout_prob = attention_probs_dropout_prob = 0.1 and RecformerEmbeddings.dropout (recformer/models.py:93,135).  torch's Philox
 * stream cannot be reproduced outside torch, so the mask here is a documented pure function (csrc/dropout.h, restated in
 * oracle/ref_cpu.py dropout_keep):   keep = lowbias32(row * 0x9E3779B1 + col * 0x85EBCA77 + key) >= floor(p * 2^32),  y = keep ? x / (1 - p) : 0,
 * recomputed in the backward kernels (nothing is stored).  key = mr_dropout_site_key(seed, step, layer, site); sites: 0 embedding
 * LayerNorm output, 1 attention probabilities (row = query token * H + head, col = key position), 2 attention-output dense, 3 FFN-output
 * dense (row = packed token, col = feature), 4 Longformer global row (row = sequence * H + head, col = key position).
 * drop_p = 0 in any *_train_* entry point is the plain kernel, bit for bit; inference never calls these. */
int mr_dropout_site_key(uint32_t seed, uint32_t step, uint32_t layer, uint32_t site, uint32_t* key_out);
/* out[t, c] = dropout(x[t, c]) (+ residual[t, c]); in place allowed (out == x).  The backward of the same site is the same call on dY. */
int mr_dropout_rows_f32(const float* x, int64_t ldx, int T, int d, float drop_p, uint32_t drop_key, const float* residual, int64_t ldr,
                        float* out, int64_t ldo, mr_stream_t stream);
/* mr_attn_f32 / mr_attn_split_work_f32 / mr_attn_global_row_f32 and their backward kernels with dropout on the attention
 * probabilities (transformers RobertaSelfAttention / LongformerSelfAttention: nn.functional.dropout(attn_probs, p, training)). */
int mr_attn_train_f32(const float* qkv, const int32_t* cu_seqlens, const int32_t* seq_order, int B, int H, int dh, int max_len, float scale,
                      int window, float drop_p, uint32_t drop_key, float* ctx, mr_stream_t stream);
int mr_attn_split_work_train_f32(const float* qkv, const int32_t* cu_seqlens, const int32_t* work, int64_t n_slots, int B, int q_rows, int H, int dh,
                                 float scale, int window, int products, float drop_p, uint32_t drop_key, float* ctx, mr_stream_t stream);
int mr_attn_global_row_train_f32(const float* qg, const float* kvg, const int32_t* cu_seqlens, int B, int H, int dh, int max_len, float scale,
                                 float drop_p, uint32_t drop_key, float* ctx, int compact, mr_stream_t stream);
int mr_attn_bwd_train_f32(const float* qkv, const float* ctx, const float* dctx, const int32_t* cu_seqlens, const int32_t* seq_order, int B, int H,
                          int dh, int max_len, float scale, int window, float drop_p, uint32_t drop_key, float* rowstat, float* dqkv,
                          mr_stream_t stream);
/* mr_attn_bwd_train_f32 on the work-list grid of mr_attn_split_work_f32 (work from mr_attn_work_plan with q_rows = 128): same results. */
int mr_attn_bwd_work_f32(const float* qkv, const float* ctx, const float* dctx, const int32_t* cu_seqlens, const int32_t* work, int64_t n_slots, int B,
                         int q_rows, int H, int dh, float scale, int window, float drop_p, uint32_t drop_key, float* rowstat, float* dqkv,
                         mr_stream_t stream);
int mr_attn_global_row_bwd_train_f32(const float* qg, const float* kvg, const float* ctx_cls, const float* dctx_cls, const int32_t* cu_seqlens,
                                     int B, int H, int dh, float scale, float drop_p, uint32_t drop_key, float* dqg, float* dkvg,
                                     mr_stream_t stream);

/* ---- scoring + evaluator -------------------------------------------------------------------- */

/* Per row of scores (nrows, ncols; leading dimension ld): the k largest entries in canonical order
 * (score descending, index ascending among equal scores, NaN ranks above everything as in torch.topk)
 * -> top_val (nrows, k) fp32, top_idx (nrows, k) int64.  Optionally (labels != NULL):
 *   row_lse[r]   = logsumexp(scores[r,:] * inv_temp)
 *   row_lab[r]   = scores[r, labels[r]] * inv_temp          (CE loss = mean(row_lse - row_lab))
 *   label_rank[r]= position of labels[r] in top_idx[r,:] or -1
 * ncols >= k; k <= mr_topk_max_k() (1024) for rows of up to 49,152 scores (one 1024-thread workgroup per row, the row's keys in
 * registers), k <= 64 for longer rows.  `--ks` is a free flag upstream (evaluator/evaluator.py:43): the Python surface rejects
 * max(ks) > mr_topk_max_k() when the Evaluator is built.
 * replaces: evaluator/evaluator.py:43 (torch.topk), the `true in pred` / `pred.index(true)` scans of
 *           evaluator/metrics.py:51-57,79-86 and the cross-entropy of module/recommender/module.py:356. */
int mr_topk_rows_f32(const float* scores, int64_t ld, int nrows, int ncols, int k, float* top_val, int64_t* top_idx,
                     const int64_t* labels, float inv_temp, float* row_lse, float* row_lab, int32_t* label_rank,
                     mr_stream_t stream);
int mr_topk_max_k(void);

/* Full-catalog scoring + top-k in one call: scores = U E^T (U: (nU, d), E: (M, d), both row-major),
 * then mr_topk_rows_f32 semantics.  scores_out != NULL: the (nU, M) block is written there (predictions were asked for).
 * scores_out == NULL, two routes with the same indices, values, label ranks and label logits (the log-sum-exp agrees to rounding):
 *   fused   -- the block is never materialised: every workgroup scores 32 users against one part of the catalog (256 / 512 / 768 items,
 *              chosen so that the launch fills the chip) into LDS and selects the part's top-k there; a second launch merges the parts'
 *              candidate lists (k <= 64, d % 32 == 0, at most 512 parts);
 *   staged  -- the scoring GEMM into `ws`, then mr_topk_rows_f32.
 * mr_score_fused_mode(2) (default, or MR_SCORE_FUSED): fused when the block would exceed 128 MB, i.e. could not stay in the Infinity Cache
 * between its writes and its reads; 1: always fused; 0: never.  Returns the previous mode; a negative argument only queries.
 * `ws`: mr_score_topk_ws_bytes_ex bytes for the call's shape under the current mode (mr_score_topk_ws_bytes: an upper bound over d, k, modes).
 * replaces: module/recommender/module.py:133-139 + evaluator/evaluator.py:43. */
size_t mr_score_topk_ws_bytes(int64_t nU, int64_t M);
size_t mr_score_topk_ws_bytes_ex(int64_t nU, int64_t M, int d, int k);
int mr_score_fused_mode(int mode);
/* test / A-B switch of mr_merge_bwd_alpha_f32: 1 = the per-vector loop for every N (bit-identical to the single-pass kernels), 0 = default;
 * any other value only queries.  Returns the previous setting.  Initial value: MR_MERGE_BWD_GENERIC, read once when the library loads. */
int mr_merge_bwd_generic(int on);
int mr_score_topk_f32(const float* U, const float* E, int64_t nU, int64_t M, int d, int k, float* top_val,
                      int64_t* top_idx, float* scores_out, const int64_t* labels, float inv_temp, float* row_lse,
                      float* row_lab, int32_t* label_rank, void* ws, size_t ws_bytes, mr_stream_t stream);

/* ---- next-row 2: distillation losses (collaborative merging optimisation, BASELINE config 5) */

/* Per-row value (and gradient) of the reference's distillation losses over logit rows z ("merged model") and t ("single
 * model"), rows x M, one launch:
 *   loss_row[r] = w_ce CE(z, label) + w_kd T^2 KL(softmax(t/T) || softmax(z/T)) + w_ent H(softmax z; log(p + 1e-8))
 *               + w_mse mean_j (z - t)^2 + w_pair relu(margin - (z[pos] - z[neg])) + w_listnet (-sum softmax(t/T) log_softmax(z/T))
 * label_src: 0 none, 1 argmax t (teacher pseudo-label), 2 argmax z (merged pseudo-label); pos / neg = best / second best of t;
 * ties go to the lowest index.  dz (may be NULL) receives grad_scale * d loss_row / d z.  The batch loss of every reference
 * class is the mean of loss_row (CE "mean", KL "batchmean", MSE "mean", ...).  t may be NULL when no term reads it.
 * replaces: module/recommender/loss_fn.py:37-215 (the eleven DistillLossBase subclasses), called per sample by
 * module/distiller/sequence/module.py:62-72. */
int mr_distill_loss_rows_f32(const float* z, int64_t ldz, const float* t, int64_t ldt, int64_t rows, int64_t M, int label_src,
                             float w_ce, float w_kd, float temperature, float w_ent, float w_mse, float w_pair, float margin,
                             float w_listnet, float* loss_row, float* dz, int64_t lddz, float grad_scale, mr_stream_t stream);

/* ---- encoder backward (collaborative-merging optimisation loop, BLaIR / RoBERTa): d loss / d merged parameters.
 * Every matrix product of the backward pass runs on mr_gemm_nt_bias_act_f32 after an operand re-layout:
 *   dX = dY W        -> gemm_nt(dY, W^T)          dW = dY^T X -> gemm_nt(dY^T, X^T)   (token dimension zero-padded to 16)
 * replaces: torch autograd through transformers' RobertaLayer (reached via module/models/encoder/_base.py:37) in
 * module/distiller/sequence/module.py:76-79 (training_step). */

/* C = A W^T (+ bias) (+ R), one weight segment, the k range cut into `splits` slices run by separate workgroups and summed in
 * ascending slice order (deterministic; not the single ascending-k chain of mr_gemm_nt_bias_act_f32 -- training graph only).
 * ws: mr_gemm_nt_splitk_ws_bytes(M, N, splits) bytes. */
size_t mr_gemm_nt_splitk_ws_bytes(int M, int N, int splits);
int mr_gemm_nt_splitk_f32(const float* A, int64_t lda, const float* W, const float* bias, int M, int N, int K, const float* R, int64_t ldr,
                          float* C, int64_t ldc, int splits, void* ws, size_t ws_bytes, mr_stream_t stream);

/* out[c][r] = in[r][c] (r < R, c < C); columns R .. R_pad - 1 of every output row are zero-filled (ldo >= R_pad). */
int mr_transpose_f32(const float* in, int64_t ldi, int R, int C, float* out, int64_t ldo, int R_pad, mr_stream_t stream);

/* out[c] = sum_r x[r][c] (bias gradients): fixed summation order -- 8 interleaved row lanes per 256-row chunk, chunks ascending.
 * ws: mr_colsum_ws_bytes(R, C) bytes (0 when R <= 256: then ws may be NULL). */
size_t mr_colsum_ws_bytes(int R, int C);
int mr_colsum_f32(const float* x, int64_t ldx, int R, int C, float* out, void* ws, size_t ws_bytes, mr_stream_t stream);

/* out[r] = sum_c x[r][c] (r < R, c < C), fixed order: the bias gradient read off the transposed dY that the weight gradient consumes. */
int mr_rowsum_f32(const float* x, int64_t ldx, int R, int C, float* out, mr_stream_t stream);

/* h = gelu_erf(u) (the training forward keeps the pre-activation u);  du = dh * d/du gelu_erf(u). */
int mr_gelu_fwd_f32(const float* u, int64_t n, float* h, mr_stream_t stream);
int mr_gelu_bwd_f32(const float* u, const float* dh, int64_t n, float* du, mr_stream_t stream);

/* LayerNorm backward over the last dimension: dx (T, d); stats (T, 2) receives (mean, rstd) of x; dgamma / dbeta (d) may both
 * be NULL.  x is the LayerNorm INPUT.  ws: mr_layernorm_bwd_ws_bytes(T, d) bytes for the parameter gradients' row-chunk partials
 * (0 when T <= 256 or dgamma == NULL). */
size_t mr_layernorm_bwd_ws_bytes(int T, int d);
int mr_layernorm_bwd_f32(const float* x, int64_t ldx, const float* dy, int64_t ldy, const float* gamma, float eps, int T, int d,
                         float* dx, int64_t lddx, float* stats, float* dgamma, float* dbeta, void* ws, size_t ws_bytes, mr_stream_t stream);

/* mr_distill_loss_rows_f32 with a per-row length: row r holds row_M[r] logits (device int32, every entry in [1, M_max]) -- ONE launch for the
 * rows of several catalogs (module/distiller/sequence/module.py:62-72 loops the samples; a step's 16 samples are spread over the domains). */
int mr_distill_loss_rows_var_f32(const float* z, int64_t ldz, const float* t, int64_t ldt, int64_t rows, int64_t M_max, const int32_t* row_M,
                                 int label_src, float w_ce, float w_kd, float temperature, float w_ent, float w_mse, float w_pair, float margin,
                                 float w_listnet, float* loss_row, float* dz, int64_t lddz, float grad_scale, mr_stream_t stream);

/* Skinny scoring of the distillation step: out[i][m] = <reps[i], E[m]> for n <= 8 representation rows against a whole catalog (M, d),
 * d % 4 == 0, d <= 1024 -- one HBM-bound stream over E (module/distiller/sequence/module.py:66: ``rep @ item_embedding.T`` per sample;
 * the MFMA tile kernel needs >= 64 rows to pay).  mr_skinny_bwd_f32: d_reps[i][:] = scale * sum_m dz[i][m] E[m][:] (its autograd),
 * per-workgroup partial sums in ws (mr_skinny_bwd_ws_bytes) added in a fixed order. */
int mr_skinny_scores_f32(const float* reps, int64_t ldr, int n, const float* E, int64_t lde, int64_t M, int d, float* out, int64_t ldo,
                         mr_stream_t stream);
size_t mr_skinny_bwd_ws_bytes(int n, int64_t M, int d);
int mr_skinny_bwd_f32(const float* dz, int64_t lddz, int n, const float* E, int64_t lde, int64_t M, int d, float scale, float* d_reps, void* ws,
                      size_t ws_bytes, mr_stream_t stream);

/* Token-sized exact-fp32 products of the collaborative-merging step (merge_train.py; autograd through transformers' Linear layers under
 * merger/weight_learning/module/_base.py:78-81 and module/distiller/sequence/module.py:59-79): C = Aop Bop^T where either operand is read
 * in either orientation, so forward (Y = X W^T: trans_a = 0, trans_b = 0), input gradient (dX = dY W: 0, 1) and weight gradient
 * (dW = dY^T X: 1, 1; K = tokens, any K >= 1) need no transposed copy of anything, and a 64 x 32 / 64 x 64 tile whose workgroup walks the
 * whole k range needs no split-K reduction: every output element is the single ascending-k fp32 FMA chain of the C oracle, bit for bit.
 *   A: trans_a = 0 -> (M, K) row-major, lda; K % 16 == 0.   trans_a = 1 -> (K, M) row-major, lda; M % 4 == 0.
 *   B: trans_b = 0 -> (N, K) row-major, ldb; K % 16 == 0; nseg_b matrices stacked along N (seg_b % 64 == 0 columns each, bias_s each).
 *      trans_b = 1 -> (K, N) row-major, ldb; N % 4 == 0; nseg_b matrices stacked along K (seg_b % 16 == 0 rows each); bias0 over all N.
 *   C: nseg_c matrices stacked along M (seg_c % 64 == 0 rows each), ldc; colsum_s (optional) receives sum_k Aop[m][k] per C segment
 *      (trans_a = 1: the bias gradient that belongs to the weight gradient).
 *   epilogue, in this order: + bias[n]; dropout with the counter mask of csrc/dropout.h under (drop_p, drop_key), row = m, column = n;
 *   + R[m][n]; epi = MR_EPI_GELU_FWD: C = v, C2 = gelu_erf(v) (the pre-activation the backward needs AND the activation);
 *   epi = MR_EPI_GELU_BWD: C = v * gelu'(E[m][n]).   bn: 0 = choose the tile width, 32 / 64 = force it.
 *   products: 0 = exact fp32 on v_mfma_f32_16x16x4_f32 (every output element the ascending-k FMA chain, bit for bit); 6 = bf16x6 split
 *   precision (three bf16 pieces per operand split while the tile is staged, six v_mfma_f32_16x16x32_bf16 products, fp32 accumulation:
 *   ~2^-24 per product over fp32's whole range -- 1e-6-sized gradients included -- at 2.7 x fewer matrix-pipe cycles). */
#define MR_EPI_NONE 0
#define MR_EPI_GELU_FWD 1
#define MR_EPI_GELU_BWD 2
int mr_gemm_tile_f32(const float* A, int64_t lda, int trans_a, const float* b0, const float* b1, const float* b2, int64_t ldb, int trans_b,
                     int nseg_b, int seg_b, const float* bias0, const float* bias1, const float* bias2, int M, int N, int K, const float* R,
                     int64_t ldr, float* c0, float* c1, float* c2, int64_t ldc, int nseg_c, int seg_c, float* colsum0, float* colsum1,
                     float* colsum2, int epi, const float* E, int64_t lde, float* C2, int64_t ldc2, float drop_p, uint32_t drop_key, int products,
                     int bn, mr_stream_t stream);

/* Softmax self-attention backward on packed sequences: qkv (T, 3 H dh) = [Q | K | V] and ctx (T, H dh) as in mr_attn_f32,
 * dctx = d loss / d ctx; rowstat (T, H, 2) is workspace; dqkv (T, 3 H dh) receives [dQ | dK | dV].  dh must be 64.
 * window < 0: full attention; window >= 0: Longformer band |i - j| <= window plus the global key 0, query row 0 excluded (it
 * belongs to the global-row kernel).  seq_order (B sequence ids, longest first; may be NULL) sets the dispatch order.  max_len >= the longest sequence (sizes the grid: one workgroup per 128 rows, head, sequence).
 * fp32 matrix-core products (v_mfma_f32_32x32x2_f32), one owner per output element, fixed summation order. */
int mr_attn_bwd_f32(const float* qkv, const float* ctx, const float* dctx, const int32_t* cu_seqlens, const int32_t* seq_order, int B, int H,
                    int dh, int max_len, float scale, int window, float* rowstat, float* dqkv, mr_stream_t stream);

/* Longformer global row (mr_attn_global_row_f32) backward: qg (B, H dh), kvg (T, 2 H dh) = [Kg | Vg], ctx_cls / dctx_cls (B, H dh) the
 * forward output and its gradient at the CLS rows; dqg (B, H dh), dkvg (T, 2 H dh). */
int mr_attn_global_row_bwd_f32(const float* qg, const float* kvg, const float* ctx_cls, const float* dctx_cls, const int32_t* cu_seqlens,
                               int B, int H, int dh, float scale, float* dqg, float* dkvg, mr_stream_t stream);

/* table[idx[t]][:] += src[t][:] (atomic adds: embedding-table gradients; with unique indices a plain row scatter). */
int mr_scatter_add_rows_f32(const float* src, int64_t lds, const int32_t* idx, int T, int d, float* table, int64_t ldt,
                            mr_stream_t stream);

/* Token-major activation x (T, C), row stride ldx -> the hi / mid bf16 pieces of x^T (C, T_pad) in the k-blocked layout
 * [T_pad / 16][C][16] (tokens T .. T_pad - 1 zero): the pre-split operand of a weight gradient dW = dY^T x in one pass. */
int mr_split_tokens_kblock_f32(const float* x, int64_t ldx, int T, int C, int T_pad, uint16_t* hi, uint16_t* mid, mr_stream_t stream);

/* Split-K form of the bf16x3 GEMM for one pre-split, k-blocked weight (hi / mid piece arenas, element offset `off`):
 * C = A W^T (+ bias) (+ R), K cut into `splits` chunks (multiples of 32) computed by separate workgroups, partial products summed in
 * chunk order from `ws` (mr_gemm_nt_bf16x3_splitk_ws_bytes).  N % 4 == 0, K % 16 == 0.  For the fine-tuning weight gradients
 * dW = dY^T X (outputs of a few dozen tiles, K = number of tokens).
 * replaces: the weight-gradient products of torch autograd through transformers' Linear layers (encoder/_base.py:37 under
 * module/recommender/module.py:168-189). */
size_t mr_gemm_nt_bf16x3_splitk_ws_bytes(int M, int N, int splits);
int mr_gemm_nt_bf16x3_splitk_f32(const float* A, int64_t lda, const uint16_t* w_hi, const uint16_t* w_mid, int64_t off, const float* bias,
                                 int M, int N, int K, const float* R, int64_t ldr, float* C, int64_t ldc, int splits, void* ws,
                                 size_t ws_bytes, mr_stream_t stream);

/* ---- fine-tuning (finetune_train.py): the optimizer step over the parameter arena ------------- */

/* One AdamW step over a flat fp32 vector of n elements (n % 4 == 0), in place: torch.optim.AdamW's update
 *   p *= 1 - lr wd;  m += (1 - beta1)(g - m);  v = beta2 v + (1 - beta2) g g;  p -= lr / (1 - beta1^step) * m / (sqrt(v) / sqrt(1 - beta2^step) + eps)
 * with the weight decay looked up per segment (seg_off: S + 1 ascending int64 element offsets, multiples of 4, seg_off[0] = 0,
 * seg_off[S] = n; seg_wd: S floats; both NULL -> `weight_decay` everywhere) and, when grad_sumsq != NULL, the gradient first scaled by
 * min(1, max_grad_norm / (sqrt(*grad_sumsq) + 1e-6)) -- torch.nn.utils.clip_grad_norm_ with the norm read from DEVICE memory.
 * `step` counts from 1.  Hyper-parameters are doubles (python floats in the reference); the arithmetic is fp32.
 * replaces: torch.optim.AdamW over the two parameter groups of module/recommender/module.py:44-72 (configure_optimizers) and
 * Lightning's gradient_clip_val (finetune_train.py:106). */
int mr_adamw_step_f32(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, const int64_t* seg_off,
                      const float* seg_wd, int S, double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step,
                      const float* grad_sumsq, float max_grad_norm, mr_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MERGEREC_HIP_H */
