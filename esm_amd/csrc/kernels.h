// kernels.h — host-side declarations of the kernel launchers (internal to libesmk.so).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace esmk {

enum { ESMK_DT_F32 = 0, ESMK_DT_F16 = 1, ESMK_DT_BF16 = 2 };

enum {
    EPI_STORE_T = 0,    // out[M,N] operand dtype = acc + bias
    EPI_STORE_F32 = 1,  // out[M,N] fp32          = acc + bias
    EPI_GELU_T = 2,     // out operand dtype      = gelu(acc + bias)
    EPI_GELU_F32 = 3,   // out fp32               = gelu(acc + bias)
    EPI_RESID_F32 = 4,  // out fp32              += acc + bias
    EPI_QKV_ROPE = 5,   // fused q,k projection: bias, q scale, RoPE, head-major store (head_dim 64)
    EPI_V_T = 6,        // v projection stored transposed [B,H,64,Tp] for the attention kernel
    EPI_MSA_CTX = 7,    // MSA row attention context: out[((zo*R + n/64)*C + m)*ldc + zi*64 + n%64] (operand dtype)
    // q, k and v in ONE launch (gemm9 only, half-height tiles, plain or LayerNorm-fold consumer form; N = 3E, W and bias = the packed [3E] q | k | v rows, E a multiple of 128): tiles
    // of columns [0,2E) run exactly as EPI_QKV_ROPE, tiles of [2E,3E) exactly as EPI_V_T — the same instruction sequence per
    // tile as the two launches, hence the same bits; what changes is how many ROUNDS of tiles the 256 CUs need (small batches)
    EPI_QKV_ALL = 8
};

struct GemmArgs {
    const void* A = nullptr;      // [M,K] operand dtype
    const void* W = nullptr;      // [N,K] operand dtype
    const float* bias = nullptr;  // [N] or null
    void* out = nullptr;
    int M = 0, N = 0, K = 0;
    int force_generic = 0;
    int force_old = 0;  // use the one-tile-per-workgroup kernels of gemm.hip (A/B measurements, tests)
    int panel_c = 0;    // gemm8: N tiles per column panel of the tile order (0 = choose)
    int half_m = 0;     // gemm8: 128-row tiles: 0 = decide from the tile count, 1 = force, -1 = never (A/B, tests)
    int xpf_kt = -1;    // gemm8, EPI_RESID_F32: K tile after which the residual tile is prefetched into the L2 (set by the launcher)
    int dbg = 0;  // timing experiments only (tools/microbench.py): 1 no staging, 2 no barrier, 4 no LDS reads
    // EPI_QKV_ROPE only
    void* q = nullptr;   // [B,H,T,64]
    void* k = nullptr;   // [B,H,T,64]
    void* vt = nullptr;  // [B,H,64,Tp]
    const float* cos = nullptr;  // [T,32]
    const float* sin = nullptr;  // [T,32]
    int T = 0, H = 0, E = 0, Tp = 0;
    float scaling = 1.f;
    // ---- generalised addressing of the persistent kernel (gemm8.hip); 0 = dense default ------------
    long long a_row_bytes = 0, w_row_bytes = 0;  // stride between consecutive operand rows (2K)
    long long a_kt_bytes = 0, w_kt_bytes = 0;    // stride between consecutive 64-wide K tiles (128)
    int a_kt_repeat = 0;                         // 1: the A stream stays on each K tile for TWO stream positions (split
                                                 // weights, W = W_hi + W_lo stored as interleaved K tiles [hi0 lo0 hi1 ...]:
                                                 // K counts the 2 K_real columns of that image, a_row_bytes = 2 K_real)
    int batch = 1, batch_inner = 1;              // batched GEMM: z = zo * batch_inner + zi
    long long a_bo = 0, a_bi = 0, w_bo = 0, w_bi = 0, o_bo = 0, o_bi = 0;  // byte offsets per zo / zi
    int n_valid = 0;                  // W rows that exist (default N): loads of rows >= n_valid are clamped
    int ldc = 0;                      // output row stride in elements (default N)
    const float* row_keep = nullptr;  // EPI_QKV_ROPE: q row m is multiplied by row_keep[m] (axial_attention.py:85-88)
    int vt_rows = 0;                  // EPI_V_T: > 0 selects the layout [B,H,R = vt_rows,64,Tp], keys not permuted
    int rowmap_R = 0, rowmap_C = 0;   // EPI_RESID_F32: GEMM row (b,c,r) is added to output row (b,r,c)
    int ctx_R = 0, ctx_C = 0;         // EPI_MSA_CTX geometry
    int head_dim = 64;                // EPI_QKV_ROPE / EPI_V_T: 64, or 128 (two 64-column slices per head)
    const int* row_pos = nullptr;     // EPI_QKV_ROPE: rotary position of row m (token-packed batches; default m % T)
    // ---- LayerNorm fold (gemm9 only) ------------------------------------------------------------
    // Producer = EPI_RESID_F32 with ln_part != null: besides out[m][n] += acc + bias it writes the operand-dtype copy
    // h16[m][n] = T(out_new - ln_mean[m]) (row stride ldh) that the next GEMM takes as its A operand, and the partial
    // row sums ln_part[(m * ln_parts + n_base / 128) * 2 + {0, 1}] = (sum d, sum d^2) over the wave's 128 columns,
    // d = out_new - ln_mean[m] (ln_mean: the row's PREVIOUS mean — any per-row constant cancels against the centred
    // weights; it only keeps the rounding of h16 relative to the row's spread instead of its offset).
    // Consumer = EPI_QKV_ROPE / EPI_V_T / EPI_GELU_T with ln_rstd != null: W holds gamma-folded, row-centred weights,
    // the accumulators start from 0 and the epilogue computes  ln_rstd[m] * acc + (bias[n] + bias2[n])
    // (bias2 = W . beta, written when the weights were folded).
    void* h16 = nullptr;
    int ldh = 0;
    float* ln_part = nullptr;
    int ln_parts = 0;
    const float* ln_mean = nullptr;
    const float* ln_rstd = nullptr;
    const float* bias2 = nullptr;
    // Precision mode f16x3, EPI_GELU_T: the output rows leave as the A operand of that mode's fc2 — per 64-column K tile
    // hi | hi | lo, lo = T(v - T(v)), row stride 3 N; full-height gemm9 instantiation of its own
    int x3_out = 0;
    int lnf_dbg = 0;  // timing experiments on the producer (results incomplete): 1 no h16 stores, 2 no statistics, 4 no mean loads
    // gemm9, EPI_RESID_F32: start-up delay (shader cycles) of one workgroup group — takes the HBM-bound read-modify-write
    // epilogues of the two groups out of lockstep (set by launch_gemm; 0 = none).  desync_group: 0 = odd XCDs are late,
    // 1 = every other workgroup of each XCD, 2 = four phases (blockIdx & 3) x desync / 2.  Results do not depend on it.
    int desync = 0, desync_group = 0;
};

hipError_t launch_gemm(const GemmArgs& p, int epi, int operand_dtype, hipStream_t st);
// gemm8.hip: persistent ping-pong kernel (K % 64 == 0, N % 8 == 0); launch_gemm prefers it
bool gemm8_supports(const GemmArgs& p, int epi);
bool gemm8_generalised(const GemmArgs& p, int epi);  // uses fields only the persistent kernel implements
bool gemm8_half_height(const GemmArgs& p);            // dense kernels: 128 x 256 tiles for this shape?
hipError_t launch_gemm8(const GemmArgs& p, int epi, int operand_dtype, hipStream_t st);
// gemm32.hip: fp32 in / fp32 accumulate / fp32 out on the exact-fp32 MFMA path (1/16 of the fp16 rate): the LM head of
// the f16x2 precision mode.  A [M,K] row stride lda, W [N,K], out [M,N] row stride ldc; K % 32 == 0, lda % 4 == 0
hipError_t launch_gemm32(const float* A, int lda, const float* W, const float* bias, float* out, int ldc, int M, int N,
                         int K, bool gelu, hipStream_t st);
// measurement hook: per-tile s_memtime stamps of workgroup-leader lanes ([workgroup][tile & 31][4])
void gemm8_set_timing(unsigned long long* dev_buf);
// gemm9.hip: the same contract on one wave per SIMD (128 x 128 wave blocks); dense operands only.  var selects the
// DMA schedule (0: 8 + 8 pieces, 1: 6 + 5 + 5) or a timing experiment (gemm9.hip)
bool gemm9_supports(const GemmArgs& p, int epi);
bool gemm_qkv_one_launch(const GemmArgs& qk);  // q / k and v as one EPI_QKV_ALL launch: supported and fewer rounds of tiles?
bool gemm9_ln_fold(const GemmArgs& p, int epi);  // the call asks for the LayerNorm-fold form of its epilogue
hipError_t launch_gemm9(const GemmArgs& p, int epi, int operand_dtype, int var, hipStream_t st);
void gemm9_set_timing(unsigned long long* dev_buf);
// which persistent kernel launch_gemm picks for dense calls: 8 (default) or 9; ESMK_GEMM_IMPL / esmk_debug_gemm_impl
void gemm_set_impl(int impl, int var);
// tuning knobs by name (esmk_debug_set): "resid_desync" (fraction of a tile's main loop), "resid_desync_group"
bool gemm_set_knob(const char* key, double value);
void attention_set_stagger(int cycles);  // attention.hip: start-up stagger of co-resident workgroups (timing only)

// ---- elementwise.hip -------------------------------------------------------------------
// per-sequence statistics of the token matrix (esm2.py:82,86-92): scale[b] for token dropout,
// key_bias[b,t] = 0 / -inf (multihead_attention.py:368-374), seq_info[2b] = #pads (esm2.py:108-109),
// seq_info[2b+1] = 1 + index of the last non-pad token
hipError_t launch_seq_stats(const int64_t* tokens, int B, int T, int pad_idx, int mask_idx,
                            int token_dropout, float* scale, float* key_bias, int* seq_info,
                            hipStream_t st, float* keep = nullptr);  // keep[b,t] = 1 - pad (optional)
// ESM-1b / ESM-1v: x += embed_positions[...] (esm1.py:133, modules.py:240-257); x *= keep (esm1.py:138-139)
// seg != null: token-packed batch, "sequence" b = segment b = rows [seg[2b], seg[2b] + seg[2b+1]), T = longest
hipError_t launch_add_positions(const int64_t* tokens, const float* pos_emb, float* x, int B, int T, int E,
                                int pad_idx, int npos, hipStream_t st, const int* seg = nullptr);
hipError_t launch_scale_rows(float* x, const float* keep, int rows, int E, hipStream_t st);
// embedding gather + token-dropout rescale + pad zeroing (esm2.py:84-95)
hipError_t launch_embed(const int64_t* tokens, const float* table, const float* scale, float* x,
                        int B, int T, int E, int vocab, int pad_idx, int mask_idx,
                        int token_dropout, hipStream_t st);
// LayerNorm(E, eps=1e-5) (modules.py:68-81): fp32 rows -> operand-dtype and/or fp32 rows
hipError_t launch_layernorm(const float* x, const float* gamma, const float* beta, void* y,
                            float* y32, int rows, int E, int operand_dtype, hipStream_t st);
// LayerNorm fold (elementwise.hip): entry of the chain, statistics from the producers' partial sums,
// load-time weight fold
hipError_t launch_rowstats(const float* x, void* y, float* mean, float* rstd, int rows, int E, int ldy, int operand_dtype,
                           hipStream_t st);
hipError_t launch_ln_finalize(const float* part, float* mean, float* rstd, int rows, int parts, int E, hipStream_t st);
hipError_t launch_fold_weight(const void* src, int src_dtype, const float* gamma, const float* beta, void* dst, int dst_dtype,
                              float* bias2, size_t rows, size_t cols, size_t dst_ld, int row_map, int d, hipStream_t st);
// MSA Transformer variants of the same kernel: output rows scaled by row_keep[row] (padded positions
// zeroed, msa_transformer.py:171-172) and/or written in (b,c,r) row order for the column-attention block
struct LnExtra {
    const float* row_keep = nullptr;
    int map_R = 0, map_C = 0;
    int ldy = 0;  // row stride of the operand-dtype output in elements (0 = E): K-padded activation rows
    int x3 = 0;   // precision mode f16x3: y rows in the hi | hi | lo layout per 64-column K tile (ldy >= 3 E), lo = T(o - T(o))
};
hipError_t launch_layernorm_ex(const float* x, const float* gamma, const float* beta, void* y,
                               float* y32, int rows, int E, int operand_dtype, LnExtra ex,
                               hipStream_t st);
// MSA embedding (msa_transformer.py:152-165, modules.py:240-257) and padding bookkeeping
hipError_t launch_msa_embed(const int64_t* tokens, const float* tok_emb, const float* pos_emb,
                            const float* msa_pos, float* x, float* keep, float* col_fill, int* any_pad,
                            int B, int R, int C, int D, int vocab, int pad_idx, int npos, hipStream_t st);
// tied row attention softmax (axial_attention.py:96-100,127)
hipError_t launch_msa_row_softmax(const float* scores, const float* keep, const int* any_pad, void* probs,
                                  float* attn_out, int B, int H, int R, int C, int ldp, int layer,
                                  int num_layers_total, int operand_dtype, hipStream_t st, int nslice = 1);
// dtype conversion of a parameter tensor into the packed image
hipError_t launch_convert(const void* src, int src_dtype, void* dst, int dst_dtype, size_t n,
                          hipStream_t st);
// [rows, cols] -> dst with row stride dst_ld; row_map / col_map = 1 spreads head_dim-d heads over 64 slots
hipError_t launch_convert2d(const void* src, int src_dtype, void* dst, int dst_dtype, size_t rows, size_t cols,
                            size_t dst_ld, int row_map, int col_map, int d, hipStream_t st);
// split-weight image (operand mode f16x2): src [rows, cols] -> dst fp16 [rows, 2 dst_ld]; 64-column K tile t of row r
// becomes hi = fp16(w) at dst[r][128 t .. +63] and lo = fp16(w - hi) at dst[r][128 t + 64 .. +127]
// (hi + lo carries ~19-22 bits of w: the MFMA takes fp16 subnormals as they are); row_map / col_map as above
// parts = 3 (operand mode f16x3): hi | lo | hi per K tile, rows of 3 dst_ld — against activation rows hi | hi | lo
// (LnExtra::x3, GemmArgs::x3_out, attention's X3 output) a plain GEMM over K' = 3 K is A_hi W_hi + A_hi W_lo + A_lo W_hi
hipError_t launch_convert2d_split(const void* src, int src_dtype, void* dst, size_t rows, size_t cols, size_t dst_ld,
                                  int row_map, int col_map, int d, hipStream_t st, int parts = 2);
// RoPE tables cos/sin[t][i] = cos/sin(t * inv_freq[i]) (rotary_embedding.py:47-61), fp32
hipError_t launch_rope_table(const float* inv_freq, float* cos, float* sin, int T, int half,
                             hipStream_t st);
hipError_t launch_copy_f32(const float* src, float* dst, size_t n, hipStream_t st);
// out[b,:] = mean of rows [first, first + min(count[b], T - first)) of x[b] ([B,T,E], any dtype code); NaN if empty
hipError_t launch_masked_row_mean(const void* x, int x_dtype, const int* count, float* out, int B, int T, int E,
                                  int first, hipStream_t st);
// contact head (modules.py:27-41,338-357)
hipError_t launch_contacts(const float* attn, const int64_t* tokens, const float* w,
                           const float* b, float* scratch, float* out, int B, int C, int T,
                           int eos_idx, int prepend_bos, int append_eos, hipStream_t st);

// contacts.hip — contact maps without the [B,L,H,T,T] attention tensor (predict_contacts, esm2.py:146-147).
// Per layer, after the attention kernel (q, k, lse still in the workspace): A[G][B,T,T] += sum_h w[layer,h] P_h
// (G = contacts_head_groups(B,T,H) accumulators, one per head group), rowsum / colsum [B,C,T] (C = L*H) masked sums
// of every channel; rowp [B, ceil(T/128), H, T] and colp [B, ceil(T/32), H, T] are per-layer scratch.
int contacts_head_groups(int B, int T, int H, int head_dim);
hipError_t launch_contacts_fused_layer(const void* q, const void* k, const float* lse, const float* key_bias,
                                       const int64_t* tokens, const float* wreg, float* acc, float* rowsum,
                                       float* colsum, float* rowp, float* colp, int B, int H, int T, int C, int layer,
                                       int head_dim, int pad_idx, int eos_idx, int prepend_bos, int append_eos,
                                       int operand_dtype, hipStream_t st);
// after the last layer: rowsum becomes r_c (in place), wt [B,C] = w_c / t_c, out [B,S,S] = sigmoid(logits)
hipError_t launch_contacts_fused_final(const float* acc, float* rowsum, const float* colsum, float* wt,
                                       const int64_t* tokens, const float* wreg, const float* bias, float* out,
                                       int B, int H, int C, int T, int head_dim, int pad_idx, int eos_idx,
                                       int prepend_bos, int append_eos, hipStream_t st);

// token-packed batch (esmk_forward_packed): per-row bookkeeping of the packed row space.  Segment s occupies
// rows [seg[2s], seg[2s] + seg[2s+1]); rows outside every segment are gaps.
//   scale_row[m] = 1 - n_mask/len of m's segment (esm2.py:91-92; 1 in gaps), key_bias[m] = 0 / -inf (pad, gap),
//   row_pos[m] = m - segment start (0 in gaps), seg_npad[s] = number of <pad> tokens inside segment s
hipError_t launch_packed_stats(const int64_t* tokens, const int* seg, int n_seg, int rows, int pad_idx,
                               int mask_idx, float* scale_row, float* key_bias, int* row_pos, int* seg_npad,
                               hipStream_t st, float* keep = nullptr);  // keep[m] = 1 - pad (0 in gaps), optional

// rows outside every segment of buf[rows][row_bytes] := 0 (the attention kernel does not write them)
hipError_t launch_zero_gap_rows(void* buf, const int* seg, int n_seg, int rows, size_t row_bytes, hipStream_t st);

// ---- attention.hip ---------------------------------------------------------------------
// toolchain guard: the inline-asm MFMA of common.h (mma_keep_c) against the builtin; a, b [64 lanes][8] operand dtype,
// c [64][16] fp32, out [3][64][16] = {asm path, builtin path, C after the calls}
hipError_t launch_mma_keep_c_selftest(const void* a, const void* b, const float* c, float* out, int operand_dtype,
                                      hipStream_t st);
// query-block work list of a token-packed batch: work[4i..4i+3] = (first row of the segment, segment length,
// first query of block i relative to the segment, segment index); npad[s] = <pad> tokens inside segment s
struct AttnSegs {
    const int* work = nullptr;
    const int* npad = nullptr;
};
hipError_t launch_attention_packed(const void* q, const void* k, const void* vt, const float* key_bias, void* ctx,
                                   int H, int rows, int Tp, AttnSegs segs, int n_items, int operand_dtype,
                                   hipStream_t st);
hipError_t launch_attention128_packed(const void* q, const void* k, const void* vt, const float* key_bias, void* ctx,
                                      int H, int rows, int Tp, AttnSegs segs, int n_items, int operand_dtype,
                                      hipStream_t st);
hipError_t launch_attention(const void* q, const void* k, const void* vt, const float* key_bias,
                            const int* seq_info, void* ctx, float* lse, int B, int H, int T, int Tp,
                            int operand_dtype, hipStream_t st);
hipError_t launch_attention_x3(const void* q, const void* k, const void* vt, const float* key_bias, const int* seq_info, void* ctx3,
                               float* lse, int B, int H, int T, int Tp, int operand_dtype, hipStream_t st);
// attention128.hip: head_dim 128 (esm2_t48_15B)
hipError_t launch_attention128(const void* q, const void* k, const void* vt, const float* key_bias,
                               const int* seq_info, void* ctx, float* lse, int B, int H, int T, int Tp,
                               int operand_dtype, hipStream_t st);
// lowp: the maps are written in the operand dtype (probs points to fp16 / bf16 storage)
hipError_t launch_attention_probs128(const void* q, const void* k, const float* lse, const float* key_bias,
                                     float* probs, int B, int H, int T, int layer, int num_layers_total,
                                     int operand_dtype, hipStream_t st, bool lowp = false);
// same kernel, MSA column attention: key_fill[b,t] != 0 REPLACES the score by -10000 (masked_fill,
// axial_attention.py:211-215) and is only applied when any_pad[0] != 0
hipError_t launch_attention_fill(const void* q, const void* k, const void* vt, const float* key_fill,
                                 const int* any_pad, void* ctx, float* lse, int B, int H, int T, int Tp,
                                 int operand_dtype, hipStream_t st);
// col_attentions[b, layer, h, c, i, j] (msa_transformer.py:193-194) from q, k and the saved log-sum-exp
hipError_t launch_attention_probs_msa(const void* q, const void* k, const float* lse, const float* key_fill,
                                      const int* any_pad, float* probs, int Bmsa, int C, int H, int R,
                                      int layer, int num_layers_total, int operand_dtype, hipStream_t st);
hipError_t launch_attention_probs(const void* q, const void* k, const float* lse,
                                  const float* key_bias, float* probs, int B, int H, int T,
                                  int layer, int num_layers_total, int operand_dtype,
                                  hipStream_t st, bool lowp = false);

}  // namespace esmk
