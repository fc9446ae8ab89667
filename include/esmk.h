/*
 * esmk.h — C ABI of libesmk.so, the MI355X (gfx950) ESM-2 forward engine.
 *
 * The reference (facebookresearch/esm) has no FFI: its boundary for this path is the Python
 * call  ESM2.forward(tokens, repr_layers, need_head_weights, return_contacts)
 * (reference esm/model/esm2.py:77-147).  libesmk.so sits directly under that call: the Python
 * class esm_amd.esm2.ESM2 keeps the reference's nn.Module surface and hands raw device
 * pointers to esmk_forward().  Every entry point below names the reference code it replaces.
 *
 * Conventions
 *   - every pointer named *_dev is a device pointer owned by the caller (a torch tensor);
 *     the library borrows it for the duration of the call and allocates nothing persistent
 *     except the small RoPE cos/sin table (freed in esmk_destroy);
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*);
 *   - return value 0 = ok, non-zero = error, message via esmk_last_error() (thread local);
 *   - a handle is bound to the device that was current at esmk_create() and is not re-entrant.
 */
#ifndef ESMK_H
#define ESMK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types for esmk_bind_weight / operand_dtype */
enum { ESMK_F32 = 0, ESMK_F16 = 1, ESMK_BF16 = 2 };

/* esmk_forward out_flags */
enum {
    ESMK_OUT_LOGITS = 1u,   /* logits [B,T,V] fp32                       (esm2.py:129)      */
    ESMK_OUT_ATTN = 2u,     /* attentions [B,L,H,T,T] fp32               (esm2.py:132-139)  */
    ESMK_OUT_CONTACTS = 4u, /* contacts [B,T-2,T-2] fp32                 (esm2.py:140-142).  Together with
                               ESMK_OUT_ATTN: computed from the attention tensor like the reference
                               (modules.py:338-357).  WITHOUT ESMK_OUT_ATTN (predict_contacts, esm2.py:146-147):
                               accumulated layer by layer, no [B,L,H,T,T] tensor exists (csrc/contacts.hip)        */
    ESMK_OUT_COL_ATTN = 8u, /* esmk_msa_forward: col_attentions [B,L,H,C,R,R] fp32 (msa_transformer.py:193-194) */
    /* `.half()` / `.bfloat16()` models return their outputs in the model dtype (esm2.py:77-144 run under
     * nn.Module.half(); ESMFold's language-model front end does exactly that, esmfold/v1/esmfold.py:61-67,118-135,
     * and stacks all L+1 representations).  With these flags the engine writes them in the OPERAND dtype directly
     * instead of fp32 + a cast pass: */
    ESMK_OUT_REPR_LOWP = 16u, /* repr_out_dev[i] are operand-dtype [B,T,E] (esmk_forward and esmk_forward_packed) */
    ESMK_OUT_ATTN_LOWP = 32u  /* attn_out_dev is operand-dtype [B,L,H,T,T]; not together with ESMK_OUT_CONTACTS on
                                 the materialised path (the contact kernels read fp32 maps) */
};

typedef struct esmk_model esmk_model;

/* Model hyper-parameters: the constructor arguments of ESM2 (esm/model/esm2.py:15-38) plus the
 * alphabet ids it copies from the Alphabet (esm/data.py:116-120). */
typedef struct esmk_config {
    int32_t num_layers;      /* L                                                        */
    int32_t embed_dim;       /* E                                                        */
    int32_t num_heads;       /* H, head_dim d = E/H                                      */
    int32_t ffn_dim;         /* 4E for ESM-2 (esm2.py:53)                                */
    int32_t vocab;           /* len(alphabet) = 33                                       */
    int32_t pad_idx, mask_idx, cls_idx, eos_idx;
    int32_t token_dropout;   /* esm2.py:86-92                                            */
    int32_t prepend_bos, append_eos; /* contact head crop (modules.py:338-347)           */
    int32_t operand_dtype;   /* ESMK_F16 or ESMK_BF16: MFMA operand type; accumulation,
                                residual stream, LayerNorm, softmax are always fp32     */
    /* ESM-1b / ESM-1v (ProteinBertModel with arch "roberta_large", esm/model/esm1.py:88-104,117-143); all
     * zero for ESM-2 */
    int32_t no_rope;         /* 1: no rotary embedding (TransformerLayer(use_rotary_embeddings=False)) */
    int32_t num_positions;   /* > 0: rows of the LearnedPositionalEmbedding table added to the token embedding
                                (esm1.py:133, modules.py:240-257); key "embed_positions.weight"   */
    int32_t ln_before;       /* 1: emb_layer_norm_before (esm1.py:136-137)                          */
    /* Precision mode "f16x2" (operand_dtype must be ESMK_F16): the weight matrices of the layer stack are kept as
     * W = W_hi + W_lo (two fp16 images, ~20 bits of every weight) and every layer GEMM runs both against the fp16
     * activations — the weight rounding, two thirds of the fp16-operand error of a 33-layer stack,
     * disappears at 2x the GEMM time.  Parameter image 2x larger.  0 = plain fp16 / bf16 operands.
     * 2 = "f16x2a" (round 6): the same for the ATTENTION projections only (q, k, v, out_proj:
     * esm/multihead_attention.py:256-261,395 — a third of the GEMM work); fc1 / fc2 stay plain fp16, the LM head runs in
     * fp32 as with 1.  Representations and logits inside 1e-3 in both norms at ~1.3x the plain step.
     * 3 = "f16x2v": the value path only (v_proj, out_proj: a sixth of the GEMM work, ~1.2x).
     * 4 = "f16x3": weights AND GEMM inputs split — every matrix is packed hi | lo | hi per 64-column K tile and every layer
     * GEMM runs as a plain launch over K' = 3 K on operand rows hi | hi | lo (A_hi W_hi + A_hi W_lo + A_lo W_hi); only q / k, v
     * and P of the attention stay fp16.  Representations, logits AND contact logits inside 1e-3 of the reference
     * (tests/test_readme.py:116 atol) at ~2.4x the step.  head_dim 64, embed_dim % 64 == 0, padded batches (esmk_forward). */
    int32_t weight_split;
    /* LayerNorm fold (reference esm/modules.py:120-140, the two LayerNorm -> Linear pairs of a TransformerLayer): 1 = the
     * q/k/v and fc1 weights are packed multiplied by the LayerNorm weight and row-centred, the residual GEMMs emit the
     * operand-dtype rows and their statistics, and the standalone per-layer LayerNorm passes disappear (plain fp16 / bf16
     * operands, head_dim <= 64; esmk_create fails otherwise); -1 = off; 0 = the library's default (environment
     * ESMK_LN_FOLD=0|1 overrides it).  With the fold the LayerNorm weight and bias of a layer MUST be packed before
     * that layer's q/k/v and fc1 weights (esmk_pack_weight fails otherwise; esmk_forward fails while a fold is stale).
     * The handle tracks the fold of ONE packed image at a time — the one esmk_pack_weight last wrote to; packing into a
     * different image starts from "nothing packed", and esmk_forward refuses an image the handle did not pack.
     * Library default since round 5: ON. */
    int32_t ln_fold;
} esmk_config;

const char* esmk_last_error(void);
const char* esmk_version(void);

/* Replaces ESM2.__init__/_init_submodules (esm2.py:15-75): records dimensions only. */
int esmk_create(const esmk_config* cfg, esmk_model** out);
void esmk_destroy(esmk_model* m);

/* RoPE inverse frequencies, fp32 host array of head_dim/2 values, exactly the buffer
 * RotaryEmbedding.__init__ builds (esm/rotary_embedding.py:40-41). The cos/sin tables
 * (rotary_embedding.py:47-61) are built on the device in fp32 from it. */
int esmk_set_rope_inv_freq(esmk_model* m, const float* inv_freq_host, int n);

/* Bytes of the packed parameter image (operand-dtype matrices + fp32 vectors). */
int esmk_packed_bytes(const esmk_model* m, size_t* bytes);

/* Replaces nn.Module.load_state_dict for the engine copy: converts ONE state-dict tensor
 * (key names as in esm2.py state_dict, e.g. "layers.3.self_attn.q_proj.weight") from its
 * device buffer into the packed image.  q/k/v projections are concatenated to one [3E,E]
 * operand; the tied lm_head.weight (esm2.py:71-75) is taken from embed_tokens.weight.
 * Unknown keys (e.g. "...rot_emb.inv_freq") return 0 and are ignored. */
int esmk_pack_weight(esmk_model* m, void* packed_dev, size_t packed_bytes, const char* key,
                     const void* src_dev, int src_dtype, const int64_t* shape, int ndim,
                     void* stream);

/* Workspace bytes for one forward of [B,T] tokens with the given outputs. */
int esmk_workspace_bytes(const esmk_model* m, int B, int T, uint32_t out_flags, size_t* bytes);

/* Replaces ESM2.forward (esm/model/esm2.py:77-144) — embedding + token dropout, the
 * TransformerLayer loop (esm/modules.py:120-142 -> esm/multihead_attention.py:159-405,
 * esm/rotary_embedding.py:63-69), final LayerNorm, RobertaLMHead (modules.py:308-314),
 * attention maps and ContactPredictionHead (modules.py:338-357).
 *   tokens_dev     int64 [B,T]
 *   repr_layers    host array of n_repr layer indices in [0,L]; repr_out_dev[i] fp32 [B,T,E]
 *   logits_out_dev fp32 [B,T,V]            (required iff ESMK_OUT_LOGITS)
 *   attn_out_dev   fp32 [B,L,H,T,T]        (required iff ESMK_OUT_ATTN)
 *   contacts_out_dev fp32 [B,T-2,T-2]      (required iff ESMK_OUT_CONTACTS)
 */
int esmk_forward(esmk_model* m, const void* packed_dev, const int64_t* tokens_dev, int B, int T,
                 const int32_t* repr_layers, int n_repr, void* const* repr_out_dev,
                 uint32_t out_flags, void* logits_out_dev, void* attn_out_dev,
                 void* contacts_out_dev, void* workspace_dev, size_t workspace_bytes,
                 void* stream);

/* ---- token-packed batches: the same forward without compute on padding -------------
 * The reference pads every batch to its longest member (esm/data.py:269-277) and ESM2.forward computes the
 * pad rows (esm2.py:94-95 only zeroes them at the input).  Here the caller lays the sequences of a batch back
 * to back in ONE row space of `rows` rows (rows % 64 == 0): segment s occupies rows
 * [segments_host[2s], segments_host[2s] + segments_host[2s+1]); segment 0 starts at row 0, starts are
 * ascending multiples of 16, rows between segments ("gaps") hold pad_idx.  Every token attends to its own
 * segment only, rotary positions restart at each segment, the token-dropout ratio (esm2.py:86-92) is per
 * segment: rows of a segment carry exactly the values esmk_forward gives that sequence alone.
 *   tokens_dev      int64 [rows]
 *   segments_host   int32 [n_seg][2] on the HOST (first row, length incl. <cls>/<eos>)
 *   repr_out_dev[i] fp32 [rows,E]; logits_out_dev fp32 [rows,V] (iff ESMK_OUT_LOGITS); gap rows are undefined
 * ESM-1b / ESM-1v handles work the same way (learned positions restart at each segment, esm/modules.py:240-257).
 * Attention maps / contacts are [T,T] per sequence and stay with esmk_forward: those flags fail with an error. */
int esmk_packed_workspace_bytes(const esmk_model* m, int n_seg, int rows, uint32_t out_flags, size_t* bytes);
int esmk_forward_packed(esmk_model* m, const void* packed_dev, const int64_t* tokens_dev,
                        const int32_t* segments_host, int n_seg, int rows, const int32_t* repr_layers,
                        int n_repr, void* const* repr_out_dev, uint32_t out_flags, void* logits_out_dev,
                        void* workspace_dev, size_t workspace_bytes, void* stream);

/* ---- MSA Transformer (reference esm/model/msa_transformer.py:20-238, esm/axial_attention.py) -------- */

/* Constructor arguments of MSATransformer (msa_transformer.py:88-144) + alphabet ids. */
typedef struct esmk_msa_config {
    int32_t num_layers, embed_dim, num_heads, ffn_dim, vocab;
    int32_t pad_idx, mask_idx, cls_idx, eos_idx, prepend_bos, append_eos;
    int32_t num_positions;               /* rows of embed_positions.weight = max_positions + pad_idx + 1 */
    int32_t has_msa_position_embedding;  /* args.embed_positions_msa (msa_transformer.py:104-112) */
    int32_t operand_dtype;               /* ESMK_F16 or ESMK_BF16 */
    int32_t weight_split;                /* 1: precision mode f16x2 (see esmk_config::weight_split; operand_dtype ESMK_F16): every weight
                                            matrix of the axial layers as W_hi + W_lo, the LM head on the fp32 MFMA path;
                                            2: f16x2a, the row / column attention projections only; 3: f16x2v, their v / out
                                            projections only */
} esmk_msa_config;

/* Replaces MSATransformer.__init__; the handle is packed with esmk_pack_weight (MSATransformer
 * state-dict keys: layers.N.{row,column}_self_attention.layer.*_proj.*, ...layer_norm.*,
 * layers.N.feed_forward_layer.layer.fc{1,2}.*, embed_positions.weight, msa_position_embedding, ...)
 * and freed with esmk_destroy. */
int esmk_msa_create(const esmk_msa_config* cfg, esmk_model** out);
int esmk_msa_workspace_bytes(const esmk_model* m, int B, int R, int C, uint32_t out_flags, size_t* bytes);

/* Replaces MSATransformer.forward (msa_transformer.py:146-220): embedding (+ LearnedPositionalEmbedding
 * modules.py:240-257), AxialTransformerLayer stack (modules.py:196-221: RowSelfAttention
 * axial_attention.py:75-130, ColumnSelfAttention :185-239, FeedForwardNetwork modules.py:395-418),
 * final LayerNorm, RobertaLMHead, ContactPredictionHead on the row attentions.
 *   tokens_dev       int64 [B,R,C]
 *   repr_out_dev[i]  fp32 [B,R,C,E]
 *   logits_out_dev   fp32 [B,R,C,V]        (iff ESMK_OUT_LOGITS)
 *   row_attn_out_dev fp32 [B,L,H,C,C]      (iff ESMK_OUT_ATTN or ESMK_OUT_CONTACTS)
 *   col_attn_out_dev fp32 [B,L,H,C,R,R]    (iff ESMK_OUT_COL_ATTN; 4.8 GB per layer for a 128 x 513 MSA)
 *   contacts_out_dev fp32 [B,C-1,C-1]      (iff ESMK_OUT_CONTACTS) */
int esmk_msa_forward(esmk_model* m, const void* packed_dev, const int64_t* tokens_dev, int B, int R, int C,
                     const int32_t* repr_layers, int n_repr, void* const* repr_out_dev, uint32_t out_flags,
                     void* logits_out_dev, void* row_attn_out_dev, void* col_attn_out_dev,
                     void* contacts_out_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* Per-kernel-class timing of esmk_forward with HIP events recorded on the launch stream
 * (measurement support for bench.py; the reference has no counterpart).
 * esmk_profile_begin() arms it; every launch of the following esmk_forward() calls is bracketed
 * by two events; esmk_profile_end() waits for them and returns one aggregated entry per class:
 * launches, total milliseconds, algorithmic FLOPs and algorithmic bytes. */
typedef struct esmk_profile_entry {
    char name[32];
    int32_t launches;
    double ms;
    double flops;
    double bytes;
} esmk_profile_entry;
int esmk_profile_begin(esmk_model* m);
int esmk_profile_end(esmk_model* m, esmk_profile_entry* out, int max_entries, int* n_out);
/* 1 if the handle runs with the LayerNorm fold (esmk_config::ln_fold resolved against the library default and
 * ESMK_LN_FOLD), 0 if not, -1 for a null / MSA handle.  Measurement scripts record the mode next to their numbers. */
int esmk_ln_fold_enabled(const esmk_model* m);

/* ---- single-kernel entry points (used by the parity tests and micro-benchmarks) -------- */

/* ESM1bLayerNorm == torch.nn.LayerNorm(E, eps=1e-5) (esm/modules.py:68-81).
 * x fp32 [rows,E] -> y (operand dtype) [rows,E] and/or y32 fp32 [rows,E] (either may be NULL). */
int esmk_op_layernorm(const float* x_dev, const float* gamma_dev, const float* beta_dev,
                      void* y_dev, float* y32_dev, int rows, int E, int operand_dtype,
                      void* stream);

/* Mean representation of every sequence of a batch, as scripts/extract.py:113-116 computes it on the host
 * (`t[i, 1 : truncate_len + 1].mean(0)`): out[b,:] (fp32 [B,E]) = mean of rows [first_row, first_row + n_b) of
 * x[b] ([B,T,E] in dtype code x_dtype), n_b = min(count_dev[b], T - first_row); an empty slice yields NaN like
 * torch.mean.  One pass over x, deterministic (no atomics). */
int esmk_op_masked_row_mean(const void* x_dev, int x_dtype, const int32_t* count_dev, float* out_dev, int B, int T,
                            int E, int first_row, void* stream);

/* nn.Linear: C[M,N] = A[M,K] . W[N,K]^T + bias, epilogue selected by `epilogue`:
 *   0 plain -> out operand dtype [M,N]           1 plain -> out fp32 [M,N]
 *   2 gelu (modules.py:17-24) -> operand dtype   3 gelu -> fp32
 *   4 residual: out fp32 [M,N] += result         (modules.py:134,140)
 * A, W operand dtype; bias fp32 (may be NULL). */
int esmk_op_linear(const void* a_dev, const void* w_dev, const float* bias_dev, void* out_dev,
                   int M, int N, int K, int epilogue, int operand_dtype, void* stream);

/* Measurement hook (no reference counterpart; tools/bench_splitk.py): S fp32 partial products
 * out[s][M,N] = A[:, sK/S:(s+1)K/S] . W[:, sK/S:(s+1)K/S]^T as ONE batched launch of the persistent kernel
 * (K/S a multiple of 64).  Small-batch study: a [4096,5120]x[1280,5120] GEMM has 80 tiles for 256 CUs. */
int esmk_debug_linear_splitk(const void* a_dev, const void* w_dev, float* partials_dev, int M, int N, int K,
                             int S, int operand_dtype, void* stream);

/* Measurement hook (no reference counterpart): when stamps_dev != NULL every following persistent
 * GEMM launch records s_memtime stamps per workgroup and tile, uint64 [256][32][4] =
 * {tile start, main loop done, epilogue done, unused}; NULL switches it off. */
int esmk_debug_gemm_timing(void* stamps_dev);

/* Split-weight GEMM of the f16x2 precision mode as single ops (tests, tools/bench_gemm9.py):
 * esmk_op_split_weight: w [N,K] (any float dtype) -> w2 fp16 [N,2K], K tiles of 64 columns interleaved hi | lo with
 * hi = fp16(w), lo = fp16(w - hi);  esmk_op_linear_split: out = a[M,K] . (w_hi + w_lo)^T + bias with the epilogues of
 * esmk_op_linear (K % 64 == 0, N % 8 == 0). */
int esmk_op_split_weight(const void* w_dev, int w_dtype, void* w2_dev, int N, int K, void* stream);
int esmk_op_linear_split(const void* a_dev, const void* w2_dev, const float* bias_dev, void* out_dev, int M, int N, int K,
                         int epilogue, void* stream);

/* Toolchain guard (no reference counterpart): the attention / contact kernels issue one MFMA per key tile through inline
 * asm (its C operand, the softmax offset broadcast, must survive); the compiler does not see that instruction's hazards.
 * Runs it beside the builtin on the same operands: a, b [64][8] operand dtype, c [64][16] fp32, out [3][64][16] fp32 =
 * {asm path, builtin path, c after the calls}; the first two must be bit-equal, the third equal to 2 c
 * (tests/test_kernels_gpu.py runs it on every GPU test run, i.e. on every toolchain the library is built with). */
int esmk_debug_mma_selftest(const void* a_dev, const void* b_dev, const float* c_dev, float* out_dev, int operand_dtype,
                            void* stream);

/* Measurement / A-B hook (no reference counterpart): which persistent GEMM kernel serves the dense nn.Linear calls
 * from now on — 8 = gemm8.hip (two waves per SIMD), 9 = gemm9.hip (one wave per SIMD, 128 x 128 wave blocks;
 * bit-identical results) wherever it applies, 0 = the library's own choice per call (default).  `variant` selects a
 * gemm9 barrier placement / timing experiment (gemm9.hip).  The environment variable ESMK_GEMM_IMPL=8|9[:variant]|auto
 * sets the same thing for a whole process. */
int esmk_debug_gemm_impl(int impl, int variant);

/* Measurement / A-B hook (no reference counterpart): named tuning knobs of the library, process wide.  Timing only —
 * no knob changes a result bit.  (Timing experiments that DO break results — kernels with their MFMAs, exponentials,
 * DMA or epilogue stores removed: the "lnf_dbg" knob, ESMK_ATTN_HACK, the gemm8 / gemm9 variant codes with parts switched
 * off — exist only in libraries built with ESMK_HIPCC_EXTRA="-DESMK_EXPERIMENTS"; the shipped library refuses them.)  "resid_desync" (>= 0): start-up delay of every other XCD's workgroups in the residual
 * GEMMs (out-projection, fc2), as a fraction of one tile's main loop, which takes the HBM-bound read-modify-write
 * epilogues of the two halves of the chip out of lockstep (gemm9.hip; environment: ESMK_RESID_DESYNC);
 * "resid_desync_group": 0 = odd XCDs late, 1 = every other workgroup of each XCD, 2 = four phases.
 * "attn_stagger" (>= 0): start-up delay, in shader cycles per wave slot, of the co-resident workgroups of the attention
 * kernel (attention.hip; environment: ESMK_ATTN_STAGGER).
 * "qkv_one_launch": 1 / 0 = always / never run the q, k and v projections of a layer as ONE GEMM launch, -1 = the library's
 * choice (one launch where it needs fewer rounds of tiles over the CUs, i.e. small batches; gemm.hip; environment:
 * ESMK_QKV_ONE_LAUNCH). */
int esmk_debug_set(const char* key, double value);

/* Fused q/k/v projection + scaling + rotary + head split (multihead_attention.py:256-284,
 * :354-355; rotary_embedding.py:11-20).  a [B*T,E]; wqkv [3E,E]; bias [3E];
 * q_out,k_out [B,H,T,64]; vt_out [B,H,64,Tp] (V transposed, keys permuted in groups of 16,
 * Tp = T rounded up to 64; see attention.hip). */
int esmk_op_qkv_rope(esmk_model* m, const void* a_dev, const void* wqkv_dev,
                     const float* bias_dev, void* q_out, void* k_out, void* vt_out, int B, int T,
                     void* stream);
/* The same with the score domain made explicit: log2_domain = 1 folds log2(e) into the q scale as well (one rounding
 * to the operand dtype), which is the q esmk_op_attention / esmk_op_attention_probs expect — the two ops then compose
 * without a conversion; log2_domain = 0 is esmk_op_qkv_rope (q scaled by d^-1/2 only, the reference's q). */
int esmk_op_qkv_rope2(esmk_model* m, const void* a_dev, const void* wqkv_dev,
                      const float* bias_dev, void* q_out, void* k_out, void* vt_out, int B, int T,
                      int log2_domain, void* stream);

/* LayerNorm fold (esmk_config::ln_fold) as single ops — the pieces esmk_forward chains per layer (modules.py:120-140):
 * esmk_op_rowstats: x fp32 [rows,E] -> y[row][c] = T(x - mean) (row stride ldy, operand dtype), mean[row], rstd[row]
 *   (eps 1e-5, biased variance: the statistics of ESM1bLayerNorm, modules.py:68-81);
 * esmk_op_fold_weight: w [N,K] of the Linear that follows a LayerNorm(gamma, beta) -> dst[n][k] = T(w[n][k] gamma[k] -
 *   mean_k(w[n][.] gamma[.])) (row stride ld), bias2[n] = sum_k w[n][k] beta[k];
 * esmk_op_linear_ln, epilogue 2 (consumer): out = T(gelu(ln_rstd[m] * (a . w^T) + bias + bias2)), a = the rows of
 *   rowstats / of a producer, w = a folded image;  epilogue 4 (producer): out fp32 [M,N] += a . w^T + bias, and
 *   h16[m][n] = T(out_new - ln_mean[m]) (row stride ldh), ln_part[m][n / 128] = (sum, sum of squares) of out_new -
 *   ln_mean[m] over the 128 columns (ln_parts >= ceil(N / 128) entries per row);
 * esmk_op_ln_finalize: ln_part -> mean[row] += sum / E, rstd[row] = rsqrt(var + 1e-5);
 * esmk_op_qkv_rope_ln: esmk_op_qkv_rope2 with folded wqkv and the rows' rstd (ln_rstd must be readable up to the next
 *   multiple of 256 rows). */
int esmk_op_rowstats(const float* x_dev, void* y_dev, float* mean_dev, float* rstd_dev, int rows, int E, int ldy,
                     int operand_dtype, void* stream);
int esmk_op_ln_finalize(const float* part_dev, float* mean_dev, float* rstd_dev, int rows, int parts, int E, void* stream);
int esmk_op_fold_weight(const void* w_dev, int w_dtype, const float* gamma_dev, const float* beta_dev, void* dst_dev,
                        int dst_dtype, float* bias2_dev, int N, int K, int ld, void* stream);
int esmk_op_linear_ln(const void* a_dev, const void* w_dev, const float* bias_dev, const float* bias2_dev, void* out_dev,
                      int M, int N, int K, int epilogue, int operand_dtype, const float* ln_rstd_dev, void* h16_dev, int ldh,
                      float* ln_part_dev, int ln_parts, const float* ln_mean_dev, int half_m, void* stream);
int esmk_op_qkv_rope_ln(esmk_model* m, const void* a_dev, const void* wqkv_dev, const float* bias_dev,
                        const float* bias2_dev, const float* ln_rstd_dev, void* q_out, void* k_out, void* vt_out, int B, int T,
                        int log2_domain, void* stream);

/* softmax(q k^T + key_bias) v  (multihead_attention.py:357-394), flash style.
 * SCORE DOMAIN: q must carry log2(e) besides d^-1/2 (esmk_forward's QKV epilogue folds both into the q scale
 * before the single rounding to the operand dtype); q.k is then the score in the log2 domain, the kernel's
 * exponentials are exp2, and lse_out is the row log2-sum-exp2 (= natural lse * log2 e).
 * esmk_op_attention_probs takes the same q and that lse.  (esmk_op_qkv_rope scales q by d^-1/2 only.)
 * key_bias fp32 [B,T] (0 or -inf, NULL = no padding); ctx_out [B*T, H*64] operand dtype;
 * lse_out fp32 [B,H,T] or NULL. */
int esmk_op_attention(const void* q_dev, const void* k_dev, const void* vt_dev,
                      const float* key_bias_dev, void* ctx_out, float* lse_out, int B, int H,
                      int T, int operand_dtype, void* stream);

/* Per-head attention probabilities (multihead_attention.py:396-403, esm2.py:132-139):
 * probs_out fp32 [B, Ltot, H, T, T] slice `layer`, rows/cols of padded tokens zeroed. */
int esmk_op_attention_probs(const void* q_dev, const void* k_dev, const float* lse_dev,
                            const float* key_bias_dev, float* probs_out, int B, int H, int T,
                            int layer, int num_layers_total, int operand_dtype, void* stream);

/* ContactPredictionHead.forward (modules.py:338-357) incl. symmetrize/apc (modules.py:27-41).
 * attn fp32 [B,C=L*H,T,T]; w fp32 [C]; b fp32 [1]; scratch fp32 >= B*C*(T+1) floats;
 * out fp32 [B,T-2,T-2] (crop follows prepend_bos/append_eos). */
int esmk_op_contacts(const float* attn_dev, const int64_t* tokens_dev, const float* w_dev,
                     const float* b_dev, float* scratch_dev, float* out_dev, int B, int C, int T,
                     int eos_idx, int prepend_bos, int append_eos, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ESMK_H */
