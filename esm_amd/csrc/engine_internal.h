// engine_internal.h — state shared by the host-side translation units of libesmk.so (engine.hip: ESM-2 /
// ESM-1b path and the generic C ABI; engine_msa.hip: MSA Transformer path).  Not part of the public ABI.
#pragma once
#include "../../include/esmk.h"
#include "kernels.h"

#include <string>
#include <vector>

namespace esmk_host {

// error reporting: message kept per thread for esmk_last_error()
int fail(const char* what, hipError_t e);
int fail(const std::string& msg);

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }
inline size_t op_size(int dt) { return dt == esmk::ESMK_DT_F32 ? 4 : 2; }

// sequential carving of byte offsets (256-byte aligned) out of one buffer
struct Carve {
    size_t off = 0;
    size_t take(size_t bytes) {
        size_t o = off;
        off = align_up(off + bytes);
        return o;
    }
};

// packed-parameter offsets of one TransformerLayer (reference esm/modules.py:84-142)
struct LayerOff {
    size_t wqkv, bqkv, wo, bo, w1, b1, w2, b2, ln1g, ln1b, ln2g, ln2b;
    size_t bqkv2 = 0, b12 = 0;  // LayerNorm fold: W . beta of the folded q/k/v and fc1 weights (GemmArgs::bias2)
};
// one AxialTransformerLayer (reference esm/modules.py:145-221)
struct AttnOff {
    size_t wqkv, bqkv, wo, bo, lng, lnb;
};
struct MsaLayerOff {
    AttnOff row, col;
    size_t w1, b1, w2, b2, flng, flnb;
};

}  // namespace esmk_host

// row counts travel as int through the kernels (row * ld products are widened to 64 bit); 2^24 rows of fp32
// residual already exceed one GPU's HBM for every supported width, so this is a guard, not a limit in practice
#define ESMK_MAX_ROWS (1LL << 24)

#define ESMK_TRY(expr)                                             \
    do {                                                           \
        hipError_t _e = (expr);                                    \
        if (_e != hipSuccess) return esmk_host::fail(#expr, _e);   \
    } while (0)

struct esmk_model {
    esmk_config cfg;
    int L, E, H, F, V, D;
    int EA = 0;  // attention width inside the engine: H * 64 (heads with head_dim < 64 are spread over 64 slots)
    int Kp = 0;  // E rounded up to the 64-wide K tile: row stride of the normalised activations
    // packed parameter image layout (byte offsets)
    size_t embed_f32, embed_op, fin_g, fin_b, lm_w, lm_b, lm_lng, lm_lnb, lm_bias, ct_w, ct_b;
    size_t lm_w32 = 0;  // f16x2 precision mode: lm_head.dense.weight in fp32 (the head runs on the fp32 MFMA path)
    std::vector<esmk_host::LayerOff> layer;
    size_t packed_bytes;
    // RoPE
    std::vector<float> inv_freq;
    float* d_inv_freq = nullptr;
    float* d_cos = nullptr;
    float* d_sin = nullptr;
    int rope_cap = 0;
    // optional per-kernel-class timing with HIP events (esmk_profile_begin / _end)
    struct ProfRec {
        int cls;
        hipEvent_t a, b;
        double flops, bytes;
    };
    bool prof_on = false;
    std::vector<ProfRec> prof;
    // esmk_forward_packed: pinned host staging of the segment / work tables; pk_event = its last upload
    int32_t* pk_host = nullptr;
    size_t pk_host_cap = 0;
    hipEvent_t pk_event = nullptr;
    // ESMK_QKV_FORK: the v projection of a layer runs on a stream of the library's own, next to the q/k projection on
    // the caller's stream (both only read the normalised rows): when neither launch fills a whole number of rounds
    // over the CUs (small batches, MSA row counts), the workgroups of one take the CUs the other leaves idle
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // LayerNorm fold: q/k/v and fc1 weights are packed gamma-folded and row-centred, the per-layer
    // LayerNorm passes become GEMM epilogue work.  fold_state[l]: bits of FoldBits — which inputs of the fold have been
    // packed, and which folded images are current (a LayerNorm parameter packed after its weights makes them stale).
    bool fold = false;
    std::vector<uint32_t> fold_state;
    // the packed image fold_state describes: the bits belong to ONE caller-owned image, so packing into (or running
    // on) another image starts from "nothing packed" instead of inheriting the previous image's bits
    const void* fold_image = nullptr;
    // MSA Transformer (esmk_msa_create)
    bool is_msa = false;
    int npos = 0, has_msa_pos = 0;
    size_t pos_emb = 0, msa_pos = 0, lnb_g = 0, lnb_b = 0;
    std::vector<esmk_host::MsaLayerOff> mlayer;
    float* d_ucos = nullptr;  // "unit" rotary tables (cos = 1, sin = 0): the MSA model has no RoPE
    float* d_usin = nullptr;
    int unit_cap = 0;
};

// Precision modes with split weights (esmk_config::weight_split): which matrices of a layer are kept as W_hi + W_lo (factor 2:
// image rows of 2 K, two MFMA passes) — 1 = f16x2: all;  2 = f16x2a: the attention projections q, k, v, out (a third of the
// GEMM work);  3 = f16x2v: the value path v, out only (a sixth of the GEMM work, most of f16x2a's accuracy: DESIGN.md I.2);
// 4 = f16x3: every matrix as hi | lo | hi (factor 3) against ACTIVATIONS laid out hi | hi | lo — weights and GEMM inputs both
// to ~20 bits: the mode that holds 1e-3 on every output, contact logits included
struct SplitPlan {
    int qk, v, o, ffn;
};
static inline SplitPlan split_plan(const esmk_model* m) {
    switch (m->cfg.weight_split) {
        case 1: return {2, 2, 2, 2};
        case 2: return {2, 2, 2, 1};
        case 3: return {1, 2, 2, 1};
        case 4: return {3, 3, 3, 3};
        default: return {1, 1, 1, 1};
    }
}
static inline bool split_x3(const esmk_model* m) { return m->cfg.weight_split == 4; }
// factor of one layer GEMM by its profiler class and epilogue (the v projection is the EPI_V_T launch of class PC_GEMM_QKV)
static inline int split_factor(const esmk_model* m, int cls, int epi);

// ---- per-kernel-class timing (esmk_profile_begin / _end): classes of both engines ---------------------------
enum {
    PC_EMBED = 0, PC_LAYERNORM, PC_GEMM_QKV, PC_ATTENTION, PC_ATTN_PROBS, PC_GEMM_OUT, PC_GEMM_FC1,
    PC_GEMM_FC2, PC_COPY, PC_LM_DENSE, PC_LM_LOGITS, PC_CONTACTS,
    PC_MSA_ROW_SCORES, PC_MSA_ROW_SOFTMAX, PC_MSA_ROW_CTX, PC_MSA_COL_ATTN, PC_LN_STATS, PC_COUNT
};
static const char* const kProfNames[PC_COUNT] = {
    "embed", "layernorm", "gemm_qkv_rope", "attention", "attention_probs", "gemm_out_proj",
    "gemm_fc1_gelu", "gemm_fc2", "repr_copy", "lm_head_dense", "lm_head_logits", "contacts",
    "msa_row_scores", "msa_row_softmax", "msa_row_context", "msa_col_attention",
    // LayerNorm fold: the row-statistics entry pass and the per-LayerNorm finalize launches (tiny; the LayerNorm passes
    // themselves are GEMM epilogue work) — "layernorm" keeps the standalone LayerNorm kernel (with the fold: the final one)
    "ln_fold_stats"};

static inline int split_factor(const esmk_model* m, int cls, int epi) {
    const SplitPlan s = split_plan(m);
    if (cls == PC_GEMM_QKV) return epi == esmk::EPI_V_T ? s.v : s.qk;
    if (cls == PC_GEMM_OUT) return s.o;
    return s.ffn;
}

// Brackets one launch with two events on the launch stream when profiling is enabled.
struct ProfScope {
    esmk_model* m;
    hipStream_t st;
    bool on;
    ProfScope(esmk_model* m_, hipStream_t st_, int cls, double flops, double bytes)
        : m(m_), st(st_), on(m_->prof_on) {
        if (!on) return;
        esmk_model::ProfRec r{cls, nullptr, nullptr, flops, bytes};
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) {
            on = false;
            return;
        }
        (void)hipEventRecord(r.a, st);
        m->prof.push_back(r);
    }
    ~ProfScope() {
        if (on) (void)hipEventRecord(m->prof.back().b, st);
    }
};

enum FoldBits : uint32_t {
    FB_LN1G = 1, FB_LN1B = 2, FB_LN2G = 4, FB_LN2B = 8, FB_WQ = 16, FB_WK = 32, FB_WV = 64, FB_W1 = 128,
    FB_ALL_W = FB_WQ | FB_WK | FB_WV | FB_W1
};

namespace esmk_host {
// "unit" rotary tables (cos = 1, sin = 0) for models without RoPE (MSA Transformer, ESM-1b)
int ensure_unit_rope(esmk_model* m, int T, hipStream_t st);
// packed image layout of the MSA Transformer (engine_msa.hip)
void plan_packed_msa(esmk_model* m);
}  // namespace esmk_host
