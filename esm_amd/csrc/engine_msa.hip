// engine_msa.hip — host side of the MSA Transformer path of libesmk.so (declared in include/esmk.h):
// packed parameter layout, workspace planning and the launch sequence that replaces MSATransformer.forward
// (reference esm/model/msa_transformer.py:146-220).  Kernels live in gemm8.hip, attention.hip, elementwise.hip.
#include "engine_internal.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <string>

using namespace esmk;
using namespace esmk_host;

namespace esmk_host {
void plan_packed_msa(esmk_model* m) {
    const size_t os = op_size(m->cfg.operand_dtype);
    const size_t E = m->E, F = m->F, V = m->V;
    // f16x2: every layer matrix as [rows, 2 cols] (hi | lo K tiles); f16x2a: the attention projections only
    const size_t ws = m->cfg.weight_split ? 2 : 1;
    const SplitPlan sp = split_plan(m);
    Carve c;
    m->embed_f32 = c.take(V * E * 4);
    m->embed_op = c.take(V * E * os);
    m->pos_emb = c.take((size_t)m->npos * E * 4);
    m->msa_pos = c.take((size_t)1024 * E * 4);
    m->lnb_g = c.take(E * 4);
    m->lnb_b = c.take(E * 4);
    m->fin_g = c.take(E * 4);
    m->fin_b = c.take(E * 4);
    m->lm_w = c.take(E * E * os);
    m->lm_w32 = c.take(ws == 2 ? E * E * 4 : 0);
    m->lm_b = c.take(E * 4);
    m->lm_lng = c.take(E * 4);
    m->lm_lnb = c.take(E * 4);
    m->lm_bias = c.take(V * 4);
    m->ct_w = c.take((size_t)m->L * m->H * 4);
    m->ct_b = c.take(4);
    m->mlayer.resize(m->L);
    auto attn = [&](AttnOff& a) {
        a.wqkv = c.take(E * E * os * (2 * sp.qk + sp.v));
        a.bqkv = c.take(3 * E * 4);
        a.wo = c.take(E * E * os * sp.o);
        a.bo = c.take(E * 4);
        a.lng = c.take(E * 4);
        a.lnb = c.take(E * 4);
    };
    for (int l = 0; l < m->L; ++l) {
        MsaLayerOff& o = m->mlayer[l];
        attn(o.row);
        attn(o.col);
        o.w1 = c.take(F * E * os * sp.ffn);
        o.b1 = c.take(F * 4);
        o.w2 = c.take(E * F * os * sp.ffn);
        o.b2 = c.take(E * 4);
        o.flng = c.take(E * 4);
        o.flnb = c.take(E * 4);
    }
    m->packed_bytes = c.off;
}
}  // namespace esmk_host

// =============================================================================================
// MSA Transformer (reference esm/model/msa_transformer.py, esm/axial_attention.py)
// =============================================================================================
namespace {

struct MsaWorkspace {
    size_t keep, col_fill, any_pad, x, h, big, scores, probs, lse, ct_scratch, total;
    size_t q, k, vt;  // inside big
    int Cp, Rp;
    int row_slices;  // the tied-score GEMM sums over R rows in this many K slices (partial maps summed by the softmax)
};

// Tied row attention scores: per (b, head) a [C, C] map contracted over R * 64, i.e. only B * H * ceil(C/256)^2
// output tiles (108 for one 128 x 513 MSA) with a very long K loop: less than half of the 256 CUs would work.
// The contraction is cut into `S` slices of R / S rows (S | R), each slice a batch entry of the same persistent
// GEMM writing its own fp32 partial map; msa_row_softmax_kernel adds the slices in index order (deterministic).
int row_score_slices(int B, int H, int R, int C) {
    const int tiles = B * H * ((C + 255) / 256) * ((C + 255) / 256);
    int best = 1;
    for (int s = 2; s <= 8 && s * tiles <= 256 + tiles / 2; ++s)
        if (R % s == 0) best = s;
    return best;
}

MsaWorkspace plan_msa_workspace(const esmk_model* m, int B, int R, int C, uint32_t flags) {
    MsaWorkspace w{};
    const size_t os = op_size(m->cfg.operand_dtype);
    const size_t N = (size_t)B * R * C, E = m->E, F = m->F, H = m->H;
    w.Cp = (C + 63) / 64 * 64;
    w.Rp = (R + 63) / 64 * 64;
    Carve c;
    w.keep = c.take(N * 4);
    w.col_fill = c.take(N * 4);
    w.any_pad = c.take(256);
    w.x = c.take(N * E * 4);
    w.h = c.take(N * E * os + 4096);
    const size_t qb = align_up(N * E * os + 4096);
    const size_t vt_row = (size_t)B * H * R * 64 * w.Cp * os;         // [B,H,R,64,Cp]
    const size_t vt_col = (size_t)B * C * H * 64 * w.Rp * os;         // [B*C,H,64,Rp]
    size_t big = 2 * qb + align_up(std::max(vt_row, vt_col));
    big = std::max(big, N * F * os);
    big = std::max(big, N * E * 4);
    w.big = c.take(big);
    w.q = w.big;
    w.k = w.big + qb;
    w.vt = w.big + 2 * qb;
    w.row_slices = row_score_slices(B, (int)H, R, C);
    w.scores = c.take((size_t)w.row_slices * B * H * C * w.Cp * 4);
    w.probs = c.take((size_t)B * H * C * w.Cp * os);
    w.lse = c.take((flags & ESMK_OUT_COL_ATTN) ? (size_t)B * C * H * R * 4 : 0);
    const int S = C - (m->cfg.prepend_bos ? 1 : 0) - (m->cfg.append_eos ? 1 : 0);
    w.ct_scratch =
        c.take((flags & ESMK_OUT_CONTACTS) ? (size_t)B * m->L * m->H * (size_t)(S > 0 ? S + 1 : 1) * 4 : 0);
    w.total = c.off;
    return w;
}

}  // namespace

extern "C" {

int esmk_msa_create(const esmk_msa_config* cfg, esmk_model** out) {
    if (!cfg || !out) return fail("esmk_msa_create: null argument");
    if (cfg->num_layers <= 0 || cfg->embed_dim <= 0 || cfg->num_heads <= 0 || cfg->ffn_dim <= 0 ||
        cfg->vocab <= 0 || cfg->num_positions <= 0)
        return fail("esmk_msa_create: non-positive dimension");
    if (cfg->embed_dim % cfg->num_heads != 0 || cfg->embed_dim / cfg->num_heads != 64)
        return fail("esmk_msa_create: head_dim must be 64 for the gfx950 attention kernels");
    if (cfg->embed_dim % 64 != 0 || cfg->ffn_dim % 64 != 0)
        return fail("esmk_msa_create: embed_dim and ffn_dim must be multiples of 64");
    if (cfg->operand_dtype != ESMK_F16 && cfg->operand_dtype != ESMK_BF16)
        return fail("esmk_msa_create: operand_dtype must be ESMK_F16 or ESMK_BF16");
    if (cfg->weight_split < 0 || cfg->weight_split > 3) return fail("esmk_msa_create: weight_split must be 0 (off), 1 (f16x2), 2 (f16x2a) or 3 (f16x2v)");
    if (cfg->weight_split != 0 && cfg->operand_dtype != ESMK_F16)
        return fail("esmk_msa_create: weight_split (precision modes f16x2 / f16x2a / f16x2v) needs operand_dtype ESMK_F16");
    esmk_model* m = new esmk_model();
    memset(&m->cfg, 0, sizeof(m->cfg));
    m->cfg.num_layers = cfg->num_layers;
    m->cfg.embed_dim = cfg->embed_dim;
    m->cfg.num_heads = cfg->num_heads;
    m->cfg.ffn_dim = cfg->ffn_dim;
    m->cfg.vocab = cfg->vocab;
    m->cfg.pad_idx = cfg->pad_idx;
    m->cfg.mask_idx = cfg->mask_idx;
    m->cfg.cls_idx = cfg->cls_idx;
    m->cfg.eos_idx = cfg->eos_idx;
    m->cfg.prepend_bos = cfg->prepend_bos;
    m->cfg.append_eos = cfg->append_eos;
    m->cfg.operand_dtype = cfg->operand_dtype;
    m->cfg.weight_split = cfg->weight_split;
    m->L = cfg->num_layers;
    m->E = cfg->embed_dim;
    m->H = cfg->num_heads;
    m->F = cfg->ffn_dim;
    m->V = cfg->vocab;
    m->D = 64;
    m->EA = m->E;
    m->Kp = m->E;
    m->is_msa = true;
    m->npos = cfg->num_positions;
    m->has_msa_pos = cfg->has_msa_position_embedding;
    plan_packed_msa(m);
    *out = m;
    return 0;
}

int esmk_msa_workspace_bytes(const esmk_model* m, int B, int R, int C, uint32_t out_flags, size_t* bytes) {
    if (!m || !bytes || !m->is_msa) return fail("esmk_msa_workspace_bytes: not an MSA model handle");
    if (B <= 0 || R <= 0 || C <= 0) return fail("esmk_msa_workspace_bytes: B, R, C must be positive");
    if ((long long)B * R * C > ESMK_MAX_ROWS) return fail("esmk_msa_workspace_bytes: B*R*C exceeds 2^24 rows");
    *bytes = plan_msa_workspace(m, B, R, C, out_flags).total;
    return 0;
}

int esmk_msa_forward(esmk_model* m, const void* packed_dev, const int64_t* tokens_dev, int B, int R, int C,
                     const int32_t* repr_layers, int n_repr, void* const* repr_out_dev, uint32_t out_flags,
                     void* logits_out_dev, void* row_attn_out_dev, void* col_attn_out_dev,
                     void* contacts_out_dev, void* workspace_dev, size_t workspace_bytes, void* stream) {
    if (!m || !m->is_msa) return fail("esmk_msa_forward: not an MSA model handle");
    if (!packed_dev || !tokens_dev || !workspace_dev) return fail("esmk_msa_forward: null argument");
    if (B <= 0 || R <= 0 || C <= 0) return fail("esmk_msa_forward: B, R, C must be positive");
    if ((long long)B * R * C > ESMK_MAX_ROWS) return fail("esmk_msa_forward: B*R*C exceeds 2^24 rows");
    if (R > 1024 && m->has_msa_pos)
        return fail("esmk_msa_forward: MSA position embedding covers a depth of 1024 alignments");  // msa_transformer.py:160-164
    if (C > 1024) return fail("esmk_msa_forward: more than 1024 columns are not supported");
    if (C > m->npos - m->cfg.pad_idx - 1)
        return fail("esmk_msa_forward: sequence length above the maximum of the positional embedding");  // modules.py:243-247
    if (out_flags & (ESMK_OUT_REPR_LOWP | ESMK_OUT_ATTN_LOWP))
        return fail("esmk_msa_forward: outputs are fp32 (ESMK_OUT_*_LOWP is an esmk_forward flag)");
    const bool want_logits = out_flags & ESMK_OUT_LOGITS;
    const bool want_contacts = out_flags & ESMK_OUT_CONTACTS;
    const bool want_attn = (out_flags & ESMK_OUT_ATTN) || want_contacts;
    if (want_logits && !logits_out_dev) return fail("esmk_msa_forward: logits buffer missing");
    if (want_attn && !row_attn_out_dev) return fail("esmk_msa_forward: row attention buffer missing");
    if (want_contacts && !contacts_out_dev) return fail("esmk_msa_forward: contacts buffer missing");
    const bool want_col = out_flags & ESMK_OUT_COL_ATTN;
    if (want_col && !col_attn_out_dev) return fail("esmk_msa_forward: column attention buffer missing");
    for (int i = 0; i < n_repr; ++i)
        if (repr_layers[i] < 0 || repr_layers[i] > m->L || !repr_out_dev[i])
            return fail("esmk_msa_forward: bad repr layer request");
    const MsaWorkspace w = plan_msa_workspace(m, B, R, C, out_flags);
    if (workspace_bytes < w.total) return fail("esmk_msa_forward: workspace too small");

    hipStream_t st = (hipStream_t)stream;
    const int op = m->cfg.operand_dtype;
    const size_t os = op_size(op);
    const int N = B * R * C, E = m->E, F = m->F, H = m->H, L = m->L, Cp = w.Cp, Rp = w.Rp;
    char* ws = (char*)workspace_dev;
    const char* pk = (const char*)packed_dev;
    float* keep = (float*)(ws + w.keep);
    float* col_fill = (float*)(ws + w.col_fill);
    int* any_pad = (int*)(ws + w.any_pad);
    float* x = (float*)(ws + w.x);
    void* h = ws + w.h;
    void* q = ws + w.q;
    void* k = ws + w.k;
    void* vt = ws + w.vt;
    void* ffn = ws + w.big;
    float* g32 = (float*)(ws + w.big);
    float* scores = (float*)(ws + w.scores);
    void* probs = ws + w.probs;
    if (ensure_unit_rope(m, std::max(R, C), st)) return 1;

    const double NE = (double)N * E;
    auto repr_copy = [&](int layer, const float* src) -> int {
        for (int i = 0; i < n_repr; ++i)
            if (repr_layers[i] == layer) {
                ProfScope ps(m, st, PC_COPY, 0, 8 * NE);
                ESMK_TRY(launch_copy_f32(src, (float*)repr_out_dev[i], (size_t)N * E, st));
            }
        return 0;
    };
    // algorithmic work of one (batched) GEMM launch: operands read once, result written once
    auto gemm = [&](int cls, const GemmArgs& a, int epi, double out_bytes_per_elem) -> int {
        const double z = a.batch > 0 ? a.batch : 1;
        const double fl = 2.0 * z * a.M * (double)(a.n_valid ? a.n_valid : a.N) * a.K;
        const double by = z * (((double)a.M * a.K + (double)a.N * a.K) * os + (double)a.M * a.N * out_bytes_per_elem);
        ProfScope ps(m, st, cls, fl, by);
        ESMK_TRY(launch_gemm(a, epi, op, st));
        return 0;
    };
    // a GEMM against a weight matrix of the layer stack: with split weights (f16x2) the same kernel runs over
    // the [N, 2K] hi | lo image, the activations' K tile kt / 2 meeting W_hi (kt even) and W_lo (kt odd)
    const int wsf = split_plan(m).qk;            // q / k weights: the v rows of the image start behind 2 E rows of this length
    auto wgemm = [&](int cls, GemmArgs a, int epi, double out_bytes_per_elem) -> int {
        if (split_factor(m, cls, epi) == 1) return gemm(cls, a, epi, out_bytes_per_elem);
        const double fl = 2.0 * a.M * (double)a.N * a.K;
        const double by = ((double)a.M * a.K + 2.0 * a.N * a.K) * os + (double)a.M * a.N * out_bytes_per_elem;
        a.a_row_bytes = (long long)a.K * (long long)os;
        a.a_kt_repeat = 1;
        a.K *= 2;
        ProfScope ps(m, st, cls, fl, by);
        ESMK_TRY(launch_gemm(a, epi, op, st));
        return 0;
    };
    auto lnorm = [&](const float* in, size_t go, size_t bo, void* y, float* y32, LnExtra ex) -> int {
        ProfScope ps(m, st, PC_LAYERNORM, 8 * NE, NE * (4 + (y ? os : 0) + (y32 ? 4 : 0)));
        ESMK_TRY(launch_layernorm_ex(in, (const float*)(pk + go), (const float*)(pk + bo), y, y32, N, E, op, ex, st));
        return 0;
    };

    static const bool env_fork = [] {
        const char* e = getenv("ESMK_QKV_FORK");
        return e != nullptr && atoi(e) != 0;
    }();
    const bool fork_v = env_fork && !m->prof_on && !m->cfg.weight_split;
    if (fork_v && !m->side_stream) {
        ESMK_TRY(hipStreamCreateWithFlags(&m->side_stream, hipStreamNonBlocking));
        ESMK_TRY(hipEventCreateWithFlags(&m->ev_fork, hipEventDisableTiming));
        ESMK_TRY(hipEventCreateWithFlags(&m->ev_join, hipEventDisableTiming));
    }

    // msa_transformer.py:152-172: token + position + MSA-row embeddings, LayerNorm, pads zeroed
    {
        ProfScope ps(m, st, PC_EMBED, 0, (double)N * 8 + 4 * NE);
        ESMK_TRY(launch_msa_embed(tokens_dev, (const float*)(pk + m->embed_f32), (const float*)(pk + m->pos_emb),
                                  m->has_msa_pos ? (const float*)(pk + m->msa_pos) : nullptr, x, keep, col_fill,
                                  any_pad, B, R, C, E, m->V, m->cfg.pad_idx, m->npos, st));
    }
    {
        LnExtra ex;
        ex.row_keep = keep;
        if (lnorm(x, m->lnb_g, m->lnb_b, nullptr, x, ex)) return 1;
    }
    if (repr_copy(0, x)) return 1;

    // q/k/v projections of one axial attention block on `rows` = N rows grouped in sequences of T tokens
    auto qkv = [&](const AttnOff& a, int T, int Tp, float scaling, const float* row_keep, int vt_rows) -> int {
        GemmArgs g;
        g.A = h;
        g.W = pk + a.wqkv;
        g.bias = (const float*)(pk + a.bqkv);
        g.M = N;
        g.N = 2 * E;
        g.K = E;
        g.q = q;
        g.k = k;
        g.vt = vt;
        g.cos = m->d_ucos;
        g.sin = m->d_usin;
        g.T = T;
        g.H = H;
        g.E = E;
        g.Tp = Tp;
        g.scaling = scaling;
        g.row_keep = row_keep;
        GemmArgs gv = g;
        gv.row_keep = nullptr;
        gv.W = pk + a.wqkv + (size_t)2 * E * E * os * wsf;
        gv.bias = (const float*)(pk + a.bqkv) + 2 * E;
        gv.N = E;
        gv.vt_rows = vt_rows;
        if (fork_v) {  // v next to q/k on the library's side stream (esmk_model::side_stream, ESMK_QKV_FORK)
            ESMK_TRY(hipEventRecord(m->ev_fork, st));
            ESMK_TRY(hipStreamWaitEvent(m->side_stream, m->ev_fork, 0));
            ESMK_TRY(launch_gemm(gv, EPI_V_T, op, m->side_stream));
            ESMK_TRY(hipEventRecord(m->ev_join, m->side_stream));
            if (gemm(PC_GEMM_QKV, g, EPI_QKV_ROPE, os)) return 1;
            ESMK_TRY(hipStreamWaitEvent(st, m->ev_join, 0));
            return 0;
        }
        if (wgemm(PC_GEMM_QKV, g, EPI_QKV_ROPE, os)) return 1;
        return wgemm(PC_GEMM_QKV, gv, EPI_V_T, os);
    };
    auto out_proj = [&](const AttnOff& a, int map_R, int map_C) -> int {
        GemmArgs g;
        g.A = h;
        g.W = pk + a.wo;
        g.bias = (const float*)(pk + a.bo);
        g.out = x;
        g.M = N;
        g.N = E;
        g.K = E;
        g.rowmap_R = map_R;
        g.rowmap_C = map_C;
        return wgemm(PC_GEMM_OUT, g, EPI_RESID_F32, 8);
    };

    for (int l = 0; l < L; ++l) {
        const MsaLayerOff& o = m->mlayer[l];
        // ---- tied row attention (axial_attention.py:75-130; NormalizedResidualBlock modules.py:376-392) ----
        if (lnorm(x, o.row.lng, o.row.lnb, h, nullptr, LnExtra())) return 1;
        if (Cp != C) ESMK_TRY(hipMemsetAsync(vt, 0, (size_t)B * H * R * 64 * Cp * os, st));
        // sequences = MSA rows (b,r) of C tokens; q scaled by d^-1/2 / sqrt(R) (axial_attention.py:36-38)
        if (qkv(o.row, C, Cp, (1.0f / sqrtf(64.0f)) / sqrtf((float)R), keep, R)) return 1;
        {   // scores[b,h,i,j] = sum_{r,d} q[r,i,b,h,d] k[r,j,b,h,d]   (axial_attention.py:90): K tile r of
            // the batched GEMM is the [C,64] matrix q[(b,r),h] (row stride 128 B)
            // slice s of MSA b is batch entry zo = b * S + s: R / S rows further down q / k, its own output map
            const int S = w.row_slices, Rs = R / S;
            GemmArgs g;
            g.A = q;
            g.W = k;
            g.out = scores;
            g.M = C;
            g.N = Cp;
            g.n_valid = C;
            g.K = Rs * 64;
            g.ldc = Cp;
            g.a_row_bytes = g.w_row_bytes = 128;
            g.a_kt_bytes = g.w_kt_bytes = (long long)H * C * 64 * os;
            g.batch = B * S * H;
            g.batch_inner = H;
            g.a_bo = g.w_bo = (long long)Rs * H * C * 64 * os;
            g.a_bi = g.w_bi = (long long)C * 64 * os;
            g.o_bo = (long long)H * C * Cp * 4;
            g.o_bi = (long long)C * Cp * 4;
            if (gemm(PC_MSA_ROW_SCORES, g, EPI_STORE_F32, 4)) return 1;
        }
        {
            const double sc = (double)B * H * C * C;
            ProfScope ps(m, st, PC_MSA_ROW_SOFTMAX, 0, sc * (4 + os + (want_attn ? 4 : 0)));
            ESMK_TRY(launch_msa_row_softmax(scores, keep, any_pad, probs,
                                            want_attn ? (float*)row_attn_out_dev : nullptr, B, H, R, C, Cp, l, L, op,
                                            st, w.row_slices));
        }
        {   // context[r,i,b,h,:] = sum_j probs[h,b,i,j] v[r,j,b,h,:]   (axial_attention.py:111)
            GemmArgs g;
            g.A = probs;
            g.W = vt;
            g.out = h;
            g.M = C;
            g.N = R * 64;
            g.K = Cp;
            g.ldc = E;
            g.a_row_bytes = g.w_row_bytes = (long long)Cp * os;
            g.batch = B * H;
            g.batch_inner = H;
            g.a_bo = (long long)H * C * Cp * os;
            g.a_bi = (long long)C * Cp * os;
            g.w_bo = (long long)H * R * 64 * Cp * os;
            g.w_bi = (long long)R * 64 * Cp * os;
            g.ctx_R = R;
            g.ctx_C = C;
            if (gemm(PC_MSA_ROW_CTX, g, EPI_MSA_CTX, os)) return 1;
        }
        if (out_proj(o.row, 0, 0)) return 1;

        // ---- column attention (axial_attention.py:185-239): every MSA column (b,c) is a sequence of R rows;
        // the normalised rows are written in (b,c,r) order so the ESM-2 attention path applies unchanged ----
        {
            LnExtra ex;
            ex.map_R = R;
            ex.map_C = C;
            if (lnorm(x, o.col.lng, o.col.lnb, h, nullptr, ex)) return 1;
        }
        if (Rp != R) ESMK_TRY(hipMemsetAsync(vt, 0, (size_t)B * C * H * 64 * Rp * os, st));
        // log2(e) folded into the q scale: the flash / map kernels work on log2-domain scores (attention.hip)
        if (qkv(o.col, R, Rp, 1.4426950408889634f / sqrtf(64.0f), nullptr, 0)) return 1;
        float* lse = want_col ? (float*)(ws + w.lse) : nullptr;
        {
            ProfScope ps(m, st, PC_MSA_COL_ATTN, 4.0 * N * (double)R * E, 4 * NE * os);
            ESMK_TRY(launch_attention_fill(q, k, vt, col_fill, any_pad, h, lse, B * C, H, R, Rp, op, st));
        }
        if (want_col) {
            ProfScope ps(m, st, PC_ATTN_PROBS, 2.0 * N * (double)R * E, 2 * NE * os + 4.0 * N * R * H);
            ESMK_TRY(launch_attention_probs_msa(q, k, lse, col_fill, any_pad, (float*)col_attn_out_dev, B, C, H, R, l,
                                                L, op, st));
        }
        if (out_proj(o.col, R, C)) return 1;

        // ---- feed forward (modules.py:395-418) ----
        if (lnorm(x, o.flng, o.flnb, h, nullptr, LnExtra())) return 1;
        {
            GemmArgs g;
            g.A = h;
            g.W = pk + o.w1;
            g.bias = (const float*)(pk + o.b1);
            g.out = ffn;
            g.M = N;
            g.N = F;
            g.K = E;
            if (wgemm(PC_GEMM_FC1, g, EPI_GELU_T, os)) return 1;
            g = GemmArgs();
            g.A = ffn;
            g.W = pk + o.w2;
            g.bias = (const float*)(pk + o.b2);
            g.out = x;
            g.M = N;
            g.N = E;
            g.K = F;
            if (wgemm(PC_GEMM_FC2, g, EPI_RESID_F32, 8)) return 1;
        }
        if (l + 1 < L && repr_copy(l + 1, x)) return 1;  // msa_transformer.py:197-198
    }

    // msa_transformer.py:200-206: final LayerNorm (representation L is the normalised stream), LM head
    float* rep_last = nullptr;
    bool wants_last = false;
    for (int i = 0; i < n_repr; ++i)
        if (repr_layers[i] == L) {
            wants_last = true;
            if (!rep_last) rep_last = (float*)repr_out_dev[i];
        }
    if (want_logits || wants_last) {
        if (lnorm(x, m->fin_g, m->fin_b, want_logits ? h : nullptr, rep_last, LnExtra())) return 1;
        for (int i = 0; i < n_repr; ++i)
            if (repr_layers[i] == L && repr_out_dev[i] != rep_last)
                ESMK_TRY(launch_copy_f32(rep_last, (float*)repr_out_dev[i], (size_t)N * E, st));
    }
    if (want_logits && m->cfg.weight_split && E % 32 == 0) {
        // f16x2: the head (modules.py:308-314) in fp32 on the exact-fp32 MFMA path, as in esmk_forward
        float* a32 = rep_last != nullptr ? rep_last : g32;
        if (a32 == g32 && lnorm(x, m->fin_g, m->fin_b, nullptr, g32, LnExtra())) return 1;
        {
            ProfScope ps(m, st, PC_LM_DENSE, 2.0 * N * (double)E * E, (2.0 * NE + (double)E * E) * 4);
            ESMK_TRY(launch_gemm32(a32, E, (const float*)(pk + m->lm_w32), (const float*)(pk + m->lm_b), x, E, N, E, E, true, st));
        }
        if (lnorm(x, m->lm_lng, m->lm_lnb, nullptr, g32, LnExtra())) return 1;  // x (the residual stream) is dead: dense output
        {
            ProfScope ps(m, st, PC_LM_LOGITS, 2.0 * N * (double)E * m->V, (NE + (double)m->V * E + (double)N * m->V) * 4);
            ESMK_TRY(launch_gemm32(g32, E, (const float*)(pk + m->embed_f32), (const float*)(pk + m->lm_bias),
                                   (float*)logits_out_dev, m->V, N, m->V, E, false, st));
        }
    } else if (want_logits) {  // modules.py:308-314
        GemmArgs g;
        g.A = h;
        g.W = pk + m->lm_w;
        g.bias = (const float*)(pk + m->lm_b);
        g.out = g32;
        g.M = N;
        g.N = E;
        g.K = E;
        if (gemm(PC_LM_DENSE, g, EPI_GELU_F32, 4)) return 1;
        if (lnorm(g32, m->lm_lng, m->lm_lnb, h, nullptr, LnExtra())) return 1;
        g = GemmArgs();
        g.A = h;
        g.W = pk + m->embed_op;
        g.bias = (const float*)(pk + m->lm_bias);
        g.out = logits_out_dev;
        g.M = N;
        g.N = m->V;
        g.K = E;
        if (gemm(PC_LM_LOGITS, g, EPI_STORE_F32, 4)) return 1;
    }
    if (want_contacts) {  // msa_transformer.py:215-217 -> modules.py:338-357 on the row attentions
        // the contact head reads tokens only for the <eos> mask, which the MSA alphabet does not append
        ProfScope ps(m, st, PC_CONTACTS, 0, 2.0 * 4 * B * (double)L * H * C * C);
        ESMK_TRY(launch_contacts((const float*)row_attn_out_dev, tokens_dev, (const float*)(pk + m->ct_w),
                                 (const float*)(pk + m->ct_b), (float*)(ws + w.ct_scratch),
                                 (float*)contacts_out_dev, B, L * H, C, m->cfg.eos_idx, m->cfg.prepend_bos,
                                 m->cfg.append_eos, st));
    }
    return 0;
}

}  // extern "C"
