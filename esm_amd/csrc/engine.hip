// engine.hip — the C ABI of libesmk.so (declared in include/esmk.h): parameter packing,
// workspace planning and the launch sequence that replaces ESM2.forward
// (reference esm/model/esm2.py:77-144).  Host code only; every kernel lives in gemm.hip,
// attention.hip and elementwise.hip.
#include "engine_internal.h"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <string>
#include <vector>

using namespace esmk;
using namespace esmk_host;

static constexpr float kLog2e = 1.4426950408889634f;
// LayerNorm fold for handles created with esmk_config::ln_fold == 0 and no ESMK_LN_FOLD in the environment:
// ON since round 5 wherever the configuration supports it (plain fp16 / bf16 operands, head_dim <= 64) — faster at every
// batch size (B = 64 + 1.1 %, B = 4 + 6.5 %) and on the fp16-operand floor numerically, like the plain mode
static constexpr bool kLnFoldDefault = true;

namespace {
thread_local std::string g_err;
}

namespace esmk_host {
int fail(const char* what, hipError_t e) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return 1;
}
int fail(const std::string& msg) {
    g_err = msg;
    return 1;
}
}  // namespace esmk_host

namespace {

void plan_packed(esmk_model* m) {
    const size_t os = op_size(m->cfg.operand_dtype);
    const size_t E = m->E, F = m->F, V = m->V, EA = m->EA, Kp = m->Kp;
    // f16x2: every layer matrix as [rows, 2 cols] (hi | lo K tiles); f16x2a (weight_split 2): the attention projections only
    const size_t ws = m->cfg.weight_split ? 2 : 1;
    const SplitPlan sp = split_plan(m);
    Carve c;
    m->embed_f32 = c.take(V * E * 4);
    m->embed_op = c.take(V * Kp * os);
    m->fin_g = c.take(E * 4);
    m->fin_b = c.take(E * 4);
    m->lm_w = c.take(E * Kp * os);
    m->lm_w32 = c.take(ws == 2 ? E * E * 4 : 0);
    m->lm_b = c.take(E * 4);
    m->lm_lng = c.take(E * 4);
    m->lm_lnb = c.take(E * 4);
    m->lm_bias = c.take(V * 4);
    m->ct_w = c.take((size_t)m->L * m->H * 4);
    m->ct_b = c.take(4);
    if (m->cfg.num_positions > 0) m->pos_emb = c.take((size_t)m->cfg.num_positions * E * 4);
    if (m->cfg.ln_before) {
        m->lnb_g = c.take(E * 4);
        m->lnb_b = c.take(E * 4);
    }
    m->layer.resize(m->L);
    for (int l = 0; l < m->L; ++l) {
        LayerOff& o = m->layer[l];
        o.wqkv = c.take(EA * Kp * os * (2 * sp.qk + sp.v));   // q, k rows | v rows (each block with its own row length)
        o.bqkv = c.take(3 * EA * 4);
        o.wo = c.take(E * EA * os * sp.o);
        o.bo = c.take(E * 4);
        o.w1 = c.take(F * Kp * os * sp.ffn);
        o.b1 = c.take(F * 4);
        o.w2 = c.take(E * F * os * sp.ffn);
        o.b2 = c.take(E * 4);
        o.ln1g = c.take(E * 4);
        o.ln1b = c.take(E * 4);
        o.ln2g = c.take(E * 4);
        o.ln2b = c.take(E * 4);
        if (m->fold) {
            o.bqkv2 = c.take(3 * EA * 4);
            o.b12 = c.take(F * 4);
        }
    }
    m->packed_bytes = c.off;
}

struct Workspace {
    size_t scale, key_bias, seq_info, keep, x, h, big, lse, ct_scratch, total;
    size_t h2 = 0, ln_part = 0, ln_mean = 0, ln_rstd = 0;  // LayerNorm fold
    int ln_parts = 0;
    size_t a3 = 0, ffn3 = 0;  // precision mode f16x3: hi | hi | lo operand rows (LayerNorm output / attention context; fc1 + GELU output)
    size_t ct_acc, ct_row, ct_col, ct_rowp, ct_colp, ct_wt;  // contacts without attention maps (contacts.hip)
    size_t q, k, vt;  // inside big
    int Tp;
    size_t row_pos, tables;  // token-packed batches only
};

// Token-packed batch (esmk_forward_packed): ONE row space of `rows` rows holding n_seg segments.  The layer
// stack sees it as B = 1, T = rows; only three kernels know about segments (token statistics, the rotary
// position in the q/k epilogue, the attention kernel's key range).
struct PackedCtx {
    int n_seg = 0, max_len = 0, n_items = 0;
    double sum_len2 = 0;                // sum of len^2: the attention work
    const int32_t* seg_host = nullptr;  // [n_seg][2] = (first row, length)
};
// query blocks of 128 rows: sum over segments of ceil(len / 128) <= rows / 128 + n_seg
inline size_t packed_items_bound(int n_seg, int rows) { return (size_t)rows / 128 + (size_t)n_seg; }

Workspace plan_workspace(const esmk_model* m, int B, int T, uint32_t flags, int packed_segs = 0) {
    Workspace w{};
    const size_t os = op_size(m->cfg.operand_dtype);
    const size_t N = (size_t)B * T, E = m->E, F = m->F, EA = m->EA, Kp = m->Kp;
    // packed: one spare (zeroed) key tile behind the rows, because a segment's last 64-key tile may start
    // anywhere and reach past the last row
    w.Tp = (T + 63) / 64 * 64 + (packed_segs > 0 ? 64 : 0);
    Carve c;
    w.scale = c.take((packed_segs > 0 ? N : (size_t)B) * 4);
    w.key_bias = c.take(N * 4);
    w.seq_info = c.take((size_t)B * 2 * 4);
    w.keep = c.take(m->cfg.num_positions > 0 ? N * 4 : 0);
    w.x = c.take(N * E * 4);
    w.h = c.take(N * std::max(Kp, EA) * os);
    if (m->fold) {
        // h: the raw rows of the residual stream in the operand dtype (A operand of q/k/v and fc1), h2: the attention
        // context; statistics per row, padded to whole 256-row tiles (the V^T epilogue loads four rows at a time)
        const size_t Np = (N + 255) / 256 * 256;
        w.ln_parts = (int)((E + 127) / 128);
        w.h2 = c.take(N * std::max(Kp, EA) * os);
        w.ln_part = c.take(Np * w.ln_parts * 2 * 4);
        w.ln_mean = c.take(Np * 4);
        w.ln_rstd = c.take(Np * 4);
    }
    if (split_x3(m)) {
        w.a3 = c.take(N * 3 * std::max(Kp, EA) * os);
        w.ffn3 = c.take(N * 3 * F * os);
    }
    const size_t qb = align_up(N * EA * os);
    const size_t vtb = align_up((size_t)B * EA * w.Tp * os);
    size_t big = 2 * qb + vtb;
    if (N * F * os > big) big = N * F * os;
    if (N * E * 4 > big) big = N * E * 4;
    w.big = c.take(big);
    w.q = w.big;
    w.k = w.big + qb;
    w.vt = w.big + 2 * qb;
    const bool attn = flags & (ESMK_OUT_ATTN | ESMK_OUT_CONTACTS);
    w.lse = c.take(attn ? (size_t)B * m->H * T * 4 : 0);
    const int S = T - (m->cfg.prepend_bos ? 1 : 0) - (m->cfg.append_eos ? 1 : 0);
    // ESMK_OUT_CONTACTS without ESMK_OUT_ATTN: no [B,L,H,T,T] tensor anywhere (contacts.hip)
    const bool fused_ct = (flags & ESMK_OUT_CONTACTS) && !(flags & ESMK_OUT_ATTN);
    w.ct_scratch = c.take(((flags & ESMK_OUT_CONTACTS) && !fused_ct)
                              ? (size_t)B * m->L * m->H * (size_t)(S > 0 ? S + 1 : 1) * 4 : 0);
    const size_t C = (size_t)m->L * m->H;
    w.ct_acc = c.take(fused_ct ? (size_t)contacts_head_groups(B, T, m->H, m->D == 128 ? 128 : 64) * B * T * T * 4 : 0);
    w.ct_row = c.take(fused_ct ? (size_t)B * C * T * 4 : 0);
    w.ct_col = c.take(fused_ct ? (size_t)B * C * T * 4 : 0);
    w.ct_rowp = c.take(fused_ct ? (size_t)B * ((T + 127) / 128) * m->H * T * 4 : 0);
    w.ct_colp = c.take(fused_ct ? (size_t)B * ((T + 31) / 32) * m->H * T * 4 : 0);
    w.ct_wt = c.take(fused_ct ? (size_t)B * C * 4 : 0);
    if (packed_segs > 0) {
        w.row_pos = c.take(N * 4);
        // [seg 2 n_seg][npad n_seg][work 4 n_items]
        w.tables = c.take(((size_t)3 * packed_segs + 4 * packed_items_bound(packed_segs, T)) * 4);
    }
    w.total = c.off;
    return w;
}

int ensure_rope(esmk_model* m, int T, hipStream_t st) {
    if (m->inv_freq.empty()) return fail("esmk_set_rope_inv_freq was not called");
    if (T <= m->rope_cap) return 0;
    int cap = 1024;
    while (cap < T) cap *= 2;
    const int half = m->D == 128 ? 64 : 32;
    if (m->d_cos) {
        ESMK_TRY(hipStreamSynchronize(st));
        ESMK_TRY(hipFree(m->d_cos));
        ESMK_TRY(hipFree(m->d_sin));
        m->d_cos = m->d_sin = nullptr;
        m->rope_cap = 0;
    }
    ESMK_TRY(hipMalloc(&m->d_cos, (size_t)cap * half * 4));
    ESMK_TRY(hipMalloc(&m->d_sin, (size_t)cap * half * 4));
    ESMK_TRY(launch_rope_table(m->d_inv_freq, m->d_cos, m->d_sin, cap, half, st));
    m->rope_cap = cap;
    return 0;
}

}  // namespace

__global__ void fill_f32_kernel(float* p, float v, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = v;
}

namespace esmk_host {
int ensure_unit_rope(esmk_model* m, int T, hipStream_t st) {
    if (T <= m->unit_cap) return 0;
    int cap = 1024;
    while (cap < T) cap *= 2;
    if (m->d_ucos) {
        ESMK_TRY(hipStreamSynchronize(st));
        ESMK_TRY(hipFree(m->d_ucos));
        ESMK_TRY(hipFree(m->d_usin));
        m->d_ucos = m->d_usin = nullptr;
        m->unit_cap = 0;
    }
    const size_t n = (size_t)cap * 64;  // row stride 32 (head_dim <= 64) or 64 (head_dim 128)
    ESMK_TRY(hipMalloc(&m->d_ucos, n * 4));
    ESMK_TRY(hipMalloc(&m->d_usin, n * 4));
    hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m->d_ucos, 1.0f, n);
    ESMK_TRY(hipGetLastError());
    ESMK_TRY(hipMemsetAsync(m->d_usin, 0, n * 4, st));
    m->unit_cap = cap;
    return 0;
}
}  // namespace esmk_host

namespace {

bool starts_with(const char* s, const char* p) { return strncmp(s, p, strlen(p)) == 0; }

size_t numel(const int64_t* shape, int ndim) {
    size_t n = 1;
    for (int i = 0; i < ndim; ++i) n *= (size_t)shape[i];
    return n;
}

}  // namespace

extern "C" {

const char* esmk_last_error(void) { return g_err.c_str(); }
#ifndef ESMK_SRC_HASH
#define ESMK_SRC_HASH "unhashed-build---"
#endif
// "esmk-src:" + the SHA-256 prefix of the sources (esm_amd/build.py: source_hash) — also read straight from the file
const char* esmk_version(void) { return "esmk 0.2 (gfx950) esmk-src:" ESMK_SRC_HASH; }

int esmk_create(const esmk_config* cfg, esmk_model** out) {
    if (!cfg || !out) return fail("esmk_create: null argument");
    if (cfg->num_layers <= 0 || cfg->embed_dim <= 0 || cfg->num_heads <= 0 || cfg->ffn_dim <= 0 ||
        cfg->vocab <= 0)
        return fail("esmk_create: non-positive dimension");
    if (cfg->embed_dim % cfg->num_heads != 0)
        return fail("esmk_create: embed_dim must be divisible by num_heads");
    if (cfg->operand_dtype != ESMK_F16 && cfg->operand_dtype != ESMK_BF16)
        return fail("esmk_create: operand_dtype must be ESMK_F16 or ESMK_BF16");
    const int d = cfg->embed_dim / cfg->num_heads;
    if ((d > 64 && d != 128) || d < 2 || (d & 1))
        return fail("esmk_create: head_dim " + std::to_string(d) +
                    " is not supported by the gfx950 attention kernels (128, or an even head_dim <= 64; smaller "
                    "heads are spread over 64 slots at pack time)");
    if (cfg->embed_dim % 8 != 0 || cfg->ffn_dim % 64 != 0)
        return fail("esmk_create: embed_dim must be a multiple of 8 and ffn_dim a multiple of 64");
    if (cfg->weight_split < 0 || cfg->weight_split > 4)
        return fail("esmk_create: weight_split must be 0 (off), 1 (f16x2), 2 (f16x2a), 3 (f16x2v) or 4 (f16x3)");
    if (cfg->weight_split == 4 && (cfg->embed_dim % 64 != 0 || cfg->embed_dim / cfg->num_heads != 64))
        return fail("esmk_create: weight_split 4 (f16x3) needs head_dim 64 and embed_dim % 64 == 0");
    if (cfg->weight_split != 0 && cfg->operand_dtype != ESMK_F16)
        return fail("esmk_create: weight_split (precision modes f16x2 / f16x2a / f16x2v / f16x3) needs operand_dtype ESMK_F16");
    // LayerNorm fold: explicit request, or the library default / ESMK_LN_FOLD where the configuration supports it
    const bool fold_ok = cfg->weight_split == 0 && d <= 64;
    if (cfg->ln_fold > 0 && !fold_ok)
        return fail("esmk_create: ln_fold needs plain fp16 / bf16 operands (no weight_split) and head_dim <= 64");
    bool fold = cfg->ln_fold > 0;
    if (cfg->ln_fold == 0 && fold_ok) {
        const char* e = getenv("ESMK_LN_FOLD");
        fold = e ? atoi(e) != 0 : kLnFoldDefault;
    }
    esmk_model* m = new esmk_model();
    m->fold = fold;
    m->fold_state.assign(cfg->num_layers, 0u);
    m->cfg = *cfg;
    m->L = cfg->num_layers;
    m->E = cfg->embed_dim;
    m->H = cfg->num_heads;
    m->F = cfg->ffn_dim;
    m->V = cfg->vocab;
    m->D = d;
    m->EA = m->H * (d == 128 ? 128 : 64);
    m->Kp = (m->E + 63) / 64 * 64;
    plan_packed(m);
    *out = m;
    return 0;
}

void esmk_destroy(esmk_model* m) {
    if (!m) return;
    if (m->d_cos) (void)hipFree(m->d_cos);
    if (m->d_sin) (void)hipFree(m->d_sin);
    if (m->d_inv_freq) (void)hipFree(m->d_inv_freq);
    if (m->d_ucos) (void)hipFree(m->d_ucos);
    if (m->d_usin) (void)hipFree(m->d_usin);
    if (m->pk_host) (void)hipHostFree(m->pk_host);
    if (m->pk_event) (void)hipEventDestroy(m->pk_event);
    if (m->ev_fork) (void)hipEventDestroy(m->ev_fork);
    if (m->ev_join) (void)hipEventDestroy(m->ev_join);
    if (m->side_stream) (void)hipStreamDestroy(m->side_stream);
    delete m;
}

int esmk_set_rope_inv_freq(esmk_model* m, const float* inv_freq_host, int n) {
    if (!m || !inv_freq_host) return fail("esmk_set_rope_inv_freq: null argument");
    if (n != m->D / 2) return fail("esmk_set_rope_inv_freq: expected head_dim/2 values");
    // 32 slots (64 for head_dim 128): slot i < d/2 carries frequency i, the rest rotate by angle 0 (they only
    // ever see zeros)
    const int slots = m->D == 128 ? 64 : 32;
    m->inv_freq.assign(slots, 0.f);
    for (int i = 0; i < n; ++i) m->inv_freq[i] = inv_freq_host[i];
    if (!m->d_inv_freq) ESMK_TRY(hipMalloc(&m->d_inv_freq, (size_t)slots * 4));
    ESMK_TRY(hipMemcpy(m->d_inv_freq, m->inv_freq.data(), (size_t)slots * 4, hipMemcpyHostToDevice));
    m->rope_cap = 0;  // tables are rebuilt on the next forward
    return 0;
}

int esmk_packed_bytes(const esmk_model* m, size_t* bytes) {
    if (!m || !bytes) return fail("esmk_packed_bytes: null argument");
    *bytes = m->packed_bytes;
    return 0;
}

int esmk_pack_weight(esmk_model* m, void* packed_dev, size_t packed_bytes, const char* key,
                     const void* src_dev, int src_dtype, const int64_t* shape, int ndim,
                     void* stream) {
    if (!m || !packed_dev || !key || !src_dev) return fail("esmk_pack_weight: null argument");
    if (packed_bytes < m->packed_bytes) return fail("esmk_pack_weight: packed buffer too small");
    hipStream_t st = (hipStream_t)stream;
    char* base = (char*)packed_dev;
    const int op = m->cfg.operand_dtype;
    const size_t os = op_size(op);
    const size_t E = m->E, F = m->F, V = m->V;
    const size_t n = numel(shape, ndim);
    auto put = [&](size_t off, int dst_dtype, size_t expect) -> int {
        if (n != expect)
            return fail(std::string("esmk_pack_weight: ") + key + " has " + std::to_string(n) +
                        " elements, expected " + std::to_string(expect));
        ESMK_TRY(launch_convert(src_dev, src_dtype, base + off, dst_dtype, n, st));
        return 0;
    };
    // [rows, cols] matrix into a destination with row stride ld; rmap / cmap spread head_dim-d heads over
    // 64 slots (elementwise.hip: head_pad_index).  The packed image is zero-initialised by the caller, so
    // padded rows / columns stay zero.
    const size_t EA = m->EA, Kp = m->Kp;
    const int hd = m->D;
    const int padmap = (hd < 64) ? 1 : 0;        // q, k, v, out_proj: heads spread over 64 slots
    const int qkmap = (hd != 64) ? 1 : 0;        // q, k additionally: slice order of 128-wide heads
    auto put2d = [&](size_t off, int dst_dtype, size_t rows, size_t cols, size_t ld, int rmap, int cmap) -> int {
        if (n != rows * cols)
            return fail(std::string("esmk_pack_weight: ") + key + " has " + std::to_string(n) +
                        " elements, expected " + std::to_string(rows * cols));
        ESMK_TRY(launch_convert2d(src_dev, src_dtype, base + off, dst_dtype, rows, cols, ld, rmap, cmap, hd, st));
        return 0;
    };
    // a matrix of the layer stack: plain operand-dtype image, or (f16x2) the hi | lo split image with rows of 2 ld
    const size_t ws = m->cfg.weight_split ? 2 : 1;
    const SplitPlan sp = split_plan(m);  // factor 2 = W_hi | W_lo image
    auto putw = [&](size_t off, size_t rows, size_t cols, size_t ld, int rmap, int cmap, size_t split = 0) -> int {
        if ((split ? split : ws) == 1) return put2d(off, op, rows, cols, ld, rmap, cmap);
        if (n != rows * cols)
            return fail(std::string("esmk_pack_weight: ") + key + " has " + std::to_string(n) +
                        " elements, expected " + std::to_string(rows * cols));
        ESMK_TRY(launch_convert2d_split(src_dev, src_dtype, base + off, rows, cols, ld, rmap, cmap, hd, st, (int)(split ? split : ws)));
        return 0;
    };
    if (!strcmp(key, "embed_tokens.weight")) {
        if (put(m->embed_f32, ESMK_DT_F32, V * E)) return 1;
        return put2d(m->embed_op, op, V, E, Kp, 0, 0);
    }
    if (m->is_msa) {
        if (!strcmp(key, "embed_positions.weight")) return put(m->pos_emb, ESMK_DT_F32, (size_t)m->npos * E);
        if (!strcmp(key, "msa_position_embedding")) return put(m->msa_pos, ESMK_DT_F32, (size_t)1024 * E);
        if (!strcmp(key, "emb_layer_norm_before.weight")) return put(m->lnb_g, ESMK_DT_F32, E);
        if (!strcmp(key, "emb_layer_norm_before.bias")) return put(m->lnb_b, ESMK_DT_F32, E);
        if (starts_with(key, "layers.")) {
            char* end = nullptr;
            const long l = strtol(key + 7, &end, 10);
            if (end == key + 7 || *end != '.' || l < 0 || l >= m->L)
                return fail(std::string("esmk_pack_weight: bad layer index in ") + key);
            const char* sub = end + 1;
            const MsaLayerOff& o = m->mlayer[l];
            const AttnOff* a = nullptr;
            if (starts_with(sub, "row_self_attention.")) { a = &o.row; sub += 19; }
            else if (starts_with(sub, "column_self_attention.")) { a = &o.col; sub += 22; }
            if (a) {
                if (!strcmp(sub, "layer.q_proj.weight")) return putw(a->wqkv, E, E, E, 0, 0, sp.qk);
                if (!strcmp(sub, "layer.k_proj.weight")) return putw(a->wqkv + E * E * os * sp.qk, E, E, E, 0, 0, sp.qk);
                if (!strcmp(sub, "layer.v_proj.weight")) return putw(a->wqkv + 2 * E * E * os * sp.qk, E, E, E, 0, 0, sp.v);
                if (!strcmp(sub, "layer.q_proj.bias")) return put(a->bqkv, ESMK_DT_F32, E);
                if (!strcmp(sub, "layer.k_proj.bias")) return put(a->bqkv + E * 4, ESMK_DT_F32, E);
                if (!strcmp(sub, "layer.v_proj.bias")) return put(a->bqkv + 2 * E * 4, ESMK_DT_F32, E);
                if (!strcmp(sub, "layer.out_proj.weight")) return putw(a->wo, E, E, E, 0, 0, sp.o);
                if (!strcmp(sub, "layer.out_proj.bias")) return put(a->bo, ESMK_DT_F32, E);
                if (!strcmp(sub, "layer_norm.weight")) return put(a->lng, ESMK_DT_F32, E);
                if (!strcmp(sub, "layer_norm.bias")) return put(a->lnb, ESMK_DT_F32, E);
                return 0;
            }
            if (!strcmp(sub, "feed_forward_layer.layer.fc1.weight")) return putw(o.w1, F, E, E, 0, 0, sp.ffn);
            if (!strcmp(sub, "feed_forward_layer.layer.fc1.bias")) return put(o.b1, ESMK_DT_F32, F);
            if (!strcmp(sub, "feed_forward_layer.layer.fc2.weight")) return putw(o.w2, E, F, F, 0, 0, sp.ffn);
            if (!strcmp(sub, "feed_forward_layer.layer.fc2.bias")) return put(o.b2, ESMK_DT_F32, E);
            if (!strcmp(sub, "feed_forward_layer.layer_norm.weight")) return put(o.flng, ESMK_DT_F32, E);
            if (!strcmp(sub, "feed_forward_layer.layer_norm.bias")) return put(o.flnb, ESMK_DT_F32, E);
            return 0;
        }
    }
    if (!strcmp(key, "lm_head.weight")) return 0;  // tied to embed_tokens.weight (esm2.py:71-75)
    if (!m->is_msa && m->cfg.num_positions > 0 && !strcmp(key, "embed_positions.weight"))
        return put(m->pos_emb, ESMK_DT_F32, (size_t)m->cfg.num_positions * E);
    if (!m->is_msa && m->cfg.ln_before && !strcmp(key, "emb_layer_norm_before.weight")) return put(m->lnb_g, ESMK_DT_F32, E);
    if (!m->is_msa && m->cfg.ln_before && !strcmp(key, "emb_layer_norm_before.bias")) return put(m->lnb_b, ESMK_DT_F32, E);
    if (!strcmp(key, "emb_layer_norm_after.weight")) return put(m->fin_g, ESMK_DT_F32, E);
    if (!strcmp(key, "emb_layer_norm_after.bias")) return put(m->fin_b, ESMK_DT_F32, E);
    if (!strcmp(key, "lm_head.dense.weight")) {
        if (m->cfg.weight_split && put(m->lm_w32, ESMK_DT_F32, E * E)) return 1;
        return put2d(m->lm_w, op, E, E, Kp, 0, 0);
    }
    if (!strcmp(key, "lm_head.dense.bias")) return put(m->lm_b, ESMK_DT_F32, E);
    if (!strcmp(key, "lm_head.layer_norm.weight")) return put(m->lm_lng, ESMK_DT_F32, E);
    if (!strcmp(key, "lm_head.layer_norm.bias")) return put(m->lm_lnb, ESMK_DT_F32, E);
    if (!strcmp(key, "lm_head.bias")) return put(m->lm_bias, ESMK_DT_F32, V);
    if (!strcmp(key, "contact_head.regression.weight"))
        return put(m->ct_w, ESMK_DT_F32, (size_t)m->L * m->H);
    if (!strcmp(key, "contact_head.regression.bias")) return put(m->ct_b, ESMK_DT_F32, 1);
    if (starts_with(key, "layers.")) {
        char* end = nullptr;
        const long l = strtol(key + 7, &end, 10);
        if (end == key + 7 || *end != '.' || l < 0 || l >= m->L)
            return fail(std::string("esmk_pack_weight: bad layer index in ") + key);
        const char* sub = end + 1;
        const LayerOff& o = m->layer[l];
        // LayerNorm fold: q/k/v and fc1 weights are packed as gamma-folded, row-centred images + W . beta (bias2); the
        // LayerNorm parameters they fold must be in the image already, and packing one of those later marks the folded
        // weights stale (esmk_forward refuses to run on a stale fold)
        if (m->fold && m->fold_image != packed_dev) {  // another image: its folds start unpacked
            m->fold_state.assign(m->L, 0u);
            m->fold_image = packed_dev;
        }
        uint32_t& fs = m->fold_state[l];
        // LayerNorm parameter of a fold: the "present" bit is set only once the copy was queued; the folded weights that
        // depend on it are stale from the moment the call is made, whether or not it succeeds
        auto ln_put = [&](size_t off, uint32_t present, uint32_t stale) -> int {
            fs &= ~(stale | present);
            if (put(off, ESMK_DT_F32, E)) return 1;
            fs |= present;
            return 0;
        };
        auto fold_put = [&](size_t woff, size_t b2off, size_t rows, int rmap, size_t lng, size_t lnb, uint32_t need,
                            uint32_t done) -> int {
            if ((fs & need) != need)
                return fail(std::string("esmk_pack_weight: ") + key + ": with the LayerNorm fold the layer's LayerNorm weight "
                            "and bias must be packed before its q/k/v and fc1 weights");
            if (n != rows * E)
                return fail(std::string("esmk_pack_weight: ") + key + " has " + std::to_string(n) + " elements, expected " +
                            std::to_string(rows * E));
            ESMK_TRY(launch_fold_weight(src_dev, src_dtype, (const float*)(base + lng), (const float*)(base + lnb), base + woff, op,
                                        (float*)(base + b2off), rows, E, Kp, rmap, hd, st));
            fs |= done;
            return 0;
        };
        if (m->fold) {
            if (!strcmp(sub, "self_attn.q_proj.weight")) return fold_put(o.wqkv, o.bqkv2, E, qkmap, o.ln1g, o.ln1b, FB_LN1G | FB_LN1B, FB_WQ);
            if (!strcmp(sub, "self_attn.k_proj.weight")) return fold_put(o.wqkv + EA * Kp * os, o.bqkv2 + EA * 4, E, qkmap, o.ln1g, o.ln1b, FB_LN1G | FB_LN1B, FB_WK);
            if (!strcmp(sub, "self_attn.v_proj.weight")) return fold_put(o.wqkv + 2 * EA * Kp * os, o.bqkv2 + 2 * EA * 4, E, padmap, o.ln1g, o.ln1b, FB_LN1G | FB_LN1B, FB_WV);
            if (!strcmp(sub, "fc1.weight")) return fold_put(o.w1, o.b12, F, 0, o.ln2g, o.ln2b, FB_LN2G | FB_LN2B, FB_W1);
            if (!strcmp(sub, "self_attn_layer_norm.weight")) return ln_put(o.ln1g, FB_LN1G, FB_WQ | FB_WK | FB_WV);
            if (!strcmp(sub, "self_attn_layer_norm.bias")) return ln_put(o.ln1b, FB_LN1B, FB_WQ | FB_WK | FB_WV);
            if (!strcmp(sub, "final_layer_norm.weight")) return ln_put(o.ln2g, FB_LN2G, FB_W1);
            if (!strcmp(sub, "final_layer_norm.bias")) return ln_put(o.ln2b, FB_LN2B, FB_W1);
        }
        // q/k/v: output rows are head dims -> spread over 64 slots; input columns padded to Kp
        if (!strcmp(sub, "self_attn.q_proj.weight")) return putw(o.wqkv, E, E, Kp, qkmap, 0, sp.qk);
        if (!strcmp(sub, "self_attn.k_proj.weight")) return putw(o.wqkv + EA * Kp * os * sp.qk, E, E, Kp, qkmap, 0, sp.qk);
        if (!strcmp(sub, "self_attn.v_proj.weight")) return putw(o.wqkv + 2 * EA * Kp * os * sp.qk, E, E, Kp, padmap, 0, sp.v);
        if (!strcmp(sub, "self_attn.q_proj.bias")) return put2d(o.bqkv, ESMK_DT_F32, 1, E, EA, 0, qkmap);
        if (!strcmp(sub, "self_attn.k_proj.bias")) return put2d(o.bqkv + EA * 4, ESMK_DT_F32, 1, E, EA, 0, qkmap);
        if (!strcmp(sub, "self_attn.v_proj.bias")) return put2d(o.bqkv + 2 * EA * 4, ESMK_DT_F32, 1, E, EA, 0, padmap);
        // out_proj consumes the attention context: its input columns follow the same slot layout
        if (!strcmp(sub, "self_attn.out_proj.weight")) return putw(o.wo, E, E, EA, 0, padmap, sp.o);
        if (!strcmp(sub, "self_attn.out_proj.bias")) return put(o.bo, ESMK_DT_F32, E);
        if (!strcmp(sub, "fc1.weight")) return putw(o.w1, F, E, Kp, 0, 0, sp.ffn);
        if (!strcmp(sub, "fc1.bias")) return put(o.b1, ESMK_DT_F32, F);
        if (!strcmp(sub, "fc2.weight")) return putw(o.w2, E, F, F, 0, 0, sp.ffn);
        if (!strcmp(sub, "fc2.bias")) return put(o.b2, ESMK_DT_F32, E);
        if (!strcmp(sub, "self_attn_layer_norm.weight")) return put(o.ln1g, ESMK_DT_F32, E);
        if (!strcmp(sub, "self_attn_layer_norm.bias")) return put(o.ln1b, ESMK_DT_F32, E);
        if (!strcmp(sub, "final_layer_norm.weight")) return put(o.ln2g, ESMK_DT_F32, E);
        if (!strcmp(sub, "final_layer_norm.bias")) return put(o.ln2b, ESMK_DT_F32, E);
        return 0;  // e.g. self_attn.rot_emb.inv_freq: rebuilt in fp32 by the engine
    }
    return 0;  // unknown keys are ignored
}

int esmk_workspace_bytes(const esmk_model* m, int B, int T, uint32_t out_flags, size_t* bytes) {
    if (!m || !bytes) return fail("esmk_workspace_bytes: null argument");
    if (m->is_msa) return fail("esmk_workspace_bytes: MSA handle (use esmk_msa_workspace_bytes)");
    if (B <= 0 || T <= 0) return fail("esmk_workspace_bytes: B and T must be positive");
    if ((long long)B * T > ESMK_MAX_ROWS) return fail("esmk_workspace_bytes: B*T exceeds 2^24 rows");
    *bytes = plan_workspace(m, B, T, out_flags).total;
    return 0;
}

static int forward_impl(esmk_model* m, const void* packed_dev, const int64_t* tokens_dev, int B, int T,
                        const int32_t* repr_layers, int n_repr, void* const* repr_out_dev,
                        uint32_t out_flags, void* logits_out_dev, void* attn_out_dev,
                        void* contacts_out_dev, void* workspace_dev, size_t workspace_bytes,
                        void* stream, const PackedCtx* pc);

int esmk_forward(esmk_model* m, const void* packed_dev, const int64_t* tokens_dev, int B, int T,
                 const int32_t* repr_layers, int n_repr, void* const* repr_out_dev,
                 uint32_t out_flags, void* logits_out_dev, void* attn_out_dev,
                 void* contacts_out_dev, void* workspace_dev, size_t workspace_bytes,
                 void* stream) {
    return forward_impl(m, packed_dev, tokens_dev, B, T, repr_layers, n_repr, repr_out_dev, out_flags,
                        logits_out_dev, attn_out_dev, contacts_out_dev, workspace_dev, workspace_bytes, stream,
                        nullptr);
}

// ---- token-packed batches (no compute on padding) --------------------------------------
static int check_segments(const char* who, const esmk_model* m, const int32_t* seg, int n_seg, int rows,
                          PackedCtx* pc) {
    const std::string w(who);
    if (!m || !seg) return fail(w + ": null argument");
    if (m->is_msa) return fail(w + ": not an ESM-2 handle");
    if (n_seg <= 0 || rows <= 0) return fail(w + ": n_seg and rows must be positive");
    if (rows % 64 != 0) return fail(w + ": rows must be a multiple of 64");
    if (rows > ESMK_MAX_ROWS) return fail(w + ": rows exceed 2^24");
    long long end = 0;
    int max_len = 0;
    size_t items = 0;
    for (int s = 0; s < n_seg; ++s) {
        const int start = seg[2 * s], len = seg[2 * s + 1];
        if (len <= 0) return fail(w + ": empty segment");
        if (start % 16 != 0) return fail(w + ": segment starts must be multiples of 16");
        if ((s == 0 && start != 0) || start < end) return fail(w + ": segments must start at row 0, ascending, disjoint");
        end = (long long)start + len;
        if (end > rows) return fail(w + ": segment past the last row");
        max_len = std::max(max_len, len);
        items += (size_t)(len + 127) / 128;
        if (pc) pc->sum_len2 += (double)len * len;
    }
    if (pc) {
        pc->n_seg = n_seg;
        pc->max_len = max_len;
        pc->n_items = (int)items;
        pc->seg_host = seg;
    }
    return 0;
}

int esmk_packed_workspace_bytes(const esmk_model* m, int n_seg, int rows, uint32_t out_flags, size_t* bytes) {
    if (!m || !bytes) return fail("esmk_packed_workspace_bytes: null argument");
    if (m->is_msa) return fail("esmk_packed_workspace_bytes: not an ESM-2 handle");
    if (n_seg <= 0 || rows <= 0 || rows % 64 != 0 || rows > ESMK_MAX_ROWS)
        return fail("esmk_packed_workspace_bytes: need n_seg > 0 and 0 < rows <= 2^24, rows % 64 == 0");
    if (out_flags & ~(uint32_t)(ESMK_OUT_LOGITS | ESMK_OUT_REPR_LOWP))
        return fail("esmk_packed_workspace_bytes: only ESMK_OUT_LOGITS / ESMK_OUT_REPR_LOWP are available");
    *bytes = plan_workspace(m, 1, rows, out_flags, n_seg).total;
    return 0;
}

int esmk_forward_packed(esmk_model* m, const void* packed_dev, const int64_t* tokens_dev,
                        const int32_t* segments_host, int n_seg, int rows, const int32_t* repr_layers,
                        int n_repr, void* const* repr_out_dev, uint32_t out_flags, void* logits_out_dev,
                        void* workspace_dev, size_t workspace_bytes, void* stream) {
    PackedCtx pc;
    if (check_segments("esmk_forward_packed", m, segments_host, n_seg, rows, &pc)) return 1;
    if (out_flags & ~(uint32_t)(ESMK_OUT_LOGITS | ESMK_OUT_REPR_LOWP))
        return fail("esmk_forward_packed: attention maps and contacts take padded batches (esmk_forward)");
    return forward_impl(m, packed_dev, tokens_dev, 1, rows, repr_layers, n_repr, repr_out_dev, out_flags,
                        logits_out_dev, nullptr, nullptr, workspace_dev, workspace_bytes, stream, &pc);
}

static int forward_impl(esmk_model* m, const void* packed_dev, const int64_t* tokens_dev, int B, int T,
                        const int32_t* repr_layers, int n_repr, void* const* repr_out_dev,
                        uint32_t out_flags, void* logits_out_dev, void* attn_out_dev,
                        void* contacts_out_dev, void* workspace_dev, size_t workspace_bytes,
                        void* stream, const PackedCtx* pc) {
    if (!m || !packed_dev || !tokens_dev || !workspace_dev) return fail("esmk_forward: null argument");
    if (m->is_msa) return fail("esmk_forward: MSA handle (use esmk_msa_forward)");
    if (B <= 0 || T <= 0) return fail("esmk_forward: B and T must be positive");
    if ((long long)B * T > ESMK_MAX_ROWS) return fail("esmk_forward: B*T exceeds 2^24 rows");
    if (n_repr > 0 && (!repr_layers || !repr_out_dev)) return fail("esmk_forward: null repr arrays");
    const bool want_logits = out_flags & ESMK_OUT_LOGITS;
    const bool want_contacts = out_flags & ESMK_OUT_CONTACTS;
    // contacts alone (predict_contacts, esm2.py:146-147): accumulated layer by layer, no attention tensor
    const bool fused_ct = want_contacts && !(out_flags & ESMK_OUT_ATTN);
    const bool want_attn = (out_flags & ESMK_OUT_ATTN) != 0;
    const int S_ct = T - (m->cfg.prepend_bos ? 1 : 0) - (m->cfg.append_eos ? 1 : 0);
    const bool repr_lowp = out_flags & ESMK_OUT_REPR_LOWP, attn_lowp = out_flags & ESMK_OUT_ATTN_LOWP;
    if (attn_lowp && want_attn && want_contacts)
        return fail("esmk_forward: ESMK_OUT_ATTN_LOWP cannot be combined with contacts computed from the attention tensor");
    if (want_logits && !logits_out_dev) return fail("esmk_forward: logits buffer missing");
    if (want_attn && !attn_out_dev) return fail("esmk_forward: attention buffer missing");
    if (want_contacts && !contacts_out_dev) return fail("esmk_forward: contacts buffer missing");
    for (int i = 0; i < n_repr; ++i)
        if (repr_layers[i] < 0 || repr_layers[i] > m->L || !repr_out_dev[i])
            return fail("esmk_forward: bad repr layer request");
    const Workspace w = plan_workspace(m, B, T, out_flags, pc ? pc->n_seg : 0);
    if (workspace_bytes < w.total) return fail("esmk_forward: workspace too small");

    hipStream_t st = (hipStream_t)stream;
    const int op = m->cfg.operand_dtype;
    const size_t os = op_size(op);
    const int N = B * T, E = m->E, F = m->F, H = m->H, L = m->L, EA = m->EA, Kp = m->Kp;
    char* ws = (char*)workspace_dev;
    const char* pk = (const char*)packed_dev;
    float* scale = (float*)(ws + w.scale);
    float* key_bias = (float*)(ws + w.key_bias);
    int* seq_info = (int*)(ws + w.seq_info);
    float* x = (float*)(ws + w.x);
    void* h = ws + w.h;
    void* q = ws + w.q;
    void* k = ws + w.k;
    void* vt = ws + w.vt;
    void* ffn = ws + w.big;
    float* g32 = (float*)(ws + w.big);
    float* lse = (want_attn || fused_ct) ? (float*)(ws + w.lse) : nullptr;

    const int T_rope = pc ? pc->max_len : T;  // longest run of positions
    if (m->cfg.no_rope) {
        if (ensure_unit_rope(m, T_rope, st)) return 1;
    } else if (ensure_rope(m, T_rope, st)) {
        return 1;
    }
    // token-packed batch: segment table, <pad> counts and the attention work list live behind the workspace
    int* row_pos = nullptr;
    AttnSegs segs;
    if (pc) {
        int* tab = (int*)(ws + w.tables);
        const size_t n_int = (size_t)3 * pc->n_seg + (size_t)4 * pc->n_items;
        if (m->pk_event) ESMK_TRY(hipEventSynchronize(m->pk_event));  // the previous upload has read the staging
        else ESMK_TRY(hipEventCreateWithFlags(&m->pk_event, hipEventDisableTiming));
        if (m->pk_host_cap < n_int) {
            if (m->pk_host) ESMK_TRY(hipHostFree(m->pk_host));
            m->pk_host = nullptr;
            m->pk_host_cap = 0;
            ESMK_TRY(hipHostMalloc((void**)&m->pk_host, 2 * n_int * 4, hipHostMallocDefault));
            m->pk_host_cap = 2 * n_int;
        }
        int32_t* hostv = m->pk_host;
        memcpy(hostv, pc->seg_host, (size_t)2 * pc->n_seg * 4);
        memset(hostv + (size_t)2 * pc->n_seg, 0, (size_t)pc->n_seg * 4);
        // query blocks, longest segments first: the tail of the grid is made of short work items
        std::vector<int> order(pc->n_seg);
        for (int s = 0; s < pc->n_seg; ++s) order[s] = s;
        std::stable_sort(order.begin(), order.end(),
                         [&](int a, int b) { return pc->seg_host[2 * a + 1] > pc->seg_host[2 * b + 1]; });
        int32_t* wk = hostv + (size_t)3 * pc->n_seg;
        for (int s : order) {
            const int start = pc->seg_host[2 * s], len = pc->seg_host[2 * s + 1];
            for (int q0 = 0; q0 < len; q0 += 128) {
                wk[0] = start;
                wk[1] = len;
                wk[2] = q0;
                wk[3] = s;
                wk += 4;
            }
        }
        ESMK_TRY(hipMemcpyAsync(tab, hostv, n_int * 4, hipMemcpyHostToDevice, st));
        ESMK_TRY(hipEventRecord(m->pk_event, st));
        row_pos = (int*)(ws + w.row_pos);
        segs.npad = tab + (size_t)2 * pc->n_seg;
        segs.work = tab + (size_t)3 * pc->n_seg;
    }

    const double NE = (double)N * E;
    auto repr_copy = [&](int layer, const float* src) -> int {
        for (int i = 0; i < n_repr; ++i)
            if (repr_layers[i] == layer) {
                ProfScope ps(m, st, PC_COPY, 0, (repr_lowp ? 4 + os : 8) * NE);
                if (repr_lowp) ESMK_TRY(launch_convert(src, ESMK_DT_F32, repr_out_dev[i], op, (size_t)N * E, st));
                else ESMK_TRY(launch_copy_f32(src, (float*)repr_out_dev[i], (size_t)N * E, st));
            }
        return 0;
    };
    auto wants_repr = [&](int layer) {
        for (int i = 0; i < n_repr; ++i)
            if (repr_layers[i] == layer) return true;
        return false;
    };
    // algorithmic bytes: operands read once + result written once (residual: read + written)
    auto gemm = [&](int cls, const GemmArgs& a, int epi, double out_bytes_per_elem) -> int {
        const double fl = 2.0 * a.M * (double)a.N * a.K;
        const double by = ((double)a.M * a.K + (double)a.N * a.K) * os + (double)a.M * a.N * out_bytes_per_elem;
        ProfScope ps(m, st, cls, fl, by);
        ESMK_TRY(launch_gemm(a, epi, op, st));
        return 0;
    };
    // a GEMM of the layer stack: with split weights (f16x2) the same kernel runs over the [N, 2K] hi | lo image, the
    // activations' K tile kt / 2 meeting W_hi (kt even) and W_lo (kt odd); FLOP / byte accounting stays algorithmic
    const int wsf = split_plan(m).qk;            // q / k weights: the v rows of the image start behind 2 EA rows of this length
    const bool any_split = m->cfg.weight_split != 0;
    // Precision mode f16x3 (weight_split 4): every layer GEMM is a PLAIN launch over K' = 3 K — weight images hi | lo | hi per K
    // tile, operand rows hi | hi | lo: a3 from the LayerNorm kernel (LnExtra::x3) and from the attention kernel's X3 output,
    // ffn3 from fc1's GELU epilogue (GemmArgs::x3_out)
    const bool x3 = split_x3(m);
    if (x3 && (pc != nullptr || m->D != 64 || Kp != E || EA != E))
        return fail("esmk_forward: the f16x3 precision mode runs padded batches of head_dim-64 models (no token-packed form)");
    void* a3 = x3 ? (void*)(ws + w.a3) : nullptr;
    void* ffn3 = x3 ? (void*)(ws + w.ffn3) : nullptr;
    auto layer_gemm = [&](int cls, GemmArgs a, int epi, double out_bytes_per_elem) -> int {
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
    auto lnorm = [&](const float* in, size_t go, size_t bo, void* y, float* y32) -> int {
        ProfScope ps(m, st, PC_LAYERNORM, 8 * NE, NE * (4 + (y ? os : 0) + (y32 ? 4 : 0)));
        LnExtra ex;
        ex.ldy = Kp;  // normalised rows are K operands: row stride = E rounded up to the 64-wide K tile
        ESMK_TRY(launch_layernorm_ex(in, (const float*)(pk + go), (const float*)(pk + bo), y, y32, N, E, op, ex, st));
        return 0;
    };
    auto ln_x3 = [&](size_t go, size_t bo) -> int {  // LayerNorm(x) -> hi | hi | lo operand rows (LnExtra::x3)
        ProfScope ps(m, st, PC_LAYERNORM, 8 * NE, NE * (4 + 3 * os));
        LnExtra ex;
        ex.ldy = 3 * E;
        ex.x3 = 1;
        ESMK_TRY(launch_layernorm_ex(x, (const float*)(pk + go), (const float*)(pk + bo), a3, nullptr, N, E, op, ex, st));
        return 0;
    };
    // LayerNorm fold: hA = raw rows of the residual stream in the operand dtype (written by rowstats for
    // layer 0, then by the residual epilogues), hB = attention context; without the fold both are `h`
    const bool fold = m->fold;
    if (fold && m->fold_image != packed_dev)
        return fail("esmk_forward: LayerNorm fold: this packed image is not the one the handle's weights were last packed "
                    "into (one image per handle at a time: re-pack, or use a second handle)");
    if (fold)
        for (int l = 0; l < L; ++l)
            if ((m->fold_state[l] & FB_ALL_W) != FB_ALL_W)
                return fail("esmk_forward: LayerNorm fold: the q/k/v or fc1 weights of layer " + std::to_string(l) +
                            " were not packed after the layer's LayerNorm parameters");
    void* hA = h;
    void* hB = fold ? (void*)(ws + w.h2) : h;
    float* ln_part = fold ? (float*)(ws + w.ln_part) : nullptr;
    float* ln_mean = fold ? (float*)(ws + w.ln_mean) : nullptr;
    float* ln_rstd = fold ? (float*)(ws + w.ln_rstd) : nullptr;
    auto producer = [&](GemmArgs& a) {  // a residual GEMM that also emits the next GEMM's rows and their statistics
        a.h16 = hA;
        a.ldh = Kp;
        a.ln_part = ln_part;
        a.ln_parts = w.ln_parts;
        a.ln_mean = ln_mean;
    };
    auto finalize = [&]() -> int {
        ProfScope ps(m, st, PC_LN_STATS, 4.0 * N * w.ln_parts, (double)N * (8.0 * w.ln_parts + 12));
        ESMK_TRY(launch_ln_finalize(ln_part, ln_mean, ln_rstd, N, w.ln_parts, E, st));
        return 0;
    };
    // pad columns [E, Kp) of the activation rows must be finite (they meet zero weight columns)
    if (Kp != E) {
        ESMK_TRY(hipMemsetAsync(h, 0, (size_t)N * std::max(Kp, EA) * os, st));
        if (fold) ESMK_TRY(hipMemsetAsync(hB, 0, (size_t)N * std::max(Kp, EA) * os, st));
    }

    // esm2.py:82-95
    {
        ProfScope ps(m, st, PC_EMBED, 0, (double)N * 8 + 4 * NE);
        const bool esm1b = m->cfg.num_positions > 0;
        float* keep = esm1b ? (float*)(ws + w.keep) : nullptr;
        if (pc) {
            ESMK_TRY(launch_packed_stats(tokens_dev, (const int*)(ws + w.tables), pc->n_seg, T, m->cfg.pad_idx,
                                         m->cfg.mask_idx, scale, key_bias, row_pos, (int*)segs.npad, st, keep));
            // the token-dropout divisor is per row: "sequences" of one token
            ESMK_TRY(launch_embed(tokens_dev, (const float*)(pk + m->embed_f32), scale, x, T, 1, E, m->V,
                                  m->cfg.pad_idx, m->cfg.mask_idx, m->cfg.token_dropout, st));
        } else {
            ESMK_TRY(launch_seq_stats(tokens_dev, B, T, m->cfg.pad_idx, m->cfg.mask_idx,
                                      m->cfg.token_dropout, scale, key_bias, seq_info, st, keep));
            ESMK_TRY(launch_embed(tokens_dev, (const float*)(pk + m->embed_f32), scale, x, B, T, E, m->V,
                                  m->cfg.pad_idx, m->cfg.mask_idx, m->cfg.token_dropout, st));
        }
        if (esm1b) {
            // esm1.py:133-139: + learned positions, emb_layer_norm_before, padded positions zeroed
            if (T_rope > m->cfg.num_positions - m->cfg.pad_idx - 1)
                return fail("esmk_forward: sequence length above the maximum of the positional embedding");
            if (pc)
                ESMK_TRY(launch_add_positions(tokens_dev, (const float*)(pk + m->pos_emb), x, pc->n_seg, pc->max_len, E,
                                              m->cfg.pad_idx, m->cfg.num_positions, st, (const int*)(ws + w.tables)));
            else
                ESMK_TRY(launch_add_positions(tokens_dev, (const float*)(pk + m->pos_emb), x, B, T, E, m->cfg.pad_idx,
                                              m->cfg.num_positions, st));
            if (m->cfg.ln_before) {
                LnExtra ex;
                ex.row_keep = keep;
                ESMK_TRY(launch_layernorm_ex(x, (const float*)(pk + m->lnb_g), (const float*)(pk + m->lnb_b), nullptr,
                                             x, N, E, op, ex, st));
            } else {
                ESMK_TRY(launch_scale_rows(x, keep, N, E, st));
            }
        }
    }
    if (repr_copy(0, x)) return 1;  // esm2.py:99-100

    // ESMK_QKV_FORK=1: q/k and v projections side by side on two streams (see esmk_model::side_stream)
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
    GemmArgs g;
    for (int l = 0; l < L; ++l) {  // esm2.py:111-121 -> modules.py:120-142
        const LayerOff& o = m->layer[l];
        // keys in [T,Tp) of V^T get probability exactly 0 but must be finite; the region is
        // shared with the FFN intermediate, so it is cleared every layer (odd T only).
        if (pc)  // only the spare key tile: every row below it is a computed (finite) row
            ESMK_TRY(hipMemset2DAsync((char*)vt + (size_t)T * os, (size_t)w.Tp * os, 0, 64 * os, (size_t)EA, st));
        else if (w.Tp != T) ESMK_TRY(hipMemsetAsync(vt, 0, (size_t)B * EA * w.Tp * os, st));
        if (x3) {
            if (ln_x3(o.ln1g, o.ln1b)) return 1;
        } else if (!fold) {
            if (lnorm(x, o.ln1g, o.ln1b, h, nullptr)) return 1;
        } else if (l == 0) {  // entry of the fold chain: rows and statistics of the embedded stream
            ProfScope ps(m, st, PC_LN_STATS, 8 * NE, NE * (4 + os));
            ESMK_TRY(launch_rowstats(x, hA, ln_mean, ln_rstd, N, E, Kp, op, st));
        }
        g = GemmArgs();
        g.A = hA;
        g.W = pk + o.wqkv;
        g.bias = (const float*)(pk + o.bqkv);
        if (fold) {
            g.ln_rstd = ln_rstd;
            g.bias2 = (const float*)(pk + o.bqkv2);
        }
        g.M = N;
        g.N = 2 * EA;
        g.K = Kp;
        if (x3) {
            g.A = a3;
            g.K = 3 * Kp;
        }
        g.q = q;
        g.k = k;
        g.vt = vt;
        g.cos = m->cfg.no_rope ? m->d_ucos : m->d_cos;
        g.sin = m->cfg.no_rope ? m->d_usin : m->d_sin;
        g.T = T;
        g.H = H;
        g.E = EA;
        g.Tp = w.Tp;
        // q carries d^-1/2 (multihead_attention.py:256-261) AND log2(e): the attention / map / contact kernels
        // work on log2-domain scores (softmax as exp2, see attention.hip)
        g.scaling = kLog2e / sqrtf((float)m->D);
        g.head_dim = m->D == 128 ? 128 : 64;
        g.row_pos = row_pos;
        GemmArgs gv = g;
        gv.row_pos = nullptr;
        gv.W = pk + o.wqkv + (size_t)2 * EA * Kp * os * wsf;     // v: weight rows [2EA,3EA)
        gv.bias = (const float*)(pk + o.bqkv) + 2 * EA;
        if (fold) gv.bias2 = (const float*)(pk + o.bqkv2) + 2 * EA;
        gv.N = EA;
        if (fork_v) {
            // v on the side stream, after everything queued so far (the LayerNorm that wrote h, the V^T clear); the
            // attention below waits for it.  Not under the per-class profiler: its events live on one stream.
            ESMK_TRY(hipEventRecord(m->ev_fork, st));
            ESMK_TRY(hipStreamWaitEvent(m->side_stream, m->ev_fork, 0));
            ESMK_TRY(launch_gemm(gv, EPI_V_T, op, m->side_stream));
            ESMK_TRY(hipEventRecord(m->ev_join, m->side_stream));
            if (gemm(PC_GEMM_QKV, g, EPI_QKV_ROPE, os)) return 1;  // q, k: weight rows [0,2EA)
            ESMK_TRY(hipStreamWaitEvent(st, m->ev_join, 0));
        } else if (!any_split && gemm_qkv_one_launch(g)) {
            // small batches: q, k and v in one launch — same tiles, same bits, fewer rounds over the CUs (kernels.h, EPI_QKV_ALL)
            GemmArgs ga = g;
            ga.N = 3 * EA;
            if (gemm(PC_GEMM_QKV, ga, EPI_QKV_ALL, os)) return 1;
        } else if (x3) {
            if (gemm(PC_GEMM_QKV, g, EPI_QKV_ROPE, os)) return 1;
            if (gemm(PC_GEMM_QKV, gv, EPI_V_T, os)) return 1;
        } else {
            if (layer_gemm(PC_GEMM_QKV, g, EPI_QKV_ROPE, os)) return 1;  // q, k: weight rows [0,2EA)
            if (layer_gemm(PC_GEMM_QKV, gv, EPI_V_T, os)) return 1;
        }
        {
            // 4 T d flop per (query, head) pair: QK^T and PV; q,k,v read + ctx written
            ProfScope ps(m, st, PC_ATTENTION, pc ? 4.0 * pc->sum_len2 * E : 4.0 * N * (double)T * E, 4 * NE * os);
            if (pc)  // gap rows of the context (the rows of h were last read by the two GEMMs above)
                ESMK_TRY(launch_zero_gap_rows(hB, (const int*)(ws + w.tables), pc->n_seg, T, (size_t)EA * os, st));
            if (pc && m->D == 128)
                ESMK_TRY(launch_attention128_packed(q, k, vt, key_bias, hB, H, T, w.Tp, segs, pc->n_items, op, st));
            else if (pc) ESMK_TRY(launch_attention_packed(q, k, vt, key_bias, hB, H, T, w.Tp, segs, pc->n_items, op, st));
            else if (m->D == 128) ESMK_TRY(launch_attention128(q, k, vt, key_bias, seq_info, hB, lse, B, H, T, w.Tp, op, st));
            else if (x3) ESMK_TRY(launch_attention_x3(q, k, vt, key_bias, seq_info, a3, lse, B, H, T, w.Tp, op, st));
            else ESMK_TRY(launch_attention(q, k, vt, key_bias, seq_info, hB, lse, B, H, T, w.Tp, op, st));
        }
        if (fused_ct && S_ct > 0) {
            // q, k and lse of this layer are still in the workspace: add the layer's channels to the
            // [B,T,T] accumulator and the per-channel masked row / column sums
            ProfScope ps(m, st, PC_ATTN_PROBS, 2.0 * N * (double)T * E, 2 * NE * os + 8.0 * N * T);
            ESMK_TRY(launch_contacts_fused_layer(q, k, lse, key_bias, tokens_dev, (const float*)(pk + m->ct_w),
                                                 (float*)(ws + w.ct_acc), (float*)(ws + w.ct_row),
                                                 (float*)(ws + w.ct_col), (float*)(ws + w.ct_rowp),
                                                 (float*)(ws + w.ct_colp), B, H, T, L * H, l,
                                                 m->D == 128 ? 128 : 64, m->cfg.pad_idx, m->cfg.eos_idx,
                                                 m->cfg.prepend_bos, m->cfg.append_eos, op, st));
        }
        if (want_attn) {
            ProfScope ps(m, st, PC_ATTN_PROBS, 2.0 * N * (double)T * E, 2 * NE * os + 4.0 * N * T * H);
            if (m->D == 128)
                ESMK_TRY(launch_attention_probs128(q, k, lse, key_bias, (float*)attn_out_dev, B, H, T, l, L, op, st,
                                                   attn_lowp));
            else
                ESMK_TRY(launch_attention_probs(q, k, lse, key_bias, (float*)attn_out_dev, B, H, T, l, L, op, st,
                                                attn_lowp));
        }
        g = GemmArgs();
        g.A = hB;
        g.W = pk + o.wo;
        g.bias = (const float*)(pk + o.bo);
        g.out = x;
        g.M = N;
        g.N = E;
        g.K = EA;
        if (fold) producer(g);
        if (x3) {
            g.A = a3;
            g.K = 3 * EA;
            if (gemm(PC_GEMM_OUT, g, EPI_RESID_F32, 8)) return 1;
            if (ln_x3(o.ln2g, o.ln2b)) return 1;
        } else {
        if (layer_gemm(PC_GEMM_OUT, g, EPI_RESID_F32, fold ? 8 + os : 8)) return 1;
        if (fold) {
            if (finalize()) return 1;
        } else if (lnorm(x, o.ln2g, o.ln2b, h, nullptr)) {
            return 1;
        }
        }
        g = GemmArgs();
        g.A = hA;
        g.W = pk + o.w1;
        g.bias = (const float*)(pk + o.b1);
        if (fold) {
            g.ln_rstd = ln_rstd;
            g.bias2 = (const float*)(pk + o.b12);
        }
        g.out = ffn;
        g.M = N;
        g.N = F;
        g.K = Kp;
        if (x3) {  // fc1 + GELU writing the hi | hi | lo rows of fc2's operand (GemmArgs::x3_out)
            g.A = a3;
            g.K = 3 * Kp;
            g.out = ffn3;
            g.x3_out = 1;
            if (gemm(PC_GEMM_FC1, g, EPI_GELU_T, 3 * os)) return 1;
        } else if (layer_gemm(PC_GEMM_FC1, g, EPI_GELU_T, os)) {
            return 1;
        }
        g = GemmArgs();
        g.A = ffn;
        g.W = pk + o.w2;
        g.bias = (const float*)(pk + o.b2);
        g.out = x;
        g.M = N;
        g.N = E;
        g.K = F;
        const bool feeds_next = fold && l + 1 < L;  // the next layer's q/k/v projections read the rows this GEMM writes
        if (feeds_next) producer(g);
        if (x3) {
            g.A = ffn3;
            g.K = 3 * F;
            if (gemm(PC_GEMM_FC2, g, EPI_RESID_F32, 8)) return 1;
        } else
        if (layer_gemm(PC_GEMM_FC2, g, EPI_RESID_F32, feeds_next ? 8 + os : 8)) return 1;
        if (feeds_next && finalize()) return 1;
        if (l + 1 < L && repr_copy(l + 1, x)) return 1;  // esm2.py:117-118
    }

    // esm2.py:123-128: final LayerNorm; representation L is the normalised stream
    float* rep_last = nullptr;
    for (int i = 0; i < n_repr; ++i)
        if (repr_layers[i] == L) {
            rep_last = (float*)repr_out_dev[i];
            break;
        }
    if (repr_lowp && rep_last != nullptr) {
        // representation L in the operand dtype: the normalised rows h ARE that tensor when their row stride is E
        void* rep_lp = rep_last;
        if (Kp == E) {
            if (lnorm(x, m->fin_g, m->fin_b, want_logits ? h : rep_lp, nullptr)) return 1;
            if (want_logits) ESMK_TRY(hipMemcpyAsync(rep_lp, h, (size_t)N * E * os, hipMemcpyDeviceToDevice, st));
        } else {  // padded row stride (E = 480): through the fp32 scratch
            if (lnorm(x, m->fin_g, m->fin_b, want_logits ? h : nullptr, g32)) return 1;
            ESMK_TRY(launch_convert(g32, ESMK_DT_F32, rep_lp, op, (size_t)N * E, st));
        }
        for (int i = 0; i < n_repr; ++i)  // duplicates of layer L, if any
            if (repr_layers[i] == L && repr_out_dev[i] != rep_lp)
                ESMK_TRY(hipMemcpyAsync(repr_out_dev[i], rep_lp, (size_t)N * E * os, hipMemcpyDeviceToDevice, st));
    } else if (want_logits || wants_repr(L)) {
        if (lnorm(x, m->fin_g, m->fin_b, want_logits ? h : nullptr, rep_last)) return 1;
        for (int i = 0; i < n_repr; ++i)  // duplicates of layer L, if any
            if (repr_layers[i] == L && repr_out_dev[i] != rep_last)
                ESMK_TRY(launch_copy_f32(rep_last, (float*)repr_out_dev[i], (size_t)N * E, st));
    }
    if (want_logits && m->cfg.weight_split && E % 32 == 0) {
        // f16x2 precision mode: the head (modules.py:308-314) in fp32 on the exact-fp32 MFMA path — two small GEMMs per
        // forward; neither its weights nor its activations are rounded to fp16, so the logits carry only the error of
        // the representation itself
        float* a32 = (!repr_lowp && rep_last != nullptr) ? rep_last : g32;
        if (a32 == g32 && lnorm(x, m->fin_g, m->fin_b, nullptr, g32)) return 1;  // the normalised stream in fp32
        {
            ProfScope ps(m, st, PC_LM_DENSE, 2.0 * N * (double)E * E, (2.0 * NE + (double)E * E) * 4);
            ESMK_TRY(launch_gemm32(a32, E, (const float*)(pk + m->lm_w32), (const float*)(pk + m->lm_b), x, E, N, E, E, true, st));
        }
        if (lnorm(x, m->lm_lng, m->lm_lnb, nullptr, g32)) return 1;  // x (the residual stream) is dead: dense output
        {
            ProfScope ps(m, st, PC_LM_LOGITS, 2.0 * N * (double)E * m->V, (NE + (double)m->V * E + (double)N * m->V) * 4);
            ESMK_TRY(launch_gemm32(g32, E, (const float*)(pk + m->embed_f32), (const float*)(pk + m->lm_bias),
                                   (float*)logits_out_dev, m->V, N, m->V, E, false, st));
        }
    } else if (want_logits) {  // modules.py:308-314
        g = GemmArgs();
        g.A = h;
        g.W = pk + m->lm_w;
        g.bias = (const float*)(pk + m->lm_b);
        g.out = g32;
        g.M = N;
        g.N = E;
        g.K = Kp;
        if (gemm(PC_LM_DENSE, g, EPI_GELU_F32, 4)) return 1;
        if (lnorm(g32, m->lm_lng, m->lm_lnb, h, nullptr)) return 1;
        g = GemmArgs();
        g.A = h;
        g.W = pk + m->embed_op;
        g.bias = (const float*)(pk + m->lm_bias);
        g.out = logits_out_dev;
        g.M = N;
        g.N = m->V;
        g.K = Kp;
        if (gemm(PC_LM_LOGITS, g, EPI_STORE_F32, 4)) return 1;
    }
    if (fused_ct && S_ct > 0) {
        ProfScope ps(m, st, PC_CONTACTS, 0, 4.0 * B * ((double)T * T * 2 + 3.0 * L * H * T));
        ESMK_TRY(launch_contacts_fused_final((const float*)(ws + w.ct_acc), (float*)(ws + w.ct_row),
                                             (const float*)(ws + w.ct_col), (float*)(ws + w.ct_wt), tokens_dev,
                                             (const float*)(pk + m->ct_w), (const float*)(pk + m->ct_b),
                                             (float*)contacts_out_dev, B, H, L * H, T, m->D == 128 ? 128 : 64,
                                             m->cfg.pad_idx, m->cfg.eos_idx, m->cfg.prepend_bos, m->cfg.append_eos,
                                             st));
    } else if (want_contacts && S_ct > 0) {  // esm2.py:140-142
        // (an empty sequence has an empty [B,0,0] contact map: nothing to compute)
        ProfScope ps(m, st, PC_CONTACTS, 0, 2.0 * 4 * B * (double)L * H * T * T);
        ESMK_TRY(launch_contacts((const float*)attn_out_dev, tokens_dev, (const float*)(pk + m->ct_w),
                                 (const float*)(pk + m->ct_b), (float*)(ws + w.ct_scratch),
                                 (float*)contacts_out_dev, B, L * H, T, m->cfg.eos_idx,
                                 m->cfg.prepend_bos, m->cfg.append_eos, st));
    }
    return 0;
}

int esmk_ln_fold_enabled(const esmk_model* m) { return (!m || m->is_msa) ? -1 : (m->fold ? 1 : 0); }

int esmk_profile_begin(esmk_model* m) {
    if (!m) return fail("esmk_profile_begin: null model");
    for (auto& r : m->prof) {
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    m->prof.clear();
    m->prof_on = true;
    return 0;
}

int esmk_profile_end(esmk_model* m, esmk_profile_entry* out, int max_entries, int* n_out) {
    if (!m || !out || !n_out) return fail("esmk_profile_end: null argument");
    m->prof_on = false;
    std::vector<esmk_profile_entry> agg(PC_COUNT);
    for (int c = 0; c < PC_COUNT; ++c) {
        memset(&agg[c], 0, sizeof(esmk_profile_entry));
        strncpy(agg[c].name, kProfNames[c], sizeof(agg[c].name) - 1);
    }
    for (auto& r : m->prof) {
        ESMK_TRY(hipEventSynchronize(r.b));
        float ms = 0.f;
        ESMK_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        agg[r.cls].launches += 1;
        agg[r.cls].ms += ms;
        agg[r.cls].flops += r.flops;
        agg[r.cls].bytes += r.bytes;
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
    }
    m->prof.clear();
    int n = 0;
    for (int c = 0; c < PC_COUNT && n < max_entries; ++c)
        if (agg[c].launches > 0) out[n++] = agg[c];
    *n_out = n;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// single-kernel entry points
// ---------------------------------------------------------------------------------------------
int esmk_op_layernorm(const float* x_dev, const float* gamma_dev, const float* beta_dev,
                      void* y_dev, float* y32_dev, int rows, int E, int operand_dtype,
                      void* stream) {
    ESMK_TRY(launch_layernorm(x_dev, gamma_dev, beta_dev, y_dev, y32_dev, rows, E, operand_dtype,
                              (hipStream_t)stream));
    return 0;
}

int esmk_op_masked_row_mean(const void* x_dev, int x_dtype, const int32_t* count_dev, float* out_dev, int B, int T,
                            int E, int first_row, void* stream) {
    if (!x_dev || !count_dev || !out_dev) return fail("esmk_op_masked_row_mean: null argument");
    if (B <= 0 || T <= 0 || E <= 0 || E % 4 != 0 || first_row < 0 || first_row > T)
        return fail("esmk_op_masked_row_mean: need B, T > 0, E a positive multiple of 4, 0 <= first_row <= T");
    if (x_dtype != ESMK_DT_F32 && x_dtype != ESMK_DT_F16 && x_dtype != ESMK_DT_BF16)
        return fail("esmk_op_masked_row_mean: x_dtype must be ESMK_F32, ESMK_F16 or ESMK_BF16");
    ESMK_TRY(launch_masked_row_mean(x_dev, x_dtype, count_dev, out_dev, B, T, E, first_row, (hipStream_t)stream));
    return 0;
}

int esmk_op_linear(const void* a_dev, const void* w_dev, const float* bias_dev, void* out_dev,
                   int M, int N, int K, int epilogue, int operand_dtype, void* stream) {
    if (epilogue < 0 || epilogue > 4) return fail("esmk_op_linear: bad epilogue");
    GemmArgs g;
    g.A = a_dev;
    g.W = w_dev;
    g.bias = bias_dev;
    g.out = out_dev;
    g.M = M;
    g.N = N;
    g.K = K;
    if (operand_dtype & 0x100) g.force_generic = 1;  // test hook: force the generic 64x64 kernel
    if (operand_dtype & 0x200) g.force_old = 1;      // test hook: one-tile-per-workgroup 256x256 kernel
    g.panel_c = (operand_dtype >> 20) & 0x3f;         // tile-order experiments (tools/microbench.py)
    g.half_m = ((operand_dtype >> 28) & 3) == 1 ? 1 : (((operand_dtype >> 28) & 3) == 2 ? -1 : 0);  // 128-row tiles: force / never
    g.dbg = (operand_dtype >> 12) & 0xff;             // timing experiments (tools/microbench.py)
    operand_dtype &= 0xff;
    ESMK_TRY(launch_gemm(g, epilogue, operand_dtype, (hipStream_t)stream));
    return 0;
}

int esmk_op_split_weight(const void* w_dev, int w_dtype, void* w2_dev, int N, int K, void* stream) {
    if (!w_dev || !w2_dev) return fail("esmk_op_split_weight: null argument");
    if (N <= 0 || K <= 0 || K % 64 != 0) return fail("esmk_op_split_weight: need N > 0 and K a positive multiple of 64");
    ESMK_TRY(launch_convert2d_split(w_dev, w_dtype, w2_dev, (size_t)N, (size_t)K, (size_t)K, 0, 0, 64, (hipStream_t)stream));
    return 0;
}

int esmk_op_linear_split(const void* a_dev, const void* w2_dev, const float* bias_dev, void* out_dev, int M, int N, int K,
                         int epilogue, void* stream) {
    if (epilogue < 0 || epilogue > 4 || epilogue == EPI_GELU_F32) return fail("esmk_op_linear_split: epilogue must be 0, 1, 2 or 4");
    if (K % 64 != 0 || N % 8 != 0) return fail("esmk_op_linear_split: need K % 64 == 0 and N % 8 == 0");
    GemmArgs g;
    g.A = a_dev;
    g.W = w2_dev;
    g.bias = bias_dev;
    g.out = out_dev;
    g.M = M;
    g.N = N;
    g.K = 2 * K;
    g.a_row_bytes = (long long)K * 2;
    g.a_kt_repeat = 1;
    ESMK_TRY(launch_gemm(g, epilogue, ESMK_DT_F16, (hipStream_t)stream));
    return 0;
}

int esmk_debug_linear_splitk(const void* a_dev, const void* w_dev, float* partials_dev, int M, int N, int K,
                             int S, int operand_dtype, void* stream) {
    if (S < 1 || K % S != 0 || (K / S) % 64 != 0) return fail("esmk_debug_linear_splitk: K/S must be a multiple of 64");
    GemmArgs g;
    g.A = a_dev;
    g.W = w_dev;
    g.out = partials_dev;
    g.M = M;
    g.N = N;
    g.K = K / S;
    g.a_row_bytes = g.w_row_bytes = (long long)K * 2;  // rows keep the full-K stride
    g.batch = S;
    g.a_bo = g.w_bo = (long long)(K / S) * 2;           // slice s starts K/S operand elements further right
    g.o_bo = (long long)M * N * 4;
    ESMK_TRY(launch_gemm(g, EPI_STORE_F32, operand_dtype, (hipStream_t)stream));
    return 0;
}

int esmk_debug_gemm_timing(void* stamps_dev) {
    gemm8_set_timing((unsigned long long*)stamps_dev);
    gemm9_set_timing((unsigned long long*)stamps_dev);
    return 0;
}

int esmk_debug_mma_selftest(const void* a_dev, const void* b_dev, const float* c_dev, float* out_dev, int operand_dtype,
                            void* stream) {
    if (!a_dev || !b_dev || !c_dev || !out_dev) return fail("esmk_debug_mma_selftest: null argument");
    ESMK_TRY(launch_mma_keep_c_selftest(a_dev, b_dev, c_dev, out_dev, operand_dtype, (hipStream_t)stream));
    return 0;
}

// ---- LayerNorm fold as single ops (tests/test_ln_fold_gpu.py) ----------------------------------------------------
int esmk_op_rowstats(const float* x_dev, void* y_dev, float* mean_dev, float* rstd_dev, int rows, int E, int ldy,
                     int operand_dtype, void* stream) {
    if (!x_dev || !y_dev || !mean_dev || !rstd_dev) return fail("esmk_op_rowstats: null argument");
    ESMK_TRY(launch_rowstats(x_dev, y_dev, mean_dev, rstd_dev, rows, E, ldy, operand_dtype, (hipStream_t)stream));
    return 0;
}

int esmk_op_ln_finalize(const float* part_dev, float* mean_dev, float* rstd_dev, int rows, int parts, int E, void* stream) {
    if (!part_dev || !mean_dev || !rstd_dev) return fail("esmk_op_ln_finalize: null argument");
    ESMK_TRY(launch_ln_finalize(part_dev, mean_dev, rstd_dev, rows, parts, E, (hipStream_t)stream));
    return 0;
}

int esmk_op_fold_weight(const void* w_dev, int w_dtype, const float* gamma_dev, const float* beta_dev, void* dst_dev,
                        int dst_dtype, float* bias2_dev, int N, int K, int ld, void* stream) {
    if (!w_dev || !gamma_dev || !beta_dev || !dst_dev || !bias2_dev) return fail("esmk_op_fold_weight: null argument");
    if (N <= 0 || K <= 0 || ld < K) return fail("esmk_op_fold_weight: need N, K > 0 and ld >= K");
    ESMK_TRY(launch_fold_weight(w_dev, w_dtype, gamma_dev, beta_dev, dst_dev, dst_dtype, bias2_dev, (size_t)N, (size_t)K,
                                (size_t)ld, 0, 64, (hipStream_t)stream));
    return 0;
}

int esmk_op_linear_ln(const void* a_dev, const void* w_dev, const float* bias_dev, const float* bias2_dev, void* out_dev,
                      int M, int N, int K, int epilogue, int operand_dtype, const float* ln_rstd_dev, void* h16_dev, int ldh,
                      float* ln_part_dev, int ln_parts, const float* ln_mean_dev, int half_m, void* stream) {
    if (epilogue != EPI_GELU_T && epilogue != EPI_RESID_F32)
        return fail("esmk_op_linear_ln: epilogue must be 2 (consumer: gelu) or 4 (producer: residual)");
    GemmArgs g;
    g.A = a_dev;
    g.W = w_dev;
    g.bias = bias_dev;
    g.bias2 = bias2_dev;
    g.out = out_dev;
    g.M = M;
    g.N = N;
    g.K = K;
    g.half_m = half_m;
    if (epilogue == EPI_GELU_T) {
        if (!ln_rstd_dev || !bias_dev) return fail("esmk_op_linear_ln: the consumer needs ln_rstd and bias");
        g.ln_rstd = ln_rstd_dev;
    } else {
        if (!h16_dev || !ln_part_dev || !ln_mean_dev) return fail("esmk_op_linear_ln: the producer needs h16, ln_part and ln_mean");
        g.h16 = h16_dev;
        g.ldh = ldh;
        g.ln_part = ln_part_dev;
        g.ln_parts = ln_parts;
        g.ln_mean = ln_mean_dev;
    }
    ESMK_TRY(launch_gemm(g, epilogue, operand_dtype, (hipStream_t)stream));
    return 0;
}

int esmk_debug_set(const char* key, double value) {
    if (!key) return fail("esmk_debug_set: null key");
    if (gemm_set_knob(key, value)) return 0;
    if (strcmp(key, "attn_stagger") == 0) {
        attention_set_stagger((int)value);
        return 0;
    }
    return fail("esmk_debug_set: unknown key");
}

int esmk_debug_gemm_impl(int impl, int variant) {
    if (impl != 8 && impl != 9 && impl != 0) return fail("esmk_debug_gemm_impl: impl must be 8, 9 or 0 (automatic choice)");
    gemm_set_impl(impl, variant);
    return 0;
}

static int qkv_rope_impl(esmk_model* m, const void* a_dev, const void* wqkv_dev, const float* bias_dev,
                         const float* bias2_dev, const float* ln_rstd_dev, void* q_out, void* k_out, void* vt_out, int B, int T,
                         int log2_domain, void* stream);

int esmk_op_qkv_rope2(esmk_model* m, const void* a_dev, const void* wqkv_dev,
                      const float* bias_dev, void* q_out, void* k_out, void* vt_out, int B, int T,
                      int log2_domain, void* stream) {
    return qkv_rope_impl(m, a_dev, wqkv_dev, bias_dev, nullptr, nullptr, q_out, k_out, vt_out, B, T, log2_domain, stream);
}

int esmk_op_qkv_rope_ln(esmk_model* m, const void* a_dev, const void* wqkv_dev, const float* bias_dev,
                        const float* bias2_dev, const float* ln_rstd_dev, void* q_out, void* k_out, void* vt_out, int B, int T,
                        int log2_domain, void* stream) {
    if (!ln_rstd_dev || !bias_dev) return fail("esmk_op_qkv_rope_ln: ln_rstd and bias are required");
    return qkv_rope_impl(m, a_dev, wqkv_dev, bias_dev, bias2_dev, ln_rstd_dev, q_out, k_out, vt_out, B, T, log2_domain, stream);
}

static int qkv_rope_impl(esmk_model* m, const void* a_dev, const void* wqkv_dev, const float* bias_dev,
                         const float* bias2_dev, const float* ln_rstd_dev, void* q_out, void* k_out, void* vt_out, int B, int T,
                         int log2_domain, void* stream) {
    if (!m) return fail("esmk_op_qkv_rope: null model");
    if (m->D != 64 || m->Kp != m->E) return fail("esmk_op_qkv_rope: single-op entry point needs head_dim 64");
    hipStream_t st = (hipStream_t)stream;
    if (ensure_rope(m, T, st)) return 1;
    const int Tp = (T + 63) / 64 * 64;
    if (Tp != T)
        ESMK_TRY(hipMemsetAsync(vt_out, 0, (size_t)B * m->H * 64 * Tp * op_size(m->cfg.operand_dtype),
                                st));
    GemmArgs g;
    g.A = a_dev;
    g.W = wqkv_dev;
    g.bias = bias_dev;
    g.bias2 = bias2_dev;
    g.ln_rstd = ln_rstd_dev;
    g.M = B * T;
    g.N = 2 * m->E;
    g.K = m->E;
    g.q = q_out;
    g.k = k_out;
    g.vt = vt_out;
    g.cos = m->d_cos;
    g.sin = m->d_sin;
    g.T = T;
    g.H = m->H;
    g.E = m->E;
    g.Tp = Tp;
    // log2_domain: q also carries log2(e), the form esmk_op_attention / esmk_op_attention_probs take (esmk_forward's own)
    g.scaling = (log2_domain ? kLog2e : 1.0f) / sqrtf((float)m->D);
    if (gemm_qkv_one_launch(g)) {  // as esmk_forward: one launch where it saves rounds of tiles
        g.N = 3 * m->E;
        ESMK_TRY(launch_gemm(g, EPI_QKV_ALL, m->cfg.operand_dtype, st));
        return 0;
    }
    ESMK_TRY(launch_gemm(g, EPI_QKV_ROPE, m->cfg.operand_dtype, st));
    g.W = (const char*)wqkv_dev + (size_t)2 * m->E * m->E * op_size(m->cfg.operand_dtype);
    g.bias = bias_dev + 2 * m->E;
    if (bias2_dev) g.bias2 = bias2_dev + 2 * m->E;
    g.N = m->E;
    ESMK_TRY(launch_gemm(g, EPI_V_T, m->cfg.operand_dtype, st));
    return 0;
}

int esmk_op_qkv_rope(esmk_model* m, const void* a_dev, const void* wqkv_dev,
                     const float* bias_dev, void* q_out, void* k_out, void* vt_out, int B, int T,
                     void* stream) {
    return esmk_op_qkv_rope2(m, a_dev, wqkv_dev, bias_dev, q_out, k_out, vt_out, B, T, 0, stream);
}

int esmk_op_attention(const void* q_dev, const void* k_dev, const void* vt_dev,
                      const float* key_bias_dev, void* ctx_out, float* lse_out, int B, int H,
                      int T, int operand_dtype, void* stream) {
    const int Tp = (T + 63) / 64 * 64;
    ESMK_TRY(launch_attention(q_dev, k_dev, vt_dev, key_bias_dev, nullptr, ctx_out, lse_out, B, H, T,
                              Tp, operand_dtype, (hipStream_t)stream));
    return 0;
}

int esmk_op_attention_probs(const void* q_dev, const void* k_dev, const float* lse_dev,
                            const float* key_bias_dev, float* probs_out, int B, int H, int T,
                            int layer, int num_layers_total, int operand_dtype, void* stream) {
    ESMK_TRY(launch_attention_probs(q_dev, k_dev, lse_dev, key_bias_dev, probs_out, B, H, T, layer,
                                    num_layers_total, operand_dtype, (hipStream_t)stream));
    return 0;
}

int esmk_op_contacts(const float* attn_dev, const int64_t* tokens_dev, const float* w_dev,
                     const float* b_dev, float* scratch_dev, float* out_dev, int B, int C, int T,
                     int eos_idx, int prepend_bos, int append_eos, void* stream) {
    ESMK_TRY(launch_contacts(attn_dev, tokens_dev, w_dev, b_dev, scratch_dev, out_dev, B, C, T,
                             eos_idx, prepend_bos, append_eos, (hipStream_t)stream));
    return 0;
}

}  // extern "C"

