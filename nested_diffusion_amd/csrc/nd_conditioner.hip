// nd_conditioner.hip -- the mapping network as ONE C-level call (gfx950 only): host-side sequencing of the ViT and
// mapping-MLP kernels of nd_vit.hip / nd_gemm_f32.hip / nd_attention.hip / nd_ops.hip.
//
// Reference (file:line relative to the reference checkout):
//   diffusion/classification_train_separately.py:249-275   cond_pred_model = {'vit', 'mlps'} (whole-module pickles)
//   diffusion/classification_train_separately.py:330-348   compute_guiding_prediction
//   mapping/models/mlp.py:23-29                             Classifier.forward
//   timm 0.4.12 vision_transformer.py (third-party, absent: restated from its published semantics)  Block.forward
//
// Nothing here allocates or synchronises, so a call can be captured into a hipGraph (nd_predict_batch does).
#include <hip/hip_runtime.h>
#include <atomic>
#include <vector>
#include <cstring>
#include <cstdlib>
#include "../../include/nested_diffusion.h"

int nd_set_err(int code, const char* fmt, ...);
// nd_ops.hip: Classifier.forward on packed activations (one input pack, hidden layers written in streaming order)
int nd_mlp_chain(const float* x, const void* const* wpk, const float* const* bias, const int* dims, float* const* hid, float* logits,
                 int M, int dtype, void* ws, size_t ws_bytes, void* stream);
// ... and the same with layers 2..4 of several MLPs sharing launches (their first layers run one by one, each behind its prefix block)
bool nd_mlp_tail_batchable(const int* dims, int M, int nm, int dtype);
int nd_mlp_chain_first(const float* x, const void* w1pk, const float* bias1, const int* dims, float* hid0, int M, int dtype, void* ws,
                       size_t ws_bytes, void* stream);
int nd_mlp_chain_tail(int nm, const nd_mlp_weights* w, const int* dims, float* const* hid0, float* const* hid1, float* const* hid2, float* logits,
                      size_t logits_stride, int M, int dtype, void* stream);

static std::atomic<unsigned long long> g_cond_serial{1};

struct nd_cond_s {
    // changes whenever anything a recorded launch sequence depends on changes (creation, workspace, any weight pointer): graphs
    // recorded against an older value must not be replayed (a freed handle's address can be handed out again by the allocator)
    unsigned long long serial = g_cond_serial.fetch_add(1);
    nd_cond_config cfg{};
    nd_patch_embed_weights pe{};
    std::vector<nd_vit_block_weights> blocks;
    std::vector<nd_mlp_weights> mlps;
    std::vector<char> have_block, have_mlp;
    bool have_pe = false;
    char* ws = nullptr;
    size_t ws_bytes = 0;
    // carved activations
    float *cols = nullptr, *tok = nullptr, *mid_tok = nullptr, *h = nullptr, *qkv = nullptr, *att = nullptr, *fc1 = nullptr;
    // ND_DTYPE_F32_SPLIT: frag32b3 images of the GEMM inputs (csrc/nd_b9.hpp): xs = the current [R, kpe | E] input, fc1s = GELU(fc1)
    void *xs = nullptr, *fc1s = nullptr;
    void* qkv_img = nullptr;          // ND_DTYPE_F32_SPLIT: the attention's operand images (nd_gemm_split_qkv), where the shape supports them
    int qkv_img_tokens = 0;           // token count the image buffer is sized for (the prefix blocks' patch count), 0: none reserved
    float* m[3] = {nullptr, nullptr, nullptr};       // the mapping MLPs' hidden activations: n_mlps slices of m_stride[l] floats each
    size_t m_stride[3] = {0, 0, 0};
    void *gemm_ws = nullptr, *lin_ws = nullptr;
    size_t gemm_ws_bytes = 0, lin_ws_bytes = 0;
};

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

static int check_cfg(const nd_cond_config* c) {
    if (!c) return nd_set_err(ND_ERR_ARG, "cfg is NULL");
    if (c->patch < 4 || (c->patch % 4) || c->img_size < c->patch || (c->img_size % c->patch) || c->in_chans < 1)
        return nd_set_err(ND_ERR_ARG, "patch must be a multiple of 4 dividing img_size");
    if (c->num_heads < 1 || c->embed_dim != 64 * c->num_heads) return nd_set_err(ND_ERR_ARG, "embed_dim must be 64 * num_heads");
    if (c->operand_dtype != ND_DTYPE_F32 && c->operand_dtype != ND_DTYPE_F16 && c->operand_dtype != ND_DTYPE_F32_SPLIT)
        return nd_set_err(ND_ERR_ARG, "unknown operand_dtype");
    const int km = c->operand_dtype == ND_DTYPE_F32 ? 16 : 32;
    const int kpe = c->in_chans * c->patch * c->patch;
    if ((c->embed_dim % km) || (c->mlp_hidden % km) || c->mlp_hidden < km || (kpe % km))
        return nd_set_err(ND_ERR_ARG, "embed_dim, mlp_hidden and in_chans*patch^2 must be multiples of %d", km);
    if (c->n_mlps < 1 || c->n_blocks < c->n_mlps) return nd_set_err(ND_ERR_ARG, "need 1 <= n_mlps <= n_blocks");
    const int kw = c->operand_dtype == ND_DTYPE_F16 ? 32 : 16;      // the mapping MLPs stream fp32 (F32, F32_SPLIT) or fp16 weights
    for (int i = 0; i < 3; ++i)
        if (c->mlp_widths[i] < kw || (c->mlp_widths[i] % kw)) return nd_set_err(ND_ERR_ARG, "mlp_widths must be multiples of %d", kw);
    if (c->num_classes < 1 || c->max_batch < 1) return nd_set_err(ND_ERR_ARG, "num_classes / max_batch invalid");
    const int ntok = (c->img_size / c->patch) * (c->img_size / c->patch);
    if (c->max_tokens < ntok || c->max_tokens > 256) return nd_set_err(ND_ERR_ARG, "max_tokens must be in [%d, 256]", ntok);
    if (!(c->ln_eps > 0.f)) return nd_set_err(ND_ERR_ARG, "ln_eps must be > 0");
    return ND_OK;
}

static inline size_t zmax(size_t a, size_t b) { return a > b ? a : b; }

// carve with base == nullptr computes the size only
static void carve(nd_cond_s* c, char* base, size_t* total) {
    const nd_cond_config& g = c->cfg;
    const size_t B = g.max_batch, N = g.max_tokens, E = g.embed_dim, Hd = g.mlp_hidden;
    const size_t ntok = (size_t)(g.img_size / g.patch) * (g.img_size / g.patch), kpe = (size_t)g.in_chans * g.patch * g.patch;
    const size_t R = B * N;
    size_t off = 0;
    auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off = al256(off + bytes); return p; };
    c->cols = (float*)take(B * ntok * kpe * 4);
    c->tok = (float*)take(R * E * 4);
    c->mid_tok = (float*)take(R * E * 4);
    c->h = (float*)take(R * E * 4);
    c->qkv = (float*)take(R * 3 * E * 4);
    c->att = (float*)take(R * E * 4);
    const bool split = g.operand_dtype == ND_DTYPE_F32_SPLIT;
    const int dt = split ? ND_DTYPE_F32 : g.operand_dtype;       // what the attention kernel and the mapping MLPs run in
    if (split) {
        c->fc1 = nullptr;
        c->xs = take(nd_split_bytes((int)R, (int)(kpe > E ? kpe : E)));
        c->fc1s = take(nd_split_bytes((int)R, (int)Hd));
        // the attention's operand images (92 MB at B = 32, 196 tokens, 12 heads) only where a block can use them: the prefix blocks run
        // at the patch count `ntok`; a token count the image form does not take (the full forward's 197: N % 4 != 0) or ND_ATT_F32=1
        // keeps the fp32 qkv buffer above and reserves nothing here
        c->qkv_img_tokens = (nd_qkv_images_supported((int)ntok, g.num_heads) && !getenv("ND_ATT_F32")) ? (int)ntok : 0;
        c->qkv_img = c->qkv_img_tokens ? take(nd_qkv_images_bytes((int)B, c->qkv_img_tokens, g.num_heads)) : nullptr;
    } else {
        c->fc1 = (float*)take(R * Hd * 4);
        c->xs = c->fc1s = nullptr;
    }
    for (int i = 0; i < 3; ++i) {                                     // packed: whole 16-row tiles; one slice per mapping MLP
        c->m_stride[i] = (((B + 15) / 16 * 16) * (size_t)g.mlp_widths[i] + 63) & ~(size_t)63;
        c->m[i] = (float*)take((size_t)g.n_mlps * c->m_stride[i] * 4);
    }
    size_t gw = 0;
    const int rows[2] = {(int)(B * ntok), (int)R};
    for (int r : rows) {
        if (split) {
            gw = zmax(gw, nd_gemm_split_workspace_bytes(r, (int)kpe, (int)E));
            gw = zmax(gw, nd_gemm_split_workspace_bytes(r, (int)E, (int)(3 * E)));
            gw = zmax(gw, nd_gemm_split_workspace_bytes(r, (int)E, (int)E));
            gw = zmax(gw, nd_gemm_split_workspace_bytes(r, (int)E, (int)Hd));
            gw = zmax(gw, nd_gemm_split_workspace_bytes(r, (int)Hd, (int)E));
            continue;
        }
        gw = zmax(gw, nd_gemm_workspace_bytes(r, (int)kpe, (int)E, dt));
        gw = zmax(gw, nd_gemm_workspace_bytes(r, (int)E, (int)(3 * E), dt));
        gw = zmax(gw, nd_gemm_workspace_bytes(r, (int)E, (int)E, dt));
        gw = zmax(gw, nd_gemm_workspace_bytes(r, (int)E, (int)Hd, dt));
        gw = zmax(gw, nd_gemm_workspace_bytes(r, (int)Hd, (int)E, dt));
    }
    c->gemm_ws_bytes = gw;
    c->gemm_ws = take(gw + 16);
    const int dims[5] = {(int)(ntok * E), g.mlp_widths[0], g.mlp_widths[1], g.mlp_widths[2], g.num_classes};
    size_t lw = 0;
    for (int l = 0; l < 4; ++l)
        for (int b = 1; b <= (int)B; ++b) lw = zmax(lw, nd_linear_workspace_bytes(b, dims[l], dims[l + 1], dt));   // plans differ with the row count
    c->lin_ws_bytes = lw;
    c->lin_ws = take(lw);
    *total = off;
}

extern "C" size_t nd_cond_workspace_bytes(const nd_cond_config* cfg) {
    if (check_cfg(cfg) != ND_OK) return 0;
    nd_cond_s tmp;
    tmp.cfg = *cfg;
    size_t total = 0;
    carve(&tmp, nullptr, &total);
    return total;
}

extern "C" int nd_cond_create(const nd_cond_config* cfg, nd_cond* out) {
    if (!out) return nd_set_err(ND_ERR_ARG, "out is NULL");
    int rc = check_cfg(cfg);
    if (rc != ND_OK) return rc;
    nd_cond_s* c = new nd_cond_s();
    c->cfg = *cfg;
    c->blocks.resize(cfg->n_blocks);
    c->mlps.resize(cfg->n_mlps);
    c->have_block.assign(cfg->n_blocks, 0);
    c->have_mlp.assign(cfg->n_mlps, 0);
    *out = c;
    return ND_OK;
}

extern "C" const nd_cond_config* nd_cond_get_config(nd_cond c) { return c ? &c->cfg : nullptr; }
unsigned long long nd_cond_serial(nd_cond c) { return c ? c->serial : 0; }
static void touch(nd_cond_s* c) { c->serial = g_cond_serial.fetch_add(1); }

extern "C" int nd_cond_destroy(nd_cond c) {
    delete c;
    return ND_OK;
}

extern "C" int nd_cond_bind_workspace(nd_cond c, void* ws, size_t bytes) {
    if (!c || !ws) return nd_set_err(ND_ERR_ARG, "conditioner / workspace is NULL");
    if ((uintptr_t)ws & 255) return nd_set_err(ND_ERR_ARG, "workspace must be 256-byte aligned");
    // size first, on a scratch copy: a rejected buffer must leave the handle's activation pointers (and the graphs recorded under
    // its serial) exactly as they were
    size_t need = 0;
    {
        nd_cond_s tmp;
        tmp.cfg = c->cfg;
        carve(&tmp, nullptr, &need);
    }
    if (bytes < need) return nd_set_err(ND_ERR_ARG, "workspace too small: %zu < %zu", bytes, need);
    carve(c, (char*)ws, &need);
    c->ws = (char*)ws;
    c->ws_bytes = bytes;
    touch(c);
    return ND_OK;
}

template <typename T>
static int all_set(const T* w, const char* what) {
    const void* const* pp = reinterpret_cast<const void* const*>(w);
    for (size_t i = 0; i < sizeof(T) / sizeof(void*); ++i)
        if (!pp[i]) return nd_set_err(ND_ERR_ARG, "%s pointer #%zu is NULL", what, i);
    return ND_OK;
}

extern "C" int nd_cond_set_patch_embed(nd_cond c, const nd_patch_embed_weights* w) {
    if (!c || !w) return nd_set_err(ND_ERR_ARG, "NULL argument");
    int rc = all_set(w, "nd_patch_embed_weights");
    if (rc != ND_OK) return rc;
    c->pe = *w;
    c->have_pe = true;
    touch(c);
    return ND_OK;
}

extern "C" int nd_cond_set_block(nd_cond c, int block, const nd_vit_block_weights* w) {
    if (!c || !w) return nd_set_err(ND_ERR_ARG, "NULL argument");
    if (block < 0 || block >= c->cfg.n_blocks) return nd_set_err(ND_ERR_ARG, "block %d outside [0,%d)", block, c->cfg.n_blocks);
    int rc = all_set(w, "nd_vit_block_weights");
    if (rc != ND_OK) return rc;
    c->blocks[block] = *w;
    c->have_block[block] = 1;
    touch(c);
    return ND_OK;
}

extern "C" int nd_cond_set_mlp(nd_cond c, int i, const nd_mlp_weights* w) {
    if (!c || !w) return nd_set_err(ND_ERR_ARG, "NULL argument");
    if (i < 0 || i >= c->cfg.n_mlps) return nd_set_err(ND_ERR_ARG, "mlp %d outside [0,%d)", i, c->cfg.n_mlps);
    int rc = all_set(w, "nd_mlp_weights");
    if (rc != ND_OK) return rc;
    c->mlps[i] = *w;
    c->have_mlp[i] = 1;
    touch(c);
    return ND_OK;
}

#define ND_COND_MAX_BATCHED 8        // = ND_INLINE_DESCS (csrc/nd_common.hpp): members whose descriptors travel by value in one launch
#define ND_TRY(call)          \
    do {                      \
        int _rc = (call);     \
        if (_rc != ND_OK) return _rc; \
    } while (0)

static int vit_block(nd_cond_s* c, int block, const float* tin, float* tout, int B, int N, void* st) {
    const nd_cond_config& g = c->cfg;
    const nd_vit_block_weights& w = c->blocks[block];
    const int E = g.embed_dim, Hd = g.mlp_hidden, R = B * N, dt = g.operand_dtype;
    if (dt == ND_DTYPE_F32_SPLIT) {
        // the same block with the four Linear layers on the bf16 matrix pipe, exact fp32 products (csrc/nd_b9.hpp): the weights are
        // frag32b3 images, every GEMM input is written as one by its producer (LayerNorm, attention and the fc1 epilogue)
        ND_TRY(nd_layernorm_split(tin, w.norm1_w, w.norm1_b, c->xs, R, E, g.ln_eps, st));
        if (c->qkv_img && N <= c->qkv_img_tokens && nd_qkv_images_supported(N, g.num_heads) && !getenv("ND_ATT_F32")) {
            // attention on the bf16 matrix pipe too: the qkv Linear writes q, k and v^T as the MFMA operand images of the attention kernel
            ND_TRY(nd_gemm_split_qkv(c->xs, w.qkv_w, w.qkv_b, c->qkv_img, B, N, g.num_heads, E, st));
            ND_TRY(nd_attention_images(c->qkv_img, c->xs, 1, B, N, g.num_heads, st));
        } else {        // N % 4 != 0 (the full forward's 197 tokens) or ND_ATT_F32=1: fp32 qkv, f32-input-MFMA attention (rounds 1-4)
            ND_TRY(nd_gemm_split(c->xs, w.qkv_w, w.qkv_b, nullptr, c->qkv, nullptr, R, E, 3 * E, ND_ACT_NONE, c->gemm_ws, c->gemm_ws_bytes, st));
            ND_TRY(nd_attention_split(c->qkv, c->xs, B, N, g.num_heads, 64, st));
        }
        ND_TRY(nd_gemm_split(c->xs, w.proj_w, w.proj_b, tin, c->mid_tok, nullptr, R, E, E, ND_ACT_NONE, c->gemm_ws, c->gemm_ws_bytes, st));
        ND_TRY(nd_layernorm_split(c->mid_tok, w.norm2_w, w.norm2_b, c->xs, R, E, g.ln_eps, st));
        ND_TRY(nd_gemm_split(c->xs, w.fc1_w, w.fc1_b, nullptr, nullptr, c->fc1s, R, E, Hd, ND_ACT_GELU, c->gemm_ws, c->gemm_ws_bytes, st));
        ND_TRY(nd_gemm_split(c->fc1s, w.fc2_w, w.fc2_b, c->mid_tok, tout, nullptr, R, Hd, E, ND_ACT_NONE, c->gemm_ws, c->gemm_ws_bytes, st));
        return ND_OK;
    }
    // x = x + attn(norm1(x))
    ND_TRY(nd_layernorm(tin, w.norm1_w, w.norm1_b, c->h, R, E, g.ln_eps, st));
    ND_TRY(nd_gemm_bias_act(c->h, w.qkv_w, w.qkv_b, nullptr, c->qkv, R, E, 3 * E, ND_ACT_NONE, dt, c->gemm_ws, c->gemm_ws_bytes, st));
    ND_TRY(nd_attention(c->qkv, c->att, B, N, g.num_heads, 64, dt, st));
    ND_TRY(nd_gemm_bias_act(c->att, w.proj_w, w.proj_b, tin, c->mid_tok, R, E, E, ND_ACT_NONE, dt, c->gemm_ws, c->gemm_ws_bytes, st));
    // x = x + mlp(norm2(x)), exact-erf GELU
    ND_TRY(nd_layernorm(c->mid_tok, w.norm2_w, w.norm2_b, c->h, R, E, g.ln_eps, st));
    ND_TRY(nd_gemm_bias_act(c->h, w.fc1_w, w.fc1_b, nullptr, c->fc1, R, E, Hd, ND_ACT_GELU, dt, c->gemm_ws, c->gemm_ws_bytes, st));
    ND_TRY(nd_gemm_bias_act(c->fc1, w.fc2_w, w.fc2_b, c->mid_tok, tout, R, Hd, E, ND_ACT_NONE, dt, c->gemm_ws, c->gemm_ws_bytes, st));
    return ND_OK;
}

extern "C" int nd_vit_block(nd_cond c, int block, const float* tok_in, float* tok_out, int B, int N, void* stream) {
    if (!c || !c->ws) return nd_set_err(ND_ERR_STATE, "conditioner workspace not bound");
    if (!tok_in || !tok_out) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    if (block < 0 || block >= c->cfg.n_blocks || !c->have_block[block]) return nd_set_err(ND_ERR_STATE, "block %d not set", block);
    if (B < 1 || B > c->cfg.max_batch || N < 1 || N > c->cfg.max_tokens)
        return nd_set_err(ND_ERR_ARG, "B=%d / N=%d outside [1,%d] / [1,%d]", B, N, c->cfg.max_batch, c->cfg.max_tokens);
    return vit_block(c, block, tok_in, tok_out, B, N, stream);
}

// The first n_used of the conditioner's mapping MLPs (and the prefix blocks they need): nd_predict_batch with an ensemble of
// fewer noise estimators than mapping MLPs (the reference samples `selected_block_indices` ∩ available checkpoints,
// classification_train_separately.py:275, 769) computes only the conditions it samples from.
int nd_guiding_prediction_first(nd_cond c, const float* images, float* logits_out, float* yhat_out, int B, int n_used, void* stream) {
    if (!c || !c->ws) return nd_set_err(ND_ERR_STATE, "conditioner workspace not bound");
    if (!images || !logits_out) return nd_set_err(ND_ERR_ARG, "NULL tensor");
    const nd_cond_config& g = c->cfg;
    if (B < 1 || B > g.max_batch) return nd_set_err(ND_ERR_ARG, "B=%d outside [1,%d]", B, g.max_batch);
    if (n_used < 1 || n_used > g.n_mlps) return nd_set_err(ND_ERR_ARG, "n_used=%d outside [1,%d]", n_used, g.n_mlps);
    if (!c->have_pe) return nd_set_err(ND_ERR_STATE, "patch embedding not set");
    for (int i = 0; i < n_used; ++i)
        if (!c->have_block[i] || !c->have_mlp[i]) return nd_set_err(ND_ERR_STATE, "block / mlp %d not set", i);
    const int E = g.embed_dim, gs = g.img_size / g.patch, ntok = gs * gs, kpe = g.in_chans * g.patch * g.patch;
    const bool split = g.operand_dtype == ND_DTYPE_F32_SPLIT;
    const int dt = split ? ND_DTYPE_F32 : g.operand_dtype;
    const int C = g.num_classes;
    // tmp = vit.patch_embed(x); vit.pos_drop is the identity in eval; no cls token, no pos_embed (:337-338, quirk Q3)
    if (split) {
        ND_TRY(nd_patchify_split(images, c->xs, B, g.in_chans, g.img_size, g.img_size, g.patch, stream));
        ND_TRY(nd_gemm_split(c->xs, c->pe.proj_w, c->pe.proj_b, nullptr, c->tok, nullptr, B * ntok, kpe, E, ND_ACT_NONE, c->gemm_ws, c->gemm_ws_bytes,
                             stream));
    } else {
        ND_TRY(nd_patchify(images, c->cols, B, g.in_chans, g.img_size, g.img_size, g.patch, stream));
        ND_TRY(nd_gemm_bias_act(c->cols, c->pe.proj_w, c->pe.proj_b, nullptr, c->tok, B * ntok, kpe, E, ND_ACT_NONE, dt, c->gemm_ws,
                                c->gemm_ws_bytes, stream));
    }
    const int dims[5] = {ntok * E, g.mlp_widths[0], g.mlp_widths[1], g.mlp_widths[2], C};
    // ND_MLP_TAIL_PER_MEMBER=1: every MLP's four layers one after the other (rounds 1-4), the A/B switch of the batched tail
    const bool batch_tail = n_used <= ND_COND_MAX_BATCHED && nd_mlp_tail_batchable(dims, B, n_used, dt) && !getenv("ND_MLP_TAIL_PER_MEMBER");
    for (int i = 0; i < n_used; ++i) {
        // member i's prefix blocks[0..i] reuse member i-1's tokens (:339-340 recomputes them from patch_embed: same values)
        ND_TRY(vit_block(c, i, c->tok, c->tok, B, ntok, stream));
        // mlps[i](tmp): reshape(-1, 196*768) -> 3 x (Linear, ReLU) -> Linear (mapping/models/mlp.py:23-29)
        const nd_mlp_weights& w = c->mlps[i];
        float* logits = logits_out + (size_t)i * B * C;
        float* hid[3] = {c->m[0] + (size_t)i * c->m_stride[0], c->m[1] + (size_t)i * c->m_stride[1], c->m[2] + (size_t)i * c->m_stride[2]};
        // layer 1 (the 150528-wide weight stream) right behind its block; layers 2..4 of all members afterwards in three shared launches
        if (batch_tail) ND_TRY(nd_mlp_chain_first(c->tok, w.w_packed[0], w.bias[0], dims, hid[0], B, dt, c->lin_ws, c->lin_ws_bytes, stream));
        else ND_TRY(nd_mlp_chain(c->tok, w.w_packed, w.bias, dims, hid, logits, B, dt, c->lin_ws, c->lin_ws_bytes, stream));
    }
    if (batch_tail) {
        float *h0[ND_COND_MAX_BATCHED], *h1[ND_COND_MAX_BATCHED], *h2[ND_COND_MAX_BATCHED];
        for (int i = 0; i < n_used; ++i) {
            h0[i] = c->m[0] + (size_t)i * c->m_stride[0]; h1[i] = c->m[1] + (size_t)i * c->m_stride[1]; h2[i] = c->m[2] + (size_t)i * c->m_stride[2];
        }
        ND_TRY(nd_mlp_chain_tail(n_used, c->mlps.data(), dims, h0, h1, h2, logits_out, (size_t)B * C, B, dt, stream));
    }
    // softmax of every condition's logits (:755-758): rows are independent and [n_used][B][C] is contiguous -- one launch, not n_used
    if (yhat_out) ND_TRY(nd_softmax_rows(logits_out, yhat_out, n_used * B, C, stream));
    return ND_OK;
}

extern "C" int nd_guiding_prediction(nd_cond c, const float* images, float* logits_out, float* yhat_out, int B, void* stream) {
    if (!c) return nd_set_err(ND_ERR_STATE, "conditioner workspace not bound");
    return nd_guiding_prediction_first(c, images, logits_out, yhat_out, B, c->cfg.n_mlps, stream);
}
