/*
 * nested_diffusion.h -- C ABI of libnd_hip.so, the MI355X (gfx950) implementation of the
 * nested-diffusion (LaDiNE) inference hot path.
 *
 * The reference (xingbpshen/nested-diffusion) has no FFI: its hot path is plain Python calling
 * torch ops.  Each entry point below therefore names the reference Python call(s) it replaces
 * (file:line relative to the reference checkout).  The host side that mirrors the reference's
 * Python API lives in nested_diffusion_amd/ and binds this library with ctypes.
 *
 * Conventions
 *  - plain C types only; every `*_dev` pointer is a DEVICE pointer owned by the caller
 *    (PyTorch-ROCm tensors in practice); the library never allocates device memory --
 *    the caller provides one workspace of nd_workspace_bytes().
 *  - `stream` is a hipStream_t passed as void* (NULL = default stream).
 *  - every function returns 0 on success, <0 on error; nd_last_error() gives the message.
 *    No C++ exception crosses the ABI.
 *  - fp32 at the ABI: every tensor handed in or out is fp32 row-major (int64 votes aside) unless a declaration says "image".
 *    INSIDE, operands live in MFMA lane order: frag16 (fp32, the weight-streaming layers), frag32b3 (three exact bf16 pieces per
 *    fp32 value: every MFMA-bound layer -- the ViT Linear layers and attention, the ConditionalLinear blocks above 128 rows -- runs on
 *    the bf16 matrix pipe with EXACT fp32 products and fp32 accumulation; csrc/nd_b9.hpp), frag32h (fp16, the fp16-operand mode).
 *    The arithmetic is the reference's fp32 arithmetic in another summation order in every default mode.
 *  - one handle per GPU per process; not thread-safe for concurrent calls on one handle.
 */
#ifndef NESTED_DIFFUSION_H
#define NESTED_DIFFUSION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ND_OK 0
#define ND_ERR_ARG (-1)
#define ND_ERR_HIP (-2)
#define ND_ERR_STATE (-3)

/* activation codes for the generic linear entry points */
#define ND_ACT_NONE 0
#define ND_ACT_SOFTPLUS 1 /* torch softplus beta=1 threshold=20 (latent_model.py:130,133,176,180,183) */
#define ND_ACT_RELU 2     /* mapping/models/mlp.py:25-27 */
#define ND_ACT_GELU 3     /* exact erf GELU, timm 0.4.12 Mlp */

typedef struct nd_handle_s *nd_handle;

/* Dimensions of one ensemble: K ConditionalModel members (latent_model.py:108-167, arch 'linear',
 * guidance=True) that share shapes.  Config dims: y_dim 2, data_dim 150528, hidden 4096,
 * feature 4096 (configs/chest_x_ray.yml:5,11,14-15). */
typedef struct {
    int32_t y_dim;       /* C  = config.data.num_classes (1..8: a library limit the reference does not have -- the step head keeps a
                          *     row's C state values and C eps sums in registers; the shipped configs use 2) */
    int32_t data_dim;    /* D  = config.model.data_dim          (multiple of 16) */
    int32_t hidden_dim;  /* H  = config.model.hidden_dim        (multiple of 16) */
    int32_t feature_dim; /* F  = config.model.feature_dim       (multiple of 16) */
    int32_t n_steps;     /* T  = config.diffusion.timesteps; embed tables hold T+1 rows */
    int32_t n_members;   /* K  (1..255) */
    int32_t max_batch;   /* largest B (images) per call */
    int32_t max_rows;    /* largest M = B * mc_trials per call */
    int32_t operand_dtype; /* ND_DTYPE_F32 (0): the reference's arithmetic.  ND_DTYPE_F16 (1): weights of the five large Linear
                            * layers and their input activations are held in fp16 (products exact, fp32 accumulation, fp32
                            * epilogues, tables and state) -- BASELINE config 5; the reference has no such mode.  Needs
                            * data_dim, hidden_dim, feature_dim % 32 == 0. */
} nd_config;
#define ND_DTYPE_F32 0
#define ND_DTYPE_F16 1
/* nd_cond_config only: fp32 arithmetic as ND_DTYPE_F32 -- exact products, fp32 sums -- with the ViT's Linear layers on the bf16 matrix
 * pipe (nd_gemm_split): their weights are handed over as frag32b3 images (nd_split_rows of the fp32 [out,in] weight, made once at
 * load), embed_dim, mlp_hidden and in_chans*patch^2 must be multiples of 32.  The mapping MLPs and the attention run as in F32. */
#define ND_DTYPE_F32_SPLIT 2

/* Device pointers to ONE member's raw parameters, exactly the tensors of
 * ConditionalModel.state_dict() (key in the comment; shapes for config dims). */
typedef struct {
    const float *enc0_w, *enc0_b;                       /* encoder_x.0.{weight [H,D], bias [H]} */
    const float *bn0_w, *bn0_b, *bn0_mean, *bn0_var;    /* encoder_x.1.* [H] */
    const float *enc3_w, *enc3_b;                       /* encoder_x.3.{weight [H,H], bias} */
    const float *bn1_w, *bn1_b, *bn1_mean, *bn1_var;    /* encoder_x.4.* [H] */
    const float *enc6_w, *enc6_b;                       /* encoder_x.6.{weight [F,H], bias} */
    const float *norm_w, *norm_b, *norm_mean, *norm_var;/* norm.* [F] */
    const float *lin1_w, *lin1_b, *emb1;                /* lin1.lin.{weight [F,2C], bias}, lin1.embed.weight [T+1,F] */
    const float *un1_w, *un1_b, *un1_mean, *un1_var;    /* unetnorm1.* */
    const float *lin2_w, *lin2_b, *emb2;                /* lin2.lin.{weight [F,F], bias}, lin2.embed.weight */
    const float *un2_w, *un2_b, *un2_mean, *un2_var;    /* unetnorm2.* */
    const float *lin3_w, *lin3_b, *emb3;                /* lin3.* */
    const float *un3_w, *un3_b, *un3_mean, *un3_var;    /* unetnorm3.* */
    const float *lin4_w, *lin4_b;                       /* lin4.{weight [C,F], bias [C]} */
} nd_member_weights;

const char *nd_last_error(void);
/* "gfx950 f32 <build id>" -- lets the host assert the native library is the one loaded. */
const char *nd_version(void);

/* ---- ensemble handle --------------------------------------------------------------------- */
int nd_create(const nd_config *cfg, nd_handle *out);
int nd_destroy(nd_handle h);
/* Bytes of device workspace the handle needs (folded tables, activations, split-K slabs). */
size_t nd_workspace_bytes(const nd_config *cfg);
int nd_bind_workspace(nd_handle h, void *workspace_dev, size_t bytes);

/* Replaces ConditionalModel(...).load_state_dict(state['noise_estimator']) + .eval() + .to(device)
 * (classification_train_separately.py:685-697, 773).  Repacks the five weight matrices into the
 * workspace in fragment order (the caller may free its tensors when the call returns) and folds eval-mode BatchNorm1d, the Linear bias and the per-timestep
 * Embedding gain into scale/shift tables:  BN(g_t * (W h + b)) = a_t * (W h) + c_t. */
int nd_load_member(nd_handle h, int member, const nd_member_weights *w, void *stream);

/* alphas / one_minus_alphas_bar_sqrt, as handed to p_sample_loop (diffusion_utils.py:133;
 * built at classification_train_separately.py:215-226).  Device arrays of length T, copied. */
int nd_set_schedule(nd_handle h, const float *alphas_dev, const float *omabs_dev, int T, void *stream);

/* xe = norm(encoder_x(x)) for members [member0, member0+n_members)  -- latent_model.py:170-171.
 * t-invariant, so evaluated ONCE per (member, batch) instead of once per denoising step.
 * x_dev [B, D] row-major (images_224_flat, classification_train_separately.py:747). */
int nd_encode(nd_handle h, int member0, int n_members, const float *x_dev, int B, void *stream);

/* One eps_theta evaluation of the t-dependent trunk on the cached xe:
 * ConditionalModel.forward lines latent_model.py:172-184.  y_dev [M,C], yhat_dev [B,C],
 * eps_out_dev [M,C]; row m uses image m % B.  Used for unit parity tests and the nn.Module mirror. */
int nd_eps_theta(nd_handle h, int member, const float *y_dev, const float *yhat_dev, int t,
                 float *eps_out_dev, int B, int mc, void *stream);

/* One reverse step for one member: p_sample (diffusion_utils.py:54-92) when t >= 1 with the draw
 * z_dev [M,C] supplied, p_sample_t_1to0 (:96-111) when t == 0 (z_dev ignored, may be NULL).
 * y_dev [M,C] -> y_out_dev [M,C]; yhat_dev / ymean_dev [B,C]. */
int nd_p_sample(nd_handle h, int member, const float *y_dev, const float *yhat_dev, const float *ymean_dev,
                const float *z_dev, int t, float *y_out_dev, int B, int mc, void *stream);

/* p_sample_loop(..., only_last_sample=True) for members [member0, member0+n_members) and mc
 * Monte-Carlo trials at once (diffusion_utils.py:133-163 driven by the member x trial loop at
 * classification_train_separately.py:767-777).  M = B*mc rows per member, row m = trial*B + image.
 *   yhat_dev  [n_members, B, C]   eps_theta condition (y_0_hat)
 *   ymean_dev [n_members, B, C]   prior mean y_T_mean (same tensor in the reference, quirk Q2)
 *   noise_dev [n_members, T, M, C] the reference's RNG draws in draw order: index 0 = initial
 *              randn_like (:139), index i>=1 = the draw of p_sample at t = T-i (:67); NULL = draw them in the library
 *              (Philox, see nd_seed): throughput mode
 *   y0_out_dev [n_members, M, C]
 *   seq_out_dev optional [n_members, T+1, M, C]: y_T, y_{T-1}, ..., y_0 (only_last_sample=False)
 * The 3T+1 kernels are replayed from one hipGraph per (member range, B, mc, T, pointers) when
 * use_graph != 0.  nd_encode must have run for these members with the same B. */
int nd_sample(nd_handle h, int member0, int n_members, const float *yhat_dev, const float *ymean_dev,
              const float *noise_dev, float *y0_out_dev, float *seq_out_dev, int B, int mc, int T,
              int use_graph, void *stream);

/* Kernel-duration probes for the roofline report: when enabled, up to 8 PAIRS of consecutive denoising steps (i, i+1) of every
 * nd_sample / nd_predict_batch graph (or eager loop) get hipEvent record nodes on the launch stream: before the head of step i, after
 * it, after the two ConditionalLinear launches of step i, and after those of step i+1.  nd_profile_read (after the stream is
 * synchronised) returns mean intervals in microseconds over the last call's probed pairs: out_us[0] head(i) + o, [1] the lin2 and
 * lin3(+lin4) launches of step i TOGETHER + o, [2] the whole unrecorded step i+1 (head + both launches) + o, [3] o = what a record
 * node adds to a loaded interval = out_us[0] - (out_us[2] - out_us[1]) -- the overheads cancel in [2] - [1], which is the head alone.
 * Mean step-block launch = (out_us[1] - out_us[3]) / 2.  *n_samples = probed pairs = min(8, (T - 1) / 2).  out_us holds 4 floats. */
int nd_set_profiling(nd_handle h, int enable);
/* Weight bytes of one step launch (block 0: lin2 of all loaded members, 1: lin3) that are read with default-policy loads and so
 * stay resident in the 256 MiB Infinity Cache from step to step; the rest is streamed from HBM with nontemporal loads.  Lets the
 * roofline report separate bytes DELIVERED to the CUs from bytes that come out of DRAM.  -1 on a bad argument. */
long long nd_resident_weight_bytes(nd_handle h, int block);
int nd_profile_read(nd_handle h, float *out_us, int *n_samples);
/* Number of hipGraph event-record nodes in the most recently BUILT nd_sample / nd_predict_batch graph (4 per probed step when
 * profiling is on, 0 when it is off; -1 if the graph could not be walked): what a test checks to know that nd_profile_read's
 * intervals are stamped on every replay and are not left over from the eager first call. */
int nd_profile_probe_nodes(nd_handle h);

/* Refilling the input buffer early.  flag: a 32-bit word in host memory the DEVICE can write (pinned / mapped host memory), or NULL to
 * switch the signal off.  With a flag set, every nd_predict_batch call or graph replay stores 1, 2, 3, ... to it (the count of calls since
 * this function was called) as soon as the last kernel that reads images_dev has run -- the im2col of the conditioner and the pack of the
 * encoder hoist, about a quarter into the batch -- so a loader may write the NEXT batch into images_dev (classification_train_separately.py:
 * 722, the .to(device) of the batch loop) while this batch's sampler is still running.  Synchronises the device; drops the recorded batch
 * graphs. */
int nd_set_input_flag(nd_handle h, uint32_t *flag_host_visible);

/* FORM OF THE SAMPLING LOOP (diffusion_utils.py:145-157, the T iterations of p_sample_loop).  mode 0: 3T+1 kernel nodes of a
 * hipGraph (head, lin2 block, lin3 + lin4 block per step).  mode 1: ONE kernel launch for the whole loop where the shape allows
 * (csrc/nd_persist.hip: 17..32 rows per member, fp32 operands, <= 8 members whose weights exceed the Infinity Cache; any other call
 * keeps mode 0): each member's workgroups stay on their CUs for all T steps and meet at per-member barriers, members `skew_us`
 * microseconds apart, so one member's reductions run under the others' weight streams.  Same bits as mode 0.  The one-launch form
 * needs every workgroup of its grid resident at once, i.e. the device to itself while it runs (one process per GPU; processes that
 * share a device select mode 0).  A handle starts in the mode ND_PERSIST names (default: see nd_create); skew_us < 0 keeps the
 * current skew.  Drops the recorded graphs.
 *   nd_loop_form      1 if the most recently recorded or enqueued loop took the one-launch form, else 0
 *   nd_persist_status synchronises the device and returns 0 when no barrier wait of a one-launch loop has been abandoned since the
 *                     last reset (1 otherwise: results of that call are invalid -- every spin is bounded, a wait of more than one
 *                     second means the grid was not resident); reset != 0 clears the flag. */
int nd_set_loop_form(nd_handle h, int mode, float skew_us);
int nd_loop_form(nd_handle h);
int nd_persist_status(nd_handle h, int reset);

/* Copy of an internal per-member activation (tests / debugging), converted from the packed layout to
 * row-major: which = 0 xe [rows<=B, F], 1 h1 [rows<=M, F], 2 h2 [rows<=M, F] -> dst_dev [rows, F]. */
int nd_member_buffer(nd_handle h, int member, int which, float *dst_dev, int rows, void *stream);

/* ---- standalone operators (mapping network + unit tests) ---------------------------------- */
/* "frag16" packing of a K-contiguous matrix [R, K] (K % 16 == 0) into the MFMA fragment order the
 * weight-streaming kernels read with fully coalesced 1 KiB wave loads (layout: DESIGN.md).  Rows are
 * padded to a multiple of 16 with zeros.  Weights are packed once (mapping-MLP load time).
 * dtype ND_DTYPE_F16: the "frag32h" image (fp16, round to nearest even, K % 32 == 0) for the fp16-operand mode. */
size_t nd_packed_bytes(int R, int K, int dtype);
int nd_pack_rows(const float *src_dev, void *dst_packed_dev, int R, int K, int dtype, void *stream);

/* out[M,N] = act(scale[n] * (x[M,K] . W[N,K]^T) + shift[n]); scale/shift may be NULL (1 / 0).
 * nn.Linear + bias (+ReLU) of mapping/models/mlp.py:25-28 with shift = bias.  x and out are row-major fp32,
 * w_packed_dev is the nd_pack_rows image (same dtype) of the [N, K] nn.Linear weight.  Skinny-M weight streaming;
 * split-K across workgroups when K is large.  workspace_dev: >= nd_linear_workspace_bytes(M, K, N, dtype).
 * dtype ND_DTYPE_F16: x is rounded to fp16 on the way in, products exact, fp32 accumulation and epilogue. */
size_t nd_linear_workspace_bytes(int M, int K, int N, int dtype);
int nd_linear(const float *x_dev, const void *w_packed_dev, const float *scale_dev, const float *shift_dev,
              float *out_dev, int M, int K, int N, int act, int dtype, void *workspace_dev, size_t workspace_bytes,
              void *stream);

/* Launch plan the weight-streaming kernel would use for a Linear of this shape (host-side, no GPU work; tests / tuning):
 * out6 = {grid.x, grid.y, grid.z (k-slabs), fragment slots per workgroup, k-chunks per slab, threads per workgroup}.
 * mode 0 fused activation, 1 lin3+lin4 projection, 2 split-K partial sums. */
int nd_skinny_plan(int K, int N, int M, int n_members, int dtype, int mode, int *out6);
/* Row fragments (16 rows each) a weight-streaming workgroup keeps per pass at M rows -- the kernel's MT template argument (1, 2, 4, or
 * 5 where five save a whole pass over the weights: 65..80 rows, the reference's batch_size 70, configs/chest_x_ray.yml:66). < 0: error. */
int nd_skinny_row_fragments(int M);

/* Which kernel one ConditionalLinear block of the sampler (latent_model.py:178-184; K = N = feature_dim) runs at
 * M = B * mc_trials rows (mc_trials = 20 at classification_train_separately.py:770-771, batch_size 70 in
 * configs/chest_x_ray.yml:66 -> M = 1400), host only.  out8 = {0 weight-streaming k_skinny | 1 LDS-tiled k_cond_gemm
 * (M > 128 rows, fp32 operands), workgroups of the tiled launch, whole tiles, remainder tiles, k-split of the remainder,
 * partial sums per (row, class) handed to the step head, 128-row tiles over M, over N}. */
int nd_step_plan(int feature_dim, int M, int n_members, int dtype, int *out8);

/* Large-M GEMM for the ViT blocks: out[M,N] = act(x[M,K] . W[N,K]^T + bias[n]) (+ residual[M,N]).
 * timm 0.4.12 Attention.qkv / proj, Mlp.fc1 (GELU) / fc2, PatchEmbed.proj as GEMM
 * (call sites classification_train_separately.py:337-340).
 * workspace_dev (>= nd_gemm_workspace_bytes(M, K, N), 16-byte aligned; may be NULL/0): lets the last, partly filled round
 * of output tiles be cut along K across the idle CUs; without it every tile is computed whole (same sums, other order).
 * dtype ND_DTYPE_F32: w_dev is the fp32 [N,K] weight.  ND_DTYPE_F16 (fp16 mode; K % 32 == 0): w_dev is an fp16 [N,K] copy of it
 * (row-major, made once by the caller), x is rounded to fp16 on the fly, products exact, fp32 accumulation / epilogue / out. */
size_t nd_gemm_workspace_bytes(int M, int K, int N, int dtype);
int nd_gemm_bias_act(const float *x_dev, const void *w_dev, const float *bias_dev, const float *residual_dev,
                     float *out_dev, int M, int K, int N, int act, int dtype, void *workspace_dev, size_t workspace_bytes,
                     void *stream);

/* The same Linear layers (timm 0.4.12 Attention.qkv / proj, Mlp.fc1 / fc2, PatchEmbed.proj; call sites
 * classification_train_separately.py:337-346) on the bf16 matrix pipe WITH EXACT fp32 PRODUCTS: every fp32 operand value is held as
 * its three exact bf16 pieces a = a1 + a2 + a3 (8 + 8 + 8 significand bits), the nine piece products of a pair are exact in fp32 and
 * are accumulated in fp32 by v_mfma_f32_16x16x32_bf16 -- the reference's arithmetic (exact products, fp32 sums) in another summation
 * order, at 9/16 of the matrix-pipe cycles of the f32-input MFMA.  Operands are "frag32b3" images (csrc/nd_b9.hpp): made ONCE by
 * nd_split_rows (weights at load) or written directly by the producing operator (nd_layernorm_split, nd_gemm_split's out_split, ...).
 *   nd_split_bytes(rows, K)   bytes of the image of a [rows, K] matrix (rows padded to 16; K % 32 == 0), 0 on a bad shape
 *   nd_split_rows             x [rows, K] fp32 row-major -> image;   nd_join_rows: the inverse (exact), for tests
 *   nd_gemm_split             out[M,N] = act(x . w^T + bias) (+ residual): x_split_dev image of [M,K], w_split_dev image of [N,K];
 *                             out_dev fp32 [M,N] and / or out_split_dev = image of the result (N % 32 == 0) for the next layer; one of the
 *                             two may be NULL.  workspace as nd_gemm_bias_act (>= nd_gemm_split_workspace_bytes, may be NULL). */
size_t nd_split_bytes(int rows, int K);
int nd_split_rows(const float *x_dev, void *out_split_dev, int rows, int K, void *stream);
int nd_join_rows(const void *in_split_dev, float *x_dev, int rows, int K, void *stream);
size_t nd_gemm_split_workspace_bytes(int M, int K, int N);
int nd_gemm_split(const void *x_split_dev, const void *w_split_dev, const float *bias_dev, const float *residual_dev, float *out_dev,
                  void *out_split_dev, int M, int K, int N, int act, void *workspace_dev, size_t workspace_bytes, void *stream);

/* nn.LayerNorm(eps) over the last dim: x [rows, dim] -> out.  timm Block.norm1/norm2 (eps 1e-6).
 * nd_layernorm_split: the same values written as the frag32b3 image of [rows, dim] (dim % 32 == 0; nd_split_bytes(rows, dim) bytes):
 * the input of the nd_gemm_split that follows each LayerNorm of a ViT block. */
int nd_layernorm(const float *x_dev, const float *gamma_dev, const float *beta_dev, float *out_dev,
                 int rows, int dim, float eps, void *stream);
int nd_layernorm_split(const float *x_dev, const float *gamma_dev, const float *beta_dev, void *out_split_dev,
                       int rows, int dim, float eps, void *stream);

/* timm 0.4.12 Attention core: qkv [B, N, 3, heads, d] (the qkv Linear's output, unpermuted) ->
 * out [B, N, heads*d] = softmax(q k^T * d^-0.5) v, heads concatenated.  d must be 64.
 * dtype ND_DTYPE_F16 (fp16 mode): q, k, v and the normalised probabilities rounded to fp16, f16 MFMA, fp32 softmax/accumulate/out. */
int nd_attention(const float *qkv_dev, float *out_dev, int B, int N, int heads, int d, int dtype, void *stream);
/* The fp32 attention with its result written as the frag32b3 image of [B*N, heads*d] (the input of the proj nd_gemm_split). */
int nd_attention_split(const float *qkv_dev, void *out_split_dev, int B, int N, int heads, int d, void *stream);

/* The same attention with BOTH contractions on the bf16 matrix pipe, exact fp32 products (timm 0.4.12 Attention.forward; call sites
 * classification_train_separately.py:339-340).  Its operands are "qkv images": per (image, head) the q and k rows as frag32b3 blocks and
 * v transposed (layout: B9AttLayout, csrc/nd_b9.hpp), written directly by the qkv Linear's epilogue:
 *   nd_qkv_images_supported  1 where the form applies: N % 4 == 0, N <= 256, heads * 64 a multiple of 128
 *   nd_qkv_images_bytes      size of the image buffer for B images of N tokens
 *   nd_gemm_split_qkv        qkv Linear: x_split_dev image of [B*N, K], w_split_dev image of the [3*heads*64, K] weight, bias [3*heads*64]
 *   nd_attention_images      softmax(q k^T / 8) v from the images; out_dev fp32 [B*N, heads*64] or (out_is_split) its frag32b3 image */
int nd_qkv_images_supported(int N, int heads);
size_t nd_qkv_images_bytes(int B, int N, int heads);
int nd_gemm_split_qkv(const void *x_split_dev, const void *w_split_dev, const float *bias_dev, void *qkv_images_dev, int B, int N, int heads,
                      int K, void *stream);
int nd_attention_images(const void *qkv_images_dev, void *out_dev, int out_is_split, int B, int N, int heads, void *stream);

/* PatchEmbed im2col: img [B, Cin, Himg, Wimg] NCHW -> cols [B * (Himg/p) * (Wimg/p), Cin*p*p] so that
 * Conv2d(k=p, s=p) becomes nd_gemm_bias_act with the conv weight viewed [embed, Cin*p*p]. */
int nd_patchify(const float *img_dev, float *cols_dev, int B, int Cin, int Himg, int Wimg, int p, void *stream);
/* The same im2col written as the frag32b3 image of its [B * (Himg/p) * (Wimg/p), Cin*p*p] result (Cin*p*p % 32 == 0). */
int nd_patchify_split(const float *img_dev, void *cols_split_dev, int B, int Cin, int Himg, int Wimg, int p, void *stream);

/* softmax over the last dim of [rows, C] (classification_train_separately.py:755-758). */
int nd_softmax_rows(const float *x_dev, float *out_dev, int rows, int C, void *stream);

/* Aggregation: samples [S, B, C] (S = K*mc, member-major then trial) ->
 *   prob_out [B, C]  = mean_s softmax(-(y-1)^2 / temperature)   (convert_to_prob + compute_ensemble_confidence,
 *                      classification_train_separately.py:392-398, 425-447)
 *   vote_out [B] int64 = mode_s argmax_c y  (ties -> smallest label; majority_voting_for_mc_samples :51-68)
 *   probs_out optional [S, B, C]: per-sample probabilities (what the reference leaves in mc_samples, quirk Q4) */
int nd_aggregate(const float *samples_dev, float *prob_out_dev, int64_t *vote_out_dev, float *probs_out_dev,
                 int S, int B, int C, float temperature, void *stream);

/* ---- in-library noise (throughput mode) ------------------------------------------------------------------
 * The reference draws its Gaussian noise inside the loop with the global torch generator (diffusion_utils.py:67, 139:
 * randn_like); CPU torch.randn cannot be reproduced on a device, so parity runs pass the draws in (noise_dev).  With
 * noise_dev == NULL, nd_sample / nd_predict_batch draw them here: Philox4x32-10 (Salmon et al., SC'11), key = the 64-bit
 * seed, counter = (global image index, trial | member << 16 | class-quad << 24, draw index i (0 = y_T, i = the draw of
 * p_sample at t = T-i), batch counter); the four 32-bit outputs give four normals by Box-Muller (classes 4q..4q+3).
 * The draws of image g do not depend on how a batch is sharded over ranks (first_image = the global index of this rank's
 * first image), and the batch counter -- kept on the device, advanced by the sampler itself -- makes successive batches
 * differ.  nd_seed resets it to 0.  Synchronises the device (rare: once per run). */
int nd_seed(nd_handle h, uint64_t seed, uint32_t first_image);
/* The same generator as a standalone operator (tests, known-answer checks): out_dev [n_members, T, mc*B, C], row = trial*B + b. */
int nd_philox_normal(float *out_dev, int n_members, int T, int B, int mc, int C, uint64_t seed, uint32_t batch_counter,
                     uint32_t first_image, void *stream);
/* Raw Philox4x32-10 blocks (known-answer test): out_dev[4*i..4*i+3] = philox(counter = ctr_dev[4*i..], key). */
int nd_philox_raw(const uint32_t *ctr_dev, uint32_t *out_dev, int n, uint32_t key0, uint32_t key1, void *stream);

/* ---- conditioner: the mapping network (classification_train_separately.py:249-275, 330-348) -------------------
 * cond_pred_model = {'vit': timm 0.4.12 vit_base_patch16_224, 'mlps': [mapping/models/mlp.py::Classifier] * K}.
 * The handle holds POINTERS to the caller's device tensors (the caller keeps them alive, as a torch module does) and one
 * caller-provided workspace for the activations. */
typedef struct nd_cond_s *nd_cond;
typedef struct {
    int32_t img_size, patch, in_chans;   /* 224, 16, 3 */
    int32_t embed_dim, num_heads;        /* 768, 12 (head dim must be 64) */
    int32_t mlp_hidden;                  /* 3072 (timm Mlp hidden = 4 * embed) */
    int32_t n_blocks;                    /* ViT blocks available to nd_vit_block (>= n_mlps) */
    int32_t n_mlps;                      /* K mapping MLPs: MLP i reads the tokens after blocks[0..i] (:336-345) */
    int32_t mlp_widths[3];               /* 4096, 2048, 128 (mapping/models/mlp.py:12-18) */
    int32_t num_classes;                 /* C */
    int32_t max_batch;
    int32_t max_tokens;                  /* tokens per image nd_vit_block may be called with (196 on the mapping path, 197 with
                                          * the cls token of the full forward) */
    int32_t operand_dtype;               /* ND_DTYPE_F32 | ND_DTYPE_F16 (Linear weights fp16 row-major, MLP weights frag32h) */
    float ln_eps;                        /* 1e-6: timm's norm_layer = partial(nn.LayerNorm, eps=1e-6) */
} nd_cond_config;
/* timm PatchEmbed.proj viewed [embed, in_chans*patch*patch] (+ bias); fp32, or fp16 in the fp16 mode. */
typedef struct { const void *proj_w; const float *proj_b; } nd_patch_embed_weights;
/* One timm Block: norm1, attn.qkv [3E,E], attn.proj [E,E], norm2, mlp.fc1 [4E,E], mlp.fc2 [E,4E]; Linear weights row-major
 * [out,in] (nn.Linear layout) in fp32 (ND_DTYPE_F32) or fp16 (ND_DTYPE_F16), or frag32b3 images of them (ND_DTYPE_F32_SPLIT: nd_split_rows;
 * the same holds for nd_patch_embed_weights.proj_w viewed [embed, in_chans*patch^2]); everything else fp32. */
typedef struct {
    const float *norm1_w, *norm1_b;
    const void *qkv_w;  const float *qkv_b;
    const void *proj_w; const float *proj_b;
    const float *norm2_w, *norm2_b;
    const void *fc1_w;  const float *fc1_b;
    const void *fc2_w;  const float *fc2_b;
} nd_vit_block_weights;
/* One mapping MLP: linear1..4 weights as nd_pack_rows images (same dtype as the config) + fp32 biases. */
typedef struct { const void *w_packed[4]; const float *bias[4]; } nd_mlp_weights;

size_t nd_cond_workspace_bytes(const nd_cond_config *cfg);
int nd_cond_create(const nd_cond_config *cfg, nd_cond *out);
int nd_cond_destroy(nd_cond c);
const nd_cond_config *nd_cond_get_config(nd_cond c);
int nd_cond_bind_workspace(nd_cond c, void *workspace_dev, size_t bytes);
int nd_cond_set_patch_embed(nd_cond c, const nd_patch_embed_weights *w);
int nd_cond_set_block(nd_cond c, int block, const nd_vit_block_weights *w);
int nd_cond_set_mlp(nd_cond c, int i, const nd_mlp_weights *w);

/* timm 0.4.12 Block.forward on tokens [B, N, embed] (pre-LN: x += attn(norm1(x)); x += mlp(norm2(x)); exact-erf GELU):
 * the calls `vit.blocks[j](tmp)` at classification_train_separately.py:339-340.  tok_out_dev may alias tok_in_dev. */
int nd_vit_block(nd_cond c, int block, const float *tok_in_dev, float *tok_out_dev, int B, int N, void *stream);

/* Diffusion.compute_guiding_prediction (:330-345) for the K mapping members: patch_embed (no cls token, no pos_embed; pos_drop
 * is the identity in eval), blocks[0..i] with the prefix SHARED between members (15 -> K block evaluations; eval-mode blocks
 * are deterministic, same values), mlps[i] right after block i.  images_dev [B, in_chans, img, img] ->
 * logits_out_dev [K, B, C]; yhat_out_dev (optional) [K, B, C] = softmax(logits, dim=1) (:755-758).
 * The never-sampled (K+1)-th element, vit(x) (:346, quirk Q1), is not computed here.  No allocation, no synchronisation:
 * capturable into a hipGraph. */
int nd_guiding_prediction(nd_cond c, const float *images_dev, float *logits_out_dev, float *yhat_out_dev, int B, void *stream);

/* ---- the whole hot path of one test batch as ONE launch (classification_train_separately.py:749-794) ----------
 * guiding prediction -> softmax -> encoder hoist of every member -> K x mc p_sample_loops -> convert_to_prob / mean / vote,
 * replayed from one hipGraph per (pointers, B, mc, T) when use_graph != 0.  Device pointers, caller-owned:
 *   images_dev [B, in_chans, img, img] (flattened [B, D] for the noise estimators, :747)
 *   noise_dev  [K, T, mc*B, C] in the reference's draw order, or NULL for in-library Philox noise (nd_seed)
 *   out->samples [K*mc, B, C] raw y_0, member-major then trial (the order of mc_samples, :767-784)
 *   out->prob [B, C], out->vote [B] int64, out->probs [K*mc, B, C] (optional), out->yhat [K, B, C]
 * K = the ensemble handle's n_members, all loaded, K <= the conditioner's n_mlps: member k is conditioned on mapping MLP k, and only
 * the first K mapping MLPs (and the prefix blocks they need) are evaluated -- the reference samples selected_block_indices ∩ the
 * available diffusion checkpoints (:275, :769), e.g. 3 noise estimators beside 5 mapping MLPs. */
typedef struct { float *samples; float *prob; int64_t *vote; float *probs; float *yhat; } nd_batch_out;
int nd_predict_batch(nd_handle h, nd_cond c, const float *images_dev, const float *noise_dev, const nd_batch_out *out,
                     int B, int mc, int T, float temperature, int use_graph, void *stream);

/* ---- reporting tail of test_atk (classification_train_separately.py:801-838) --------------------- */
/* Per-image spread of the S = K*mc per-sample probabilities (what the reference keeps in pred_mc, quirk Q4):
 *   piw_out[B,C] = quantile(q_hi) - quantile(q_lo) over the samples (torch.quantile semantics: linear
 *                  interpolation at rank q*(S-1));  compute_mean_piws_for_class :108-114 uses 0.025 / 0.975
 *   var_out[B,C] = unbiased variance over the samples;  calculate_variances :166-172
 * probs_dev [S,B,C]. */
int nd_sample_stats(const float *probs_dev, float *piw_out_dev, float *var_out_dev, int S, int B, int C,
                    float q_lo, float q_hi, void *stream);

/* The numbers test_atk prints, over the whole test set (N images):
 *   out[0]        majority-vote accuracy                                   (compute_accuracy :801-807)
 *   out[1]        ECE: n_bins-bin l1 calibration error (torchmetrics 0.11.4 MulticlassCalibrationError) of
 *                 convert_to_prob(prob_mean) -- the reference passes the averaged probabilities with prob_in=False,
 *                 so they are transformed once more (:413-423, :812); kept as written
 *   out[2+c], out[2+C+c]       mean PIW of class c over correct / incorrect votes (NaN if none; :124-140)
 *   out[2+2C+c], out[2+3C+c]   mean variance of class c over correct / incorrect votes (0 if none; :150-172)
 * piw_dev, var_dev, prob_mean_dev [N,C]; vote_dev, target_dev [N] int64; out_dev [2+4C]. */
int nd_report(const float *piw_dev, const float *var_dev, const float *prob_mean_dev, const int64_t *vote_dev,
              const int64_t *target_dev, float *out_dev, int N, int C, float temperature, int n_bins, void *stream);

/* ---- input perturbations of the robustness protocol (diffusion/utils.py:272-414; applied at
 * classification_train_separately.py:726-737).  Images are [B, C, H, W] fp32, contiguous. ------------------- */
/* add_noise (:272-279): out = x + z * std, z = the randn_like draw (supplied, like the sampler's noise). */
int nd_img_add_noise(const float *x_dev, const float *z_dev, float *out_dev, size_t n, float std, void *stream);
/* adjust_brightness (:390-399): out = clamp(x + k, 0, 1). */
int nd_img_brightness(const float *x_dev, float *out_dev, size_t n, float k, void *stream);
/* adjust_contrast (:402-414): m = per-image mean; out = clamp(m + (x - m) * k, 0, 1).  mean_ws_dev: B floats. */
int nd_img_contrast(const float *x_dev, float *out_dev, float *mean_ws_dev, int B, size_t per_image, float k, void *stream);
/* torch interpolate(mode='bilinear', align_corners=False) [B,C,Hi,Wi] -> [B,C,Ho,Wo]: down_up_sample (:372-387) is two
 * calls.  With crop_dev != NULL ([B][2] int32 = top, left) the source is the crop_size x crop_size window of each
 * image: random_crop_and_resize (:282-312; torchvision Resize on tensors = the same interpolate). */
int nd_img_resize_bilinear(const float *x_dev, float *out_dev, int B, int C, int Hi, int Wi, int Ho, int Wo,
                           const int32_t *crop_dev, int crop_size, void *stream);
/* random_cover_new (:315-349): zero n_rects squares of side `side` per image, in place; rects_dev [B][n_rects][2]
 * int32 = (top, left), chosen on the host with the reference's rejection sampling. */
int nd_img_cover(float *x_dev, int B, int C, int H, int W, const int32_t *rects_dev, int n_rects, int side, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* NESTED_DIFFUSION_H */
