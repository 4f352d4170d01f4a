/*
 * dmvae_hip.h -- C ABI of the MI355X (gfx950) DMVAE training-step library.
 *
 * This is the drop-in boundary for the ONE hot path of ffs97/deep-mixture-vae:
 * everything that the reference executes inside
 *     session.run([self.loss, self.train_step], feed_dict)      code/base_models.py:126-129
 * (forward MLPs, reparameterisation, mixture KL, Bernoulli reconstruction
 * loss, autodiff backward, Adam) plus the host batch assembly that feeds it.
 * The reference has no FFI of its own (pure Python over TensorFlow 1.x), so
 * each entry point names the TensorFlow op / reference line it replaces.
 *
 * Conventions
 *   - plain C: device pointers + sizes, no C++ / torch types.
 *   - `stream` is a hipStream_t passed as void*; every call only ENQUEUES work
 *     on it (safe under stream capture); nothing synchronises or allocates.
 *   - every function returns 0 on success, a negative DMVAE_E* code for an
 *     argument error, or a positive hipError_t; dmvae_last_error() gives text.
 *   - matrices are row-major with an explicit leading dimension (elements).
 *   - "act" buffers are bf16 when dtype == DMVAE_BF16, float when DMVAE_F32.
 *   - not re-entrant on the same stream/plan; distinct streams are independent.
 */
#ifndef DMVAE_HIP_H
#define DMVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: dmvae_buffers.arena_elems, dmvae_config.model, dmvae_latent_args.mfma_ws(_bytes), dmvae_config.adam_ieee and
 *    dmvae_prof_row.kernel_ms were added after version 1; a client compiled against an older header passes shorter
 *    structs, so every binding checks dmvae_abi_version() == DMVAE_ABI_VERSION when it loads the library. */
/* 3: dmvae_plan_set_stage_groups added (no struct changed). */
/* 4: dmvae_plan_prefetch_batch / dmvae_plan_swap_batch added (a second bf16 batch buffer in the plan's workspace:
 *    dmvae_sizes.work_bytes grows; no struct layout changed). */
/* 5: dmvae_heads_latent_fwd / dmvae_heads_latent_ok / dmvae_heads_args added; DMVAE_ADAM_SHADOW flag of dmvae_adam_tf
 *    (no existing struct changed). */
#define DMVAE_ABI_VERSION 5

enum { DMVAE_F32 = 0, DMVAE_BF16 = 1 };

enum {
    DMVAE_EINVAL = -1,   /* bad argument (shape not tile aligned, null pointer ...) */
    DMVAE_EUNSUPPORTED = -2,
    DMVAE_ESTATE = -3
};

/* ---- operand layouts of dmvae_gemm -------------------------------------
 * C[M,N] = sum_k A(m,k) * B(k,n)
 *   DMVAE_GEMM_FWD : A = act [M][lda] (k contiguous),  B = W [K][ldb] (n contiguous)
 *                    tf.layers.dense / tf.matmul forward   base_models.py:221-248,291-293, includes/layers.py:32
 *   DMVAE_GEMM_DX  : A = dY  [M][lda] (k contiguous),  B = W [N][ldb] (k contiguous)
 *                    dX = dY . W^T  (tf.gradients of the above, base_models.py:110)
 *   DMVAE_GEMM_DW  : A = X   [K][lda] (m contiguous),  B = dY [K][ldb] (n contiguous)
 *                    dW = X^T . dY  (tf.gradients of the above)
 * M, N must be multiples of 64 and K of 64 (callers pad; see dmvae_plan).
 */
enum { DMVAE_GEMM_FWD = 0, DMVAE_GEMM_DX = 1, DMVAE_GEMM_DW = 2 };

/* ---- fused epilogues ---------------------------------------------------- */
enum {
    DMVAE_EPI_BIAS_RELU = 0, /* out(act) = relu(acc + bias[n])                    activation=relu dense      */
    DMVAE_EPI_BIAS_F32 = 1,  /* out(f32) = acc + bias[n]                          linear heads               */
    DMVAE_EPI_BIAS_RECON = 2,/* l = acc + bias; loss partial + out(act) = dLoss/dl  base_models.py:72-85       */
    DMVAE_EPI_RELU_MASK = 3, /* out(act) = acc * (aux0(act)[m][n] > 0)            ReLU backward              */
    DMVAE_EPI_LATENT = 4,    /* dZ = acc: out[m][n] = dZ + aux0, out[m][d_off+n] = dZ*aux2 + aux1   priors.py:86-89 backward */
    DMVAE_EPI_STORE_F32 = 5, /* out(f32) = acc            (dW, split_k == 1)                                    */
    DMVAE_EPI_ATOMIC_F32 = 6,/* out(f32) += acc via atomics (dW, split_k > 1; out pre-zeroed)                   */
    DMVAE_EPI_BIAS_SIGMOID = 7,/* out(f32) = sigmoid(acc + bias)  reconstructed_X  base_models.py:295-296      */
    DMVAE_EPI_ADAM = 8       /* dW only, grouped bf16 launch with a dmvae_adam_ctx: the gradient tile never
                              * leaves the registers -- the TF-Adam update of the matching parameter / m / v
                              * elements (same offset in every arena) is applied in the epilogue; out = the
                              * tile's address in the GRADIENT arena (locates the offset; written only when
                              * ctx->store_grad), out2 = the bias gradient's address there (updated likewise) */
};

typedef struct dmvae_epilogue {
    int32_t kind;        /* DMVAE_EPI_*                                              */
    int32_t m_valid;     /* rows  < m_valid are real (RECON masks the rest), 0 = all */
    int32_t n_valid;     /* cols  < n_valid are real (RECON masks the rest), 0 = all */
    int32_t d_off;       /* LATENT: column offset of the log_var half                */
    int32_t recon_kind;  /* RECON: 0 = binary (sigmoid xent), 1 = real (0.5*sq err)  */
    float scale;         /* RECON: 1/B                                               */
    void* out;           /* primary output                                           */
    int64_t ldo;
    void* out2;          /* RECON: optional f32 copy of the logits (may be NULL).
                          * DW layout: optional f32 bias gradient db[N] = sum_k B[k][n]
                          * (ones-operand MFMA in the first tile row; may be NULL)    */
    int64_t ldo2;
    const float* bias;   /* [N] or NULL                                              */
    const void* aux0;    /* RELU_MASK: forward activation (act); RECON: x (f32); LATENT: gmu (f32) */
    int64_t ld0;
    const void* aux1;    /* LATENT: glv (f32) */
    int64_t ld1;
    const void* aux2;    /* LATENT: clv = eps*0.5*exp(lv/2) (f32) */
    int64_t ld2;
    float* partials;     /* RECON: per-workgroup loss partial sums, length >= dmvae_gemm_partials() */
} dmvae_epilogue;

/* GEMM with fused epilogue.  dtype selects bf16 MFMA (v_mfma_f32_16x16x32_bf16,
 * fp32 accumulate) or exact-f32 MFMA (v_mfma_f32_16x16x4_f32).  split_k > 1 is
 * valid only with DMVAE_EPI_ATOMIC_F32. */
int dmvae_gemm(void* stream, int dtype, int layout, int M, int N, int K,
               const void* A, int64_t lda, const void* B, int64_t ldb,
               const dmvae_epilogue* epi, int split_k);
/* number of loss partials dmvae_gemm writes for an (M,N) RECON launch: one per 64x64 cell of the output,
 * whatever tile the kernel uses (a larger tile writes its sum to its first cell and zeros to the others) */
int dmvae_gemm_partials(int dtype, int M, int N);

/* Grouped weight-gradient GEMMs: n (<= 16) independent DMVAE_GEMM_DW problems
 * (dW_l = X_l^T . dY_l, optional fused db_l through epi.out2; epi.kind must be
 * DMVAE_EPI_STORE_F32) enqueued as ONE grid for bf16 -- every tf.gradients(loss, W_l)
 * of base_models.py:110 in one launch.  Each problem alone covers a fraction of the
 * 256 CUs; together they keep the chip full without split-K (bit-reproducible). */
typedef struct dmvae_gemm_problem {
    int32_t M, N, K, reserved;
    const void* A; int64_t lda;
    const void* B; int64_t ldb;
    dmvae_epilogue epi;
} dmvae_gemm_problem;
int dmvae_gemm_grouped_dw(void* stream, int dtype, const dmvae_gemm_problem* probs, int n);
/* the general form: n independent problems of one layout sharing an epilogue kind.  bf16 groups
 * exist for (DW, STORE_F32), (FWD, BIAS_F32) and (DX, RELU_MASK); f32 issues one launch each. */
int dmvae_gemm_grouped(void* stream, int dtype, int layout, const dmvae_gemm_problem* probs, int n);

/* dW GEMMs with the optimizer step fused into their epilogue (DMVAE_EPI_ADAM; bf16): replaces the
 * tf.gradients + AdamOptimizer.apply_gradients pair of base_models.py:95-110 for every weight and
 * bias in one launch.  The four arenas (param, grad, m, v) share one layout; t = state->adam_t
 * (already advanced for this step).  [seg_off, seg_off + seg_n) is an extra element range of the
 * arenas (multiple of 4 elements) whose gradient is already in `grad` -- the prior tables -- updated
 * by one more workgroup of the same launch. */
typedef struct dmvae_adam_ctx {
    float* param; float* grad; float* m; float* v;
    void* param_bf16;          /* bf16 shadow refreshed with the new parameters (may be NULL) */
    const void* state;         /* dmvae_state: lr and adam_t                                  */
    float beta1, beta2, epsilon, grad_scale;
    int32_t store_grad;        /* also write the gradients to `grad`                          */
    int32_t ieee;              /* bf16 mode: 1 = IEEE square root and division in the update quotient (as fp32 mode),
                                * 0 = the hardware v_sqrt_f32 / v_rcp_f32 (1 ulp each; dmvae_adam_tf below)      */
    int64_t seg_off, seg_n;
} dmvae_adam_ctx;
int dmvae_gemm_grouped_dw_adam(void* stream, const dmvae_gemm_problem* probs, int n, const dmvae_adam_ctx* ctx);

/* ---- latent kernel: softmax + reparameterisation + mixture KL + all KL gradients
 * replaces priors.py:86-89 (Z), :104-147 (KL_Z exact / relaxed), :170-181
 * (Gumbel-Softmax), :183-201 (KL_C), base_models.py:249 (softmax) and their
 * tf.gradients.  mode 0 = exact (cluster_sample False, live), 1 = relaxed, 2 = VaDE (below). */
typedef struct dmvae_latent_args {
    int32_t B;            /* rows that are real; rows [B, B_pad) are written as zeros */
    int32_t B_pad;
    int32_t D, K;
    int32_t mode;         /* 0 exact, 1 relaxed (Gumbel-Softmax weights)              */
    int32_t act_dtype;    /* dtype of Z_act / dlogits_act                              */
    float kl_ratio;       /* used when state == NULL                                  */
    float temperature;
    float inv_B;          /* 1 / (batch size the loss averages over)                  */
    uint64_t seed;        /* Philox key when eps / gumbel are NULL                    */
    uint64_t noise_step;  /* Philox stream position when state == NULL                */
    const float* mean; int64_t ld_mean;       /* [B_pad][>=D] */
    const float* log_var; int64_t ld_log_var; /* [B_pad][>=D] */
    const float* logits; int64_t ld_logits;   /* [B_pad][>=K] */
    const float* eps; int64_t ld_eps;         /* [B][D] or NULL -> on-device Philox N(0,1)   */
    const float* gumbel; int64_t ld_gumbel;   /* [B][K] or NULL -> on-device Philox Gumbel    */
    const float* prior_means;                 /* [K][D] */
    const float* prior_log_vars;              /* [K][D] */
    void* Z_act; int64_t ld_Z;                /* [B_pad][ld_Z] act dtype; cols >= D zeroed    */
    float* Z_f32; int64_t ld_Zf;              /* optional f32 copy or NULL                    */
    float* weights; int64_t ld_w;             /* optional [B_pad][K]: softmax(logits) / zeta  */
    float* gmu; float* glv; float* clv; int64_t ld_g;  /* [B_pad][ld_g] f32: KL grads wrt mean/log_var, reparam coef */
    void* dlogits_act; int64_t ld_dl;         /* [B_pad][ld_dl] act dtype; cols >= K zeroed   */
    float* dprior_partials;                   /* [nblocks][2][K][D] f32 (deterministic two-pass) */
    float* loss_partials;                     /* [nblocks][2]: sum_b KL_Z_b, sum_b KL_C_b      */
    const void* state;                        /* optional dmvae_state* (device): kl_ratio, noise_step read from it */
    /* Large prior tables (mode 0, K * D >= 4096: dmvae_latent_ws_bytes() > 0): scratch for the MFMA form of the
     * (row, cluster, dimension) contractions -- three exact-f32 MFMA GEMMs between two row-wise kernels
     * (csrc/latent_mfma.hip).  NULL (or too small a size) = the one-kernel form.  With the scratch, the loss partials
     * are written for the same dmvae_latent_nblocks() blocks, but the prior-table gradient arrives COMPLETE in row 0
     * of dprior_partials (rows 1.. are not touched), and on-device noise comes from a differently keyed Philox stream. */
    void* mfma_ws; int64_t mfma_ws_bytes;
} dmvae_latent_args;
int dmvae_latent_nblocks(int B_pad, int D, int K);
/* mode 2 (VaDE, csrc/latent_vade.hip): weights = get_cluster_probs(Z) of the SAMPLE (priors.py:91-102) for both KL terms,
 * gradients through them included; logits / gumbel / dlogits_act are ignored (may be NULL); `weights` receives the
 * responsibilities.  It writes dmvae_latent_nblocks_vade(B_pad) = B_pad / 16 blocks of partials. */
int dmvae_latent_nblocks_vade(int B_pad);
int64_t dmvae_latent_ws_bytes(int B_pad, int D, int K, int mode);   /* 0 when the MFMA form does not apply */
int dmvae_latent_fwd(void* stream, const dmvae_latent_args* a);

/* ---- the two head layers' forward pass AND the latent stage as one launch (csrc/heads_latent.hip) -----------------
 * [mean | log_var] = hz . W_mv + b_mv and logits = hc . W_lg + b_lg (the three linear tf.layers.dense of base_models.py:229-249:
 * 231-239 mean / log_var off the z-head's hidden layer, 241-248 logits off the c-head's) followed, on the same rows in the same
 * workgroup, by everything dmvae_latent_fwd computes (priors.py:86-201).  `a` as for dmvae_latent_fwd, except that a->mean,
 * a->log_var (= a->mean + Dp, one [B_pad][2 Dp] buffer) and a->logits ([B_pad][Kp]) are OUTPUTS here: the heads' f32 results are
 * written there as dmvae_gemm with DMVAE_EPI_BIAS_F32 would write them -- same bits -- and every other output is bit-identical to
 * dmvae_gemm_grouped + dmvae_latent_fwd on the same inputs.  bf16 operands only; Dp = 64 | 128, Kp = 64, K * D < 4096 (the one-kernel
 * latent form), Hp a multiple of 64, B_pad a multiple of 16 and at most 4096 (one round of 256 workgroups: every workgroup streams
 * both weight matrices); anything else: DMVAE_EUNSUPPORTED (dmvae_heads_latent_ok() == 0) -- use the two calls. */
typedef struct dmvae_heads_args {
    const void* hz; int64_t lda;     /* bf16 [B_pad][lda]: columns [0, Hp) = relu hidden layer of the z-head, [Hp, 2 Hp) = of the c-head */
    int32_t Hp, Dp, Kp, reserved;    /* padded widths: head hidden layer, latent_dim, n_classes                                       */
    const void* W_mv; int64_t ld_mv; /* bf16 [Hp][ld_mv]: columns [0, Dp) mean's kernel, [Dp, 2 Dp) log_var's                         */
    const void* W_lg; int64_t ld_lg; /* bf16 [Hp][ld_lg]: the logits kernel                                                           */
    const float* b_mv;               /* f32 [2 Dp] */
    const float* b_lg;               /* f32 [Kp]   */
    /* K slices (optional; small batches: few 16-row blocks, each a long K chain): kslices (2 .. 8, Hp % (kslices * 64) == 0) workgroups per block,
     * whose partial tiles the last one to arrive adds in ascending order before the latent stage -- deterministic; the results then differ from the
     * unsliced call by the f32 summation order.  kslice_ws: >= dmvae_heads_latent_kslice_floats() floats; kslice_tick: B_pad / 16 ints, zero on
     * entry (left zero).  kslices <= 1: none. */
    int32_t kslices, reserved2;
    float* kslice_ws; int64_t kslice_ws_floats;
    int32_t* kslice_tick;
} dmvae_heads_args;
int64_t dmvae_heads_latent_kslice_floats(int B_pad, int Dp, int kslices);
int dmvae_heads_latent_ok(int B_pad, int D, int K, int Dp, int Kp, int Hp, int mode);
int dmvae_heads_latent_fwd(void* stream, const dmvae_heads_args* heads, const dmvae_latent_args* a);

/* ---- stand-alone reconstruction loss (fused form: DMVAE_EPI_BIAS_RECON) --
 * tf.nn.sigmoid_cross_entropy_with_logits + reduce_sum/mean, base_models.py:72-85 */
int dmvae_recon_fwd_bwd(void* stream, int act_dtype, int recon_kind, int B, int B_pad, int I, int I_pad,
                        const float* logits, int64_t ldl, const float* x, int64_t ldx, float inv_B,
                        void* dlogits_act, int64_t ldd, float* partials /* [dmvae_recon_nblocks] */);
int dmvae_recon_nblocks(int B_pad, int I_pad);

/* ---- column sums (bias gradients, reduction of per-block partials) ------
 * out[n] = sum_m in[m][n]; tf.gradients of the bias add. */
int dmvae_colsum(void* stream, int in_dtype, const void* in, int64_t ld, int M, int N, float* out);

/* ---- device-resident step state (read by kernels so a captured graph can be replayed) */
typedef struct dmvae_state {
    uint64_t adam_t;        /* completed Adam updates (t of the next update = adam_t + 1) */
    uint64_t noise_step;    /* Philox stream position                                  */
    uint32_t batch_cursor;  /* next batch within the epoch                             */
    uint32_t batches_per_epoch;
    float kl_ratio;
    float lr;
    float epoch_weight;     /* 1/epoch_len: loss += batch_loss * epoch_weight  (base_models.py:130) */
    float lr_t;             /* lr*sqrt(1-b2^t)/(1-b1^t) for t = adam_t, written when a plan step advances adam_t
                             * (read by the fused dW + Adam epilogue; the stand-alone Adam kernel recomputes it) */
    float epoch_loss, epoch_recon, epoch_klz, epoch_klc;   /* running epoch means   */
    float last_loss, last_recon, last_klz, last_klc;       /* last batch            */
} dmvae_state;

/* loss = recon + kl_ratio*(KL_C + KL_Z)  (base_models.py:87-93); sums the
 * per-block partials deterministically, updates state (epoch accumulators,
 * batch_cursor, noise_step). */
int dmvae_loss_finalize(void* stream, const float* recon_partials, int n_recon,
                        const float* latent_partials, int n_latent, float inv_B, void* state);

/* ---- TF-1.x Adam on a flat arena (tf.train.AdamOptimizer, base_models.py:95-110)
 * lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps).  t = state->adam_t+1
 * (or t_host when state is NULL).  grad is multiplied by grad_scale first (1/world).
 * Optionally refreshes a bf16 shadow of the parameters and zeroes grad (flags & DMVAE_ADAM_ZERO_GRAD).
 * Arithmetic follows the mode: param_bf16 == NULL (fp32 parity mode) -> IEEE sqrt and division; param_bf16 != NULL (bf16
 * throughput mode, where the forward pass reads the 8-bit-mantissa shadow) -> the quotient lr_t*m / (sqrt(v)+eps) with
 * the hardware square root and reciprocal (1 ulp each), exactly as the fused dW + Adam epilogues of that mode compute it
 * (DMVAE_EPI_ADAM: bit-identical to this kernel on the same inputs); flags & DMVAE_ADAM_IEEE keeps IEEE there too.
 * dmvae_adam_finish bumps state->adam_t (separate 1-thread kernel so that all
 * chunks of one update see the same t). */
#define DMVAE_ADAM_ZERO_GRAD 1
#define DMVAE_ADAM_IEEE 2
/* the same update (same bits) on the low-footprint kernel that can share a CU with a macro-tile GEMM running on another stream
 * (<= 48 VGPRs, a 32 KiB LDS ring filled by LDS-DMA, four waves per workgroup); bits 8..23 of flags: workgroups (0 = 256) */
#define DMVAE_ADAM_SHADOW 4
int dmvae_adam_tf(void* stream, int64_t n, float* param, float* grad, float* m, float* v,
                  void* param_bf16, float lr, float beta1, float beta2, float epsilon,
                  float grad_scale, int flags, uint64_t t_host, const void* state);
int dmvae_adam_finish(void* stream, void* state);

/* ---- batch assembly (Dataset.get_batches, includes/utils.py:449-463) -----
 * row r of the batch = data[perm[first + r]] (perm NULL: data[first + r]);
 * first = cursor*batch (cursor from state when state != NULL).  Writes the act
 * copy [B_pad][ld_act] and an f32 copy (either may be NULL); pad rows/cols = 0. */
int dmvae_gather_rows(void* stream, int act_dtype, const float* data, int64_t n_rows, int dim,
                      const int32_t* perm, int64_t first, int batch, int n_valid, int B_pad,
                      void* out_act, int64_t ld_act, float* out_f32, int64_t ld_f32,
                      const void* state);

/* ---- noise (np.random.randn / sample_gumbel, priors.py:67-68,157-158) ---- */
int dmvae_philox_normal(void* stream, float* out, int64_t n, uint64_t seed, uint64_t step, uint32_t stream_id);
int dmvae_philox_gumbel(void* stream, float* out, int64_t n, uint64_t seed, uint64_t step, uint32_t stream_id);

/* ---- casts ---- */
int dmvae_cast_f32_to_bf16(void* stream, const float* in, void* out, int64_t n);
int dmvae_cast_bf16_to_f32(void* stream, const void* in, float* out, int64_t n);

/* ======================================================================
 * Step plan: the whole session.run([loss, train_step]) as one enqueue.
 * ====================================================================== */
#define DMVAE_MAX_LAYERS 8
/* Encoder trunk.  MLP: tf.layers.dense x n_enc on the 784 inputs (base_models.py:218-226, the branch
 * BASELINE.json names).  CNN: the checked-in `self.cnn = True` branch (base_models.py:156,176-216): the
 * batch as 28x28x1 images, six 3x3 SAME convolutions (32,32,64,64,128,128) with 2x2 SAME max-pools behind
 * the 2nd / 4th / 6th, flattened (h,w,c) to 2048, then ONE FullyConnected layer enc[0] (n_enc must be 1,
 * input_dim 784).  The conv layers run as the step's GEMMs in conv mode -- the patch matrix is implicit,
 * activations are stored with a zero border (csrc/conv.hip).  Their tensors come first in the arena:
 * "W_conv<i>" [9*Cin][Cout] = the HWIO kernel flattened (ld = Cout padded to 64) and "b_conv<i>" [Cout]. */
#define DMVAE_TRUNK_MLP 0
#define DMVAE_TRUNK_CNN 1
/* Model.  DMVAE: base_models.py:150-302 (trunk, z-head and c-head with a hidden layer each, q(c|x) = softmax(logits)).
 * VaDE: base_models.py:435-562 -- n_enc FullyConnected trunk layers, mean / log_var linear straight off the trunk (no
 * head hidden layers, no logits: head_dim is ignored), q(c|x) := p(c|z) = get_cluster_probs(Z) (priors.py:91-102) as the
 * mixture weights AND the categorical probabilities; the latent stage is dmvae_latent_fwd mode 2.  Tensor table: W_enc<i>,
 * b_enc<i>, W_mean, b_mean, W_logvar, b_logvar, W_dec<i>, b_dec<i>, W_out, b_out, prior_means, prior_log_vars.
 * With DMVAE_TRUNK_CNN (base_models.py:456-488): the same conv / pool stack in front, enc[0] = 128 (fc 2048 -> 128). */
#define DMVAE_MODEL_DMVAE 0
#define DMVAE_MODEL_VADE 1

typedef struct dmvae_config {
    int32_t input_dim, latent_dim, n_classes;
    int32_t n_enc; int32_t enc[DMVAE_MAX_LAYERS];   /* trunk widths (reference: 500,500)        */
    int32_t head_dim;                                /* z-/c-head hidden width (reference: 2000) */
    int32_t n_dec; int32_t dec[DMVAE_MAX_LAYERS];   /* decoder widths (reference: 2000,500,500) */
    int32_t input_type;                              /* 0 binary, 1 real                         */
    int32_t dtype;                                   /* DMVAE_F32 (parity) or DMVAE_BF16         */
    int32_t max_batch;                               /* largest per-rank batch                   */
    int32_t mode;                                    /* 0 exact KL, 1 relaxed (Gumbel-Softmax)   */
    float temperature;
    float beta1, beta2, adam_eps;
    uint64_t seed;
    int32_t deterministic;                           /* 1: no float atomics anywhere             */
    int32_t trunk;                                   /* DMVAE_TRUNK_MLP (default) or DMVAE_TRUNK_CNN */
    int32_t model;                                   /* DMVAE_MODEL_DMVAE (default) or DMVAE_MODEL_VADE */
    int32_t adam_ieee;                               /* bf16 plans: 1 = TF-Adam with the IEEE square root and division (exactly the
                                                      * fp32 mode's arithmetic on the fp32 master weights); 0 (default) = hardware
                                                      * sqrt / reciprocal, 1 ulp each (-0.06 ms at cfg5)                */
} dmvae_config;

typedef struct dmvae_tensor_info {
    char name[32];
    int64_t offset;      /* element offset in the parameter arena */
    int32_t rows, cols;  /* logical shape (rows = 1 for biases)   */
    int64_t ld;          /* row stride in elements                */
} dmvae_tensor_info;

typedef struct dmvae_sizes {
    int64_t param_elems;     /* f32 arena (also grad, m, v; and the bf16 shadow in elements) */
    int64_t work_bytes;      /* activation / scratch workspace                               */
    int32_t batch_pad;       /* padded batch rows                                            */
    int32_t input_pad;       /* padded input columns                                         */
    int32_t n_tensors;
    int32_t reserved;
} dmvae_sizes;

typedef struct dmvae_buffers {
    float* param; float* grad; float* m; float* v;
    void* param_bf16;     /* bf16 shadow (may be NULL when dtype == DMVAE_F32) */
    void* work;           /* work_bytes, 256-byte aligned, zero-initialised by the caller */
    void* state;          /* dmvae_state on the device */
    int64_t arena_elems;  /* elements allocated for each of param / grad / m / v (/ shadow): >= param_elems; a caller that
                           * cuts the arenas into equal slices per rank (reduce-scatter -> sharded Adam -> all-gather) pads
                           * them, and dmvae_plan_update_range then accepts ranges up to this size.  0 = param_elems */
} dmvae_buffers;

typedef struct dmvae_plan dmvae_plan;

int dmvae_plan_create(const dmvae_config* cfg, dmvae_plan** out);
void dmvae_plan_destroy(dmvae_plan* p);
int dmvae_plan_sizes(const dmvae_plan* p, dmvae_sizes* out);
int dmvae_plan_tensor(const dmvae_plan* p, int index, dmvae_tensor_info* out);
int dmvae_plan_bind(dmvae_plan* p, const dmvae_buffers* b);

/* batch assembly into the plan's input buffers (see dmvae_gather_rows) */
int dmvae_plan_load_batch(dmvae_plan* p, void* stream, const float* data, int64_t n_rows,
                          const int32_t* perm, int64_t first, int n_valid, int use_state_cursor);
/* The same for a caller that goes straight on to ONE dmvae_plan_forward_backward / _train_step on this batch.  On bf16 plans whose
 * output layer the small-tile kernel runs (input_dim a multiple of 4) only the bf16 copy of the batch is written: the f32 copy's one
 * reader in a step, the reconstruction epilogue of the output layer (base_models.py:72-85), fetches its target rows from `data`
 * through `perm` itself.  `data` and `perm` must stay alive and unchanged until that step has run; the "x" view is not valid
 * afterwards.  Any other plan: exactly dmvae_plan_load_batch. */
int dmvae_plan_load_batch_step(dmvae_plan* plan, void* stream, const float* data, int64_t n_rows, const int32_t* perm,
                               int64_t first, int n_valid, int use_state_cursor);
/* Batch assembly overlapped with the step (Dataset.get_batches, includes/utils.py:449-463, one batch ahead): arms the assembly of the NEXT
 * batch inside the next dmvae_plan_forward_backward / _train_step -- by workgroups riding in one of its launches where that launch leaves
 * them room, else by a gather launch of its own -- into the other of the plan's two bf16 batch buffers; with use_state_cursor the rows are
 * those of the device cursor after that pass has advanced it.  dmvae_plan_swap_batch (host state only) then makes that batch the current
 * one, in place of the dmvae_plan_load_batch_step in front of the following step.  DMVAE_EUNSUPPORTED on plans that assemble their batches
 * with dmvae_plan_load_batch (f32, conv trunk, output layer on the macro tile).  `data` / `perm`: alive and unchanged until the step that
 * CONSUMES the prefetched batch has run -- i.e. the step AFTER the one that assembles it (its reconstruction epilogue reads the targets
 * through them).  A captured step holds the buffer it was captured with: capture one graph per buffer and alternate them. */
int dmvae_plan_prefetch_batch(dmvae_plan* plan, const float* data, int64_t n_rows, const int32_t* perm, int64_t first, int n_valid,
                              int use_state_cursor);
int dmvae_plan_swap_batch(dmvae_plan* plan);
/* forward + loss + backward: fills the grad arena and the loss partials.
 * eps / gumbel: caller-supplied noise (parity mode) or NULL (on-device Philox). */
int dmvae_plan_forward_backward(dmvae_plan* p, void* stream, int n_valid,
                                const float* eps, int64_t ld_eps, const float* gumbel, int64_t ld_gumbel,
                                float inv_B);
/* The same work cut into three segments for data parallelism: after segment k one contiguous
 * bucket of the gradient arena is final, so its all-reduce can run while the next segment computes
 * (the reference has no counterpart: it is single-process).  Segment 0 = forward, loss, decoder
 * backward; 1 = heads backward; 2 = trunk backward; call them in order with the same arguments.
 * dmvae_plan_grad_buckets: bounds[5] in arena elements.  The arena holds every weight matrix first and, from
 * bounds[3] (a multiple of 4096) on, the small fp32 tensors: every bias, then the prior tables.  Segment 0 completes
 * the weights [bounds[2], bounds[3]), segment 1 [bounds[1], bounds[2]), segment 2 [bounds[0], bounds[1]); the tail
 * [bounds[3], bounds[4]) is complete when segment 2 has run.  A data-parallel caller can reduce-scatter the weight
 * buckets, update its owned slices and all-gather the bf16 SHADOW of the weights (the GEMMs read nothing else of
 * them), and all-reduce + update the tail on every rank (the epilogues and the latent kernel read it in fp32).
 * Results are bit-identical to dmvae_plan_forward_backward. */
int dmvae_plan_forward_backward_stage(dmvae_plan* p, void* stream, int stage, int n_valid,
                                      const float* eps, int64_t ld_eps, const float* gumbel, int64_t ld_gumbel,
                                      float inv_B);
int dmvae_plan_grad_buckets(const dmvae_plan* p, int64_t bounds[5]);
/* How many weight-gradient launches the staged backward issues: 3 (default) = one per segment, as above; 2 = segment 0's
 * weight-gradient problems stay queued and go out with segment 1's in ONE grouped launch, so that [bounds[1], bounds[3])
 * -- decoder + heads, 86 % of the weights of the MNIST-shaped stacks -- is complete when segment 1 has run and its
 * collective can overlap segment 2 (the trunk), at the price of two weight-gradient grids per step instead of three
 * (MNIST-sized arenas: three small grids fill the chip worse than they hide wire time).
 * Results stay bit-identical to dmvae_plan_forward_backward. */
int dmvae_plan_set_stage_groups(dmvae_plan* p, int n_groups);
/* Adam (+ bf16 shadow refresh); grad_scale = 1/world_size.  The loss finalize at the end of
 * dmvae_plan_forward_backward has already advanced state->adam_t for this step: every
 * forward_backward is to be followed by exactly one update. */
int dmvae_plan_update(dmvae_plan* p, void* stream, float grad_scale);
/* the same on the arena elements [lo, hi) only (4-aligned): with the bucketed gradient exchange each
 * bucket is updated as soon as its all-reduce has landed, while later buckets are still in flight */
int dmvae_plan_update_range(dmvae_plan* p, void* stream, float grad_scale, int64_t lo, int64_t hi);
/* forward + loss + backward + Adam as ONE sequence with the update fused into the dW launch
 * (single-process training on a bf16 plan: no gradient exchange between backward and update).
 * Equivalent to dmvae_plan_forward_backward followed by dmvae_plan_update(grad_scale = 1), except
 * that the gradient arena is not written.  f32 plans run exactly that pair. */
int dmvae_plan_train_step(dmvae_plan* p, void* stream, int n_valid,
                          const float* eps, int64_t ld_eps, const float* gumbel, int64_t ld_gumbel,
                          float inv_B);
/* inference pieces used by get_accuracy / reconstruction / sampling:
 * encode: X (loaded batch) -> mean, log_var, logits (f32, in the workspace)
 * decode: Z (f32 [n][latent_dim], caller) -> sigmoid/identity reconstruction (f32, workspace) */
int dmvae_plan_encode(dmvae_plan* p, void* stream, int n_valid);
int dmvae_plan_decode(dmvae_plan* p, void* stream, const float* Z, int64_t ldz, int n_valid);
/* workspace views: name in {"mean","log_var","logits","recon","weights","Z","xlogits","x"};
 * returns the device pointer, leading dimension (elements) and dtype. */
int dmvae_plan_view(const dmvae_plan* p, const char* name, void** ptr, int64_t* ld, int32_t* dtype);

/* Measurement and tuning entry points (per-kernel timing for bench.py's roofline leg, probes, tile knobs) are
 * declared in dmvae_hip_debug.h: exported by the same library, not part of the drop-in boundary. */
int dmvae_abi_version(void);
const char* dmvae_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* DMVAE_HIP_H */
