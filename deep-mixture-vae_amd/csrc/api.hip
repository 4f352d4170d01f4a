// extern "C" surface of libdmvae_hip.so (include/dmvae_hip.h): argument
// checking, the step plan (arena layout + the launch sequence of one
// session.run([loss, train_step])), and HIP-event profiling of every launch.
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "kernels.h"
#include "measure.h"

namespace dmvae {

// ------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return 0;
}

// ------------------------------------------------------------------ profiling
// One record per ProfScope: the event BRACKET around the scope's launches (e0, e1: holds the launches' dispatch besides the
// kernels) and, per kernel launched inside it, the pair bound to that dispatch (k: the kernel's own begin -> end).
struct ProfRec { const char* name; double flops, bytes; hipEvent_t e0, e1; std::vector<std::pair<hipEvent_t, hipEvent_t>> k; };
static bool g_prof_on = false;
static std::vector<ProfRec> g_prof;
static std::vector<hipEvent_t> g_event_pool;
static std::mutex g_prof_mu;
static thread_local int g_prof_cur = -1;       // slot of the innermost open ProfScope of this thread

static hipEvent_t get_event() {
    if (!g_event_pool.empty()) { hipEvent_t e = g_event_pool.back(); g_event_pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
ProfScope::ProfScope(hipStream_t s_, const char* name, double flops, double bytes) : s(s_), slot(-1), outer(-1) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    ProfRec r{name, flops, bytes, get_event(), get_event(), {}};
    (void)hipEventRecord(r.e0, s);
    slot = (int)g_prof.size();
    g_prof.push_back(r);
    outer = g_prof_cur;
    g_prof_cur = slot;
}
ProfScope::~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    (void)hipEventRecord(g_prof[slot].e1, s);
    g_prof_cur = outer;
}
bool prof_launch_events(hipEvent_t* e0, hipEvent_t* e1) {
    if (!g_prof_on || g_prof_cur < 0) return false;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (g_prof_cur >= (int)g_prof.size()) return false;
    *e0 = get_event();
    *e1 = get_event();
    g_prof[g_prof_cur].k.emplace_back(*e0, *e1);
    return true;
}

static const char* gemm_name(int dtype, int layout) {
    static const char* n[2][3] = {{"gemm_f32_fwd", "gemm_f32_dx", "gemm_f32_dw"}, {"gemm_bf16_fwd", "gemm_bf16_dx", "gemm_bf16_dw"}};
    return n[dtype][layout];
}

static inline int pad64(int x) { return (x + 63) / 64 * 64; }

static int gemm_checked(hipStream_t s, int dtype, int layout, int M, int N, int K, const void* A, int64_t lda,
                        const void* B, int64_t ldb, const dmvae_epilogue* epi, int split, GemmArgs* deferred = nullptr,
                        int conv_p = 0, int conv_c = 0) {
    DMVAE_REQUIRE(dtype == DMVAE_F32 || dtype == DMVAE_BF16, "dmvae_gemm: bad dtype %d", dtype);
    DMVAE_REQUIRE(layout >= 0 && layout <= 2, "dmvae_gemm: bad layout %d", layout);
    DMVAE_REQUIRE(A && B && epi && epi->out, "dmvae_gemm: null pointer");
    DMVAE_REQUIRE(M > 0 && N > 0 && K > 0 && M % 64 == 0 && N % 64 == 0 && K % 64 == 0,
                  "dmvae_gemm: M=%d N=%d K=%d must be positive multiples of 64 (pad the operands)", M, N, K);
    DMVAE_REQUIRE(split >= 1 && K % (split * 64) == 0, "dmvae_gemm: split_k=%d does not divide K=%d into multiples of 64", split, K);
    DMVAE_REQUIRE(split == 1 || epi->kind == DMVAE_EPI_ATOMIC_F32, "dmvae_gemm: split_k > 1 needs DMVAE_EPI_ATOMIC_F32");
    const int esz = dtype == DMVAE_BF16 ? 2 : 4;
    DMVAE_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && (lda * esz) % 16 == 0 && (ldb * esz) % 16 == 0,
                  "dmvae_gemm: operands must be 16-byte aligned with 16-byte multiple row strides");
    GemmArgs a;
    a.A = A; a.lda = lda; a.B = B; a.ldb = ldb;
    a.M = M; a.N = N; a.K = K; a.k_split = K / split; a.group_m = 8;
    a.conv_p = conv_p; a.conv_c = conv_c;
    DMVAE_REQUIRE(conv_c == 0 || ((conv_c == 32 || conv_c % 64 == 0) && lda >= conv_c &&
                                  (layout == DMVAE_GEMM_DW ? M == pad64(9 * conv_c) : (K == pad64(9 * conv_c) && split == 1))),
                  "dmvae_gemm: conv mode: 32 or a multiple of 64 channels per tap (<= lda), 9 taps padded to 64 along K (M for the weight gradient)");
    a.epi = *epi;
    if (a.epi.m_valid <= 0) a.epi.m_valid = M;
    if (a.epi.n_valid <= 0) a.epi.n_valid = N;
    if (deferred) { *deferred = a; return 0; }      // caller batches it into a grouped launch
    if (dtype == DMVAE_BF16) return gemm_bf16_dispatch(s, layout, a, split);   // (profiled per template instantiation inside)
    const double bytes = 4.0 * ((double)M * K + (double)K * N + (double)M * N);
    ProfScope ps(s, gemm_name(dtype, layout), 2.0 * M * N * (double)K, bytes);
    return gemm_f32_dispatch(s, layout, a, split);
}

// RECON loss partials live on the 64x64 cell grid of the output whatever tile a kernel uses
static int gemm_partials(int, int M, int N) { return (M / 64) * (N / 64); }

}  // namespace dmvae

using namespace dmvae;

constexpr int KSPLIT_MAX_ROWS = 256, KSPLIT_MAX_SLICES = 8;      // K slices of thin dense launches (ksplit_for): batches of at most this many rows

// tuning knob (dmvae_debug_set_knob 10): K slices of the dense weight-gradient group (0 = the plan's rule, 1 = none, 2 / 4 = forced where
// the plan has the slabs); see dmvae_plan::dw_slices_max
static int g_dw_slices = 0;
// tuning knob (dmvae_debug_set_knob 11): with K slices, the layers whose shape divides by 256 on the macro tile (1, default) or every
// layer on the small tiles (0)
static int g_dw_macro = 1;
static int g_pf_rides = 1;                  // tuning knob (dmvae_debug_set_knob 17): a prefetched batch's gather rides in the dZ GEMM where it has the room (1), or is a launch of its own (0)
static int g_fin_rides = 1;                 // tuning knob (dmvae_debug_set_knob 16): the step_finalize blocks ride in the dZ GEMM where it has the room, else in the heads' dX launch
                                            // (1, default); always in the heads' dX launch (2); a launch of their own (0)
static int g_heads_dx_form = 0;             // tuning knob (dmvae_debug_set_knob 12): the dX of the two head layers as one grouped launch (0 / 1) or as two launches (2)

// ====================================================================== plan
static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

static int conv_dw_split(int64_t K, int tiles);

struct PLayer {
    std::string name;
    int in, out, in_pad, out_pad;
    int64_t w_off, b_off, ldw;    // arena element offsets; ldw = row stride of W
};

// One 3x3 SAME convolution of the CNN trunk (base_models.py:181-201), as the GEMMs it runs as (conv.hip).
// Activations are zero-bordered: [Bp][P][P][ld] (P = hw + 2) with P + 1 zero guard rows on either end; an
// o_* offset points at the guard, rows0() at padded pixel 0.
struct PConv {
    std::string name;
    int cin, cout, hw, pool;          // channels, image side of its input = output, 2x2 SAME max-pool behind it
    int cin_ld, cout_ld;              // channel stride of the input / output activation: the channel count itself (32, 64, 128; the image: 1)
    int cin_np, cout_np;              // channels padded to 64 = N of the GEMM that produces such an activation, ld of the weight matrix
    int P;                            // hw + 2
    int kdim, ktdim;                  // K of the forward / input-gradient GEMM: 9 * cin / 9 * cout padded to 64 (first layer: an explicit 9-column patch matrix)
    int64_t w_off, b_off;             // W [kdim][cout_ld]: HWIO flattened, rows (tap, c < cin), zero pad rows / columns; b [cout_ld]
    int64_t o_act, o_dact;            // relu output and its gradient
    int64_t o_pool, o_dpool;          // pooled output (zero-bordered for the next conv; the last one plain = the flat trunk output) and its gradient
};

struct dmvae_plan {
    dmvae_config cfg;
    std::vector<PConv> conv;          // cfg.trunk == DMVAE_TRUNK_CNN only
    int flat = 0;                     // 4*4*128 = 2048: width of the flattened pool output feeding enc0
    int64_t o_dflat = 0, o_wt = 0, o_c0part = 0, o_cslab = 0;     // o_cslab: K-slice slabs of the conv weight gradients (cfg.deterministic)
    int64_t conv_param_end = 0;
    int Bp, Ip, Dp, Kp, Hp, Tp;   // padded batch / input / latent / classes / head / trunk
    int es;                        // bytes per activation element
    std::vector<PLayer> enc, dec;
    PLayer zc, mv, lg, out;        // fused z|c hidden, fused mean|log_var, logits, output layer
    int64_t prior_off;             // prior_means [K][D] then prior_log_vars [K][D], contiguous
    int64_t tail_off = 0;          // [tail_off, param_elems): every bias, then the prior tables -- the tensors the epilogues and the
                                   // latent kernel read in fp32 (0.3 % of the arena), kept apart from the weight matrices
    int64_t param_elems;
    std::vector<dmvae_tensor_info> tensors;
    // workspace byte offsets
    int64_t o_x, o_xf, o_hzc, o_mv, o_lg, o_Z, o_Zf, o_gmu, o_glv, o_clv, o_w, o_dlg, o_recon, o_dl, o_dmv, o_dhzc;
    std::vector<int64_t> o_enc, o_dec, o_denc, o_ddec;
    int64_t o_rpart, o_lpart, o_dprior, o_cs, o_lws = 0;
    int64_t lws_bytes = 0;            // scratch of the MFMA form of the latent contractions (0: the one-kernel form)
    // K slices of the thin dense launches of a SMALL batch (ksplit_for): slabs of f32 partial tiles + one ticket per tile (zero between launches)
    int64_t o_ksws = -1, ksws_elems = 0, o_ktick = -1;
    int n_rpart, n_lblk, n_pblk;      // loss-partial blocks of the latent kernel; rows of its prior-table gradient partials (as of plan creation)
    int n_lblk_cap = 0;               // rows the partial-sum buffers hold
    int64_t cs_elems;
    int64_t work_bytes;
    dmvae_buffers buf;
    bool bound;
    std::vector<GemmArgs> dw_queue;   // dW problems queued (bf16: flushed as grouped launches)
    // Large batches (Bp >= 8192, MNIST-shaped layers): the dW group's tile count is fixed by the parameter shapes while its K = the
    // batch grows -- few, very long workgroups (cfg3: 256 K tiles each) that fill the CUs' slots unevenly and re-fetch their operand
    // panels (1.75x at cfg3, profiles/r03_cfg3_pmc_traffic.txt).  Then every problem is cut into dw_slices K slices, each a set of
    // ordinary workgroups of the SAME grouped launch storing its partial product (and bias partial) into its own slab -- a full image
    // of the gradient arena per slice, [dw_slices_max][param_elems] floats at o_dwslab -- and the slabs are added in ascending order
    // by the Adam kernel (adam_slabs: fused step) or by slab_reduce into the gradient arena (backward alone).  No float atomics.
    int dw_slices_max = 1;            // 1: the plan has no slabs
    int64_t o_dwslab = 0;
    int dw_slices_now = 1;            // slices of the pass being enqueued
    bool dw_macro_now = false;        // ... and whether its 256-divisible layers take the macro tile (their bias gradient: a bias-only strip)
    int dw_macro_tiles = 0;           // 256x256 tiles of those layers (per K slice)
    bool fused_update = false;        // set for the duration of dmvae_plan_train_step on a bf16 plan
    bool staged = false;              // dmvae_plan_forward_backward_stage: every segment launches its own dW group
    int stage_groups = 3;             // ... or (2) segments 0 and 1 share one: dmvae_plan_set_stage_groups
    // dmvae_plan_load_batch_step: the batch was assembled for a step that follows at once -- no f32 copy of it exists; the output
    // layer's reconstruction epilogue reads its targets from the dataset through the same permutation (bf16 plans whose output
    // layer runs on the small-tile kernel, input_dim a multiple of 4)
    struct { const float* data; int64_t n_rows; const int32_t* perm; int64_t first; int n_valid; const void* st; } tsrc{};
    bool tsrc_valid = false;          // cleared by dmvae_plan_load_batch (which writes the f32 copy)
    bool tsrc_used = false;           // a forward pass has consumed that batch: the device cursor has moved on (step_finalize), so a second
                                      // pass without a reload would pair the old bf16 batch with the NEXT batch's targets -- refused
    bool tgt_gather = false;          // the plan is eligible
    // dmvae_plan_prefetch_batch: the NEXT batch is assembled while this step runs -- by workgroups riding in the trunk's dX launch where that
    // launch leaves them CUs of their own (forward_backward_impl), else by a gather launch of its own -- into the OTHER of two bf16 batch
    // buffers; dmvae_plan_swap_batch then makes it the current one.  The step itself no longer starts with a gather.
    int64_t o_x2 = -1;                // the second bf16 batch buffer (bf16 dense plans)
    int xsel = 0;                     // the current one
    struct { const float* data; int64_t n_rows; const int32_t* perm; int64_t first; int n_valid; const void* st; bool armed, done; dmvae_gather_args gat; } pf{};
    bool vade = false;                // cfg.model == DMVAE_MODEL_VADE: no head hidden layers, no logits; latent mode 2
    // bias gradients of the macro-tile path: the GEMM that PRODUCES a dY (256x256 kernel, ReLU-gate / recon epilogue) leaves
    // its column sums per 256-row tile, [Bp/256][width] floats per dY tensor; the dW problem of that layer picks them up
    int64_t o_cs_dl = -1, o_cs_dhzc = -1;
    std::vector<int64_t> o_cs_ddec, o_cs_denc;
    std::map<const void*, std::pair<const float*, int64_t>> csum_of;     // dY base pointer -> (partials, ld), filled during a pass
};

static void add_tensor(dmvae_plan* p, const std::string& name, int64_t off, int rows, int cols, int64_t ld) {
    dmvae_tensor_info t;
    memset(&t, 0, sizeof(t));
    snprintf(t.name, sizeof(t.name), "%s", name.c_str());
    t.offset = off; t.rows = rows; t.cols = cols; t.ld = ld;
    p->tensors.push_back(t);
}

extern "C" int dmvae_plan_create(const dmvae_config* c, dmvae_plan** out) {
    DMVAE_REQUIRE(c && out, "dmvae_plan_create: null argument");
    DMVAE_REQUIRE(c->input_dim > 0 && c->latent_dim > 0 && c->n_classes > 0 && (c->head_dim > 0 || c->model == DMVAE_MODEL_VADE), "dmvae_plan_create: bad dims");
    DMVAE_REQUIRE(c->model == DMVAE_MODEL_DMVAE || (c->model == DMVAE_MODEL_VADE && c->mode == 0),
                  "dmvae_plan_create: model %d (0 DMVAE, 1 VaDE; VaDE runs the exact KL)", c->model);
    DMVAE_REQUIRE(c->n_enc >= 1 && c->n_enc <= DMVAE_MAX_LAYERS && c->n_dec >= 1 && c->n_dec <= DMVAE_MAX_LAYERS, "dmvae_plan_create: 1..%d layers", DMVAE_MAX_LAYERS);
    DMVAE_REQUIRE(c->dtype == DMVAE_F32 || c->dtype == DMVAE_BF16, "dmvae_plan_create: bad dtype");
    DMVAE_REQUIRE(c->max_batch > 0, "dmvae_plan_create: max_batch must be > 0");
    dmvae_plan* p = new dmvae_plan();
    p->cfg = *c;
    p->bound = false;
    p->es = c->dtype == DMVAE_BF16 ? 2 : 4;
    p->Bp = (c->max_batch + 127) / 128 * 128;
    p->Ip = pad64(c->input_dim);
    p->Dp = pad64(c->latent_dim);
    p->Kp = pad64(c->n_classes);
    p->vade = c->model == DMVAE_MODEL_VADE;
    p->Hp = pad64(p->vade ? 64 : c->head_dim);
    // Arena layout: every weight matrix first (trunk, heads, decoder, output), then -- from tail_off, a multiple of 4096 elements --
    // the SMALL fp32 tensors: every bias in the same layer order, then the prior tables.  The GEMMs read the weights through the bf16
    // shadow; the biases (epilogues) and the prior tables (latent kernel) are read in fp32.  Kept apart, a data-parallel job can
    // reduce-scatter / shard-update the weight range and all-gather its bf16 SHADOW (half the bytes) while the tail
    // is all-reduced whole and updated on every rank (dmvae_plan_grad_buckets).  boff counts tail-relative until tail_off is known.
    int64_t off = 0, boff = 0;
    auto place = [&](PLayer& L, const std::string& name, int in, int out_, int in_pad, int out_pad) {
        L.name = name; L.in = in; L.out = out_; L.in_pad = in_pad; L.out_pad = out_pad; L.ldw = out_pad;
        L.w_off = off; off += (int64_t)in_pad * out_pad;
        L.b_off = boff; boff += out_pad;
    };
    int prev = c->input_dim, prev_pad = p->Ip;
    if (c->trunk == DMVAE_TRUNK_CNN) {
        // the checked-in encoder (base_models.py:176-216): reshape (-1,28,28,1); conv 1-32, 32-32, pool, 32-64, 64-64,
        // pool, 64-128, 128-128, pool (SAME: 28 -> 14 -> 7 -> 4); flatten (h,w,c) = 2048; FullyConnected 2048 -> enc[0]
        if (c->input_dim != 784 || c->n_enc != 1) { delete p; set_error("dmvae_plan_create: the CNN trunk takes 784 inputs and one fc layer"); return DMVAE_EINVAL; }
        static const int spec[6][4] = {{1, 32, 28, 0}, {32, 32, 28, 1}, {32, 64, 14, 0}, {64, 64, 14, 1}, {64, 128, 7, 0}, {128, 128, 7, 1}};
        int ld_in = 1;
        for (int i = 0; i < 6; ++i) {
            PConv L;
            L.name = "conv" + std::to_string(i);
            L.cin = spec[i][0]; L.cout = spec[i][1]; L.hw = spec[i][2]; L.pool = spec[i][3];
            L.cin_ld = ld_in; L.cout_ld = L.cout; L.cin_np = pad64(L.cin); L.cout_np = pad64(L.cout); L.P = L.hw + 2;
            L.kdim = pad64(9 * L.cin); L.ktdim = pad64(9 * L.cout);
            L.w_off = off; off += (int64_t)L.kdim * L.cout_np;
            L.b_off = boff; boff += L.cout_np;
            add_tensor(p, "W_" + L.name, L.w_off, 9 * L.cin, L.cout, L.cout_np);
            add_tensor(p, "b_" + L.name, L.b_off, 1, L.cout, L.cout_np);
            p->conv.push_back(L);
            ld_in = L.cout_ld;
        }
        p->conv_param_end = off;
        p->flat = 4 * 4 * 128;
        prev = p->flat; prev_pad = p->flat;
    }
    for (int i = 0; i < c->n_enc; ++i) {
        PLayer L;
        place(L, "enc" + std::to_string(i), prev, c->enc[i], prev_pad, pad64(c->enc[i]));
        add_tensor(p, "W_" + L.name, L.w_off, L.in, L.out, L.ldw);
        add_tensor(p, "b_" + L.name, L.b_off, 1, L.out, L.out_pad);
        p->enc.push_back(L);
        prev = c->enc[i]; prev_pad = L.out_pad;
    }
    p->Tp = prev_pad;
    if (!p->vade) {
        place(p->zc, "zc", prev, 2 * p->Hp, p->Tp, 2 * p->Hp);
        add_tensor(p, "W_zh", p->zc.w_off, prev, c->head_dim, p->zc.ldw);
        add_tensor(p, "b_zh", p->zc.b_off, 1, c->head_dim, 2 * p->Hp);
        add_tensor(p, "W_ch", p->zc.w_off + p->Hp, prev, c->head_dim, p->zc.ldw);
        add_tensor(p, "b_ch", p->zc.b_off + p->Hp, 1, c->head_dim, 2 * p->Hp);
    }
    // [mean | log_var]: off the z-hidden layer (DMVAE, base_models.py:234-239) or straight off the trunk (VaDE, :501-507)
    const int mv_in = p->vade ? prev : c->head_dim, mv_in_pad = p->vade ? p->Tp : p->Hp;
    place(p->mv, "mv", mv_in, 2 * p->Dp, mv_in_pad, 2 * p->Dp);
    add_tensor(p, "W_mean", p->mv.w_off, mv_in, c->latent_dim, p->mv.ldw);
    add_tensor(p, "b_mean", p->mv.b_off, 1, c->latent_dim, 2 * p->Dp);
    add_tensor(p, "W_logvar", p->mv.w_off + p->Dp, mv_in, c->latent_dim, p->mv.ldw);
    add_tensor(p, "b_logvar", p->mv.b_off + p->Dp, 1, c->latent_dim, 2 * p->Dp);
    if (!p->vade) {
        place(p->lg, "logits", c->head_dim, c->n_classes, p->Hp, p->Kp);
        add_tensor(p, "W_logits", p->lg.w_off, c->head_dim, c->n_classes, p->lg.ldw);
        add_tensor(p, "b_logits", p->lg.b_off, 1, c->n_classes, p->Kp);
    }
    prev = c->latent_dim; prev_pad = p->Dp;
    for (int i = 0; i < c->n_dec; ++i) {
        PLayer L;
        place(L, "dec" + std::to_string(i), prev, c->dec[i], prev_pad, pad64(c->dec[i]));
        add_tensor(p, "W_" + L.name, L.w_off, L.in, L.out, L.ldw);
        add_tensor(p, "b_" + L.name, L.b_off, 1, L.out, L.out_pad);
        p->dec.push_back(L);
        prev = c->dec[i]; prev_pad = L.out_pad;
    }
    place(p->out, "out", prev, c->input_dim, prev_pad, p->Ip);
    add_tensor(p, "W_out", p->out.w_off, prev, c->input_dim, p->out.ldw);
    add_tensor(p, "b_out", p->out.b_off, 1, c->input_dim, p->Ip);
    p->tail_off = align_up(off, 4096);
    for (auto& L : p->conv) L.b_off += p->tail_off;
    for (auto& L : p->enc) L.b_off += p->tail_off;
    for (auto& L : p->dec) L.b_off += p->tail_off;
    p->zc.b_off += p->tail_off; p->mv.b_off += p->tail_off; p->lg.b_off += p->tail_off; p->out.b_off += p->tail_off;
    for (auto& t : p->tensors)
        if (t.name[0] == 'b' && t.name[1] == '_') t.offset += p->tail_off;
    off = p->tail_off + boff;
    p->prior_off = off;
    const int KD = c->n_classes * c->latent_dim;
    add_tensor(p, "prior_means", off, c->n_classes, c->latent_dim, c->latent_dim);
    add_tensor(p, "prior_log_vars", off + KD, c->n_classes, c->latent_dim, c->latent_dim);
    off += align_up(2 * (int64_t)KD, 64);
    p->param_elems = off;

    // ---- workspace
    int64_t w = 0;
    auto take = [&](int64_t bytes) { int64_t o = w; w += align_up(bytes, 256); return o; };
    const int64_t Bp = p->Bp, es = p->es;
    p->o_xf = take(Bp * p->Ip * 4);
    p->o_x = (c->dtype == DMVAE_F32) ? p->o_xf : take(Bp * p->Ip * es);
    if (c->dtype == DMVAE_BF16 && p->conv.empty()) p->o_x2 = take(Bp * p->Ip * es);
    {
        int64_t wmax = 0;
        for (size_t i = 0; i < p->conv.size(); ++i) {
            PConv& L = p->conv[i];
            auto bordered = [&](int P, int ld) { return take((Bp * P * P + 2 * (P + 1)) * ld * es); };
            L.o_act = bordered(L.P, L.cout_ld);
            L.o_dact = bordered(L.P, L.cout_ld);
            const int ho = (L.hw + 1) / 2;
            const bool last = i + 1 == p->conv.size();
            L.o_pool = L.pool ? (last ? take(Bp * ho * ho * L.cout_ld * es) : bordered(ho + 2, L.cout_ld)) : 0;
            L.o_dpool = (L.pool && !last) ? bordered(ho + 2, L.cout_ld) : 0;
            if (i > 0) wmax = std::max<int64_t>(wmax, (int64_t)L.cin_np * L.ktdim);
        }
        if (!p->conv.empty()) {
            p->o_wt = take(wmax * es);
            p->o_dflat = take(Bp * p->flat * es);
            p->o_c0part = take((int64_t)conv_first_dw_blocks(p->conv.front().hw, Bp) * 320 * 4);
            {      // split-K slabs instead of float atomics (always: measured FASTER than the atomic form, 3.06 vs 3.23 ms/step): the largest layer's split x (kdim + 1) x cout_np floats
                int64_t mx = 0;
                for (size_t i = 1; i < p->conv.size(); ++i) {
                    const PConv& L = p->conv[i];
                    const int sp = conv_dw_split((int64_t)Bp * L.P * L.P, (L.kdim / 64) * (L.cout_np / 64));
                    mx = std::max<int64_t>(mx, (int64_t)sp * (L.kdim + 1) * L.cout_np);
                }
                p->o_cslab = take(mx * 4);
            }
        }
    }
    for (auto& L : p->enc) p->o_enc.push_back(take(Bp * L.out_pad * es));
    p->o_hzc = take(Bp * 2 * p->Hp * es);
    p->o_mv = take(Bp * 2 * p->Dp * 4);
    p->o_lg = take(Bp * p->Kp * 4);
    p->o_Z = take(Bp * p->Dp * es);
    p->o_Zf = take(Bp * p->Dp * 4);
    p->o_gmu = take(Bp * p->Dp * 4);
    p->o_glv = take(Bp * p->Dp * 4);
    p->o_clv = take(Bp * p->Dp * 4);
    p->o_w = take(Bp * p->Kp * 4);
    p->o_dlg = take(Bp * p->Kp * es);
    for (auto& L : p->dec) p->o_dec.push_back(take(Bp * L.out_pad * es));
    p->o_recon = take(Bp * p->Ip * 4);
    p->o_dl = take(Bp * p->Ip * es);
    for (auto& L : p->dec) p->o_ddec.push_back(take(Bp * L.out_pad * es));
    p->o_dmv = take(Bp * 2 * p->Dp * es);
    p->o_dhzc = take(Bp * 2 * p->Hp * es);
    for (auto& L : p->enc) p->o_denc.push_back(take(Bp * L.out_pad * es));
    if (c->dtype == DMVAE_BF16 && p->Bp % 256 == 0 && !getenv("DMVAE_NO_CSUM")) {
        const int64_t tr = p->Bp / 256;
        p->o_cs_dl = take(tr * p->Ip * 4);
        p->o_cs_dhzc = take(tr * 2 * p->Hp * 4);
        for (auto& L : p->dec) p->o_cs_ddec.push_back(take(tr * L.out_pad * 4));
        for (auto& L : p->enc) p->o_cs_denc.push_back(take(tr * L.out_pad * 4));
    }
    p->n_rpart = gemm_partials(c->dtype, p->Bp, p->Ip);
    p->o_rpart = take((int64_t)p->n_rpart * 4);
    p->n_lblk = p->vade ? latent_vade_nblocks(p->Bp) : latent_nblocks(p->Bp, c->latent_dim, c->n_classes);
    // The block count of the latent kernel follows a tuning knob (14) that may change after the plan exists: the partial-sum buffers are
    // sized for the LARGEST count any setting can produce (16 rows per block), the count in use is taken at enqueue time
    // (latent_blocks_now) and checked against this capacity.
    p->n_lblk_cap = std::max(p->n_lblk, (p->Bp + 15) / 16);
    p->o_lpart = take((int64_t)p->n_lblk_cap * 2 * 4);
    p->lws_bytes = (!p->vade && latent_mfma_applies(c->latent_dim, c->n_classes, c->mode)) ? latent_mfma_ws_bytes(p->Bp, c->latent_dim, c->n_classes) : 0;
    p->n_pblk = p->lws_bytes ? 1 : p->n_lblk;          // the MFMA form delivers the prior-table gradient complete, in one row
    p->o_dprior = take((int64_t)(p->lws_bytes ? 1 : p->n_lblk_cap) * 2 * KD * 4);
    if (p->lws_bytes) p->o_lws = take(p->lws_bytes);
    if (c->dtype == DMVAE_BF16 && p->Bp <= 2048) {
        // dense launches (<= KSPLIT_MAX_ROWS rows; the largest user: a 512-wide output, 8 slices, 64 x 64 tiles) and the fused heads + latent launch
        // (blocks x slices <= 256: up to 2048 rows)
        int64_t need = p->Bp <= KSPLIT_MAX_ROWS ? (int64_t)(p->Bp / 64) * 8 * KSPLIT_MAX_SLICES * 4096 : 0;
        if (p->Dp == 64 || p->Dp == 128)
            for (int S = 2; S <= KSPLIT_MAX_SLICES; S *= 2)
                if ((p->Bp / 16) * S <= 256) need = std::max(need, heads_latent_kslice_floats(p->Bp, p->Dp, S));
        if (need > 0) {
            p->ksws_elems = need;
            p->o_ksws = take(p->ksws_elems * 4);
            p->o_ktick = take(256 * 4);
        }
    }
    int maxN = std::max(std::max(2 * p->Hp, p->Ip), p->flat);
    for (auto& L : p->enc) maxN = std::max(maxN, L.out_pad);
    for (auto& L : p->dec) maxN = std::max(maxN, L.out_pad);
    maxN = std::max(maxN, 2 * KD);
    p->cs_elems = (int64_t)64 * maxN;
    p->o_cs = take(p->cs_elems * 4);
    {   // K slices of the dW group (see dw_slices_max): bf16, dense trunk, a batch of >= 8192 rows, no layer on the macro tile
        bool macro = false;
        auto chk = [&](const PLayer& L) { macro = macro || gemm_bf16_256_ok(DMVAE_GEMM_DW, DMVAE_EPI_ADAM, L.in_pad, L.out_pad, (int)Bp, false); };
        for (auto& L : p->enc) chk(L);
        for (auto& L : p->dec) chk(L);
        if (!p->vade) { chk(p->zc); chk(p->lg); }
        chk(p->mv); chk(p->out);
        // every layer's problem + a bias strip for each layer the macro tile may take must fit ONE grouped launch queue: the fused
        // step sums the slabs once, behind it
        int n_dw = 0, n_elig = 0, tiles = 0;
        auto cnt = [&](const PLayer& L) {
            ++n_dw;
            if (L.in_pad % 256 == 0 && L.out_pad % 256 == 0) { ++n_elig; tiles += (L.in_pad / 256) * (L.out_pad / 256); }
        };
        for (auto& L : p->enc) cnt(L);
        for (auto& L : p->dec) cnt(L);
        if (!p->vade) { cnt(p->zc); cnt(p->lg); }
        cnt(p->mv); cnt(p->out);
        if (c->dtype == DMVAE_BF16 && p->conv.empty() && Bp >= 8192 && Bp % (4 * 64) == 0 && !macro && n_dw + n_elig < DMVAE_MAX_GROUP) {
            p->dw_slices_max = 4;
            p->o_dwslab = take((int64_t)p->dw_slices_max * p->param_elems * 4);
            p->dw_macro_tiles = tiles;
        }
    }
    // the step path may leave the batch's f32 copy out (dmvae_plan_load_batch_step): bf16, 16-byte aligned dataset rows, and an
    // output layer that the small-tile kernel runs (the macro tile fetches its targets in row batches from the copy)
    p->tgt_gather = c->dtype == DMVAE_BF16 && c->input_dim % 4 == 0 &&
                    !gemm_bf16_256_ok(DMVAE_GEMM_FWD, DMVAE_EPI_BIAS_RECON, p->Bp, p->Ip, p->dec.back().out_pad, false);
    p->work_bytes = w;
    *out = p;
    return 0;
}

extern "C" void dmvae_plan_destroy(dmvae_plan* p) {
    if (!p) return;
    delete p;
}

extern "C" int dmvae_plan_sizes(const dmvae_plan* p, dmvae_sizes* o) {
    DMVAE_REQUIRE(p && o, "dmvae_plan_sizes: null argument");
    o->param_elems = p->param_elems;
    o->work_bytes = p->work_bytes;
    o->batch_pad = p->Bp;
    o->input_pad = p->Ip;
    o->n_tensors = (int)p->tensors.size();
    o->reserved = 0;
    return 0;
}
extern "C" int dmvae_plan_tensor(const dmvae_plan* p, int i, dmvae_tensor_info* o) {
    DMVAE_REQUIRE(p && o && i >= 0 && i < (int)p->tensors.size(), "dmvae_plan_tensor: bad index %d", i);
    *o = p->tensors[i];
    return 0;
}
extern "C" int dmvae_plan_bind(dmvae_plan* p, const dmvae_buffers* b) {
    DMVAE_REQUIRE(p && b && b->param && b->grad && b->m && b->v && b->work && b->state, "dmvae_plan_bind: null buffer");
    DMVAE_REQUIRE(p->cfg.dtype == DMVAE_F32 || b->param_bf16, "dmvae_plan_bind: bf16 plan needs the bf16 parameter shadow");
    DMVAE_REQUIRE((uintptr_t)b->work % 256 == 0 && (uintptr_t)b->param % 256 == 0 && (uintptr_t)b->grad % 256 == 0, "dmvae_plan_bind: buffers must be 256-byte aligned");
    p->buf = *b;
    if (p->buf.arena_elems == 0) p->buf.arena_elems = p->param_elems;
    DMVAE_REQUIRE(p->buf.arena_elems >= p->param_elems && p->buf.arena_elems % 4 == 0, "dmvae_plan_bind: arena_elems %lld < the plan's %lld parameters (or not a multiple of 4)",
                  (long long)p->buf.arena_elems, (long long)p->param_elems);
    p->bound = true;
    return 0;
}

#define WS(p, off) (reinterpret_cast<char*>((p)->buf.work) + (off))
#define TRY(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

// weights as seen by the GEMMs: bf16 shadow or the f32 master
static inline const void* Wp(const dmvae_plan* p, int64_t off) {
    return p->cfg.dtype == DMVAE_BF16 ? (const void*)(reinterpret_cast<const bf16_t*>(p->buf.param_bf16) + off)
                                      : (const void*)(p->buf.param + off);
}
static inline const void* act_off(const dmvae_plan* p, int64_t base, int64_t elems) { return WS(p, base) + elems * p->es; }
static inline char* XB(const dmvae_plan* p, int other = 0) { return WS(p, ((p->xsel ^ other) && p->o_x2 >= 0) ? p->o_x2 : p->o_x); }      // the current (other = 1: the next) act copy of the batch

extern "C" int dmvae_plan_load_batch(dmvae_plan* p, void* stream, const float* data, int64_t n_rows, const int32_t* perm,
                                     int64_t first, int n_valid, int use_state_cursor) {
    DMVAE_REQUIRE(p && p->bound && data, "dmvae_plan_load_batch: plan not bound / null data");
    DMVAE_REQUIRE(n_valid >= 0 && n_valid <= p->cfg.max_batch, "dmvae_plan_load_batch: n_valid=%d exceeds max_batch=%d", n_valid, p->cfg.max_batch);
    hipStream_t s = (hipStream_t)stream;
    void* xa = p->cfg.dtype == DMVAE_BF16 ? (void*)XB(p) : nullptr;
    p->tsrc_valid = false;
    p->pf.armed = p->pf.done = false;
    return gather_launch(s, p->cfg.dtype, data, n_rows, p->cfg.input_dim, perm, first, p->cfg.max_batch, n_valid, p->Bp, xa, p->Ip,
                         reinterpret_cast<float*>(WS(p, p->o_xf)), p->Ip, p->Ip, use_state_cursor ? p->buf.state : nullptr);
}

// The same for a caller that goes straight on to dmvae_plan_forward_backward / _train_step (one step on this batch, nothing else
// reads it).  On an eligible plan (tgt_gather) only the bf16 copy of the batch is written: the f32 copy's one reader in a step,
// the reconstruction epilogue of the output layer, fetches its target rows from `data` through `perm` itself -- the same bytes
// read, 4 B per input element less written and re-read (cfg2: the gather 9.9 -> 6 us).  The "x" view is then NOT valid.
// Otherwise exactly dmvae_plan_load_batch.  `data` / `perm` must stay alive and unchanged until the step has run.
extern "C" int dmvae_plan_load_batch_step(dmvae_plan* p, void* stream, const float* data, int64_t n_rows, const int32_t* perm,
                                          int64_t first, int n_valid, int use_state_cursor) {
    DMVAE_REQUIRE(p && p->bound && data, "dmvae_plan_load_batch_step: plan not bound / null data");
    DMVAE_REQUIRE(n_valid >= 0 && n_valid <= p->cfg.max_batch, "dmvae_plan_load_batch_step: n_valid=%d exceeds max_batch=%d", n_valid, p->cfg.max_batch);
    if (!p->tgt_gather) return dmvae_plan_load_batch(p, stream, data, n_rows, perm, first, n_valid, use_state_cursor);
    p->tsrc.data = data; p->tsrc.n_rows = n_rows; p->tsrc.perm = perm; p->tsrc.first = first; p->tsrc.n_valid = n_valid;
    p->tsrc.st = use_state_cursor ? p->buf.state : nullptr;
    p->tsrc_valid = true;
    p->tsrc_used = false;
    p->pf.armed = p->pf.done = false;
    if constexpr (MEAS_NO_STEP_GATHER) {      // measurement build 11 only (measure.h): timing of a step without its gather
        static int calls = 0;
        if (++calls > 2) return 0;
    }
    return gather_launch((hipStream_t)stream, DMVAE_BF16, data, n_rows, p->cfg.input_dim, perm, first, p->cfg.max_batch, n_valid, p->Bp,
                         XB(p), p->Ip, nullptr, p->Ip, p->Ip, use_state_cursor ? p->buf.state : nullptr);
}

// Arms the assembly of the NEXT batch inside the next forward + backward pass (see dmvae_plan::pf); with use_state_cursor the rows are those
// of the device cursor AFTER that pass's step_finalize has advanced it.  Needs a current batch assembled by dmvae_plan_load_batch_step (or
// made current by dmvae_plan_swap_batch) on an eligible plan; `data` / `perm` must stay alive and unchanged until the step that CONSUMES the
// prefetched batch has run (the step after the one that assembles it: that step's reconstruction epilogue still reads the targets through them).
extern "C" int dmvae_plan_prefetch_batch(dmvae_plan* p, const float* data, int64_t n_rows, const int32_t* perm, int64_t first, int n_valid,
                                         int use_state_cursor) {
    DMVAE_REQUIRE(p && p->bound && data, "dmvae_plan_prefetch_batch: plan not bound / null data");
    DMVAE_REQUIRE(n_valid >= 0 && n_valid <= p->cfg.max_batch, "dmvae_plan_prefetch_batch: n_valid=%d exceeds max_batch=%d", n_valid, p->cfg.max_batch);
    if (!p->tgt_gather || p->o_x2 < 0) { set_error("dmvae_plan_prefetch_batch: this plan assembles its batches with dmvae_plan_load_batch (f32, conv trunk, or an output layer on the macro tile)"); return DMVAE_EUNSUPPORTED; }
    DMVAE_REQUIRE(p->tsrc_valid && !p->tsrc_used, "dmvae_plan_prefetch_batch: no current batch (dmvae_plan_load_batch_step or dmvae_plan_swap_batch first)");
    p->pf.data = data; p->pf.n_rows = n_rows; p->pf.perm = perm; p->pf.first = first; p->pf.n_valid = n_valid;
    p->pf.st = use_state_cursor ? p->buf.state : nullptr;
    p->pf.armed = true; p->pf.done = false;
    return 0;
}
// Makes the batch a pass has prefetched the current one (host state only: the other batch buffer and its target source).
extern "C" int dmvae_plan_swap_batch(dmvae_plan* p) {
    DMVAE_REQUIRE(p && p->bound, "dmvae_plan_swap_batch: plan not bound");
    DMVAE_REQUIRE(p->pf.done, "dmvae_plan_swap_batch: no pass has run since dmvae_plan_prefetch_batch");
    p->xsel ^= 1;
    p->tsrc.data = p->pf.data; p->tsrc.n_rows = p->pf.n_rows; p->tsrc.perm = p->pf.perm; p->tsrc.first = p->pf.first; p->tsrc.n_valid = p->pf.n_valid;
    p->tsrc.st = p->pf.st;
    p->tsrc_valid = true; p->tsrc_used = false;
    p->pf.armed = p->pf.done = false;
    return 0;
}

// K slices for the thin dense launches of a small batch.  At 100 rows (the reference's own default, train.py:215-216) every GEMM of the
// step is 2 x N / 64 tiles -- a few CUs, each walking the whole K chain and streaming its own weight panel: the launch lasts as long as that chain
// (K = 4096: 64 K tiles, 18 us).  Cut into S slices the chain is S times shorter on S times the CUs; the slices' partial tiles meet in a fixed-order
// sum by the last one to arrive (GemmArgs::tick): deterministic, no float atomics, no extra launch.  Returns S (1 = none).
static int g_ksplit = 1;              // tuning knob (dmvae_debug_set_knob 21): 1 = the rule below, 0 = never
static int ksplit_for(const dmvae_plan* p, int N, int K, int epi) {
    if (!g_ksplit || p->o_ksws < 0 || p->Bp > KSPLIT_MAX_ROWS || p->cfg.dtype != DMVAE_BF16 || K < 2048) return 1;
    if (epi != DMVAE_EPI_BIAS_RELU && epi != DMVAE_EPI_RELU_MASK && epi != DMVAE_EPI_BIAS_F32 && epi != DMVAE_EPI_LATENT) return 1;
    const int tiles = (epi == DMVAE_EPI_LATENT ? (p->Bp + 15) / 16 : p->Bp / 64) * (N / 64);      // (the dZ GEMM runs on 16-row tiles; a slab is sized for a 64 x 64 tile either way)
    int S = std::min(KSPLIT_MAX_SLICES, K / (g_ksplit == 2 ? 256 : 512));      // (knob 21 = 2: slices of 256 instead of 512)
    while (S > 1 && (K % (S * 64) || tiles * S > 256 || (int64_t)tiles * S * 4096 > p->ksws_elems)) --S;
    return tiles <= 64 ? std::max(S, 1) : 1;      // (a launch that already covers a quarter of the chip keeps its K chain)
}
// a dense problem of the plan with its K slices, if the rule gives it any
static int gemm_plan(dmvae_plan* p, hipStream_t s, int layout, int N, int K, const void* A, int64_t lda, const void* B, int64_t ldb, const dmvae_epilogue* e,
                     GemmArgs* deferred) {
    const int S = deferred ? 1 : ksplit_for(p, N, K, e->kind);
    if (S == 1) return gemm_checked(s, p->cfg.dtype, layout, p->Bp, N, K, A, lda, B, ldb, e, 1, deferred);
    GemmArgs a;
    TRY(gemm_checked(s, p->cfg.dtype, layout, p->Bp, N, K, A, lda, B, ldb, e, 1, &a));
    if (gemm_bf16_256_ok(layout, e->kind, p->Bp, N, K, false)) return gemm_bf16_dispatch(s, layout, a, 1);
    a.k_split = K / S;
    a.ws = reinterpret_cast<float*>(WS(p, p->o_ksws)); a.ws_elems = p->ksws_elems;
    a.tick = reinterpret_cast<int*>(WS(p, p->o_ktick));
    return gemm_bf16_dispatch(s, layout, a, S);
}

static int fwd_dense(dmvae_plan* p, hipStream_t s, const void* A, int64_t lda, int Kdim, const PLayer& L, int N, int64_t w_col,
                     int kind, void* out, int64_t ldo, GemmArgs* deferred = nullptr) {
    dmvae_epilogue e;
    memset(&e, 0, sizeof(e));
    e.kind = kind; e.out = out; e.ldo = ldo; e.bias = p->buf.param + L.b_off + w_col;
    return gemm_plan(p, s, DMVAE_GEMM_FWD, N, Kdim, A, lda, reinterpret_cast<const char*>(Wp(p, L.w_off + w_col)), L.ldw, &e, deferred);
}

// pointer to padded pixel 0 of a zero-bordered activation (skips the P + 1 guard rows)
static inline char* rows0(const dmvae_plan* p, int64_t off, int P, int ld) { return WS(p, off) + (int64_t)(P + 1) * ld * p->es; }

// CNN trunk forward (base_models.py:176-216).  First layer (one input channel): a direct kernel; every other
// convolution is ONE conv-mode GEMM straight off the zero-bordered activation (conv.hip) + bias + ReLU, then
// the border rows (which computed relu(bias)) are re-zeroed; SAME max-pools behind conv1 / conv3 / conv5.
static int conv_trunk_forward(dmvae_plan* p, hipStream_t s, const void** flat) {
    const int dt = p->cfg.dtype;
    const void* in = nullptr;                       // zero-bordered input of the next convolution, at padded pixel 0
    for (size_t i = 0; i < p->conv.size(); ++i) {
        const PConv& L = p->conv[i];
        const int M = p->Bp * L.P * L.P;
        dmvae_epilogue e;
        memset(&e, 0, sizeof(e));
        e.kind = DMVAE_EPI_BIAS_RELU; e.out = rows0(p, L.o_act, L.P, L.cout_ld); e.ldo = L.cout_ld; e.bias = p->buf.param + L.b_off;
        e.n_valid = L.cout;                          // a 32-channel output is stored 32 wide: the GEMM's other 32 columns are dropped
        if (i == 0) {   // one input channel: direct kernel, writes the border rows as zeros itself
            TRY(conv_first_fwd_launch(s, dt, WS(p, p->o_x), p->Ip, L.hw, p->Bp, Wp(p, L.w_off), L.cout_np, p->buf.param + L.b_off, L.cout,
                                      rows0(p, L.o_act, L.P, L.cout_ld), L.cout_ld));
        } else {
            TRY(gemm_checked(s, dt, DMVAE_GEMM_FWD, M, L.cout_np, L.kdim, in, L.cin_ld, Wp(p, L.w_off), L.cout_np, &e, 1, nullptr, L.P, L.cin));
            TRY(zero_border_launch(s, dt, rows0(p, L.o_act, L.P, L.cout_ld), L.P, L.cout_ld, p->Bp));
        }
        in = rows0(p, L.o_act, L.P, L.cout_ld);
        if (L.pool) {
            const bool last = i + 1 == p->conv.size();
            const int Po = (L.hw + 1) / 2 + 2;
            void* out = last ? (void*)WS(p, L.o_pool) : (void*)rows0(p, L.o_pool, Po, L.cout_ld);
            TRY(maxpool2_fwd_launch(s, dt, in, L.hw, L.cout_ld, p->Bp, out, last ? 0 : 1));
            in = out;
        }
    }
    *flat = in;
    return 0;
}

// largest split of K that keeps K-slices whole multiples of 64 and long enough to amortise a tile
static int conv_dw_split(int64_t K, int tiles) {
    const int64_t units = K / 64;
    int64_t target = std::min<int64_t>(units / 16, std::max(1, 2048 / std::max(1, tiles)));
    for (int64_t d = std::max<int64_t>(1, target); d >= 8; --d)       // a multiple of 8: one XCD per slice (gemm_bf16_kernel)
        if (units % d == 0 && d % 8 == 0) return (int)d;
    for (int64_t d = std::max<int64_t>(1, target); d > 1; --d)
        if (units % d == 0) return (int)d;
    return 1;
}

// CNN trunk backward from d_flat (gradient of the flattened pool output).  Per layer, last to first:
//   un-pool + ReLU gate (maxpool2_bwd_relu)  ->  dY, zero-bordered;
//   dW: ONE DW-layout GEMM with M = (tap, channel): a tile row reads the layer's input shifted by its tap's row
//       offset, against dY over all padded rows (split-K, fp32 atomics into the zeroed gradient range); db rides along;
//   d(input): ONE conv-mode GEMM of dY with the flipped kernel, gated by the input activation's ReLU (which
//       also zeroes the border rows: the activation is zero there).
static int conv_trunk_backward(dmvae_plan* p, hipStream_t s) {
    const int dt = p->cfg.dtype;
    hipError_t me = hipMemsetAsync(p->buf.grad + p->conv.front().w_off, 0, (size_t)(p->conv_param_end - p->conv.front().w_off) * 4, s);
    if (me != hipSuccess) { set_error("conv gradient memset: %s", hipGetErrorString(me)); return (int)me; }
    for (int i = (int)p->conv.size() - 1; i >= 0; --i) {
        const PConv& L = p->conv[i];
        const int M = p->Bp * L.P * L.P;
        const bool last = i + 1 == (int)p->conv.size();
        char* dact = rows0(p, L.o_dact, L.P, L.cout_ld);
        if (L.pool) {
            const int Po = (L.hw + 1) / 2 + 2;
            const void* dpool = last ? (const void*)WS(p, p->o_dflat) : (const void*)rows0(p, L.o_dpool, Po, L.cout_ld);
            TRY(maxpool2_bwd_relu_launch(s, dt, rows0(p, L.o_act, L.P, L.cout_ld), dpool, L.hw, L.cout_ld, p->Bp, dact, last ? 0 : 1));
        }
        dmvae_epilogue e;
        memset(&e, 0, sizeof(e));
        e.kind = DMVAE_EPI_ATOMIC_F32; e.ldo = L.cout_np; e.n_valid = L.cout;
        if (i == 0) {
            TRY(conv_first_dw_launch(s, dt, WS(p, p->o_x), p->Ip, L.hw, p->Bp, dact, L.cout_ld, L.cout, p->buf.grad + L.w_off, L.cout_np, p->buf.grad + L.b_off,
                                     reinterpret_cast<float*>(WS(p, p->o_c0part))));
            break;
        }
        const PConv& Lp = p->conv[i - 1];
        const char* in = Lp.pool ? rows0(p, Lp.o_pool, L.P, L.cin_ld) : rows0(p, Lp.o_act, L.P, L.cin_ld);
        e.out = p->buf.grad + L.w_off; e.out2 = p->buf.grad + L.b_off;
        const int sp = conv_dw_split(M, (L.kdim / 64) * (L.cout_np / 64));
        if (sp > 1) {
            // K slice y stores its partial product (and bias-gradient partial) into slab y; the slabs are then added in
            // ascending order: bit-reproducible, no float atomics (the atomic form drifts after a few Adam steps)
            float* slab = reinterpret_cast<float*>(WS(p, p->o_cslab));
            const int64_t wn = (int64_t)L.kdim * L.cout_np;
            e.kind = DMVAE_EPI_STORE_F32; e.out = slab; e.out2 = slab + (int64_t)sp * wn;
            GemmArgs g;
            TRY(gemm_checked(s, dt, DMVAE_GEMM_DW, L.kdim, L.cout_np, M, in, L.cin_ld, dact, L.cout_ld, &e, 1, &g, L.P, L.cin));
            g.k_split = M / sp; g.slab_stride = wn; g.slab_stride2 = L.cout_np;
            if (dt == DMVAE_BF16) TRY(gemm_bf16_dispatch(s, DMVAE_GEMM_DW, g, sp));
            else {
                ProfScope ps(s, "gemm_f32_dw", 2.0 * g.M * g.N * (double)g.K, 4.0 * ((double)g.M * g.K + (double)g.K * g.N + (double)g.M * g.N));
                TRY(gemm_f32_dispatch(s, DMVAE_GEMM_DW, g, sp));
            }
            TRY(slab_reduce_launch(s, slab, wn, sp, wn, p->buf.grad + L.w_off));
            TRY(slab_reduce_launch(s, slab + (int64_t)sp * wn, L.cout_np, sp, L.cout_np, p->buf.grad + L.b_off));
        } else
        TRY(gemm_checked(s, dt, DMVAE_GEMM_DW, L.kdim, L.cout_np, M, in, L.cin_ld, dact, L.cout_ld, &e, sp, nullptr, L.P, L.cin));
        if (L.kdim > 9 * L.cin) {   // rows past the ninth tap are padding: the GEMM filled them with a copy of tap 8
            me = hipMemsetAsync(p->buf.grad + L.w_off + (int64_t)9 * L.cin * L.cout_np, 0, (size_t)(L.kdim - 9 * L.cin) * L.cout_np * 4, s);
            if (me != hipSuccess) { set_error("conv gradient pad memset: %s", hipGetErrorString(me)); return (int)me; }
        }
        TRY(conv_wflip_launch(s, dt, Wp(p, L.w_off), L.cin, L.cin_np, L.cout, L.cout_np, WS(p, p->o_wt), L.ktdim));
        memset(&e, 0, sizeof(e));
        e.kind = DMVAE_EPI_RELU_MASK; e.ldo = L.cin_ld; e.aux0 = in; e.ld0 = L.cin_ld; e.n_valid = L.cin;
        e.out = Lp.pool ? rows0(p, Lp.o_dpool, L.P, L.cin_ld) : rows0(p, Lp.o_dact, L.P, L.cin_ld);
        TRY(gemm_checked(s, dt, DMVAE_GEMM_DX, M, L.cin_np, L.ktdim, dact, L.cout_ld, WS(p, p->o_wt), L.ktdim, &e, 1, nullptr, L.P, L.cout));
    }
    return 0;
}

// heads = false: stop behind the [z|c]-hidden layer (the step then runs the two head layers inside the latent launch: heads_latent.hip)
static int encode_impl(dmvae_plan* p, hipStream_t s, bool heads = true) {
    const void* in = p->conv.empty() ? XB(p) : WS(p, p->o_x);
    int64_t ld = p->Ip;
    int kd = p->Ip;
    if (!p->conv.empty()) {
        TRY(conv_trunk_forward(p, s, &in));
        ld = kd = p->flat;
    }
    for (size_t i = 0; i < p->enc.size(); ++i) {
        const PLayer& L = p->enc[i];
        TRY(fwd_dense(p, s, in, ld, kd, L, L.out_pad, 0, DMVAE_EPI_BIAS_RELU, WS(p, p->o_enc[i]), L.out_pad));
        in = WS(p, p->o_enc[i]); ld = L.out_pad; kd = L.out_pad;
    }
    if (p->vade)      // VaDE: [mean | log_var] straight off the trunk (base_models.py:501-507), no logits
        return fwd_dense(p, s, in, ld, kd, p->mv, 2 * p->Dp, 0, DMVAE_EPI_BIAS_F32, WS(p, p->o_mv), 2 * p->Dp);
    TRY(fwd_dense(p, s, in, ld, kd, p->zc, 2 * p->Hp, 0, DMVAE_EPI_BIAS_RELU, WS(p, p->o_hzc), 2 * p->Hp));
    if (!heads) return 0;
    // the two head layers [mean|log_var] = hz.Wmv and logits = hc.Wl are independent siblings:
    // bf16 issues them as one grouped grid (each alone is 64..128 workgroups)
    GemmArgs q[2];
    const bool grp = p->cfg.dtype == DMVAE_BF16;
    TRY(fwd_dense(p, s, WS(p, p->o_hzc), 2 * p->Hp, p->Hp, p->mv, 2 * p->Dp, 0, DMVAE_EPI_BIAS_F32, WS(p, p->o_mv), 2 * p->Dp, grp ? &q[0] : nullptr));
    TRY(fwd_dense(p, s, act_off(p, p->o_hzc, p->Hp), 2 * p->Hp, p->Hp, p->lg, p->Kp, 0, DMVAE_EPI_BIAS_F32, WS(p, p->o_lg), p->Kp, grp ? &q[1] : nullptr));
    if (grp) TRY(gemm_bf16_grouped(s, DMVAE_GEMM_FWD, q, 2));
    return 0;
}

static int decode_hidden(dmvae_plan* p, hipStream_t s) {
    const void* in = WS(p, p->o_Z);
    int64_t ld = p->Dp;
    int kd = p->Dp;
    for (size_t i = 0; i < p->dec.size(); ++i) {
        const PLayer& L = p->dec[i];
        TRY(fwd_dense(p, s, in, ld, kd, L, L.out_pad, 0, DMVAE_EPI_BIAS_RELU, WS(p, p->o_dec[i]), L.out_pad));
        in = WS(p, p->o_dec[i]); ld = L.out_pad; kd = L.out_pad;
    }
    return 0;
}

extern "C" int dmvae_plan_encode(dmvae_plan* p, void* stream, int n_valid) {
    DMVAE_REQUIRE(p && p->bound, "dmvae_plan_encode: plan not bound");
    (void)n_valid;
    return encode_impl(p, (hipStream_t)stream);
}

extern "C" int dmvae_plan_decode(dmvae_plan* p, void* stream, const float* Z, int64_t ldz, int n_valid) {
    DMVAE_REQUIRE(p && p->bound && Z, "dmvae_plan_decode: plan not bound / null Z");
    DMVAE_REQUIRE(n_valid >= 0 && n_valid <= p->cfg.max_batch, "dmvae_plan_decode: n_valid=%d exceeds max_batch", n_valid);
    hipStream_t s = (hipStream_t)stream;
    // Z (f32, caller) -> padded act buffer: a row gather with identity order
    void* za = p->cfg.dtype == DMVAE_BF16 ? (void*)WS(p, p->o_Z) : nullptr;
    float* zf = p->cfg.dtype == DMVAE_BF16 ? reinterpret_cast<float*>(WS(p, p->o_Zf)) : reinterpret_cast<float*>(WS(p, p->o_Z));
    DMVAE_REQUIRE(ldz == p->cfg.latent_dim, "dmvae_plan_decode: Z must be contiguous [n][latent_dim]");
    TRY(gather_launch(s, p->cfg.dtype, Z, n_valid, p->cfg.latent_dim, nullptr, 0, p->cfg.max_batch, n_valid, p->Bp, za, p->Dp, zf, p->Dp, p->Dp, nullptr));
    TRY(decode_hidden(p, s));
    const PLayer& L = p->out;
    const int last = (int)p->dec.size() - 1;
    const int kind = p->cfg.input_type == 0 ? DMVAE_EPI_BIAS_SIGMOID : DMVAE_EPI_BIAS_F32;
    return fwd_dense(p, s, WS(p, p->o_dec[last]), p->dec[last].out_pad, p->dec[last].out_pad, L, p->Ip, 0, kind, WS(p, p->o_recon), p->Ip);
}

// Split-K of the dW GEMMs: off.  Measured (tools/gemm_sweep.py) fp32-atomic split-K loses on every
// dW shape of the step except 512x512 (-13 %), so dW is a plain store and the whole step is
// bit-reproducible; cfg.deterministic is kept in the ABI for a future slab-reduce split-K.
static int dw_split(const dmvae_plan*, int, int) { return 1; }

static int launch_dw_queue(dmvae_plan* p, hipStream_t target, bool with_prior);

// dW = X^T dY into the grad arena, db = colsum(dY)
static int grad_dense(dmvae_plan* p, hipStream_t s, const void* X, int64_t ldx, int Mdim, const void* dY, int64_t ldy, int N,
                      int64_t w_off, int64_t ldw, int64_t b_off) {
    dmvae_epilogue e;
    memset(&e, 0, sizeof(e));
    const int split = dw_split(p, Mdim, N);
    const int nsl = p->cfg.dtype == DMVAE_BF16 ? p->dw_slices_now : 1;      // K slices into slabs (dw_slices_max)
    float* gbase = nsl > 1 ? reinterpret_cast<float*>(WS(p, p->o_dwslab)) : p->buf.grad;
    e.kind = nsl > 1 ? DMVAE_EPI_STORE_F32 : p->fused_update ? DMVAE_EPI_ADAM : (split > 1 ? DMVAE_EPI_ATOMIC_F32 : DMVAE_EPI_STORE_F32);
    e.out = gbase + w_off; e.ldo = ldw;
    e.out2 = gbase + b_off;             // db = column sums of dY, fused (ones-operand MFMA)
    if (p->cfg.dtype == DMVAE_BF16) {         // bf16: queued, all dW problems of the step go out as ONE grouped launch
        if ((int)p->dw_queue.size() >= DMVAE_MAX_GROUP) TRY(launch_dw_queue(p, s, false));    // deep stacks: a full group goes out early
        GemmArgs a;
        TRY(gemm_checked(s, DMVAE_BF16, DMVAE_GEMM_DW, Mdim, N, p->Bp, X, ldx, dY, ldy, &e, 1, &a));
        a.ws = reinterpret_cast<float*>(WS(p, p->o_cs)); a.ws_elems = p->cs_elems;     // bias-gradient slab sums of a 256x256-tile problem
        if (nsl > 1) { a.k_split = p->Bp / nsl; a.slab_stride = a.slab_stride2 = p->param_elems; }       // slice y -> slab y (gemm_bf16_body)
        if (nsl > 1 && p->dw_macro_now && gemm_bf16_256_slice_ok(Mdim, N, p->Bp / nsl)) {
            // this layer's slices run on the macro tile (gemm_bf16.hip peel_large_dw); the macro tile has no ones-operand pass, so the
            // bias gradient comes from a bias-only strip of the grouped small-tile launch: same K slices, same slabs
            GemmArgs b = a;
            b.M = 64; b.bias_only = 1;
            a.epi.out2 = nullptr;
            p->dw_queue.push_back(b);
        }
        auto cs = p->csum_of.find(dY);
        if (cs != p->csum_of.end() && cs->second.second == N) {      // ... unless dY's producer left its column sums
            a.csum_in = cs->second.first; a.csum_ld = N; a.csum_rows = p->Bp / 256;
        }
        p->dw_queue.push_back(a);
        return 0;
    }
    return gemm_checked(s, p->cfg.dtype, DMVAE_GEMM_DW, Mdim, N, p->Bp, X, ldx, dY, ldy, &e, split);
}

// Launch the queued weight-gradient problems as ONE grouped grid (with the Adam update in its
// epilogue under dmvae_plan_train_step).  Whole pass: everything queues until the last call (group 2) -- one launch;
// staged (dmvae_plan_forward_backward_stage, data parallel): every group is launched where it is flushed, so that its gradient
// bucket is final when the segment ends.  (The groups on a side stream beside the dX chain were measured twice and removed:
// 0.4235 vs 0.391 ms in round 1, 0.358 vs 0.292 ms in round 2 -- a fork / join pair in a replayed graph costs 15-20 us.)
static int flush_dw(dmvae_plan* p, hipStream_t s, int group) {
    if (p->dw_queue.empty()) return 0;
    if (!p->staged && group != 2) return 0;
    if (p->staged && p->stage_groups == 2 && group == 0) return 0;      // two launches: segment 0's problems go out with segment 1's
    return launch_dw_queue(p, s, group == 2);
}

// one grouped launch of whatever is queued; with_prior: also the prior tables' Adam (once per fused step)
static int launch_dw_queue(dmvae_plan* p, hipStream_t target, bool with_prior) {
    if (p->dw_queue.empty()) return 0;
    int rc;
    if (p->dw_slices_now > 1) {    // K slices into slabs, then their fixed-order sum: inside the Adam kernel (fused step) or into the gradient arena
        const float* slab = reinterpret_cast<const float*>(WS(p, p->o_dwslab));
        // the group's WEIGHT range and its BIAS range (the biases live in the arena tail, behind every weight matrix): kept apart,
        // because under the staged backward the span between them belongs to other segments -- buckets whose collective may
        // already be in flight (earlier segments) or whose gradients are not written yet (later ones)
        int64_t wlo = p->param_elems, whi = 0, blo = p->param_elems, bhi = 0;
        for (auto& a : p->dw_queue) {
            const int64_t w0 = reinterpret_cast<const float*>(a.epi.out) - slab;
            wlo = std::min(wlo, w0);
            whi = std::max(whi, w0 + (int64_t)a.M * a.epi.ldo);
            if (a.epi.out2) {
                const int64_t b0 = reinterpret_cast<const float*>(a.epi.out2) - slab;
                blo = std::min(blo, b0);
                bhi = std::max(bhi, b0 + (int64_t)a.N);
            }
        }
        const int64_t lo = std::min(wlo, blo), hi = std::max(whi, bhi);
        rc = gemm_bf16_grouped_dw(target, p->dw_queue.data(), (int)p->dw_queue.size());
        p->dw_queue.clear();
        if (rc) return rc;
        if (p->fused_update) {
            DMVAE_REQUIRE(with_prior && lo == 0, "dW slabs: the fused step flushes ONE group holding every layer");
            AdamArgs a;
            a.n = p->param_elems; a.p = p->buf.param; a.g = p->buf.grad; a.m = p->buf.m; a.v = p->buf.v;
            a.pb = reinterpret_cast<bf16_t*>(p->buf.param_bf16);
            a.lr = 0.f; a.b1 = p->cfg.beta1; a.b2 = p->cfg.beta2; a.eps = p->cfg.adam_eps; a.gscale = 1.f;
            a.zero_grad = 0; a.ieee = p->cfg.adam_ieee;
            a.t_host = ~0ull;                                   // t = state->adam_t, advanced by this step's step_finalize
            a.st = reinterpret_cast<const dmvae_state*>(p->buf.state);
            // [hi, param_elems): the prior tables (gradient complete in the arena, written by step_finalize) and the arena's zero tail
            return adam_slabs_launch(target, a, slab, p->dw_slices_now, p->param_elems, hi, p->param_elems);
        }
        if (!p->staged || bhi <= blo || blo <= whi)       // whole pass: every layer is in this group, the span between is only the tail's alignment pad (zeros)
            return slab_reduce_launch(target, slab + lo, hi - lo, p->dw_slices_now, p->param_elems, p->buf.grad + lo);
        // staged (data parallel): this segment's weights and this segment's biases, nothing in between (one reduce over
        // [lo, hi) rewrote the weight gradients of EARLIER segments -- whose reduce-scatter / all-reduce was already running -- with
        // local sums, and wrote stale slab values over biases that later segments had yet to produce)
        TRY(slab_reduce_launch(target, slab + wlo, whi - wlo, p->dw_slices_now, p->param_elems, p->buf.grad + wlo));
        return slab_reduce_launch(target, slab + blo, bhi - blo, p->dw_slices_now, p->param_elems, p->buf.grad + blo);
    }
    if (p->fused_update) {    // dmvae_plan_train_step: the Adam update rides in the epilogue of this launch
        dmvae_adam_ctx c;
        memset(&c, 0, sizeof(c));
        c.param = p->buf.param; c.grad = p->buf.grad; c.m = p->buf.m; c.v = p->buf.v; c.param_bf16 = p->buf.param_bf16;
        c.state = p->buf.state; c.beta1 = p->cfg.beta1; c.beta2 = p->cfg.beta2; c.epsilon = p->cfg.adam_eps; c.grad_scale = 1.f;
        c.store_grad = 0;
        c.ieee = p->cfg.adam_ieee;
        c.seg_off = p->prior_off;                       // prior tables: gradient written by step_finalize
        c.seg_n = with_prior ? ((2 * (int64_t)p->cfg.n_classes * p->cfg.latent_dim + 3) & ~(int64_t)3) : 0;
        rc = gemm_bf16_grouped_dw_adam(target, p->dw_queue.data(), (int)p->dw_queue.size(), c);
    } else {
        rc = gemm_bf16_grouped_dw(target, p->dw_queue.data(), (int)p->dw_queue.size());
    }
    p->dw_queue.clear();
    return rc;
}

// cs_off >= 0: where this dY tensor's column-sum partials live ([Bp/256][cs_ld] floats at workspace offset cs_off; this
// GEMM writes columns cs_col .. cs_col + N); taken only when the 256x256 kernel runs the GEMM, and then registered under
// cs_key (the dY tensor's base pointer) for the weight-gradient problem of that layer
static int dx_dense(dmvae_plan* p, hipStream_t s, const void* dY, int64_t ldy, int Kdim, int64_t w_off, int64_t ldw, int N,
                    const void* Yfwd, int64_t ldyf, void* out, int64_t ldo, GemmArgs* deferred = nullptr,
                    const void* cs_key = nullptr, int64_t cs_off = -1, int64_t cs_ld = 0, int64_t cs_col = 0, const GemmRiders* riders = nullptr) {
    dmvae_epilogue e;
    memset(&e, 0, sizeof(e));
    e.kind = DMVAE_EPI_RELU_MASK; e.out = out; e.ldo = ldo; e.aux0 = Yfwd; e.ld0 = ldyf;
    if (riders) {       // (the caller has asked gemm_bf16_riders_room: a small-tile launch)
        GemmArgs a;
        TRY(gemm_checked(s, p->cfg.dtype, DMVAE_GEMM_DX, p->Bp, N, Kdim, dY, ldy, Wp(p, w_off), ldw, &e, 1, &a));
        return gemm_bf16_dispatch(s, DMVAE_GEMM_DX, a, 1, riders);
    }
    if (!deferred && cs_off >= 0 && p->cfg.dtype == DMVAE_BF16 && gemm_bf16_256_ok(DMVAE_GEMM_DX, DMVAE_EPI_RELU_MASK, p->Bp, N, Kdim, false)) {
        GemmArgs a;
        TRY(gemm_checked(s, p->cfg.dtype, DMVAE_GEMM_DX, p->Bp, N, Kdim, dY, ldy, Wp(p, w_off), ldw, &e, 1, &a));
        float* part = reinterpret_cast<float*>(WS(p, cs_off));
        a.csum_out = part + cs_col; a.csum_ld = cs_ld;
        p->csum_of[cs_key] = std::make_pair((const float*)part, cs_ld);
        return gemm_bf16_dispatch(s, DMVAE_GEMM_DX, a, 1);
    }
    return gemm_plan(p, s, DMVAE_GEMM_DX, N, Kdim, dY, ldy, Wp(p, w_off), ldw, &e, deferred);
}

// stage < 0: the whole forward + backward.  stage 0 / 1 / 2: the three segments after which one
// contiguous BUCKET of the gradient arena is complete (dmvae_plan_grad_buckets): 0 = forward, loss,
// decoder backward -> [dec0 .. out, prior tables]; 1 = heads -> [zh|ch, mean|log_var, logits];
// 2 = trunk -> [enc*].  Each segment launches its own dW group, so a data-parallel caller can start
// the bucket's all-reduce while the next segment runs.
static int forward_backward_impl(dmvae_plan* p, void* stream, int n_valid, const float* eps, int64_t ld_eps,
                                 const float* gumbel, int64_t ld_gumbel, float inv_B, int stage) {
    DMVAE_REQUIRE(p && p->bound, "dmvae_plan_forward_backward: plan not bound");
    DMVAE_REQUIRE(n_valid > 0 && n_valid <= p->cfg.max_batch, "dmvae_plan_forward_backward: n_valid=%d out of range", n_valid);
    DMVAE_REQUIRE(stage >= -1 && stage <= 2, "dmvae_plan_forward_backward_stage: stage %d (0, 1, 2)", stage);
    hipStream_t s = (hipStream_t)stream;
    const dmvae_config& c = p->cfg;
    const int dt = c.dtype;
    const bool all = stage < 0;
    p->staged = !all;
    if (all || stage == 0) {      // K slices of this pass's dW groups (dw_slices_max).  MEASURED (tools/knob_step.py <cfg> 10 1 2 4, same box):
        // cfg3 (16384 rows) 1.0065 / 0.9778 / 0.9855 ms per step for 1 / 2 / 4 slices; cfg4 (8192 rows) 0.6623 / 0.7003 / 0.7126 -- the
        // slabs' extra 8 B per parameter and the separate Adam pass pay only once a tile's K loop is > 128 K tiles: two slices from 16384 rows
        // With the 256-divisible layers (three quarters of the flops of the MNIST-shaped stacks: [z|c]-hidden, 2048 -> 512, 512 -> 512)
        // on the macro tile: four slices make 4 x 56 = 224 macro workgroups -- one round of the chip at 1.3-1.4 PFLOP/s against the
        // small tiles' 0.74 in this layout (both operands through the transposing LDS read).
        // MEASURED (same box, tools/knob_step.py <cfg> 11 0 1): cfg3 (16384 rows) 0.9565 (small tiles, two slices) -> 0.9388 ms per step
        // (macro launch 98 us at 1.40 PFLOP/s + the small-tile remainder 120 us + the Adam pass 40 us); cfg4 (8192 rows) 0.6474
        // (unsliced, fused Adam) vs 0.7248: a 2048-deep slice does not carry the slabs' extra pass -- from 16384 rows only.
        const bool macro = g_dw_macro && p->dw_slices_max >= 4 && 4 * p->dw_macro_tiles >= 160 && p->Bp >= 16384 && gemm_bf16_256_slice_ok(256, 256, p->Bp / 4);
        const int want = g_dw_slices > 0 ? g_dw_slices : macro ? 4 : (p->Bp >= 16384 ? 2 : 1);
        p->dw_slices_now = (p->dw_slices_max > 1 && want > 1 && p->Bp % (want * 64) == 0) ? std::min(want, p->dw_slices_max) : 1;
        p->dw_macro_now = g_dw_macro && p->dw_slices_now > 1;
    }
    const int nd = (int)p->dec.size(), ne = (int)p->enc.size();
    auto cso = [](const std::vector<int64_t>& v, int i) -> int64_t { return v.empty() ? -1 : v[i]; };      // column-sum partials of a dY (bf16, Bp % 256 == 0)
    // Loss scalars, Adam t / lr_t and the prior-table gradients need only the forward partials.  Whole pass, bf16:
    // the 1 + 80 blocks of step_finalize ride as extra workgroups of the grouped heads-dX launch further down (one
    // kernel boundary less); staged (data parallel: the prior tables belong to this segment's bucket) and f32: a
    // launch of their own, here.  MEASURED earlier: as a side BRANCH of the graph (fork / join events) the step
    // took 0.344 ms against 0.326 ms -- a second branch costs more than the launch it hides.
    const int KD2 = 2 * c.n_classes * c.latent_dim;
    // heads forward + latent stage as one launch where it applies (bf16, <= 4096 rows, Dp <= 128, K * D < 4096): blocks of 16 rows
    const bool hl_fused = dt == DMVAE_BF16 && !p->vade && heads_latent_ok(p->Bp, c.latent_dim, c.n_classes, p->Dp, p->Kp, p->Hp, c.mode, 16) && p->mv.ldw >= 2 * p->Dp && p->lg.ldw >= p->Kp;
    const int n_lblk = p->vade ? p->n_lblk : hl_fused ? p->Bp / 16 : latent_nblocks(p->Bp, c.latent_dim, c.n_classes);      // under the knob's CURRENT value: what latent_launch will use
    const int n_pblk = p->lws_bytes ? 1 : n_lblk;
    DMVAE_REQUIRE(n_lblk <= p->n_lblk_cap, "dmvae_plan_forward_backward: the latent kernel would run %d blocks, the plan's partial-sum buffers hold %d", n_lblk, p->n_lblk_cap);
    const dmvae_finalize_args fin = step_finalize_args(reinterpret_cast<float*>(WS(p, p->o_rpart)), p->n_rpart, reinterpret_cast<float*>(WS(p, p->o_lpart)),
                                                       n_lblk, inv_B, p->buf.state, 1, c.beta1, c.beta2,
                                                       reinterpret_cast<float*>(WS(p, p->o_dprior)), n_pblk, KD2, p->buf.grad + p->prior_off);
    // (wide heads -- the 4096-wide configuration -- take the 256x256 macro-tile kernel one by one instead of the grouped grid)
    const bool heads_big = dt == DMVAE_BF16 && gemm_bf16_256_ok(DMVAE_GEMM_DX, DMVAE_EPI_RELU_MASK, p->Bp, p->Hp, 2 * p->Dp, false) &&
                           gemm_bf16_256_ok(DMVAE_GEMM_DX, DMVAE_EPI_RELU_MASK, p->Bp, p->Hp, p->Kp, false);
    const bool fin_rides = all && dt == DMVAE_BF16 && !heads_big && !p->vade && g_fin_rides;
    // ... and WHERE they ride.  In the heads' dX launch (every slot of the chip taken by a real workgroup) they cost that launch 2.2 us at 4096 rows
    // (MEASURED, knob 16: 19.4 us with them, 17.2 without; as a launch of their own +4.3 us per step).  The dZ GEMM two launches earlier has
    // their inputs too (it runs behind the reconstruction layer) and, while tiles + riders <= 256, leaves every rider a CU of its own.
    bool fin_in_dz = false, fin_in_out = false;
    const int fin_lead = (fin.nblocks + 7) & ~7;
    GemmArgs dz_shape{}, out_shape{};           // (shapes only: what gemm_bf16_riders_room looks at)
    dz_shape.M = p->Bp; dz_shape.N = p->Dp; dz_shape.K = p->dec.empty() ? 64 : p->dec[0].out_pad; dz_shape.epi.kind = DMVAE_EPI_LATENT; dz_shape.k_split = dz_shape.K;
    out_shape.M = p->Bp; out_shape.N = p->dec.empty() ? 64 : p->dec.back().out_pad; out_shape.K = p->Ip; out_shape.epi.kind = DMVAE_EPI_RELU_MASK;
    const int dz_room = (dt == DMVAE_BF16 && !p->dec.empty()) ? gemm_bf16_riders_room(dz_shape, true) : -1;
    const int out_room = (dt == DMVAE_BF16 && !p->dec.empty()) ? gemm_bf16_riders_room(out_shape, false) : -1;
    // knob 16 = 3: step_finalize as second workgroups of the output layer's dX launch whenever it has the room (free there: 0.2717 vs 0.2721 ms
    // against the dZ placement) -- where it also goes by default when the dZ launch runs on its two-wave 16-row tiles, which cannot carry it
    if (fin_rides && g_fin_rides == 3 && out_room >= fin_lead) fin_in_out = true;
    if (fin_rides && g_fin_rides == 1) fin_in_dz = dz_room >= fin_lead && gemm_bf16_carries_finalize(dz_shape);
    if (fin_rides && g_fin_rides == 1 && !fin_in_dz && out_room >= fin_lead) fin_in_out = true;      // (free there: 9.3 vs 9.2 us for that launch)
    // A prefetched batch (dmvae_plan_prefetch_batch): its gather takes the idle CUs of the dZ launch -- behind step_finalize, which has moved the
    // batch cursor on: those blocks then ride one launch EARLIER, as second workgroups of the output layer's dX launch (their inputs, the loss
    // partials, are complete behind the output layer's forward GEMM).  Without the room for both: step_finalize where it was, the gather as a
    // launch of its own in front of the trunk's backward pass.
    GemmRiders rid_dz, rid_out;
    if (p->pf.armed && (all || stage == 0)) {
        p->pf.gat = gather_args(DMVAE_BF16, p->pf.data, p->pf.n_rows, c.input_dim, p->pf.perm, p->pf.first, c.max_batch, p->pf.n_valid, p->Bp, XB(p, 1), p->Ip,
                                nullptr, p->Ip, p->Ip, p->pf.st);
        if (fin_rides && g_fin_rides == 1 && g_pf_rides && dz_room >= 64 && out_room >= fin_lead) {
            fin_in_dz = false; fin_in_out = true;
            rid_out.fin = fin; rid_out.nfin = fin_lead;
            rid_dz.gat = p->pf.gat;
            rid_dz.gat_last = g_pf_rides != 2;           // behind the tiles: the tiles then get CUs of their own first
            rid_dz.ngat = (g_pf_rides == 3 ? std::min(2 * dz_room, 512) : std::min(dz_room, 256)) & ~7;
        }
    }
    if (fin_in_dz) { rid_dz.fin = fin; rid_dz.nfin = fin_lead; }
    if (fin_in_out) { rid_out.fin = fin; rid_out.nfin = fin_lead; }
  if (all || stage == 0) {
    p->dw_queue.clear();
    p->csum_of.clear();
    TRY(encode_impl(p, s, !hl_fused));

    dmvae_latent_args la;
    memset(&la, 0, sizeof(la));
    la.B = n_valid; la.B_pad = p->Bp; la.D = c.latent_dim; la.K = c.n_classes; la.mode = p->vade ? 2 : c.mode; la.act_dtype = dt;
    la.kl_ratio = 1.f; la.temperature = c.temperature; la.inv_B = inv_B; la.seed = c.seed; la.noise_step = 0;
    la.mean = reinterpret_cast<float*>(WS(p, p->o_mv)); la.ld_mean = 2 * p->Dp;
    la.log_var = la.mean + p->Dp; la.ld_log_var = 2 * p->Dp;
    la.logits = reinterpret_cast<float*>(WS(p, p->o_lg)); la.ld_logits = p->Kp;
    la.eps = eps; la.ld_eps = ld_eps; la.gumbel = gumbel; la.ld_gumbel = ld_gumbel;
    la.prior_means = p->buf.param + p->prior_off;
    la.prior_log_vars = la.prior_means + (int64_t)c.n_classes * c.latent_dim;
    la.Z_act = WS(p, p->o_Z); la.ld_Z = p->Dp;
    la.Z_f32 = dt == DMVAE_BF16 ? reinterpret_cast<float*>(WS(p, p->o_Zf)) : nullptr; la.ld_Zf = p->Dp;
    la.weights = reinterpret_cast<float*>(WS(p, p->o_w)); la.ld_w = p->Kp;
    la.gmu = reinterpret_cast<float*>(WS(p, p->o_gmu)); la.glv = reinterpret_cast<float*>(WS(p, p->o_glv));
    la.clv = reinterpret_cast<float*>(WS(p, p->o_clv)); la.ld_g = p->Dp;
    la.dlogits_act = WS(p, p->o_dlg); la.ld_dl = p->Kp;
    la.dprior_partials = reinterpret_cast<float*>(WS(p, p->o_dprior));
    la.loss_partials = reinterpret_cast<float*>(WS(p, p->o_lpart));
    la.state = p->buf.state;
    la.mfma_ws = p->lws_bytes ? WS(p, p->o_lws) : nullptr; la.mfma_ws_bytes = p->lws_bytes;
    if (hl_fused) {
        dmvae_heads_args ha;
        memset(&ha, 0, sizeof(ha));
        ha.hz = WS(p, p->o_hzc); ha.lda = 2 * p->Hp; ha.Hp = p->Hp; ha.Dp = p->Dp; ha.Kp = p->Kp;
        ha.W_mv = Wp(p, p->mv.w_off); ha.ld_mv = p->mv.ldw; ha.W_lg = Wp(p, p->lg.w_off); ha.ld_lg = p->lg.ldw;
        ha.b_mv = p->buf.param + p->mv.b_off; ha.b_lg = p->buf.param + p->lg.b_off;
        // few blocks (a small batch): K slices, while blocks x slices stay within one round of the chip (the scheme of ksplit_for)
        if (g_ksplit && p->o_ksws >= 0) {
            int S = KSPLIT_MAX_SLICES;
            while (S > 1 && (p->Hp % (S * 64) || (p->Bp / 16) * S > 256 || heads_latent_kslice_floats(p->Bp, p->Dp, S) > p->ksws_elems)) S /= 2;
            if (S > 1) {
                ha.kslices = S; ha.kslice_ws = reinterpret_cast<float*>(WS(p, p->o_ksws)); ha.kslice_ws_floats = p->ksws_elems;
                ha.kslice_tick = reinterpret_cast<int32_t*>(WS(p, p->o_ktick));
            }
        }
        TRY(heads_latent_launch(s, &la, &ha));
    } else {
        TRY(latent_launch(s, &la));
    }

    TRY(decode_hidden(p, s));
    {   // output layer + reconstruction loss + dLoss/dlogits in one epilogue
        const PLayer& L = p->out;
        dmvae_epilogue e;
        memset(&e, 0, sizeof(e));
        e.kind = DMVAE_EPI_BIAS_RECON; e.m_valid = n_valid; e.n_valid = c.input_dim; e.recon_kind = c.input_type; e.scale = inv_B;
        e.out = WS(p, p->o_dl); e.ldo = p->Ip; e.bias = p->buf.param + L.b_off;
        e.aux0 = WS(p, p->o_xf); e.ld0 = p->Ip; e.partials = reinterpret_cast<float*>(WS(p, p->o_rpart));
        const int Kd = p->dec[nd - 1].out_pad;
        if (p->tsrc_valid) {      // dmvae_plan_load_batch_step: the targets come from the dataset through the permutation (gemm_bf16_body)
            e.recon_kind |= 0x100;
            e.aux0 = p->tsrc.data; e.ld0 = c.input_dim; e.aux1 = p->tsrc.perm; e.ld1 = p->tsrc.first;
            e.aux2 = p->tsrc.st; e.d_off = c.max_batch; e.ld2 = p->tsrc.n_rows;
            DMVAE_REQUIRE(p->tsrc.n_valid == n_valid, "dmvae_plan_forward_backward: n_valid=%d, but dmvae_plan_load_batch_step assembled %d rows", n_valid, p->tsrc.n_valid);
            DMVAE_REQUIRE(!p->tsrc_used, "dmvae_plan_forward_backward: the batch assembled by dmvae_plan_load_batch_step has been consumed by an earlier pass "
                                         "(it has no f32 copy and the batch cursor has advanced): load the batch again, or use dmvae_plan_load_batch");
            // (only a batch addressed by the DEVICE cursor goes stale: step_finalize advances the cursor.  With an explicit `first` the bf16
            //  batch and the targets are both addressed by it, and a second pass over the same batch -- gradient checks, repeat passes -- is valid)
            if (p->tsrc.st) p->tsrc_used = true;
        }
        if (dt == DMVAE_BF16 && p->o_cs_dl >= 0 && gemm_bf16_256_ok(DMVAE_GEMM_FWD, DMVAE_EPI_BIAS_RECON, p->Bp, p->Ip, Kd, false)) {
            GemmArgs a;      // the macro-tile kernel also leaves dLoss/dlogits' column sums = the output bias gradient
            TRY(gemm_checked(s, dt, DMVAE_GEMM_FWD, p->Bp, p->Ip, Kd, WS(p, p->o_dec[nd - 1]), Kd, Wp(p, L.w_off), L.ldw, &e, 1, &a));
            float* part = reinterpret_cast<float*>(WS(p, p->o_cs_dl));
            a.csum_out = part; a.csum_ld = p->Ip;
            p->csum_of[(const void*)WS(p, p->o_dl)] = std::make_pair((const float*)part, (int64_t)p->Ip);
            TRY(gemm_bf16_dispatch(s, DMVAE_GEMM_FWD, a, 1));
        } else
        TRY(gemm_checked(s, dt, DMVAE_GEMM_FWD, p->Bp, p->Ip, Kd, WS(p, p->o_dec[nd - 1]), Kd, Wp(p, L.w_off), L.ldw, &e, 1));
    }
    if (!fin_rides)
        TRY(step_finalize_launch(s, fin.rp, fin.nr, fin.lp, fin.nl, inv_B, p->buf.state, 1, c.beta1, c.beta2, fin.part, fin.nblk, fin.ncol, fin.gout));
    // ---- backward: decoder
    TRY(grad_dense(p, s, WS(p, p->o_dec[nd - 1]), p->dec[nd - 1].out_pad, p->dec[nd - 1].out_pad, WS(p, p->o_dl), p->Ip, p->Ip,
                   p->out.w_off, p->out.ldw, p->out.b_off));
    TRY(dx_dense(p, s, WS(p, p->o_dl), p->Ip, p->Ip, p->out.w_off, p->out.ldw, p->dec[nd - 1].out_pad,
                 WS(p, p->o_dec[nd - 1]), p->dec[nd - 1].out_pad, WS(p, p->o_ddec[nd - 1]), p->dec[nd - 1].out_pad, nullptr,
                 WS(p, p->o_ddec[nd - 1]), cso(p->o_cs_ddec, nd - 1), p->dec[nd - 1].out_pad, 0, fin_in_out ? &rid_out : nullptr));
    for (int i = nd - 1; i >= 0; --i) {
        const PLayer& L = p->dec[i];
        const void* xin = i > 0 ? WS(p, p->o_dec[i - 1]) : WS(p, p->o_Z);
        const int64_t ldx = i > 0 ? p->dec[i - 1].out_pad : p->Dp;
        TRY(grad_dense(p, s, xin, ldx, L.in_pad, WS(p, p->o_ddec[i]), L.out_pad, L.out_pad, L.w_off, L.ldw, L.b_off));
        if (i > 0) {
            TRY(dx_dense(p, s, WS(p, p->o_ddec[i]), L.out_pad, L.out_pad, L.w_off, L.ldw, L.in_pad,
                         WS(p, p->o_dec[i - 1]), p->dec[i - 1].out_pad, WS(p, p->o_ddec[i - 1]), p->dec[i - 1].out_pad, nullptr,
                         WS(p, p->o_ddec[i - 1]), cso(p->o_cs_ddec, i - 1), p->dec[i - 1].out_pad));
        } else {   // dZ -> [dmean | dlog_var] through the reparameterisation + KL gradients
            dmvae_epilogue e;
            memset(&e, 0, sizeof(e));
            e.kind = DMVAE_EPI_LATENT; e.out = WS(p, p->o_dmv); e.ldo = 2 * p->Dp; e.d_off = p->Dp;
            e.aux0 = la.gmu; e.ld0 = p->Dp; e.aux1 = la.glv; e.ld1 = p->Dp; e.aux2 = la.clv; e.ld2 = p->Dp;
            const int S = ksplit_for(p, p->Dp, L.out_pad, DMVAE_EPI_LATENT);      // a small batch: K slices (GemmArgs::tick)
            if (rid_dz.nfin || rid_dz.ngat || S > 1) {      // the step_finalize blocks, or the next batch's gather, ride here (see above)
                GemmArgs ga;
                TRY(gemm_checked(s, dt, DMVAE_GEMM_DX, p->Bp, p->Dp, L.out_pad, WS(p, p->o_ddec[0]), L.out_pad, Wp(p, L.w_off), L.ldw, &e, 1, &ga));
                if (S > 1) {
                    ga.k_split = L.out_pad / S;
                    ga.ws = reinterpret_cast<float*>(WS(p, p->o_ksws)); ga.ws_elems = p->ksws_elems;
                    ga.tick = reinterpret_cast<int*>(WS(p, p->o_ktick));
                }
                TRY(gemm_bf16_dispatch(s, DMVAE_GEMM_DX, ga, S, (rid_dz.nfin || rid_dz.ngat) ? &rid_dz : nullptr));
                if (rid_dz.ngat) { p->pf.armed = false; p->pf.done = true; }
            } else
            TRY(gemm_checked(s, dt, DMVAE_GEMM_DX, p->Bp, p->Dp, L.out_pad, WS(p, p->o_ddec[0]), L.out_pad, Wp(p, L.w_off), L.ldw, &e, 1));
        }
    }
    TRY(flush_dw(p, s, 0));   // decoder dW group (launched here only when staged / DMVAE_DW_OVERLAP)
  }
  if ((all || stage == 1) && p->vade) {
    // ---- backward: VaDE's heads are the one linear layer [mean | log_var] off the trunk
    TRY(grad_dense(p, s, WS(p, p->o_enc[ne - 1]), p->Tp, p->Tp, WS(p, p->o_dmv), 2 * p->Dp, 2 * p->Dp, p->mv.w_off, p->mv.ldw, p->mv.b_off));
    TRY(dx_dense(p, s, WS(p, p->o_dmv), 2 * p->Dp, 2 * p->Dp, p->mv.w_off, p->mv.ldw, p->Tp, WS(p, p->o_enc[ne - 1]), p->Tp, WS(p, p->o_denc[ne - 1]), p->Tp,
                 nullptr, WS(p, p->o_denc[ne - 1]), cso(p->o_cs_denc, ne - 1), p->Tp));
    TRY(flush_dw(p, s, 1));
  } else if (all || stage == 1) {
    // ---- backward: heads
    TRY(grad_dense(p, s, WS(p, p->o_hzc), 2 * p->Hp, p->Hp, WS(p, p->o_dmv), 2 * p->Dp, 2 * p->Dp, p->mv.w_off, p->mv.ldw, p->mv.b_off));
    TRY(grad_dense(p, s, act_off(p, p->o_hzc, p->Hp), 2 * p->Hp, p->Hp, WS(p, p->o_dlg), p->Kp, p->Kp, p->lg.w_off, p->lg.ldw, p->lg.b_off));
    {   // d(z-hidden) and d(c-hidden): independent siblings writing the two halves of dhzc -> one grouped grid (bf16)
        GemmArgs q[2];
        const bool grp = dt == DMVAE_BF16 && !heads_big;
        TRY(dx_dense(p, s, WS(p, p->o_dmv), 2 * p->Dp, 2 * p->Dp, p->mv.w_off, p->mv.ldw, p->Hp, WS(p, p->o_hzc), 2 * p->Hp, WS(p, p->o_dhzc), 2 * p->Hp,
                     grp ? &q[0] : nullptr, WS(p, p->o_dhzc), p->o_cs_dhzc, 2 * p->Hp, 0));
        TRY(dx_dense(p, s, WS(p, p->o_dlg), p->Kp, p->Kp, p->lg.w_off, p->lg.ldw, p->Hp, act_off(p, p->o_hzc, p->Hp), 2 * p->Hp,
                     const_cast<void*>(act_off(p, p->o_dhzc, p->Hp)), 2 * p->Hp, grp ? &q[1] : nullptr, WS(p, p->o_dhzc), p->o_cs_dhzc, 2 * p->Hp, p->Hp));
        if (grp && g_heads_dx_form == 2) {       // z-hidden as its own launch, c-hidden (and the riding finalize blocks) as a group of one
            TRY(gemm_bf16_dispatch(s, DMVAE_GEMM_DX, q[0], 1));
            TRY(gemm_bf16_grouped(s, DMVAE_GEMM_DX, q + 1, 1, (fin_rides && !fin_in_dz && !fin_in_out) ? &fin : nullptr));
        } else
        if (grp) TRY(gemm_bf16_grouped(s, DMVAE_GEMM_DX, q, 2, (fin_rides && !fin_in_dz && !fin_in_out) ? &fin : nullptr));
    }
    // ---- backward: trunk
    TRY(grad_dense(p, s, WS(p, p->o_enc[ne - 1]), p->Tp, p->Tp, WS(p, p->o_dhzc), 2 * p->Hp, 2 * p->Hp, p->zc.w_off, p->zc.ldw, p->zc.b_off));
    TRY(flush_dw(p, s, 1));      // heads dW group ([mean|log_var], logits, [zh|ch]): launched here when staged
  }
  if (all || stage == 2) {
    // a prefetched batch that did not ride in the dZ launch: a gather of its own, here -- step_finalize has run (in every placement it has, an
    // earlier launch of this stream), so the device cursor is the next batch's
    if (p->pf.armed) {
        const dmvae_gather_args& g = p->pf.gat;
        TRY(gather_launch(s, DMVAE_BF16, g.data, g.n_rows, g.dim, g.perm, g.first, g.batch, g.n_valid, g.B_pad, g.out_act, g.ld_act, nullptr, g.ld_f32, g.cols_pad, g.st));
        p->pf.armed = false; p->pf.done = true;
    }
    if (!p->vade) {
        TRY(dx_dense(p, s, WS(p, p->o_dhzc), 2 * p->Hp, 2 * p->Hp, p->zc.w_off, p->zc.ldw, p->Tp, WS(p, p->o_enc[ne - 1]), p->Tp,
                     WS(p, p->o_denc[ne - 1]), p->Tp, nullptr, WS(p, p->o_denc[ne - 1]), cso(p->o_cs_denc, ne - 1), p->Tp));
    }
    for (int i = ne - 1; i >= 0; --i) {
        const PLayer& L = p->enc[i];
        const bool cnn = !p->conv.empty();
        const void* flat = cnn ? (p->conv.back().pool ? WS(p, p->conv.back().o_pool) : WS(p, p->conv.back().o_act)) : nullptr;
        const void* xin = i > 0 ? WS(p, p->o_enc[i - 1]) : (cnn ? flat : XB(p));
        const int64_t ldx = i > 0 ? p->enc[i - 1].out_pad : (cnn ? p->flat : p->Ip);
        TRY(grad_dense(p, s, xin, ldx, L.in_pad, WS(p, p->o_denc[i]), L.out_pad, L.out_pad, L.w_off, L.ldw, L.b_off));
        if (i > 0)
            TRY(dx_dense(p, s, WS(p, p->o_denc[i]), L.out_pad, L.out_pad, L.w_off, L.ldw, L.in_pad,
                         WS(p, p->o_enc[i - 1]), p->enc[i - 1].out_pad, WS(p, p->o_denc[i - 1]), p->enc[i - 1].out_pad, nullptr,
                         WS(p, p->o_denc[i - 1]), cso(p->o_cs_denc, i - 1), p->enc[i - 1].out_pad));
        else if (cnn)   // gradient of the flattened pool output (gate: pool output > 0, see conv_trunk_backward)
            TRY(dx_dense(p, s, WS(p, p->o_denc[0]), L.out_pad, L.out_pad, L.w_off, L.ldw, L.in_pad, flat, p->flat, WS(p, p->o_dflat), p->flat));
    }
    TRY(flush_dw(p, s, 2));   // trunk dW group (everything queued so far when not staged)
    if (!p->conv.empty()) TRY(conv_trunk_backward(p, s));
  }
    return 0;
}

extern "C" int dmvae_plan_forward_backward(dmvae_plan* p, void* stream, int n_valid, const float* eps, int64_t ld_eps,
                                           const float* gumbel, int64_t ld_gumbel, float inv_B) {
    return forward_backward_impl(p, stream, n_valid, eps, ld_eps, gumbel, ld_gumbel, inv_B, -1);
}
extern "C" int dmvae_plan_forward_backward_stage(dmvae_plan* p, void* stream, int stage, int n_valid, const float* eps, int64_t ld_eps,
                                                 const float* gumbel, int64_t ld_gumbel, float inv_B) {
    DMVAE_REQUIRE(stage >= 0 && stage <= 2, "dmvae_plan_forward_backward_stage: stage %d (0, 1, 2)", stage);
    return forward_backward_impl(p, stream, n_valid, eps, ld_eps, gumbel, ld_gumbel, inv_B, stage);
}
extern "C" int dmvae_plan_set_stage_groups(dmvae_plan* p, int n_groups) {
    DMVAE_REQUIRE(p && (n_groups == 2 || n_groups == 3), "dmvae_plan_set_stage_groups: 2 or 3 weight-gradient launches per staged backward");
    p->stage_groups = n_groups;
    return 0;
}
extern "C" int dmvae_plan_grad_buckets(const dmvae_plan* p, int64_t bounds[5]) {
    DMVAE_REQUIRE(p && bounds, "dmvae_plan_grad_buckets: null pointer");
    bounds[0] = 0;                                   // stage 2 completes [bounds[0], bounds[1])  (trunk weights)
    bounds[1] = p->vade ? p->mv.w_off : p->zc.w_off; // stage 1 completes [bounds[1], bounds[2])  (heads weights)
    bounds[2] = p->dec.empty() ? p->out.w_off : p->dec[0].w_off;   // stage 0 completes [bounds[2], bounds[3])  (decoder weights)
    bounds[3] = p->tail_off;                         // [bounds[3], bounds[4]): the tail -- every bias, the prior tables: complete
    bounds[4] = p->param_elems;                      // when the LAST segment has run (each segment writes its layers' biases)
    return 0;
}

extern "C" int dmvae_plan_train_step(dmvae_plan* p, void* stream, int n_valid, const float* eps, int64_t ld_eps,
                                     const float* gumbel, int64_t ld_gumbel, float inv_B) {
    DMVAE_REQUIRE(p && p->bound, "dmvae_plan_train_step: plan not bound");
    if (p->cfg.dtype != DMVAE_BF16 || !p->conv.empty()) {     // f32, or conv gradients (not in the dW group): the plain pair
        TRY(dmvae_plan_forward_backward(p, stream, n_valid, eps, ld_eps, gumbel, ld_gumbel, inv_B));
        return dmvae_plan_update(p, stream, 1.f);
    }
    p->fused_update = true;               // dW problems are queued with DMVAE_EPI_ADAM; flush_dw launches them with the Adam context
    const int rc = dmvae_plan_forward_backward(p, stream, n_valid, eps, ld_eps, gumbel, ld_gumbel, inv_B);
    p->fused_update = false;
    return rc;
}

extern "C" int dmvae_plan_update(dmvae_plan* p, void* stream, float grad_scale) {
    DMVAE_REQUIRE(p && p->bound, "dmvae_plan_update: plan not bound");
    return dmvae_plan_update_range(p, stream, grad_scale, 0, p->param_elems);
}
extern "C" int dmvae_plan_update_range(dmvae_plan* p, void* stream, float grad_scale, int64_t lo, int64_t hi) {
    DMVAE_REQUIRE(p && p->bound, "dmvae_plan_update_range: plan not bound");
    DMVAE_REQUIRE(lo >= 0 && hi > lo && hi <= p->buf.arena_elems && lo % 4 == 0 && hi % 4 == 0,
                  "dmvae_plan_update_range: [%lld, %lld) must be a 4-aligned range of the %lld-element arena", (long long)lo, (long long)hi, (long long)p->buf.arena_elems);
    hipStream_t s = (hipStream_t)stream;
    AdamArgs a;
    a.n = hi - lo; a.p = p->buf.param + lo; a.g = p->buf.grad + lo; a.m = p->buf.m + lo; a.v = p->buf.v + lo;
    a.pb = p->cfg.dtype == DMVAE_BF16 ? reinterpret_cast<bf16_t*>(p->buf.param_bf16) + lo : nullptr;
    a.lr = 0.f; a.b1 = p->cfg.beta1; a.b2 = p->cfg.beta2; a.eps = p->cfg.adam_eps; a.gscale = grad_scale;
    a.zero_grad = 0;    // every gradient element is overwritten each step (no atomic accumulation)
    a.ieee = p->cfg.adam_ieee;
    a.t_host = ~0ull;    // t = state->adam_t, already advanced by this step's loss_finalize (saves a launch)
    a.st = reinterpret_cast<const dmvae_state*>(p->buf.state);
    return adam_launch(s, a);
}

extern "C" int dmvae_plan_view(const dmvae_plan* p, const char* name, void** ptr, int64_t* ld, int32_t* dtype) {
    DMVAE_REQUIRE(p && p->bound && name && ptr && ld && dtype, "dmvae_plan_view: null argument / plan not bound");
    const std::string n(name);
    char* base = reinterpret_cast<char*>(p->buf.work);
    *dtype = DMVAE_F32;
    if (n == "mean") { *ptr = base + p->o_mv; *ld = 2 * p->Dp; }
    else if (n == "log_var") { *ptr = base + p->o_mv + (int64_t)p->Dp * 4; *ld = 2 * p->Dp; }
    else if (n == "logits") { *ptr = base + p->o_lg; *ld = p->Kp; }
    else if (n == "weights") { *ptr = base + p->o_w; *ld = p->Kp; }
    else if (n == "recon") { *ptr = base + p->o_recon; *ld = p->Ip; }
    else if (n == "x") { *ptr = base + p->o_xf; *ld = p->Ip; }
    else if (n == "Z") { *ptr = base + (p->cfg.dtype == DMVAE_BF16 ? p->o_Zf : p->o_Z); *ld = p->Dp; }
    else if (n == "dxlogits") { *ptr = base + p->o_dl; *ld = p->Ip; *dtype = p->cfg.dtype; }
    else if (n == "hzc" && !p->vade) { *ptr = base + p->o_hzc; *ld = 2 * p->Hp; *dtype = p->cfg.dtype; }   // [z-hidden | c-hidden], halves Hp apart
    else if (n.rfind("enc", 0) == 0 && n.size() == 4 && n[3] - '0' < (int)p->enc.size()) {
        const int i = n[3] - '0';
        *ptr = base + p->o_enc[i]; *ld = p->enc[i].out_pad; *dtype = p->cfg.dtype;
    } else if (n.rfind("dec", 0) == 0 && n.size() == 4 && n[3] - '0' < (int)p->dec.size()) {
        const int i = n[3] - '0';
        *ptr = base + p->o_dec[i]; *ld = p->dec[i].out_pad; *dtype = p->cfg.dtype;
    }
    else if (n.rfind("conv", 0) == 0 && n.size() == 5 && n[4] - '0' >= 0 && n[4] - '0' < (int)p->conv.size()) {
        const PConv& L = p->conv[n[4] - '0'];       // zero-bordered [B][hw+2][hw+2][cout_ld] relu outputs, at padded pixel 0
        *ptr = rows0(p, L.o_act, L.P, L.cout_ld); *ld = L.cout_ld; *dtype = p->cfg.dtype;
    } else if (n == "flat" && !p->conv.empty()) { *ptr = base + p->conv.back().o_pool; *ld = p->flat; *dtype = p->cfg.dtype; }
    else { set_error("dmvae_plan_view: unknown view '%s'", name); return DMVAE_EINVAL; }
    return 0;
}

// ====================================================================== thin C wrappers
extern "C" int dmvae_gemm(void* stream, int dtype, int layout, int M, int N, int K, const void* A, int64_t lda,
                          const void* B, int64_t ldb, const dmvae_epilogue* epi, int split_k) {
    DMVAE_REQUIRE(!epi || (epi->recon_kind == 0 || epi->recon_kind == 1), "dmvae_gemm: recon_kind %d (0 binary, 1 real)", epi ? epi->recon_kind : 0);
    return gemm_checked((hipStream_t)stream, dtype, layout, M, N, K, A, lda, B, ldb, epi, split_k);
}
extern "C" int dmvae_gemm_partials(int dtype, int M, int N) { return gemm_partials(dtype, M, N); }

extern "C" int dmvae_gemm_grouped(void* stream, int dtype, int layout, const dmvae_gemm_problem* probs, int n) {
    DMVAE_REQUIRE(probs && n >= 1 && n <= DMVAE_MAX_GROUP, "dmvae_gemm_grouped: 1..%d problems", DMVAE_MAX_GROUP);
    hipStream_t s = (hipStream_t)stream;
    std::vector<GemmArgs> q(n);
    for (int i = 0; i < n; ++i) {
        const dmvae_gemm_problem& pr = probs[i];
        if (dtype == DMVAE_BF16) TRY(gemm_checked(s, dtype, layout, pr.M, pr.N, pr.K, pr.A, pr.lda, pr.B, pr.ldb, &pr.epi, 1, &q[i]));
        else TRY(gemm_checked(s, dtype, layout, pr.M, pr.N, pr.K, pr.A, pr.lda, pr.B, pr.ldb, &pr.epi, 1));   // f32: one launch each
    }
    return dtype == DMVAE_BF16 ? gemm_bf16_grouped(s, layout, q.data(), n) : 0;
}
extern "C" int dmvae_gemm_grouped_dw(void* stream, int dtype, const dmvae_gemm_problem* probs, int n) {
    DMVAE_REQUIRE(probs && n >= 1, "dmvae_gemm_grouped_dw: no problems");
    for (int i = 0; i < n; ++i)
        DMVAE_REQUIRE(probs[i].epi.kind == DMVAE_EPI_STORE_F32, "dmvae_gemm_grouped_dw: problems must use DMVAE_EPI_STORE_F32");
    return dmvae_gemm_grouped(stream, dtype, DMVAE_GEMM_DW, probs, n);
}
extern "C" int dmvae_gemm_grouped_dw_adam(void* stream, const dmvae_gemm_problem* probs, int n, const dmvae_adam_ctx* ctx) {
    DMVAE_REQUIRE(probs && ctx && n >= 1 && n <= DMVAE_MAX_GROUP, "dmvae_gemm_grouped_dw_adam: 1..%d problems and a context", DMVAE_MAX_GROUP);
    hipStream_t s = (hipStream_t)stream;
    std::vector<GemmArgs> q(n);
    for (int i = 0; i < n; ++i) {
        const dmvae_gemm_problem& pr = probs[i];
        TRY(gemm_checked(s, DMVAE_BF16, DMVAE_GEMM_DW, pr.M, pr.N, pr.K, pr.A, pr.lda, pr.B, pr.ldb, &pr.epi, 1, &q[i]));
    }
    return gemm_bf16_grouped_dw_adam(s, q.data(), n, *ctx);
}

extern "C" int dmvae_latent_nblocks(int B_pad, int D, int K) { return latent_nblocks(B_pad, D, K); }
extern "C" int dmvae_latent_nblocks_vade(int B_pad) { return latent_vade_nblocks(B_pad); }
extern "C" int64_t dmvae_latent_ws_bytes(int B_pad, int D, int K, int mode) {
    return (B_pad > 0 && B_pad % 64 == 0 && latent_mfma_applies(D, K, mode)) ? latent_mfma_ws_bytes(B_pad, D, K) : 0;
}
extern "C" int dmvae_latent_fwd(void* stream, const dmvae_latent_args* a) {
    DMVAE_REQUIRE(a && a->mode >= 0 && a->mode <= 2, "dmvae_latent_fwd: bad mode %d", a ? a->mode : -1);
    DMVAE_REQUIRE(a->mean && a->log_var && (a->logits || a->mode == 2) && a->prior_means && a->prior_log_vars && a->Z_act && a->gmu && a->glv &&
                  a->clv && (a->dlogits_act || a->mode == 2) && a->dprior_partials && a->loss_partials, "dmvae_latent_fwd: null pointer");
    DMVAE_REQUIRE(a->ld_Z >= a->D && (a->ld_dl >= a->K || a->mode == 2) && a->ld_g >= a->D, "dmvae_latent_fwd: leading dimension too small");
    return latent_launch((hipStream_t)stream, a);
}
extern "C" int64_t dmvae_heads_latent_kslice_floats(int B_pad, int Dp, int kslices) { return heads_latent_kslice_floats(B_pad, Dp, kslices); }
extern "C" int dmvae_heads_latent_ok(int B_pad, int D, int K, int Dp, int Kp, int Hp, int mode) { return heads_latent_ok(B_pad, D, K, Dp, Kp, Hp, mode, 16) ? 1 : 0; }
extern "C" int dmvae_heads_latent_fwd(void* stream, const dmvae_heads_args* h, const dmvae_latent_args* a) {
    DMVAE_REQUIRE(h && a && a->mode >= 0 && a->mode <= 1, "dmvae_heads_latent_fwd: bad arguments / mode");
    DMVAE_REQUIRE(h->hz && h->W_mv && h->W_lg && h->b_mv && h->b_lg && a->mean && a->log_var && a->logits && a->prior_means && a->prior_log_vars && a->Z_act && a->gmu &&
                  a->glv && a->clv && a->dlogits_act && a->dprior_partials && a->loss_partials, "dmvae_heads_latent_fwd: null pointer");
    DMVAE_REQUIRE(a->ld_Z >= a->D && a->ld_dl >= a->K && a->ld_g >= a->D, "dmvae_heads_latent_fwd: leading dimension too small");
    return heads_latent_launch((hipStream_t)stream, a, h);
}
extern "C" int dmvae_recon_nblocks(int B_pad, int I_pad) { return recon_nblocks(B_pad, I_pad); }
extern "C" int dmvae_recon_fwd_bwd(void* stream, int act_dtype, int recon_kind, int B, int B_pad, int I, int I_pad, const float* logits,
                                   int64_t ldl, const float* x, int64_t ldx, float inv_B, void* dl, int64_t ldd, float* partials) {
    DMVAE_REQUIRE(logits && x && dl && partials, "dmvae_recon_fwd_bwd: null pointer");
    DMVAE_REQUIRE(I_pad % 4 == 0 && ldl % 4 == 0 && ldx % 4 == 0 && ldd % 4 == 0 && B <= B_pad && I <= I_pad, "dmvae_recon_fwd_bwd: dims must be multiples of 4");
    return recon_launch((hipStream_t)stream, act_dtype, recon_kind, B, B_pad, I, I_pad, logits, ldl, x, ldx, inv_B, dl, ldd, partials);
}
extern "C" int dmvae_colsum(void* stream, int in_dtype, const void* in, int64_t ld, int M, int N, float* out) {
    DMVAE_REQUIRE(in && out && M > 0 && N > 0, "dmvae_colsum: bad argument");
    if (M > 64) TRY(colsum_prepare(N));   // two-pass path: lazily sized scratch (not capture-safe; plans bring their own)
    return colsum_launch((hipStream_t)stream, in_dtype, in, ld, M, N, out, nullptr, 0);
}
extern "C" int dmvae_loss_finalize(void* stream, const float* rp, int nr, const float* lp, int nl, float inv_B, void* state) {
    DMVAE_REQUIRE(rp && lp && state, "dmvae_loss_finalize: null pointer");
    return loss_finalize_launch((hipStream_t)stream, rp, nr, lp, nl, inv_B, state, 0);
}
extern "C" int dmvae_adam_tf(void* stream, int64_t n, float* param, float* grad, float* m, float* v, void* param_bf16, float lr,
                             float beta1, float beta2, float epsilon, float grad_scale, int flags, uint64_t t_host, const void* state) {
    DMVAE_REQUIRE(param && grad && m && v && n > 0 && n % 4 == 0, "dmvae_adam_tf: n must be a positive multiple of 4");
    DMVAE_REQUIRE(state || t_host >= 1, "dmvae_adam_tf: t starts at 1");
    AdamArgs a;
    a.n = n; a.p = param; a.g = grad; a.m = m; a.v = v; a.pb = reinterpret_cast<bf16_t*>(param_bf16);
    a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = epsilon; a.gscale = grad_scale; a.zero_grad = (flags & DMVAE_ADAM_ZERO_GRAD) != 0; a.ieee = (flags & DMVAE_ADAM_IEEE) != 0;
    a.t_host = t_host; a.st = reinterpret_cast<const dmvae_state*>(state);
    if (flags & DMVAE_ADAM_SHADOW) return adam_shadow_launch((hipStream_t)stream, a, (flags >> 8) & 0xffff);
    return adam_launch((hipStream_t)stream, a);
}
extern "C" int dmvae_adam_finish(void* stream, void* state) {
    DMVAE_REQUIRE(state, "dmvae_adam_finish: null state");
    return adam_finish_launch((hipStream_t)stream, state);
}
extern "C" int dmvae_gather_rows(void* stream, int act_dtype, const float* data, int64_t n_rows, int dim, const int32_t* perm,
                                 int64_t first, int batch, int n_valid, int B_pad, void* out_act, int64_t ld_act, float* out_f32,
                                 int64_t ld_f32, const void* state) {
    DMVAE_REQUIRE(data && (out_act || out_f32), "dmvae_gather_rows: null pointer");
    const int64_t ld = out_act ? ld_act : ld_f32;
    DMVAE_REQUIRE(ld % 4 == 0 && ld >= dim && (!out_act || !out_f32 || ld_act == ld_f32), "dmvae_gather_rows: ld must be a multiple of 4, >= dim, equal for both outputs");
    return gather_launch((hipStream_t)stream, act_dtype, data, n_rows, dim, perm, first, batch, n_valid, B_pad, out_act, ld_act, out_f32, ld_f32, (int)ld, state);
}
extern "C" int dmvae_philox_normal(void* stream, float* out, int64_t n, uint64_t seed, uint64_t step, uint32_t sid) {
    DMVAE_REQUIRE(out && n > 0, "dmvae_philox_normal: bad argument");
    return philox_launch((hipStream_t)stream, out, n, seed, step, sid, 0);
}
extern "C" int dmvae_philox_gumbel(void* stream, float* out, int64_t n, uint64_t seed, uint64_t step, uint32_t sid) {
    DMVAE_REQUIRE(out && n > 0, "dmvae_philox_gumbel: bad argument");
    return philox_launch((hipStream_t)stream, out, n, seed, step, sid, 1);
}
extern "C" int dmvae_cast_f32_to_bf16(void* stream, const float* in, void* out, int64_t n) {
    DMVAE_REQUIRE(in && out && n > 0, "dmvae_cast: bad argument");
    return cast_launch((hipStream_t)stream, in, out, n, 1);
}
extern "C" int dmvae_cast_bf16_to_f32(void* stream, const void* in, float* out, int64_t n) {
    DMVAE_REQUIRE(in && out && n > 0, "dmvae_cast: bad argument");
    return cast_launch((hipStream_t)stream, in, out, n, 0);
}

extern "C" int dmvae_debug_spin(void* stream, int microseconds) { return spin_launch((hipStream_t)stream, microseconds); }
extern "C" int dmvae_debug_strip_fwd2(void* stream, int B_pad, int K0, const void* X, int64_t ldx, const void* W0, int64_t ld0, const float* b0,
                                      const void* W1, int64_t ld1, const float* b1, void* Y1, int64_t ldy1, void* Y2, int64_t ldy2) {
    return strip_fwd2_launch((hipStream_t)stream, B_pad, K0, X, ldx, W0, ld0, b0, W1, ld1, b1, Y1, ldy1, Y2, ldy2);
}
extern "C" int dmvae_debug_stamps(void** device_ptr) {
    if (!device_ptr) return DMVAE_EINVAL;
    *device_ptr = gemm_bf16_stamps();
    if (!*device_ptr) *device_ptr = heads_dx_phase_table();      // measurement build 10: the phase table of heads_dx.hip (measure.h)
    return *device_ptr ? 0 : DMVAE_ESTATE;
}

extern "C" int dmvae_debug_anatomy(void** device_ptr) {
    if (!device_ptr) return DMVAE_EINVAL;
    *device_ptr = gemm_bf16_anatomy();
    return *device_ptr ? 0 : DMVAE_ESTATE;
}
extern "C" int dmvae_debug_anatomy256(void** device_ptr) {
    if (!device_ptr) return DMVAE_EINVAL;
    *device_ptr = gemm_bf16_256_anatomy();
    return *device_ptr ? 0 : DMVAE_ESTATE;
}

extern "C" int dmvae_prof_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof_on = on != 0;
    return 0;
}
extern "C" int dmvae_prof_collect(dmvae_prof_row* rows, int max_rows) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    std::map<std::string, dmvae_prof_row> agg;
    std::vector<std::string> order;
    for (auto& r : g_prof) {
        (void)hipEventSynchronize(r.e1);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, r.e0, r.e1);
        // the kernels' own durations: per dispatch, stop - start of the pair hipExtLaunchKernelGGL bound to it
        double kms = 0.0;
        bool kok = !r.k.empty();
        for (auto& pr : r.k) {
            float d = 0.f;
            if (hipEventSynchronize(pr.second) != hipSuccess || hipEventElapsedTime(&d, pr.first, pr.second) != hipSuccess || !(d > 0.f)) kok = false;
            kms += d;
            g_event_pool.push_back(pr.first);
            g_event_pool.push_back(pr.second);
        }
        (void)hipGetLastError();
        auto it = agg.find(r.name);
        if (it == agg.end()) {
            dmvae_prof_row row;
            memset(&row, 0, sizeof(row));
            snprintf(row.name, sizeof(row.name), "%s", r.name);
            it = agg.emplace(r.name, row).first;
            order.push_back(r.name);
        }
        it->second.launches += 1;
        it->second.total_ms += ms;
        it->second.kernel_ms += kok ? kms : 0.0;
        it->second.kernel_launches += kok ? (int64_t)r.k.size() : 0;
        it->second.flops += r.flops;
        it->second.bytes += r.bytes;
        g_event_pool.push_back(r.e0);
        g_event_pool.push_back(r.e1);
    }
    g_prof.clear();
    g_prof_cur = -1;
    int n = 0;
    for (auto& k : order) {
        if (n >= max_rows) break;
        rows[n++] = agg[k];
    }
    return n;
}

extern "C" int dmvae_debug_set_tile(int bm, int bn) {
    DMVAE_REQUIRE((bm == 0 && bn == 0) || ((bm == 64 || bm == 128) && (bn == 64 || bn == 128)), "dmvae_debug_set_tile: 64 or 128 (0,0 = heuristic)");
    gemm_bf16_force_tile(bm * 1000 + bn);
    return 0;
}

extern "C" int dmvae_debug_set_knob(int which, int value) {
    if (which == 10) { g_dw_slices = value; return 0; }
    if (which == 11) { g_dw_macro = value; return 0; }
    if (which == 12) { g_heads_dx_form = value; return 0; }
    if (which == 13) { heads_dx_stream_set(value); return 0; }
    if (which == 16) { g_fin_rides = value; return 0; }
    if (which == 17) { g_pf_rides = value; return 0; }
    if (which == 18 || which == 20) { gemm_bf16_set_knob(which, value); return 0; }
    if (which == 19) { heads_latent_set(value); return 0; }
    if (which == 21) { g_ksplit = value; return 0; }
    if (which == 14) { latent_set_blocks_target(value); return 0; }      // (the block count in use is taken at enqueue time and checked against the plan's capacity)
    DMVAE_REQUIRE(which >= 0 && which <= 9 && which != 3, "dmvae_debug_set_knob: knob 0 = supertile rows, 1 = 8-wave workgroups, 2 = per-problem tiles in grouped grids, (3: removed, the deep-ring policy), 4 = XCD runs per tile class in grouped grids, 5 = short-K conv tiles, 6 = 256x256 tile policy, 7 = short-K workgroups, 8 = first-tile stagger of the merged dW grid, 9 = waves per workgroup of a grouped dX launch; 10 / 11 = K slices of the dW groups, 12 = heads dX as one or two launches, 13 = heads dX on the streaming kernel (1) or the grouped tiles (0), 14 = blocks the latent kernel's geometry aims at (512), 19 = fused heads + latent launch, 20 = XCD partition of the grouped dW launch, 21 = K slices of a small batch's thin launches");
    gemm_bf16_set_knob(which, value);
    return 0;
}

extern "C" int dmvae_abi_version(void) { return DMVAE_ABI_VERSION; }
extern "C" const char* dmvae_last_error(void) { return g_err; }
