// Launchers implemented in the kernel translation units (host side, namespace dmvae).
#pragma once
#include "gemm_epilogue.h"

namespace dmvae {

struct AdamArgs {
    int64_t n;
    float* p; float* g; float* m; float* v; bf16_t* pb;
    float lr, b1, b2, eps, gscale;
    int zero_grad;
    int ieee;               // bf16 mode with the IEEE square root and division (dmvae_config.adam_ieee)
    uint64_t t_host;
    const dmvae_state* st;
};

#define DMVAE_MAX_GROUP 16
// riders: the step_finalize blocks and / or the gather of the next batch as extra workgroups of this launch (dense small-tile DX launches with the
// LATENT or RELU_MASK epilogue: gemm_bf16_dx_riders_kernel; ask gemm_bf16_riders_room / gemm_bf16_carries_finalize first)
int gemm_bf16_dispatch(hipStream_t s, int layout, const GemmArgs& a, int split, const GemmRiders* riders = nullptr);
int gemm_bf16_riders_room(const GemmArgs& a, bool own_cu);
bool gemm_bf16_carries_finalize(const GemmArgs& a);      // false: this launch's tile has fewer than four waves (step_finalize_block needs 256 threads)
int gemm_bf16_grouped_dw(hipStream_t s, const GemmArgs* probs, int nprob);
// fin: the step_finalize blocks ride as extra workgroups of this launch (DX / RELU_MASK groups only)
int gemm_bf16_grouped(hipStream_t s, int layout, const GemmArgs* probs, int nprob, const dmvae_finalize_args* fin = nullptr);
dmvae_finalize_args step_finalize_args(const float* rp, int nr, const float* lp, int nl, float inv_B, void* st, int bump_adam,
                                       float b1, float b2, const float* part, int nblk, int ncol, float* gout);
int gemm_bf16_tile_m(int M, int N, int split);
int gemm_auto_group_m(int tiles_m, int tiles_n, int bm, int bn, double run = 0.0);
// 256x256 macro-tile kernel (gemm_bf16_256.hip): takes the large dense problems, one workgroup per CU
bool gemm_bf16_256_ok(int layout, int epi, int M, int N, int K, bool conv);
int gemm_bf16_256_launch(hipStream_t s, int layout, const GemmArgs& a, const dmvae_adam_ctx* ctx = nullptr);
void gemm_bf16_256_set_policy(int v);
void gemm_bf16_256_set_stagger(int v);
bool gemm_bf16_256_slice_ok(int M, int N, int k_split);   // may K slices of this dW problem run on the macro tile (into slabs)?
bool gemm_bf16_256_rides();      // policy: may sub-chip problems ride in the merged dW grid (knob 6 >= 1, merging on)
int gemm_bf16_256_dw_all(hipStream_t s, const GemmArgs* probs, int n, const dmvae_adam_ctx* ctx);
// streaming form of the dX of the two head layers (heads_dx.hip): takes the problems of a grouped DX / RELU_MASK launch with K = 64 / 128 / 256
// (taken[i]) as one launch, the riding step_finalize blocks with them; the caller launches whatever is left as before
void* heads_dx_phase_table();      // measurement build 10 only (measure.h); nullptr otherwise
int heads_dx_stream_launch(hipStream_t s, const GemmArgs* probs, int nprob, const dmvae_finalize_args* fin, bool* taken);
void heads_dx_stream_set(int v);
void gemm_bf16_force_tile(int t);
void gemm_bf16_set_knob(int which, int v);
int gemm_f32_dispatch(hipStream_t s, int layout, const GemmArgs& a, int split);
// three STORE_F32 problems (forward / dX / dW layout, in that order) with split1 / split2 / split3 K slices as ONE grid (latent_mfma.hip)
int gemm_f32_trio(hipStream_t s, const GemmArgs& g1, int split1, const GemmArgs& g2, int split2, const GemmArgs& g3, int split3);
int latent_nblocks(int B_pad, int D, int K);
int latent_launch(hipStream_t s, const dmvae_latent_args* a);
void latent_set_blocks_target(int v);
// VaDE's latent stage (latent_vade.hip): mode 2
int latent_vade_nblocks(int B_pad);
int latent_vade_launch(hipStream_t s, const dmvae_latent_args* a);
// the two head layers' forward pass + the latent stage as one launch (heads_latent.hip); rows_per_latent_block: what latent_nblocks implies (must be 16)
bool heads_latent_ok(int B_pad, int D, int K, int Dp, int Kp, int Hp, int mode, int rows_per_latent_block);
int heads_latent_launch(hipStream_t s, const dmvae_latent_args* a, const dmvae_heads_args* h);
void heads_latent_set(int v);
int64_t heads_latent_kslice_floats(int B_pad, int Dp, int slices);
// measurement kernel (strip_fwd2.hip): two consecutive 512-wide dense layers as one row-strip kernel
int strip_fwd2_launch(hipStream_t s, int B_pad, int K0, const void* X, int64_t ldx, const void* W0, int64_t ld0, const float* b0,
                      const void* W1, int64_t ld1, const float* b1, void* Y1, int64_t ldy1, void* Y2, int64_t ldy2);
// MFMA form for large prior tables (latent_mfma.hip)
bool latent_mfma_applies(int D, int K, int mode);
int64_t latent_mfma_ws_bytes(int B_pad, int D, int K);
int latent_mfma_launch(hipStream_t s, const dmvae_latent_args* a, float* ws, int64_t ws_bytes);
int adam_launch(hipStream_t s, const AdamArgs& a);
// the same update shaped to run beside a macro-tile GEMM (<= 48 VGPRs, 32 KiB LDS ring filled by LDS-DMA, `blocks` workgroups of four waves; 0 = 256)
int adam_shadow_launch(hipStream_t s, const AdamArgs& a, int blocks);
int adam_finish_launch(hipStream_t s, void* st);
// TF-Adam on a gradient that arrives as nslab K-slice slabs (slab j at slabs + j * stride; same element offsets as the arenas);
// [seg_lo, seg_hi): elements whose gradient is complete in a.g instead (prior tables)
int adam_slabs_launch(hipStream_t s, const AdamArgs& a, const float* slabs, int nslab, int64_t stride, int64_t seg_lo, int64_t seg_hi);
int slab_reduce_launch(hipStream_t s, const float* slabs, int64_t n, int nslab, int64_t stride, float* out);
int colsum_prepare(int64_t max_n);
float* colsum_global_scratch(int64_t* elems);
int colsum_launch(hipStream_t s, int in_dtype, const void* in, int64_t ld, int M, int N, float* out, float* ws, int64_t ws_elems);
int recon_nblocks(int B_pad, int I_pad);
int recon_launch(hipStream_t s, int act_dtype, int recon_kind, int B, int B_pad, int I, int I_pad, const float* logits, int64_t ldl,
                 const float* x, int64_t ldx, float inv_B, void* dl, int64_t ldd, float* partials);
int loss_finalize_launch(hipStream_t s, const float* rp, int nr, const float* lp, int nl, float inv_B, void* st, int bump_adam);
int step_finalize_launch(hipStream_t s, const float* rp, int nr, const float* lp, int nl, float inv_B, void* st, int bump_adam,
                         float b1, float b2, const float* part, int nblk, int ncol, float* gout);
int gemm_bf16_grouped_dw_adam(hipStream_t s, const GemmArgs* probs, int nprob, const dmvae_adam_ctx& ctx);
dmvae_gather_args gather_args(int act_dtype, const float* data, int64_t n_rows, int dim, const int32_t* perm, int64_t first, int batch, int n_valid,
                              int B_pad, void* out_act, int64_t ld_act, float* out_f32, int64_t ld_f32, int cols_pad, const void* st);
int gather_launch(hipStream_t s, int act_dtype, const float* data, int64_t n_rows, int dim, const int32_t* perm, int64_t first, int batch,
                  int n_valid, int B_pad, void* out_act, int64_t ld_act, float* out_f32, int64_t ld_f32, int cols_pad, const void* st);
int philox_launch(hipStream_t s, float* out, int64_t n, uint64_t seed, uint64_t step, uint32_t sid, int gumbel);
int cast_launch(hipStream_t s, const void* in, void* out, int64_t n, int to_bf16);
int spin_launch(hipStream_t s, int us);
void* gemm_bf16_stamps();
void* gemm_bf16_anatomy();
void* gemm_bf16_256_anatomy();
// CNN trunk (conv.hip)
int conv_first_fwd_launch(hipStream_t s, int dtype, const void* x, int64_t bstride, int H, int64_t n_img, const void* W, int ldw,
                          const float* bias, int cout, void* out, int ld);
int conv_first_dw_blocks(int H, int64_t n_img);
int conv_first_dw_launch(hipStream_t s, int dtype, const void* x, int64_t bstride, int H, int64_t n_img, const void* dY, int ld, int cout,
                         float* dW, int ldw, float* db, float* part);
int zero_border_launch(hipStream_t s, int dtype, void* a, int P, int ld, int64_t n_img);
int maxpool2_fwd_launch(hipStream_t s, int dtype, const void* in, int H, int ld, int64_t n_img, void* out, int out_border);
int maxpool2_bwd_relu_launch(hipStream_t s, int dtype, const void* in, const void* dout, int H, int ld, int64_t n_img, void* din, int dout_border);
int conv_wflip_launch(hipStream_t s, int dtype, const void* W, int cin, int cin_ld, int cout, int ldw, void* Wt, int Kt);

}  // namespace dmvae
