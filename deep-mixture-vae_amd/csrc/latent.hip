// Fused latent kernel: softmax(logits) [base_models.py:249], Gaussian
// reparameterisation [priors.py:86-89], Gumbel-Softmax [priors.py:170-181],
// mixture-of-Gaussians KL exact / relaxed [priors.py:104-147], categorical KL
// [priors.py:183-201] AND every gradient of kl_ratio*(KL_C + KL_Z) -- none of
// them depends on the decoder, so the "backward" of this stage reduces to
//     dmean = dZ + gmu,   dlog_var = dZ * clv + glv
// which is fused into the epilogue of the GEMM that produces dZ
// (DMVAE_EPI_LATENT).  The [B,K,D] broadcast tensor TensorFlow materialises is
// never formed.
//
// Work decomposition (algorithmic bytes in DESIGN.md):
//   block = 256 threads = 4 waves, owns RB consecutive rows of the batch;
//   prior tables (mu_k, and exp(-logvar_k) or logvar_k) are staged in LDS in
//   D-chunks of DC columns (KL is separable in d); SIXTEEN lanes per row.
//   Everything a thread needs from global memory for its first row and the
//   first chunk is requested at the top of the kernel, before any dependent
//   work: one memory latency instead of four in sequence (at cfg2 the kernel is
//   pure latency: 9.7 MB in 18 us before).
//   Barriers are lds_barrier() (LDS traffic only): __syncthreads() also drains the
//   wave's global stores, and with five write-only output arrays per row that
//   was 2-3 us of write-acknowledgement wait at each of four barriers
//   (tools/latent_time.py with the DMVAE_ABLATE=7 phase stamps).
//   phase 1a: lane l of a row owns columns d = l, l+16, ...: Z, the reparam
//            coefficient, and the sums over k (dKL/dmean, dKL/dlog_var) -- a
//            loop over k with private accumulators, no cross-lane traffic;
//   phase 1b: lane l of a row owns clusters k = l, l+16, ...: the sums over d
//            (per-(row,k) KL terms), again private accumulators reading the
//            row's mu / e^lv from LDS -- this replaced a 16-lane shuffle
//            reduction per (row, k, chunk), which dominated at K*D = 50*256;
//   phase 2: threads over (k, d) pairs, loop over the block's rows held in
//            LDS -> per-block partial of the prior-table gradients, written
//            to [nblocks][2][K][D] and summed in a fixed order by
//            step_finalize (deterministic; no float atomics).
#include "latent_body.h"

namespace dmvae {

template <int MODE, int DSL>
__global__ __launch_bounds__(256) void latent_fwd_kernel(LatentLaunch L) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    latent_body<MODE, DSL, false>(L, (lds_f*)lds, nullptr, (int)blockIdx.x, LatentTile{nullptr, 0, 0, 0}, [] {}, [] { return true; });
}

static int g_latent_blocks = 512;          // tuning knob (dmvae_debug_set_knob 14): blocks the geometry aims at (rows per block = 16 .. 64, a power of two)
void latent_set_blocks_target(int v) { g_latent_blocks = v < 64 ? 512 : v; }
static void latent_geometry(int B_pad, int D, int K, int& RB, int& DC, int& nchunks, size_t& lds_bytes) {
    const int want = (B_pad + g_latent_blocks - 1) / g_latent_blocks;      // aim at >= 256..512 blocks
    RB = 16;                                    // power of two in [16, 64]: divides B_pad (multiple of 64)
    while (RB < want && RB < 64) RB *= 2;
    DC = 256;
    while (DC > 16 && DC / 2 >= D) DC /= 2;     // no wider than D rounded up to a power of two >= 16
    auto bytes = [&](int dc) { return latent_lds_bytes(K, RB, dc); };
    while (DC > 16 && bytes(DC) > 60 * 1024) DC /= 2;
    nchunks = (D + DC - 1) / DC;
    lds_bytes = bytes(DC);
}

int latent_nblocks(int B_pad, int D, int K) {
    int RB, DC, nc; size_t lb;
    latent_geometry(B_pad, D, K, RB, DC, nc, lb);
    return (B_pad + RB - 1) / RB;
}

template <int MODE, int DSL>
static void latent_launch_t(hipStream_t s, const LatentLaunch& L, int nblk, size_t lb) {
    if (lb > 64 * 1024) {
        static bool set = false;
        if (!set) { (void)hipFuncSetAttribute((const void*)latent_fwd_kernel<MODE, DSL>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); set = true; }
    }
    DMVAE_LAUNCH((latent_fwd_kernel<MODE, DSL>), dim3(nblk), dim3(256), lb, s, L);
}

int latent_launch(hipStream_t s, const dmvae_latent_args* a) {
    LatentLaunch L;
    L.a = *a;
    size_t lb;
    latent_geometry(a->B_pad, a->D, a->K, L.RB, L.DC, L.nchunks, lb);
    if (a->B_pad % 64 != 0 || a->B > a->B_pad || a->D < 1 || a->K < 1) {
        set_error("dmvae_latent_fwd: B_pad=%d must be a multiple of 64 and >= B=%d", a->B_pad, a->B);
        return DMVAE_EINVAL;
    }
    if (lb > 150 * 1024) {
        set_error("dmvae_latent_fwd: K=%d D=%d needs %zu B of LDS (> 150 KiB); not supported yet", a->K, a->D, lb);
        return DMVAE_EUNSUPPORTED;
    }
    if (a->mode == 2) return latent_vade_launch(s, a);
    if (a->mfma_ws && latent_mfma_applies(a->D, a->K, a->mode) && a->mfma_ws_bytes >= latent_mfma_ws_bytes(a->B_pad, a->D, a->K))
        return latent_mfma_launch(s, a, reinterpret_cast<float*>(a->mfma_ws), a->mfma_ws_bytes);
    const int nblk = (a->B_pad + L.RB - 1) / L.RB;
    const double bytes = 4.0 * ((double)a->B * (6.0 * a->D + 3.0 * a->K) + 2.0 * a->K * a->D * (nblk + 1));
    ProfScope ps(s, a->mode == 0 ? "latent_fwd_exact" : "latent_fwd_relaxed", 6.0 * a->B * (double)a->K * a->D, bytes);
#define LAT(MODE_) \
    switch (L.DC) { \
        case 16: latent_launch_t<MODE_, 1>(s, L, nblk, lb); break; \
        case 32: latent_launch_t<MODE_, 2>(s, L, nblk, lb); break; \
        case 64: latent_launch_t<MODE_, 4>(s, L, nblk, lb); break; \
        case 128: latent_launch_t<MODE_, 8>(s, L, nblk, lb); break; \
        default: latent_launch_t<MODE_, 16>(s, L, nblk, lb); break; \
    }
    if (a->mode == 0) { LAT(0) } else { LAT(1) }
#undef LAT
    return check_launch("latent_fwd");
}

}  // namespace dmvae
