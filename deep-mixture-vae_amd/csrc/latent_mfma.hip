// MFMA form of the latent kernel's (row, cluster, dimension) contractions -- exact mixture KL
// (code/priors.py:131-145, cluster_sample False) for LARGE prior tables (K * D >~ 4000: BASELINE.json configs[3]
// K=50 D=256, configs[4] K=256 D=512), where latent.hip's VALU loops over (row, k, d) dominate (150 us of a 0.8 ms
// step; 2.5 ms of 14 ms).  With ip = exp(-prior_log_var), pm = prior_mean, e = exp(log_var), mu = mean, w = softmax:
//
//   A_bd  = sum_k w_bk ip_kd,  C_bd = sum_k w_bk pm_kd ip_kd          [A | C] = W . [ip | pm ip]            (G1)
//           -> dKL/dmean = r/B (mu A - C),  dKL/dlog_var = r/2B (e A - 1)
//   S_bk  = sum_d (e_bd + (mu_bd - pm_kd)^2) ip_kd
//         = [e + mu^2 | mu] . [ip | -2 pm ip]^T + c2_k,   c2_k = sum_d pm^2 ip                              (G2)
//           -> t_bk = S_bk + sum_d plv_kd - sum_d lv_bd - D,  KL_Z = mean_b 1/2 sum_k w_bk t_bk,  dKL/dw = r/2B t
//   G     = W^T . [e + mu^2 | mu | 1]     (contraction over the batch)                                      (G3)
//           -> d prior_mean = -r/B ip (G_mu - pm Wsum),  d prior_log_var = r/2B (Wsum - ip (G_e - 2 pm G_mu + pm^2 Wsum))
//
// three GEMMs in EXACT f32 (v_mfma_f32_16x16x4_f32 through gemm_f32.hip: the expanded squares cancel, bf16 operands
// would not hold the 2e-5 relative KL tolerance) between two row-wise kernels:
//   latent_pre   softmax, KL_C, reparameterisation (Z, its coefficient), W and [e + mu^2 | mu | 1] -- 16-byte accesses
//   latent_post  the gradients wrt mean / log_var / logits, KL_Z; its extra workgroups sum the G slabs (fixed order)
//                into the prior-table gradients.
// The [B, K, D] tensor TensorFlow materialises is still never formed; the intermediates are [B, 2D] and [B, K].
// No float atomics: G3's split over the batch writes slabs.  Device noise: one Philox block per four columns of a
// row, keyed by the element index (a different, equally valid stream than latent.hip's geometry-keyed one).
#include <string.h>

#include <algorithm>

#include "kernels.h"

namespace dmvae {

static inline int pad64i(int x) { return (x + 63) / 64 * 64; }

struct LatentMfmaWs {      // float offsets into the caller's scratch
    int Dp, Kp, XW, nsplit, nsplit_s;      // K slices of G3 (over the batch) and of G2 (over 2D)
    int64_t T1, T2, c2, ck, Wm, X1, AC, S, RL, G, total;
};
static LatentMfmaWs latent_mfma_layout(int Bp, int D, int K) {
    LatentMfmaWs w;
    w.Dp = pad64i(D); w.Kp = pad64i(K); w.XW = 2 * w.Dp + 64;
    const int tiles = (w.Kp / 64) * (w.XW / 64);
    int ns = 1;
    while (ns < 32 && tiles * ns < 512 && (Bp / 64) % (2 * ns) == 0) ns *= 2;      // slices of the batch: fill the chip, stay multiples of 64 rows
    w.nsplit = ns;
    // G2 = [B, 2D] x [2D, K]: with few clusters (N = K padded to 64) its grid is B / 64 workgroups of a long K loop -- cut 2D
    int nss = 1;
    while (nss < 8 && (Bp / 64) * (w.Kp / 64) * nss < 512 && (2 * w.Dp / 64) % (2 * nss) == 0) nss *= 2;
    w.nsplit_s = nss;
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t r = o; o += (n + 63) / 64 * 64; return r; };
    w.T1 = take((int64_t)w.Kp * 2 * w.Dp); w.T2 = take((int64_t)w.Kp * 2 * w.Dp);
    w.c2 = take(w.Kp); w.ck = take(w.Kp);
    w.Wm = take((int64_t)Bp * w.Kp); w.X1 = take((int64_t)Bp * w.XW);
    w.AC = take((int64_t)Bp * 2 * w.Dp); w.S = take((int64_t)nss * Bp * w.Kp); w.RL = take(Bp);
    w.G = take((int64_t)ns * w.Kp * w.XW);
    w.total = o;
    return w;
}

bool latent_mfma_applies(int D, int K, int mode) { return mode == 0 && (int64_t)D * K >= 4096; }
int64_t latent_mfma_ws_bytes(int B_pad, int D, int K) { return 4 * latent_mfma_layout(B_pad, D, K).total; }

// ---- prior tables -> GEMM operands (tiny: K * D elements)
__device__ __forceinline__ void latent_tables_block(const int k, const float* __restrict__ pm, const float* __restrict__ plv, int K, int D, int Kp, int Dp,
                                                    float* __restrict__ T1, float* __restrict__ T2, float* __restrict__ c2, float* __restrict__ ck, float* red) {
    // one block per (padded) cluster row k
    float s2 = 0.f, sl = 0.f;
    for (int d = threadIdx.x; d < Dp; d += 256) {
        float ip = 0.f, m = 0.f;
        if (k < K && d < D) {
            const float lv = plv[(int64_t)k * D + d];
            m = pm[(int64_t)k * D + d];
            ip = __expf(-lv);
            s2 += m * m * ip;
            sl += lv;
        }
        T1[(int64_t)k * 2 * Dp + d] = ip;
        T1[(int64_t)k * 2 * Dp + Dp + d] = m * ip;
        T2[(int64_t)k * 2 * Dp + d] = ip;
        T2[(int64_t)k * 2 * Dp + Dp + d] = -2.f * m * ip;
    }
    const float a = block_sum_256(s2, red);
    const float b = block_sum_256(sl, red + 4);
    if (threadIdx.x == 0) { c2[k] = a; ck[k] = b; }
}

struct LatentMfmaArgs {
    dmvae_latent_args a;
    LatentMfmaWs w;
    float* ws;
    int RB;          // rows per workgroup: latent.hip's geometry, so that both paths write the same number of loss partials
};

// ---- rows, before the GEMMs.  16 lanes per row, 16 rows per 256-thread block; a lane owns quads of columns.
__global__ __launch_bounds__(256) void latent_pre_kernel(LatentMfmaArgs L, int nrow_blocks) {
    const dmvae_latent_args& a = L.a;
    if ((int)blockIdx.x >= nrow_blocks) {        // extra workgroups: the prior tables as GEMM operands
        __shared__ float tred[8];
        latent_tables_block((int)blockIdx.x - nrow_blocks, a.prior_means, a.prior_log_vars, a.K, a.D, L.w.Kp, L.w.Dp, L.ws + L.w.T1, L.ws + L.w.T2,
                            L.ws + L.w.c2, L.ws + L.w.ck, tred);
        return;
    }
    const int lane16 = threadIdx.x & 15, rsub = threadIdx.x >> 4;
    const int D = a.D, K = a.K, Dp = L.w.Dp, Kp = L.w.Kp, XW = L.w.XW;
    const dmvae_state* st = reinterpret_cast<const dmvae_state*>(a.state);
    const uint64_t nstep = st ? st->noise_step : a.noise_step;
    __shared__ float red[16];
    float kc_acc = 0.f;
  for (int pass = 0; pass < L.RB / 16; ++pass) {
    const int b = blockIdx.x * L.RB + pass * 16 + rsub;
    const bool valid = b < a.B;
    float* Wm = L.ws + L.w.Wm + (int64_t)b * Kp;
    float* X1 = L.ws + L.w.X1 + (int64_t)b * XW;

    // softmax over K (lane owns k = lane16 + 16 i), KL_C.  Loads of a pad row read row 0 (their values are not used): every load below is
    // unconditional -- a load inside a divergent `if` (or `cond ? load : 0`) is waited for at the join, and the three passes over the logits
    // were twelve memory round trips per row at K = 50 (round 4; the same arithmetic in the same order)
    const int64_t br = valid ? b : 0;
    const float* lrow = a.logits + br * a.ld_logits;
    const float logK = __logf((float)K);
    float kc = 0.f;
    if (K <= 64) {            // a lane's clusters fit four registers: one batch of loads serves the three passes
        float v[4];
        bool in[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = lane16 + 16 * j;
            in[j] = k < K;
            v[j] = lrow[in[j] ? k : 0];
            v[j] = valid ? v[j] : 0.f;
        }
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) mx = in[j] ? fmaxf(mx, v[j]) : mx;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
        float se = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) se += in[j] ? __expf(v[j] - mx) : 0.f;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) se += __shfl_xor(se, o, 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = lane16 + 16 * j;
            if (k < Kp) {
                float q = 0.f;
                if (valid && in[j]) {
                    q = __expf(v[j] - mx) / se;
                    kc += q * (__logf(q + 1e-20f) + logK);
                }
                Wm[k] = q;
                if (a.weights && in[j]) a.weights[(int64_t)b * a.ld_w + k] = q;
            }
        }
    } else {
        float mx = -INFINITY;
        for (int k0 = lane16; k0 < K; k0 += 64) {          // four loads in flight per lane and trip
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = lrow[k0 + 16 * j < K ? k0 + 16 * j : 0];
#pragma unroll
            for (int j = 0; j < 4; ++j) mx = k0 + 16 * j < K ? fmaxf(mx, valid ? v[j] : 0.f) : mx;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 16));
        float se = 0.f;
        for (int k0 = lane16; k0 < K; k0 += 64) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = lrow[k0 + 16 * j < K ? k0 + 16 * j : 0];
#pragma unroll
            for (int j = 0; j < 4; ++j) se += k0 + 16 * j < K ? __expf((valid ? v[j] : 0.f) - mx) : 0.f;
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) se += __shfl_xor(se, o, 16);
        for (int k0 = lane16; k0 < Kp; k0 += 64) {
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = lrow[k0 + 16 * j < K ? k0 + 16 * j : 0];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + 16 * j;
                if (k < Kp) {
                    float q = 0.f;
                    if (valid && k < K) {
                        q = __expf(v[j] - mx) / se;
                        kc += q * (__logf(q + 1e-20f) + logK);
                    }
                    Wm[k] = q;
                    if (a.weights && k < K) a.weights[(int64_t)b * a.ld_w + k] = q;
                }
            }
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) kc += __shfl_xor(kc, o, 16);
    kc_acc += kc;

    // columns in quads: Z, reparameterisation coefficient, [e + mu^2 | mu]
    const bool vec = (D % 4 == 0) && (a.ld_mean % 4 == 0) && (a.ld_log_var % 4 == 0) && (a.ld_g % 4 == 0) && (a.ld_Z % 4 == 0) &&
                     (!a.eps || a.ld_eps % 4 == 0) && (!a.Z_f32 || a.ld_Zf % 4 == 0);
    float lvsum = 0.f;
    if (vec) {
        // four quads per lane and trip: their 12 loads issued together, unconditionally (pad rows read row 0, pad quads read quad 0); the outputs
        // as 8- / 16-byte stores.  Same arithmetic as the general loop below.
        const float* mrow = a.mean + br * a.ld_mean;
        const float* vrow = a.log_var + br * a.ld_log_var;
        const float* erow = a.eps ? a.eps + br * a.ld_eps : nullptr;
        const int nq = Dp / 4;
        for (int q0 = lane16; q0 < nq; q0 += 64) {
            float4 m4[4], l4[4], e4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int d0 = 4 * (q0 + 16 * u);
                const int off = (q0 + 16 * u < nq && d0 < D) ? d0 : 0;
                m4[u] = *reinterpret_cast<const float4*>(mrow + off);
                l4[u] = *reinterpret_cast<const float4*>(vrow + off);
                e4[u] = erow ? *reinterpret_cast<const float4*>(erow + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q4 = q0 + 16 * u, d0 = 4 * q4;
                if (q4 >= nq) continue;
                const bool in = valid && d0 < D;            // (D % 4 == 0 here: the whole quad)
                float mu[4] = {in ? m4[u].x : 0.f, in ? m4[u].y : 0.f, in ? m4[u].z : 0.f, in ? m4[u].w : 0.f};
                const float lv[4] = {in ? l4[u].x : 0.f, in ? l4[u].y : 0.f, in ? l4[u].z : 0.f, in ? l4[u].w : 0.f};
                float ep[4] = {in ? e4[u].x : 0.f, in ? e4[u].y : 0.f, in ? e4[u].z : 0.f, in ? e4[u].w : 0.f};
                if (in && !a.eps) philox_normal4(a.seed, nstep, 0u, (uint64_t)b * (Dp / 4) + q4, ep);
                float z[4], cl[4], x1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float e = __expf(lv[j]), sd = __expf(0.5f * lv[j]);
                    z[j] = in ? mu[j] + sd * ep[j] : 0.f;
                    cl[j] = in ? ep[j] * 0.5f * sd : 0.f;
                    x1[j] = in ? e + mu[j] * mu[j] : 0.f;
                    lvsum += in ? lv[j] : 0.f;
                }
                *reinterpret_cast<float4*>(X1 + d0) = make_float4(x1[0], x1[1], x1[2], x1[3]);
                *reinterpret_cast<float4*>(X1 + Dp + d0) = make_float4(mu[0], mu[1], mu[2], mu[3]);
                if (d0 + 4 <= a.ld_Z) {
                    if (a.act_dtype == DMVAE_BF16) {
                        uint2 pk;
                        pk.x = pack2bf(z[0], z[1]); pk.y = pack2bf(z[2], z[3]);
                        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(a.Z_act) + (int64_t)b * a.ld_Z + d0) = pk;
                    } else *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.Z_act) + (int64_t)b * a.ld_Z + d0) = make_float4(z[0], z[1], z[2], z[3]);
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (d0 + j < a.ld_Z) {
                            if (a.act_dtype == DMVAE_BF16) reinterpret_cast<bf16_t*>(a.Z_act)[(int64_t)b * a.ld_Z + d0 + j] = f2bf(z[j]);
                            else reinterpret_cast<float*>(a.Z_act)[(int64_t)b * a.ld_Z + d0 + j] = z[j];
                        }
                }
                if (d0 < D) {
                    if (a.Z_f32) *reinterpret_cast<float4*>(a.Z_f32 + (int64_t)b * a.ld_Zf + d0) = make_float4(z[0], z[1], z[2], z[3]);
                    *reinterpret_cast<float4*>(a.clv + (int64_t)b * a.ld_g + d0) = make_float4(cl[0], cl[1], cl[2], cl[3]);
                }
            }
        }
    } else
    for (int q4 = lane16; q4 < Dp / 4; q4 += 16) {
        const int d0 = 4 * q4;
        float mu[4] = {0.f, 0.f, 0.f, 0.f}, lv[4] = {0.f, 0.f, 0.f, 0.f}, ep[4] = {0.f, 0.f, 0.f, 0.f};
        if (valid && d0 < D) {
            if (vec) {
                const float4 m4 = *reinterpret_cast<const float4*>(a.mean + (int64_t)b * a.ld_mean + d0);
                const float4 l4 = *reinterpret_cast<const float4*>(a.log_var + (int64_t)b * a.ld_log_var + d0);
                mu[0] = m4.x; mu[1] = m4.y; mu[2] = m4.z; mu[3] = m4.w;
                lv[0] = l4.x; lv[1] = l4.y; lv[2] = l4.z; lv[3] = l4.w;
                if (a.eps) {
                    const float4 e4 = *reinterpret_cast<const float4*>(a.eps + (int64_t)b * a.ld_eps + d0);
                    ep[0] = e4.x; ep[1] = e4.y; ep[2] = e4.z; ep[3] = e4.w;
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (d0 + j < D) {
                        mu[j] = a.mean[(int64_t)b * a.ld_mean + d0 + j];
                        lv[j] = a.log_var[(int64_t)b * a.ld_log_var + d0 + j];
                        if (a.eps) ep[j] = a.eps[(int64_t)b * a.ld_eps + d0 + j];
                    }
            }
            if (!a.eps) philox_normal4(a.seed, nstep, 0u, (uint64_t)b * (Dp / 4) + q4, ep);
        }
        float z[4], cl[4], x1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = valid && d0 + j < D;
            const float e = __expf(lv[j]), sd = __expf(0.5f * lv[j]);
            z[j] = ok ? mu[j] + sd * ep[j] : 0.f;
            cl[j] = ok ? ep[j] * 0.5f * sd : 0.f;
            x1[j] = ok ? e + mu[j] * mu[j] : 0.f;
            if (!ok) mu[j] = 0.f;
            lvsum += ok ? lv[j] : 0.f;
        }
        *reinterpret_cast<float4*>(X1 + d0) = make_float4(x1[0], x1[1], x1[2], x1[3]);
        *reinterpret_cast<float4*>(X1 + Dp + d0) = make_float4(mu[0], mu[1], mu[2], mu[3]);
        // outputs: Z (act dtype, pad columns up to ld_Z zeroed: they are K padding of the first decoder GEMM), f32 copy, coefficient
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = d0 + j;
            if (d < a.ld_Z) {
                if (a.act_dtype == DMVAE_BF16) reinterpret_cast<bf16_t*>(a.Z_act)[(int64_t)b * a.ld_Z + d] = f2bf(z[j]);
                else reinterpret_cast<float*>(a.Z_act)[(int64_t)b * a.ld_Z + d] = z[j];
            }
            if (d < D) {
                if (a.Z_f32) a.Z_f32[(int64_t)b * a.ld_Zf + d] = z[j];
                a.clv[(int64_t)b * a.ld_g + d] = cl[j];
            }
        }
    }
    for (int d = Dp + lane16; d < a.ld_Z; d += 16) {      // (a Z buffer wider than D padded to 64)
        if (a.act_dtype == DMVAE_BF16) reinterpret_cast<bf16_t*>(a.Z_act)[(int64_t)b * a.ld_Z + d] = 0;
        else reinterpret_cast<float*>(a.Z_act)[(int64_t)b * a.ld_Z + d] = 0.f;
    }
    for (int c = lane16; c < 64; c += 16) X1[2 * Dp + c] = (c == 0 && valid) ? 1.f : 0.f;      // the ones column: G3 then also yields sum_b w_bk
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) lvsum += __shfl_xor(lvsum, o, 16);
    if (lane16 == 0) L.ws[L.w.RL + b] = lvsum;
  }
    if (lane16 == 0) red[rsub] = kc_acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float c = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) c += red[i];
        a.loss_partials[2 * blockIdx.x + 1] = c;
    }
}

// ---- rows, after the GEMMs; workgroups >= nrow_blocks: the prior-table gradients from the G slabs
__global__ __launch_bounds__(256) void latent_post_kernel(LatentMfmaArgs L, int nrow_blocks) {
    const dmvae_latent_args& a = L.a;
    const int D = a.D, K = a.K, Dp = L.w.Dp, Kp = L.w.Kp, XW = L.w.XW;
    const dmvae_state* st = reinterpret_cast<const dmvae_state*>(a.state);
    const float klr = st ? st->kl_ratio : a.kl_ratio;
    const float rB = klr * a.inv_B, rB2 = 0.5f * rB;
    if ((int)blockIdx.x >= nrow_blocks) {
        const int64_t KD = (int64_t)K * D;
        for (int64_t idx = (int64_t)((int)blockIdx.x - nrow_blocks) * 256 + threadIdx.x; idx < KD; idx += (int64_t)((int)gridDim.x - nrow_blocks) * 256) {
            const int k = (int)(idx / D), d = (int)(idx - (int64_t)k * D);
            float ge = 0.f, gm = 0.f, wsum = 0.f;
            for (int s = 0; s < L.w.nsplit; ++s) {                     // slabs in ascending order
                const float* g = L.ws + L.w.G + ((int64_t)s * Kp + k) * XW;
                ge += g[d]; gm += g[Dp + d]; wsum += g[2 * Dp];
            }
            const float ip = L.ws[L.w.T1 + (int64_t)k * 2 * Dp + d];
            const float pmv = a.prior_means[idx];
            a.dprior_partials[idx] = -rB * ip * (gm - pmv * wsum);
            a.dprior_partials[KD + idx] = rB2 * (wsum - ip * (ge - 2.f * pmv * gm + pmv * pmv * wsum));
        }
        return;
    }
    const int lane16 = threadIdx.x & 15, rsub = threadIdx.x >> 4;
    __shared__ float red[16];
    float klz_acc = 0.f;
  for (int pass = 0; pass < L.RB / 16; ++pass) {
    const int b = blockIdx.x * L.RB + pass * 16 + rsub;
    const bool valid = b < a.B;
    const float* AC = L.ws + L.w.AC + (int64_t)b * 2 * Dp;
    const float* S0 = L.ws + L.w.S + (int64_t)b * Kp;
    const int64_t sstr = (int64_t)a.B_pad * Kp;       // slab stride of G2's K slices
    auto Ssum = [&](int k) { float v = S0[k]; for (int sl = 1; sl < L.w.nsplit_s; ++sl) v += S0[sl * sstr + k]; return v; };
    const float* Wm = L.ws + L.w.Wm + (int64_t)b * Kp;
    const float* X1 = L.ws + L.w.X1 + (int64_t)b * XW;
    // gradients wrt mean / log_var.  Aligned shapes: four quads per lane and trip, every load unconditional (a pad row reads row 0 of log_var),
    // 16-byte stores -- the general loop below fetched log_var one element at a time inside `if (d < D)`, a memory round trip each
    const int64_t br = valid ? b : 0;
    if (D % 4 == 0 && a.ld_log_var % 4 == 0 && a.ld_g % 4 == 0) {
        const float* vrow = a.log_var + br * a.ld_log_var;
        const int nq = D / 4;
        for (int q0 = lane16; q0 < nq; q0 += 64) {
            float4 A4[4], C4[4], M4[4], V4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int d0 = q0 + 16 * u < nq ? 4 * (q0 + 16 * u) : 0;
                A4[u] = *reinterpret_cast<const float4*>(AC + d0);
                C4[u] = *reinterpret_cast<const float4*>(AC + Dp + d0);
                M4[u] = *reinterpret_cast<const float4*>(X1 + Dp + d0);
                V4[u] = *reinterpret_cast<const float4*>(vrow + d0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (q0 + 16 * u >= nq) continue;
                const int d0 = 4 * (q0 + 16 * u);
                const float A[4] = {A4[u].x, A4[u].y, A4[u].z, A4[u].w}, Cc[4] = {C4[u].x, C4[u].y, C4[u].z, C4[u].w};
                const float mu[4] = {M4[u].x, M4[u].y, M4[u].z, M4[u].w}, lv[4] = {V4[u].x, V4[u].y, V4[u].z, V4[u].w};
                float gm[4], gl[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float e = valid ? __expf(lv[j]) : 0.f;
                    gm[j] = valid ? rB * (mu[j] * A[j] - Cc[j]) : 0.f;
                    gl[j] = valid ? rB2 * (e * A[j] - 1.f) : 0.f;
                }
                *reinterpret_cast<float4*>(a.gmu + (int64_t)b * a.ld_g + d0) = make_float4(gm[0], gm[1], gm[2], gm[3]);
                *reinterpret_cast<float4*>(a.glv + (int64_t)b * a.ld_g + d0) = make_float4(gl[0], gl[1], gl[2], gl[3]);
            }
        }
    } else
    for (int q4 = lane16; q4 < (D + 3) / 4; q4 += 16) {
        const int d0 = 4 * q4;
        const float4 A4 = *reinterpret_cast<const float4*>(AC + d0);
        const float4 C4 = *reinterpret_cast<const float4*>(AC + Dp + d0);
        const float4 M4 = *reinterpret_cast<const float4*>(X1 + Dp + d0);
        const float A[4] = {A4.x, A4.y, A4.z, A4.w}, Cc[4] = {C4.x, C4.y, C4.z, C4.w}, mu[4] = {M4.x, M4.y, M4.z, M4.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int d = d0 + j;
            if (d < D) {
                const float e = valid ? __expf(a.log_var[(int64_t)b * a.ld_log_var + d]) : 0.f;
                a.gmu[(int64_t)b * a.ld_g + d] = valid ? rB * (mu[j] * A[j] - Cc[j]) : 0.f;
                a.glv[(int64_t)b * a.ld_g + d] = valid ? rB2 * (e * A[j] - 1.f) : 0.f;
            }
        }
    }
    // KL_Z and the gradient wrt the logits
    const float rl = L.ws[L.w.RL + b], logK = __logf((float)K);
    float s_wdw = 0.f, s_qdq = 0.f, klz = 0.f;
    if (K <= 64) {            // a lane's clusters fit four registers: one batch of unconditional loads serves both passes (same arithmetic, same order)
        float w[4], t[4], dq[4];
        bool in[4];
        float ss[4], c2v[4], ckv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = lane16 + 16 * j;
            in[j] = k < K;
            const int kk = in[j] ? k : 0;
            w[j] = Wm[kk]; ss[j] = S0[kk]; c2v[j] = L.ws[L.w.c2 + kk]; ckv[j] = L.ws[L.w.ck + kk];
        }
        for (int sl = 1; sl < L.w.nsplit_s; ++sl) {
            float sv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) sv[j] = S0[sl * sstr + (in[j] ? lane16 + 16 * j : 0)];
#pragma unroll
            for (int j = 0; j < 4; ++j) ss[j] += sv[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            t[j] = ss[j] + c2v[j] + ckv[j] - rl - (float)D;
            dq[j] = rB * (__logf(w[j] + 1e-20f) + w[j] / (w[j] + 1e-20f) + logK);
            if (in[j]) {
                klz += 0.5f * w[j] * t[j];
                s_wdw += w[j] * (rB2 * t[j]);
                s_qdq += w[j] * dq[j];
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            s_wdw += __shfl_xor(s_wdw, o, 16); s_qdq += __shfl_xor(s_qdq, o, 16); klz += __shfl_xor(klz, o, 16);
        }
        for (int k = lane16; k < a.ld_dl; k += 16) {
            float dl = 0.f;
            const int j = (k - lane16) >> 4;
            if (k < K && valid) {
                const float wj = j == 0 ? w[0] : j == 1 ? w[1] : j == 2 ? w[2] : w[3];
                const float tj = j == 0 ? t[0] : j == 1 ? t[1] : j == 2 ? t[2] : t[3];
                const float dj = j == 0 ? dq[0] : j == 1 ? dq[1] : j == 2 ? dq[2] : dq[3];
                dl = wj * (dj - s_qdq) + wj * (rB2 * tj - s_wdw);
            }
            if (a.act_dtype == DMVAE_BF16) reinterpret_cast<bf16_t*>(a.dlogits_act)[(int64_t)b * a.ld_dl + k] = f2bf(dl);
            else reinterpret_cast<float*>(a.dlogits_act)[(int64_t)b * a.ld_dl + k] = dl;
        }
        klz_acc += valid ? klz : 0.f;
        continue;
    }
    for (int k = lane16; k < K; k += 16) {
        const float w = Wm[k];
        const float t = Ssum(k) + L.ws[L.w.c2 + k] + L.ws[L.w.ck + k] - rl - (float)D;
        klz += 0.5f * w * t;
        const float dq = rB * (__logf(w + 1e-20f) + w / (w + 1e-20f) + logK);
        s_wdw += w * (rB2 * t);
        s_qdq += w * dq;
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
        s_wdw += __shfl_xor(s_wdw, o, 16); s_qdq += __shfl_xor(s_qdq, o, 16); klz += __shfl_xor(klz, o, 16);
    }
    for (int k = lane16; k < a.ld_dl; k += 16) {
        float dl = 0.f;
        if (k < K && valid) {
            const float w = Wm[k];
            const float t = Ssum(k) + L.ws[L.w.c2 + k] + L.ws[L.w.ck + k] - rl - (float)D;
            const float dq = rB * (__logf(w + 1e-20f) + w / (w + 1e-20f) + logK);
            dl = w * (dq - s_qdq) + w * (rB2 * t - s_wdw);
        }
        if (a.act_dtype == DMVAE_BF16) reinterpret_cast<bf16_t*>(a.dlogits_act)[(int64_t)b * a.ld_dl + k] = f2bf(dl);
        else reinterpret_cast<float*>(a.dlogits_act)[(int64_t)b * a.ld_dl + k] = dl;
    }
    klz_acc += valid ? klz : 0.f;
  }
    if (lane16 == 0) red[rsub] = klz_acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float z = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) z += red[i];
        a.loss_partials[2 * blockIdx.x] = z;
    }
}

static GemmArgs f32_problem(int M, int N, int K, const float* A, int64_t lda, const float* B, int64_t ldb, float* out, int64_t ldo,
                            int split, int64_t slab_stride) {
    GemmArgs g;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.M = M; g.N = N; g.K = K; g.k_split = K / split; g.group_m = 8; g.conv_p = 0; g.conv_c = 0;
    memset(&g.epi, 0, sizeof(g.epi));
    g.epi.kind = DMVAE_EPI_STORE_F32; g.epi.out = out; g.epi.ldo = ldo; g.epi.m_valid = M; g.epi.n_valid = N;
    g.slab_stride = slab_stride;
    return g;
}

int latent_mfma_launch(hipStream_t s, const dmvae_latent_args* a, float* ws, int64_t ws_bytes) {
    const LatentMfmaWs w = latent_mfma_layout(a->B_pad, a->D, a->K);
    if (4 * w.total > ws_bytes) { set_error("dmvae_latent_fwd: MFMA-path scratch too small (%lld < %lld bytes)", (long long)ws_bytes, (long long)(4 * w.total)); return DMVAE_EINVAL; }
    LatentMfmaArgs L;
    L.a = *a; L.w = w; L.ws = ws;
    L.RB = a->B_pad / latent_nblocks(a->B_pad, a->D, a->K);        // latent.hip's rows per workgroup (16, 32 or 64)
    const int nrb = a->B_pad / L.RB;
    const double BD = (double)a->B * a->D, BK = (double)a->B * a->K;
    {   // the prior-table operands ride as Kp extra workgroups of the row kernel (one launch less)
        ProfScope ps(s, "latent_pre", 0.0, 4.0 * (BD * (2.0 + (a->eps ? 1.0 : 0.0) + 2.0 + 2.0 + 1.0) + BK * 3.0) + 16.0 * a->K * a->D);
        DMVAE_LAUNCH(latent_pre_kernel, dim3(nrb + w.Kp), dim3(256), 0, s, L, nrb);
    }
    int rc = check_launch("latent_pre");
    if (rc) return rc;
    {
        ProfScope ps(s, "latent_gemm_f32", 2.0 * a->B_pad * (double)w.Kp * (2.0 * w.Dp + 2.0 * w.Dp + w.XW), 4.0 * a->B_pad * (3.0 * w.Kp + 2.0 * w.XW + 2.0 * w.Dp));
        // G1 (forward layout), G2 (dX layout, K slices into slabs), G3 (dW layout, batch slices into slabs): all three read only what
        // latent_pre wrote -- ONE grid (gemm_f32_trio) instead of three launches
        const GemmArgs g1 = f32_problem(a->B_pad, 2 * w.Dp, w.Kp, ws + w.Wm, w.Kp, ws + w.T1, 2 * w.Dp, ws + w.AC, 2 * w.Dp, 1, 0);
        const GemmArgs g2 = f32_problem(a->B_pad, w.Kp, 2 * w.Dp, ws + w.X1, w.XW, ws + w.T2, 2 * w.Dp, ws + w.S, w.Kp, w.nsplit_s, (int64_t)a->B_pad * w.Kp);
        const GemmArgs g3 = f32_problem(w.Kp, w.XW, a->B_pad, ws + w.Wm, w.Kp, ws + w.X1, w.XW, ws + w.G, w.XW, w.nsplit, (int64_t)w.Kp * w.XW);
        rc = gemm_f32_trio(s, g1, 1, g2, w.nsplit_s, g3, w.nsplit);
    }
    if (rc) return rc;
    {
        const int extra = (int)std::min<int64_t>(256, ((int64_t)a->K * a->D + 255) / 256);
        ProfScope ps(s, "latent_post", 0.0, 4.0 * (BD * (2.0 + 1.0 + 1.0 + 2.0) + BK * 3.0) + 4.0 * w.nsplit * w.Kp * w.XW);
        DMVAE_LAUNCH(latent_post_kernel, dim3(nrb + extra), dim3(256), 0, s, L, nrb);
    }
    return check_launch("latent_post");
}

}  // namespace dmvae
