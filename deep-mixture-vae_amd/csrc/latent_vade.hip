// Latent stage of VaDE (code/base_models.py:435-562): the mixture weights are not a softmax of
// encoder logits but the responsibilities of the SAMPLE, gamma = get_cluster_probs(Z) (code/priors.py:91-102):
//     u_bk = -1/2 [ sum_d (z_bd - pm_kd)^2 ip_kd + sum_d plv_kd ],   gamma_b = softmax_k(u_b),   ip = exp(-plv)
// and they weight both the exact mixture KL (priors.py:131-145) and the categorical KL (priors.py:183-201, "probs").
// One kernel: reparameterisation, gamma, KL_Z, KL_C and every gradient of kl_ratio * (KL_C + KL_Z) -- including the
// path THROUGH gamma into Z (hence mean / log_var) and into the prior tables:
//     G_k  = dL/dgamma_k = r/B [ T_k / 2 + log(gamma_k + e0) + gamma_k / (gamma_k + e0) + log K ],  T_k = sum_d t_kd
//     du_k = gamma_k (G_k - sum_j gamma_j G_j)
//     dZ_d (latent part) = - sum_k du_k (z_d - pm_kd) ip_kd
// so the two arrays handed to the dZ GEMM's DMVAE_EPI_LATENT epilogue become
//     gmu' = dKL/dmean|gamma + dZ_lat,     glv' = dKL/dlog_var|gamma + dZ_lat * clv        (clv = eps/2 exp(lv/2))
// and the backward of the stage is again  dmean = dZ + gmu',  dlog_var = dZ * clv + glv'.
// Geometry: 16 lanes per row, 16 rows per 256-thread workgroup, the prior tables whole in LDS (VaDE's tables are
// small: 10 x 10 in the reference); larger tables than ~50 KiB return DMVAE_EUNSUPPORTED.  Per-block partials of the
// prior-table gradients, summed in a fixed order by step_finalize (no float atomics).
#include "kernels.h"

namespace dmvae {

__device__ __forceinline__ float sum16(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
    return v;
}
__device__ __forceinline__ float max16(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 16));
    return v;
}

__global__ __launch_bounds__(256) void latent_vade_kernel(dmvae_latent_args a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K = a.K, D = a.D, DP = D + 1;
    constexpr int RB = 16;
    float* tpm = lds;                 // [K][DP] prior means
    float* tip = tpm + K * DP;        // [K][DP] exp(-prior_log_var)
    float* ck = tip + K * DP;         // [K] sum_d prior_log_var
    float* gam = ck + K;              // [RB][K] gamma
    float* dus = gam + RB * K;        // [RB][K] dL/du
    float* rz = dus + RB * K;         // [RB][DP] z
    float* rmu = rz + RB * DP;        // [RB][DP] mu
    float* re = rmu + RB * DP;        // [RB][DP] exp(log_var)
    float* rcl = re + RB * DP;        // [RB][DP] reparameterisation coefficient eps/2 exp(lv/2)
    float* red = rcl + RB * DP;       // [32]

    const int tid = threadIdx.x, lr = tid & 15, rsub = tid >> 4;
    const int b = blockIdx.x * RB + rsub;
    const bool valid = b < a.B;
    const dmvae_state* st = reinterpret_cast<const dmvae_state*>(a.state);
    const float klr = st ? st->kl_ratio : a.kl_ratio;
    const uint64_t nstep = st ? st->noise_step : a.noise_step;
    const float rB = klr * a.inv_B, rB2 = 0.5f * rB, logK = __logf((float)K);

    for (int idx = tid; idx < K * D; idx += 256) {
        const int k = idx / D, d = idx - k * D;
        tpm[k * DP + d] = a.prior_means[idx];
        tip[k * DP + d] = __expf(-a.prior_log_vars[idx]);
    }
    for (int k = rsub; k < K; k += 16) {
        float s = 0.f;
        for (int d0 = lr; d0 < D; d0 += 64) {      // four loads in flight per lane (latent.hip: one load per trip was a round trip per trip); same order of additions
            float t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = d0 + 16 * i;
                t[i] = a.prior_log_vars[(int64_t)k * D + (d < D ? d : 0)];
                t[i] = d < D ? t[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s += t[i];
        }
        s = sum16(s);
        if (lr == 0) ck[k] = s;
    }
    // reparameterisation: lane owns d = lr + 16 i
    float lvsum = 0.f;
    for (int d = lr; d < a.ld_Z; d += 16) {
        float z = 0.f;
        if (d < D) {
            float mu = 0.f, e = 0.f, cl = 0.f;
            if (valid) {
                mu = a.mean[(int64_t)b * a.ld_mean + d];
                const float lv = a.log_var[(int64_t)b * a.ld_log_var + d];
                const float ep = a.eps ? a.eps[(int64_t)b * a.ld_eps + d] : philox_normal_at(a.seed, nstep, 0u, (uint64_t)b * D + d);
                const float sd = __expf(0.5f * lv);
                e = __expf(lv);
                z = mu + sd * ep;
                cl = ep * 0.5f * sd;
                lvsum += lv;
            }
            rz[rsub * DP + d] = z; rmu[rsub * DP + d] = mu; re[rsub * DP + d] = e; rcl[rsub * DP + d] = cl;
            if (a.Z_f32) a.Z_f32[(int64_t)b * a.ld_Zf + d] = z;
            a.clv[(int64_t)b * a.ld_g + d] = cl;
        }
        if (a.act_dtype == DMVAE_BF16) reinterpret_cast<bf16_t*>(a.Z_act)[(int64_t)b * a.ld_Z + d] = f2bf(z);     // pad columns: zeros
        else reinterpret_cast<float*>(a.Z_act)[(int64_t)b * a.ld_Z + d] = z;
    }
    lvsum = sum16(lvsum);
    __syncthreads();

    // responsibilities: lane owns k = lr + 16 j
    float mx = -INFINITY;
    for (int k = lr; k < K; k += 16) {
        float su = 0.f, stt = 0.f;
        for (int d = 0; d < D; ++d) {
            const float pmv = tpm[k * DP + d], ipv = tip[k * DP + d];
            const float dz = rz[rsub * DP + d] - pmv, dm = rmu[rsub * DP + d] - pmv;
            su += dz * dz * ipv;
            stt += (re[rsub * DP + d] + dm * dm) * ipv;
        }
        const float u = -0.5f * (su + ck[k]);
        gam[rsub * K + k] = u;                                   // u for now
        dus[rsub * K + k] = ck[k] - lvsum - (float)D + stt;        // T_k for now
        mx = fmaxf(mx, u);
    }
    mx = max16(mx);
    float se = 0.f;
    for (int k = lr; k < K; k += 16) {
        const float ex = __expf(gam[rsub * K + k] - mx);
        gam[rsub * K + k] = ex;
        se += ex;
    }
    se = sum16(se);
    float sgG = 0.f, klz = 0.f, klc = 0.f;
    for (int k = lr; k < K; k += 16) {
        const float g = valid ? gam[rsub * K + k] / se : 0.f;
        const float T = dus[rsub * K + k];
        const float lg = __logf(g + 1e-20f);
        const float G = rB * (0.5f * T + lg + g / (g + 1e-20f) + logK);
        gam[rsub * K + k] = g;
        dus[rsub * K + k] = G;                                     // G for now
        sgG += g * G;
        klz += 0.5f * g * T;
        klc += g * (lg + logK);
        if (a.weights) a.weights[(int64_t)b * a.ld_w + k] = g;
    }
    sgG = sum16(sgG); klz = sum16(klz); klc = sum16(klc);
    for (int k = lr; k < K; k += 16) dus[rsub * K + k] = gam[rsub * K + k] * (dus[rsub * K + k] - sgG);      // du
    __syncthreads();

    // gradients wrt mean / log_var (direct + through gamma(Z)): lane owns d
    for (int d = lr; d < D; d += 16) {
        const float mu = rmu[rsub * DP + d], z = rz[rsub * DP + d], e = re[rsub * DP + d];
        float gm = 0.f, A = 0.f, dzl = 0.f;
        for (int k = 0; k < K; ++k) {
            const float g = gam[rsub * K + k], pmv = tpm[k * DP + d], ipv = tip[k * DP + d];
            gm += g * (mu - pmv) * ipv;
            A += g * ipv;
            dzl -= dus[rsub * K + k] * (z - pmv) * ipv;
        }
        const float cl = rcl[rsub * DP + d];
        a.gmu[(int64_t)b * a.ld_g + d] = valid ? rB * gm + dzl : 0.f;
        a.glv[(int64_t)b * a.ld_g + d] = valid ? rB2 * (e * A - 1.f) + dzl * cl : 0.f;
    }

    // prior-table gradient partials of this block: threads over (k, d), rows in ascending order
    for (int idx = tid; idx < K * D; idx += 256) {
        const int k = idx / D, d = idx - k * D;
        const float pmv = tpm[k * DP + d], ipv = tip[k * DP + d];
        float a1 = 0.f, a2 = 0.f;
        for (int r = 0; r < RB; ++r) {
            const float g = gam[r * K + k], du = dus[r * K + k];
            const float dm = rmu[r * DP + d] - pmv, dz = rz[r * DP + d] - pmv;
            a1 += -rB * g * dm * ipv + du * dz * ipv;
            a2 += rB2 * g * (1.f - (re[r * DP + d] + dm * dm) * ipv) + du * 0.5f * (dz * dz * ipv - 1.f);
        }
        float* o = a.dprior_partials + (int64_t)blockIdx.x * 2 * K * D;
        o[idx] = a1;
        o[(int64_t)K * D + idx] = a2;
    }
    if (lr == 0) { red[rsub] = valid ? klz : 0.f; red[16 + rsub] = valid ? klc : 0.f; }
    __syncthreads();
    if (tid == 0) {
        float z = 0.f, c = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { z += red[i]; c += red[16 + i]; }
        a.loss_partials[2 * blockIdx.x] = z;
        a.loss_partials[2 * blockIdx.x + 1] = c;
    }
}

static size_t vade_lds_bytes(int D, int K) {
    return sizeof(float) * ((size_t)2 * K * (D + 1) + K + (size_t)2 * 16 * K + (size_t)4 * 16 * (D + 1) + 32);
}
int latent_vade_nblocks(int B_pad) { return B_pad / 16; }

int latent_vade_launch(hipStream_t s, const dmvae_latent_args* a) {
    if (a->B_pad % 64 != 0 || a->B > a->B_pad || a->D < 1 || a->K < 1) {
        set_error("dmvae_latent_fwd (VaDE): B_pad=%d must be a multiple of 64 and >= B=%d", a->B_pad, a->B);
        return DMVAE_EINVAL;
    }
    const size_t lb = vade_lds_bytes(a->D, a->K);
    if (lb > 60 * 1024) {
        set_error("dmvae_latent_fwd (VaDE): K=%d D=%d needs %zu B of LDS: the VaDE latent stage keeps its prior tables whole in LDS", a->K, a->D, lb);
        return DMVAE_EUNSUPPORTED;
    }
    const int nblk = a->B_pad / 16;
    ProfScope ps(s, "latent_vade", 14.0 * a->B * (double)a->K * a->D, 4.0 * ((double)a->B * (7.0 * a->D + a->K) + 2.0 * a->K * a->D * (nblk + 1)));
    DMVAE_LAUNCH(latent_vade_kernel, dim3(nblk), dim3(256), lb, s, *a);
    return check_launch("latent_vade");
}

}  // namespace dmvae
