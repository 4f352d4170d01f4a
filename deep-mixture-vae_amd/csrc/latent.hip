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
// Work decomposition (HBM-bound; algorithmic bytes in DESIGN.md):
//   block = 256 threads = 4 waves, owns RB consecutive rows of the batch;
//   prior tables (mu_k, and exp(-logvar_k) or logvar_k) are staged in LDS in
//   D-chunks of DC columns (KL is separable in d);
//   phase 1: one wave per row, lanes over d (coalesced 4-byte loads, 256 B per
//            wave instruction), loop over k with the table row broadcast from
//            LDS, one wavefront reduction per (row, k);
//   phase 2: threads over (k, d) pairs, loop over the block's rows held in
//            LDS -> per-block partial of the prior-table gradients, written
//            to [nblocks][2][K][D] and summed by dmvae_colsum in a fixed
//            order (deterministic; no float atomics).
#include "kernels.h"

namespace dmvae {

struct LatentLaunch {
    dmvae_latent_args a;
    int RB;       // rows per block (multiple of 4)
    int DC;       // columns per chunk (16..256)
    int nchunks;
};

constexpr int LAT_DS = 4;   // d slots per lane: DC <= 256

template <int MODE>   // 0 exact, 1 relaxed
__global__ __launch_bounds__(256) void latent_fwd_kernel(LatentLaunch L) {
    const dmvae_latent_args& a = L.a;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int K = a.K, D = a.D, RB = L.RB, DC = L.DC;
    float* t1 = lds;                  // [K][DC] prior means
    float* t2 = t1 + K * DC;          // [K][DC] exp(-prior_log_var) (exact) | prior_log_var (relaxed)
    float* ck = t2 + K * DC;          // [K]     sum_d prior_log_var
    float* ws = ck + K;               // [RB][K] mixture weights (softmax or zeta); 0 for pad rows
    float* qs = ws + RB * K;          // [RB][K] softmax(logits)
    float* sk = qs + RB * K;          // [RB][K] exact: sum_d (e+(mu-pm)^2)*ip ; relaxed: dLoss/dzeta
    float* r1 = sk + RB * K;          // [RB][DC] exact: mu   | relaxed: dLoss/d(bar mean)
    float* r2 = r1 + RB * DC;         // [RB][DC] exact: e^lv | relaxed: dLoss/d(bar log_var)
    float* rowlv = r2 + RB * DC;      // [RB] sum_d log_var (exact) | sum_d relaxed KL integrand
    float* red = rowlv + RB;          // [8]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * RB;
    const dmvae_state* st = reinterpret_cast<const dmvae_state*>(a.state);
    const float klr = st ? st->kl_ratio : a.kl_ratio;
    const uint64_t nstep = st ? st->noise_step : a.noise_step;
    const float rB = klr * a.inv_B;          // r / B
    const float rB2 = 0.5f * rB;             // r / (2B)
    const float logK = __logf((float)K);

    // ---- prologue: c_k, per-row softmax / zeta, KL_C ----
    for (int k = wave; k < K; k += 4) {
        float s = 0.f;
        for (int d = lane; d < D; d += 64) s += a.prior_log_vars[(int64_t)k * D + d];
        s = wave_sum(s);
        if (lane == 0) ck[k] = s;
    }
    float klc_acc = 0.f, klz_acc = 0.f;   // per-wave (lane 0 meaningful)
    for (int r = wave; r < RB; r += 4) {
        const int b = row0 + r;
        const bool valid = b < a.B;
        // softmax over k (lanes stride k)
        float mx = -INFINITY, mz = -INFINITY;
        for (int k = lane; k < K; k += 64) {
            const float lg = valid ? a.logits[(int64_t)b * a.ld_logits + k] : 0.f;
            mx = fmaxf(mx, lg);
            if (MODE == 1) {
                const float gk = !valid ? 0.f
                    : (a.gumbel ? a.gumbel[(int64_t)b * a.ld_gumbel + k]
                                : philox_gumbel_at(a.seed, nstep, 1u, (uint64_t)b * K + k));
                mz = fmaxf(mz, (lg + gk) / a.temperature);
            }
        }
        mx = wave_max(mx);
        if (MODE == 1) mz = wave_max(mz);
        float se = 0.f, sz = 0.f;
        for (int k = lane; k < K; k += 64) {
            const float lg = valid ? a.logits[(int64_t)b * a.ld_logits + k] : 0.f;
            const float ex = __expf(lg - mx);
            qs[r * K + k] = ex;
            se += ex;
            if (MODE == 1) {
                const float gk = !valid ? 0.f
                    : (a.gumbel ? a.gumbel[(int64_t)b * a.ld_gumbel + k]
                                : philox_gumbel_at(a.seed, nstep, 1u, (uint64_t)b * K + k));
                const float ez = __expf((lg + gk) / a.temperature - mz);
                ws[r * K + k] = ez;
                sz += ez;
            }
        }
        se = wave_sum(se);
        if (MODE == 1) sz = wave_sum(sz);
        float kc = 0.f;
        for (int k = lane; k < K; k += 64) {
            const float q = valid ? qs[r * K + k] / se : 0.f;
            qs[r * K + k] = q;
            if (MODE == 1) ws[r * K + k] = valid ? ws[r * K + k] / sz : 0.f;
            else ws[r * K + k] = q;
            sk[r * K + k] = 0.f;
            kc += valid ? q * (__logf(q + 1e-20f) + logK) : 0.f;
        }
        kc = wave_sum(kc);
        klc_acc += kc;
        if (lane == 0) rowlv[r] = 0.f;
        // zero the pad columns of Z (they are K-dim padding of the first decoder GEMM)
        for (int d = D + lane; d < a.ld_Z; d += 64) {
            if (a.act_dtype == DMVAE_BF16) reinterpret_cast<bf16_t*>(a.Z_act)[(int64_t)b * a.ld_Z + d] = 0;
            else reinterpret_cast<float*>(a.Z_act)[(int64_t)b * a.ld_Z + d] = 0.f;
        }
    }
    __syncthreads();

    // ---- D-chunk loop ----
    for (int c = 0; c < L.nchunks; ++c) {
        const int d0 = c * DC;
        const int dc = min(DC, D - d0);
        for (int idx = tid; idx < K * dc; idx += 256) {
            const int k = idx / dc, d = idx - k * dc;
            const float pmv = a.prior_means[(int64_t)k * D + d0 + d];
            const float plv = a.prior_log_vars[(int64_t)k * D + d0 + d];
            t1[k * DC + d] = pmv;
            t2[k * DC + d] = (MODE == 0) ? __expf(-plv) : plv;
        }
        __syncthreads();

        // phase 1: one wave per row
        for (int r = wave; r < RB; r += 4) {
            const int b = row0 + r;
            const bool valid = b < a.B;
            float mu[LAT_DS], e[LAT_DS], lvv[LAT_DS];
            bool ok[LAT_DS];
            float lvsum = 0.f;
#pragma unroll
            for (int i = 0; i < LAT_DS; ++i) {
                const int d = lane + 64 * i;
                ok[i] = d < dc;
                mu[i] = 0.f; e[i] = 0.f; lvv[i] = 0.f;
                if (ok[i]) {
                    const int dg = d0 + d;
                    float z = 0.f, cl = 0.f;
                    if (valid) {
                        mu[i] = a.mean[(int64_t)b * a.ld_mean + dg];
                        lvv[i] = a.log_var[(int64_t)b * a.ld_log_var + dg];
                        e[i] = __expf(lvv[i]);
                        const float sd = __expf(0.5f * lvv[i]);
                        const float ep = a.eps ? a.eps[(int64_t)b * a.ld_eps + dg]
                                               : philox_normal_at(a.seed, nstep, 0u, (uint64_t)b * D + dg);
                        z = mu[i] + sd * ep;
                        cl = ep * 0.5f * sd;
                        lvsum += lvv[i];
                    }
                    if (a.act_dtype == DMVAE_BF16) reinterpret_cast<bf16_t*>(a.Z_act)[(int64_t)b * a.ld_Z + dg] = f2bf(z);
                    else reinterpret_cast<float*>(a.Z_act)[(int64_t)b * a.ld_Z + dg] = z;
                    if (a.Z_f32) a.Z_f32[(int64_t)b * a.ld_Zf + dg] = z;
                    a.clv[(int64_t)b * a.ld_g + dg] = cl;
                }
            }
            if (MODE == 0) {
                float gm[LAT_DS] = {0.f, 0.f, 0.f, 0.f}, A[LAT_DS] = {0.f, 0.f, 0.f, 0.f};
                for (int k = 0; k < K; ++k) {
                    const float wk = ws[r * K + k];
                    float part = 0.f;
#pragma unroll
                    for (int i = 0; i < LAT_DS; ++i) {
                        if (ok[i]) {
                            const int d = lane + 64 * i;
                            const float ipk = t2[k * DC + d];
                            const float diff = mu[i] - t1[k * DC + d];
                            part += (e[i] + diff * diff) * ipk;
                            gm[i] += wk * diff * ipk;
                            A[i] += wk * ipk;
                        }
                    }
                    part = wave_sum(part);
                    if (lane == 0) sk[r * K + k] += part;
                }
                lvsum = wave_sum(lvsum);
                if (lane == 0) rowlv[r] += lvsum;
#pragma unroll
                for (int i = 0; i < LAT_DS; ++i) {
                    if (ok[i]) {
                        const int d = lane + 64 * i, dg = d0 + d;
                        a.gmu[(int64_t)b * a.ld_g + dg] = valid ? rB * gm[i] : 0.f;
                        a.glv[(int64_t)b * a.ld_g + dg] = valid ? rB2 * (e[i] * A[i] - 1.f) : 0.f;
                        r1[r * DC + d] = mu[i];
                        r2[r * DC + d] = e[i];
                    }
                }
            } else {
                float bm[LAT_DS] = {0.f, 0.f, 0.f, 0.f}, bl[LAT_DS] = {0.f, 0.f, 0.f, 0.f};
                for (int k = 0; k < K; ++k) {
                    const float wk = ws[r * K + k];
#pragma unroll
                    for (int i = 0; i < LAT_DS; ++i) {
                        if (ok[i]) {
                            const int d = lane + 64 * i;
                            bm[i] += wk * t1[k * DC + d];
                            bl[i] += wk * t2[k * DC + d];
                        }
                    }
                }
                float dbm[LAT_DS], dbl[LAT_DS], integ = 0.f;
#pragma unroll
                for (int i = 0; i < LAT_DS; ++i) {
                    dbm[i] = 0.f; dbl[i] = 0.f;
                    if (ok[i]) {
                        const int d = lane + 64 * i, dg = d0 + d;
                        const float ib = __expf(-bl[i]);
                        const float diff = mu[i] - bm[i];
                        const float gmu = rB * diff * ib;
                        if (valid) {
                            integ += bl[i] - lvv[i] - 1.f + (e[i] + diff * diff) * ib;
                            dbm[i] = -gmu;
                            dbl[i] = rB2 * (1.f - (e[i] + diff * diff) * ib);
                        }
                        a.gmu[(int64_t)b * a.ld_g + dg] = valid ? gmu : 0.f;
                        a.glv[(int64_t)b * a.ld_g + dg] = valid ? rB2 * (e[i] * ib - 1.f) : 0.f;
                        r1[r * DC + d] = dbm[i];
                        r2[r * DC + d] = dbl[i];
                    }
                }
                integ = wave_sum(integ);
                if (lane == 0) rowlv[r] += integ;
                for (int k = 0; k < K; ++k) {
                    float part = 0.f;
#pragma unroll
                    for (int i = 0; i < LAT_DS; ++i) {
                        if (ok[i]) {
                            const int d = lane + 64 * i;
                            part += dbm[i] * t1[k * DC + d] + dbl[i] * t2[k * DC + d];
                        }
                    }
                    part = wave_sum(part);
                    if (lane == 0) sk[r * K + k] += part;
                }
            }
        }
        __syncthreads();

        // phase 2: prior-table gradient partials of this block, threads over (k, d)
        for (int idx = tid; idx < K * dc; idx += 256) {
            const int k = idx / dc, d = idx - k * dc;
            float a1 = 0.f, a2 = 0.f;
            if (MODE == 0) {
                const float pmv = t1[k * DC + d], ipv = t2[k * DC + d];
                for (int r = 0; r < RB; ++r) {
                    const float wk = ws[r * K + k];
                    const float diff = r1[r * DC + d] - pmv;
                    a1 += wk * diff;
                    a2 += wk * (1.f - (r2[r * DC + d] + diff * diff) * ipv);
                }
                a1 = -rB * ipv * a1;
                a2 = rB2 * a2;
            } else {
                for (int r = 0; r < RB; ++r) {
                    const float wk = ws[r * K + k];
                    a1 += wk * r1[r * DC + d];
                    a2 += wk * r2[r * DC + d];
                }
            }
            float* o = a.dprior_partials + (int64_t)blockIdx.x * 2 * K * D;
            o[(int64_t)k * D + d0 + d] = a1;
            o[(int64_t)K * D + (int64_t)k * D + d0 + d] = a2;
        }
        __syncthreads();
    }

    // ---- finalize rows: KL_Z, dlogits ----
    for (int r = wave; r < RB; r += 4) {
        const int b = row0 + r;
        const bool valid = b < a.B;
        const float rl = rowlv[r];
        // pass 1: weighted sums
        float s_wdw = 0.f, s_qdq = 0.f, klz = 0.f;
        for (int k = lane; k < K; k += 64) {
            const float w = ws[r * K + k], q = qs[r * K + k];
            float dw;
            if (MODE == 0) {
                const float t = sk[r * K + k] + ck[k] - rl - (float)D;
                klz += 0.5f * w * t;
                dw = rB2 * t;
            } else {
                dw = sk[r * K + k];
            }
            const float dq = rB * (__logf(q + 1e-20f) + q / (q + 1e-20f) + logK);
            s_wdw += w * dw;
            s_qdq += q * dq;
        }
        s_wdw = wave_sum(s_wdw);
        s_qdq = wave_sum(s_qdq);
        klz = wave_sum(klz);
        if (MODE == 1) klz = 0.5f * rl;
        if (valid) klz_acc += klz;
        const float wscale = (MODE == 1) ? 1.0f / a.temperature : 1.0f;
        for (int k = lane; k < a.ld_dl; k += 64) {
            float dl = 0.f;
            if (k < K && valid) {
                const float w = ws[r * K + k], q = qs[r * K + k];
                float dw;
                if (MODE == 0) dw = rB2 * (sk[r * K + k] + ck[k] - rl - (float)D);
                else dw = sk[r * K + k];
                const float dq = rB * (__logf(q + 1e-20f) + q / (q + 1e-20f) + logK);
                dl = q * (dq - s_qdq) + wscale * w * (dw - s_wdw);
            }
            if (a.act_dtype == DMVAE_BF16) reinterpret_cast<bf16_t*>(a.dlogits_act)[(int64_t)b * a.ld_dl + k] = f2bf(dl);
            else reinterpret_cast<float*>(a.dlogits_act)[(int64_t)b * a.ld_dl + k] = dl;
            if (a.weights && k < K) a.weights[(int64_t)b * a.ld_w + k] = ws[r * K + k];
        }
    }
    // block loss partials, fixed order
    if (lane == 0) { red[wave] = klz_acc; red[4 + wave] = klc_acc; }
    __syncthreads();
    if (tid == 0) {
        a.loss_partials[2 * blockIdx.x + 0] = (red[0] + red[1]) + (red[2] + red[3]);
        a.loss_partials[2 * blockIdx.x + 1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}

static void latent_geometry(int B_pad, int D, int K, int& RB, int& DC, int& nchunks, size_t& lds_bytes) {
    const int want = (B_pad + 511) / 512;      // aim at ~512 blocks
    RB = 4;                                      // power of two <= 64: divides B_pad (multiple of 64)
    while (RB < want && RB < 64) RB *= 2;
    DC = 256;
    while (DC > 64 && DC / 2 >= D) DC /= 2;          // no wider than D rounded up to 64
    auto bytes = [&](int dc) {
        return sizeof(float) * ((size_t)2 * K * dc + K + (size_t)3 * RB * K + (size_t)2 * RB * dc + RB + 8);
    };
    while (DC > 16 && bytes(DC) > 60 * 1024) DC /= 2;
    nchunks = (D + DC - 1) / DC;
    lds_bytes = bytes(DC);
}

int latent_nblocks(int B_pad, int D, int K) {
    int RB, DC, nc; size_t lb;
    latent_geometry(B_pad, D, K, RB, DC, nc, lb);
    return (B_pad + RB - 1) / RB;
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
    const int nblk = (a->B_pad + L.RB - 1) / L.RB;
    static bool attr_set[2] = {false, false};
    if (lb > 64 * 1024 && !attr_set[a->mode]) {
        if (a->mode == 0) hipFuncSetAttribute((const void*)latent_fwd_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        else hipFuncSetAttribute((const void*)latent_fwd_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr_set[a->mode] = true;
    }
    const double bytes = 4.0 * ((double)a->B * (6.0 * a->D + 3.0 * a->K) + 2.0 * a->K * a->D * (nblk + 1));
    ProfScope ps(s, a->mode == 0 ? "latent_fwd_exact" : "latent_fwd_relaxed", 6.0 * a->B * (double)a->K * a->D, bytes);
    if (a->mode == 0) hipLaunchKernelGGL(latent_fwd_kernel<0>, dim3(nblk), dim3(256), lb, s, L);
    else hipLaunchKernelGGL(latent_fwd_kernel<1>, dim3(nblk), dim3(256), lb, s, L);
    return check_launch("latent_fwd");
}

}  // namespace dmvae
