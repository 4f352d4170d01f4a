// The latent stage of one block of rows as a device function, shared by latent_fwd_kernel (latent.hip: its inputs come from global
// memory) and heads_latent_kernel (heads_latent.hip: the workgroup computes them itself -- the two head GEMMs -- and hands them over in LDS).
// priors.py:86-201, base_models.py:249: see latent.hip.
#pragma once
#include "kernels.h"

namespace dmvae {

struct LatentLaunch {
    dmvae_latent_args a;
    int RB;       // rows per block (multiple of 16)
    int DC;       // columns per chunk = 16 * DSL
    int nchunks;
};

__device__ __forceinline__ float row_sum16(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 16);
    return v;
}
__device__ __forceinline__ float row_max16(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 16));
    return v;
}

#include "measure.h"      // MEAS_LAT_STAMP: phase stamps of block 0 in the abl7 measurement build (tools/latent_time.py); empty in the product build

// The block's mean / log_var / logits rows when they come from LDS instead of global memory (FUSED: heads_latent.hip, where the workgroup has just
// computed them): p[r * ld + d] = mean, p[r * ld + lv_col + d] = log_var, p[r * ld + lg_col + k] = logits of the block's row r.
// LDS pointers carry their address space in the TYPE: as plain float* (generic pointers the optimiser resolves to LDS only late) the same code compiled
// to 20 % more waits -- loads no longer batched across LDS traffic that might, for all the early passes knew, alias global memory -- and ran 7 % slower.
typedef __attribute__((address_space(3))) float lds_f;
struct LatentTile { const lds_f* p; int ld, lv_col, lg_col; };

// One block of the latent stage: 256 threads, rows [blk * RB, blk * RB + RB).  lds: the block's latent arrays (latent_lds_bytes; with lds_rows: its
// first latent_lds_head_bytes there, the remaining latent_lds_rows_bytes at lds_rows).
// FUSED = false: latent_fwd_kernel below (mean / log_var / logits from global memory; top / mid do nothing).
// FUSED = true (heads_latent.hip): top() runs first of all (it issues the first operand tiles of the head GEMMs), mid() runs after the
// prior tables are staged and before the first use of the heads' outputs (it runs the GEMMs' K loop and leaves `tile` complete, behind a
// barrier; it returns false in a workgroup that has nothing further to do: a K slice that was not the last to arrive).  The arithmetic on the rows is the same code in the same order either way: same bits.
template <int MODE, int DSL, bool FUSED, class TOP, class MID>   // MODE 0 exact, 1 relaxed; DSL = columns per lane per chunk (DC = 16*DSL)
__device__ __forceinline__ void latent_body(const LatentLaunch& L, lds_f* lds, lds_f* lds_rows, const int blk, const LatentTile tile, TOP&& top, MID&& mid) {
    const dmvae_latent_args& a = L.a;
    MEAS_LAT_STAMP(0);
    top();          // (first of all: the reads of the step state below are a memory round trip the compiler waits for before anything that follows them)
    const int K = a.K, D = a.D, RB = L.RB;
    constexpr int DC = 16 * DSL, DCP = DC + 1;   // +1: rows of one column land on distinct banks (phase 1b / 2)
    lds_f* t1 = lds;                  // [K][DCP] prior means
    lds_f* t2 = t1 + K * DCP;         // [K][DCP] exp(-prior_log_var) (exact) | prior_log_var (relaxed)
    lds_f* ck = t2 + K * DCP;         // [K]      sum_d prior_log_var
    lds_f* ws = ck + K;               // [RB][K]  mixture weights (softmax or zeta); 0 for pad rows
    lds_f* qs = ws + RB * K;          // [RB][K]  softmax(logits)
    lds_f* sk = qs + RB * K;          // [RB][K]  exact: sum_d (e+(mu-pm)^2)*ip ; relaxed: dLoss/dzeta
    // (lds_rows != nullptr: the arrays from here on -- first written AFTER mid() -- live there instead of behind sk: the fused kernel overlays them on its idle ring)
    lds_f* r1 = lds_rows ? lds_rows : sk + RB * K;          // [RB][DCP] exact: mu   | relaxed: dLoss/d(bar mean)
    lds_f* r2 = r1 + RB * DCP;        // [RB][DCP] exact: e^lv | relaxed: dLoss/d(bar log_var)
    lds_f* rowlv = r2 + RB * DCP;     // [RB] sum_d log_var (exact) | sum_d relaxed KL integrand
    lds_f* red = rowlv + RB;          // [32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15;                       // lane within the row group
    const int rsub = wave * 4 + (lane >> 4);        // row slot of this 16-lane group within a pass of 16 rows
    const int row0 = blk * RB;
    const dmvae_state* st = reinterpret_cast<const dmvae_state*>(a.state);
    const float klr = st ? st->kl_ratio : a.kl_ratio;
    const uint64_t nstep = st ? st->noise_step : a.noise_step;
    const float rB = klr * a.inv_B;          // r / B
    const float rB2 = 0.5f * rB;             // r / (2B)
    const float logK = __logf((float)K);
    const int dc0 = min(DC, D);

    // ---- early global loads: tables of chunk 0, and this group's first row (logits, mean, log_var, eps)
    constexpr int TPF = 4, KF = 4;
    float pm_pre[TPF], plv_pre[TPF];
#pragma unroll
    for (int j = 0; j < TPF; ++j) {
        const int idx = tid + 256 * j;
        pm_pre[j] = 0.f; plv_pre[j] = 0.f;
        if (idx < K * dc0) {
            const int k = idx / dc0, d = idx - k * dc0;
            pm_pre[j] = a.prior_means[(int64_t)k * D + d];
            plv_pre[j] = a.prior_log_vars[(int64_t)k * D + d];
        }
    }
    const int b0 = row0 + rsub;
    const bool valid0 = b0 < a.B;
    float lg_pre[KF], mu_pre[DSL], lv_pre[DSL], ep_pre[DSL];
#pragma unroll
    for (int i = 0; i < KF; ++i) {
        const int k = lr + 16 * i;
        lg_pre[i] = 0.f;
        if constexpr (!FUSED) lg_pre[i] = (valid0 && k < K) ? a.logits[(int64_t)b0 * a.ld_logits + k] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < DSL; ++i) {
        const int d = lr + 16 * i;
        mu_pre[i] = 0.f; lv_pre[i] = 0.f; ep_pre[i] = 0.f;
        if (valid0 && d < dc0) {
            if constexpr (!FUSED) {
                mu_pre[i] = a.mean[(int64_t)b0 * a.ld_mean + d];
                lv_pre[i] = a.log_var[(int64_t)b0 * a.ld_log_var + d];
            }
            if (a.eps) ep_pre[i] = a.eps[(int64_t)b0 * a.ld_eps + d];
        }
    }

    // FUSED: the K loop of the head GEMMs runs NOW -- behind the requests above and those of c_k's summands (when one pass covers them: K <= 16, D <= 64), in
    // front of everything that consumes them: the tables reach LDS after the K loop, from registers, instead of holding its start back by their round
    // trips (3.4 us into the kernel, profiles/r05_heads_latent_phases.txt); a K slice that is not the last to arrive never stages them at all
    const bool ck_hoist = FUSED && K <= 16 && D <= 64;
    float ck_pre[4] = {0.f, 0.f, 0.f, 0.f};
    if (ck_hoist) {
        const int k = tid >> 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int d = lr + 16 * i;
            ck_pre[i] = a.prior_log_vars[(int64_t)(k < K ? k : 0) * D + (d < D ? d : 0)];
        }
    }
    MEAS_LAT_STAMP(1);
    if constexpr (FUSED) {
        MEAS_LAT_STAMP(11);
        if (!mid()) return;
        MEAS_LAT_STAMP(12);
    }
    // ---- prologue: c_k, tables of chunk 0 into LDS, per-row softmax / zeta, KL_C ----
    for (int k = tid >> 4; k < K; k += 16) {
        // four loads in flight per lane (index clamped, value selected): one load per trip of `for d: s += table[d]` was a memory round trip
        // per trip, on the path to phase 1 -- most of the "prior-table staging 2.0 us" of profiles/r04_latent_phases.txt.  Same order of additions.
        float s = 0.f;
        for (int d0 = lr; d0 < D; d0 += 64) {
            float t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int d = d0 + 16 * i;
                t[i] = ck_hoist ? ck_pre[i] : a.prior_log_vars[(int64_t)k * D + (d < D ? d : 0)];
                t[i] = d < D ? t[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) s += t[i];
        }
        s = row_sum16(s);
        if (lr == 0) ck[k] = s;
    }
#pragma unroll
    for (int j = 0; j < TPF; ++j) {
        const int idx = tid + 256 * j;
        if (idx < K * dc0) {
            const int k = idx / dc0, d = idx - k * dc0;
            t1[k * DCP + d] = pm_pre[j];
            t2[k * DCP + d] = (MODE == 0) ? __expf(-plv_pre[j]) : plv_pre[j];
        }
    }
    for (int idx = tid + 256 * TPF; idx < K * dc0; idx += 256) {
        const int k = idx / dc0, d = idx - k * dc0;
        const float plv = a.prior_log_vars[(int64_t)k * D + d];
        t1[k * DCP + d] = a.prior_means[(int64_t)k * D + d];
        t2[k * DCP + d] = (MODE == 0) ? __expf(-plv) : plv;
    }
    if constexpr (!FUSED) { if (!mid()) return; }      // (nothing, unfused; FUSED: mid() ran above -- K slices: only the last slice of a block to arrive goes on, heads_latent.hip)
    float klc_acc = 0.f, klz_acc = 0.f;   // per 16-lane group (all lanes of the group hold the same value)
    for (int r = rsub; r < RB; r += 16) {
        const bool first = !FUSED && r == rsub;
        const int b = row0 + r;
        const bool valid = b < a.B;
        // this lane's clusters k = lr + 16 i: the first KF logits (and Gumbel draws) live in registers
        float lg[KF], gk[KF];
        if (!first) {                 // (later rows of the block: one batch of unconditional loads, row and column clamped -- see phase 1a)
            const int64_t bb = valid ? b : 0;
#pragma unroll
            for (int i = 0; i < KF; ++i) {
                const int k = lr + 16 * i;
                if constexpr (FUSED) lg[i] = tile.p[r * tile.ld + tile.lg_col + (k < K ? k : 0)];
                else lg[i] = a.logits[bb * a.ld_logits + (k < K ? k : 0)];
            }
        }
#pragma unroll
        for (int i = 0; i < KF; ++i) {
            const int k = lr + 16 * i;
            lg[i] = first ? lg_pre[i] : ((valid && k < K) ? lg[i] : 0.f);
            gk[i] = 0.f;
            if (MODE == 1 && valid && k < K)
                gk[i] = a.gumbel ? a.gumbel[(int64_t)b * a.ld_gumbel + k] : philox_gumbel_at(a.seed, nstep, 1u, (uint64_t)b * K + k);
        }
        auto for_k = [&](auto&& fn) {     // fn(k, logit, gumbel) over this lane's clusters
#pragma unroll
            for (int i = 0; i < KF; ++i) {
                const int k = lr + 16 * i;
                if (k < K) fn(k, lg[i], gk[i]);
            }
            for (int k = lr + 16 * KF; k < K; k += 16) {
                float l = 0.f;
                if constexpr (FUSED) l = valid ? tile.p[r * tile.ld + tile.lg_col + k] : 0.f;
                else l = valid ? a.logits[(int64_t)b * a.ld_logits + k] : 0.f;
                float g = 0.f;
                if (MODE == 1 && valid)
                    g = a.gumbel ? a.gumbel[(int64_t)b * a.ld_gumbel + k] : philox_gumbel_at(a.seed, nstep, 1u, (uint64_t)b * K + k);
                fn(k, l, g);
            }
        };
        float mx = -INFINITY, mz = -INFINITY;
        for_k([&](int, float l, float g) {
            mx = fmaxf(mx, l);
            if (MODE == 1) mz = fmaxf(mz, (l + g) / a.temperature);
        });
        mx = row_max16(mx);
        if (MODE == 1) mz = row_max16(mz);
        float se = 0.f, sz = 0.f;
        for_k([&](int k, float l, float g) {
            const float ex = __expf(l - mx);
            qs[r * K + k] = ex;
            se += ex;
            if (MODE == 1) {
                const float ez = __expf((l + g) / a.temperature - mz);
                ws[r * K + k] = ez;
                sz += ez;
            }
        });
        se = row_sum16(se);
        if (MODE == 1) sz = row_sum16(sz);
        float kc = 0.f;
        for (int k = lr; k < K; k += 16) {
            const float q = valid ? qs[r * K + k] / se : 0.f;
            qs[r * K + k] = q;
            if (MODE == 1) ws[r * K + k] = valid ? ws[r * K + k] / sz : 0.f;
            else ws[r * K + k] = q;
            sk[r * K + k] = 0.f;
            kc += valid ? q * (__logf(q + 1e-20f) + logK) : 0.f;
        }
        kc = row_sum16(kc);
        klc_acc += kc;
        if (lr == 0) rowlv[r] = 0.f;
        // zero the pad columns of Z (they are K-dim padding of the first decoder GEMM)
        for (int d = D + lr; d < a.ld_Z; d += 16) {
            if (a.act_dtype == DMVAE_BF16) reinterpret_cast<bf16_t*>(a.Z_act)[(int64_t)b * a.ld_Z + d] = 0;
            else reinterpret_cast<float*>(a.Z_act)[(int64_t)b * a.ld_Z + d] = 0.f;
        }
    }
    lds_barrier();
    MEAS_LAT_STAMP(2);

    // ---- D-chunk loop ----
    for (int c = 0; c < L.nchunks; ++c) {
        const int d0 = c * DC;
        const int dc = min(DC, D - d0);
        if (c > 0) {                  // chunk 0 was staged in the prologue
            for (int idx = tid; idx < K * dc; idx += 256) {
                const int k = idx / dc, d = idx - k * dc;
                const float pmv = a.prior_means[(int64_t)k * D + d0 + d];
                const float plv = a.prior_log_vars[(int64_t)k * D + d0 + d];
                t1[k * DCP + d] = pmv;
                t2[k * DCP + d] = (MODE == 0) ? __expf(-plv) : plv;
            }
            lds_barrier();
        }

        // phase 1a: sixteen lanes per row, column d = lr + 16*i; sums over k in private accumulators
        for (int r = rsub; r < RB; r += 16) {
            const bool first = r == rsub && c == 0;           // (its eps quad -- and, unfused, its mean / log_var -- were fetched at entry)
            const int b = row0 + r;
            const bool valid = b < a.B;
            float mu[DSL], e[DSL], lvv[DSL];
            bool ok[DSL];
            float lvsum = 0.f;
            // device noise: one Philox block per four of this lane's columns, keyed by (row, chunk, lane, group)
            float nz[DSL < 4 ? 4 : DSL];
            if (!a.eps && valid) {
#pragma unroll
                for (int j = 0; j < (DSL + 3) / 4; ++j) {
                    float q4[4];
                    philox_normal4(a.seed, nstep, 0u, ((((uint64_t)b * L.nchunks + c) * 16 + lr) * ((DSL + 3) / 4)) + j, q4);
#pragma unroll
                    for (int u = 0; u < 4; ++u) nz[4 * j + u] = q4[u];
                }
            }
            MEAS_LAT_STAMP(7);
            // this row's mean / log_var / eps: the block's first row was fetched at kernel entry; every later row (32 and 64 rows per block:
            // batches >= 16 384) in ONE batch of unconditional loads (row and column clamped, values used only where valid): inside the
            // `if (ok) { if (valid) {` below each of them was a memory round trip of its own, waited for at the join
            float mu_r[DSL], lv_r[DSL], ep_r[DSL];
            if constexpr (FUSED) {
#pragma unroll
                for (int i = 0; i < DSL; ++i) {
                    const int d = lr + 16 * i;
                    const int dgc = d0 + (d < dc ? d : 0);
                    mu_r[i] = tile.p[r * tile.ld + dgc];
                    lv_r[i] = tile.p[r * tile.ld + tile.lv_col + dgc];
                    ep_r[i] = first ? ep_pre[i] : (a.eps ? a.eps[(valid ? (int64_t)b : 0) * a.ld_eps + dgc] : 0.f);
                }
            } else if (first) {
#pragma unroll
                for (int i = 0; i < DSL; ++i) { mu_r[i] = mu_pre[i]; lv_r[i] = lv_pre[i]; ep_r[i] = ep_pre[i]; }
            } else {
                const int64_t bb = valid ? b : 0;               // (row 0 exists)
#pragma unroll
                for (int i = 0; i < DSL; ++i) {
                    const int d = lr + 16 * i;
                    const int dgc = d0 + (d < dc ? d : 0);
                    mu_r[i] = a.mean[bb * a.ld_mean + dgc];
                    lv_r[i] = a.log_var[bb * a.ld_log_var + dgc];
                    ep_r[i] = a.eps ? a.eps[bb * a.ld_eps + dgc] : 0.f;
                }
            }
#pragma unroll
            for (int i = 0; i < DSL; ++i) {
                const int d = lr + 16 * i;
                ok[i] = d < dc;
                mu[i] = 0.f; e[i] = 0.f; lvv[i] = 0.f;
                if (ok[i]) {
                    const int dg = d0 + d;
                    float z = 0.f, cl = 0.f;
                    if (valid) {
                        mu[i] = mu_r[i];
                        lvv[i] = lv_r[i];
                        e[i] = __expf(lvv[i]);
                        const float sd = __expf(0.5f * lvv[i]);
                        const float ep = a.eps ? ep_r[i] : nz[i];
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
            MEAS_LAT_STAMP(8);
            if (MODE == 0) {
                float gm[DSL], A[DSL];
#pragma unroll
                for (int i = 0; i < DSL; ++i) { gm[i] = 0.f; A[i] = 0.f; }
                // one wave per SIMD: nothing hides an LDS round trip, so every loop below is unrolled
                // to keep a batch of independent reads in flight
                // (columns past the chunk read table padding and are never stored: no branch in the loop, so the LDS reads of an
                //  unrolled trip are issued together -- with `if (ok[i])` around the body every (k, column) pair was its own LDS
                //  round trip: 1.8 us of the 4 us this phase took at D = 64, K = 10)
#pragma unroll 5
                for (int k = 0; k < K; ++k) {
                    const float wk = ws[r * K + k];
#pragma unroll
                    for (int i = 0; i < DSL; ++i) {
                        const int d = lr + 16 * i;
                        const float ipk = t2[k * DCP + d];
                        gm[i] += wk * (mu[i] - t1[k * DCP + d]) * ipk;
                        A[i] += wk * ipk;
                    }
                }
                MEAS_LAT_STAMP(9);
                lvsum = row_sum16(lvsum);
                if (lr == 0) rowlv[r] += lvsum;
#pragma unroll
                for (int i = 0; i < DSL; ++i) {
                    if (ok[i]) {
                        const int d = lr + 16 * i, dg = d0 + d;
                        a.gmu[(int64_t)b * a.ld_g + dg] = valid ? rB * gm[i] : 0.f;
                        a.glv[(int64_t)b * a.ld_g + dg] = valid ? rB2 * (e[i] * A[i] - 1.f) : 0.f;
                        r1[r * DCP + d] = mu[i];
                        r2[r * DCP + d] = e[i];
                    }
                }
            } else {
                float bm[DSL], bl[DSL];
#pragma unroll
                for (int i = 0; i < DSL; ++i) { bm[i] = 0.f; bl[i] = 0.f; }
#pragma unroll 5
                for (int k = 0; k < K; ++k) {
                    const float wk = ws[r * K + k];
#pragma unroll
                    for (int i = 0; i < DSL; ++i) {
                        const int d = lr + 16 * i;
                        bm[i] += wk * t1[k * DCP + d];
                        bl[i] += wk * t2[k * DCP + d];
                    }
                }
                float integ = 0.f;
#pragma unroll
                for (int i = 0; i < DSL; ++i) {
                    if (ok[i]) {
                        const int d = lr + 16 * i, dg = d0 + d;
                        const float ib = __expf(-bl[i]);
                        const float diff = mu[i] - bm[i];
                        const float gmu = rB * diff * ib;
                        float dbm = 0.f, dbl = 0.f;
                        if (valid) {
                            integ += bl[i] - lvv[i] - 1.f + (e[i] + diff * diff) * ib;
                            dbm = -gmu;
                            dbl = rB2 * (1.f - (e[i] + diff * diff) * ib);
                        }
                        a.gmu[(int64_t)b * a.ld_g + dg] = valid ? gmu : 0.f;
                        a.glv[(int64_t)b * a.ld_g + dg] = valid ? rB2 * (e[i] * ib - 1.f) : 0.f;
                        r1[r * DCP + d] = dbm;
                        r2[r * DCP + d] = dbl;
                    }
                }
                integ = row_sum16(integ);
                if (lr == 0) rowlv[r] += integ;
            }
        }
        MEAS_LAT_STAMP(10);
        lds_barrier();
        MEAS_LAT_STAMP(3);

        // phase 1b: sixteen lanes per row, cluster k = lr + 16*j; sums over d in private accumulators
        for (int r = rsub; r < RB; r += 16) {
            const lds_f* x1 = r1 + r * DCP;
            const lds_f* x2 = r2 + r * DCP;
            for (int k = lr; k < K; k += 16) {
                const lds_f* p1 = t1 + k * DCP;
                const lds_f* p2 = t2 + k * DCP;
                float s[4] = {0.f, 0.f, 0.f, 0.f};
                int d = 0;
                for (; d + 7 < dc; d += 8) {      // 32 independent LDS reads per trip
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (MODE == 0) {
                            const float f = x1[d + u] - p1[d + u];
                            s[u & 3] += (x2[d + u] + f * f) * p2[d + u];
                        } else {
                            s[u & 3] += x1[d + u] * p1[d + u] + x2[d + u] * p2[d + u];
                        }
                    }
                }
                for (; d < dc; ++d) {
                    if (MODE == 0) {
                        const float f = x1[d] - p1[d];
                        s[0] += (x2[d] + f * f) * p2[d];
                    } else {
                        s[0] += x1[d] * p1[d] + x2[d] * p2[d];
                    }
                }
                sk[r * K + k] += (s[0] + s[1]) + (s[2] + s[3]);     // (r, k) belongs to this lane alone
            }
        }

        MEAS_LAT_STAMP(4);
        // phase 2: prior-table gradient partials of this block, threads over (k, d)
        for (int idx = tid; idx < K * dc; idx += 256) {
            const int k = idx / dc, d = idx - k * dc;
            float a1 = 0.f, a2 = 0.f;
            if (MODE == 0) {
                const float pmv = t1[k * DCP + d], ipv = t2[k * DCP + d];
#pragma unroll 8
                for (int r = 0; r < RB; ++r) {
                    const float wk = ws[r * K + k];
                    const float diff = r1[r * DCP + d] - pmv;
                    a1 += wk * diff;
                    a2 += wk * (1.f - (r2[r * DCP + d] + diff * diff) * ipv);
                }
                a1 = -rB * ipv * a1;
                a2 = rB2 * a2;
            } else {
#pragma unroll 8
                for (int r = 0; r < RB; ++r) {
                    const float wk = ws[r * K + k];
                    a1 += wk * r1[r * DCP + d];
                    a2 += wk * r2[r * DCP + d];
                }
            }
            float* o = a.dprior_partials + (int64_t)blk * 2 * K * D;
            o[(int64_t)k * D + d0 + d] = a1;
            o[(int64_t)K * D + (int64_t)k * D + d0 + d] = a2;
        }
        lds_barrier();
        MEAS_LAT_STAMP(5);
    }

    // ---- finalize rows: KL_Z, dlogits ----
    for (int r = rsub; r < RB; r += 16) {
        const int b = row0 + r;
        const bool valid = b < a.B;
        const float rl = rowlv[r];
        float s_wdw = 0.f, s_qdq = 0.f, klz = 0.f;
        for (int k = lr; k < K; k += 16) {
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
        s_wdw = row_sum16(s_wdw);
        s_qdq = row_sum16(s_qdq);
        klz = row_sum16(klz);
        if (MODE == 1) klz = 0.5f * rl;
        if (valid) klz_acc += klz;
        const float wscale = (MODE == 1) ? 1.0f / a.temperature : 1.0f;
        for (int k = lr; k < a.ld_dl; k += 16) {
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
    // block loss partials, fixed order over the 16 row groups
    if (lr == 0) { red[rsub] = klz_acc; red[16 + rsub] = klc_acc; }
    lds_barrier();
    if (tid == 0) {
        float z = 0.f, c = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) { z += red[i]; c += red[16 + i]; }
        a.loss_partials[2 * blk + 0] = z;
        a.loss_partials[2 * blk + 1] = c;
    }
    MEAS_LAT_STAMP(6);
}


// ... split into what is written before mid() (tables, c_k; the per-row weights are written after it but sized with them) and the row arrays written after it
__host__ __device__ inline size_t latent_lds_head_bytes(int K, int RB, int dc) { return sizeof(float) * ((size_t)2 * K * (dc + 1) + K + (size_t)3 * RB * K); }
__host__ __device__ inline size_t latent_lds_rows_bytes(int RB, int dc) { return sizeof(float) * ((size_t)2 * RB * (dc + 1) + RB + 32); }
// LDS floats the block's latent arrays take (latent_body's layout): tables 2 K (DC + 1) | c_k K | weights, softmax, sk 3 RB K | rows 2 RB (DC + 1) | RB | 32
__host__ __device__ inline size_t latent_lds_bytes(int K, int RB, int dc) {
    return sizeof(float) * ((size_t)2 * K * (dc + 1) + K + (size_t)3 * RB * K + (size_t)2 * RB * (dc + 1) + RB + 32);
}

}  // namespace dmvae
