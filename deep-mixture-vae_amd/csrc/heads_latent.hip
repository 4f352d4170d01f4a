// Heads forward + latent stage as ONE launch (bf16 plans, batches of at most one round of the chip: B_pad / 16 <= 256 workgroups).
//
// Replaces, for a block of 16 rows at a time, the two launches
//     [mean | log_var] = hz . W_mv + b_mv,   logits = hc . W_lg + b_lg          code/base_models.py:229-249  (tf.layers.dense x 3)
//     softmax, reparameterisation, mixture KL, categorical KL and all their gradients                code/priors.py:86-201
// i.e. gemm_bf16_grouped_kernel<64, 64, FWD, BIAS_F32> followed by latent_fwd_kernel, with an f32 [B, 2 D + K] round trip through HBM between
// them.  The stage is row-independent (no cross-row coupling before the per-block partial sums), so a workgroup that owns 16 rows can run
// both head problems for them (K = 2048 each, accumulators 2 D + K columns wide) and then the latent phases on rows that never left the CU:
// one launch boundary, the round trip and the latent kernel's entry latency less (profiles/r04_latent_phases.txt).
//
//   * 256 threads = 4 waves.  One K tile (64 deep) of the ring = {A: 16 rows of hz + 16 rows of hc (one 32-row k-contiguous tile: 4 KiB),
//     W_mv: 64 x 2 Dp (n-contiguous, 16 KiB per 128 columns), W_lg: 64 x 64 (8 KiB)}: 28 KiB (Dp = 64) / 44 KiB (Dp = 128), filled by LDS-DMA
//     (global_load_lds_dwordx4), NSTAGE = 4 / 3 slots, counted vmcnt, one barrier per K tile -- the loop of gemm_bf16_body (gemm_bf16.hip).
//   * wave w multiplies the [mean | log_var] column tiles [w * 2 DP, (w + 1) * 2 DP) (16 wide each) against the hz fragment and, if w < ceil(K / 16),
//     logits column tile w against the hc fragment: the same v_mfma_f32_16x16x32_bf16 with the operands in the same order and the K steps in
//     the same order as the tiles of the grouped launch it replaces -> the same bits.
//   * the f32 result (+ bias, as DMVAE_EPI_BIAS_F32) is parked in the idle ring as a [16][2 Dp + 64] tile, written to global memory for the views /
//     the backward pass exactly as before (mean / log_var / logits), and handed to latent_body (latent_body.h, FUSED): the SAME code that
//     latent_fwd_kernel runs, reading its three inputs from LDS.  The first ring tiles are requested before, and the K loop runs after, the
//     latent stage's input-independent prologue (prior tables -> LDS, c_k), which therefore hides behind the first tiles' latency.
//   * every workgroup streams the whole of both weight matrices (L2-resident: 768 KiB at D = 64): that is why the form is limited to one round
//     of the chip -- at 16 384 rows it would read them four times per CU (the grouped 64-row tiles + latent_fwd_kernel stay).
#include "gemm_tile.h"
#include "latent_body.h"

namespace dmvae {

struct HeadsLatentLaunch {
    LatentLaunch L;                      // a.mean / a.log_var / a.logits: the heads' f32 OUTPUTS here (still written: views, evaluation, tests)
    const bf16_t* hz; int64_t lda;       // [B_pad][lda]: columns [0, Hp) the z-head's hidden layer, [Hp, 2 Hp) the c-head's
    int Hp;
    const bf16_t* Wmv; int64_t ldmv;     // [Hp][ldmv >= 2 Dp], n-contiguous (the bf16 shadow)
    const bf16_t* Wlg; int64_t ldlg;     // [Hp][ldlg >= 64]
    const float* bmv; const float* blg;  // f32 biases: 2 Dp / 64 entries
    int nt_lg;                           // 16-column tiles of the logits head that hold real classes: ceil(K / 16)
    // K slices (small batches: few 16-row blocks, each a long K chain): gridDim.y workgroups per block, each over Hp / gridDim.y of the contraction; the
    // partial f32 tiles meet in slab [block][slice] of ks_ws, the last slice to arrive (ticket ks_tick[block]) adds them in ascending order, adds the
    // biases and runs the latent stage (the scheme of GemmArgs::tick, gemm_bf16.hip)
    float* ks_ws; int* ks_tick;
};

template <int DP> struct HLGeom {        // DP = Dp / 64
    static constexpr int NMV = 128 * DP;
    static constexpr int A_ELEMS = 32 * BK, MVH_ELEMS = 128 * BK, LG_ELEMS = 64 * BK;
    static constexpr int STAGE = A_ELEMS + DP * MVH_ELEMS + LG_ELEMS;        // bf16 elements per ring slot
#ifndef DMVAE_HL_NSTAGE1
#define DMVAE_HL_NSTAGE1 4      // (a measurement build may set 5: 140 KiB of ring at Dp = 64)
#endif
    static constexpr int NSTAGE = DP == 1 ? DMVAE_HL_NSTAGE1 : 3;
    static constexpr int LOADS = 1 + 4 * DP + 2;                             // LDS-DMA instructions per lane per K tile
    static constexpr int TLD = NMV + 64 + 4;                                 // floats per row of the parked f32 tile (+ 4: rows start on different banks)
    static constexpr size_t RING_BYTES = (size_t)NSTAGE * STAGE * 2;
    static constexpr size_t TILE_BYTES = (size_t)16 * TLD * 4;
};
// LDS of one workgroup: [latent tables, c_k, per-row weights (latent_lds_head_bytes, rounded up to 1 KiB)] [ring].  After the K loop the ring holds
// the latent stage's row arrays (latent_lds_rows_bytes, first written after the K loop) and, behind them, the parked f32 tile.
__host__ __device__ inline size_t hl_head_bytes(int K, int dc) { return (latent_lds_head_bytes(K, 16, dc) + 1023) & ~(size_t)1023; }
__host__ __device__ inline size_t hl_rows_bytes(int dc) { return (latent_lds_rows_bytes(16, dc) + 15) & ~(size_t)15; }
template <int DP> __host__ __device__ inline size_t hl_lds_bytes(int K, int dc) {
    const size_t over = hl_rows_bytes(dc) + HLGeom<DP>::TILE_BYTES;
    return hl_head_bytes(K, dc) + (over > HLGeom<DP>::RING_BYTES ? over : HLGeom<DP>::RING_BYTES);
}

template <int MODE, int DSL, int DP>
__global__ __launch_bounds__(256) void heads_latent_kernel(HeadsLatentLaunch H) {
    using G = HLGeom<DP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char hl_smem[];
    float* lat = reinterpret_cast<float*>(hl_smem);
    unsigned char* ring_b = hl_smem + hl_head_bytes(H.L.a.K, 16 * DSL);
    bf16_t* ring = reinterpret_cast<bf16_t*>(ring_b);
    float* lat_rows = reinterpret_cast<float*>(ring_b);                       // (after the K loop)
    float* tile = reinterpret_cast<float*>(ring_b + hl_rows_bytes(16 * DSL));  // (after the K loop)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int row0 = (int)blockIdx.x * 16;
    const int nsl = (int)gridDim.y;                       // K slices (1 = none)
    const int nk = H.Hp / BK / nsl;
    const int k0 = (int)blockIdx.y * nk;                  // first K tile of this slice
    const bool has_lg = wave < H.nt_lg;

    const bf16_t* Ag = H.hz + (int64_t)row0 * H.lda + (int64_t)k0 * BK;
    const int64_t stepMV = (int64_t)BK * H.ldmv, stepLG = (int64_t)BK * H.ldlg;

    // loop-invariant per-lane addressing.  A: one k-contiguous 32-row tile whose rows 16..31 are rows 0..15 of the c-head's columns.
    // (Every wave requests its share of every operand.  MEASURED, round 5: loaders specialised by latency class -- wave 0 only the activation tiles,
    //  first-touch data, into a 12-slot ring of their own; waves 1..3 only the weight tiles, L2 hits, 4 slots -- bit-identical and SLOWER: 32.4 us for
    //  the launch against 25.3.  A CU's vector-memory path returns in order ACROSS waves too: the deeper the first-touch requests run ahead, the
    //  longer every L2 hit queued behind them waits.  profiles/r05_heads_latent.txt)
    unsigned goA[1], goMV[4], goLG[2];
    {
        const int row = wave * 8 + (lane >> 3);
        goA[0] = 2u * (unsigned)((row & 15) * (int)H.lda + (row >> 4) * H.Hp + swz_kc(row, lane & 7) * 8);
    }
    stage_offsets<128, false, 4, BK>(H.ldmv, wave, lane, goMV);
    stage_offsets<64, false, 4, BK>(H.ldlg, wave, lane, goLG);
    // (MEASURED, round 5: requesting only the 16 columns of W_lg that K <= 16 classes multiply -- a [64][16] K tile by global_load_lds_dword, a fifth of the
    //  ring's bytes less -- is bit-identical and changes nothing: 26.7 us for the launch, 0.2820 vs 0.2807 ms per step.  With the 5-slot ring (nothing
    //  either) that rules out both the bytes and the depth of the ring as what this K loop waits for.  Not kept.)
    unsigned short foAz[2], foAc[2], foMV[2][2 * DP][2], foLG[2][2], unused;
    constexpr int MVHALF = DP == 1 ? 0 : 1;
    const int mv_half = MVHALF ? (wave >> 1) : 0;                              // Dp = 128: waves 0, 1 hold mean's columns, 2, 3 log_var's
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        frag_offsets<32, true>(0, ks, lane, foAz[ks], unused);
        frag_offsets<32, true>(16, ks, lane, foAc[ks], unused);
#pragma unroll
        for (int j = 0; j < 2 * DP; ++j)
            frag_offsets<128, false>((wave * 2 * DP + j) * 16 - mv_half * 128, ks, lane, foMV[ks][j][0], foMV[ks][j][1]);
        frag_offsets<64, false>((wave & 3) * 16, ks, lane, foLG[ks][0], foLG[ks][1]);
    }
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) bf16_t*)ring) + 1024u * (unsigned)wave);

    auto issue = [&](int t, int slot) {
        const unsigned s = lds_w + 2u * (unsigned)(slot * G::STAGE);
        glds_tile(Ag + (int64_t)t * BK, goA, s, 4096u);
        glds_tile(H.Wmv + (k0 + t) * stepMV, goMV, s + 2u * G::A_ELEMS, 4096u);
        if constexpr (DP == 2) glds_tile(H.Wmv + 128 + (k0 + t) * stepMV, goMV, s + 2u * (G::A_ELEMS + G::MVH_ELEMS), 4096u);
        glds_tile(H.Wlg + (k0 + t) * stepLG, goLG, s + 2u * (G::A_ELEMS + DP * G::MVH_ELEMS), 4096u);
    };
    f32x4 acc[2 * DP], accl = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2 * DP; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    struct Frag { bf16x8 az, ac, mv[2 * DP], lg; };
    auto rd = [&](int slot, int ks, Frag& f) {
        const bf16_t* As = ring + slot * G::STAGE;
        const bf16_t* Ms = As + G::A_ELEMS + mv_half * G::MVH_ELEMS;
        const bf16_t* Ls = As + G::A_ELEMS + DP * G::MVH_ELEMS;
        f.az = read_frag<true>(As, foAz[ks], 0);
#pragma unroll
        for (int j = 0; j < 2 * DP; ++j) f.mv[j] = read_frag<false>(Ms, foMV[ks][j][0], foMV[ks][j][1]);
        if (has_lg) {
            f.ac = read_frag<true>(As, foAc[ks], 0);
            f.lg = read_frag<false>(Ls, foLG[ks][0], foLG[ks][1]);
        }
    };
    auto mma = [&](const Frag& f) {
        // operands swapped as everywhere (D[row = n][col = m]): a lane owns 4 consecutive n of one row m
#pragma unroll
        for (int j = 0; j < 2 * DP; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.mv[j], f.az, acc[j], 0, 0, 0);
        if (has_lg) accl = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.lg, f.ac, accl, 0, 0, 0);
    };

    // bias quads of this lane's columns (f32, arena tail): requested now, used after the K loop
    float bq[2 * DP][4], bl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 2 * DP; ++j) loadf4(H.bmv, (wave * 2 * DP + j) * 16 + g * 4, bq[j]);
    if (has_lg) loadf4(H.blg, wave * 16 + g * 4, bl);

    const LatentTile lt{(const lds_f*)tile, G::TLD, 64 * DP, G::NMV};
    latent_body<MODE, DSL, true>(
        H.L, (lds_f*)lat, (lds_f*)lat_rows, (int)blockIdx.x, lt,
        [&] {          // top: the first ring tiles, before anything else of the block touches memory
#pragma unroll
            for (int t = 0; t < G::NSTAGE; ++t)
                if (t < nk) issue(t, t);
        },
        [&] {          // mid: the K loop (gemm_bf16_body's pipeline), then the parked f32 tile
            Frag f0, f1;
            if (nk >= G::NSTAGE) wait_vmcnt<G::LOADS*(G::NSTAGE - 1)>();
            else wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            rd(0, 0, f0);
            for (int kt = 0; kt < nk; kt += G::NSTAGE) {
#pragma unroll
                for (int s = 0; s < G::NSTAGE; ++s) {
                    if (kt + s < nk) {
                        rd(s, 1, f1);
                        mma(f0);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        if (kt + s + G::NSTAGE <= nk) wait_vmcnt<G::LOADS*(G::NSTAGE - 2)>();
                        else wait_vmcnt<0>();
                        __builtin_amdgcn_s_barrier();
                        if (kt + s + G::NSTAGE < nk) issue(kt + s + G::NSTAGE, s);
                        if (kt + s + 1 < nk) rd((s + 1) % G::NSTAGE, 0, f0);
                        mma(f1);
                    }
                }
            }
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();          // every wave is done with the ring: it becomes the f32 tile
            { const LatentLaunch& L = H.L; (void)L; MEAS_LAT_STAMP(13); }      // (measurement build 7: the K loop has ended)
            const bool sliced = nsl > 1;                   // (then the biases are added behind the sum of the slices)
#pragma unroll
            for (int j = 0; j < 2 * DP; ++j) {
                f32x4 v = acc[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += sliced ? 0.f : bq[j][e];
                *reinterpret_cast<f32x4*>(tile + li * G::TLD + (wave * 2 * DP + j) * 16 + g * 4) = v;
            }
            {
                f32x4 v = accl;                    // (a wave without a logits tile: zeros -- the pad columns of the logits buffer)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += sliced ? 0.f : bl[e];
                *reinterpret_cast<f32x4*>(tile + li * G::TLD + G::NMV + wave * 16 + g * 4) = v;
            }
            __syncthreads();
            if (sliced) {
                // the partial tile to its slab, a ticket, and the last slice to arrive adds them up (agent-scope 16-byte accesses, no cache-wide fence:
                // see gemm_bf16_body); the biases after the sum
                constexpr int TQ = 16 * G::TLD / 4;
                f32x4* slab0 = reinterpret_cast<f32x4*>(H.ks_ws) + (int64_t)blockIdx.x * nsl * TQ;
                for (int q = tid; q < TQ; q += 256) {
                    const f32x4 v = reinterpret_cast<const f32x4*>(tile)[q];
                    f32x4* dst = slab0 + (int64_t)blockIdx.y * TQ + q;
                    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                int* flag = reinterpret_cast<int*>(lat_rows);          // (the row arrays are not in use yet)
                if (tid == 0) *flag = __hip_atomic_fetch_add(H.ks_tick + blockIdx.x, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                const bool last = *flag == nsl - 1;
                __syncthreads();                                       // (everyone has read the flag before the row arrays are written)
                if (!last) return false;
                for (int q = tid; q < TQ; q += 256) {
                    const f32x4 *p0, *p1, *p2, *p3, *p4, *p5, *p6, *p7;
                    auto at = [&](int sl) { return slab0 + (int64_t)(sl < nsl ? sl : nsl - 1) * TQ + q; };
                    p0 = at(0); p1 = at(1); p2 = at(2); p3 = at(3); p4 = at(4); p5 = at(5); p6 = at(6); p7 = at(7);
                    f32x4 u0, u1, u2, u3, u4, u5, u6, u7;
                    asm volatile(
                        "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\tglobal_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\t"
                        "global_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\tglobal_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\t"
                        "s_waitcnt vmcnt(0)"
                        : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5), "=&v"(u6), "=&v"(u7)
                        : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7)
                        : "memory");
                    f32x4 v = u0;
                    if (1 < nsl) v += u1;
                    if (2 < nsl) v += u2;
                    if (3 < nsl) v += u3;
                    if (4 < nsl) v += u4;
                    if (5 < nsl) v += u5;
                    if (6 < nsl) v += u6;
                    if (7 < nsl) v += u7;
                    const int c = (q * 4) % G::TLD;                    // column of the quad (TLD is a multiple of 4); the 4 pad columns take no bias
                    f32x4 b = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (c < G::NMV) b = *reinterpret_cast<const f32x4*>(H.bmv + c);
                    else if (c < G::NMV + 64) b = *reinterpret_cast<const f32x4*>(H.blg + (c - G::NMV));
                    reinterpret_cast<f32x4*>(tile)[q] = v + b;
                }
                if (tid == 0) __hip_atomic_store(H.ks_tick + blockIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
            }
            // mean | log_var | logits to global memory, row-contiguous quads (what DMVAE_EPI_BIAS_F32 stored)
            constexpr int QPR = (G::NMV + 64) / 4;
            const dmvae_latent_args& a = H.L.a;
            for (int q = tid; q < 16 * QPR; q += 256) {
                const int r = q / QPR, c = (q - r * QPR) * 4;
                const f32x4 v = *reinterpret_cast<const f32x4*>(tile + r * G::TLD + c);
                float* dst = c < G::NMV ? const_cast<float*>(a.mean) + (int64_t)(row0 + r) * a.ld_mean + c
                                        : const_cast<float*>(a.logits) + (int64_t)(row0 + r) * a.ld_logits + (c - G::NMV);
                *reinterpret_cast<f32x4*>(dst) = v;
            }
            return true;
        });
}

// ---------------------------------------------------------------- host side
static int g_heads_latent = 1;       // tuning knob (dmvae_debug_set_knob 19): 1 = the fused launch where it applies (default), 0 = never
void heads_latent_set(int v) { g_heads_latent = v; }

// the chunk width latent_fwd_kernel would use for (D, K) with 16-row blocks (latent.hip latent_geometry): the fused kernel must use the same one
// (it keys the device noise stream)
static int hl_dc(int D, int K) {
    int DC = 256;
    while (DC > 16 && DC / 2 >= D) DC /= 2;
    while (DC > 16 && latent_lds_bytes(K, 16, DC) > 60 * 1024) DC /= 2;
    return DC;
}

bool heads_latent_ok(int B_pad, int D, int K, int Dp, int Kp, int Hp, int mode, int rows_per_latent_block) {
    if (!g_heads_latent || mode < 0 || mode > 1) return false;
    // one round of the chip (knob 19 = 2 lifts that: a row-strip measurement at large batches); the partial sums are per 16-row block
    if (B_pad % 16 || (B_pad / 16 > 256 && g_heads_latent < 2) || rows_per_latent_block != 16) return false;
    if ((Dp != 64 && Dp != 128) || Kp != 64 || K > 64 || D > Dp || Hp % BK || Hp < BK) return false;
    if (latent_mfma_applies(D, K, mode)) return false;
    const int DC = hl_dc(D, K);
    if (DC < D || DC > 128) return false;                                                  // one chunk
    const size_t lds = Dp == 64 ? hl_lds_bytes<1>(K, DC) : hl_lds_bytes<2>(K, DC);
    return lds <= 160 * 1024;
}

template <int MODE, int DSL, int DP>
static void hl_launch_t(hipStream_t s, const HeadsLatentLaunch& H, int nblk, int nsl, size_t lds) {
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute((const void*)heads_latent_kernel<MODE, DSL, DP>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
    DMVAE_LAUNCH((heads_latent_kernel<MODE, DSL, DP>), dim3(nblk, nsl), dim3(256), lds, s, H);
}
// floats of slab space K slices need (dmvae_heads_args::kslice_ws): blocks x slices x the parked tile
int64_t heads_latent_kslice_floats(int B_pad, int Dp, int slices) { return (int64_t)(B_pad / 16) * slices * 16 * (Dp == 64 ? HLGeom<1>::TLD : HLGeom<2>::TLD); }

int heads_latent_launch(hipStream_t s, const dmvae_latent_args* a, const dmvae_heads_args* h) {
    const int Dp = h->Dp, Kp = h->Kp;
    if (!heads_latent_ok(a->B_pad, a->D, a->K, Dp, Kp, h->Hp, a->mode, 16)) {
        set_error("dmvae_heads_latent_fwd: not applicable (bf16 heads with Dp = 64 | 128, Kp = 64, K * D < 4096, B_pad <= 4096, Hp %% 64 == 0; B_pad=%d D=%d K=%d Dp=%d Kp=%d Hp=%d mode=%d)",
                  a->B_pad, a->D, a->K, Dp, Kp, h->Hp, a->mode);
        return DMVAE_EUNSUPPORTED;
    }
    if (a->act_dtype != DMVAE_BF16 || a->log_var != a->mean + Dp || a->ld_mean != 2 * Dp || a->ld_log_var != 2 * Dp || a->ld_logits != Kp || a->B > a->B_pad) {
        set_error("dmvae_heads_latent_fwd: outputs must be one [B_pad][2 Dp] buffer (mean | log_var) and one [B_pad][Kp] buffer (logits), bf16 activations");
        return DMVAE_EINVAL;
    }
    if (h->ld_mv < 2 * Dp || h->ld_lg < Kp || h->lda < 2 * h->Hp || (h->lda | h->ld_mv | h->ld_lg) % 8) {
        set_error("dmvae_heads_latent_fwd: leading dimensions (lda >= 2 Hp, ld_mv >= 2 Dp, ld_lg >= Kp, multiples of 8)");
        return DMVAE_EINVAL;
    }
    HeadsLatentLaunch H;
    H.L.a = *a; H.L.RB = 16; H.L.DC = hl_dc(a->D, a->K); H.L.nchunks = 1;
    H.hz = reinterpret_cast<const bf16_t*>(h->hz); H.lda = h->lda; H.Hp = h->Hp;
    H.Wmv = reinterpret_cast<const bf16_t*>(h->W_mv); H.ldmv = h->ld_mv;
    H.Wlg = reinterpret_cast<const bf16_t*>(h->W_lg); H.ldlg = h->ld_lg;
    H.bmv = h->b_mv; H.blg = h->b_lg;
    H.nt_lg = (a->K + 15) / 16;
    const int nblk = a->B_pad / 16;
    const int nsl = h->kslices > 1 ? h->kslices : 1;
    H.ks_ws = reinterpret_cast<float*>(h->kslice_ws); H.ks_tick = h->kslice_tick;
    if (nsl > 1 && (nsl > 8 || h->Hp % (nsl * BK) || !h->kslice_ws || !h->kslice_tick || h->kslice_ws_floats < heads_latent_kslice_floats(a->B_pad, Dp, nsl))) {
        set_error("dmvae_heads_latent_fwd: %d K slices need Hp %% (slices * 64) == 0, at most 8, a slab space of %lld floats and B_pad / 16 zeroed tickets", nsl,
                  (long long)heads_latent_kslice_floats(a->B_pad, Dp, nsl));
        return DMVAE_EINVAL;
    }
    const size_t lds = Dp == 64 ? hl_lds_bytes<1>(a->K, H.L.DC) : hl_lds_bytes<2>(a->K, H.L.DC);
    const double flops = 2.0 * a->B_pad * (double)h->Hp * (2 * Dp + Kp) + 6.0 * a->B * (double)a->K * a->D;
    const double bytes = 2.0 * (2.0 * a->B_pad * h->Hp + (double)h->Hp * (2 * Dp + Kp)) + 4.0 * ((double)a->B * (8.0 * a->D + 4.0 * a->K) + 2.0 * a->K * a->D * (nblk + 1));
    ProfScope ps(s, a->mode == 0 ? "heads_latent_exact" : "heads_latent_relaxed", flops, bytes);
#define HL(MODE_) \
    switch (H.L.DC) { \
        case 16: hl_launch_t<MODE_, 1, 1>(s, H, nblk, nsl, lds); break; \
        case 32: hl_launch_t<MODE_, 2, 1>(s, H, nblk, nsl, lds); break; \
        case 64: hl_launch_t<MODE_, 4, 1>(s, H, nblk, nsl, lds); break; \
        default: hl_launch_t<MODE_, 8, 2>(s, H, nblk, nsl, lds); break; \
    }
    if ((H.L.DC == 128) != (Dp == 128)) { set_error("dmvae_heads_latent_fwd: chunk width %d does not go with Dp = %d", H.L.DC, Dp); return DMVAE_EINVAL; }
    if (a->mode == 0) { HL(0) } else { HL(1) }
#undef HL
    return check_launch("heads_latent");
}

}  // namespace dmvae
