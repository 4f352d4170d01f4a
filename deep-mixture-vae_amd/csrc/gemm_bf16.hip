// bf16 MFMA GEMM for gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate) with
// fused epilogues.  Replaces the tf.matmul / tf.layers.dense forward ops of
// code/base_models.py:221-248,279-293 and their tf.gradients (dX, dW).
//
// One kernel template, three operand layouts (include/dmvae_hip.h):
//   FWD: A [M][K] k-contiguous,  B = W [K][N] n-contiguous
//   DX : A [M][K] k-contiguous,  B = W [N][K] k-contiguous
//   DW : A = X [K][M] m-contiguous, B = dY [K][N] n-contiguous (K = batch)
// so no transposed copy of weights, activations or gradients exists in HBM:
// a k-contiguous operand is read into its MFMA fragment with one
// ds_read_b128, an m/n-contiguous operand with two ds_read_b64_tr_b16 (the
// gfx950 transposing LDS read).
//
// Data movement: tiles go global -> LDS directly (global_load_lds_dwordx4,
// 16 B per lane, no VGPR staging) into an NSTAGE-deep LDS ring; NSTAGE-1 K
// tiles are in flight while one is multiplied (its fragments double-buffered in
// registers, so a slot is refilled as soon as its second half has been read),
// retired by a COUNTED s_waitcnt vmcnt(N) and one raw s_barrier per K step.
// What bounds these kernels is the CU's intake (L2 -> LDS, ~70 GB/s per CU,
// DESIGN.md section 6), so: as few bytes per flop as the grid size allows, TWO
// resident workgroups per CU whose load / compute / store phases overlap (rings
// <= 72 KiB, register budget for 4 waves per SIMD), and the epilogue's global
// reads issued before the K loop.  The LDS-DMA is issued
// from inline asm: through __builtin_amdgcn_global_load_lds hipcc (ROCm 7.2)
// treats it as a may-alias LDS store and drains the ring with vmcnt(0) before
// the first ds_read of every K step (measured: ring depth then buys nothing).
// The ring loop is unrolled NSTAGE times so every ring slot is a compile-time
// constant: LDS read addresses are (loop-invariant VGPR + immediate), global
// addresses are (wave-uniform SGPR tile pointer + loop-invariant 32-bit VGPR
// offset) -- the K loop carries no per-lane address arithmetic.
//
// An LDS-DMA write is lane-linear (wave-uniform base + lane*16 B), so tiles
// cannot be padded; bank conflicts are removed (SQ_LDS_BANK_CONFLICT = 0,
// profiles/) by an XOR swizzle of the 16-byte chunk index, applied to the
// per-lane SOURCE address when filling and again when reading:
//   k-contiguous tile, 128-B rows : chunk ^= (row >> 1) & 7          (ds_read_b128)
//   n-contiguous tile, 256-B rows : chunk ^= ((k & 3) | ((k >> 3) & 1) << 2) << 1
//   n-contiguous tile, 128-B rows : chunk ^= (((k >> 1) & 1) | ((k >> 3) & 1) << 1) << 1
// (the last two keep each transposing read's 8 rows x 32 B of a 32-lane half
// on distinct bank groups).
//
// Block = 4 waves (2 x 2) or 8 waves (4 x 2), BK = 64.  Epilogue: the fp32 tile is parked in the
// idle ring and re-read row-contiguous (see the body), then the fused op of include/dmvae_hip.h.
#include <algorithm>
#include <cmath>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "gemm_tile.h"
#include "measure.h"      // MEAS_*: stamps and ablation switches of the measurement builds; all of it empty / false in the product build

MEAS_TABLES_BF16

namespace dmvae {

// One output tile of one GEMM problem.  bid_raw = tile id within the problem, nwg = number of
// workgroups of the launch when the launch is this single problem (XCD-aware remap), else 0.
// NW = waves per workgroup: 4 (2 x 2) or 8 (4 x 2: two waves per SIMD share one tile's LDS traffic).
// BKT = K depth of one ring slot (64; 32 is available to the dW layout: a 128x128 tile then gets a
// 4-slot ring in the same 64 KiB, i.e. 48 KiB instead of 32 KiB in flight).
// CONV: conv mode (GemmArgs::conv_c, gemm_epilogue.h) -- a compile-time variant, so the dense kernels carry none of it.
// BONLY (DW layout, STORE_F32): a BIAS-ONLY strip -- the problem's A operand is never loaded or multiplied; only the column sums of
// B = dY (the ones-operand MFMA) are produced, into epi.out2.  The weight gradient of such a layer is computed elsewhere (the
// 256x256 macro tile, gemm_bf16_256.hip, which has no ones-operand pass); this strip rides in the grouped launch beside it.
template <int BM, int BN, int LAYOUT, int EPI, int NSTAGE, int NW, int BKT = BK, bool CONV = false, bool BONLY = false>
__device__ __forceinline__ void gemm_bf16_body(const GemmArgs& a, const int bid_raw, const int gstart, const int nwg, bf16_t* smem,
                                               const dmvae_adam_ctx* ac = nullptr, const int kslice = -1) {
    constexpr bool A_KC = (LAYOUT != DMVAE_GEMM_DW);
    constexpr bool B_KC = (LAYOUT == DMVAE_GEMM_DX);
    constexpr int A_ELEMS = BM * BKT, B_ELEMS = BN * BKT, STAGE = A_ELEMS + B_ELEMS;
    constexpr int WM = NW / 2;                       // waves along M (2 along N)
    constexpr int TM = BM / (16 * WM), TN = BN / 32; // 16x16 tiles per wave (wave tile = BM/WM x BN/2)
    static_assert(!BONLY || (LAYOUT == DMVAE_GEMM_DW && EPI == DMVAE_EPI_STORE_F32 && !CONV), "bias-only strips: dense DW / STORE_F32");
    constexpr int LOADS = ((BONLY ? 0 : BM) + BN) * BKT / (512 * NW);   // LDS-DMA instructions per lane per K tile
    static_assert(NSTAGE >= 2 && NSTAGE <= 8 && LOADS * (NSTAGE - 1) <= 63, "ring depth / vmcnt range");
    // smem: NSTAGE * STAGE elements, the kernel's ONLY LDS object (owned by the __global__ wrapper)

    MEAS_ANAT(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = a.N / BN;
    // XCD-aware tile order (speed only, never correctness): workgroups are dealt round-robin
    // over the 8 XCDs, so ids b and b+8 share an L2.  Give each XCD a CONTIGUOUS run of tile
    // ids (bijective for any grid size) ...
    // The problem owns the global workgroup ids [gstart, gstart + nwg) (gstart = 0 for a plain
    // launch, the problem's first id inside a grouped grid).
    const int bid = nwg < 0 ? bid_raw : xcd_run_index(gstart + bid_raw, gstart, gstart + nwg);   // nwg < 0: the caller did the remap
    // ... and walk the tiles of a run in supertiles of group_m tile-rows (column-major inside a
    // supertile): the ~32 tiles an XCD works on at one time then form a group_m x (32/group_m)
    // block, i.e. few A and few B panels, each reused from L2 by several workgroups.
    int tm, tn;
    {
        const int tiles_m = a.M / BM, gm_max = a.group_m;
        const int gsz = gm_max * tiles_n, grp = bid / gsz, first = grp * gm_max;
        const int gm = min(tiles_m - first, gm_max), in = bid - grp * gsz;
        tm = first + in % gm;
        tn = in / gm;
    }
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = (kslice >= 0 ? kslice : (int)blockIdx.y) * a.k_split;
    const int nk = a.k_split / BKT;

    const bf16_t* Ag = reinterpret_cast<const bf16_t*>(a.A);
    const bf16_t* Bg = reinterpret_cast<const bf16_t*>(a.B);
    if (!A_KC && CONV)       // conv-mode weight gradient: the tap shift of every column sits in the per-lane offsets (below),
        Ag += ((int64_t)kbeg - (a.conv_p + 1)) * a.lda;     // taken relative to the most negative shift so that they stay unsigned
    else
        Ag += A_KC ? ((int64_t)m0 * a.lda + kbeg) : ((int64_t)kbeg * a.lda + m0);
    Bg += B_KC ? ((int64_t)n0 * a.ldb + kbeg) : ((int64_t)kbeg * a.ldb + n0);
    const int64_t stepA = A_KC ? (int64_t)BKT : (int64_t)BKT * a.lda;
    const int64_t stepB = B_KC ? (int64_t)BKT : (int64_t)BKT * a.ldb;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // DW layout: the bias gradient db[n] = sum_k dY[k][n] rides along as one extra MFMA per B
    // fragment against an all-ones A operand (exact: products with 1.0, fp32 accumulate), in
    // the workgroups of the first tile row only -- no separate column-sum pass over dY.
    constexpr bool DW = (LAYOUT == DMVAE_GEMM_DW);
    const bool do_bias = DW && !MEAS_NO_BIAS_MFMA && a.epi.out2 != nullptr && (tm == 0 || BONLY);
    f32x4 bacc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) bacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const s16x8 ones_bits = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
    const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_bits);

    // loop-invariant per-lane addressing: global source byte offsets and LDS fragment offsets
    unsigned goA[BM * BKT / (512 * NW)], goB[BN * BKT / (512 * NW)];
    stage_offsets<BM, A_KC, NW, BKT>(a.lda, wave, lane, goA);
    stage_offsets<BN, B_KC, NW, BKT>(a.ldb, wave, lane, goB);
    // Conv mode: K (or, for the weight gradient, M) runs over (tap, channel) with conv_c channels per tap, and an
    // operand row holds lda >= conv_c channels.  conv_c >= 64: a 64-wide tile lies inside one tap.  conv_c = 32 (the
    // 32-channel layers, stored with 32 zero pad channels): a tile covers TWO taps -- 16-byte chunks 0..3 belong to
    // the first, 4..7 to the second (selA), each reading channels 0..31 of its own shifted row.
    unsigned selA[BM * BKT / (512 * NW)];
#pragma unroll
    for (int i = 0; i < BM * BKT / (512 * NW); ++i) selA[i] = 0;
    if constexpr (CONV) {
        static_assert(A_KC || BM == 64, "conv-mode weight gradient: 64-row tiles");
#pragma unroll
        for (int i = 0; i < BM * BKT / (512 * NW); ++i) {
            const int row = i * (8 * NW) + wave * 8 + (lane >> 3);            // both layouts: 8 rows x 8 chunks per wave pass
            if constexpr (A_KC) {
                const int k = swz_kc(row, lane & 7) * 8;                          // position inside the 64-wide K tile
                selA[i] = (unsigned)(k / a.conv_c);
                goA[i] = 2u * (unsigned)(row * (int)a.lda + k % a.conv_c);
            } else {
                const int m = m0 + swz_nc<64>(row, lane & 7) * 8;                // (tap, channel) of this chunk's columns
                const int tap = min(m / a.conv_c, 8);                             // rows past 9 taps are padding (their gradient is zeroed)
                goA[i] = 2u * (unsigned)((row + (tap / 3 - 1) * a.conv_p + (tap % 3 - 1) + a.conv_p + 1) * (int)a.lda + m % a.conv_c);
            }
        }
    }
    unsigned short foA[BKT / 32][TM][2], foB[BKT / 32][TN][2];
#pragma unroll
    for (int ks = 0; ks < BKT / 32; ++ks) {
#pragma unroll
        for (int i = 0; i < TM; ++i) frag_offsets<BM, A_KC>(wm * (BM / WM) + i * 16, ks, lane, foA[ks][i][0], foA[ks][i][1]);
#pragma unroll
        for (int j = 0; j < TN; ++j) frag_offsets<BN, B_KC>(wn * (BN / 2) + j * 16, ks, lane, foB[ks][j][0], foB[ks][j][1]);
    }
    // LDS byte address of this wave's first DMA chunk in ring slot 0 (wave-uniform)
    const unsigned lds_w = __builtin_amdgcn_readfirstlane(
        (unsigned)(size_t)((__attribute__((address_space(3))) bf16_t*)smem) + 1024u * (unsigned)wave);

    // issue the loads of K tile t (< nk) into ring slot `slot`
    // conv mode: tiles are issued in K order, so (tap, channel) of a tile's first column is a running counter
    int cv_tap = 0, cv_c0 = 0;
    int64_t cv_off = 0;
    unsigned cv_delta = 0;                              // bytes from the tile's first tap to its second (conv_c = 32)
    auto issue = [&](int t, int slot) {                 // t < nk
        const int tc = t;
        const unsigned s = lds_w + 2u * (unsigned)(slot * STAGE);
        if constexpr (A_KC && CONV) {
            if (t < nk) {
                const int t0 = min(cv_tap, 8), t1 = min(cv_tap + 1, 8);       // taps past the ninth: padding columns, zero weights
                cv_off = conv_tap_offset(t0, cv_c0, a.conv_p, a.lda);
                cv_delta = (unsigned)(2 * (int)(conv_tap_offset(t1, 0, a.conv_p, a.lda) - conv_tap_offset(t0, 0, a.conv_p, a.lda)));
                cv_c0 += BKT;
                while (cv_c0 >= a.conv_c) { cv_c0 -= a.conv_c; ++cv_tap; }
            }
            unsigned o[BM * BKT / (512 * NW)];
#pragma unroll
            for (int i = 0; i < BM * BKT / (512 * NW); ++i) o[i] = goA[i] + selA[i] * cv_delta;
            glds_tile(Ag + cv_off, o, s, 1024u * NW);
        } else if constexpr (!BONLY) {
            glds_tile(Ag + tc * stepA, goA, s, 1024u * NW);
        }
        glds_tile(Bg + tc * stepB, goB, s + 2u * A_ELEMS, 1024u * NW);
    };
    // fragments of K sub-step ks (32 deep) of ring slot `slot`
    auto rd = [&](int slot, int ks, bf16x8 (&af)[TM], bf16x8 (&bfr)[TN]) {
        const bf16_t* As = smem + slot * STAGE;
        const bf16_t* Bs = As + A_ELEMS;
        if constexpr (MEAS_NO_LDS_READ) {
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = ones;
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = ones;
        } else {
            if constexpr (!BONLY) {
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i] = read_frag<A_KC>(As, foA[ks][i][0], foA[ks][i][1]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = read_frag<B_KC>(Bs, foB[ks][j][0], foB[ks][j][1]);
        }
    };
    auto mma = [&](const bf16x8 (&af)[TM], const bf16x8 (&bfr)[TN]) {
        if constexpr (MEAS_NO_MFMA) {      // (fragments still read: kept live)
#pragma unroll
            for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(bfr[j]));
        } else if constexpr (!BONLY) {
            // the wave that has its fragments goes first: two waves share a SIMD, the other one is waiting on
            // LDS or the barrier anyway (measured on the step, tools/ab_libs.sh: 0.3136 -> 0.3110 ms)
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    // operands swapped: D[row = n][col = m] -> each lane owns 4 consecutive n of one m
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        if constexpr (DW) {
            if (do_bias) {
#pragma unroll
                for (int j = 0; j < TN; ++j) bacc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], ones, bacc[j], 0, 0, 0);
            }
        }
    };

    // Software pipeline over the two 32-deep sub-steps of a K tile, fragments double-buffered in
    // registers: the LDS reads of the NEXT sub-step are issued before the MFMAs of the current
    // one, so a wave alone on its SIMD (no second resident workgroup to hide behind) keeps the
    // matrix core fed.  Per K tile kt (slot s):
    //     read f1 <- (s, ks 1) | mma f0 | lgkmcnt(0): slot s is now fully in registers
    //     vmcnt: tile kt+1 landed | barrier: ... for every wave, and every wave is done with slot s
    //     refill slot s with tile kt+NSTAGE | read f0 <- (s+1, ks 0) | mma f1
    // NSTAGE tiles are in flight after the prologue, NSTAGE-1 while a tile is multiplied.
    static_assert(BKT == 64, "the pipelined loop handles two 32-deep sub-steps per ring slot");
    bf16x8 f0a[TM], f0b[TN], f1a[TM], f1b[TN];
    // RELU_MASK (every dX GEMM): the forward activations that gate this tile are fetched NOW, in the
    // epilogue's row-contiguous quad order, so their latency hides behind the K loop instead of
    // standing between the last MFMA and the stores.  Issued before the first tile load: they are
    // the oldest vector-memory operations, so the counted vmcnt waits below stay exact.
    constexpr int NQ = BM * (BN / 4) / (64 * NW);
    float4 target[EPI == DMVAE_EPI_BIAS_RECON ? NQ : 1];       // BIAS_RECON: likewise the f32 reconstruction targets
    unsigned target_zero = 0;                                   // bit q: quad q's target is zeros (a row / column outside the batch), whatever was loaded
    if constexpr (EPI == DMVAE_EPI_BIAS_RECON) {
        // recon_kind bit 8 (the plan's step path, dmvae_plan_load_batch_step): no f32 copy of the batch exists -- row m's targets are
        // row perm[first + m] of the dataset itself: aux0 = data [ld2 rows][ld0 = input_dim floats], aux1 = perm (int32 or null),
        // ld1 = first, or batch_cursor * d_off when aux2 = the device state; rows >= m_valid and pad columns are zeros (masked anyway)
        const bool tg = (a.epi.recon_kind & 0x100) != 0;
        int64_t tfirst = 0;
        if (tg) {
            const dmvae_state* tst = reinterpret_cast<const dmvae_state*>(a.epi.aux2);
            tfirst = tst ? (int64_t)tst->batch_cursor * a.epi.d_off : a.epi.ld1;
        }
        if (tg) {
            // EVERY load unconditional (indices clamped, the result selected afterwards) and each stage issued for all NQ quads before the next: a
            // load inside a divergent `if` is waited for at the join, and the first form -- [permutation entry -> vmcnt(0) -> target row] per quad,
            // inside ifs -- ran its 2 NQ round trips one after the other in front of the K loop (seen in the ISA; round 4)
            const int32_t* perm = reinterpret_cast<const int32_t*>(a.epi.aux1);
            int64_t src[NQ];
            bool ok[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int idx = q * (64 * NW) + tid;
                const int m = m0 + idx / (BN / 4), n = n0 + (idx % (BN / 4)) * 4;
                const int64_t sr = tfirst + m;
                ok[q] = m < a.epi.m_valid && n + 3 < (int)a.epi.ld0 && sr >= 0 && sr < a.epi.ld2;      // (the permutation has ld2 = n_rows entries)
                src[q] = ok[q] ? sr : 0;
            }
            if (perm) {
#pragma unroll
                for (int q = 0; q < NQ; ++q) src[q] = (int64_t)perm[src[q]];
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int idx = q * (64 * NW) + tid;
                const int n = n0 + (idx % (BN / 4)) * 4;
                ok[q] = ok[q] && src[q] >= 0 && src[q] < a.epi.ld2;
                target[q] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.epi.aux0) + (ok[q] ? src[q] * a.epi.ld0 + n : 0));
                if (!ok[q]) target_zero |= 1u << q;          // (selected in the epilogue: touching the value here would wait for it in front of the ring's first loads)
            }
        } else {
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int idx = q * (64 * NW) + tid;
                const int ml = idx / (BN / 4), c = idx % (BN / 4);
                target[q] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(a.epi.aux0) + (int64_t)(m0 + ml) * a.epi.ld0 + n0 + c * 4);
            }
        }
    }
    constexpr bool LAT_PRE = EPI == DMVAE_EPI_LATENT && NQ <= 4;       // (12 registers per quad: the small tiles only)
    float lat[LAT_PRE ? NQ : 1][12];            // LATENT (the dZ GEMM): the reparameterisation / KL gradient quads, likewise
    if constexpr (LAT_PRE) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = q * (64 * NW) + tid;
            const int ml = idx / (BN / 4), c = idx % (BN / 4);
            loadf4(a.epi.aux0, (int64_t)(m0 + ml) * a.epi.ld0 + n0 + c * 4, lat[q]);
            loadf4(a.epi.aux1, (int64_t)(m0 + ml) * a.epi.ld1 + n0 + c * 4, lat[q] + 4);
            loadf4(a.epi.aux2, (int64_t)(m0 + ml) * a.epi.ld2 + n0 + c * 4, lat[q] + 8);
        }
    }
    // the bias quad of this thread's column block: the same for every row it handles (64 NW is a multiple of BN / 4)
    constexpr bool HAS_BIAS = (EPI == DMVAE_EPI_BIAS_RELU || EPI == DMVAE_EPI_BIAS_F32 || EPI == DMVAE_EPI_BIAS_SIGMOID || EPI == DMVAE_EPI_BIAS_RECON);
    static_assert((64 * NW) % (BN / 4) == 0, "a thread keeps its column block across the epilogue's passes");
    float bq[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (HAS_BIAS) loadf4(a.epi.bias, n0 + (tid % (BN / 4)) * 4, bq);
    uint2 gate[EPI == DMVAE_EPI_RELU_MASK ? NQ : 1];
    if constexpr (EPI == DMVAE_EPI_RELU_MASK) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int idx = q * (64 * NW) + tid;
            const int ml = idx / (BN / 4), c = idx % (BN / 4);
            if constexpr (MEAS_NO_MASK_READ) gate[q] = make_uint2(0x3f803f80u, 0x3f803f80u);
            else gate[q] = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(a.epi.aux0) + (int64_t)(m0 + ml) * a.epi.ld0 + n0 + c * 4);
        }
    }
    // Only real tiles are issued.  (Round 1 clamped tile ids past the K range to the last tile to keep every vmcnt count a
    // constant: NSTAGE extra tile loads per workgroup that the epilogue then had to wait for -- 3 of 11 at K = 512.)  The waits
    // of the last NSTAGE - 1 iterations, where fewer tiles are in flight than the constant assumes, drain everything instead.
#pragma unroll
    for (int t = 0; t < NSTAGE; ++t)
        if (t < nk) issue(t, t);
    if (nk >= NSTAGE) wait_vmcnt<LOADS*(NSTAGE - 1)>();        // tile 0 has landed (this wave's share)
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    MEAS_ANAT(1);
    rd(0, 0, f0a, f0b);
    for (int kt = 0; kt < nk; kt += NSTAGE) {
#pragma unroll
        for (int s = 0; s < NSTAGE; ++s) {                      // ring slot s is a compile-time constant here
            if (kt + s < nk) {
                rd(s, 1, f1a, f1b);
                mma(f0a, f0b);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (kt + s + NSTAGE <= nk) wait_vmcnt<LOADS*(NSTAGE - 2)>();   // tile kt+s+1 has landed (this wave's share)
                else wait_vmcnt<0>();                                          // K tail: fewer tiles in flight than the constant assumes
                __builtin_amdgcn_s_barrier();
                if constexpr (!MEAS_NO_KLOOP_LOADS) {
                    if (kt + s + NSTAGE < nk) issue(kt + s + NSTAGE, s);
                }
                if (kt + s + 1 < nk) rd((s + 1) % NSTAGE, 0, f0a, f0b);
                mma(f1a, f1b);
            }
        }
    }
    wait_vmcnt<0>();                                            // (nothing is in flight any more)
    __builtin_amdgcn_s_barrier();
    MEAS_KLOOP_END();
    MEAS_ANAT(2);

    // Epilogue through LDS.  The MFMA result map gives a lane 4 consecutive n of ONE row and its 15
    // neighbours 15 OTHER rows: stored straight from the accumulators a wave instruction touches
    // 16 rows x 32 B (bf16) -- partial-line writes (and mask / target reads) that ran the
    // epilogue at ~1.4 TB/s.  The fp32 tile is instead parked in the (now idle) ring, 16-byte
    // chunk index XOR (row & 7) (conflict-free ds_write_b128 / ds_read_b128), and re-read
    // row-contiguous: a wave then covers whole 256..512-byte row segments.
    float loss = 0.f;
    const int li = lane & 15, g = lane >> 4;
    constexpr int CH = BN / 4, NT = 64 * NW;
    static_assert(BM * BN * 4 <= NSTAGE * STAGE * 2, "fp32 tile must fit in the ring");
    // deterministic split-K (conv weight gradient, cfg.deterministic): K slice y stores into its own slab, summed in a fixed
    // order by slab_reduce afterwards -- no float atomics
    // ... and likewise the K slices of a dense weight-gradient problem (GemmArgs::slab_stride != 0: the dW group of a large batch,
    // whose tile count is fixed by the parameter shapes while K grows with the batch; the slabs are added by adam_slabs / slab_reduce)
    dmvae_epilogue epi_s = a.epi;
    if constexpr (EPI == DMVAE_EPI_STORE_F32 && (CONV || LAYOUT == DMVAE_GEMM_DW)) {
        const int ys = kslice >= 0 ? kslice : (int)blockIdx.y;
        epi_s.out = reinterpret_cast<float*>(epi_s.out) + (int64_t)ys * a.slab_stride;
        if (epi_s.out2) epi_s.out2 = reinterpret_cast<float*>(epi_s.out2) + (int64_t)ys * a.slab_stride2;
    }
    float* ct = reinterpret_cast<float*>(smem);
    if constexpr (MEAS_NO_EPILOGUE) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(acc[i][j]));
    } else if constexpr (!BONLY) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int ml = wm * (BM / WM) + i * 16 + li;
            const int c = (wn * (BN / 2) + j * 16) / 4 + g;
            *reinterpret_cast<f32x4*>(ct + ml * BN + ((c ^ (ml & 7)) << 2)) = acc[i][j];
        }
    __syncthreads();
    // K slices with a fused epilogue (GemmArgs::tick): this workgroup's tile is a PARTIAL product.  It goes to its slab; the last slice to arrive
    // adds all slabs in ascending order (its own included, from memory: the order does not depend on who is last) and carries on into the epilogue.
    constexpr int KSPLIT_MAX_S = 8;
    constexpr bool KSLICE_OK = !CONV && LAYOUT != DMVAE_GEMM_DW && BM * BN <= 4096 && NW <= 4 && BM * BN * 4 + 16 <= NSTAGE * STAGE * 2 &&      // (the 64-row and thinner tiles: what a small batch runs on)
                               (EPI == DMVAE_EPI_BIAS_RELU || EPI == DMVAE_EPI_RELU_MASK || EPI == DMVAE_EPI_BIAS_F32 || EPI == DMVAE_EPI_LATENT);
    if constexpr (KSLICE_OK) {
        if (a.tick != nullptr && gridDim.y > 1) {
            const int S = (int)gridDim.y, tile_lin = tm * tiles_n + tn;
            constexpr int Q = BM * BN;                          // floats per tile
            static_assert(KSPLIT_MAX_S == 8, "the combine issues eight loads per quad");
            // The slabs cross XCDs (the slices of a tile land on different L2s).  A release / acquire FENCE at agent scope does it -- and was measured: it
            // writes back and invalidates the XCD's whole L2 per workgroup, every other workgroup there loses its weight panels (the step +10.6 % at 100
            // rows instead of faster).  So the slab traffic itself is agent-scope: write-through stores, L2-bypassing loads (sc1), one dword each; the
            // stores are complete (vmcnt(0)) before the barrier in front of the ticket, the ticket is an agent-scope atomic.  No cache-wide operation.
            // (16-byte accesses with the sc1 bit from inline asm: the dword forms of __hip_atomic_load ran one round trip after the other -- 128 of them per thread
            //  for eight slabs: the sliced launch took twice the unsliced one.)  Up to eight slabs: all eight loads of a quad are issued together, surplus
            //  ones re-read the last slab and are not added.
            f32x4* slab0 = reinterpret_cast<f32x4*>(a.ws) + (int64_t)tile_lin * S * (Q / 4);
            const f32x4* ct4 = reinterpret_cast<const f32x4*>(ct);
            for (int q = tid; q < Q / 4; q += 64 * NW) {        // the LDS image as it is: the slab is this tile's own
                const f32x4 v = ct4[q];
                f32x4* dst = slab0 + (int64_t)blockIdx.y * (Q / 4) + q;
                asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(v) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* flag = reinterpret_cast<int*>(ct + BM * BN);
            if (tid == 0) *flag = __hip_atomic_fetch_add(a.tick + tile_lin, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            if (*flag != S - 1) return;
            for (int q = tid; q < Q / 4; q += 64 * NW) {
                const f32x4 *p0, *p1, *p2, *p3, *p4, *p5, *p6, *p7;
                auto at = [&](int sl) { return slab0 + (int64_t)(sl < S ? sl : S - 1) * (Q / 4) + q; };
                p0 = at(0); p1 = at(1); p2 = at(2); p3 = at(3); p4 = at(4); p5 = at(5); p6 = at(6); p7 = at(7);
                f32x4 u0, u1, u2, u3, u4, u5, u6, u7;
                asm volatile(
                    "global_load_dwordx4 %0, %8, off sc1\n\tglobal_load_dwordx4 %1, %9, off sc1\n\tglobal_load_dwordx4 %2, %10, off sc1\n\tglobal_load_dwordx4 %3, %11, off sc1\n\t"
                    "global_load_dwordx4 %4, %12, off sc1\n\tglobal_load_dwordx4 %5, %13, off sc1\n\tglobal_load_dwordx4 %6, %14, off sc1\n\tglobal_load_dwordx4 %7, %15, off sc1\n\t"
                    "s_waitcnt vmcnt(0)"
                    : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(u4), "=&v"(u5), "=&v"(u6), "=&v"(u7)
                    : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(p4), "v"(p5), "v"(p6), "v"(p7)
                    : "memory");
                f32x4 v = u0;                                   // ascending slice order, whoever arrived last
                if (1 < S) v += u1;
                if (2 < S) v += u2;
                if (3 < S) v += u3;
                if (4 < S) v += u4;
                if (5 < S) v += u5;
                if (6 < S) v += u6;
                if (7 < S) v += u7;
                reinterpret_cast<f32x4*>(ct)[q] = v;
            }
            if (tid == 0) __hip_atomic_store(a.tick + tile_lin, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch (and the next replay of a captured one)
            __syncthreads();
        }
    }
    if constexpr (EPI == DMVAE_EPI_ADAM && !CONV) {
        // the update in batches of NB quads whose parameter / m / v loads are in flight together, SOFTWARE-PIPELINED over the batches (adam_pipelined,
        // gemm_tile.h: the loads of batch i + 1 are issued before the stores of batch i -- vmcnt counts both in issue order, so a load behind stores is
        // seen only once they are acknowledged; batch after batch, as until round 5, every batch paid a full round trip with nothing in flight)
        constexpr int NQE = BM * CH / NT;
        constexpr int NB = NQE % 4 == 0 ? 4 : (NQE % 2 == 0 ? 2 : 1);
        const unsigned base = (unsigned)((reinterpret_cast<const float*>(a.epi.out) - ac->grad) + (int64_t)m0 * a.epi.ldo + n0);
#ifndef DMVAE_ADAM_EPI_BATCHWISE
#ifdef DMVAE_ADAM_EPI_NB
        constexpr int PNB = NQE % DMVAE_ADAM_EPI_NB == 0 ? DMVAE_ADAM_EPI_NB : NB, PDEPTH = DMVAE_ADAM_EPI_DEPTH;
#else
        constexpr int PNB = NB, PDEPTH = 2;
#endif
        adam_pipelined<PNB, NQE / PNB, PDEPTH>(*ac,
            [&](int i, int b) {
                const int idx = (i * PNB + b) * NT + tid;
                return base + (unsigned)(idx / CH) * (unsigned)a.epi.ldo + (unsigned)(idx % CH) * 4u;
            },
            [&](int i, int b, float (&gv)[4]) {
                const int idx = (i * PNB + b) * NT + tid;
                const int ml = idx / CH, c = idx % CH;
                const f32x4 t = *reinterpret_cast<const f32x4*>(ct + ml * BN + ((c ^ (ml & 7)) << 2));
                gv[0] = t[0]; gv[1] = t[1]; gv[2] = t[2]; gv[3] = t[3];
            });
#else
#pragma unroll
        for (int q0 = 0; q0 < NQE; q0 += NB) {
            unsigned off[NB];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int idx = (q0 + b) * NT + tid;
                off[b] = base + (unsigned)(idx / CH) * (unsigned)a.epi.ldo + (unsigned)(idx % CH) * 4u;
            }
            adam_quads<NB>(*ac, off, [&](int b, float (&gv)[4]) {
                const int idx = (q0 + b) * NT + tid;
                const int ml = idx / CH, c = idx % CH;
                const f32x4 t = *reinterpret_cast<const f32x4*>(ct + ml * BN + ((c ^ (ml & 7)) << 2));
                gv[0] = t[0]; gv[1] = t[1]; gv[2] = t[2]; gv[3] = t[3];
            });
        }
#endif
    } else
#pragma unroll
    for (int q = 0; q < BM * CH / NT; ++q) {
        const int idx = q * NT + tid;
        const int ml = idx / CH, c = idx % CH;
        bool padcol = false;
        if constexpr (CONV) {     // a 32-channel activation is stored 32 wide: the tile's other columns are not part of the output
            padcol = n0 + c * 4 >= a.epi.n_valid;
            if (padcol && !(EPI == DMVAE_EPI_STORE_F32 && a.slab_stride)) continue;      // (slabs are summed whole: their pad columns must hold zeros)
        }
        const f32x4 t = *reinterpret_cast<const f32x4*>(ct + ml * BN + ((c ^ (ml & 7)) << 2));
        float v[4] = {t[0], t[1], t[2], t[3]};
        if (padcol) { v[0] = 0.f; v[1] = 0.f; v[2] = 0.f; v[3] = 0.f; }
        if constexpr (EPI == DMVAE_EPI_ADAM) {
            // the gradient quad updates the matching parameter / m / v elements (same offset in every arena)
            const int64_t off = (reinterpret_cast<const float*>(a.epi.out) - ac->grad) + (int64_t)(m0 + ml) * a.epi.ldo + n0 + c * 4;
            adam_quad(*ac, off, v);
        } else if constexpr (EPI == DMVAE_EPI_RELU_MASK) {
            const uint2 y = gate[q];                 // prefetched before the K loop
            v[0] = __uint_as_float(y.x << 16) > 0.f ? v[0] : 0.f;
            v[1] = __uint_as_float(y.x & 0xffff0000u) > 0.f ? v[1] : 0.f;
            v[2] = __uint_as_float(y.y << 16) > 0.f ? v[2] : 0.f;
            v[3] = __uint_as_float(y.y & 0xffff0000u) > 0.f ? v[3] : 0.f;
            ActIO<bf16_t>::store4(a.epi.out, (int64_t)(m0 + ml) * a.epi.ldo + n0 + c * 4, v);
        } else if constexpr (EPI == DMVAE_EPI_BIAS_RECON) {
            const bool tz = (target_zero >> q) & 1u;
            const float xq[4] = {tz ? 0.f : target[q].x, tz ? 0.f : target[q].y, tz ? 0.f : target[q].z, tz ? 0.f : target[q].w};
            epilogue_quad<EPI, bf16_t>(a.epi, m0 + ml, n0 + c * 4, v, loss, xq, nullptr, bq);
        } else if constexpr (LAT_PRE) {
            epilogue_quad<EPI, bf16_t>(a.epi, m0 + ml, n0 + c * 4, v, loss, lat[q]);
        } else {
            epilogue_quad<EPI, bf16_t>(epi_s, m0 + ml, n0 + c * 4, v, loss, nullptr, nullptr, HAS_BIAS ? bq : nullptr);
        }
    }
    }
    if constexpr (DW) {
        if (do_bias && wm == 0 && li == 0) {      // D[row = n][col = any m]: column 0 of the wm == 0 waves writes
            float* db = reinterpret_cast<float*>(epi_s.out2);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * (BN / 2) + j * 16 + g * 4;
                if constexpr (CONV) {
                    if (n >= a.epi.n_valid) {
                        if (EPI == DMVAE_EPI_STORE_F32 && a.slab_stride2) *reinterpret_cast<float4*>(db + n) = make_float4(0.f, 0.f, 0.f, 0.f);
                        continue;
                    }
                }
                if constexpr (EPI == DMVAE_EPI_ATOMIC_F32) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(db + n + e, bacc[j][e]);
                } else if constexpr (EPI == DMVAE_EPI_ADAM) {
                    float v[4] = {bacc[j][0], bacc[j][1], bacc[j][2], bacc[j][3]};
                    adam_quad(*ac, (db - ac->grad) + n, v);
                } else {
                    *reinterpret_cast<float4*>(db + n) = make_float4(bacc[j][0], bacc[j][1], bacc[j][2], bacc[j][3]);
                }
            }
        }
    }
    if constexpr (EPI == DMVAE_EPI_BIAS_RECON) {
        float* red = reinterpret_cast<float*>(smem);
        lds_barrier();                                           // every wave is done reading the parked tile (no wait for the global stores)
        const float t = block_sum_waves<NW>(loss, red);
        // partials live on the 64x64 cell grid of the output (dmvae_gemm_partials is tile independent): the tile's sum
        // goes to its first cell, zeros to its others
        constexpr int CM = BM / 64, CN = BN / 64;
        if (tid < CM * CN) a.epi.partials[(tm * CM + tid / CN) * (a.N / 64) + tn * CN + tid % CN] = tid == 0 ? t : 0.f;
    }
    MEAS_ANAT_DRAIN();
}

template <int BM, int BN, int LAYOUT, int EPI, int NSTAGE, int NW>
// second bound = waves per SIMD the register budget must allow: two resident workgroups per CU (the
// rings are sized for that); without it the 128x128 dX instantiation took 130 VGPRs = one workgroup per CU
// (the recon epilogue with its prefetched targets would spill under that budget: left unconstrained)
__global__ __launch_bounds__(64 * NW, (EPI == DMVAE_EPI_BIAS_RECON ? 1 : NW / 2)) void gemm_bf16_kernel(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[NSTAGE * (BM + BN) * BK];
    gemm_bf16_body<BM, BN, LAYOUT, EPI, NSTAGE, NW>(a, blockIdx.x, 0, gridDim.x, smem);
}

// A dX GEMM (DX layout; LATENT epilogue = the dZ GEMM, RELU_MASK = a layer's dX) carrying RIDERS in the first ids of its grid (GemmRiders,
// gemm_epilogue.h): the step_finalize blocks (common.h) and / or the gather of the NEXT batch (gather_rows_block).  Where they ride (api.hip):
//   * step_finalize in the dZ GEMM at 4096 rows (64 tiles on 256 CUs: every rider a CU of its own) -- in the heads' dX launch, which fills every
//     slot of the chip, the same riders cost that launch 2.2 us (MEASURED round 4, knob 16);
//   * with a prefetched batch (dmvae_plan_prefetch_batch): the gather in the dZ GEMM -- memory work on the ~190 CUs that launch leaves idle -- and
//     step_finalize one launch earlier, in the output layer's dX (one 8-wave tile per CU: the riders are second workgroups).  The gather as
//     second workgroups of the TRUNK's dX (one tile per CU, K = 4096) was measured first and gives nothing: 0.2700 vs 0.2694 ms -- a CU's
//     vector-memory path is what its K loop is bound by, and the riders share it.
// Same tile code, same bits.
template <int BM, int BN, int EPI, int NSTAGE, int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void gemm_bf16_dx_riders_kernel(GemmArgs a, GemmRiders r) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[NSTAGE * (BM + BN) * BK];
    const int lead = r.nfin + (r.gat_last ? 0 : r.ngat);
    if (blockIdx.y > 0 && ((int)blockIdx.x < lead || (r.gat_last && (int)blockIdx.x >= (int)gridDim.x - r.ngat))) return;      // (K slices of the host GEMM, GemmArgs::tick: the riders ride once)
    if (r.gat_last && (int)blockIdx.x >= (int)gridDim.x - r.ngat) {
        if constexpr (MEAS_EMPTY_GATHER_RIDERS) return;
        gather_rows_block<bf16_t, 16>((int)blockIdx.x - ((int)gridDim.x - r.ngat), 64 * NW, r.gat);
        return;
    }
    if ((int)blockIdx.x < r.nfin) {
        if (NW > 4 && threadIdx.x >= 256) return;          // (step_finalize_block is written for four waves; a finished wave leaves the barrier count)
        if ((int)blockIdx.x < r.fin.nblocks) step_finalize_block((int)blockIdx.x, r.fin, reinterpret_cast<float(*)[17]>(smem));
        return;
    }
    if ((int)blockIdx.x < lead) {
        if constexpr (MEAS_EMPTY_GATHER_RIDERS) return;
        gather_rows_block<bf16_t, 16>((int)blockIdx.x - r.nfin, 64 * NW, r.gat);      // (two rider blocks per idle CU, 16 quads in flight per thread: one pass at 4096 rows)
        return;
    }
    gemm_bf16_body<BM, BN, DMVAE_GEMM_DX, EPI, NSTAGE, NW>(a, (int)blockIdx.x - lead, lead, (int)gridDim.x - lead - (r.gat_last ? r.ngat : 0), smem);
}

// The same tile in conv mode (implicit 3x3 convolution, csrc/conv.hip): its own kernel, so the dense
// instantiations above are bit-for-bit what they were (a run-time flag in their K loop cost 1.6 % of the step).
template <int BM, int BN, int LAYOUT, int EPI, int NSTAGE, int NW>
__global__ __launch_bounds__(64 * NW, NW / 2) void gemm_bf16_conv_kernel(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[NSTAGE * (BM + BN) * BK];
    if constexpr (LAYOUT == DMVAE_GEMM_DW) {
        // weight gradient, split-K: the tiles of ONE K slice are the nine taps of the same activation rows (and
        // share the dY rows) -- put them on one XCD so the slice is fetched into one L2, once.  Workgroups are
        // dealt x-fastest round-robin over the 8 XCDs: linear id L -> XCD L & 7, its j-th workgroup there.
        if ((gridDim.y & 7) == 0) {
            const int L = (int)(blockIdx.y * gridDim.x + blockIdx.x), xcd = L & 7, j = L >> 3, T = (int)gridDim.x;
            gemm_bf16_body<BM, BN, LAYOUT, EPI, NSTAGE, NW, BK, true>(a, j % T, 0, -1, smem, nullptr, (j / T) * 8 + xcd);
            return;
        }
    }
    gemm_bf16_body<BM, BN, LAYOUT, EPI, NSTAGE, NW, BK, true>(a, blockIdx.x, 0, gridDim.x, smem);
}

// Grouped launch: several independent GEMM problems of one layout / epilogue / tile shape in ONE
// grid (the nine dW = X^T dY products of a training step: each alone fills a fraction of the 256
// CUs, together they keep every CU at two resident workgroups).  Problem i owns the workgroups
// [start[i], start[i+1]).
void* gemm_bf16_anatomy() { return MEAS_SYMBOL(g_anat); }      // (nullptr in the product build: no stamp tables, measure.h)
void* gemm_bf16_stamps() { return MEAS_SYMBOL(g_stamps); }
struct GroupedArgs {
    int nprob;
    int kind[DMVAE_MAX_GROUP];
    int start[DMVAE_MAX_GROUP + 1];
    int cls_start[DMVAE_MAX_GROUP], cls_end[DMVAE_MAX_GROUP];
    int nsl[DMVAE_MAX_GROUP];         // K slices of problem i (K / k_split; 1 = none): its workgroups are slice-major, [slice][tile]
    GemmArgs p[DMVAE_MAX_GROUP];
    dmvae_adam_ctx adam;      // DMVAE_EPI_ADAM launches only
    dmvae_finalize_args fin;  // DMVAE_EPI_RELU_MASK launches: fin.nblocks extra workgroups run step_finalize (0 = none)
    int lead, lead_work;      // ... in the first `lead` (fin.nblocks rounded up to 8) workgroup ids of the grid; ADAM launches: the lead_work workgroups of the extra segment, likewise
    // xcut (round 5, weight-gradient groups): the XCD runs are cut over the WHOLE item sequence by streamed bytes instead of per tile-shape class by
    // count: XCD x (workgroup ids with id % 8 == x) owns the contiguous items [xr0[x], xr0[x] + xcnt[x]); the ids beyond its count return at once
    int xcut, xr0[8], xcnt[8];
};
// SHORTK: every problem has K <= 128 (one or two K tiles: the dX of the narrow heads).  Such a workgroup is all prologue
// and epilogue -- what helps is MORE of them per CU: 64x64 tiles on a 2-slot ring = 32 KiB of LDS, four to five workgroups
// per CU instead of two.
template <int BM, int BN, int LAYOUT, int EPI, int NSTAGE, int NW = 4, bool SHORTK = false>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 1) void gemm_bf16_grouped_kernel(GroupedArgs g) {      // (eight waves: <= 128 VGPRs, so that two workgroups share a CU)
    // every problem takes the largest tile its shape divides (traffic per flop ~ (BM+BN)/(BM*BN)):
    // kind 0 = 128x128 / 2 stages, 1 = 128x64 / 3, 2 = 64x64 / 4  -- one LDS array of the largest ring
    __shared__ __attribute__((aligned(16))) bf16_t smem[SHORTK ? 2 * (64 + 64) * BK : 3 * (128 + 64) * BK];   // 72 KiB >= 2*(128+128)*64, 4*(64+64)*64
    const dmvae_adam_ctx* ac = EPI == DMVAE_EPI_ADAM ? &g.adam : nullptr;
    if constexpr (EPI == DMVAE_EPI_ADAM) {
        if ((int)blockIdx.x < g.lead) {              // the extra workgroup(s): arena segment whose gradient is already in memory (first ids, see below)
            if ((int)blockIdx.x >= g.lead_work) return;
            const int64_t n4 = g.adam.seg_n >> 2;
            for (int64_t q = (int64_t)blockIdx.x * (64 * NW) + threadIdx.x; q < n4; q += (int64_t)g.lead_work * (64 * NW)) {
                const int64_t off = g.adam.seg_off + 4 * q;
                const float4 gq = *reinterpret_cast<const float4*>(g.adam.grad + off);
                const float gv[4] = {gq.x, gq.y, gq.z, gq.w};
                adam_quad(g.adam, off, gv);
            }
            return;
        }
    }
    // A CLASS is a run of consecutive problems with the same tile shape (cls[] holds each
    // problem's class bounds as workgroup ids).  The XCD runs are cut over the class, not over
    // each problem: an XCD then works on one or two problems with large blocks of their tiles
    // instead of an eighth of every problem, so far fewer operand panels are fetched by more
    // than one L2.  The workgroup mix per XCD / CU is unchanged (a permutation inside a class).
    int bx = (int)blockIdx.x;
    if constexpr (EPI == DMVAE_EPI_ADAM) bx -= g.lead;
    if constexpr (EPI == DMVAE_EPI_RELU_MASK) {
        // the riding workgroups (loss scalars / Adam step / prior-table gradients: chains of dependent memory latencies, no bandwidth)
        // hold the FIRST ids of the grid, so they run under the tiles instead of behind the last one; g.lead is a multiple of 8, which
        // keeps id % 8 -- the XCD a workgroup lands on -- what the tile mapping below assumes
        if (bx < g.lead) {
            if (NW > 4 && threadIdx.x >= 256) return;    // (written for four waves; a finished wave leaves the barrier count)
            if (bx < g.fin.nblocks) step_finalize_block(bx, g.fin, reinterpret_cast<float(*)[17]>(smem));
            return;
        }
        bx -= g.lead;
    }
    int i = 0, item;
    if (g.xcut) {
        const int x = bx & 7, j = bx >> 3;
        if (j >= g.xcnt[x]) return;
        item = g.xr0[x] + j;
    } else {
        while (i + 1 < g.nprob && bx >= g.start[i + 1]) ++i;
        item = g.cls_start[i] + xcd_run_index(bx, g.cls_start[i], g.cls_end[i]);
    }
    i = 0;
    while (i + 1 < g.nprob && item >= g.start[i + 1]) ++i;
    int bid = item - g.start[i];
    const int kind = g.kind[i];
    const int gs = 0, cnt = -1;                           // the body takes bid as the tile id
    int ksl = -1;                                         // K slice of a split problem (DW / STORE_F32 into slabs), else blockIdx.y = 0
    if (g.nsl[i] > 1) {
        const int tiles_i = (g.start[i + 1] - g.start[i]) / g.nsl[i];
        ksl = bid / tiles_i;
        bid -= ksl * tiles_i;
    }
    MEAS_WG_BEGIN();
    // (a 4-slot ring of K depth 32 in the same 64 KiB was measured for the 128x128 dW tiles: no gain)
    if constexpr (SHORTK) {
        gemm_bf16_body<64, 64, LAYOUT, EPI, 2, NW>(g.p[i], bid, gs, cnt, smem, ac, ksl);
    } else {
        if (kind == 0) gemm_bf16_body<128, 128, LAYOUT, EPI, 2, NW>(g.p[i], bid, gs, cnt, smem, ac, ksl);
        else if (kind == 1) gemm_bf16_body<128, 64, LAYOUT, EPI, 3, NW>(g.p[i], bid, gs, cnt, smem, ac, ksl);
        else if (kind == 3) {      // bias-only strip (see BONLY): M = 64 by construction, one workgroup per 64 columns and K slice
            if constexpr (LAYOUT == DMVAE_GEMM_DW && EPI == DMVAE_EPI_STORE_F32)
                gemm_bf16_body<64, 64, LAYOUT, EPI, 4, NW, BK, false, true>(g.p[i], bid, gs, cnt, smem, ac, ksl);
        }
        else gemm_bf16_body<BM, BN, LAYOUT, EPI, NSTAGE, NW>(g.p[i], bid, gs, cnt, smem, ac, ksl);
    }
    MEAS_WG_END(LAYOUT, kind);
}

// ---------------------------------------------------------------- host side
static int g_thin = 2;                     // tuning knob (dmvae_debug_set_knob 18): thin tiles for the dZ GEMM (gemm_bf16_thin_rows): 0 never, 1 down to 32 rows, 2 down to 16
static int g_group_m = 0;                   // tuning knob (dmvae_debug_set_knob 0): supertile rows, 0 = automatic
// Supertile height.  An XCD owns a run of R = tiles/8 consecutive tile ids and walks it in
// supertiles gm tile-rows high, so its L2 sees ~gm A panels (BM x K each) and ~R/gm B panels
// (BN x K each); fabric traffic ~ gm*BM + (R/gm)*BN is least at gm = sqrt(R*BN/BM).  Measured on
// the step (tools/knob_ab.py 0 ...): fixed 8 -> 0.3356 ms, 4 (= this rule for N = 512) -> 0.3300 ms.
int gemm_auto_group_m(int tiles_m, int tiles_n, int bm, int bn, double run) {
    if (g_group_m > 0) return g_group_m;
    const double R = run > 0.0 ? run : std::max(1.0, tiles_m * (double)tiles_n / 8.0);
    int gm = (int)(std::sqrt(R * bn / bm) + 0.5);
    const int need = (int)((R + tiles_n - 1) / tiles_n);       // rows a run spans anyway
    gm = std::max(gm, need);
    return std::max(1, std::min(gm, tiles_m));
}
static int g_shortk = 0;                    // tuning knob (dmvae_debug_set_knob 7): K <= 128 problems as 64x64 / 2-slot workgroups (four to five per CU).
                                            // MEASURED (tools/knob_ab.py 7 0 1): cfg2 0.2948 (off) vs 0.2985 ms (on), cfg4 0.6834 vs 0.6845: off
static int g_conv_short = 2;                // tuning knob (dmvae_debug_set_knob 5): >= 1 short-K conv tiles as 4-wave / 2-slot workgroups, 2 also 3-slot rings for the 64x64 weight-gradient tiles (tools/cnn_knob.py: 3.539 / 3.290 / 3.270 ms)
static int g_grouped_cls = 1;               // tuning knob (dmvae_debug_set_knob 4): XCD runs cut per tile-shape class (1) or per problem (0)
static int g_dx_group_nw = 0;               // tuning knob (dmvae_debug_set_knob 9): waves per workgroup of a grouped dX launch, 0 = automatic, 4, 8
static int g_grouped_mixed = 1;             // tuning knob (dmvae_debug_set_knob 2): 0 all 64x64, 1 planned per-problem tiles, 2 largest tile each shape divides
static int g_grouped_xcut = 0;              // tuning knob (dmvae_debug_set_knob 20): weight-gradient groups: XCD runs cut by streamed bytes over the whole sequence (1) or per tile-shape class by count (0, default).
                                            // MEASURED (round 5, tools/xcut_ab.sh, profiles/r05_dw_refetch.txt): 1 fetches less -- 651 -> 569 MB per launch at 8192 rows, as tools/dw_refetch_model.py
                                            // predicts -- and is SLOWER: 0.2732 -> 0.2854 ms per step at 4096 rows, 0.6092 -> 0.6463 at 8192, 0.9103 -> 0.9310 at 16 384
template <int BM, int BN, int LAYOUT, int EPI, int NSTAGE, int NW = 4>
static const char* kernel_name(bool grouped) {   // the template instantiation, as rocprofv3 prints it
    static char nm[2][64];
    static bool init = false;
    if (!init) {
        snprintf(nm[0], 64, "gemm_bf16_kernel<%d, %d, %d, %d, %d, %d>", BM, BN, LAYOUT, EPI, NSTAGE, NW);
        snprintf(nm[1], 64, "gemm_bf16_grouped_kernel<%d, %d, %d, %d, %d, %d>", BM, BN, LAYOUT, EPI, NSTAGE, NW);
        init = true;
    }
    return nm[grouped ? 1 : 0];
}
static double gemm_bytes(const GemmArgs& a, int layout = DMVAE_GEMM_FWD) {    // algorithmic: each operand once + the output once
    const int k = a.epi.kind;
    if (k == DMVAE_EPI_ADAM) return 2.0 * ((double)a.M * a.K + (double)a.K * a.N);   // the update's bytes are added per launch
    const double osz = (k == DMVAE_EPI_STORE_F32 || k == DMVAE_EPI_ATOMIC_F32 || k == DMVAE_EPI_BIAS_F32 || k == DMVAE_EPI_BIAS_SIGMOID) ? 4.0 : 2.0;
    if (a.conv_c) {
        // conv mode: the [pixels][9 * Cin] patch matrix is VIRTUAL -- what exists in HBM is the activation, [rows][lda]
        // (lda = its channel count), read once; the weights / the other activation; the real output columns
        const double nv = a.epi.n_valid > 0 ? a.epi.n_valid : a.N;
        if (layout == DMVAE_GEMM_DW) return 2.0 * ((double)a.K * a.lda + (double)a.K * a.ldb) + 4.0 * a.M * nv;
        return 2.0 * ((double)a.M * a.lda + (double)a.K * a.N) + osz * a.M * nv;
    }
    return 2.0 * ((double)a.M * a.K + (double)a.K * a.N) + osz * a.M * a.N;
}

template <int BM, int BN, int LAYOUT, int EPI, int NSTAGE, int NW = 4>
static int launch(hipStream_t s, const GemmArgs& a, int split, const GemmRiders* riders = nullptr) {
    dim3 grid((a.M / BM) * (a.N / BN), split);
    if constexpr (LAYOUT == DMVAE_GEMM_DX && (EPI == DMVAE_EPI_LATENT || EPI == DMVAE_EPI_RELU_MASK)) {
        if (riders && (split == 1 || a.tick)) {      // riders in the first ids of this grid (gemm_bf16_dx_riders_kernel); with K slices (GemmArgs::tick) in slice 0 only
            GemmRiders r = *riders;
            double rbytes = 0.0;
            if (r.ngat) { r.gat.nblocks = r.ngat; rbytes = (double)r.gat.n_valid * r.gat.dim * (4 + 2); }
            ProfScope ps(s, kernel_name<BM, BN, LAYOUT, EPI, NSTAGE, NW>(false), 2.0 * a.M * a.N * (double)a.K, gemm_bytes(a) + rbytes);
            DMVAE_LAUNCH((gemm_bf16_dx_riders_kernel<BM, BN, EPI, NSTAGE, NW>), dim3(grid.x + r.nfin + r.ngat, split), dim3(64 * NW), 0, s, a, r);
            return check_launch("gemm_bf16 (dX + riders)");
        }
    }
    if (riders) { set_error("gemm_bf16: riders are carried by the dense DX launches with the LATENT or RELU_MASK epilogue only"); return DMVAE_EINVAL; }
    ProfScope ps(s, kernel_name<BM, BN, LAYOUT, EPI, NSTAGE, NW>(false), 2.0 * a.M * a.N * (double)a.K, gemm_bytes(a));
    DMVAE_LAUNCH((gemm_bf16_kernel<BM, BN, LAYOUT, EPI, NSTAGE, NW>), grid, dim3(64 * NW), 0, s, a);
    return check_launch("gemm_bf16");
}

template <int BM, int BN, int LAYOUT, int EPI, int NSTAGE, int NW = 4>
static int launch_conv(hipStream_t s, const GemmArgs& a, int split) {
    dim3 grid((a.M / BM) * (a.N / BN), split);
    static const std::string nm = [] {     // (initialised once, thread-safe)
        char b[64];
        snprintf(b, sizeof(b), "gemm_bf16_conv_kernel<%d, %d, %d, %d, %d, %d>", BM, BN, LAYOUT, EPI, NSTAGE, NW);
        return std::string(b);
    }();
    // algorithmic flops: real pixels (no border rows), 9 taps x real channels, real output columns
    const double P = a.conv_p, inner = (P - 2.0) * (P - 2.0) / (P * P), nv = a.epi.n_valid > 0 ? a.epi.n_valid : a.N;
    const double fl = 2.0 * (LAYOUT == DMVAE_GEMM_DW ? (double)a.K : (double)a.M) * inner * 9.0 * a.conv_c * nv;
    ProfScope ps(s, nm.c_str(), fl, gemm_bytes(a, LAYOUT));
    DMVAE_LAUNCH((gemm_bf16_conv_kernel<BM, BN, LAYOUT, EPI, NSTAGE, NW>), grid, dim3(64 * NW), 0, s, a);
    return check_launch("gemm_bf16_conv");
}

// Grouped launch of n independent problems that share a layout and an epilogue kind.  Tile per
// problem = the largest shape it divides, downgraded for the whole group while the grid would
// not give every CU a workgroup.
template <int LAYOUT, int EPI>
static int grouped_launch(hipStream_t s, const GemmArgs* probs, int nprob, const dmvae_adam_ctx* ctx = nullptr, const dmvae_finalize_args* fin = nullptr) {
    auto best_kind = [](const GemmArgs& p) { return p.bias_only ? 3 : (p.M % 128 == 0 && p.N % 128 == 0) ? 0 : (p.M % 128 == 0 ? 1 : 2); };
    auto nslices = [](const GemmArgs& p) { return (p.k_split > 0 && p.k_split < p.K) ? p.K / p.k_split : 1; };
    auto tiles = [&](const GemmArgs& p, int kind) {      // workgroups: tiles x K slices
        if (kind == 3) return (p.N / 64) * nslices(p);
        return (p.M / (kind == 2 ? 64 : 128)) * (p.N / (kind == 0 ? 128 : 64)) * nslices(p);
    };
    // bytes one workgroup of this kind streams into LDS: these kernels run at the per-CU L2->LDS
    // intake rate (~70 GB/s), so a workgroup's duration is proportional to it
    auto wg_bytes = [&](const GemmArgs& p, int kind) {
        if (kind == 3) return 2.0 * 64 * (double)(p.K / nslices(p));
        return 2.0 * ((kind == 2 ? 64 : 128) + (kind == 0 ? 128 : 64)) * (double)(p.K / nslices(p));
    };
    // Tile plan.  Every workgroup is resident at once (<= 2 per CU) and the dispatcher deals them
    // breadth-first in launch order (measured, tools/stamps.py): launched longest-first, workgroup
    // j lands on CU j mod 256, so CU c streams s[c] + s[c+256] + ... bytes.  Every problem may use
    // the largest tile its shape divides or a smaller one (more, shorter workgroups that pack the
    // CUs' second slots evenly): take the assignment with the least-loaded busiest CU, ties to
    // the fewest bytes in total.  <= 3^nprob candidates, searched once per set of shapes.
    int kinds[DMVAE_MAX_GROUP];
    bool shortk = g_shortk && LAYOUT != DMVAE_GEMM_DW && EPI != DMVAE_EPI_ADAM;
    for (int i = 0; i < nprob; ++i) shortk = shortk && probs[i].K <= 128;
    for (int i = 0; i < nprob; ++i) kinds[i] = (g_grouped_mixed && !shortk) ? best_kind(probs[i]) : 2;
    if (g_grouped_mixed == 1 && !shortk) {
        static std::mutex mu;
        static std::map<std::vector<int>, std::vector<int>> memo;
        std::vector<int> key;
        for (int i = 0; i < nprob; ++i) { key.push_back(probs[i].M); key.push_back(probs[i].N); key.push_back(probs[i].K); key.push_back(nslices(probs[i]) * 2 + (probs[i].bias_only ? 1 : 0)); }
        std::lock_guard<std::mutex> lk(mu);
        auto it = memo.find(key);
        if (it == memo.end()) {
            std::vector<int> cur(kinds, kinds + nprob), best = cur;
            double best_max = 1e300, best_sum = 1e300;
            std::vector<double> s;
            for (;;) {
                s.clear();
                double sum = 0.0;
                for (int i = 0; i < nprob; ++i) {
                    const int t = tiles(probs[i], cur[i]);
                    s.insert(s.end(), (size_t)t, wg_bytes(probs[i], cur[i]));
                    sum += t * wg_bytes(probs[i], cur[i]);
                }
                std::sort(s.begin(), s.end(), std::greater<double>());
                double load[256] = {0.0}, mx = 0.0;
                for (size_t j = 0; j < s.size(); ++j) load[j & 255] += s[j];
                for (double l : load) mx = std::max(mx, l);
                if (mx < best_max * 0.99 || (mx < best_max * 1.01 && sum < best_sum)) { best_max = std::min(mx, best_max); best_sum = sum; best = cur; }
                int i = 0;                               // next assignment: odometer over [best_kind, 2]
                while (i < nprob && (cur[i] == 2 || kinds[i] == 3)) { cur[i] = kinds[i]; ++i; }      // (a bias-only strip has one form)
                if (i == nprob) break;
                ++cur[i];
            }
            it = memo.emplace(key, best).first;
        }
        for (int i = 0; i < nprob; ++i) kinds[i] = it->second[i];
    }
    int order[DMVAE_MAX_GROUP];                      // longest workgroups first
    for (int i = 0; i < nprob; ++i) order[i] = i;
    std::stable_sort(order, order + nprob, [&](int x, int y) { return wg_bytes(probs[x], kinds[x]) > wg_bytes(probs[y], kinds[y]); });
    GroupedArgs g;
    g.nprob = nprob;
    int total = 0, n = 0;
    double flops = 0.0, bytes = 0.0;
    for (int o = 0; o < nprob; ++o) {
        const int i = order[o], kind = kinds[i];
        g.start[n] = total;
        g.kind[n] = kind;
        g.nsl[n] = nslices(probs[i]);
        g.p[n] = probs[i];
        g.p[n].group_m = gemm_auto_group_m(probs[i].M / (kind >= 2 ? 64 : 128), probs[i].N / (kind == 0 ? 128 : 64),
                                      kind >= 2 ? 64 : 128, kind == 0 ? 128 : 64);
        total += tiles(probs[i], kind);
        flops += 2.0 * probs[i].M * probs[i].N * (double)probs[i].K;
        bytes += gemm_bytes(probs[i]);
        ++n;
    }
    for (int i = nprob; i < DMVAE_MAX_GROUP; ++i) { g.kind[i] = 2; g.nsl[i] = 1; }
    for (int i = nprob; i <= DMVAE_MAX_GROUP; ++i) g.start[i] = total;
    // classes: runs of consecutive problems with one tile shape (see the kernel); knob 4 = 0 makes
    // every problem its own class (each problem spread over all eight XCDs)
    for (int i = 0; i < DMVAE_MAX_GROUP; ++i) { g.cls_start[i] = g.start[std::min(i, nprob)]; g.cls_end[i] = g.start[std::min(i + 1, nprob)]; }
    if (g_grouped_cls)
        for (int lo = 0; lo < nprob;) {
            int hi = lo + 1;
            // (a class is problems of one tile shape AND comparable K: an XCD's run is a share of the class by workgroup COUNT, so
            //  with the heads' dX pair -- K = 2 D next to K = 64 -- in one class, four XCDs got the long workgroups and four waited)
            while (hi < nprob && g.kind[hi] == g.kind[lo] && 2 * std::min(g.p[hi].K / g.nsl[hi], g.p[lo].K / g.nsl[lo]) > std::max(g.p[hi].K / g.nsl[hi], g.p[lo].K / g.nsl[lo])) ++hi;
            const double run = std::max(1.0, (g.start[hi] - g.start[lo]) / 8.0);
            for (int i = lo; i < hi; ++i) {
                g.cls_start[i] = g.start[lo];
                g.cls_end[i] = g.start[hi];
                const int kind = g.kind[i], bm = kind >= 2 ? 64 : 128, bn = kind == 0 ? 128 : 64;
                const int t = (g.start[i + 1] - g.start[i]) / g.nsl[i];
                g.p[i].group_m = gemm_auto_group_m(g.p[i].M / bm, g.p[i].N / bn, bm, bn, std::min<double>(t, run));
            }
            lo = hi;
        }
    // Weight-gradient groups (round 5, knob 20 = 1; NOT the default -- it fetches less and runs slower, see g_grouped_xcut): cut the XCD runs over the WHOLE sequence by streamed bytes.  Per class by count, a problem that is alone in its
    // class (the 832-wide first / last layer on 64x64 or 128x64 tiles) is spread over all eight XCDs and each of the eight L2s fetches most of its
    // operand panels: tools/dw_refetch_model.py -- at 8192 rows the partition alone makes the launch fetch 481 MB of operands for 309 MB (measured
    // with the update's reads: 647 MB for 384), of which the first layer 77.6 MB for 22.0.  Cut by bytes, every XCD gets one contiguous stretch of the
    // longest-first sequence -- a small problem lands in one or two L2s (floor 411 MB; 734 for 981 at 16 384 rows) -- and the same bytes to stream; the
    // tile COUNT per XCD then differs (128x128 tiles up front, 64x64 at the end), so the grid is 8 x the largest count and the surplus ids return at once.
    g.xcut = 0;
    int grid_items = total;
    if (LAYOUT == DMVAE_GEMM_DW && g_grouped_xcut && total >= 64) {
        double tot_b = 0.0;
        std::vector<double> cost((size_t)total);
        for (int i = 0; i < nprob; ++i)
            for (int j = g.start[i]; j < g.start[i + 1]; ++j) { cost[j] = wg_bytes(g.p[i], g.kind[i]); tot_b += cost[j]; }
        int cuts[9] = {0};
        double acc = 0.0;
        int x = 1;
        for (int j = 0; j < total && x < 8; ++j) {
            acc += cost[j];
            while (x < 8 && acc >= tot_b * x / 8.0 - 1e-6) cuts[x++] = j + 1;
        }
        while (x < 8) cuts[x++] = total;
        cuts[8] = total;
        int mx = 0;
        for (int k = 0; k < 8; ++k) { g.xr0[k] = cuts[k]; g.xcnt[k] = cuts[k + 1] - cuts[k]; mx = std::max(mx, g.xcnt[k]); }
        g.xcut = 1;
        grid_items = 8 * mx;
        for (int i = 0; i < nprob; ++i) {        // supertile height from the largest share of the problem's tiles that one XCD holds
            int share = 1;
            for (int k = 0; k < 8; ++k) share = std::max(share, std::min(cuts[k + 1], g.start[i + 1]) - std::max(cuts[k], g.start[i]));
            const int kind = g.kind[i], bm = kind >= 2 ? 64 : 128, bn = kind == 0 ? 128 : 64;
            const int t = (g.start[i + 1] - g.start[i]) / g.nsl[i];
            g.p[i].group_m = gemm_auto_group_m(g.p[i].M / bm, g.p[i].N / bn, bm, bn, std::min<double>(t, share));
        }
    }
    int extra = 0;
    g.fin = dmvae_finalize_args{};
    g.lead = 0; g.lead_work = 0;
    if (fin && EPI == DMVAE_EPI_RELU_MASK) { g.fin = *fin; g.lead = (fin->nblocks + 7) & ~7; extra = g.lead; }
    g.adam = dmvae_adam_ctx{};
    if (ctx) {
        g.adam = *ctx;
        // the update's traffic: p, m, v read + written, the bf16 shadow written (the gradient stays in registers)
        double elems = (double)ctx->seg_n;
        for (int i = 0; i < nprob; ++i) elems += (double)probs[i].M * probs[i].N + (probs[i].epi.out2 ? probs[i].N : 0);
        bytes += elems * (24.0 + (ctx->param_bf16 ? 2.0 : 0.0) + (ctx->store_grad ? 4.0 : 0.0)) + 4.0 * ctx->seg_n;
        if (ctx->seg_n > 0) { g.lead_work = (int)std::min<int64_t>(4, (ctx->seg_n / 4 + 255) / 256); g.lead = 8; extra = g.lead; }
    }
    // reported under a name that says what the grid is (rocprofv3 prints the <64, 64, ...> instantiation: the 64x64
    // tile is only the smallest of the three the kernel dispatches to per problem)
    static const std::string pname = [] {
        char b[64];
        snprintf(b, sizeof(b), "gemm_bf16_grouped_mixed_tiles<L%d, E%d>", LAYOUT, EPI);
        return std::string(b);
    }();
    ProfScope ps(s, pname.c_str(), flops, bytes);
    // (8-wave workgroups in the grouped grids were measured too: 0.3071 vs 0.3014 ms/step, not kept)
    // (ONE workgroup per CU at a time -- the launch padding its LDS request -- with half of the CUs dealt a long tile first and the others a short one, so
    //  that the long tiles' Adam epilogues fall into two bunches under the other half's K loops, was measured in round 5 for the weight-gradient group:
    //  bit-identical, cfg2 0.2700 -> 0.2943 ms, cfg4 0.6099 -> 0.7127.  Two resident workgroups overlapping their K loops are worth more than un-bunched epilogues.)
    if constexpr (LAYOUT != DMVAE_GEMM_DW) {
        if (shortk) {
            DMVAE_LAUNCH((gemm_bf16_grouped_kernel<64, 64, LAYOUT, EPI, 2, 4, true>), dim3(total + extra), dim3(256), 0, s, g);
            return check_launch("gemm_bf16_grouped");
        }
    }
    // the dX of the two head layers is all mask read and output written (K = 2 D | the padded class count): eight waves per
    // workgroup keep twice the epilogue loads / stores in flight per CU
    if constexpr (LAYOUT == DMVAE_GEMM_DX && EPI == DMVAE_EPI_RELU_MASK) {
        if (g_dx_group_nw == 8) {
            DMVAE_LAUNCH((gemm_bf16_grouped_kernel<64, 64, LAYOUT, EPI, 4, 8>), dim3(total + extra), dim3(512), 0, s, g);
            return check_launch("gemm_bf16_grouped");
        }
    }
    DMVAE_LAUNCH((gemm_bf16_grouped_kernel<64, 64, LAYOUT, EPI, 4, 4>), dim3(grid_items + extra), dim3(256), 0, s, g);
    return check_launch("gemm_bf16_grouped");
}

// instantiated groups: the nine dW of a step (DW / STORE_F32), the sibling head layers
// [mean|log_var] + logits (FWD / BIAS_F32) and their two dX (DX / RELU_MASK)
int gemm_bf16_grouped(hipStream_t s, int layout, const GemmArgs* probs, int nprob, const dmvae_finalize_args* fin) {
    if (nprob < 1 || nprob > DMVAE_MAX_GROUP) { set_error("dmvae_gemm_grouped: 1..%d problems", DMVAE_MAX_GROUP); return DMVAE_EINVAL; }
    const int epi = probs[0].epi.kind;
    for (int i = 1; i < nprob; ++i)
        if (probs[i].epi.kind != epi) { set_error("dmvae_gemm_grouped: all problems must share the epilogue kind"); return DMVAE_EINVAL; }
    if (layout == DMVAE_GEMM_DW && epi == DMVAE_EPI_STORE_F32) return grouped_launch<DMVAE_GEMM_DW, DMVAE_EPI_STORE_F32>(s, probs, nprob);
    if (layout == DMVAE_GEMM_FWD && epi == DMVAE_EPI_BIAS_F32) return grouped_launch<DMVAE_GEMM_FWD, DMVAE_EPI_BIAS_F32>(s, probs, nprob);
    if (fin && !(layout == DMVAE_GEMM_DX && epi == DMVAE_EPI_RELU_MASK)) { set_error("dmvae_gemm_grouped: step_finalize rides on DX / RELU_MASK groups only"); return DMVAE_EINVAL; }
    if (layout == DMVAE_GEMM_DX && epi == DMVAE_EPI_RELU_MASK) {
        // the tiny-K problems (the dX of the two head layers: K = 2 D | the padded class count) stream through heads_dx.hip; the step_finalize
        // riders go with them; what does not qualify (K = 512 at D = 256) stays on the grouped tiles
        bool taken[DMVAE_MAX_GROUP];
        const int rc = heads_dx_stream_launch(s, probs, nprob, fin, taken);
        if (rc) return rc;
        GemmArgs rest[DMVAE_MAX_GROUP];
        int nrest = 0, ntaken = 0;
        for (int i = 0; i < nprob; ++i) {
            if (taken[i]) ++ntaken;
            else rest[nrest++] = probs[i];
        }
        if (nrest == 0) return 0;
        return grouped_launch<DMVAE_GEMM_DX, DMVAE_EPI_RELU_MASK>(s, rest, nrest, nullptr, ntaken ? nullptr : fin);
    }
    set_error("dmvae_gemm_grouped: layout %d with epilogue %d is not instantiated", layout, epi);
    return DMVAE_EUNSUPPORTED;
}
// dW problems large enough for the 256x256 macro tile (gemm_bf16_256.hip) leave the group: each is a launch of its own
// (its grid covers the chip by itself), with the bias gradient from slab column sums and -- under ctx -- its Adam
// update.  Returns the problems that stay grouped.
static int peel_large_dw(hipStream_t s, const GemmArgs* probs, int nprob, const dmvae_adam_ctx* ctx, std::vector<GemmArgs>& rest) {
    std::vector<GemmArgs> large;
    // a problem joins when its own grid covers the chip -- or, once such problems exist, when it merely divides by 256 (K >= 1024):
    // inside the merged grid (gemm_bf16_256_dw_all) a 16..64-tile problem rides at the macro tile's rate instead of the small tiles'
    bool any = false;
    for (int i = 0; i < nprob; ++i)
        any = any || (probs[i].k_split == probs[i].K && gemm_bf16_256_ok(DMVAE_GEMM_DW, probs[i].epi.kind, probs[i].M, probs[i].N, probs[i].K, probs[i].conv_c != 0));
    for (int i = 0; i < nprob; ++i) {
        const GemmArgs& p = probs[i];
        // K slices into slabs (the dW group of a large batch): the caller (csrc/api.hip grad_dense) has taken the bias gradient off the
        // problems it wants on the macro tile (a bias-only strip of the grouped launch produces it)
        if (p.k_split != p.K && !p.bias_only && !p.epi.out2 && p.slab_stride && gemm_bf16_256_slice_ok(p.M, p.N, p.k_split)) { large.push_back(p); continue; }
        const bool own = p.k_split == p.K && gemm_bf16_256_ok(DMVAE_GEMM_DW, p.epi.kind, p.M, p.N, p.K, p.conv_c != 0);
        const bool rides = any && gemm_bf16_256_rides() && p.k_split == p.K && !p.conv_c && p.M % 256 == 0 && p.N % 256 == 0 && p.K >= 1024 &&
                           (p.epi.kind == DMVAE_EPI_STORE_F32 || p.epi.kind == DMVAE_EPI_ADAM);
        if (own || rides) large.push_back(p);
        else rest.push_back(p);
    }
    return large.empty() ? 0 : gemm_bf16_256_dw_all(s, large.data(), (int)large.size(), ctx);      // merged into one grid where possible
}
int gemm_bf16_grouped_dw(hipStream_t s, const GemmArgs* probs, int nprob) {
    if (nprob < 1 || nprob > DMVAE_MAX_GROUP) { set_error("dmvae_gemm_grouped_dw: 1..%d problems", DMVAE_MAX_GROUP); return DMVAE_EINVAL; }
    std::vector<GemmArgs> rest;
    for (int i = 0; i < nprob; ++i)
        if (probs[i].epi.kind != DMVAE_EPI_STORE_F32) return gemm_bf16_grouped(s, DMVAE_GEMM_DW, probs, nprob, nullptr);   // (reports the error)
    const int rc = peel_large_dw(s, probs, nprob, nullptr, rest);
    if (rc) return rc;
    return rest.empty() ? 0 : gemm_bf16_grouped(s, DMVAE_GEMM_DW, rest.data(), (int)rest.size(), nullptr);
}
// the dW group with the Adam update in the epilogue (every problem's epilogue kind = DMVAE_EPI_ADAM)
int gemm_bf16_grouped_dw_adam(hipStream_t s, const GemmArgs* probs, int nprob, const dmvae_adam_ctx& ctx) {
    if (nprob < 1 || nprob > DMVAE_MAX_GROUP) { set_error("dmvae_gemm_grouped_dw_adam: 1..%d problems", DMVAE_MAX_GROUP); return DMVAE_EINVAL; }
    for (int i = 0; i < nprob; ++i)
        if (probs[i].epi.kind != DMVAE_EPI_ADAM || !probs[i].epi.out) { set_error("dmvae_gemm_grouped_dw_adam: every problem needs epilogue kind ADAM and out = its gradient address"); return DMVAE_EINVAL; }
    if (!ctx.param || !ctx.grad || !ctx.m || !ctx.v || !ctx.state || (ctx.seg_off & 3) || (ctx.seg_n & 3) || ctx.seg_n < 0) {
        set_error("dmvae_gemm_grouped_dw_adam: null arena / state, or a segment that is not a multiple of 4 elements");
        return DMVAE_EINVAL;
    }
    for (int i = 0; i < nprob; ++i) {               // the fused update addresses the arenas with 32-bit element offsets (adam_quads)
        const int64_t end = (reinterpret_cast<const float*>(probs[i].epi.out) - ctx.grad) + (int64_t)probs[i].M * probs[i].epi.ldo;
        if (end < 0 || end > ADAM_QUADS_MAX_ELEMS) { set_error("dmvae_gemm_grouped_dw_adam: gradient %d ends %lld elements into the arena (limit 2^30)", i, (long long)end); return DMVAE_EUNSUPPORTED; }
    }
    std::vector<GemmArgs> rest;
    dmvae_adam_ctx c1 = ctx;
    c1.seg_n = 0;                                   // the extra arena segment (prior tables) stays with the grouped launch
    const int rc = peel_large_dw(s, probs, nprob, &c1, rest);
    if (rc) return rc;
    if (!rest.empty()) return grouped_launch<DMVAE_GEMM_DW, DMVAE_EPI_ADAM>(s, rest.data(), (int)rest.size(), &ctx);
    if (ctx.seg_n > 0) {                            // nothing left to ride on: the segment's update as the stand-alone kernel (same bits)
        AdamArgs a;
        a.n = ctx.seg_n; a.p = ctx.param + ctx.seg_off; a.g = ctx.grad + ctx.seg_off; a.m = ctx.m + ctx.seg_off; a.v = ctx.v + ctx.seg_off;
        a.pb = ctx.param_bf16 ? reinterpret_cast<bf16_t*>(ctx.param_bf16) + ctx.seg_off : nullptr;
        a.lr = 0.f; a.b1 = ctx.beta1; a.b2 = ctx.beta2; a.eps = ctx.epsilon; a.gscale = ctx.grad_scale; a.zero_grad = 0; a.ieee = ctx.ieee;
        a.t_host = ~0ull; a.st = reinterpret_cast<const dmvae_state*>(ctx.state);
        return adam_launch(s, a);
    }
    return 0;
}

static int g_force_tile = 0;   // debug override (dmvae_debug_set_tile): BM*1000+BN, 0 = heuristic
void gemm_bf16_force_tile(int t) { g_force_tile = t; }
static int g_nw8 = 1;          // tuning knob (dmvae_debug_set_knob 1): 8-wave workgroups for the 128-row tiles
void gemm_bf16_set_knob(int which, int v) {
    if (which == 0) g_group_m = v < 0 ? 0 : v;
    if (which == 1) g_nw8 = v;
    if (which == 2) g_grouped_mixed = v;
    if (which == 4) g_grouped_cls = v;
    if (which == 5) g_conv_short = v;
    if (which == 6) gemm_bf16_256_set_policy(v);
    if (which == 7) g_shortk = v;
    if (which == 8) gemm_bf16_256_set_stagger(v);
    if (which == 9) g_dx_group_nw = v;
    if (which == 18) g_thin = v;
    if (which == 20) g_grouped_xcut = v;
}

// A thin launch (few 64 x 64 tiles, long K: the dZ GEMM at 4096 rows is 64 tiles of 32 K tiles) is bound by the per-CU intake of its activations --
// written by the previous launch from other XCDs, nothing of them in this XCD's L2: ~35 GB/s per CU -- times the CUs it occupies (MEASURED round 4:
// neither an 8-slot ring nor two K pipelines per workgroup moved it).  32- / 16-row tiles put two / four times the CUs on the same bytes; the
// weight panel each workgroup re-reads comes from L2.  The dZ launch at 4096 rows (rocprofv3, 302 replays): 11.49 us on 64 tiles of 64 rows,
// 10.19 on 128 of 32 rows, 8.92 on 256 of 16 rows (two waves per workgroup; the step_finalize blocks then ride in the output layer's dX launch).
static int gemm_bf16_thin_rows(const GemmArgs& a) {      // 0: the general tiles; 32 / 16: rows of the thin tile -- the thinnest that still fits one round of 256 CUs
    if (!g_thin || a.conv_c || a.N != 64 || a.K < 1024 || (a.k_split != a.K && !a.tick)) return 0;      // (K slices joined by tickets keep the thin tile)
    if (g_thin >= 2 && a.M % 16 == 0 && a.M / 16 <= 256) return 16;
    if (a.M % 32 == 0 && a.M / 32 <= 256) return 32;
    return 0;
}

// Tile choice, BM*1000+BN.  These GEMMs run at the per-CU L2->LDS streaming rate, so the figure
// of merit is (flops per byte loaded) x (fraction of the 256 CUs that get a workgroup):
//   intensity ~ BM*BN/(BM+BN):  128x128 64, 128x64 / 64x128 42.7, 64x64 32.
// M, N are multiples of 64.
int gemm_bf16_tile_m(int M, int N, int split) {
    if (g_force_tile) {
        const int bm = g_force_tile / 1000, bn = g_force_tile % 1000;
        if (M % bm == 0 && N % bn == 0) return g_force_tile;
    }
    static const int cand[4][2] = {{128, 128}, {128, 64}, {64, 128}, {64, 64}};
    double best = -1.0;
    int pick = 64 * 1000 + 64;
    for (auto& c : cand) {
        if (M % c[0] || N % c[1]) continue;
        const double wgs = (double)(M / c[0]) * (N / c[1]) * split;
        const double score = (double)c[0] * c[1] / (c[0] + c[1]) * (wgs >= 256.0 ? 1.0 : wgs / 256.0);
        if (score > best * 1.0001) { best = score; pick = c[0] * 1000 + c[1]; }
    }
    // one 128x128 workgroup per CU (no second resident workgroup to overlap its load / multiply / store phases with) loses to two
    // 128x64 ones: MEASURED on the step at 8192 rows (tools/tile_step.py cfg4 0,0 128,64: 0.6308 -> 0.6226 ms with every launch on
    // 128x64; the 500-wide layers are the ones this rule moves)
    // (the same step one size down -- 64x64 instead of a single 128x64 workgroup per CU, the 500-wide layers at 4096 rows -- loses:
    //  0.2741 -> 0.2819 ms)
    if (pick == 128128 && (double)(M / 128) * (N / 128) * split <= 256.0 && (double)(M / 128) * (N / 64) * split >= 384.0) pick = 128064;
    return pick;
}

// Ring depth (measured, tools/gemm_sweep.py): two workgroups per CU with a shallow ring each
// beat one workgroup with a deep ring on every shape of the step (the K loop is issue-bound at
// one wave per SIMD; a second resident workgroup fills the gaps):
//   128x128: 2 x 32 KiB, 128x64 / 64x128: 3 x 24 KiB, 64x64: 4 x 16 KiB  (<= 72 KiB -> 2 blocks/CU)
template <int LAYOUT, int EPI>
static int launch_tiled(hipStream_t s, const GemmArgs& a0, int split, const GemmRiders* riders = nullptr) {
    GemmArgs a = a0;
    int t = gemm_bf16_tile_m(a.M, a.N, split);
    if (a.conv_c) {          // conv mode (csrc/conv.hip): the three (layout, epilogue) pairs a convolution layer uses
        constexpr bool ok = (LAYOUT == DMVAE_GEMM_FWD && EPI == DMVAE_EPI_BIAS_RELU) || (LAYOUT == DMVAE_GEMM_DX && EPI == DMVAE_EPI_RELU_MASK) ||
                            (LAYOUT == DMVAE_GEMM_DW && (EPI == DMVAE_EPI_ATOMIC_F32 || EPI == DMVAE_EPI_STORE_F32));
        if constexpr (ok) {
            if (LAYOUT == DMVAE_GEMM_DW) t = 64 * 1000 + (a.N % 128 == 0 ? 128 : 64);   // a tile row stays inside one tap
            a.group_m = gemm_auto_group_m(a.M / (t / 1000), a.N / (t % 1000), t / 1000, t % 1000);
            if constexpr (LAYOUT == DMVAE_GEMM_DW) {
                if (t == 64128) return launch_conv<64, 128, LAYOUT, EPI, 3>(s, a, split);
                if (g_conv_short == 2) return launch_conv<64, 64, LAYOUT, EPI, 3>(s, a, split);     // 48 KiB ring -> three workgroups per CU
                return launch_conv<64, 64, LAYOUT, EPI, 4>(s, a, split);
            } else {
                switch (t) {
                    case 128128: return launch_conv<128, 128, LAYOUT, EPI, 2, 8>(s, a, split);
                    case 128064:
                        // short K (the 32-channel layers: 5 K tiles): a 2-slot ring and 4 waves -> three workgroups per CU
                        if (a.K <= 640 && g_conv_short) return launch_conv<128, 64, LAYOUT, EPI, 2, 4>(s, a, split);
                        return launch_conv<128, 64, LAYOUT, EPI, 3, 8>(s, a, split);
                    case 64128: return launch_conv<64, 128, LAYOUT, EPI, 3>(s, a, split);
                    default: return launch_conv<64, 64, LAYOUT, EPI, 4>(s, a, split);
                }
            }
        } else {
            set_error("dmvae_gemm(bf16): conv mode is built for FWD+BIAS_RELU, DX+RELU_MASK and DW+ATOMIC_F32");
            return DMVAE_EUNSUPPORTED;
        }
    }
    // (256x128 tiles -- one 8-wave workgroup per CU, 3 x 48 KiB ring, 25 % fewer bytes per flop -- were
    //  measured three times and not kept: at M = 4096 the 2048-wide forward layer 16.7 vs 15.9 us, its dX
    //  17.1 vs 19.0 us; whole step at B = 16384: 1.0753 vs 1.0706 ms, at B = 8192 / D 256 / K 50: 0.8079 vs
    //  0.7922 ms.  Two resident workgroups overlapping their load / compute / store phases beat the
    //  smaller intake of one.)
    if (riders && (a.conv_c || LAYOUT != DMVAE_GEMM_DX || gemm_bf16_256_ok(LAYOUT, EPI, a.M, a.N, a.K, false))) { set_error("gemm_bf16: riders are carried by the dense small-tile DX launches only (ask gemm_bf16_riders_room first)"); return DMVAE_EINVAL; }
    if (split == 1 && gemm_bf16_256_ok(LAYOUT, EPI, a.M, a.N, a.K, false)) return gemm_bf16_256_launch(s, LAYOUT, a);
    a.group_m = gemm_auto_group_m(a.M / (t / 1000), a.N / (t % 1000), t / 1000, t % 1000);
    if (g_shortk && a.K <= 128 && split == 1 && LAYOUT != DMVAE_GEMM_DW && EPI != DMVAE_EPI_BIAS_RECON) {
        // one or two K tiles: the workgroup is prologue + epilogue; small LDS footprint -> more workgroups per CU
        a.group_m = gemm_auto_group_m(a.M / 64, a.N / 64, 64, 64);
        return launch<64, 64, LAYOUT, EPI, 2>(s, a, split, riders);
    }
    // (A single DEEP ring for launches of at most one workgroup per CU -- 4 x 32 KiB for 128x128, 6 x 24 KiB for 128x64, i.e. one
    //  workgroup per CU by LDS -- was knob 3 until round 4: 0.2926 vs 0.2860 ms per step at cfg2; the K loop of a 4096x512x512 layer
    //  took 2.92 vs 2.96 us and its first tile landed 0.5 us later.  Removed with its 18 instantiations, which also could not meet
    //  the two-workgroups-per-CU register bound of the kernel template.)
    if constexpr (LAYOUT == DMVAE_GEMM_DX && EPI == DMVAE_EPI_LATENT) {
        const int thin = gemm_bf16_thin_rows(a);   // the dZ GEMM on 32- or 16-row tiles: two / four times the CUs streaming its activations
        if (thin == 32) {
            a.group_m = gemm_auto_group_m(a.M / 32, a.N / 64, 32, 64);
            return launch<32, 64, LAYOUT, EPI, 4>(s, a, split, riders);
        }
        if (thin == 16) {
            if (riders && riders->nfin) { set_error("gemm_bf16: the 16-row tile (two waves) cannot carry the step_finalize blocks"); return DMVAE_EINVAL; }
            a.group_m = gemm_auto_group_m(a.M / 16, a.N / 64, 16, 64);
            return launch<16, 64, LAYOUT, EPI, 4, 2>(s, a, split, riders);
        }
    }
    switch (t) {
        case 128128:
            return g_nw8 ? launch<128, 128, LAYOUT, EPI, 2, 8>(s, a, split, riders) : launch<128, 128, LAYOUT, EPI, 2>(s, a, split, riders);
        case 128064:
            // (three workgroups per CU on a 2-slot 48 KiB ring: 0.331 vs 0.300 ms/step -- the long-K layers need the third slot)
            return g_nw8 ? launch<128, 64, LAYOUT, EPI, 3, 8>(s, a, split, riders) : launch<128, 64, LAYOUT, EPI, 3>(s, a, split, riders);
        case 64128: return launch<64, 128, LAYOUT, EPI, 3>(s, a, split, riders);
        default: return launch<64, 64, LAYOUT, EPI, 4>(s, a, split, riders);
    }
}

// Workgroup slots a dense small-tile DX launch of this shape leaves free for riders that want a CU of their own (256 - tiles), or that are
// content to be the second resident workgroup of a CU (512 - tiles); < 0: the launch runs on another kernel (macro tile, conv) -- no riders.
int gemm_bf16_riders_room(const GemmArgs& a, bool own_cu) {
    if (a.conv_c || gemm_bf16_256_ok(DMVAE_GEMM_DX, a.epi.kind, a.M, a.N, a.K, false)) return -1;
    if (a.epi.kind != DMVAE_EPI_LATENT && a.epi.kind != DMVAE_EPI_RELU_MASK) return -1;
    int t = gemm_bf16_tile_m(a.M, a.N, 1);
    if (g_shortk && a.K <= 128) t = 64064;
    if (a.epi.kind == DMVAE_EPI_LATENT && gemm_bf16_thin_rows(a)) t = gemm_bf16_thin_rows(a) * 1000 + 64;
    const int tiles = (a.M / (t / 1000)) * (a.N / (t % 1000));
    return (own_cu ? 256 : 512) - tiles;
}

bool gemm_bf16_carries_finalize(const GemmArgs& a) { return !(a.epi.kind == DMVAE_EPI_LATENT && gemm_bf16_thin_rows(a) == 16); }

int gemm_bf16_dispatch(hipStream_t s, int layout, const GemmArgs& a, int split, const GemmRiders* riders) {
    const int epi = a.epi.kind;
#define CASE(L, E) \
    if (layout == L && epi == E) return launch_tiled<L, E>(s, a, split, riders);
    CASE(DMVAE_GEMM_FWD, DMVAE_EPI_BIAS_RELU)
    CASE(DMVAE_GEMM_FWD, DMVAE_EPI_BIAS_F32)
    CASE(DMVAE_GEMM_FWD, DMVAE_EPI_BIAS_RECON)
    CASE(DMVAE_GEMM_FWD, DMVAE_EPI_BIAS_SIGMOID)
    CASE(DMVAE_GEMM_FWD, DMVAE_EPI_STORE_F32)
    CASE(DMVAE_GEMM_DX, DMVAE_EPI_STORE_F32)
    CASE(DMVAE_GEMM_DX, DMVAE_EPI_RELU_MASK)
    CASE(DMVAE_GEMM_DX, DMVAE_EPI_LATENT)
    CASE(DMVAE_GEMM_DW, DMVAE_EPI_STORE_F32)
    CASE(DMVAE_GEMM_DW, DMVAE_EPI_ATOMIC_F32)
#undef CASE
    set_error("dmvae_gemm(bf16): layout %d with epilogue %d is not instantiated", layout, epi);
    return DMVAE_EUNSUPPORTED;
}

}  // namespace dmvae
