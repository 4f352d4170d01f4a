// Measurement scaffolding of the kernels, in ONE place.  The product library is built with DMVAE_ABLATE == 0: every
// MEAS_* stamp macro below is then empty, every MEAS_NO_* predicate is a constant false (the branch it guards is discarded at
// compile time) and none of the stamp tables exists -- the product kernels execute no stamp and carry no table.
//
// tools/ablate.sh N builds a separately named library (build/libdmvae_hip_abl<N>.so, never the product one, selected with
// DMVAE_HIP_LIB) with -DDMVAE_ABLATE=N.  Results of builds 1-5, 8, 9, 11 and 12 are WRONG by construction: timing only.
//   1  no MFMA (fragments still read)            2  no LDS fragment reads            3  no global -> LDS loads inside the K loop
//   4  = 1 + 2                                   5  no epilogue (accumulators kept live)
//   6  per-workgroup stamps: tools/stamps.py (placement, K-loop / epilogue timeline of a grouped launch), tools/anatomy.py (phases
//      of a small GEMM), tools/anatomy256.py / tools/clock256.py (merged macro-tile dW grid; the clock held inside its K loop)
//   7  phase timeline of block 0 of the latent kernel (tools/latent_time.py)
//   8  no ReLU-mask read in the dX epilogue (the upper bound of what a 1-bit mask could save)
//   9  no bias-gradient MFMAs in the first tile row of the weight-gradient tiles (the upper bound of moving that work elsewhere; bias gradients are then zero)
//  10  phase split of every workgroup of heads_dx.hip (tools/heads_dx_phases.py): shader-clock sums per phase, kept in registers
//      until the workgroup ends (a stamp STORE inside the walk would count in the walk's own vmcnt arithmetic)
//  11  the batch gather of dmvae_plan_load_batch_step not enqueued after its first two calls (the step then trains on a stale batch): the upper
//      bound of a gather that runs under another launch (the next batch fetched by riders of the weight-gradient launch)
//  12  the gather blocks that ride in the dZ launch (a prefetched batch) return at once: what their presence costs that launch without their work
// Stamp values go to tables of their own (or, build 7, behind the loss partials in the caller's buffer); no output is computed
// from them (MI355X_MICROARCH.md, DVFS give-back item 6).
#pragma once

#ifndef DMVAE_ABLATE
#define DMVAE_ABLATE 0
#endif

namespace dmvae {
constexpr bool MEAS_NO_MFMA = DMVAE_ABLATE == 1 || DMVAE_ABLATE == 4;
constexpr bool MEAS_NO_LDS_READ = DMVAE_ABLATE == 2 || DMVAE_ABLATE == 4;
constexpr bool MEAS_NO_KLOOP_LOADS = DMVAE_ABLATE == 3;
constexpr bool MEAS_NO_EPILOGUE = DMVAE_ABLATE == 5;
constexpr bool MEAS_NO_MASK_READ = DMVAE_ABLATE == 8;
constexpr bool MEAS_NO_BIAS_MFMA = DMVAE_ABLATE == 9;
constexpr bool MEAS_STAMPS = DMVAE_ABLATE == 6;
constexpr bool MEAS_NO_STEP_GATHER = DMVAE_ABLATE == 11;
constexpr bool MEAS_EMPTY_GATHER_RIDERS = DMVAE_ABLATE == 12;
}  // namespace dmvae

#if DMVAE_ABLATE == 6
// gemm_bf16.hip: g_anat[wg][8] = {entry, first K tile landed, K loop done, epilogue issued, stores acknowledged, HW_ID << 32 | XCC_ID};
// g_mid[wg] = end of the K loop; g_stamps[wg][4] = {begin, end, HW_ID << 32 | XCC_ID, K-loop ticks << 16 | layout << 8 | tile kind}
#define MEAS_TABLES_BF16                                  \
    __device__ unsigned long long g_mid[2048];            \
    __device__ unsigned long long g_anat[2048 * 8];       \
    __device__ unsigned long long g_stamps[2048 * 4];
#define MEAS_HWID() (((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32) | (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20))
#define MEAS_ANAT(i) do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_anat[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define MEAS_KLOOP_END() do { if (threadIdx.x == 0 && blockIdx.x < 2048) g_mid[blockIdx.x] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define MEAS_ANAT_DRAIN()                                                                             \
    do {                                                                                              \
        MEAS_ANAT(3);                                                                                 \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                              \
        MEAS_ANAT(4);                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 2048) g_anat[blockIdx.x * 8 + 5] = MEAS_HWID();          \
    } while (0)
#define MEAS_WG_BEGIN() const unsigned long long meas_t0_ = __builtin_amdgcn_s_memrealtime()
#define MEAS_WG_END(layout, kind)                                                                                          \
    do {                                                                                                                   \
        if (threadIdx.x == 0 && blockIdx.x < 2048) {                                                                       \
            unsigned long long* st_ = g_stamps + 4 * blockIdx.x;                                                           \
            st_[0] = meas_t0_;                                                                                             \
            st_[1] = __builtin_amdgcn_s_memrealtime();                                                                     \
            st_[2] = MEAS_HWID();                                                                                          \
            st_[3] = ((g_mid[blockIdx.x] - meas_t0_) << 16) | ((unsigned long long)(layout) << 8) | (unsigned)(kind);       \
        }                                                                                                                  \
    } while (0)
// gemm_bf16_256.hip: g_anat256[wg][8] = {entry, K loop done, epilogue done (100 MHz ticks), HW_ID << 32 | XCC_ID,
//                                        entry, K loop done (shader cycles, s_memtime), 0, 0}
#define MEAS_TABLES_256 __device__ unsigned long long g_anat256[4096 * 8];
#define MEAS_ANAT256(i)                                                                                  \
    do {                                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                                     \
            g_anat256[blockIdx.x * 8 + (i)] = __builtin_amdgcn_s_memrealtime();                          \
            if ((i) < 2) g_anat256[blockIdx.x * 8 + 4 + (i)] = __builtin_amdgcn_s_memtime();             \
        }                                                                                                \
    } while (0)
#define MEAS_ANAT256_DRAIN()                                                                             \
    do {                                                                                                 \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                 \
        __syncthreads();                                                                                 \
        MEAS_ANAT256(2);                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 4096) g_anat256[blockIdx.x * 8 + 3] = MEAS_HWID();          \
    } while (0)
#define MEAS_SYMBOL(sym) ([] { void* p_ = nullptr; return hipGetSymbolAddress(&p_, HIP_SYMBOL(sym)) == hipSuccess ? p_ : nullptr; }())
#else
#define MEAS_TABLES_BF16
#define MEAS_TABLES_256
#define MEAS_ANAT(i) do { } while (0)
#define MEAS_KLOOP_END() do { } while (0)
#define MEAS_ANAT_DRAIN() do { } while (0)
#define MEAS_WG_BEGIN() do { } while (0)
#define MEAS_WG_END(layout, kind) do { } while (0)
#define MEAS_ANAT256(i) do { } while (0)
#define MEAS_ANAT256_DRAIN() do { } while (0)
#define MEAS_SYMBOL(sym) (static_cast<void*>(nullptr))      // the product build has no stamp tables: dmvae_debug_* report DMVAE_ESTATE
#endif

#if DMVAE_ABLATE == 10  // heads_dx.hip: g_hdx[wg][16] = {entry, W landed, W fragments read + tile 0's gates parked, end of walk, stores acknowledged (s_memtime cycles),
                       //   sums over the steps: [5] wait for dY (first step only), [6] barrier behind it, [7] fragments .. park barrier, [8] epilogue (stores issued),
                       //   [9] wait for the next tile's gates + their park; [10], [11] entry and end in 100 MHz ticks, [12] HW_ID << 32 | XCC_ID, [13] steps}
#define MEAS_TABLES_HDX __device__ unsigned long long g_hdx[1024 * 16];
#define MEAS_HDX_BEGIN() unsigned long long hx_[12] = {}; unsigned long long hx_last_; hx_[0] = __builtin_amdgcn_s_memtime(); hx_[10] = __builtin_amdgcn_s_memrealtime()
#define MEAS_HDX_MARK(i) do { hx_last_ = __builtin_amdgcn_s_memtime(); hx_[i] = hx_last_; } while (0)
#define MEAS_HDX_ACC(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); hx_[i] += t_ - hx_last_; hx_last_ = t_; } while (0)
#define MEAS_HDX_END(steps)                                                                                                        \
    do {                                                                                                                           \
        MEAS_HDX_MARK(3);                                                                                                          \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                           \
        MEAS_HDX_MARK(4);                                                                                                          \
        hx_[11] = __builtin_amdgcn_s_memrealtime();                                                                                \
        if (threadIdx.x == 0 && blockIdx.x < 1024) {                                                                               \
            for (int i_ = 0; i_ < 12; ++i_) g_hdx[blockIdx.x * 16 + i_] = hx_[i_];                                                 \
            g_hdx[blockIdx.x * 16 + 13] = (unsigned long long)(steps);                                                             \
            g_hdx[blockIdx.x * 16 + 12] = (((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) << 32) | (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20)); \
        }                                                                                                                          \
    } while (0)
#define MEAS_SYMBOL_HDX(sym) ([] { void* p_ = nullptr; return hipGetSymbolAddress(&p_, HIP_SYMBOL(sym)) == hipSuccess ? p_ : nullptr; }())
#else
#define MEAS_TABLES_HDX
#define MEAS_HDX_BEGIN() do { } while (0)
#define MEAS_HDX_MARK(i) do { } while (0)
#define MEAS_HDX_ACC(i) do { } while (0)
#define MEAS_HDX_END(steps) do { } while (0)
#define MEAS_SYMBOL_HDX(sym) (static_cast<void*>(nullptr))
#endif

#if DMVAE_ABLATE == 7   // latent.hip: phase timeline of block 0, behind the loss partials (100 MHz ticks)
#define MEAS_LAT_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) reinterpret_cast<unsigned long long*>(L.a.loss_partials + 2 * gridDim.x)[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define MEAS_LAT_STAMP(i) do { } while (0)
#endif
