/*
 * dmvae_hip_debug.h -- measurement and tuning entry points of libdmvae_hip.so.
 *
 * NOT part of the drop-in boundary (include/dmvae_hip.h): nothing here replaces a line of the reference.
 * bench.py's roofline leg uses dmvae_prof_* / dmvae_debug_spin; tools/ uses the probes and knobs.
 */
#ifndef DMVAE_HIP_DEBUG_H
#define DMVAE_HIP_DEBUG_H

#include "dmvae_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- per-kernel timing with HIP events (bench.py roofline leg) ----------
 * enable(1): every launch made through this library is bracketed by a
 * hipEvent pair on its stream AND issued through hipExtLaunchKernelGGL with an
 * event pair bound to the dispatch itself.  collect() synchronises those events and
 * returns, per kernel family, launches / bracket ms / kernel ms / algorithmic flops / bytes.
 * Eager launches only (not under stream capture). */
typedef struct dmvae_prof_row {
    char name[48];
    int64_t launches;          /* profiled scopes (one per library call that launches this kernel family)       */
    double total_ms;           /* event BRACKETS around them: kernel + its dispatch / event markers              */
    double flops;   /* algorithmic */
    double bytes;   /* algorithmic */
    double kernel_ms;          /* the kernels' own begin -> end (the pair hipExtLaunchKernelGGL binds to a       */
    int64_t kernel_launches;   /* dispatch = what rocprofv3 --kernel-trace reports), and how many dispatches;    */
                               /* 0 when the runtime returned no dispatch timestamps                              */
} dmvae_prof_row;
int dmvae_prof_enable(int on);
/* keeps `stream` busy for ~microseconds (<= 20 ms) so the host can run ahead of the GPU and the
 * event brackets of the following launches contain no host launch latency */
int dmvae_debug_spin(void* stream, int microseconds);
int dmvae_prof_collect(dmvae_prof_row* rows, int max_rows);   /* returns number of rows */
/* measurement kernel (csrc/strip_fwd2.hip; tools/strip2_probe.py): Y1 = relu(X . W0 + b0), Y2 = relu(Y1 . W1 + b1), two 512-wide dense layers
 * (base_models.py:220-226) as ONE row-strip launch -- 32 rows per workgroup, Y1 kept in LDS (and written), both weight matrices streamed from L2;
 * bf16 operands, B_pad % 32 == 0, K0 % 64 == 0; same bits as two dmvae_gemm calls with DMVAE_EPI_BIAS_RELU.  Not on the step path. */
int dmvae_debug_strip_fwd2(void* stream, int B_pad, int K0, const void* X, int64_t ldx, const void* W0, int64_t ld0, const float* b0,
                           const void* W1, int64_t ld1, const float* b1, void* Y1, int64_t ldy1, void* Y2, int64_t ldy2);
/* measurement builds only (tools/ablate.sh 6): device pointer of the per-workgroup stamp table of the
 * grouped GEMM kernel, 2048 x {start, end (100 MHz ticks), HW_ID<<32 | XCC_ID, layout<<32 | tile kind};
 * the product library never writes it */
int dmvae_debug_stamps(void** device_ptr);
/* DMVAE_ABLATE=6 builds: per-workgroup phase stamps of the last small-tile bf16 GEMM launch, 2048 x 8 uint64 (100 MHz ticks):
 * {entry, first K tile landed, K loop done, epilogue issued, stores acknowledged, HW_ID << 32 | XCC_ID} (tools/anatomy.py) */
int dmvae_debug_anatomy(void** device_ptr);
/* likewise for the last 256x256 macro-tile launch: 4096 x 8 uint64 {entry, K loop done, epilogue done (100 MHz ticks), HW_ID << 32 | XCC_ID,
 * entry, K loop done (shader cycles, s_memtime), 0, 0} (tools/anatomy256.py; tools/clock256.py: the clock held inside the K loop) */
int dmvae_debug_anatomy256(void** device_ptr);
/* tuning aid: force the bf16 GEMM tile (64|128 x 64|128); (0,0) restores the heuristic */
int dmvae_debug_set_tile(int bm, int bn);
/* tuning aid: knob 0 = supertile height (tile rows) of the L2-friendly tile order,
 *             knob 1 = 8-wave workgroups for the 128-row tiles (0|1),
 *             knob 2 = per-problem tile shapes in the grouped dW grid (0 = all 64x64, 1 = planned, 2 = largest),
 *             (knob 3, the deep-ring policy, was removed in round 4 with its instantiations: measured slower, see csrc/gemm_bf16.hip;
 *              setting it is refused),
 *             knob 4 = XCD runs of a grouped grid cut per tile-shape class (1) or per problem (0),
 *             knob 5 = conv-mode tiles: >= 1 short-K tiles as 4-wave / 2-slot workgroups (three per CU),
 *                      2 also 3-slot rings for the 64x64 weight-gradient tiles,
 *             knob 6 = 256x256 macro-tile kernel (csrc/gemm_bf16_256.hip): 0 never, 1 when its grid covers the chip
 *                      (default), 2 whenever M and N divide by 256,
 *             knob 7 = problems with K <= 128 as 64x64 tiles on a 2-slot ring (32 KiB: four to five workgroups per CU) (0|1;
 *                      default 0: measured slower on the whole step),
 *             knob 8 = merged weight-gradient grid of the 256x256 kernel: first-tile delay, units of 3.4 us spread over the
 *                      256 CUs (default 0 = none, measured best; -1 = one launch per problem),
 *             knob 10 = K slices of the dense weight-gradient group of plans with >= 8192 batch rows (slabs + fixed-order sum in the
 *                      Adam kernel): 0 = the plan's rule (4 where the 256-divisible layers take the macro tile, else 2 from 16384 rows, else none),
 *                      1 = none, 2 | 4 = forced,
 *             knob 11 = with K slices: the 256-divisible layers' slices on the 256x256 macro tile (1, default) or everything on the small tiles (0)
 *             knob 9  = waves per workgroup of a grouped dX launch (0 / 4 = four, 8 = eight at <= 128 VGPRs; tools/heads_dx_probe.py, tools/knob_step.py:
 *                       85.6 vs 90.0 us alone at 16384 rows, 0.9785 vs 0.9651 ms in the step -- four)
 *             knob 12 = the dX of the two head layers as one grouped launch (0 / 1, default) or as two launches (2: 0.9424 vs 0.9366 ms at 16384 rows),
 *             knob 13 = that dX on the streaming kernel (csrc/heads_dx.hip; 1, default, when every problem of the group has K = 64 / 128 / 256) or
 *                       on the grouped tiles (0): the pair alone 20.4 -> 17.4 us at 4096 rows, 80.3 -> 76.8 us at 16384; step -0.6 % at 16384 rows,
 *             knob 14 = blocks the latent kernel's geometry aims at (512; rows per block = 16 .. 64).  The plan sizes its partial sums for the largest
 *                       count and takes the count at enqueue time, so it may change while a plan exists.  1024 at 16384 rows: 0.9500 vs 0.9493 ms,
 *             knob 16 = where the step_finalize blocks run: 1 (default) as riders of the dZ GEMM while tiles + riders <= 256 (4096 rows), else of the
 *                       heads' dX launch; 2 always the heads' dX launch; 0 a launch of their own (0.2796 vs 0.2753 ms at 4096 rows),
 *             knob 17 = where the gather of a prefetched batch (dmvae_plan_prefetch_batch) runs: 1 (default) on the idle CUs of the dZ GEMM, one block
 *                       each, in the last ids of its grid (12.98 us for that launch at 4096 rows, 11.2 without riders); 2 the same in the first ids
 *                       (14.1 us); 3 two blocks per idle CU, last ids (15.5 us); 0 a launch of its own in front of the trunk's backward pass.
 *                       (Those figures: the 64-row dZ tiles of knob 18 = 0.  With the default thin dZ tiles a 4096-row launch is 256 workgroups and has
 *                       no room for riders, so there the gather is a launch of its own whatever this knob says; riders run at <= 2048 rows.)
 *             knob 18 = thin tiles for the dZ GEMM (a 64-wide output with K >= 1024): 2 (default) the thinnest of 16 / 32 rows that still fits one
 *                       round of 256 CUs, 1 down to 32 rows only, 0 the general 64 x 64 tiles (that launch at 4096 rows: 8.92 / 10.19 / 11.49 us),
 *             knob 19 = heads forward + latent stage as ONE launch (csrc/heads_latent.hip) where it applies -- bf16, at most 4096 rows, Dp <= 128, K * D < 4096 -- (1,
 *                       default) or as the grouped heads GEMM + latent_fwd_kernel (0): that pair 26.3 us, the fused launch 25.3 us at 4096 rows; step -0.3 .. -0.8 %,
 *             knob 20 = XCD partition of the grouped weight-gradient launch: runs per tile-shape class by count (0, default) or cut over the whole sequence by streamed
 *                       bytes (1: fetches 13 % less at 8192 rows and is 2 .. 6 % slower on the step at every size: profiles/r05_dw_refetch.txt),
 *             knob 21 = K slices of the thin launches of a small batch (<= 256 rows: the dense layers with K >= 2048 and the dZ GEMM; up to 2048 rows: the fused
 *                       heads + latent launch while blocks x slices <= 256), joined by the last workgroup to arrive (GemmArgs::tick): 1 (default) slices of
 *                       512, 2 slices of 256 (0.1397 vs 0.1402 ms at 100 rows: the same), 0 none */
int dmvae_debug_set_knob(int which, int value);

#ifdef __cplusplus
}
#endif
#endif /* DMVAE_HIP_DEBUG_H */
