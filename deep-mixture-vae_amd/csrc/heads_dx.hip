// Streaming form of the dX of the two head layers (round 4): d(z-hidden) = d[mean|log_var] . W_mv^T and d(c-hidden) = dlogits . W_l^T
// through the ReLU mask of [hz | hc] -- the tf.gradients of code/base_models.py:229-248 with respect to the two head hidden layers.
//
// Why its own kernel.  These two problems have K = 2 D (128 .. 512) and K = the padded class count (64) against N = 2048 outputs per
// row: one to four K tiles of multiply work per 128 x 128 tile, against B x 2048 x 2 B of mask read and as much output written per
// problem.  As tiles of the general kernel (gemm_bf16.hip) a workgroup is [operand loads 2 us] -> [K loop <= 1 us] -> [epilogue + store
// drain 3 us]: the memory system is busy in the first and last part only, and the launch moved its bytes at 2.4-3.3 TB/s where a copy of
// the same bytes runs at 6.3-7.1 TB/s (tools/heads_dx_probe.py; DESIGN_LOG.md).  Here a workgroup overlaps its own phases across tiles:
//   * it owns a COLUMN SLICE of one problem (CW = 128 or 64 columns of W: 16 - 32 KB) and loads that slice's MFMA fragments into
//     REGISTERS once; W is never touched again;
//   * it then walks `steps` consecutive 64-row tiles of dY.  Per step: the 64 x K block of dY arrives by LDS-DMA into ONE slot, is read
//     into fragments at once (the slot is free again before the multiply starts), and the NEXT block's DMA plus the NEXT tile's ReLU
//     gates are issued BEFORE this tile's MFMAs, epilogue and stores -- they are in flight under them;
//   * the epilogue is the general kernel's: the fp32 tile parked in LDS, re-read row-contiguous, gated, stored as 8 B per lane.
// Every vector-memory operation of the loop (LDS-DMA, gate loads, output stores) is issued from inline asm, so the counted
// s_waitcnt vmcnt(N) below are exact by construction (vmcnt counts loads, stores and LDS-DMA together, in issue order: a wait for
// the gates of step s is written as "everything issued after them may stay in flight", never as a drain).
// THE GATES NEVER STAY IN REGISTERS ACROSS A CONTROL-FLOW JOIN.  An asm load's destination counts, for hipcc, as written at the asm
// statement: kept in registers from one step to the next (the first form of this kernel), the compiler's copies at the joins --
// v_mov_b64 of a gate register at the loop header, and between the prologue and the single-step / multi-step paths -- read them while
// the loads were still in flight: a few per cent of the masks of a workgroup's FIRST tile came out wrong, on some launches
// (tools/heads_dx_diag.py; found by tests/test_gpu_kernels.py, proven in the ISA).  Now a step's gate registers are local to that
// branch-free step: loaded at its top (for the NEXT tile), and at its end waited for and written to thread-private LDS slots, from
// which the next step's epilogue reads them; the prologue does the same for tile 0.
// MEASURED (round 4, profiles/r04_heads_dx_stream.txt).  The pair alone, interleaved graph replays on shared buffers, two runs: 20.4 -> 17.4 and
// 20.3 -> 17.6 us at cfg2 (3.3 -> 3.8-3.9 TB/s of mask + output; a plain copy of the same bytes: 6.3 TB/s), 80.3 -> 76.8 and 78.8 -> 75.7 us at
// cfg3 (copy: 7.0 TB/s).  In the STEP (tools/knob_step.py <cfg> 13 0 1 1 0: one engine, graphs on the same buffers, same-value spread 0.15 %):
// cfg2 0.2758 vs 0.2757 ms -- nothing; cfg3 0.9383 -> 0.9327 ms (-0.6 %); cfg4 does not take this kernel.  It does NOT reach the memory
// system's rate (VERDICT r3 asked >= 5 TB/s): a step of the walk takes ~4 us, one HBM round trip -- the next tile's gates are requested
// ONE step ahead and the step ends waiting for them.  THAT READING WAS WRONG, and the two forms it called for were built and lose
// (DESIGN_LOG.md R4.2, profiles/r04_heads_dx_phases.txt): per-workgroup phase sums (measurement build 10, tools/heads_dx_phases.py)
// show a step waiting 0.08 us for its operands; with the gates three tiles ahead by LDS-DMA the rate did not move, and with the park
// removed as well (columns of W permuted so that a lane stores 16 B straight from its accumulators, 16 rows x 64 B per wave
// instruction; half the VALU work, one barrier per step instead of three) it FELL: 17.8 vs 17.2 us at cfg2, 79.7 vs 71.3 at cfg3 -- the
// vector-memory path of a CU is what is busy, it serves requests in order, and it prefers this form's 4 rows x 256 B per instruction.
// Arithmetic: the same v_mfma_f32_16x16x32_bf16 chain per output element, k ascending, as gemm_bf16_body -- bit-identical results
// (tests/test_gpu_kernels.py::test_heads_dx_stream_equals_the_grouped_kernel, and every step test runs through it).
#include <algorithm>
#include <string>
#include <type_traits>

#include "gemm_tile.h"
#include "measure.h"

namespace dmvae {
MEAS_TABLES_HDX
void* heads_dx_phase_table() { return MEAS_SYMBOL_HDX(g_hdx); }      // (nullptr in the product build, measure.h)

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

struct HeadsDxArgs {
    int nprob;
    int start[3];             // workgroups of problem i: [start[i], start[i + 1]) behind the `lead` riding ones
    int kind[2];              // 0: K = 64 (CW 128), 1: K = 128 (CW 128), 2: K = 256 (CW 64)
    int chunks[2], steps[2];  // row chunks per problem, 64-row tiles per chunk
    GemmArgs p[2];
    dmvae_finalize_args fin;  // fin.nblocks workgroups of step_finalize ride in the first `lead` ids (a multiple of 8), 0 = none
    int lead;
};

__device__ __forceinline__ void gld2(u32x2& d, const void* p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p) : "memory"); }
__device__ __forceinline__ void gst2(void* p, u32x2 d) { asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(d) : "memory"); }
// wait until at most N vector-memory operations are outstanding; the operands pin the gate registers: nothing the compiler
// schedules may read them before this statement (an asm load's destination counts as written at its own statement)
template <int N, int NQ>
__device__ __forceinline__ void wait_gates(u32x2 (&g)[NQ]) {
    if constexpr (NQ == 8)
        asm volatile("s_waitcnt vmcnt(%8)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]) : "n"(N) : "memory");
    else
        asm volatile("s_waitcnt vmcnt(%4)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]) : "n"(N) : "memory");
}

// One workgroup (4 waves = 2 (M) x 2 (N)): column slice `slice` (CW = 32 CB columns) of problem a, row tiles [t0, t0 + steps) of 64 rows.
// KS = K / 32 (k-steps), CB = 16-column blocks per wave.  smem: X = 32 KiB (W staging, then the fp32 park) | A slot = 64 x K bf16 (32 KiB) |
// gate slots = 256 threads x NQ quads x 8 B (16 KiB), thread-private.
// (A variant whose waits never credit a store -- in case stores could be acknowledged ahead of older loads -- was built to test that
//  suspicion of the first form's fault: it failed the same way; the cause was the register hazard above.)
template <int KS, int CB>
__device__ __forceinline__ void heads_dx_body(const GemmArgs& a, const int slice, const int t0, const int steps, bf16_t* smem) {
    constexpr int K = 32 * KS, KT = K / 64, CW = 32 * CB;
    constexpr int NQ = 64 * (CW / 4) / 256;              // output quads per thread per tile
    constexpr int NAG = KT * 2;                          // LDS-DMA instructions per lane for one 64 x K block of dY
    static_assert(KT >= 1 && (NQ == 8 || NQ == 4) && NQ + NAG <= 63, "tile shape / vmcnt range");
    static_assert(CW * K * 2 <= 32768 && 64 * CW * 4 <= 32768 && 64 * K * 2 <= 32768, "W slice / fp32 park share the 32 KiB region X; the dY slot is 32 KiB");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, li = lane & 15, g = lane >> 4;
    const int n0 = slice * CW;
    bf16_t* Xs = smem;                                   // [KT][CW][64] W images, later float[64][CW]
    bf16_t* As = smem + 16384;                           // [KT][64][64] dY images (byte offset 32768)
    u32x2* gslot = reinterpret_cast<u32x2*>(smem + 32768) + tid;     // this thread's gate quads: quad q at gslot[q * 256] (byte offset 65536)
    const unsigned lds_x = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) bf16_t*)Xs) + 1024u * (unsigned)wave);
    const unsigned lds_a = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) bf16_t*)As) + 1024u * (unsigned)wave);

    const bf16_t* Ag = reinterpret_cast<const bf16_t*>(a.A) + (int64_t)t0 * 64 * a.lda;
    const bf16_t* Wg = reinterpret_cast<const bf16_t*>(a.B) + (int64_t)n0 * a.ldb;
    unsigned goA[2], goW[CW * 64 / 2048];
    stage_offsets<64, true, 4, 64>(a.lda, wave, lane, goA);
    stage_offsets<CW, true, 4, 64>(a.ldb, wave, lane, goW);

    auto issue_a = [&](int s) {      // the 64 x K block of dY of step s: KT images of [64][64]
        const bf16_t* src = Ag + (int64_t)s * 64 * a.lda;
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) glds_tile(src + kt * 64, goA, lds_a + 2u * (unsigned)(kt * 64 * 64), 4096u);
    };
    // per-lane addresses of this thread's quads: row ml = idx / (CW/4), column quad c = idx % (CW/4), idx = q * 256 + tid
    const int ml0 = tid / (CW / 4), cq = tid % (CW / 4);
    constexpr int RQ = 256 / (CW / 4);                   // rows between a thread's consecutive quads
    const bf16_t* gate_p = reinterpret_cast<const bf16_t*>(a.epi.aux0) + (int64_t)(t0 * 64 + ml0) * a.epi.ld0 + n0 + cq * 4;
    bf16_t* out_p = reinterpret_cast<bf16_t*>(a.epi.out) + (int64_t)(t0 * 64 + ml0) * a.epi.ldo + n0 + cq * 4;
    auto issue_gates = [&](u32x2 (&gt)[NQ], int s) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) gld2(gt[q], gate_p + ((int64_t)s * 64 + q * RQ) * a.epi.ld0);
    };
    auto park_gates = [&](const u32x2 (&gt)[NQ]) {       // ONLY behind the wait that covers them: registers -> this thread's LDS slots
#pragma unroll
        for (int q = 0; q < NQ; ++q) gslot[q * 256] = gt[q];
    };

    // ---- prologue (branch-free).  Issue order:  W slice | gates of tile 0 | dY block 0
    MEAS_HDX_BEGIN();
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) glds_tile(Wg + kt * 64, goW, lds_x + 2u * (unsigned)(kt * CW * 64), 4096u);
    bf16x8 wf[CB][KS];                                   // this wave's W fragments: resident for the whole walk
    {
        u32x2 g0[NQ];
        issue_gates(g0, 0);
        issue_a(0);
        wait_vmcnt<NQ + NAG>();                          // the W slice has landed (this wave's share)
        __builtin_amdgcn_s_barrier();
        MEAS_HDX_MARK(1);
#pragma unroll
        for (int j = 0; j < CB; ++j)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) {
                unsigned short lo, hi;
                frag_offsets<CW, true>(wn * (CW / 2) + j * 16, kk & 1, lane, lo, hi);
                wf[j][kk] = read_frag<true>(Xs + (kk >> 1) * CW * 64, lo, hi);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        wait_gates<NAG, NQ>(g0);                         // tile 0's gates have landed (younger: dY block 0 only) ...
        park_gates(g0);                                  // ... and leave the registers before any control flow
    }
    unsigned short foA[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) { unsigned short hi; frag_offsets<64, true>(wm * 32 + i * 16, ks, lane, foA[i][ks], hi); }
    __builtin_amdgcn_s_barrier();                        // every wave has its W fragments: X is free for the park

    float* ct = reinterpret_cast<float*>(Xs);
    MEAS_HDX_MARK(2);
    // One step, branch-free.  Issue order per wave:  ... || step s: dY block s + 1, gates(s + 1), stores(s) || step s + 1: ...
    auto step = [&](auto firstc, auto lastc, int s) {
        constexpr bool FIRST = decltype(firstc)::value, LAST = decltype(lastc)::value;
        // dY block s has landed (this wave's share): the first step waits for it here (nothing younger is outstanding); later steps
        // retired it with the wait for gates(s) at the end of step s - 1 (the block is older than those gates)
        if constexpr (FIRST) wait_vmcnt<0>();
        MEAS_HDX_ACC(5);
        __builtin_amdgcn_s_barrier();
        MEAS_HDX_ACC(6);
        bf16x8 af[2][KS];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) af[i][kk] = read_frag<true>(As + (kk >> 1) * 64 * 64, foA[i][kk & 1], 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();                    // every wave holds block s in registers: the slot is free
        u32x2 gn[NQ];                                    // the NEXT tile's gates: local to this step
        if constexpr (!LAST) {
            issue_a(s + 1);
            issue_gates(gn, s + 1);
        }
        f32x4 acc[2][CB];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < CB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)                  // k ascending per accumulator: the order of gemm_bf16_body
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < CB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j][kk], af[i][kk], acc[i][j], 0, 0, 0);     // operands swapped: a lane owns 4 consecutive n of one m
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < CB; ++j) {
                const int ml = wm * 32 + i * 16 + li;
                const int c = (wn * (CW / 2) + j * 16) / 4 + g;
                *reinterpret_cast<f32x4*>(ct + ml * CW + ((c ^ (ml & 7)) << 2)) = acc[i][j];
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // the fp32 tile is parked
        MEAS_HDX_ACC(7);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int ml = ml0 + q * RQ;
            const f32x4 t = *reinterpret_cast<const f32x4*>(ct + ml * CW + ((cq ^ (ml & 7)) << 2));
            const u32x2 y = gslot[q * 256];              // this tile's gates: parked by this thread at the end of the previous step / the prologue
            const float v0 = __uint_as_float(y[0] << 16) > 0.f ? t[0] : 0.f;
            const float v1 = __uint_as_float(y[0] & 0xffff0000u) > 0.f ? t[1] : 0.f;
            const float v2 = __uint_as_float(y[1] << 16) > 0.f ? t[2] : 0.f;
            const float v3 = __uint_as_float(y[1] & 0xffff0000u) > 0.f ? t[3] : 0.f;
            u32x2 o;
            o[0] = pack2bf(v0, v1);
            o[1] = pack2bf(v2, v3);
            gst2(out_p + ((int64_t)s * 64 + q * RQ) * a.epi.ldo, o);
        }
        MEAS_HDX_ACC(8);
        if constexpr (!LAST) {
            wait_gates<NQ, NQ>(gn);                      // gates(s + 1) -- and the older dY block s + 1 -- have landed; younger: this tile's stores only
            park_gates(gn);                              // (this thread read its slots above: same thread, program order)
        }
        MEAS_HDX_ACC(9);
        // (no barrier here: a wave writes the next park only behind the two barriers at the top of the next step, which it
        //  passes after every wave has finished its reads of this one)
    };
    using T = std::true_type; using F = std::false_type;
    if (steps == 1) { step(T{}, T{}, 0); MEAS_HDX_END(steps); return; }
    step(T{}, F{}, 0);
    int s = 1;
    for (; s + 1 < steps; ++s) step(F{}, F{}, s);
    step(F{}, T{}, s);
    MEAS_HDX_END(steps);
}


__global__ __launch_bounds__(256, 2) void heads_dx_stream_kernel(HeadsDxArgs h) {
    // 80 KiB: X (32 KiB) | dY slot (up to 64 x 256 bf16 = 32 KiB) | gate slots (16 KiB) -> two workgroups per CU (2 x 81920 B = the CU's 160 KiB)
    __shared__ __attribute__((aligned(16))) bf16_t smem[40960];
    int bx = (int)blockIdx.x;
    if (bx < h.lead) {            // the riding step_finalize blocks hold the FIRST ids of the grid (they run under the tiles, gemm_bf16.hip)
        if (bx < h.fin.nblocks) step_finalize_block(bx, h.fin, reinterpret_cast<float(*)[17]>(smem));
        return;
    }
    bx -= h.lead;
    const int i = (h.nprob > 1 && bx >= h.start[1]) ? 1 : 0;
    const int w = bx - h.start[i];
    const GemmArgs& a = h.p[i];
    const int kind = h.kind[i];
    const int nsl = a.N / (kind == 2 ? 64 : 128);
    const int chunk = w / nsl, slice = w - chunk * nsl;            // consecutive ids: neighbouring column slices of the same rows
    const int tiles = a.M / 64;
    const int t0 = chunk * h.steps[i];
    const int steps = min(h.steps[i], tiles - t0);
    if (steps <= 0) return;
    if (kind == 0) heads_dx_body<2, 4>(a, slice, t0, steps, smem);
    else if (kind == 1) heads_dx_body<4, 4>(a, slice, t0, steps, smem);
    else heads_dx_body<8, 2>(a, slice, t0, steps, smem);
}

// ---------------------------------------------------------------- host side
static int g_heads_dx_stream = 1;      // tuning knob (dmvae_debug_set_knob 13): 1 = the qualifying problems of a grouped DX / RELU_MASK launch take this kernel, 0 = the grouped tiles
void heads_dx_stream_set(int v) { g_heads_dx_stream = v; }

static int heads_dx_kind(const GemmArgs& p) {
    if (p.conv_c || p.M % 64 || p.N % 128 || p.k_split != p.K || p.epi.kind != DMVAE_EPI_RELU_MASK || !p.epi.aux0 || !p.epi.out) return -1;
    if ((p.lda * 2) % 16 || (p.ldb * 2) % 16 || p.epi.ld0 % 8 || p.epi.ldo % 8) return -1;
    if (((uintptr_t)p.epi.aux0 | (uintptr_t)p.epi.out | (uintptr_t)p.A | (uintptr_t)p.B) % 16) return -1;      // 16-byte units: LDS-DMA of the gates, dwordx4 stores
    return p.K == 64 ? 0 : p.K == 128 ? 1 : p.K == 256 ? 2 : -1;
}

// a grouped DX / RELU_MASK launch of one or two problems that ALL qualify (K = 64, 128 or 256) as ONE streaming launch, with the riding
// step_finalize blocks; taken[i] says which problems it took (all or none) -- the caller launches what is left, riders included, as before
int heads_dx_stream_launch(hipStream_t s, const GemmArgs* probs, int nprob, const dmvae_finalize_args* fin, bool* taken) {
    for (int i = 0; i < nprob; ++i) taken[i] = false;
    if (!g_heads_dx_stream) return 0;
    HeadsDxArgs h{};
    int n = 0;
    double flops = 0.0, bytes = 0.0;
    // ALL problems of the group or none.  MEASURED (round 4, tools/heads_dx_ab.py, the pair alone, interleaved on shared buffers): with
    // K = 2 D = 512 (cfg4) only the class-logits problem qualifies; streaming it and leaving the other on the grouped tiles makes two launches
    // out of one grid: 48.9 vs 44.1 us for the pair.
    if (nprob > 2) return 0;
    for (int i = 0; i < nprob; ++i)
        if (heads_dx_kind(probs[i]) < 0) return 0;
    for (int i = 0; i < nprob; ++i) {
        taken[i] = true;
        h.p[n] = probs[i];
        h.kind[n] = heads_dx_kind(probs[i]);
        ++n;
    }
    if (n == 0) return 0;
    // Row chunks per problem.  ONE round of the chip (two resident workgroups per CU = 512 slots) where the batch allows -- never more than 512
    // workgroups: rounding the chunk count to nearest gave 528 at 16 384 rows with D = 128, and the 16 workgroups of the second round started when
    // the first ones ended (launch span 72 us for walks of 45-56 us) -- and at least two steps each.  With two problems the slots are shared so
    // that the walks END together: a step costs 2.1-2.2 us on a 128-column slice, 1.9 on a 64-column one (measurement build 10,
    // profiles/r04_heads_dx_phases.txt: with equal step counts the 128-column workgroups of cfg3 ran 60 us, the others 52).
    auto step_cost = [](int kind) { return kind == 2 ? 1.93 : kind == 1 ? 2.10 : 2.20; };
    auto shape = [&](int i, int want, int& chunks, int& steps) {
        const int tiles = h.p[i].M / 64;
        chunks = std::max(1, std::min(tiles, want));
        steps = (tiles + chunks - 1) / chunks;
        if (steps < 2 && tiles >= 2) steps = 2;
        chunks = (tiles + steps - 1) / steps;
    };
    int nsl[2] = {0, 0};
    for (int i = 0; i < n; ++i) nsl[i] = h.p[i].N / (h.kind[i] == 2 ? 64 : 128);
    if (n == 1) {
        shape(0, 512 / nsl[0], h.chunks[0], h.steps[0]);
    } else {
        double best = 1e30;
        for (int c0 = 1; c0 * nsl[0] < 512; ++c0) {
            const int c1 = (512 - c0 * nsl[0]) / nsl[1];
            if (c1 < 1) break;
            int ch0, st0, ch1, st1;
            shape(0, c0, ch0, st0);
            shape(1, c1, ch1, st1);
            if (ch0 * nsl[0] + ch1 * nsl[1] > 512) continue;
            const double t = std::max(st0 * step_cost(h.kind[0]), st1 * step_cost(h.kind[1]));
            if (t < best) { best = t; h.chunks[0] = ch0; h.steps[0] = st0; h.chunks[1] = ch1; h.steps[1] = st1; }
        }
        if (best == 1e30) {                // (cannot happen for the shapes heads_dx_kind admits; the caller then launches the grouped tiles)
            for (int i = 0; i < nprob; ++i) taken[i] = false;
            return 0;
        }
    }
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const GemmArgs& p = h.p[i];
        h.start[i] = total;
        total += h.chunks[i] * nsl[i];
        flops += 2.0 * p.M * p.N * (double)p.K;
        bytes += 2.0 * ((double)p.M * p.K + (double)p.K * p.N) + 2.0 * p.M * p.N;
    }
    for (int i = n; i <= 2; ++i) h.start[i] = total;
    h.nprob = n;
    h.lead = 0;
    if (fin) { h.fin = *fin; h.lead = (fin->nblocks + 7) & ~7; }
    static const std::string nm = "heads_dx_stream_kernel";
    ProfScope ps(s, nm.c_str(), flops, bytes);
    DMVAE_LAUNCH(heads_dx_stream_kernel, dim3(total + h.lead), dim3(256), 0, s, h);
    return check_launch("heads_dx_stream");
}

}  // namespace dmvae
