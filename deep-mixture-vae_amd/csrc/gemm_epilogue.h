// Fused GEMM epilogues shared by the bf16 and f32 MFMA kernels.
//
// Both kernels issue their MFMAs with the operands swapped (first = B
// fragment, second = A fragment), so with the gfx950 C/D map
//   col = lane & 15, row = 4*(lane >> 4) + reg
// every lane ends up holding FOUR CONSECUTIVE n of ONE row m of C: the
// epilogue works on (m, n..n+3) quads -> 8-byte bf16 / 16-byte f32 stores and
// vector loads of bias / mask / target.
#pragma once
#include "common.h"

namespace dmvae {

struct GemmArgs {
    const void* A; int64_t lda;
    const void* B; int64_t ldb;
    int M, N, K, k_split;   // k_split: contraction depth handled by one blockIdx.y
    int group_m;            // supertile height (tile rows) of the L2-friendly tile order
    // Implicit 3x3 SAME convolution (csrc/conv.hip): conv_c != 0 makes the k-contiguous A operand a
    // patch matrix that is never formed.  A's rows are the pixels of zero-bordered images in HBM
    // ([B][P][P][lda], P = side + 2) and K runs over (tap, channel) with conv_c <= lda channels per tap
    // (32, or a multiple of 64), padded to a multiple of 64 with zero-weight columns: the columns of tap t
    // are the SAME rows shifted by (t/3 - 1) * P + (t%3 - 1) pixels -- a wave-uniform pointer offset per K
    // tile (conv_c = 32: two taps per tile, the second a wave-uniform delta on half of the lanes); the
    // per-lane offsets of the LDS-DMA stay loop-invariant.  In the DW layout (weight gradient, M = the
    // (tap, channel) rows) the shift belongs to the tile's columns instead.
    int conv_p, conv_c;
    dmvae_epilogue epi;
    // scratch for the bias-gradient column sums of a 256x256-tile dW problem (gemm_bf16_256.hip): >= 64 * N floats,
    // the plan's own; nullptr = the library's lazily allocated global scratch (not capture-safe on first use)
    float* ws = nullptr; int64_t ws_elems = 0;
    // f32 kernel, DMVAE_EPI_STORE_F32 with split-K: K slice y stores its partial product at out + y * slab_stride (deterministic
    // split: the caller adds the slabs in a fixed order); 0 = plain store
    int64_t slab_stride = 0;
    int64_t slab_stride2 = 0;      // likewise for the DW layout's fused bias gradient (epi.out2)
    // 256x256 kernel: column sums of the tile's OUTPUT as it is stored (RELU_MASK / BIAS_RECON epilogues: the dY of the next
    // weight-gradient GEMM) -> csum_out[tile_row][csum_ld]: the bias gradient of that layer without re-reading dY.
    float* csum_out = nullptr; int64_t csum_ld = 0;
    // DW layout, 256x256 kernel: such partials of this problem's dY, [csum_rows][csum_ld] (replaces the slab column sums)
    const float* csum_in = nullptr; int csum_rows = 0;
    // DW layout, grouped small-tile launch: a BIAS-ONLY strip -- M = 64 pro forma, A is never read; only epi.out2 (the column sums of B)
    // is produced.  The layer's weight gradient itself runs on the macro tile (K slices into slabs, csrc/api.hip launch_dw_queue).
    int bias_only = 0;
    // K slices of a FUSED-epilogue problem (small-tile kernel, FWD / DX layouts; round 5): grid.y = K / k_split workgroups per tile, each storing its
    // f32 partial tile to slab [tile][slice] of `ws` (>= tiles * slices * BM * BN floats); the LAST one to arrive (a ticket per tile in `tick`, which
    // starts at zero and is reset by that workgroup) adds the slabs in ascending slice order and runs the epilogue: deterministic, no float atomics.
    // For launches that occupy a fraction of the chip with a long K chain (a 100-row batch: every GEMM < 32 CUs).
    int* tick = nullptr;
};

// element offset of K position k (multiple of the tile depth) of a conv-mode A operand
__host__ __device__ __forceinline__ int64_t conv_tap_offset(int tap, int c0, int P, int64_t lda) {
    return (int64_t)((tap / 3 - 1) * P + (tap % 3 - 1)) * lda + c0;
}

template <typename T> struct ActIO;
template <> struct ActIO<bf16_t> {
    static __device__ __forceinline__ void store4(void* base, int64_t off, const float v[4]) {
        uint2 p;
        p.x = pack2bf(v[0], v[1]);
        p.y = pack2bf(v[2], v[3]);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(base) + off) = p;
    }
    static __device__ __forceinline__ void load4(const void* base, int64_t off, float v[4]) {
        const uint2 p = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16_t*>(base) + off);
        v[0] = __uint_as_float(p.x << 16);
        v[1] = __uint_as_float(p.x & 0xffff0000u);
        v[2] = __uint_as_float(p.y << 16);
        v[3] = __uint_as_float(p.y & 0xffff0000u);
    }
};
template <> struct ActIO<float> {
    static __device__ __forceinline__ void store4(void* base, int64_t off, const float v[4]) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + off) = make_float4(v[0], v[1], v[2], v[3]);
    }
    static __device__ __forceinline__ void load4(const void* base, int64_t off, float v[4]) {
        const float4 p = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + off);
        v[0] = p.x; v[1] = p.y; v[2] = p.z; v[3] = p.w;
    }
};

__device__ __forceinline__ void loadf4(const void* base, int64_t off, float v[4]) {
    ActIO<float>::load4(base, off, v);
}

// sigmoid cross entropy with logits, TF form: max(l,0) - l*x + log(1+exp(-|l|))
__device__ __forceinline__ float xent_logits(float l, float x) {
    return fmaxf(l, 0.f) - l * x + log1pf(__expf(-fabsf(l)));
}
__device__ __forceinline__ float sigmoidf_(float l) { return 1.0f / (1.0f + __expf(-l)); }
// Batch assembly (Dataset.get_batches, includes/utils.py:449-463: row r <- data[perm[first + r]]) as a device function: the gather kernel
// (elementwise.hip) runs it over its own grid, and `nblocks` workgroups of ANOTHER launch can run it as riders (the next batch fetched under
// the trunk's dX GEMM, gemm_bf16.hip / api.hip dmvae_plan_prefetch_batch).  16-byte loads of the f32 source row; writes the act copy (bf16:
// 8-byte stores) and / or the f32 copy; pad rows and pad columns are zeros.  st != nullptr: first = st->batch_cursor * batch.
struct dmvae_gather_args {
    const float* data; int64_t n_rows; int dim; const int32_t* perm;
    int64_t first; int batch, n_valid, B_pad;
    void* out_act; int64_t ld_act; float* out_f32; int64_t ld_f32; int cols_pad;
    const dmvae_state* st;
    int nblocks;                      // workgroups that share the work (grid-stride)
};
// A thread's chain is permutation entry -> source row -> store: two dependent memory round trips per quad.  U quads per pass, each stage issued
// for all of them before the next stage starts, and every load UNCONDITIONAL (indices clamped, the result selected afterwards): a load inside a
// divergent `if` is waited for at the join, so the "batched" first form ran its 16 chains one after the other -- as riders (few blocks: a rider
// block holds its host kernel's LDS) the gather then took 16-20 us inside an 11-us launch (rocprofv3, profiles/r04_prefetch.txt).
// Fast path: input_dim a multiple of 4 and a 16-byte aligned dataset (every quad inside a row is then one aligned 16-byte load).
template <typename ACT, int U, bool PERM>
__device__ __forceinline__ void gather_rows_pass(const int base, const int nthreads, const int total, const int quads, const int64_t first, const dmvae_gather_args& g) {
    int r[U], c[U];
    int64_t idx[U];
    bool inb[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int i = base + u * nthreads;
        r[u] = i < total ? i / quads : -1;
        c[u] = i < total ? (i - r[u] * quads) * 4 : 0;
        const int64_t sr = first + r[u];
        inb[u] = r[u] >= 0 && r[u] < g.n_valid && sr >= 0 && sr < g.n_rows;      // (the permutation has n_rows entries: a batch that runs past them reads zeros)
        idx[u] = inb[u] ? sr : 0;
    }
    if constexpr (PERM) {
#pragma unroll
        for (int u = 0; u < U; ++u) idx[u] = (int64_t)g.perm[idx[u]];
    }
    float4 q[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        inb[u] = inb[u] && idx[u] >= 0 && idx[u] < g.n_rows && c[u] < g.dim;
        q[u] = *reinterpret_cast<const float4*>(g.data + (inb[u] ? idx[u] * g.dim + c[u] : 0));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (r[u] < 0) continue;
        const float v[4] = {inb[u] ? q[u].x : 0.f, inb[u] ? q[u].y : 0.f, inb[u] ? q[u].z : 0.f, inb[u] ? q[u].w : 0.f};
        if (g.out_act) ActIO<ACT>::store4(g.out_act, (int64_t)r[u] * g.ld_act + c[u], v);
        if (g.out_f32) ActIO<float>::store4(g.out_f32, (int64_t)r[u] * g.ld_f32 + c[u], v);
    }
}
template <typename ACT, int U = 8>        // (U = 16 for riders: gemm_bf16_dx_riders_kernel)
__device__ __forceinline__ void gather_rows_block(const int bx, const int nthreads, const dmvae_gather_args& g) {
    int64_t first = g.first;
    if (g.st) first = (int64_t)g.st->batch_cursor * g.batch;
    const int quads = g.cols_pad >> 2;
    const int total = g.B_pad * quads;                  // (< 2^31: gather_launch / the plan's sizes)
    if (g.dim % 4 == 0 && (reinterpret_cast<uintptr_t>(g.data) & 15) == 0 && g.n_rows > 0) {
        for (int base = bx * nthreads * U + (int)threadIdx.x; base < total; base += g.nblocks * nthreads * U) {
            if (g.perm) gather_rows_pass<ACT, U, true>(base, nthreads, total, quads, first, g);
            else gather_rows_pass<ACT, U, false>(base, nthreads, total, quads, first, g);
        }
        return;
    }
    for (int i = bx * nthreads + (int)threadIdx.x; i < total; i += g.nblocks * nthreads) {      // any input_dim / alignment: one quad at a time, element loads
        const int r = i / quads, c = (i - r * quads) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (r < g.n_valid) {
            int64_t src = first + r;
            if (g.perm) src = (src >= 0 && src < g.n_rows) ? (int64_t)g.perm[src] : -1;
            if (src >= 0 && src < g.n_rows) {
                const float* p = g.data + src * g.dim + c;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = (c + j < g.dim) ? p[j] : 0.f;
            }
        }
        if (g.out_act) ActIO<ACT>::store4(g.out_act, (int64_t)r * g.ld_act + c, v);
        if (g.out_f32) ActIO<float>::store4(g.out_f32, (int64_t)r * g.ld_f32 + c, v);
    }
}

// Workgroups of other work riding in the FIRST ids of a dX GEMM launch (gemm_bf16_dx_riders_kernel): nfin ids for the step_finalize blocks
// (fin.nblocks of them work), then ngat ids for the gather of the next batch (all work); nfin, ngat multiples of 8, so that id % 8 -- the XCD a
// workgroup lands on -- stays what the tile mapping assumes.
struct GemmRiders {
    dmvae_finalize_args fin; int nfin = 0;
    dmvae_gather_args gat; int ngat = 0;
    int gat_last = 0;                 // 1: the gather blocks take the LAST ids of the grid instead (behind the tiles; not counted in the tile mapping's lead)
};

// cross entropy AND sigmoid from ONE exponential: e = exp(-|l|);
//   xent = max(l,0) - l*x + log(1+e),  sigmoid = l >= 0 ? 1/(1+e) : e/(1+e)
// (3 transcendental issues per element instead of log1pf + 2 expf; log(1+e) loses e below 6e-8,
// an absolute error <= 6e-8 per element, far inside the 1e-3 loss tolerance over 784 elements)
__device__ __forceinline__ void xent_sigmoid(float l, float x, float& xent, float& sig) {
    const float e = __expf(-fabsf(l));
    const float inv = __frcp_rn(1.0f + e);
    xent = fmaxf(l, 0.f) - l * x + __logf(1.0f + e);
    sig = l >= 0.f ? inv : e * inv;
}

// One quad of the epilogue.  `loss` accumulates the RECON contribution.
// xpre: RECON only, the target quad x[m][n..n+3] when the caller fetched it ahead of time.
// stored: optional, receives the four values as they were stored in the ACT output (RELU_MASK, BIAS_RECON: the gradient quad)
template <int EPI, typename ACT>
// bpre / ypre: the bias quad / the forward-activation (ReLU gate) quad already in registers.  Loaded inside, every call sits
// behind the previous call's stores (the compiler cannot tell the arrays apart): one memory round trip per quad.
__device__ __forceinline__ void epilogue_quad(const dmvae_epilogue& e, int m, int n, float v[4], float& loss, const float* xpre = nullptr, float* stored = nullptr,
                                              const float* bpre = nullptr, const float* ypre = nullptr) {
    if constexpr (EPI == DMVAE_EPI_BIAS_RELU) {
        float b[4];
        if (bpre) { b[0] = bpre[0]; b[1] = bpre[1]; b[2] = bpre[2]; b[3] = bpre[3]; }
        else loadf4(e.bias, n, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j] + b[j], 0.f);
        ActIO<ACT>::store4(e.out, (int64_t)m * e.ldo + n, v);
    } else if constexpr (EPI == DMVAE_EPI_BIAS_F32) {
        float b[4];
        if (bpre) { b[0] = bpre[0]; b[1] = bpre[1]; b[2] = bpre[2]; b[3] = bpre[3]; }
        else loadf4(e.bias, n, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += b[j];
        ActIO<float>::store4(e.out, (int64_t)m * e.ldo + n, v);
    } else if constexpr (EPI == DMVAE_EPI_BIAS_SIGMOID) {
        float b[4];
        if (bpre) { b[0] = bpre[0]; b[1] = bpre[1]; b[2] = bpre[2]; b[3] = bpre[3]; }
        else loadf4(e.bias, n, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = sigmoidf_(v[j] + b[j]);
        ActIO<float>::store4(e.out, (int64_t)m * e.ldo + n, v);
    } else if constexpr (EPI == DMVAE_EPI_BIAS_RECON) {
        float b[4], x[4], d[4];
        if (bpre) { b[0] = bpre[0]; b[1] = bpre[1]; b[2] = bpre[2]; b[3] = bpre[3]; }
        else loadf4(e.bias, n, b);
        if (xpre) { x[0] = xpre[0]; x[1] = xpre[1]; x[2] = xpre[2]; x[3] = xpre[3]; }
        else loadf4(e.aux0, (int64_t)m * e.ld0 + n, x);
        const bool rowok = m < e.m_valid;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float l = v[j] + b[j];
            v[j] = l;
            const bool ok = rowok && (n + j) < e.n_valid;
            if ((e.recon_kind & 0xff) == 0) {
                float xe, sg;
                xent_sigmoid(l, x[j], xe, sg);
                loss += ok ? xe : 0.f;
                d[j] = ok ? (sg - x[j]) * e.scale : 0.f;
            } else {
                const float r = l - x[j];
                loss += ok ? 0.5f * r * r : 0.f;
                d[j] = ok ? r * e.scale : 0.f;
            }
        }
        ActIO<ACT>::store4(e.out, (int64_t)m * e.ldo + n, d);
        if (stored) { stored[0] = d[0]; stored[1] = d[1]; stored[2] = d[2]; stored[3] = d[3]; }
        if (e.out2) ActIO<float>::store4(e.out2, (int64_t)m * e.ldo2 + n, v);
    } else if constexpr (EPI == DMVAE_EPI_RELU_MASK) {
        float y[4];
        if (ypre) { y[0] = ypre[0]; y[1] = ypre[1]; y[2] = ypre[2]; y[3] = ypre[3]; }
        else ActIO<ACT>::load4(e.aux0, (int64_t)m * e.ld0 + n, y);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = y[j] > 0.f ? v[j] : 0.f;
        ActIO<ACT>::store4(e.out, (int64_t)m * e.ldo + n, v);
        if (stored) { stored[0] = v[0]; stored[1] = v[1]; stored[2] = v[2]; stored[3] = v[3]; }
    } else if constexpr (EPI == DMVAE_EPI_LATENT) {
        float gm[4], gl[4], cl[4], o0[4], o1[4];
        if (xpre) {       // xpre[12]: the three quads already in registers
#pragma unroll
            for (int j = 0; j < 4; ++j) { gm[j] = xpre[j]; gl[j] = xpre[4 + j]; cl[j] = xpre[8 + j]; }
        } else {
            loadf4(e.aux0, (int64_t)m * e.ld0 + n, gm);
            loadf4(e.aux1, (int64_t)m * e.ld1 + n, gl);
            loadf4(e.aux2, (int64_t)m * e.ld2 + n, cl);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o0[j] = v[j] + gm[j];
            o1[j] = v[j] * cl[j] + gl[j];
        }
        ActIO<ACT>::store4(e.out, (int64_t)m * e.ldo + n, o0);
        ActIO<ACT>::store4(e.out, (int64_t)m * e.ldo + e.d_off + n, o1);
    } else if constexpr (EPI == DMVAE_EPI_STORE_F32) {
        ActIO<float>::store4(e.out, (int64_t)m * e.ldo + n, v);
    } else if constexpr (EPI == DMVAE_EPI_ATOMIC_F32) {
        float* o = reinterpret_cast<float*>(e.out) + (int64_t)m * e.ldo + n;
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(o + j, v[j]);
    }
}

}  // namespace dmvae
