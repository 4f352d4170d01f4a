// HBM-bound kernels of the DMVAE step: TF-style Adam, column sums (bias
// gradients / reduction of per-block partials), stand-alone reconstruction
// loss, batch assembly, Philox noise, casts, loss finalize.
// All loads/stores are 16 B per lane where the layout allows.
#include <algorithm>

#include "kernels.h"

namespace dmvae {

// ---------------------------------------------------------------- Adam (TF 1.x)
// tf.train.AdamOptimizer, base_models.py:95-110 -- epsilon OUTSIDE the bias
// correction: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps).
// 4 elements per lane: 16-B loads of p, g, m, v; 16-B stores of p, m, v (+8 B bf16).
__global__ __launch_bounds__(256) void adam_tf_kernel(AdamArgs a) {
    // state given: t = adam_t + 1, or adam_t itself when the step's loss_finalize already advanced it (t_host == ~0)
    const uint64_t t = a.st ? (a.t_host == ~0ull ? a.st->adam_t : a.st->adam_t + 1) : a.t_host;
    const float lr = a.st ? a.st->lr : a.lr;
    const float lr_t = adam_lr_t(lr, a.b1, a.b2, t);
    const int64_t n4 = a.n >> 2;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 p = reinterpret_cast<float4*>(a.p)[i];
        float4 g = reinterpret_cast<float4*>(a.g)[i];
        float4 m = reinterpret_cast<float4*>(a.m)[i];
        float4 v = reinterpret_cast<float4*>(a.v)[i];
        float* pp = &p.x; float* gp = &g.x; float* mp = &m.x; float* vp = &v.x;
        if (a.pb && !a.ieee) {           // bf16 mode: the same arithmetic as the fused dW epilogues (adam_elem<true>)
#pragma unroll
            for (int j = 0; j < 4; ++j) adam_elem<true>(pp[j], mp[j], vp[j], gp[j], a.gscale, a.b1, a.b2, a.eps, lr_t);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) adam_elem<false>(pp[j], mp[j], vp[j], gp[j], a.gscale, a.b1, a.b2, a.eps, lr_t);
        }
        reinterpret_cast<float4*>(a.p)[i] = p;
        reinterpret_cast<float4*>(a.m)[i] = m;
        reinterpret_cast<float4*>(a.v)[i] = v;
        if (a.pb) {
            uint2 q;
            q.x = pack2bf(p.x, p.y);
            q.y = pack2bf(p.z, p.w);
            reinterpret_cast<uint2*>(a.pb)[i] = q;
        }
        if (a.zero_grad) reinterpret_cast<float4*>(a.g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
__global__ void adam_finish_kernel(dmvae_state* st) { st->adam_t += 1; }

// The same update for a gradient that arrives as nslab (2..16) K-slice slabs of the weight-gradient GEMMs (split-K without float
// atomics: the dW group of a large batch, csrc/api.hip grad_dense): element i of the gradient = slab 0 + slab 1 + ... in ASCENDING
// order -- the order slab_reduce_kernel uses for <= 16 slabs, including its trailing + 0 (a sum of -0s becomes +0 there), so the
// fused form and slab_reduce -> adam_tf give the same bits.  Elements of [seg_lo, seg_hi) (the prior tables: their gradient comes
// complete from step_finalize) take a.g instead.  30 + 4 nslab bytes per parameter.
__global__ __launch_bounds__(256) void adam_slabs_kernel(AdamArgs a, const float* __restrict__ slabs, int nslab, int64_t stride, int64_t seg_lo, int64_t seg_hi) {
    const uint64_t t = a.st ? (a.t_host == ~0ull ? a.st->adam_t : a.st->adam_t + 1) : a.t_host;
    const float lr_t = adam_lr_t(a.st ? a.st->lr : a.lr, a.b1, a.b2, t);
    const int64_t n4 = a.n >> 2, stride4 = stride >> 2;
    const int64_t step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += step) {
        float4 p = reinterpret_cast<float4*>(a.p)[i];
        float4 m = reinterpret_cast<float4*>(a.m)[i];
        float4 v = reinterpret_cast<float4*>(a.v)[i];
        float4 g;
        if (4 * i >= seg_lo && 4 * i < seg_hi) g = reinterpret_cast<const float4*>(a.g)[i];
        else {
            g = reinterpret_cast<const float4*>(slabs)[i];
            for (int s = 1; s < nslab; ++s) {
                const float4 b = reinterpret_cast<const float4*>(slabs)[(int64_t)s * stride4 + i];
                g.x += b.x; g.y += b.y; g.z += b.z; g.w += b.w;
            }
            if (nslab < 16) { g.x += 0.f; g.y += 0.f; g.z += 0.f; g.w += 0.f; }
        }
        float* pp = &p.x; float* gp = &g.x; float* mp = &m.x; float* vp = &v.x;
        if (a.pb && !a.ieee) {
#pragma unroll
            for (int j = 0; j < 4; ++j) adam_elem<true>(pp[j], mp[j], vp[j], gp[j], a.gscale, a.b1, a.b2, a.eps, lr_t);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) adam_elem<false>(pp[j], mp[j], vp[j], gp[j], a.gscale, a.b1, a.b2, a.eps, lr_t);
        }
        reinterpret_cast<float4*>(a.p)[i] = p;
        reinterpret_cast<float4*>(a.m)[i] = m;
        reinterpret_cast<float4*>(a.v)[i] = v;
        if (a.pb) {
            uint2 q;
            q.x = pack2bf(p.x, p.y);
            q.y = pack2bf(p.z, p.w);
            reinterpret_cast<uint2*>(a.pb)[i] = q;
        }
        if (a.zero_grad) reinterpret_cast<float4*>(a.g)[i] = g;      // here: KEEP the summed gradient (store_grad), never zero it
    }
}

// ---------------------------------------------------------------- Adam in the SHADOW of a macro-tile GEMM
// The same update as adam_tf_kernel (same adam_elem: same bits), shaped to run BESIDE the 256x256 macro-tile GEMM
// (gemm_bf16_256.hip: one workgroup of 8 waves, 128 KiB of LDS, 2 x 232 VGPRs per SIMD lane on every CU) instead of behind it: what that
// kernel leaves free on a CU is 48 VGPRs per SIMD lane, 32 KiB of LDS and wave slots.  So: one workgroup of four waves (one per SIMD)
// per CU, <= 48 VGPRs, and the loads in flight live in LDS, not in registers -- every wave owns a private ring of SHADOW_NST stages x
// {p, m, v, g} x 1 KiB filled by LDS-DMA (global_load_lds_dwordx4, 64 lanes x 16 B per instruction): 32 KiB in flight per CU.  No
// barrier anywhere (a wave reads only what it requested itself; vmcnt retires in order).  lr_t is READ from the step state (state->lr_t as
// step_finalize left it for t = state->adam_t: the t_host == ~0 convention of adam_tf_kernel); the host refuses the other forms.
constexpr int SHADOW_NST = 2;
__device__ __forceinline__ void shadow_glds(const float* base, unsigned voff, unsigned lds) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(base), "v"(voff), "s"(lds)
        : "memory");
}
template <bool FAST>
__global__ __launch_bounds__(256) void adam_shadow_kernel(AdamArgs a) {
    __shared__ __attribute__((aligned(16))) float ring[4 * SHADOW_NST * 4 * 256];      // [wave][stage][array][256 floats]: 32 KiB
    const float lr_t = a.st->lr_t;                          // (the double-precision pow of adam_lr_t alone takes more registers than this kernel may use)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int nchunk = (int)(a.n >> 8);                    // chunks of 256 elements = 64 lanes x one quad (the host hands the remainder to adam_tf_kernel; n < 2^39)
    const int cstep = (int)gridDim.x * 4;
    float* mine = ring + wave * (SHADOW_NST * 4 * 256);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) float*)mine));
    const unsigned voff = (unsigned)lane * 16u;
    auto issue = [&](int c, int slot) {                     // the four 1-KiB rows of chunk c -> stage `slot`
        const int64_t e = (int64_t)c << 8;
        const unsigned l = lds0 + (unsigned)slot * 4096u;
        shadow_glds(a.p + e, voff, l);
        shadow_glds(a.m + e, voff, l + 1024u);
        shadow_glds(a.v + e, voff, l + 2048u);
        shadow_glds(a.g + e, voff, l + 3072u);
    };
    const int c0 = (int)blockIdx.x * 4 + wave;
    const int mycount = c0 < nchunk ? (nchunk - 1 - c0) / cstep + 1 : 0;      // chunks of this wave: c0 + k * cstep
#pragma unroll
    for (int s = 0; s < SHADOW_NST; ++s)
        if (s < mycount) issue(c0 + s * cstep, s);
    int slot = 0;
    for (int k = 0; k < mycount; ++k) {
        const int c = c0 + k * cstep;
        // chunk c has landed when at most the loads of the SHADOW_NST - 1 younger chunks are outstanding; the stores of the previous
        // trips are OLDER than those loads, so the same count also covers them (no assumption about how stores and loads interleave)
        if (k + SHADOW_NST - 1 < mycount) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (SHADOW_NST - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const float* st = mine + slot * 1024 + lane * 4;
        float4 p = *reinterpret_cast<const float4*>(st);
        float4 m = *reinterpret_cast<const float4*>(st + 256);
        float4 v = *reinterpret_cast<const float4*>(st + 512);
        const float4 g = *reinterpret_cast<const float4*>(st + 768);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the stage is in registers: its LDS may be refilled
        if (k + SHADOW_NST < mycount) issue(c + SHADOW_NST * cstep, slot);
        float* pp = &p.x; float* mp = &m.x; float* vp = &v.x; const float* gp = &g.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) adam_elem<FAST>(pp[j], mp[j], vp[j], gp[j], a.gscale, a.b1, a.b2, a.eps, lr_t);
        const int64_t i = ((int64_t)c << 6) + lane;
        reinterpret_cast<float4*>(a.p)[i] = p;
        reinterpret_cast<float4*>(a.m)[i] = m;
        reinterpret_cast<float4*>(a.v)[i] = v;
        if (a.pb) {
            uint2 q;
            q.x = pack2bf(p.x, p.y);
            q.y = pack2bf(p.z, p.w);
            reinterpret_cast<uint2*>(a.pb)[i] = q;
        }
        if (a.zero_grad) reinterpret_cast<float4*>(a.g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        slot = slot + 1 == SHADOW_NST ? 0 : slot + 1;
    }
}
// n elements (multiple of 4); the part that is not a whole number of 256-element chunks goes through adam_tf_kernel
int adam_shadow_launch(hipStream_t s, const AdamArgs& a0, int blocks) {
    if (!a0.st || a0.t_host != ~0ull) { set_error("adam (shadow form): needs the step state with t already advanced (state != NULL, t_host = ~0): it reads state->lr_t"); return DMVAE_EINVAL; }
    AdamArgs a = a0;
    const int64_t body = (a.n >> 8) << 8;
    if (body > 0) {
        a.n = body;
        if (blocks <= 0) blocks = 256;
        blocks = (int)std::min<int64_t>(blocks, (body >> 8) / 4 + 1);
        ProfScope ps(s, "adam_shadow", 12.0 * body, (28.0 + (a.pb ? 2.0 : 0.0) + (a.zero_grad ? 4.0 : 0.0)) * body);
        if (a.pb && !a.ieee) DMVAE_LAUNCH(adam_shadow_kernel<true>, dim3(blocks), dim3(256), 0, s, a);      // (the arithmetic follows the mode: adam_elem)
        else DMVAE_LAUNCH(adam_shadow_kernel<false>, dim3(blocks), dim3(256), 0, s, a);
        const int rc = check_launch("adam_shadow");
        if (rc) return rc;
    }
    if (body < a0.n) {
        AdamArgs r = a0;
        r.n = a0.n - body; r.p += body; r.g += body; r.m += body; r.v += body;
        if (r.pb) r.pb += body;
        return adam_launch(s, r);
    }
    return 0;
}

int adam_launch(hipStream_t s, const AdamArgs& a) {
    const int64_t n4 = a.n >> 2;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    ProfScope ps(s, "adam_tf", 12.0 * a.n, (28.0 + (a.pb ? 2.0 : 0.0) + (a.zero_grad ? 4.0 : 0.0)) * a.n);
    DMVAE_LAUNCH(adam_tf_kernel, dim3(blocks), dim3(256), 0, s, a);
    return check_launch("adam_tf");
}
int adam_slabs_launch(hipStream_t s, const AdamArgs& a, const float* slabs, int nslab, int64_t stride, int64_t seg_lo, int64_t seg_hi) {
    if (a.n % 4 || stride % 4 || seg_lo % 4 || seg_hi % 4 || nslab < 2 || nslab > 16) { set_error("adam_slabs: sizes must be multiples of 4, 2..16 slabs"); return DMVAE_EINVAL; }
    const int64_t n4 = a.n >> 2;
    const int blocks = (int)std::max<int64_t>(1, std::min<int64_t>(2048, (n4 + 255) / 256));
    ProfScope ps(s, "adam_slabs", (12.0 + nslab) * a.n, (24.0 + (a.pb ? 2.0 : 0.0) + 4.0 * nslab) * a.n);
    DMVAE_LAUNCH(adam_slabs_kernel, dim3(blocks), dim3(256), 0, s, a, slabs, nslab, stride, seg_lo, seg_hi);
    return check_launch("adam_slabs");
}
int adam_finish_launch(hipStream_t s, void* st) {
    DMVAE_LAUNCH(adam_finish_kernel, dim3(1), dim3(1), 0, s, reinterpret_cast<dmvae_state*>(st));
    return check_launch("adam_finish");
}

// ---------------------------------------------------------------- column sums
// out[n] = sum_m in[m][n].  Two passes in a fixed order (no atomics):
//   pass 1: grid (N/256-col strips) x (row slabs of 256 rows): each thread sums
//           its column over the slab (coalesced across the 256 lanes) -> ws[slab][n]
//   pass 2: each thread sums its column over the slabs.
// When M <= 256 a single pass writes out directly.
template <typename T>
__global__ __launch_bounds__(256) void colsum_kernel(const T* in, int64_t ld, int M, int N, int rows_per_slab, float* out, int64_t ldo) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int r0 = blockIdx.y * rows_per_slab;
    const int r1 = min(M, r0 + rows_per_slab);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        float v0, v1, v2, v3;
        if constexpr (sizeof(T) == 2) {
            v0 = bf2f(in[(int64_t)(r + 0) * ld + n]); v1 = bf2f(in[(int64_t)(r + 1) * ld + n]);
            v2 = bf2f(in[(int64_t)(r + 2) * ld + n]); v3 = bf2f(in[(int64_t)(r + 3) * ld + n]);
        } else {
            v0 = in[(int64_t)(r + 0) * ld + n]; v1 = in[(int64_t)(r + 1) * ld + n];
            v2 = in[(int64_t)(r + 2) * ld + n]; v3 = in[(int64_t)(r + 3) * ld + n];
        }
        s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; r < r1; ++r) {
        if constexpr (sizeof(T) == 2) s0 += bf2f(in[(int64_t)r * ld + n]);
        else s0 += in[(int64_t)r * ld + n];
    }
    out[(int64_t)blockIdx.y * ldo + n] = (s0 + s1) + (s2 + s3);
}

static float* g_colsum_ws = nullptr;     // [COLSUM_MAX_SLABS][COLSUM_MAX_N] scratch, allocated once (outside capture)
static int64_t g_colsum_ws_elems = 0;
constexpr int COLSUM_SLABS = 64;

int colsum_prepare(int64_t max_n) {      // called from plan bind / first use, never under capture
    const int64_t need = (int64_t)COLSUM_SLABS * max_n;
    if (need <= g_colsum_ws_elems) return 0;
    if (g_colsum_ws) (void)hipFree(g_colsum_ws);
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&g_colsum_ws), need * sizeof(float));
    if (e != hipSuccess) { set_error("colsum scratch hipMalloc failed: %s", hipGetErrorString(e)); g_colsum_ws_elems = 0; return (int)e; }
    g_colsum_ws_elems = need;
    return 0;
}

float* colsum_global_scratch(int64_t* elems) { *elems = g_colsum_ws_elems; return g_colsum_ws; }

int colsum_launch(hipStream_t s, int in_dtype, const void* in, int64_t ld, int M, int N, float* out, float* ws, int64_t ws_elems) {
    const int strips = (N + 255) / 256;
    const double bytes = (double)M * N * (in_dtype == DMVAE_BF16 ? 2 : 4);
    ProfScope ps(s, "colsum", (double)M * N, bytes);
    if (M <= 64) {
        if (in_dtype == DMVAE_BF16) DMVAE_LAUNCH(colsum_kernel<bf16_t>, dim3(strips, 1), dim3(256), 0, s, (const bf16_t*)in, ld, M, N, M, out, (int64_t)0);
        else DMVAE_LAUNCH(colsum_kernel<float>, dim3(strips, 1), dim3(256), 0, s, (const float*)in, ld, M, N, M, out, (int64_t)0);
        return check_launch("colsum");
    }
    if (!ws) { ws = g_colsum_ws; ws_elems = g_colsum_ws_elems; }
    int rps = (M + COLSUM_SLABS - 1) / COLSUM_SLABS;
    if (rps < 8) rps = 8;
    const int slabs = (M + rps - 1) / rps;
    if ((int64_t)slabs * N > ws_elems) {
        set_error("dmvae_colsum: scratch too small (%lld < %lld); call through a bound plan or with N <= prepared", (long long)ws_elems, (long long)slabs * N);
        return DMVAE_ESTATE;
    }
    if (in_dtype == DMVAE_BF16) DMVAE_LAUNCH(colsum_kernel<bf16_t>, dim3(strips, slabs), dim3(256), 0, s, (const bf16_t*)in, ld, M, N, rps, ws, (int64_t)N);
    else DMVAE_LAUNCH(colsum_kernel<float>, dim3(strips, slabs), dim3(256), 0, s, (const float*)in, ld, M, N, rps, ws, (int64_t)N);
    DMVAE_LAUNCH(colsum_kernel<float>, dim3(strips, 1), dim3(256), 0, s, (const float*)ws, (int64_t)N, slabs, N, slabs, out, (int64_t)0);
    return check_launch("colsum");
}

// ---------------------------------------------------------------- slab reduction (deterministic split-K)
// out[i] = sum over the slabs in a FIXED order; n a multiple of 4, 16-byte accesses.  A block = 16 slab groups x 16 quads:
// thread (g, q) adds slabs g, g + 16, g + 32, ... in ascending order, then the 16 group sums are added in ascending order
// (the slab count is often in the hundreds while n is a few thousand quads: one thread per quad would run 20 blocks).
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int64_t n4, int nslab, int64_t stride4, float* __restrict__ out) {
    __shared__ float4 red[16][17];
    const int q = threadIdx.x & 15, g = threadIdx.x >> 4;
    for (int64_t base = (int64_t)blockIdx.x * 16; base < n4; base += (int64_t)gridDim.x * 16) {
        const int64_t i = base + q;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n4)
            for (int s = g; s < nslab; s += 16) {
                const float4 b = reinterpret_cast<const float4*>(slabs)[(int64_t)s * stride4 + i];
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
        red[g][q] = a;
        __syncthreads();
        if (g == 0 && i < n4) {
            float4 t = red[0][q];
#pragma unroll
            for (int k = 1; k < 16; ++k) { const float4 b = red[k][q]; t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w; }
            reinterpret_cast<float4*>(out)[i] = t;
        }
        __syncthreads();
    }
}
int slab_reduce_launch(hipStream_t s, const float* slabs, int64_t n, int nslab, int64_t stride, float* out) {
    if (n % 4 || stride % 4) { set_error("slab_reduce: sizes must be multiples of 4"); return DMVAE_EINVAL; }
    ProfScope ps(s, "slab_reduce", (double)n * nslab, 4.0 * n * (nslab + 1));
    const int blocks = (int)std::min<int64_t>(4096, (n / 4 + 15) / 16);
    DMVAE_LAUNCH(slab_reduce_kernel, dim3(std::max(1, blocks)), dim3(256), 0, s, slabs, n / 4, nslab, stride / 4, out);
    return check_launch("slab_reduce");
}

// ---------------------------------------------------------------- stand-alone recon loss
// base_models.py:72-85.  Each thread handles 4 consecutive columns (16-B loads).
template <typename ACT>
__global__ __launch_bounds__(256) void recon_kernel(int recon_kind, int B, int B_pad, int I, int I_pad,
                                                    const float* logits, int64_t ldl, const float* x, int64_t ldx,
                                                    float inv_B, void* dl, int64_t ldd, float* partials) {
    __shared__ float red[4];
    const int quads = I_pad >> 2;
    const int64_t total = (int64_t)B_pad * quads;
    float loss = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / quads), n = (int)(i % quads) * 4;
        float l[4], xv[4], d[4];
        loadf4(logits, (int64_t)m * ldl + n, l);
        loadf4(x, (int64_t)m * ldx + n, xv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = m < B && (n + j) < I;
            if (recon_kind == 0) {
                loss += ok ? xent_logits(l[j], xv[j]) : 0.f;
                d[j] = ok ? (sigmoidf_(l[j]) - xv[j]) * inv_B : 0.f;
            } else {
                const float r = l[j] - xv[j];
                loss += ok ? 0.5f * r * r : 0.f;
                d[j] = ok ? r * inv_B : 0.f;
            }
        }
        ActIO<ACT>::store4(dl, (int64_t)m * ldd + n, d);
    }
    const float t = block_sum_256(loss, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}
int recon_nblocks(int B_pad, int I_pad) {
    int64_t q = (int64_t)B_pad * (I_pad / 4);
    int64_t b = (q + 255) / 256;
    return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}
int recon_launch(hipStream_t s, int act_dtype, int recon_kind, int B, int B_pad, int I, int I_pad,
                 const float* logits, int64_t ldl, const float* x, int64_t ldx, float inv_B,
                 void* dl, int64_t ldd, float* partials) {
    const int nb = recon_nblocks(B_pad, I_pad);
    ProfScope ps(s, "recon_fwd_bwd", 10.0 * B * I, (double)B * I * (8 + (act_dtype == DMVAE_BF16 ? 2 : 4)));
    if (act_dtype == DMVAE_BF16) DMVAE_LAUNCH(recon_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, recon_kind, B, B_pad, I, I_pad, logits, ldl, x, ldx, inv_B, dl, ldd, partials);
    else DMVAE_LAUNCH(recon_kernel<float>, dim3(nb), dim3(256), 0, s, recon_kind, B, B_pad, I, I_pad, logits, ldl, x, ldx, inv_B, dl, ldd, partials);
    return check_launch("recon_fwd_bwd");
}

// ---------------------------------------------------------------- loss finalize
// loss = recon + kl_ratio*(KL_C + KL_Z); base_models.py:87-93 and :130.
__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* rp, int nr, const float* lp, int nl, float inv_B, dmvae_state* st, int bump_adam) {
    __shared__ float red[4];
    float a = 0.f, z = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < nr; i += 256) a += rp[i];
    for (int i = threadIdx.x; i < nl; i += 256) { z += lp[2 * i]; c += lp[2 * i + 1]; }
    const float recon = block_sum_256(a, red) * inv_B;
    const float klz = block_sum_256(z, red) * inv_B;
    const float klc = block_sum_256(c, red) * inv_B;
    if (threadIdx.x == 0) {
        const float loss = recon + st->kl_ratio * (klc + klz);
        st->last_loss = loss; st->last_recon = recon; st->last_klz = klz; st->last_klc = klc;
        st->epoch_loss += loss * st->epoch_weight;
        st->epoch_recon += recon * st->epoch_weight;
        st->epoch_klz += klz * st->epoch_weight;
        st->epoch_klc += klc * st->epoch_weight;
        st->noise_step += 1;
        if (bump_adam) st->adam_t += 1;      // the update that follows in the same step reads t = adam_t
        st->batch_cursor = (st->batches_per_epoch > 0) ? (st->batch_cursor + 1) % st->batches_per_epoch : 0;
    }
}
int loss_finalize_launch(hipStream_t s, const float* rp, int nr, const float* lp, int nl, float inv_B, void* st, int bump_adam) {
    DMVAE_LAUNCH(loss_finalize_kernel, dim3(1), dim3(256), 0, s, rp, nr, lp, nl, inv_B, reinterpret_cast<dmvae_state*>(st), bump_adam);
    return check_launch("loss_finalize");
}

// ---------------------------------------------------------------- step finalize
// Everything of a training step that only needs the forward pass's per-block partials, in ONE
// launch (a kernel boundary costs ~4.6 us on this chip; the plan runs it on a side branch beside
// the backward GEMMs):  block 0 = the loss scalars (loss_finalize_kernel's work);
// blocks 1.. = the prior-table gradients, d/d(prior) = sum over the latent kernel's blocks of
// their partials [nblk][ncol], summed in a FIXED order: 16 row groups x 16 columns per block, each
// thread adds its rows in ascending order, then the 16 groups are added in ascending order.
__global__ __launch_bounds__(256) void step_finalize_kernel(dmvae_finalize_args f) {
    __shared__ float red[16][17];
    step_finalize_block((int)blockIdx.x, f, red);
}
dmvae_finalize_args step_finalize_args(const float* rp, int nr, const float* lp, int nl, float inv_B, void* st, int bump_adam,
                                       float b1, float b2, const float* part, int nblk, int ncol, float* gout) {
    dmvae_finalize_args f;
    f.rp = rp; f.nr = nr; f.lp = lp; f.nl = nl; f.inv_B = inv_B; f.st = reinterpret_cast<dmvae_state*>(st); f.bump_adam = bump_adam;
    f.b1 = b1; f.b2 = b2; f.part = part; f.nblk = nblk; f.ncol = ncol; f.gout = gout; f.nblocks = 1 + (ncol + 15) / 16;
    return f;
}
int step_finalize_launch(hipStream_t s, const float* rp, int nr, const float* lp, int nl, float inv_B, void* st, int bump_adam,
                         float b1, float b2, const float* part, int nblk, int ncol, float* gout) {
    ProfScope ps(s, "step_finalize", (double)nblk * ncol, 4.0 * ((double)nblk * ncol + nr + 2.0 * nl));
    const dmvae_finalize_args f = step_finalize_args(rp, nr, lp, nl, inv_B, st, bump_adam, b1, b2, part, nblk, ncol, gout);
    DMVAE_LAUNCH(step_finalize_kernel, dim3(f.nblocks), dim3(256), 0, s, f);
    return check_launch("step_finalize");
}

// ---------------------------------------------------------------- batch assembly
// Dataset.get_batches (includes/utils.py:449-463): row r <- data[perm[first+r]] -- gather_rows_block (gemm_epilogue.h) over a grid of its own.
template <typename ACT>
__global__ __launch_bounds__(256) void gather_kernel(dmvae_gather_args g) { gather_rows_block<ACT, 4>((int)blockIdx.x, 256, g); }
dmvae_gather_args gather_args(int act_dtype, const float* data, int64_t n_rows, int dim, const int32_t* perm, int64_t first, int batch, int n_valid,
                              int B_pad, void* out_act, int64_t ld_act, float* out_f32, int64_t ld_f32, int cols_pad, const void* st) {
    dmvae_gather_args g;
    g.data = data; g.n_rows = n_rows; g.dim = dim; g.perm = perm; g.first = first; g.batch = batch; g.n_valid = n_valid; g.B_pad = B_pad;
    g.out_act = out_act; g.ld_act = ld_act; g.out_f32 = out_f32; g.ld_f32 = ld_f32; g.cols_pad = cols_pad; g.st = (const dmvae_state*)st;
    const int64_t q = (int64_t)B_pad * (cols_pad / 4);
    g.nblocks = (int)std::min<int64_t>(2048, std::max<int64_t>(1, (q + 1023) / 1024));      // four quads per thread and pass
    return g;
}
int gather_launch(hipStream_t s, int act_dtype, const float* data, int64_t n_rows, int dim, const int32_t* perm,
                  int64_t first, int batch, int n_valid, int B_pad, void* out_act, int64_t ld_act,
                  float* out_f32, int64_t ld_f32, int cols_pad, const void* st) {
    const dmvae_gather_args g = gather_args(act_dtype, data, n_rows, dim, perm, first, batch, n_valid, B_pad, out_act, ld_act, out_f32, ld_f32, cols_pad, st);
    if ((int64_t)B_pad * (cols_pad / 4) >= (int64_t(1) << 31) - int64_t(2048) * 256 * 8) { set_error("dmvae_gather_rows: batch x columns too large for 32-bit quad indices"); return DMVAE_EINVAL; }
    ProfScope ps(s, "gather_rows", 0.0, (double)n_valid * dim * (4 + 4 + (act_dtype == DMVAE_BF16 ? 2 : 4)));
    if (act_dtype == DMVAE_BF16) DMVAE_LAUNCH(gather_kernel<bf16_t>, dim3(g.nblocks), dim3(256), 0, s, g);
    else DMVAE_LAUNCH(gather_kernel<float>, dim3(g.nblocks), dim3(256), 0, s, g);
    return check_launch("gather_rows");
}

// ---------------------------------------------------------------- profiling aid
// Keeps the stream busy for `us` microseconds (bounded) so that the host can enqueue the
// launches that follow before the GPU reaches them: HIP-event brackets around those launches
// then measure kernel time, not host launch latency.  Spins on the constant 100 MHz counter.
__global__ void spin_kernel(unsigned long long ticks) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}
int spin_launch(hipStream_t s, int us) {
    if (us < 0) us = 0;
    if (us > 20000) us = 20000;
    DMVAE_LAUNCH(spin_kernel, dim3(1), dim3(1), 0, s, (unsigned long long)us * 100ull);
    return check_launch("spin");
}

// ---------------------------------------------------------------- noise + casts
__global__ __launch_bounds__(256) void philox_kernel(float* out, int64_t n, uint64_t seed, uint64_t step, uint32_t sid, int gumbel) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = gumbel ? philox_gumbel_at(seed, step, sid, (uint64_t)i) : philox_normal_at(seed, step, sid, (uint64_t)i);
}
int philox_launch(hipStream_t s, float* out, int64_t n, uint64_t seed, uint64_t step, uint32_t sid, int gumbel) {
    int nb = (int)((n + 255) / 256);
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    DMVAE_LAUNCH(philox_kernel, dim3(nb), dim3(256), 0, s, out, n, seed, step, sid, gumbel);
    return check_launch("philox");
}

__global__ __launch_bounds__(256) void cast_f2b_kernel(const float* in, bf16_t* out, int64_t n) {
    const int64_t n4 = n >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(in)[i];
        uint2 q; q.x = pack2bf(v.x, v.y); q.y = pack2bf(v.z, v.w);
        reinterpret_cast<uint2*>(out)[i] = q;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = f2bf(in[i]);
}
__global__ __launch_bounds__(256) void cast_b2f_kernel(const bf16_t* in, float* out, int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) out[i] = bf2f(in[i]);
}
int cast_launch(hipStream_t s, const void* in, void* out, int64_t n, int to_bf16) {
    int nb = (int)((n / 4 + 255) / 256);
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    if (to_bf16) DMVAE_LAUNCH(cast_f2b_kernel, dim3(nb), dim3(256), 0, s, (const float*)in, (bf16_t*)out, n);
    else DMVAE_LAUNCH(cast_b2f_kernel, dim3(nb), dim3(256), 0, s, (const bf16_t*)in, (float*)out, n);
    return check_launch("cast");
}

}  // namespace dmvae
