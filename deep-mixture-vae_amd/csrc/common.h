// Shared device/host helpers for the gfx950 DMVAE kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/dmvae_hip_debug.h"

typedef unsigned short bf16_t;   // raw bf16 bits in memory
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WAVE 64

// round-to-nearest-even f32 -> bf16 through the hardware convert (keeps NaN a NaN)
__device__ __forceinline__ bf16_t f2bf(float f) {
    __bf16 h = (__bf16)f;
    return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ float bf2f(bf16_t h) {
    return __uint_as_float(((unsigned int)h) << 16);
}
// both halves in ONE v_cvt_pk_bf16_f32 (two scalar converts cost a convert each and an or / sdwa to join them; same rounding)
__device__ __forceinline__ unsigned int pack2bf(float lo, float hi) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const bf16x2_ h = __builtin_convertvector(f32x2_{lo, hi}, bf16x2_);
    return __builtin_bit_cast(unsigned int, h);
}

// TF-1.x Adam (base_models.py:95-110), shared by the stand-alone kernel and the dW-epilogue form so
// that both produce the same bits: every product and sum rounded on its own (no FMA contraction).
__device__ __forceinline__ float adam_lr_t(float lr, float b1, float b2, uint64_t t) {
    // pow on the exact integer t; double keeps 1 - b2^t accurate for small t
    const double b1t = pow((double)b1, (double)t), b2t = pow((double)b2, (double)t);
    return (float)((double)lr * sqrt(1.0 - b2t) / (1.0 - b1t));
}
// FAST (the bf16 throughput mode: fused dW epilogues and the stand-alone kernel whenever a bf16 shadow is written): the update
// quotient lr_t m / (sqrt(v) + eps) with the hardware square root and reciprocal (v_sqrt_f32, v_rcp_f32: 1 ulp each) instead
// of the IEEE sequences (~25 of the ~35 instructions per element; the fused epilogue of a 256x256 tile was ALU-bound on them).
// The fp32 parity mode keeps IEEE sqrt and division.  Both forms of one mode share this function: same bits.
template <bool FAST = false>
__device__ __forceinline__ void adam_elem(float& p, float& m, float& v, float g, float gscale, float b1, float b2, float eps, float lr_t) {
    // plain operators under contract(off): HIP's __fmul_rn / __fadd_rn are header functions whose * and +
    // still carry the contract flag, so each caller fused them differently (1-ulp differences in m)
#pragma clang fp contract(off)
    const float gj = g * gscale;
    const float m1 = b1 * m, m2 = (1.f - b1) * gj;
    m = m1 + m2;
    const float v1 = b2 * v, v2 = ((1.f - b2) * gj) * gj;
    v = v1 + v2;
    if constexpr (FAST) {
        const float num = lr_t * m, den = __builtin_amdgcn_sqrtf(v) + eps;
        const float quo = num * __builtin_amdgcn_rcpf(den);
        p = p - quo;
    } else {
        const float num = lr_t * m, den = sqrtf(v) + eps;
        p = p - num / den;
    }
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global
// store the wave has in flight (vmcnt(0): the write acknowledgement, 2-3 us from HBM) -- wasted
// when the stores are write-only outputs nobody in the workgroup reads back.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Fixed-order block sum for 256-thread blocks; result valid in thread 0.
__device__ __forceinline__ float block_sum_256(float v, float* red /* >= 4 floats LDS */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

// step_finalize (elementwise.hip) as a device function, so that its 1 + ceil(ncol/16) blocks can also ride
// as extra workgroups of another launch of the step (the grouped heads-dX GEMM): saves a kernel boundary.
struct dmvae_finalize_args {
    const float* rp; int nr;          // reconstruction-loss partials
    const float* lp; int nl;          // latent kernel partials {kl_z, kl_c} per block
    float inv_B; dmvae_state* st; int bump_adam; float b1, b2;
    const float* part; int nblk, ncol; float* gout;      // prior-table gradient partials [nblk][ncol] -> gout[ncol]
    int nblocks;                      // 1 + ceil(ncol / 16)
};
// block 0 = loss scalars, epoch accumulators, Adam t / lr_t, batch cursor;  blocks 1.. = the prior-table
// gradients summed in a FIXED order: 16 row groups x 16 columns per block, each thread adds its rows in ascending
// order, then the 16 groups are added in ascending order.  256 threads; red = 16 x 17 floats of LDS.
__device__ __forceinline__ void step_finalize_block(const int blk, const dmvae_finalize_args& f, float (*red)[17]) {
    if (blk == 0) {
        float a = 0.f, z = 0.f, c = 0.f;
        for (int i = threadIdx.x; i < f.nr; i += 256) a += f.rp[i];
        for (int i = threadIdx.x; i < f.nl; i += 256) { z += f.lp[2 * i]; c += f.lp[2 * i + 1]; }
        float* r4 = &red[0][0];
        const float recon = block_sum_256(a, r4) * f.inv_B;
        const float klz = block_sum_256(z, r4) * f.inv_B;
        const float klc = block_sum_256(c, r4) * f.inv_B;
        if (threadIdx.x == 0) {
            dmvae_state* st = f.st;
            const float loss = recon + st->kl_ratio * (klc + klz);
            st->last_loss = loss; st->last_recon = recon; st->last_klz = klz; st->last_klc = klc;
            st->epoch_loss += loss * st->epoch_weight;
            st->epoch_recon += recon * st->epoch_weight;
            st->epoch_klz += klz * st->epoch_weight;
            st->epoch_klc += klc * st->epoch_weight;
            st->noise_step += 1;
            if (f.bump_adam) {     // the update of this step (stand-alone or fused into the dW launch) uses t = adam_t
                st->adam_t += 1;
                st->lr_t = adam_lr_t(st->lr, f.b1, f.b2, st->adam_t);
            }
            st->batch_cursor = (st->batches_per_epoch > 0) ? (st->batch_cursor + 1) % st->batches_per_epoch : 0;
        }
        return;
    }
    const int col = (blk - 1) * 16 + (threadIdx.x & 15), rg = threadIdx.x >> 4;
    float s = 0.f;
    if (col < f.ncol) {
#pragma unroll 8                          // (eight loads in flight per lane; the order of the additions is unchanged)
        for (int r = rg; r < f.nblk; r += 16) s += f.part[(int64_t)r * f.ncol + col];
    }
    red[rg][threadIdx.x & 15] = s;
    __syncthreads();
    if (threadIdx.x < 16 && col < f.ncol) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][threadIdx.x];
        f.gout[col] = t;
    }
}

// Fixed-order block sum over NW waves; result valid in thread 0.
template <int NW>
__device__ __forceinline__ float block_sum_waves(float v, float* red /* >= NW floats LDS */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    float r = 0.f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < NW; ++i) r += red[i];
    }
    __syncthreads();
    return r;
}

// ---- Philox4x32-10 (counter-based; same stream on every replay of a graph) ----
__host__ __device__ __forceinline__ void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}
// uniform in (0,1]: never 0 so log() is finite
__host__ __device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f); }

// element `idx` of noise stream (seed, step, stream_id): one Philox block gives 4 values;
// value j = idx & 3 of block idx >> 2.
__device__ __forceinline__ void philox_block(uint64_t seed, uint64_t step, uint32_t stream_id, uint64_t blk, uint32_t out[4]) {
    out[0] = (uint32_t)blk; out[1] = (uint32_t)(blk >> 32) ^ (stream_id << 24);
    out[2] = (uint32_t)step; out[3] = (uint32_t)(step >> 32);
    philox4x32_10(out, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__device__ __forceinline__ float philox_normal_at(uint64_t seed, uint64_t step, uint32_t stream_id, uint64_t idx) {
    uint32_t r[4];
    philox_block(seed, step, stream_id, idx >> 1, r);
    // Box-Muller: block idx>>1 yields two normals from (r0,r1)
    const float rad = sqrtf(-2.0f * __logf(u01(r[0])));
    const float ang = 6.283185307179586f * u01(r[1]);
    return (idx & 1) ? rad * __sinf(ang) : rad * __cosf(ang);
}
// four independent N(0,1) draws from ONE Philox block (two Box-Muller pairs): the latent kernel's
// lanes own four columns each, so this replaces four blocks of ten rounds by one
__device__ __forceinline__ void philox_normal4(uint64_t seed, uint64_t step, uint32_t stream_id, uint64_t blk, float (&n)[4]) {
    uint32_t r[4];
    philox_block(seed, step, stream_id, blk, r);
    const float rad0 = sqrtf(-2.0f * __logf(u01(r[0]))), ang0 = 6.283185307179586f * u01(r[1]);
    const float rad1 = sqrtf(-2.0f * __logf(u01(r[2]))), ang1 = 6.283185307179586f * u01(r[3]);
    n[0] = rad0 * __cosf(ang0); n[1] = rad0 * __sinf(ang0);
    n[2] = rad1 * __cosf(ang1); n[3] = rad1 * __sinf(ang1);
}
__device__ __forceinline__ float philox_gumbel_at(uint64_t seed, uint64_t step, uint32_t stream_id, uint64_t idx) {
    uint32_t r[4];
    philox_block(seed, step, stream_id, idx >> 2, r);
    const float U = u01(r[idx & 3]);
    // includes/utils.py:17-19: -log(eps - log(U + eps)), eps = 1e-20 (vanishes in f32 for U in (0,1])
    return -__logf(1e-20f - __logf(U));
}

// ---- host side: errors + launch profiling ----
namespace dmvae {
void set_error(const char* fmt, ...);
struct ProfScope {   // brackets a launch with a hipEvent pair when profiling is on
    ProfScope(hipStream_t s, const char* name, double flops, double bytes);
    ~ProfScope();
    hipStream_t s; int slot, outer;
};
// profiling on and a ProfScope open on this thread: a fresh event pair for ONE kernel dispatch, kept with that scope.
// The pair goes to hipExtLaunchKernelGGL, which binds both events to the dispatch itself, so their elapsed time is the
// kernel's own begin -> end timestamps (the dispatch's completion signal: what rocprofv3 --kernel-trace reports),
// with none of the dispatch / event-marker time an event BRACKET around the launch contains.
bool prof_launch_events(hipEvent_t* e0, hipEvent_t* e1);
int check_launch(const char* what);
}  // namespace dmvae

// every kernel launch of the library: a plain launch, or -- while bench.py's roofline leg profiles eager steps -- the same
// launch with its dispatch timestamps captured (never under stream capture: profiling is an eager-mode facility)
#define DMVAE_LAUNCH(kernel, grid, block, lds, stream, ...)                                                        \
    do {                                                                                                           \
        hipEvent_t pe0_ = nullptr, pe1_ = nullptr;                                                                 \
        if (::dmvae::prof_launch_events(&pe0_, &pe1_))                                                             \
            hipExtLaunchKernelGGL(kernel, grid, block, (unsigned)(lds), stream, pe0_, pe1_, 0u, __VA_ARGS__);      \
        else                                                                                                       \
            hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                     \
    } while (0)

#define DMVAE_REQUIRE(cond, ...)                      \
    do {                                              \
        if (!(cond)) {                                \
            dmvae::set_error(__VA_ARGS__);            \
            return DMVAE_EINVAL;                      \
        }                                             \
    } while (0)
