// The narrow middle of the backward pass as ONE kernel over 16-row blocks (bf16 plans, DMVAE model):
//     dZ        = d(dec0) . W_dec0^T                                  (tf.gradients of base_models.py:280-283)
//     dmean     = dZ + gmu,  dlog_var = dZ * clv + glv                (priors.py:86-89 backward + KL gradients, latent.hip)
//     d z-hidden = ([dmean | dlog_var] . W_mv^T) * (z-hidden > 0)     (base_models.py:229-239)
//     d c-hidden = (dlogits . W_logits^T) * (c-hidden > 0)            (base_models.py:241-248)
// i.e. the dZ GEMM with its DMVAE_EPI_LATENT epilogue and the two head dX GEMMs: three launches whose GEMMs are tiny
// (N = 64 or K <= 128: the 128-row tile kernels spend their 11 + 24 us at cfg2 on prologues, epilogues and 64 of 256 CUs).
// Every contraction here has its weight operand K-CONTIGUOUS in memory (W^T products of row-major [in][out] weights), so
// both MFMA operands are plain 16-byte global loads straight into registers -- no LDS staging, no transposing reads; a
// block streams the three weight matrices (1 MB at cfg2) once for its 16 rows.  Wave w of 4 owns latent column tiles
// w, w + 4, ... in phase 1 and every fourth 64-column group of the 2 x head_dim outputs in phase 3; the [16][2D]
// intermediate goes through LDS (it is phase 3's second operand), the outputs leave through a wave-private LDS tile in
// 128-byte row segments.  The step_finalize workgroups ride on this launch as they did on the grouped heads-dX launch.
//
// MEASURED (tools/knob_ab.py 9 0 1; correct: tests/test_gpu_configs.py ..._with_the_fused_middle_backward) and NOT FASTER:
// cfg2 0.2973 -> 0.3246 ms/step, cfg4 0.691 -> 0.864, cfg3 1.049 -> 1.334.  The kernel takes ~60 us where the three launches
// take 35: fragment-shaped global loads (a wave instruction = 16 rows x 64 B) are bound by the texture-addresser, a CU
// streams its 1.3 MB at ~22 GB/s instead of the ~70 GB/s that full 128-byte lines through LDS-DMA reach (the CDNA guide's
// "x through LDS in full lines" rule, re-learnt).  Off by default (knob 9); kept as the measured negative result.
#include <algorithm>

#include "kernels.h"

namespace dmvae {

__device__ __forceinline__ bf16x8 ld_frag(const bf16_t* p) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const s16x8*>(p));
}

constexpr int MB_MAXKS = 16;       // 2 Dp / 32 <= 16  (Dp <= 256)

__global__ __launch_bounds__(256) void mid_bwd_kernel(MidBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if ((int)blockIdx.x >= a.nrow_blocks) {
        step_finalize_block((int)blockIdx.x - a.nrow_blocks, a.fin, reinterpret_cast<float(*)[17]>(lds_raw));
        return;
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * 16;
    const int D2 = 2 * a.Dp;
    bf16_t* A3 = reinterpret_cast<bf16_t*>(lds_raw);                 // [16][2 Dp + 8]  [dmean | dlog_var] (row stride padded by 16 B)
    const int s3 = D2 + 8;
    bf16_t* A3b = A3 + 16 * s3;                                      // [16][Kp + 8]    dlogits
    const int s3b = a.Kp + 8;
    float* stg = reinterpret_cast<float*>(A3b + 16 * s3b) + wave * (16 * 68);     // wave-private [16][64 + 4] f32

    // dlogits rows of this block -> LDS (phase 3's second operand for the c half)
    for (int i = tid; i < 16 * (a.Kp / 8); i += 256) {
        const int r = i / (a.Kp / 8), c = i % (a.Kp / 8);
        *reinterpret_cast<uint4*>(A3b + r * s3b + c * 8) = *reinterpret_cast<const uint4*>(a.dlg + (int64_t)(row0 + r) * a.ld_dlg + c * 8);
    }

    // ---- phase 1: dZ tile(s) of this wave, K = N0; both operands straight from global memory
    const bf16_t* arow = a.ddec0 + (int64_t)(row0 + li) * a.ld_dd + 8 * g;
    for (int t = wave; t < a.Dp / 16; t += 4) {
        const bf16_t* wrow = a.Wd0 + (int64_t)(t * 16 + li) * a.ld_wd0 + 8 * g;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        const int nks = a.N0 / 32;
#pragma unroll 8
        for (int ks = 0; ks < nks; ++ks)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(wrow + ks * 32), ld_frag(arow + ks * 32), acc, 0, 0, 0);
        // reparameterisation + KL backward on the quad dZ[m = li][d0 .. d0 + 3]
        const int d0 = t * 16 + 4 * g;
        const int64_t go = (int64_t)(row0 + li) * a.ld_g + d0;
        const float4 gm = *reinterpret_cast<const float4*>(a.gmu + go);
        const float4 gl = *reinterpret_cast<const float4*>(a.glv + go);
        const float4 cl = *reinterpret_cast<const float4*>(a.clv + go);
        const float dm[4] = {acc[0] + gm.x, acc[1] + gm.y, acc[2] + gm.z, acc[3] + gm.w};
        const float dl[4] = {acc[0] * cl.x + gl.x, acc[1] * cl.y + gl.y, acc[2] * cl.z + gl.z, acc[3] * cl.w + gl.w};
        uint2 pm, pl;
        pm.x = pack2bf(dm[0], dm[1]); pm.y = pack2bf(dm[2], dm[3]);
        pl.x = pack2bf(dl[0], dl[1]); pl.y = pack2bf(dl[2], dl[3]);
        *reinterpret_cast<uint2*>(a.dmv + (int64_t)(row0 + li) * a.ld_dmv + d0) = pm;
        *reinterpret_cast<uint2*>(a.dmv + (int64_t)(row0 + li) * a.ld_dmv + a.Dp + d0) = pl;
        *reinterpret_cast<uint2*>(A3 + li * s3 + d0) = pm;
        *reinterpret_cast<uint2*>(A3 + li * s3 + a.Dp + d0) = pl;
    }
    __syncthreads();

    // ---- phase 3: second operands (rows of this block) from LDS into registers, once
    bf16x8 amv[MB_MAXKS], alg[4];
    const int ksz = D2 / 32, ksc = a.Kp / 32;
#pragma unroll
    for (int ks = 0; ks < MB_MAXKS; ++ks)
        if (ks < ksz) amv[ks] = ld_frag(A3 + li * s3 + ks * 32 + 8 * g);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        if (ks < ksc) alg[ks] = ld_frag(A3b + li * s3b + ks * 32 + 8 * g);

    // 64-column groups of [d z-hidden | d c-hidden]; group gi covers output columns gi*64 .. +63
    const int ngrp = 2 * a.Hp / 64, zgrp = a.Hp / 64;
    for (int gi = wave; gi < ngrp; gi += 4) {
        const bool zpart = gi < zgrp;
        const int n0 = (zpart ? gi : gi - zgrp) * 64;                 // row of W_mv / W_logits
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (zpart) {
            const bf16_t* w = a.Wmv + (int64_t)(n0 + li) * a.ld_wmv + 8 * g;
#pragma unroll
            for (int ks = 0; ks < MB_MAXKS; ++ks)
                if (ks < ksz) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(w + (int64_t)j * 16 * a.ld_wmv + ks * 32), amv[ks], acc[j], 0, 0, 0);
                }
        } else {
            const bf16_t* w = a.Wlg + (int64_t)(n0 + li) * a.ld_wlg + 8 * g;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                if (ks < ksc) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ld_frag(w + (int64_t)j * 16 * a.ld_wlg + ks * 32), alg[ks], acc[j], 0, 0, 0);
                }
        }
        // acc[j]: lane holds out[m = li][gi*64 + j*16 + 4g .. +3] -> wave-private tile, then 128-byte row segments
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(stg + li * 68 + j * 16 + 4 * g) = acc[j];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int item = pass * 64 + lane, r = item >> 3, c8 = (item & 7) * 8;      // row r, columns c8 .. c8 + 7 of the group
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(stg + r * 68 + c8);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(stg + r * 68 + c8 + 4);
            const int64_t o = (int64_t)(row0 + r) * a.ld_h + gi * 64 + c8;
            const uint4 y = *reinterpret_cast<const uint4*>(a.hzc + o);                     // forward activations: the ReLU gates
            const unsigned yw[4] = {y.x, y.y, y.z, y.w};
            const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            unsigned ow[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float lo = __uint_as_float(yw[q] << 16) > 0.f ? v[2 * q] : 0.f;
                const float hi = __uint_as_float(yw[q] & 0xffff0000u) > 0.f ? v[2 * q + 1] : 0.f;
                ow[q] = pack2bf(lo, hi);
            }
            *reinterpret_cast<uint4*>(a.dhzc + (int64_t)(row0 + r) * a.ld_dh + gi * 64 + c8) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // tile read before the next group overwrites it
    }
}

bool mid_bwd_applies(int Bp, int Dp, int Kp, int Hp, int N0) {
    return Bp % 16 == 0 && Dp % 16 == 0 && 2 * Dp / 32 <= MB_MAXKS && Kp % 32 == 0 && Kp / 32 <= 4 && Hp % 64 == 0 && N0 % 256 == 0;
}

int mid_bwd_launch(hipStream_t s, const MidBwdArgs& a0) {
    MidBwdArgs a = a0;
    if (!mid_bwd_applies(a.Bp, a.Dp, a.Kp, a.Hp, a.N0)) { set_error("mid_bwd: shape not supported"); return DMVAE_EINVAL; }
    a.nrow_blocks = a.Bp / 16;
    const size_t lds = std::max<size_t>(16 * 17 * 4, (size_t)16 * (2 * a.Dp + 8) * 2 + (size_t)16 * (a.Kp + 8) * 2 + 4 * 16 * 68 * 4);
    const double flops = 2.0 * a.Bp * ((double)a.Dp * a.N0 + (double)a.Hp * (2 * a.Dp + a.Kp));
    const double bytes = 2.0 * a.Bp * ((double)a.N0 + 4.0 * a.Hp + 2.0 * a.Dp + a.Kp) + 12.0 * a.Bp * a.Dp;
    ProfScope ps(s, "mid_bwd", flops, bytes);
    DMVAE_LAUNCH(mid_bwd_kernel, dim3(a.nrow_blocks + a.fin.nblocks), dim3(256), lds, s, a);
    return check_launch("mid_bwd");
}

}  // namespace dmvae
