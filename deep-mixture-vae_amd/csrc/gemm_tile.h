// Device helpers shared by the bf16 MFMA GEMM kernels (gemm_bf16.hip: 64..128-wide tiles, two workgroups per CU;
// gemm_bf16_256.hip: the 256x256 macro tile, one workgroup per CU): swizzled LDS tile images, LDS-DMA issue,
// MFMA fragment reads, the TF-Adam quad of the fused dW epilogue, XCD-aware workgroup order.
#pragma once
#include "kernels.h"

namespace dmvae {

constexpr int BK = 64;

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

// swizzled 16-byte chunk of a k-contiguous tile row (8 chunks per 128-B row)
__device__ __forceinline__ int swz_kc(int row, int chunk) { return chunk ^ ((row >> 1) & 7); }
// swizzled chunk of an n-contiguous tile row k; C = tile columns (128 -> 16 chunks, 64 -> 8 chunks)
template <int C>
__device__ __forceinline__ int swz_nc(int k, int chunk) {
    if constexpr (C == 128) return chunk ^ ((((k & 3) | (((k >> 3) & 1) << 2))) << 1);
    else return chunk ^ (((((k >> 1) & 1) | (((k >> 3) & 1) << 1))) << 1);
}

// Per-lane source BYTE offsets (relative to the tile origin) of the R/32 loads of one operand
// tile; loop invariant, computed once.
//   KC : tile [R rows][64 k]   : a wave instruction covers 8 rows x 128 B
//   !KC: tile [64 k][R cols]   : R = 128: 4 k-rows x 256 B;  R = 64: 8 k-rows x 128 B
// BKT = K depth of the tile: 64, or 32 for the n-contiguous (dW) operands only (half the k-rows).
template <int R, bool KC, int NW, int BKT>
__device__ __forceinline__ void stage_offsets(int64_t ld, int wave, int lane, unsigned (&off)[R * BKT / (512 * NW)]) {
    static_assert(BKT == 64 || !KC, "a k-contiguous tile row is 64 elements");
#pragma unroll
    for (int i = 0; i < R * BKT / (512 * NW); ++i) {
        int row, c;
        if constexpr (KC) {
            row = i * (8 * NW) + wave * 8 + (lane >> 3);
            c = swz_kc(row, lane & 7);
        } else if constexpr (R == 128) {
            row = i * (4 * NW) + wave * 4 + (lane >> 4);
            c = swz_nc<128>(row, lane & 15);
        } else {
            row = i * (8 * NW) + wave * 8 + (lane >> 3);
            c = swz_nc<64>(row, lane & 7);
        }
        off[i] = 2u * (unsigned)(row * (int)ld + c * 8);
    }
}

// LDS-DMA of one operand tile: NL wave instructions of 64 lanes x 16 B each,
//   LDS[lds + stride*i + lane*16 ..) <- *(tile + off[i])        (i < NL; stride = 1 KiB x waves)
// tile = wave-uniform pointer (SGPR pair), off = per-lane 32-bit byte offsets, lds = wave-uniform
// LDS byte address of this wave's first chunk (a pass of all waves covers 1 KiB x waves for every
// tile shape).  M0 (the DMA's LDS base) is compiler-reserved: saved/restored in the statement.
__device__ __forceinline__ void glds_tile(const void* tile, const unsigned (&off)[1], unsigned lds, unsigned) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(tile), "v"(off[0]), "s"(lds)
        : "memory");
}
__device__ __forceinline__ void glds_tile(const void* tile, const unsigned (&off)[2], unsigned lds, unsigned stride) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
        "s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(tile), "v"(off[0]), "v"(off[1]), "s"(lds), "s"(lds + stride)
        : "memory");
}
__device__ __forceinline__ void glds_tile(const void* tile, const unsigned (&off)[4], unsigned lds, unsigned stride) {
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %6\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
        "s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %3, %1\n\t"
        "s_mov_b32 m0, %8\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %4, %1\n\t"
        "s_mov_b32 m0, %9\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %5, %1\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "s"(tile), "v"(off[0]), "v"(off[1]), "v"(off[2]), "v"(off[3]), "s"(lds), "s"(lds + stride), "s"(lds + 2u * stride), "s"(lds + 3u * stride)
        : "memory");
}

// LDS element offsets of the fragment for index i0 + (lane & 15), k = ks*32 + 8*(lane >> 4) .. +7
// inside its operand tile (loop invariant): {lo, hi}; a k-contiguous operand needs one
// ds_read_b128 (lo only), an n-contiguous one two transposing reads.
template <int R, bool KC>
__device__ __forceinline__ void frag_offsets(int i0, int ks, int lane, unsigned short& lo, unsigned short& hi) {
    const int li = lane & 15, g = lane >> 4;
    if constexpr (KC) {
        const int row = i0 + li;
        lo = (unsigned short)(row * 64 + swz_kc(row, ks * 4 + g) * 8);
        hi = 0;
    } else {
        const int q = li >> 2, p = li & 3;
        const int k0 = ks * 32 + g * 8 + q, k1 = k0 + 4;
        const int c = (i0 >> 3) + (p >> 1);
        lo = (unsigned short)(k0 * R + swz_nc<R>(k0, c) * 8 + (p & 1) * 4);
        hi = (unsigned short)(k1 * R + swz_nc<R>(k1, c) * 8 + (p & 1) * 4);
    }
}
// The same fragment from ONE offset per 16-row / 16-column block (that of ks = 0): for an n-contiguous operand the second
// transposing read sits 4 k-rows further and the ks = 1 fragment 32 k-rows further, and neither move changes the swizzle
// (k & 3 and (k >> 3) & 1 are the same: k0 & 7 < 4) -- constant offsets, folded into the ds_read's immediate; for a
// k-contiguous operand ks = 1 flips chunk bit 2 (chunk = ks*4 + g, XOR-swizzled): element offset ^ 32.
template <int R, bool KC, int KS>
__device__ __forceinline__ bf16x8 read_frag_ks(const bf16_t* s, unsigned lo) {
    if constexpr (KC) {
        const s16x8 v = *reinterpret_cast<const s16x8*>(s + (KS ? (lo ^ 32u) : lo));
        return __builtin_bit_cast(bf16x8, v);
    } else {
        const s16x4 l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(s + lo + KS * 32 * R));
        const s16x4 h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(s + lo + KS * 32 * R + 4 * R));
        const s16x8 v = __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    }
}
template <bool KC>
__device__ __forceinline__ bf16x8 read_frag(const bf16_t* s, unsigned lo, unsigned hi) {
    if constexpr (KC) {
        const s16x8 v = *reinterpret_cast<const s16x8*>(s + lo);
        return __builtin_bit_cast(bf16x8, v);
    } else {
        const s16x4 l = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(s + lo));
        const s16x4 h = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(s + hi));
        const s16x8 v = __builtin_shufflevector(l, h, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    }
}

// TF-Adam on four consecutive arena elements whose gradient g[0..3] is in registers
// (DMVAE_EPI_ADAM, and the extra prior-table segment of the same launch).
__device__ __forceinline__ void adam_quad(const dmvae_adam_ctx& c, int64_t off, const float (&g)[4]) {
    const dmvae_state* st = reinterpret_cast<const dmvae_state*>(c.state);
    const float lr_t = st->lr_t;
    float4 p = *reinterpret_cast<const float4*>(c.param + off);
    float4 m = *reinterpret_cast<const float4*>(c.m + off);
    float4 v = *reinterpret_cast<const float4*>(c.v + off);
    float* pp = &p.x; float* mp = &m.x; float* vp = &v.x;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (c.param_bf16 && !c.ieee) adam_elem<true>(pp[j], mp[j], vp[j], g[j], c.grad_scale, c.beta1, c.beta2, c.epsilon, lr_t);       // (wave-uniform: the arithmetic follows the mode, see adam_elem)
        else adam_elem<false>(pp[j], mp[j], vp[j], g[j], c.grad_scale, c.beta1, c.beta2, c.epsilon, lr_t);
    }
    *reinterpret_cast<float4*>(c.param + off) = p;
    *reinterpret_cast<float4*>(c.m + off) = m;
    *reinterpret_cast<float4*>(c.v + off) = v;
    if (c.param_bf16) {
        uint2 q;
        q.x = pack2bf(p.x, p.y);
        q.y = pack2bf(p.z, p.w);
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(c.param_bf16) + off) = q;
    }
    if (c.store_grad) *reinterpret_cast<float4*>(c.grad + off) = make_float4(g[0], g[1], g[2], g[3]);
}

// NB quads at once: every load of the batch is issued before the first store.  One quad at a time the compiler must keep
// quad j + 1's loads behind quad j's stores (same arrays, offsets it cannot tell apart), and vmcnt(0) then also drains those
// stores: a full memory round trip per quad, the matrix cores idle meanwhile.  Same arithmetic per element: bit-identical.
// Offsets are 32-bit ELEMENT offsets into the arenas (the host checks the arena against ADAM_QUADS_MAX_ELEMS): one VGPR per
// quad beside the three uniform base pointers, instead of three 64-bit addresses kept from the loads to the stores.
constexpr int64_t ADAM_QUADS_MAX_ELEMS = (int64_t)1 << 30;
template <int NB, class G>
__device__ __forceinline__ void adam_quads(const dmvae_adam_ctx& c, const unsigned (&off)[NB], G&& grad_of) {     // grad_of(b, g[4]): the gradient quad of batch item b
    const dmvae_state* st = reinterpret_cast<const dmvae_state*>(c.state);
    const float lr_t = st->lr_t;
    float4 p[NB], m[NB], v[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        p[b] = *reinterpret_cast<const float4*>(c.param + off[b]);
        m[b] = *reinterpret_cast<const float4*>(c.m + off[b]);
        v[b] = *reinterpret_cast<const float4*>(c.v + off[b]);
    }
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        float g[4];
        grad_of(b, g);
        float* pp = &p[b].x; float* mp = &m[b].x; float* vp = &v[b].x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (c.param_bf16 && !c.ieee) adam_elem<true>(pp[j], mp[j], vp[j], g[j], c.grad_scale, c.beta1, c.beta2, c.epsilon, lr_t);       // (wave-uniform: the arithmetic follows the mode, see adam_elem)
            else adam_elem<false>(pp[j], mp[j], vp[j], g[j], c.grad_scale, c.beta1, c.beta2, c.epsilon, lr_t);
        }
        *reinterpret_cast<float4*>(c.param + off[b]) = p[b];
        *reinterpret_cast<float4*>(c.m + off[b]) = m[b];
        *reinterpret_cast<float4*>(c.v + off[b]) = v[b];
        if (c.param_bf16) {
            uint2 q;
            q.x = pack2bf(p[b].x, p[b].y);
            q.y = pack2bf(p[b].z, p[b].w);
            *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(c.param_bf16) + off[b]) = q;
        }
        if (c.store_grad) *reinterpret_cast<float4*>(c.grad + off[b]) = make_float4(g[0], g[1], g[2], g[3]);
    }
}

// The same over NBATCH batches, software-pipelined: the loads of batches i + 1 .. i + DEPTH - 1 are issued BEFORE the stores of batch i.  vmcnt
// counts loads and stores in issue order, so a load that follows stores is only seen once those stores are acknowledged; issued ahead of
// them, the next batches' parameters arrive while this batch is computed and written.  DEPTH buffers of NB quads cost 12 DEPTH NB registers
// and keep (DEPTH - 1) NB quads of loads in flight while a batch is waited for: for one register budget, small batches on a deep ring hold
// more in flight than two large ones (2 x 4: 4 of 8 quads; 4 x 2: 6 of 8; 8 x 1: 7 of 8).  Same arithmetic per element in the same order
// per address: bit-identical for every (NB, DEPTH).
template <int NB, int NBATCH, int DEPTH = 2, class OFF, class G>
__device__ __forceinline__ void adam_pipelined(const dmvae_adam_ctx& c, OFF&& off_of, G&& grad_of) {      // off_of(i, b) -> element offset; grad_of(i, b, g[4])
    static_assert(DEPTH >= 2, "at least the batch in hand and the next one");
    const dmvae_state* st = reinterpret_cast<const dmvae_state*>(c.state);
    const float lr_t = st->lr_t;
    float4 p[DEPTH][NB], m[DEPTH][NB], v[DEPTH][NB];
    auto load = [&](int i, int buf) {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const unsigned o = off_of(i, b);
            p[buf][b] = *reinterpret_cast<const float4*>(c.param + o);
            m[buf][b] = *reinterpret_cast<const float4*>(c.m + o);
            v[buf][b] = *reinterpret_cast<const float4*>(c.v + o);
        }
    };
#pragma unroll
    for (int i = 0; i < DEPTH - 1; ++i)
        if (i < NBATCH) load(i, i);
#pragma unroll
    for (int i = 0; i < NBATCH; ++i) {
        const int buf = i % DEPTH;
        if (i + DEPTH - 1 < NBATCH) load(i + DEPTH - 1, (i + DEPTH - 1) % DEPTH);
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const unsigned o = off_of(i, b);
            float g[4];
            grad_of(i, b, g);
            float* pp = &p[buf][b].x; float* mp = &m[buf][b].x; float* vp = &v[buf][b].x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (c.param_bf16 && !c.ieee) adam_elem<true>(pp[j], mp[j], vp[j], g[j], c.grad_scale, c.beta1, c.beta2, c.epsilon, lr_t);       // (wave-uniform: the arithmetic follows the mode, see adam_elem)
                else adam_elem<false>(pp[j], mp[j], vp[j], g[j], c.grad_scale, c.beta1, c.beta2, c.epsilon, lr_t);
            }
            *reinterpret_cast<float4*>(c.param + o) = p[buf][b];
            *reinterpret_cast<float4*>(c.m + o) = m[buf][b];
            *reinterpret_cast<float4*>(c.v + o) = v[buf][b];
            if (c.param_bf16) {
                uint2 q;
                q.x = pack2bf(p[buf][b].x, p[buf][b].y);
                q.y = pack2bf(p[buf][b].z, p[buf][b].w);
                *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(c.param_bf16) + o) = q;
            }
            if (c.store_grad) *reinterpret_cast<float4*>(c.grad + o) = make_float4(g[0], g[1], g[2], g[3]);
        }
    }
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// XCD-aware order (speed only, never correctness): workgroups are dealt round-robin over the 8
// XCDs, so ids b and b+8 share an L2.  Of the workgroup ids [gstart, gend), XCD x owns those with
// id & 7 == x; give each XCD a CONTIGUOUS run of work items: returns the item index in
// [0, gend - gstart) of workgroup gid (a bijection for any range).
__device__ __forceinline__ int xcd_run_index(const int gid, const int gstart, const int gend) {
    const int xcd = gid & 7;
    int run0 = 0;                                        // items owned by the XCD labels below ours
    for (int y = 0; y < xcd; ++y) {
        const int first = gstart + ((y - gstart) & 7);
        run0 += first < gend ? ((gend - 1 - first) >> 3) + 1 : 0;
    }
    return run0 + ((gid - (gstart + ((xcd - gstart) & 7))) >> 3);
}

}  // namespace dmvae
