// Row-strip forward of TWO consecutive narrow dense layers as one kernel -- the prototype VERDICT r4 #7 asked for (measured, not estimated).
//     Y1 = relu(X . W0 + b0)   [B][512]        Y2 = relu(Y1 . W1 + b1)   [B][512]         tf.layers.dense x 2, code/base_models.py:220-226 (enc0 -> enc1), :280-289 (dec1 -> dec2)
// A workgroup owns 32 rows through both layers: Y1's 32 x 512 tile stays in LDS (bf16, as eight k-contiguous 32 x 64 tiles: the A operand of layer 2
// is read from there by the same ds_read_b128 the tiled kernels use) and is ALSO written to global memory (the backward pass gates on it and
// multiplies by it); both weight matrices are streamed from L2 in 64 x 256 K tiles (two column halves per layer: 4 passes), so a CU reads
// 2 (K0 + 512) x 512 bytes of weights per strip whatever the batch, where the tiled kernels (gemm_bf16.hip) read a 128-row activation panel per tile.
// Same MFMA, same operand order, same K order, same bias / ReLU / bf16 rounding as two dmvae_gemm launches with DMVAE_EPI_BIAS_RELU: same bits.
//
// NOT on the step path: a measurement kernel behind dmvae_debug_strip_fwd2 (include/dmvae_hip_debug.h), timed against the two launches it would
// replace by tools/strip2_probe.py.  Result and decision: profiles/r05_row_strip.txt, DESIGN_LOG R5.
#include <type_traits>

#include "gemm_tile.h"

namespace dmvae {

struct Strip2Args {
    const bf16_t* X; int64_t ldx; int K0;         // [B_pad][ldx], K0 (multiple of 64) columns used
    const bf16_t* W0; int64_t ld0; const float* b0;      // [K0][ld0 >= 512]
    const bf16_t* W1; int64_t ld1; const float* b1;      // [512][ld1 >= 512]
    bf16_t* Y1; int64_t ldy1;
    bf16_t* Y2; int64_t ldy2;
};

constexpr int S2_R = 32, S2_N = 512, S2_NP = 256;           // rows per strip, layer width, columns per pass
constexpr int S2_A = S2_R * BK, S2_WH = 128 * BK;            // elements: A tile (4 KiB), one 128-column half of a weight K tile (16 KiB)
constexpr int S2_STAGE = S2_A + 2 * S2_WH;                   // 36 KiB
constexpr int S2_NST = 3;
constexpr int S2_Y1 = S2_R * S2_N;                           // 32 KiB: Y1 as eight [32][64] k-contiguous tiles

__global__ __launch_bounds__(256) void strip_fwd2_kernel(Strip2Args a) {
    extern __shared__ __attribute__((aligned(16))) bf16_t s2_smem[];
    bf16_t* y1t = s2_smem;                                   // [8][32][64] (swz_kc)
    bf16_t* ring = s2_smem + S2_Y1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, g = lane >> 4;
    const int row0 = (int)blockIdx.x * S2_R;

    unsigned goA[1], goW0[4], goW1[4];
    stage_offsets<S2_R, true, 4, BK>(a.ldx, wave, lane, goA);
    stage_offsets<128, false, 4, BK>(a.ld0, wave, lane, goW0);
    stage_offsets<128, false, 4, BK>(a.ld1, wave, lane, goW1);
    // wave w multiplies columns [64 w, 64 w + 64) of the pass: half w >> 1, column (w & 1) * 64 + 16 j inside it; both 16-row tiles
    unsigned short foA[2][2], foB[2][4][2], unused;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int i = 0; i < 2; ++i) frag_offsets<S2_R, true>(i * 16, ks, lane, foA[ks][i], unused);
#pragma unroll
        for (int j = 0; j < 4; ++j) frag_offsets<128, false>((wave & 1) * 64 + j * 16, ks, lane, foB[ks][j][0], foB[ks][j][1]);
    }
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) bf16_t*)ring) + 1024u * (unsigned)wave);
    const bf16_t* Xg = a.X + (int64_t)row0 * a.ldx;

    struct Frag { bf16x8 a[2], b[4]; };
    f32x4 acc[2][4];

    // one pass = one 32 x 256 output block: LAYER 0 streams X tiles and W0 tiles, LAYER 1 streams W1 tiles and reads its A operand from y1t
    auto issue = [&](auto LAYER, int pass, int t, int slot) {
        constexpr int L = decltype(LAYER)::value;
        const unsigned s = lds_w + 2u * (unsigned)(slot * S2_STAGE);
        if constexpr (L == 0) {
            glds_tile(Xg + (int64_t)t * BK, goA, s, 4096u);
            const bf16_t* w = a.W0 + (int64_t)t * BK * a.ld0 + pass * S2_NP;
            glds_tile(w, goW0, s + 2u * S2_A, 4096u);
            glds_tile(w + 128, goW0, s + 2u * (S2_A + S2_WH), 4096u);
        } else {
            const bf16_t* w = a.W1 + (int64_t)t * BK * a.ld1 + pass * S2_NP;
            glds_tile(w, goW1, s + 2u * S2_A, 4096u);
            glds_tile(w + 128, goW1, s + 2u * (S2_A + S2_WH), 4096u);
        }
    };
    auto rd = [&](auto LAYER, int t, int slot, int ks, Frag& f) {
        constexpr int L = decltype(LAYER)::value;
        const bf16_t* st = ring + slot * S2_STAGE;
        const bf16_t* As = L == 0 ? st : y1t + t * S2_A;
        const bf16_t* Bs = st + S2_A + (wave >> 1) * S2_WH;
#pragma unroll
        for (int i = 0; i < 2; ++i) f.a[i] = read_frag<true>(As, foA[ks][i], 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) f.b[j] = read_frag<false>(Bs, foB[ks][j][0], foB[ks][j][1]);
    };
    auto mma = [&](const Frag& f) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.b[j], f.a[i], acc[i][j], 0, 0, 0);
    };
    auto prologue = [&](auto LAYER, int pass, int nk) {
#pragma unroll
        for (int t = 0; t < S2_NST; ++t)
            if (t < nk) issue(LAYER, pass, t, t);
    };
    auto kloop = [&](auto LAYER, int pass, int nk) {          // the pipeline of gemm_bf16_body; the pass's first tiles are already requested
        constexpr int L = decltype(LAYER)::value;
        constexpr int LOADS = L == 0 ? 9 : 8;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        Frag f0, f1;
        if (nk >= S2_NST) wait_vmcnt<LOADS*(S2_NST - 1)>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        rd(LAYER, 0, 0, 0, f0);
        for (int kt = 0; kt < nk; kt += S2_NST) {
#pragma unroll
            for (int s = 0; s < S2_NST; ++s) {
                const int t = kt + s;
                if (t < nk) {
                    rd(LAYER, t, s, 1, f1);
                    mma(f0);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (t + S2_NST <= nk) wait_vmcnt<LOADS*(S2_NST - 2)>();
                    else wait_vmcnt<0>();
                    __builtin_amdgcn_s_barrier();
                    if (t + S2_NST < nk) issue(LAYER, pass, t + S2_NST, s);
                    if (t + 1 < nk) rd(LAYER, t + 1, (s + 1) % S2_NST, 0, f0);
                    mma(f1);
                }
            }
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                        // the ring is idle: the next pass may request its first tiles
    };
    // bias + ReLU + bf16 of this wave's 32 x 64 block of the pass (DMVAE_EPI_BIAS_RELU's arithmetic); a lane owns 4 consecutive columns of one row
    auto finish = [&](const float* bias, int pass, auto&& sink) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = pass * S2_NP + wave * 64 + j * 16 + g * 4;
            float b[4];
            loadf4(bias, n, b);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(acc[i][j][e] + b[e], 0.f);
                uint2 q;
                q.x = pack2bf(v[0], v[1]); q.y = pack2bf(v[2], v[3]);
                sink(i * 16 + li, n, q);
            }
        }
    };
    const int nk0 = a.K0 / BK;
    using L0 = std::integral_constant<int, 0>;
    using L1 = std::integral_constant<int, 1>;
    auto to_y1 = [&](int m, int n, uint2 q) {                 // into the k-contiguous tile n / 64 (the layout frag_offsets<.., true> reads)
        const int c = (n & 63) >> 3;
        *reinterpret_cast<uint2*>(y1t + (n >> 6) * S2_A + m * 64 + swz_kc(m, c) * 8 + (n & 7)) = q;
    };
    auto to_y2 = [&](int m, int n, uint2 q) { *reinterpret_cast<uint2*>(a.Y2 + (int64_t)(row0 + m) * a.ldy2 + n) = q; };

    prologue(L0{}, 0, nk0);
    kloop(L0{}, 0, nk0);
    prologue(L0{}, 1, nk0);                                   // (requested before this pass's epilogue: the epilogue hides behind their latency)
    finish(a.b0, 0, to_y1);
    kloop(L0{}, 1, nk0);
    prologue(L1{}, 0, S2_N / BK);
    finish(a.b0, 1, to_y1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                             // Y1 is complete in LDS
    // Y1 to global memory, row-contiguous 16-byte chunks (1 KiB per row): 32 rows x 64 chunks over 256 threads
#pragma unroll
    for (int q = 0; q < (S2_R * S2_N / 8) / 256; ++q) {
        const int idx = q * 256 + tid;
        const int m = idx >> 6, ch = idx & 63;                // chunk ch of row m: tile ch >> 3, chunk ch & 7 inside it
        const uint4 v = *reinterpret_cast<const uint4*>(y1t + (ch >> 3) * S2_A + m * 64 + swz_kc(m, ch & 7) * 8);
        *reinterpret_cast<uint4*>(a.Y1 + (int64_t)(row0 + m) * a.ldy1 + ch * 8) = v;
    }
    kloop(L1{}, 0, S2_N / BK);
    prologue(L1{}, 1, S2_N / BK);
    finish(a.b1, 0, to_y2);
    kloop(L1{}, 1, S2_N / BK);
    finish(a.b1, 1, to_y2);
}

int strip_fwd2_launch(hipStream_t s, int B_pad, int K0, const void* X, int64_t ldx, const void* W0, int64_t ld0, const float* b0,
                      const void* W1, int64_t ld1, const float* b1, void* Y1, int64_t ldy1, void* Y2, int64_t ldy2) {
    if (B_pad % S2_R || K0 % BK || K0 < BK || ldx < K0 || ld0 < S2_N || ld1 < S2_N || ldy1 < S2_N || ldy2 < S2_N || (ldx | ld0 | ld1 | ldy1 | ldy2) % 8 || !X || !W0 || !W1 || !b0 || !b1 || !Y1 || !Y2) {
        set_error("strip_fwd2: B_pad %% 32, K0 %% 64, two 512-wide layers, leading dimensions multiples of 8");
        return DMVAE_EINVAL;
    }
    Strip2Args a;
    a.X = reinterpret_cast<const bf16_t*>(X); a.ldx = ldx; a.K0 = K0;
    a.W0 = reinterpret_cast<const bf16_t*>(W0); a.ld0 = ld0; a.b0 = b0;
    a.W1 = reinterpret_cast<const bf16_t*>(W1); a.ld1 = ld1; a.b1 = b1;
    a.Y1 = reinterpret_cast<bf16_t*>(Y1); a.ldy1 = ldy1; a.Y2 = reinterpret_cast<bf16_t*>(Y2); a.ldy2 = ldy2;
    const size_t lds = (size_t)(S2_Y1 + S2_NST * S2_STAGE) * 2;          // 32 + 108 KiB
    static bool set = false;
    if (!set) { (void)hipFuncSetAttribute((const void*)strip_fwd2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); set = true; }
    ProfScope ps(s, "strip_fwd2", 2.0 * B_pad * (double)S2_N * (K0 + S2_N), 2.0 * ((double)B_pad * (K0 + 2 * S2_N) + (double)(K0 + S2_N) * S2_N));
    DMVAE_LAUNCH(strip_fwd2_kernel, dim3(B_pad / S2_R), dim3(256), lds, s, a);
    return check_launch("strip_fwd2");
}

}  // namespace dmvae
