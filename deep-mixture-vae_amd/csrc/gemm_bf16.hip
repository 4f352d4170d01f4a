// bf16 MFMA GEMM for gfx950 (v_mfma_f32_16x16x32_bf16, fp32 accumulate) with
// fused epilogues.  Replaces the tf.matmul / tf.layers.dense forward ops of
// code/base_models.py:221-248,279-293 and their tf.gradients (dX, dW).
//
// One kernel template, three operand layouts (include/dmvae_hip.h):
//   FWD: A [M][K] k-contiguous,  B = W [K][N] n-contiguous
//   DX : A [M][K] k-contiguous,  B = W [N][K] k-contiguous
//   DW : A = X [K][M] m-contiguous, B = dY [K][N] n-contiguous (K = batch)
// Tiles are copied global -> LDS in the orientation they have in memory
// (16-byte loads along the contiguous dimension).  A k-contiguous operand is
// read into its MFMA fragment with one ds_read_b128; an m/n-contiguous operand
// with two ds_read_b64_tr_b16 (gfx950 transposing LDS read), so no transposed
// copy of weights, activations or gradients ever exists in HBM.
//
// Block = 256 threads = 4 waves (2 x 2), BK = 64, register-staged double
// buffering: the global loads of tile t+1 are in flight while tile t is
// multiplied, one barrier per K step.
#include "kernels.h"

namespace dmvae {

constexpr int BK = 64;
constexpr int LDK = BK + 8;     // k-contiguous tile: row stride 144 B
constexpr int TRPAD = 16;       // m/n-contiguous tile: row stride (C + 16) * 2 B

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int R, bool KC>
__device__ __forceinline__ void stage_load(const bf16_t* __restrict__ g, int64_t ld, uint4 (&r)[R / 32], int tid) {
#pragma unroll
    for (int i = 0; i < R / 32; ++i) {
        const int c = tid + i * 256;
        int row, cc;
        if constexpr (KC) { row = c >> 3; cc = c & 7; }
        else { row = c / (R / 8); cc = c % (R / 8); }
        r[i] = *reinterpret_cast<const uint4*>(g + (int64_t)row * ld + cc * 8);
    }
}
template <int R, bool KC>
__device__ __forceinline__ void stage_store(bf16_t* s, const uint4 (&r)[R / 32], int tid) {
#pragma unroll
    for (int i = 0; i < R / 32; ++i) {
        const int c = tid + i * 256;
        int row, cc;
        if constexpr (KC) { row = c >> 3; cc = c & 7; }
        else { row = c / (R / 8); cc = c % (R / 8); }
        constexpr int LD = KC ? LDK : (R + TRPAD);
        *reinterpret_cast<uint4*>(s + row * LD + cc * 8) = r[i];
    }
}

// fragment for index i0 + (lane & 15), k = ks*32 + 8*(lane >> 4) .. +7
template <int R, bool KC>
__device__ __forceinline__ bf16x8 read_frag(const bf16_t* s, int i0, int ks, int lane) {
    const int li = lane & 15, g = lane >> 4;
    if constexpr (KC) {
        const s16x8 v = *reinterpret_cast<const s16x8*>(s + (i0 + li) * LDK + ks * 32 + g * 8);
        return __builtin_bit_cast(bf16x8, v);
    } else {
        constexpr int LD = R + TRPAD;
        const int q = li >> 2, p = li & 3;
        const bf16_t* a0 = s + (ks * 32 + g * 8 + q) * LD + i0 + p * 4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * LD));
        const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <int BM, int BN, int LAYOUT, int EPI>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmArgs a) {
    constexpr bool A_KC = (LAYOUT != DMVAE_GEMM_DW);
    constexpr bool B_KC = (LAYOUT == DMVAE_GEMM_DX);
    constexpr int A_ELEMS = A_KC ? BM * LDK : BK * (BM + TRPAD);
    constexpr int B_ELEMS = B_KC ? BN * LDK : BK * (BN + TRPAD);
    constexpr int TM = BM / 32, TN = BN / 32;   // 16x16 tiles per wave (wave tile = BM/2 x BN/2)
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * (A_ELEMS + B_ELEMS)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = a.N / BN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int kbeg = blockIdx.y * a.k_split;
    const int nk = a.k_split / BK;

    const bf16_t* Ag = reinterpret_cast<const bf16_t*>(a.A);
    const bf16_t* Bg = reinterpret_cast<const bf16_t*>(a.B);
    Ag += A_KC ? ((int64_t)m0 * a.lda + kbeg) : ((int64_t)kbeg * a.lda + m0);
    Bg += B_KC ? ((int64_t)n0 * a.ldb + kbeg) : ((int64_t)kbeg * a.ldb + n0);
    const int64_t stepA = A_KC ? (int64_t)BK : (int64_t)BK * a.lda;
    const int64_t stepB = B_KC ? (int64_t)BK : (int64_t)BK * a.ldb;

    f32x4 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 ra[BM / 32], rb[BN / 32];
    stage_load<BM, A_KC>(Ag, a.lda, ra, tid);
    stage_load<BN, B_KC>(Bg, a.ldb, rb, tid);
    stage_store<BM, A_KC>(smem, ra, tid);
    stage_store<BN, B_KC>(smem + A_ELEMS, rb, tid);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const bf16_t* As = smem + (kt & 1) * (A_ELEMS + B_ELEMS);
        const bf16_t* Bs = As + A_ELEMS;
        const bool more = kt + 1 < nk;
        if (more) {
            Ag += stepA; Bg += stepB;
            stage_load<BM, A_KC>(Ag, a.lda, ra, tid);
            stage_load<BN, B_KC>(Bg, a.ldb, rb, tid);
        }
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            bf16x8 af[TM], bfr[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = read_frag<BM, A_KC>(As, wm * (BM / 2) + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) bfr[j] = read_frag<BN, B_KC>(Bs, wn * (BN / 2) + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    // operands swapped: D[row = n][col = m] -> each lane owns 4 consecutive n of one m
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        if (more) {
            bf16_t* nxt = smem + ((kt + 1) & 1) * (A_ELEMS + B_ELEMS);
            stage_store<BM, A_KC>(nxt, ra, tid);
            stage_store<BN, B_KC>(nxt + A_ELEMS, rb, tid);
        }
        __syncthreads();
    }

    float loss = 0.f;
    const int li = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int m = m0 + wm * (BM / 2) + i * 16 + li;
            const int n = n0 + wn * (BN / 2) + j * 16 + g * 4;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            epilogue_quad<EPI, bf16_t>(a.epi, m, n, v, loss);
        }
    if constexpr (EPI == DMVAE_EPI_BIAS_RECON) {
        float* red = reinterpret_cast<float*>(smem);   // all waves are past the last barrier of the K loop
        const float t = block_sum_256(loss, red);
        if (tid == 0) a.epi.partials[blockIdx.x] = t;
    }
}

// ---------------------------------------------------------------- host side
template <int BM, int BN, int LAYOUT, int EPI>
static int launch(hipStream_t s, const GemmArgs& a, int split) {
    dim3 grid((a.M / BM) * (a.N / BN), split);
    hipLaunchKernelGGL((gemm_bf16_kernel<BM, BN, LAYOUT, EPI>), grid, dim3(256), 0, s, a);
    return check_launch("gemm_bf16");
}

template <int LAYOUT, int EPI>
static int launch_tiled(hipStream_t s, const GemmArgs& a, int split) {
    // Largest tile that still yields >= ~1.5 workgroups per CU; M, N multiples of 64 guaranteed.
    const bool m128 = a.M % 128 == 0, n128 = a.N % 128 == 0;
    auto wgs = [&](int bm, int bn) { return (long)(a.M / bm) * (a.N / bn) * split; };
    if (m128 && n128 && wgs(128, 128) >= 384) return launch<128, 128, LAYOUT, EPI>(s, a, split);
    if (m128 && wgs(128, 64) >= 384) return launch<128, 64, LAYOUT, EPI>(s, a, split);
    if (n128 && wgs(64, 128) >= 384) return launch<64, 128, LAYOUT, EPI>(s, a, split);
    return launch<64, 64, LAYOUT, EPI>(s, a, split);
}

int gemm_bf16_dispatch(hipStream_t s, int layout, const GemmArgs& a, int split) {
    const int epi = a.epi.kind;
#define CASE(L, E) \
    if (layout == L && epi == E) return launch_tiled<L, E>(s, a, split);
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

int gemm_bf16_tile_m(int M, int N, int split) {   // BM the dispatcher will pick (for partial counts)
    const bool m128 = M % 128 == 0, n128 = N % 128 == 0;
    auto wgs = [&](int bm, int bn) { return (long)(M / bm) * (N / bn) * split; };
    if (m128 && n128 && wgs(128, 128) >= 384) return 128 * 1000 + 128;
    if (m128 && wgs(128, 64) >= 384) return 128 * 1000 + 64;
    if (n128 && wgs(64, 128) >= 384) return 64 * 1000 + 128;
    return 64 * 1000 + 64;
}

}  // namespace dmvae
