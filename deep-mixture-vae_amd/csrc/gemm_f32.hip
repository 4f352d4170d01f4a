// Exact-f32 MFMA GEMM (v_mfma_f32_16x16x4_f32: bit-for-bit a k-ordered fmaf
// chain) -- the PARITY-mode matmul: same layouts and fused epilogues as
// gemm_bf16.hip, float operands and float activations.  Used to hold the
// "ELBO within 1e-3 of the reference" bar, which bf16 operands cannot.
//
// 64x64 tile, BK = 16, 4 waves (2 x 2) of 32x32.  Both operand tiles live in
// LDS as [k][i] (row stride 80 floats: the two k rows of a 32-lane half land
// 16 banks apart -> conflict-free ds_read_b32); a k-contiguous operand is
// transposed on the way into LDS.
#include "kernels.h"

namespace dmvae {

constexpr int FBM = 64, FBN = 64, FBK = 16, FLD = 80;

template <bool KC>
__device__ __forceinline__ float4 f32_stage_load(const float* __restrict__ g, int64_t ld, int tid) {
    if constexpr (KC) {   // tile [64 rows][16 k]: thread -> (row = tid>>2, k = 4*(tid&3))
        return *reinterpret_cast<const float4*>(g + (int64_t)(tid >> 2) * ld + (tid & 3) * 4);
    } else {              // tile [16 k][64 cols]: thread -> (k = tid>>4, col = 4*(tid&15))
        return *reinterpret_cast<const float4*>(g + (int64_t)(tid >> 4) * ld + (tid & 15) * 4);
    }
}
template <bool KC>
__device__ __forceinline__ void f32_stage_store(float* s, const float4& r, int tid) {
    if constexpr (KC) {
        const int row = tid >> 2, k = (tid & 3) * 4;
        s[(k + 0) * FLD + row] = r.x;
        s[(k + 1) * FLD + row] = r.y;
        s[(k + 2) * FLD + row] = r.z;
        s[(k + 3) * FLD + row] = r.w;
    } else {
        *reinterpret_cast<float4*>(s + (tid >> 4) * FLD + (tid & 15) * 4) = r;
    }
}

// one output tile (bx) of K slice by of problem a; smem = 4 * FBK * FLD floats
template <int LAYOUT, int EPI>
__device__ __forceinline__ void gemm_f32_body(const GemmArgs& a, const int bx, const int by, float* smem) {
    constexpr bool A_KC = (LAYOUT != DMVAE_GEMM_DW);
    constexpr bool B_KC = (LAYOUT == DMVAE_GEMM_DX);
    constexpr int T = FBK * FLD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, g = lane >> 4;
    const int tiles_n = a.N / FBN;
    const int m0 = (bx / tiles_n) * FBM, n0 = (bx % tiles_n) * FBN;
    const int kbeg = by * a.k_split;
    const int nk = a.k_split / FBK;

    const float* Ag = reinterpret_cast<const float*>(a.A);
    const float* Bg = reinterpret_cast<const float*>(a.B);
    if (!A_KC && a.conv_c) {   // conv-mode weight gradient (gemm_epilogue.h): column m = (tap, channel); this thread's columns
        const int m = m0 + (tid & 15) * 4, tap = min(m / a.conv_c, 8);   // (rows past the ninth tap are padding, zeroed by the caller)
        Ag += (int64_t)kbeg * a.lda + conv_tap_offset(tap, m % a.conv_c, a.conv_p, a.lda) - (tid & 15) * 4;
    } else
        Ag += A_KC ? ((int64_t)m0 * a.lda + kbeg) : ((int64_t)kbeg * a.lda + m0);
    Bg += B_KC ? ((int64_t)n0 * a.ldb + kbeg) : ((int64_t)kbeg * a.ldb + n0);
    const int64_t stepA = A_KC ? (int64_t)FBK : (int64_t)FBK * a.lda;
    const int64_t stepB = B_KC ? (int64_t)FBK : (int64_t)FBK * a.ldb;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr bool DW = (LAYOUT == DMVAE_GEMM_DW);
    const bool do_bias = DW && a.epi.out2 != nullptr && m0 == 0;   // bias gradient db[n] = sum_k dY[k][n], see gemm_bf16.hip
    f32x4 bacc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    float4 ra = f32_stage_load<A_KC>(Ag, a.lda, tid);
    float4 rb = f32_stage_load<B_KC>(Bg, a.ldb, tid);
    f32_stage_store<A_KC>(smem, ra, tid);
    f32_stage_store<B_KC>(smem + T, rb, tid);
    __syncthreads();

    auto compute = [&](int buf) {
        const float* As = smem + buf * 2 * T;
        const float* Bs = As + T;
#pragma unroll
        for (int ks = 0; ks < FBK / 4; ++ks) {
            float av[2], bv[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) av[i] = As[(ks * 4 + g) * FLD + wm * 32 + i * 16 + li];
#pragma unroll
            for (int j = 0; j < 2; ++j) bv[j] = Bs[(ks * 4 + g) * FLD + wn * 32 + j * 16 + li];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j], av[i], acc[i][j], 0, 0, 0);
            if constexpr (DW) {
                if (do_bias) {
#pragma unroll
                    for (int j = 0; j < 2; ++j) bacc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j], 1.0f, bacc[j], 0, 0, 0);
                }
            }
        }
    };
    // conv mode (GemmArgs::conv_c, see gemm_epilogue.h): the k-contiguous A operand is an implicit patch
    // matrix -- K tile t reads the same rows shifted by its tap; (tap, channel) is a running counter
    const float* const Ag0 = Ag;
    int cv_tap = 0, cv_c0 = 0;
    if constexpr (A_KC) {
        if (a.conv_c) {
            Ag = Ag0 + conv_tap_offset(0, 0, a.conv_p, a.lda);
            ra = f32_stage_load<A_KC>(Ag, a.lda, tid);
            f32_stage_store<A_KC>(smem, ra, tid);
            __syncthreads();
        }
    }
    for (int kt = 0; kt < nk - 1; ++kt) {   // branch-free steady state, last step peeled
        Ag += stepA; Bg += stepB;
        if constexpr (A_KC) {
            if (a.conv_c) {
                cv_c0 += FBK;
                if (cv_c0 >= a.conv_c) { cv_c0 = 0; ++cv_tap; }
                Ag = Ag0 + conv_tap_offset(min(cv_tap, 8), cv_c0, a.conv_p, a.lda);     // past the ninth tap: padding columns, zero weights
            }
        }
        ra = f32_stage_load<A_KC>(Ag, a.lda, tid);
        rb = f32_stage_load<B_KC>(Bg, a.ldb, tid);
        compute(kt & 1);
        float* nxt = smem + ((kt + 1) & 1) * 2 * T;
        f32_stage_store<A_KC>(nxt, ra, tid);
        f32_stage_store<B_KC>(nxt + T, rb, tid);
        __syncthreads();
    }
    compute((nk - 1) & 1);
    __syncthreads();

    float loss = 0.f;
    dmvae_epilogue epi = a.epi;
    if constexpr (EPI == DMVAE_EPI_STORE_F32) {      // K-slice slabs
        epi.out = reinterpret_cast<float*>(epi.out) + (int64_t)by * a.slab_stride;
        if (epi.out2) epi.out2 = reinterpret_cast<float*>(epi.out2) + (int64_t)by * a.slab_stride2;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + wm * 32 + i * 16 + li;
            const int n = n0 + wn * 32 + j * 16 + g * 4;
            float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
            if (a.conv_c && n >= a.epi.n_valid) {              // conv mode: a 32-channel activation is stored 32 wide
                if (!(EPI == DMVAE_EPI_STORE_F32 && a.slab_stride)) continue;
                v[0] = v[1] = v[2] = v[3] = 0.f;               // (slabs are summed whole: their pad columns must hold zeros)
            }
            epilogue_quad<EPI, float>(epi, m, n, v, loss);
        }
    if constexpr (DW) {
        if (do_bias && wm == 0 && li == 0) {
            float* db = reinterpret_cast<float*>(epi.out2);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wn * 32 + j * 16 + g * 4;
                if (a.conv_c && n >= a.epi.n_valid) {
                    if (EPI == DMVAE_EPI_STORE_F32 && a.slab_stride2) *reinterpret_cast<float4*>(db + n) = make_float4(0.f, 0.f, 0.f, 0.f);
                    continue;
                }
                if constexpr (EPI == DMVAE_EPI_ATOMIC_F32) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) atomicAdd(db + n + e, bacc[j][e]);
                } else {
                    *reinterpret_cast<float4*>(db + n) = make_float4(bacc[j][0], bacc[j][1], bacc[j][2], bacc[j][3]);
                }
            }
        }
    }
    if constexpr (EPI == DMVAE_EPI_BIAS_RECON) {
        const float t = block_sum_256(loss, smem);
        if (tid == 0) a.epi.partials[bx] = t;
    }
}

template <int LAYOUT, int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float smem[4 * FBK * FLD];
    gemm_f32_body<LAYOUT, EPI>(a, (int)blockIdx.x, (int)blockIdx.y, smem);
}

// The three contractions of the latent stage's MFMA form (latent_mfma.hip: G1 forward layout, G2 dX layout, G3 dW layout; all
// STORE_F32, all reading only what latent_pre wrote) as ONE grid: workgroups [0, n1) = G1, [n1, n1 + n2) = G2, the rest G3; inside a
// problem the id is slice-major (K slice = id / tiles).  Three launches of 6-25 us each were boundary-bound.
struct F32Trio { GemmArgs p[3]; int start[4]; int tiles[3]; };
__global__ __launch_bounds__(256) void gemm_f32_trio_kernel(F32Trio t) {
    __shared__ __attribute__((aligned(16))) float smem[4 * FBK * FLD];
    const int b = (int)blockIdx.x;
    if (b < t.start[1]) gemm_f32_body<DMVAE_GEMM_FWD, DMVAE_EPI_STORE_F32>(t.p[0], b % t.tiles[0], b / t.tiles[0], smem);
    else if (b < t.start[2]) gemm_f32_body<DMVAE_GEMM_DX, DMVAE_EPI_STORE_F32>(t.p[1], (b - t.start[1]) % t.tiles[1], (b - t.start[1]) / t.tiles[1], smem);
    else gemm_f32_body<DMVAE_GEMM_DW, DMVAE_EPI_STORE_F32>(t.p[2], (b - t.start[2]) % t.tiles[2], (b - t.start[2]) / t.tiles[2], smem);
}
int gemm_f32_trio(hipStream_t s, const GemmArgs& g1, int split1, const GemmArgs& g2, int split2, const GemmArgs& g3, int split3) {
    F32Trio t;
    const GemmArgs* g[3] = {&g1, &g2, &g3};
    const int sp[3] = {split1, split2, split3};
    int total = 0;
    for (int i = 0; i < 3; ++i) {
        if (g[i]->epi.kind != DMVAE_EPI_STORE_F32 || g[i]->M % FBM || g[i]->N % FBN || sp[i] < 1 || g[i]->K % (sp[i] * FBK) || g[i]->conv_c || g[i]->epi.out2) {
            set_error("gemm_f32_trio: STORE_F32 problems on the 64 x 64 x %d tile grid only", FBK); return DMVAE_EINVAL;
        }
        t.p[i] = *g[i];
        t.p[i].k_split = g[i]->K / sp[i];
        t.tiles[i] = (g[i]->M / FBM) * (g[i]->N / FBN);
        t.start[i] = total;
        total += t.tiles[i] * sp[i];
    }
    t.start[3] = total;
    DMVAE_LAUNCH(gemm_f32_trio_kernel, dim3(total), dim3(256), 0, s, t);
    return check_launch("gemm_f32_trio");
}

template <int LAYOUT, int EPI>
static int launch(hipStream_t s, const GemmArgs& a, int split) {
    dim3 grid((a.M / FBM) * (a.N / FBN), split);
    DMVAE_LAUNCH((gemm_f32_kernel<LAYOUT, EPI>), grid, dim3(256), 0, s, a);
    return check_launch("gemm_f32");
}

int gemm_f32_dispatch(hipStream_t s, int layout, const GemmArgs& a, int split) {
    const int epi = a.epi.kind;
#define CASE(L, E) \
    if (layout == L && epi == E) return launch<L, E>(s, a, split);
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
    set_error("dmvae_gemm(f32): layout %d with epilogue %d is not instantiated", layout, epi);
    return DMVAE_EUNSUPPORTED;
}

}  // namespace dmvae
