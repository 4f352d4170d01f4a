// CNN encoder trunk of the checked-in DeepMixtureVAE (base_models.py:176-216; Convolution /
// MaxPooling at includes/layers.py:39-77): the kernels around the GEMMs.
//
// A 3x3 SAME stride-1 convolution in NHWC / HWIO is the GEMM  [pixels][9*Cin] x [9*Cin][Cout]; its
// weight gradient is X_patches^T dY and its input gradient is the same convolution of dY with the
// flipped, channel-transposed kernel.  The patch matrix is NEVER FORMED (the one-channel first layer is
// a direct kernel, 9 multiply-adds per output): activations live in HBM with a zero border, [B][P][P][C], P = side + 2, plus P + 1
// zero guard rows on either end, and the GEMMs run over the PADDED pixel index space:
//   * row m of the patch matrix, tap (ky,kx), is row m + (ky-1)*P + (kx-1) of the activation
//     itself -- a wave-uniform pointer offset per K tile (GemmArgs::conv_c, gemm_epilogue.h); the
//     zero border supplies the SAME padding; rows that ARE border pixels compute garbage and are
//     re-zeroed (zero_border) or come out zero by themselves (ReLU gate of a zero activation);
//   * the weight gradient is ONE DW-layout GEMM with M = (tap, channel): a tile row reads the
//     activation shifted by its tap's row offset, against dY over all padded rows (dY's border
//     rows are zero); split-K with fp32 atomics.
// The cost is the border (15 % more rows at 28x28, 65 % at 7x7); what it buys is that no patch matrix
// (9x the activation bytes) is written or read.  K counts the REAL channels: a 32-channel layer (stored
// with 32 zero pad channels) puts two taps into one 64-wide K tile.
//
// T is the activation type of the plan (float in parity mode, bf16 otherwise); an activation is stored
// exactly as wide as its channel count (32, 64, 128): a GEMM that produces 32 channels runs a 64-wide N tile
// and drops the other columns in its epilogue (dmvae_epilogue::n_valid).
#include <algorithm>
#include "common.h"
#include "kernels.h"

namespace dmvae {

template <typename T> struct Vec16;                       // 16-byte vector of T
template <> struct Vec16<float> { typedef float4 type; static constexpr int N = 4; };
template <> struct Vec16<bf16_t> { typedef uint4 type; static constexpr int N = 8; };

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16_t v) { return bf2f(v); }

// 16 bytes of channels as scalars
template <typename T> struct Lanes {
    typename Vec16<T>::type v;
    __device__ __forceinline__ T& operator[](int i) { return reinterpret_cast<T*>(&v)[i]; }
};

// 8 consecutive channels <-> 8 floats
__device__ __forceinline__ void store8(float* p, const float v[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void store8(bf16_t* p, const float v[8]) {
    uint4 q;
    q.x = pack2bf(v[0], v[1]); q.y = pack2bf(v[2], v[3]); q.z = pack2bf(v[4], v[5]); q.w = pack2bf(v[6], v[7]);
    *reinterpret_cast<uint4*>(p) = q;
}
__device__ __forceinline__ void load8(const float* p, float v[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void load8(const bf16_t* p, float v[8]) {
    const uint4 q = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u);
    v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
    v[4] = __uint_as_float(q.z << 16); v[5] = __uint_as_float(q.z & 0xffff0000u);
    v[6] = __uint_as_float(q.w << 16); v[7] = __uint_as_float(q.w & 0xffff0000u);
}

// First layer (ONE input channel, 32 outputs, base_models.py:182): 9 multiply-adds per output -- no GEMM.  Direct
// form: out[b, y, x, co] = relu(b[co] + sum_t x[b, y+ty-1, x+tx-1] * W[t][co]), written into the interior of the
// zero-bordered activation (its border and guard rows are zero from allocation and nothing ever writes them).
// x = the batch as loaded, [B][bstride] with the H*H pixels of an image contiguous; W = [9][ldw], fp32 bias.
// Thread = (4 consecutive pixels of an image row, 8 channels): one 3 x 6 input window, 72 weights and 8 biases in
// registers; H is a multiple of 4.
template <typename T>
__device__ __forceinline__ void first_window(const T* __restrict__ img, int H, int y, int x0, float (&xv)[3][6]) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = y + dy - 1;
#pragma unroll
        for (int dx = 0; dx < 6; ++dx) {
            const int xx = x0 + dx - 1;
            xv[dy][dx] = (yy >= 0 && yy < H && xx >= 0 && xx < H) ? to_f(img[yy * H + xx]) : 0.f;
        }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void conv_first_fwd_kernel(const T* __restrict__ x, int64_t bstride, int H, int n_units, const T* __restrict__ W, int ldw,
                                                             const float* __restrict__ bias, T* __restrict__ out, int ld) {
    const int c = (threadIdx.x & 3) * 8;
    float w[9][8], bv[8];
#pragma unroll
    for (int t = 0; t < 9; ++t) load8(W + t * ldw + c, w[t]);
#pragma unroll
    for (int j = 0; j < 8; ++j) bv[j] = bias[c + j];
    const int P = H + 2, S = H / 4, U = H * S;
    for (int u = blockIdx.x * 64 + (threadIdx.x >> 2); u < n_units; u += gridDim.x * 64) {
        const int b = u / U, r = u - b * U, y = r / S, x0 = (r - y * S) * 4;
        float xv[3][6];
        first_window(x + (int64_t)b * bstride, H, y, x0, xv);
        T* o = out + ((int64_t)b * P * P + (int64_t)(y + 1) * P + x0 + 1) * ld + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = bv[j];
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv[t / 3][i + t % 3], w[t][j], acc[j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], 0.f);
            store8(o + (int64_t)i * ld, acc);
        }
    }
}

// ... and its weight / bias gradient: dW[t][co] = sum over pixels x[pix + tap t] * dY[pix][co], db = sum dY.
// A block walks a slab of units (4 pixels of a row each): thread = (unit lane, 8 channels) with 9 + 1 accumulators
// x 8 channels in registers; reduced over the 16 unit lanes of a wave by shuffles, over the 4 waves through LDS,
// then written as the block's partial [10][32]; a second tiny kernel adds the blocks in a fixed order, so this
// layer's gradient is deterministic (no atomics).
template <typename T>
__global__ __launch_bounds__(256) void conv_first_dw_kernel(const T* __restrict__ x, int64_t bstride, int H, int n_units, const T* __restrict__ dY,
                                                            float* __restrict__ part, int units_per_block, int ld) {
    const int P = H + 2, S = H / 4, U = H * S;
    const int cq = threadIdx.x & 3, c = cq * 8, lane = threadIdx.x >> 2;      // 64 unit lanes x 4 channel octets
    float acc[10][8];
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    const int u_begin = blockIdx.x * units_per_block, u_end = min(n_units, u_begin + units_per_block);
    for (int u = u_begin + lane; u < u_end; u += 64) {
        const int b = u / U, r = u - b * U, y = r / S, x0 = (r - y * S) * 4;
        float xv[3][6];
        first_window(x + (int64_t)b * bstride, H, y, x0, xv);
        const T* gp = dY + ((int64_t)b * P * P + (int64_t)(y + 1) * P + x0 + 1) * ld + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float g[8];
            load8(gp + (int64_t)i * ld, g);
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[t][j] = fmaf(xv[t / 3][i + t % 3], g[j], acc[t][j]);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[9][j] += g[j];
        }
    }
    __shared__ float red[4][4][80];
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 10; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[t][j];
#pragma unroll
            for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);      // over the 16 unit lanes of this wave (lane bits 2..5)
            if ((threadIdx.x & 63) < 4) red[wave][cq][t * 8 + j] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * 80; i += 256) {
        const int q = i / 80, v = i % 80, t = v >> 3, co = q * 8 + (v & 7);
        part[(int64_t)blockIdx.x * 320 + t * 32 + co] = red[0][q][v] + red[1][q][v] + red[2][q][v] + red[3][q][v];
    }
}
// second stage: dW[t][co] / db[co] = sum over the blocks' partials [nblk][10][32] in a fixed order (deterministic).
// Block t (10 of them) owns the 32 outputs of tap t (t = 9: the bias); its 1024 threads = 32 row lanes x 32 outputs.
__global__ __launch_bounds__(1024) void conv_first_dw_reduce_kernel(const float* __restrict__ part, int nblk, float* __restrict__ dW, int ldw, float* __restrict__ db) {
    __shared__ float red[32][33];
    const int t = blockIdx.x, co = threadIdx.x & 31, rl = threadIdx.x >> 5;
    float s0 = 0.f, s1 = 0.f;
    int b = rl;
    for (; b + 32 < nblk; b += 64) {
        s0 += part[(int64_t)b * 320 + t * 32 + co];
        s1 += part[(int64_t)(b + 32) * 320 + t * 32 + co];
    }
    if (b < nblk) s0 += part[(int64_t)b * 320 + t * 32 + co];
    red[rl][co] = s0 + s1;
    __syncthreads();
    if (rl == 0) {
        float sum = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) sum += red[r][co];
        if (t < 9) dW[t * ldw + co] = sum; else db[co] = sum;
    }
}

// zero the border pixels of [B][P][P][ld] (4P - 4 pixels per image)
template <typename T>
__global__ __launch_bounds__(256) void zero_border_kernel(T* __restrict__ a, int P, int ld, int64_t n_img) {
    constexpr int V = Vec16<T>::N;
    const int nbp = 4 * P - 4, cv = ld / V;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_img * nbp * cv; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % cv) * V;
        const int64_t q = e / cv;
        const int j = (int)(q % nbp);
        const int64_t b = q / nbp;
        int yy, xx;
        if (j < P) { yy = 0; xx = j; }
        else if (j < 2 * P) { yy = P - 1; xx = j - P; }
        else { const int t = j - 2 * P; yy = 1 + (t >> 1); xx = (t & 1) ? P - 1 : 0; }
        typename Vec16<T>::type z = {};
        *reinterpret_cast<typename Vec16<T>::type*>(a + ((b * P + yy) * P + xx) * ld + c) = z;
    }
}

// tf.nn.max_pool ksize 2, strides 2, padding SAME: out = ceil(H/2); the pad (bottom / right, odd H) never
// wins.  in: zero-bordered [B][H+2][H+2][ld]; out: zero-bordered [B][Ho+2][Ho+2][ld] (out_border) or plain
// [B][Ho][Ho][ld] (the flattened trunk output).  One thread = one window x 16 bytes of channels.
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ in, int H, int ld, int64_t n_win, T* __restrict__ out, int out_border) {
    constexpr int V = Vec16<T>::N;
    const int Ho = (H + 1) / 2, P = H + 2, Po = Ho + 2 * out_border, cv = ld / V;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_win * cv; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % cv) * V;
        const int64_t q = e / cv;
        const int xo = (int)(q % Ho), yo = (int)((q / Ho) % Ho);
        const int64_t b = q / ((int64_t)Ho * Ho);
        const T* base = in + (b * P * P) * ld + c;
        Lanes<T> best;
        best.v = *reinterpret_cast<const typename Vec16<T>::type*>(base + ((int64_t)(2 * yo + 1) * P + 2 * xo + 1) * ld);
#pragma unroll
        for (int t = 1; t < 4; ++t) {
            const int y = 2 * yo + (t >> 1), x = 2 * xo + (t & 1);
            if (y < H && x < H) {
                Lanes<T> v;
                v.v = *reinterpret_cast<const typename Vec16<T>::type*>(base + ((int64_t)(y + 1) * P + x + 1) * ld);
#pragma unroll
                for (int j = 0; j < V; ++j)
                    if (to_f(v[j]) > to_f(best[j])) best[j] = v[j];
            }
        }
        *reinterpret_cast<typename Vec16<T>::type*>(out + ((b * Po + yo + out_border) * Po + xo + out_border) * ld + c) = best.v;
    }
}

// Gradient of ReLU -> max-pool: a pixel receives its window's gradient iff it is the FIRST maximum of the
// window in row-major order (TF's MaxPoolGrad) and its own value is positive (the ReLU in front of the
// pool).  One thread = one window x 16 bytes of channels: reads the window once, writes its four pixels
// (interior pixels of the zero-bordered din; the border is never written and stays zero).
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_relu_kernel(const T* __restrict__ in, const T* __restrict__ dout, int H, int ld,
                                                                int64_t n_win, T* __restrict__ din, int dout_border) {
    constexpr int V = Vec16<T>::N;
    const int Ho = (H + 1) / 2, P = H + 2, Po = Ho + 2 * dout_border, cv = ld / V;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_win * cv; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % cv) * V;
        const int64_t q = e / cv;
        const int xo = (int)(q % Ho), yo = (int)((q / Ho) % Ho);
        const int64_t b = q / ((int64_t)Ho * Ho);
        const int64_t img = (b * P * P) * ld + c;
        Lanes<T> x[4], g;
        g.v = *reinterpret_cast<const typename Vec16<T>::type*>(dout + ((b * Po + yo + dout_border) * Po + xo + dout_border) * ld + c);
        bool live[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int y = 2 * yo + (t >> 1), xx = 2 * xo + (t & 1);
            live[t] = y < H && xx < H;
            if (live[t]) x[t].v = *reinterpret_cast<const typename Vec16<T>::type*>(in + img + ((int64_t)(y + 1) * P + xx + 1) * ld);
        }
        Lanes<T> o[4];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float bf = to_f(x[0][j]);
            int first = 0;
#pragma unroll
            for (int t = 1; t < 4; ++t)
                if (live[t] && to_f(x[t][j]) > bf) { bf = to_f(x[t][j]); first = t; }
#pragma unroll
            for (int t = 0; t < 4; ++t) o[t][j] = (t == first && bf > 0.f) ? g[j] : T(0);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int y = 2 * yo + (t >> 1), xx = 2 * xo + (t & 1);
            if (live[t]) *reinterpret_cast<typename Vec16<T>::type*>(din + img + ((int64_t)(y + 1) * P + xx + 1) * ld) = o[t].v;
        }
    }
}

// Kernel of the input-gradient convolution.  W = [(tap, ci < cin)][ldw] (HWIO flattened); the result
// Wt[ci < cin_ld][Kt], column (tap', co < cout) = W[(8 - tap', ci)][co]: taps flipped ((2-ky)*3 + (2-kx) = 8 - tap),
// channels transposed; zero in the pad rows ci >= cin and the pad columns >= 9 * cout.
template <typename T>
__global__ __launch_bounds__(256) void conv_wflip_kernel(const T* __restrict__ W, int cin, int cin_ld, int cout, int ldw, T* __restrict__ Wt, int Kt) {
    const int total = cin_ld * Kt;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int ci = e / Kt, k = e - ci * Kt;
        const int tap = k / cout, co = k - tap * cout;
        T v = T(0);
        if (ci < cin && tap < 9) v = W[(int64_t)((8 - tap) * cin + ci) * ldw + co];
        Wt[e] = v;
    }
}

static int grid_for(int64_t n) {
    int64_t nb = (n + 255) / 256;
    return (int)(nb < 1 ? 1 : (nb > 65536 ? 65536 : nb));
}
static inline int esize(int dtype) { return dtype == DMVAE_BF16 ? 2 : 4; }

int conv_first_fwd_launch(hipStream_t s, int dtype, const void* x, int64_t bstride, int H, int64_t n_img, const void* W, int ldw,
                          const float* bias, int cout, void* out, int ld) {
    const int64_t n_units = n_img * H * (H / 4);
    if ((ld != 32 && ld != 64) || cout != 32 || H % 4 || n_units >= (1ll << 31)) { set_error("conv_first_fwd: 32 channels, side a multiple of 4 (ld=%d cout=%d H=%d)", ld, cout, H); return DMVAE_EINVAL; }
    ProfScope ps(s, "conv_first_fwd", 2.0 * n_img * H * H * 9 * cout, (double)n_img * H * H * ld * esize(dtype));
    const int nb = (int)std::min<int64_t>((n_units + 63) / 64, 256 * 16);
    if (dtype == DMVAE_BF16) DMVAE_LAUNCH((conv_first_fwd_kernel<bf16_t>), dim3(nb), dim3(256), 0, s, (const bf16_t*)x, bstride, H, (int)n_units, (const bf16_t*)W, ldw, bias, (bf16_t*)out, ld);
    else DMVAE_LAUNCH((conv_first_fwd_kernel<float>), dim3(nb), dim3(256), 0, s, (const float*)x, bstride, H, (int)n_units, (const float*)W, ldw, bias, (float*)out, ld);
    return check_launch("conv_first_fwd");
}

// few, long blocks: every block ends with a 80-value shuffle + LDS reduction (512 blocks: 157 us, 2048: 190 us at B = 4096)
int conv_first_dw_blocks(int H, int64_t n_img) { return (int)std::min<int64_t>(512, (n_img * H * (H / 4) + 63) / 64); }
// part: conv_first_dw_blocks() x 320 floats of scratch
int conv_first_dw_launch(hipStream_t s, int dtype, const void* x, int64_t bstride, int H, int64_t n_img, const void* dY, int ld, int cout,
                         float* dW, int ldw, float* db, float* part) {
    const int64_t n_units = n_img * H * (H / 4);
    if ((ld != 32 && ld != 64) || cout != 32 || H % 4 || n_units >= (1ll << 31) || !part) { set_error("conv_first_dw: 32 channels, side a multiple of 4, scratch (ld=%d cout=%d H=%d)", ld, cout, H); return DMVAE_EINVAL; }
    ProfScope ps(s, "conv_first_dw", 2.0 * n_img * H * H * 9 * cout, (double)n_img * H * H * ld * esize(dtype));
    const int nb = conv_first_dw_blocks(H, n_img);
    const int upb = (int)((n_units + nb - 1) / nb);
    if (dtype == DMVAE_BF16) DMVAE_LAUNCH((conv_first_dw_kernel<bf16_t>), dim3(nb), dim3(256), 0, s, (const bf16_t*)x, bstride, H, (int)n_units, (const bf16_t*)dY, part, upb, ld);
    else DMVAE_LAUNCH((conv_first_dw_kernel<float>), dim3(nb), dim3(256), 0, s, (const float*)x, bstride, H, (int)n_units, (const float*)dY, part, upb, ld);
    DMVAE_LAUNCH(conv_first_dw_reduce_kernel, dim3(10), dim3(1024), 0, s, (const float*)part, nb, dW, ldw, db);
    return check_launch("conv_first_dw");
}

int zero_border_launch(hipStream_t s, int dtype, void* a, int P, int ld, int64_t n_img) {
    if (ld % 32) { set_error("zero_border: channel stride %d must be a multiple of 32", ld); return DMVAE_EINVAL; }
    const int64_t n = n_img * (4 * P - 4) * (ld / (16 / esize(dtype)));
    ProfScope ps(s, "zero_border", 0.0, (double)n * 16);
    if (dtype == DMVAE_BF16) DMVAE_LAUNCH((zero_border_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s, (bf16_t*)a, P, ld, n_img);
    else DMVAE_LAUNCH((zero_border_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, (float*)a, P, ld, n_img);
    return check_launch("zero_border");
}

int maxpool2_fwd_launch(hipStream_t s, int dtype, const void* in, int H, int ld, int64_t n_img, void* out, int out_border) {
    const int Ho = (H + 1) / 2;
    const int64_t n_win = n_img * Ho * Ho;
    if (ld % 32) { set_error("maxpool2: channel stride %d must be a multiple of 32", ld); return DMVAE_EINVAL; }
    ProfScope ps(s, "maxpool2_fwd", 0.0, ((double)n_img * H * H + n_win) * ld * esize(dtype));
    const int nb = grid_for(n_win * (ld / (16 / esize(dtype))));
    if (dtype == DMVAE_BF16) DMVAE_LAUNCH((maxpool2_fwd_kernel<bf16_t>), dim3(nb), dim3(256), 0, s, (const bf16_t*)in, H, ld, n_win, (bf16_t*)out, out_border);
    else DMVAE_LAUNCH((maxpool2_fwd_kernel<float>), dim3(nb), dim3(256), 0, s, (const float*)in, H, ld, n_win, (float*)out, out_border);
    return check_launch("maxpool2_fwd");
}

int maxpool2_bwd_relu_launch(hipStream_t s, int dtype, const void* in, const void* dout, int H, int ld, int64_t n_img, void* din, int dout_border) {
    const int Ho = (H + 1) / 2;
    const int64_t n_win = n_img * Ho * Ho;
    if (ld % 32) { set_error("maxpool2: channel stride %d must be a multiple of 32", ld); return DMVAE_EINVAL; }
    ProfScope ps(s, "maxpool2_bwd_relu", 0.0, (2.0 * n_img * H * H + n_win) * ld * esize(dtype));
    const int nb = grid_for(n_win * (ld / (16 / esize(dtype))));
    if (dtype == DMVAE_BF16) DMVAE_LAUNCH((maxpool2_bwd_relu_kernel<bf16_t>), dim3(nb), dim3(256), 0, s, (const bf16_t*)in, (const bf16_t*)dout, H, ld, n_win, (bf16_t*)din, dout_border);
    else DMVAE_LAUNCH((maxpool2_bwd_relu_kernel<float>), dim3(nb), dim3(256), 0, s, (const float*)in, (const float*)dout, H, ld, n_win, (float*)din, dout_border);
    return check_launch("maxpool2_bwd_relu");
}

int conv_wflip_launch(hipStream_t s, int dtype, const void* W, int cin, int cin_ld, int cout, int ldw, void* Wt, int Kt) {
    if (Kt < 9 * cout || cin_ld < cin) { set_error("conv_wflip: pads smaller than the kernel"); return DMVAE_EINVAL; }
    ProfScope ps(s, "conv_wflip", 0.0, 2.0 * cin_ld * Kt * esize(dtype));
    const int nb = grid_for((int64_t)cin_ld * Kt);
    if (dtype == DMVAE_BF16) DMVAE_LAUNCH((conv_wflip_kernel<bf16_t>), dim3(nb), dim3(256), 0, s, (const bf16_t*)W, cin, cin_ld, cout, ldw, (bf16_t*)Wt, Kt);
    else DMVAE_LAUNCH((conv_wflip_kernel<float>), dim3(nb), dim3(256), 0, s, (const float*)W, cin, cin_ld, cout, ldw, (float*)Wt, Kt);
    return check_launch("conv_wflip");
}

}  // namespace dmvae
