// CNN encoder trunk of the checked-in DeepMixtureVAE (base_models.py:176-216; Convolution /
// MaxPooling at includes/layers.py:39-77): the kernels around the GEMMs.
//
// A 3x3 SAME stride-1 convolution in NHWC / HWIO is the GEMM  [pixels][9*Cin] x [9*Cin][Cout]; its
// weight gradient is X_patches^T dY and its input gradient is the same convolution of dY with the
// flipped, channel-transposed kernel.  The patch matrix is NEVER FORMED (except for the one-channel
// first layer): activations live in HBM with a zero border, [B][P][P][C], P = side + 2, plus P + 1
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
// T is the activation type of the plan (float in parity mode, bf16 otherwise); channel strides are
// multiples of 64 (32-channel layers carry 32 zero channels).
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

// First layer (one input channel): explicit patch matrix over the padded pixel space,
//   out[m][k] = x[b, y+ky-1, x+kx-1]  (k = ky*3+kx < 9, m = interior pixel (y,x) of image b), else 0.
// x = the batch as loaded, [B][bstride] with the 784 pixels of an image contiguous.
template <typename T>
__global__ __launch_bounds__(256) void im2col_first_kernel(const T* __restrict__ x, int64_t bstride, int H, int n_rows, T* __restrict__ out, int Kpad) {
    constexpr int V = Vec16<T>::N;
    const int P = H + 2, R = P * P, kv = Kpad / V;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < (int64_t)n_rows * kv; e += (int64_t)gridDim.x * 256) {
        const int m = (int)(e / kv), k = (int)(e - (int64_t)m * kv) * V;
        const int b = m / R, r = m - b * R, yy = r / P, xx = r - yy * P;
        alignas(16) T v[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            const int t = k + j, y = yy - 1 + t / 3 - 1, xq = xx - 1 + t % 3 - 1;
            const bool live = t < 9 && yy >= 1 && yy <= H && xx >= 1 && xx <= H && y >= 0 && y < H && xq >= 0 && xq < H;
            v[j] = live ? x[(int64_t)b * bstride + y * H + xq] : T(0);
        }
        *reinterpret_cast<typename Vec16<T>::type*>(out + (int64_t)m * Kpad + k) = *reinterpret_cast<const typename Vec16<T>::type*>(v);
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

int im2col_first_launch(hipStream_t s, int dtype, const void* x, int64_t bstride, int H, int64_t n_img, void* out, int Kpad) {
    const int64_t n_rows = n_img * (H + 2) * (H + 2);
    if (Kpad % 64 || n_rows >= (1ll << 31)) { set_error("im2col_first: Kpad=%d / %lld rows", Kpad, (long long)n_rows); return DMVAE_EINVAL; }
    ProfScope ps(s, "im2col_first", 0.0, (double)n_rows * Kpad * esize(dtype));
    const int nb = grid_for(n_rows * (Kpad / (16 / esize(dtype))));
    if (dtype == DMVAE_BF16) hipLaunchKernelGGL((im2col_first_kernel<bf16_t>), dim3(nb), dim3(256), 0, s, (const bf16_t*)x, bstride, H, (int)n_rows, (bf16_t*)out, Kpad);
    else hipLaunchKernelGGL((im2col_first_kernel<float>), dim3(nb), dim3(256), 0, s, (const float*)x, bstride, H, (int)n_rows, (float*)out, Kpad);
    return check_launch("im2col_first");
}

int zero_border_launch(hipStream_t s, int dtype, void* a, int P, int ld, int64_t n_img) {
    if (ld % 64) { set_error("zero_border: channel stride %d must be a multiple of 64", ld); return DMVAE_EINVAL; }
    const int64_t n = n_img * (4 * P - 4) * (ld / (16 / esize(dtype)));
    ProfScope ps(s, "zero_border", 0.0, (double)n * 16);
    if (dtype == DMVAE_BF16) hipLaunchKernelGGL((zero_border_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s, (bf16_t*)a, P, ld, n_img);
    else hipLaunchKernelGGL((zero_border_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, (float*)a, P, ld, n_img);
    return check_launch("zero_border");
}

int maxpool2_fwd_launch(hipStream_t s, int dtype, const void* in, int H, int ld, int64_t n_img, void* out, int out_border) {
    const int Ho = (H + 1) / 2;
    const int64_t n_win = n_img * Ho * Ho;
    if (ld % 64) { set_error("maxpool2: channel stride %d must be a multiple of 64", ld); return DMVAE_EINVAL; }
    ProfScope ps(s, "maxpool2_fwd", 0.0, ((double)n_img * H * H + n_win) * ld * esize(dtype));
    const int nb = grid_for(n_win * (ld / (16 / esize(dtype))));
    if (dtype == DMVAE_BF16) hipLaunchKernelGGL((maxpool2_fwd_kernel<bf16_t>), dim3(nb), dim3(256), 0, s, (const bf16_t*)in, H, ld, n_win, (bf16_t*)out, out_border);
    else hipLaunchKernelGGL((maxpool2_fwd_kernel<float>), dim3(nb), dim3(256), 0, s, (const float*)in, H, ld, n_win, (float*)out, out_border);
    return check_launch("maxpool2_fwd");
}

int maxpool2_bwd_relu_launch(hipStream_t s, int dtype, const void* in, const void* dout, int H, int ld, int64_t n_img, void* din, int dout_border) {
    const int Ho = (H + 1) / 2;
    const int64_t n_win = n_img * Ho * Ho;
    if (ld % 64) { set_error("maxpool2: channel stride %d must be a multiple of 64", ld); return DMVAE_EINVAL; }
    ProfScope ps(s, "maxpool2_bwd_relu", 0.0, (2.0 * n_img * H * H + n_win) * ld * esize(dtype));
    const int nb = grid_for(n_win * (ld / (16 / esize(dtype))));
    if (dtype == DMVAE_BF16) hipLaunchKernelGGL((maxpool2_bwd_relu_kernel<bf16_t>), dim3(nb), dim3(256), 0, s, (const bf16_t*)in, (const bf16_t*)dout, H, ld, n_win, (bf16_t*)din, dout_border);
    else hipLaunchKernelGGL((maxpool2_bwd_relu_kernel<float>), dim3(nb), dim3(256), 0, s, (const float*)in, (const float*)dout, H, ld, n_win, (float*)din, dout_border);
    return check_launch("maxpool2_bwd_relu");
}

int conv_wflip_launch(hipStream_t s, int dtype, const void* W, int cin, int cin_ld, int cout, int ldw, void* Wt, int Kt) {
    if (Kt < 9 * cout || cin_ld < cin) { set_error("conv_wflip: pads smaller than the kernel"); return DMVAE_EINVAL; }
    ProfScope ps(s, "conv_wflip", 0.0, 2.0 * cin_ld * Kt * esize(dtype));
    const int nb = grid_for((int64_t)cin_ld * Kt);
    if (dtype == DMVAE_BF16) hipLaunchKernelGGL((conv_wflip_kernel<bf16_t>), dim3(nb), dim3(256), 0, s, (const bf16_t*)W, cin, cin_ld, cout, ldw, (bf16_t*)Wt, Kt);
    else hipLaunchKernelGGL((conv_wflip_kernel<float>), dim3(nb), dim3(256), 0, s, (const float*)W, cin, cin_ld, cout, ldw, (float*)Wt, Kt);
    return check_launch("conv_wflip");
}

}  // namespace dmvae
