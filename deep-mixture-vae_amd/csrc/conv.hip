// CNN encoder trunk of the checked-in DeepMixtureVAE (base_models.py:176-216; Convolution /
// MaxPooling at includes/layers.py:39-77): the data-movement kernels around the GEMMs.
//
// A 3x3 SAME stride-1 convolution in NHWC / HWIO is the GEMM  [pixels][9*Cin] x [9*Cin][Cout]; its
// weight gradient is the DW-layout GEMM over the same patch matrix (K = pixels, split-K) and its
// input gradient is the same convolution of dY with the flipped, transposed kernel.  Round 1 forms
// the patch matrix explicitly (im2col3x3) and runs the step's own MFMA GEMMs on it, so the conv
// layers inherit their epilogues (bias+ReLU, ReLU gate, bias gradient) and their parity tests; the
// patch matrices of the forward pass are KEPT for the weight-gradient GEMMs (HBM is 288 GB).  The
// implicit form (patch addresses generated inside the GEMM's LDS-DMA loads) is the next step.
//
// Activations are [B][H][W][ld] with ld = channels padded to 64 (pads are exact zeros); T is the
// activation type of the plan (float in parity mode, bf16 otherwise).
#include <algorithm>
#include "common.h"
#include "kernels.h"

namespace dmvae {

template <typename T> struct Vec16;                       // 16-byte vector of T
template <> struct Vec16<float> { typedef float4 type; static constexpr int N = 4; };
template <> struct Vec16<bf16_t> { typedef uint4 type; static constexpr int N = 8; };

// out[pix][k], k = (ky*3+kx)*C + c  <-  in[b, y+ky-1, x+kx-1, c]; 0 outside the image and for k >= 9C
// (tf.nn.conv2d padding='SAME').  Vector path: C a multiple of the 16-byte vector; scalar path: C = 1.
template <typename T, bool VEC>
__global__ __launch_bounds__(256) void im2col3x3_kernel(const T* __restrict__ in, int64_t bstride, int ldc, int C, int H, int W,
                                                        int n_pix, T* __restrict__ out, int Kpad) {
    constexpr int V = Vec16<T>::N;                  // every thread writes 16 bytes; !VEC: gathered element by element (C = 1)
    constexpr int PB = 64;                          // pixels per block pass: all index arithmetic stays 32-bit
    const int kv = Kpad / V, HW = H * W;
    for (int p0 = blockIdx.x * PB; p0 < n_pix; p0 += gridDim.x * PB) {
        const int np = min(PB, n_pix - p0);
        for (int li = threadIdx.x; li < np * kv; li += 256) {
            const int lp = li / kv, k = (li - lp * kv) * V;
            const int pix = p0 + lp;
            const int tap = k / C, c = k - tap * C;
            const int b = pix / HW, r = pix - b * HW, y = r / W, x = r - y * W;
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            const bool live = tap < 9 && yy >= 0 && yy < H && xx >= 0 && xx < W;
            const int64_t src = (int64_t)b * bstride + (int64_t)(yy * W + xx) * ldc + c;
            const int64_t dst = (int64_t)pix * Kpad + k;
            if constexpr (VEC) {
                typename Vec16<T>::type v = {};
                if (live) v = *reinterpret_cast<const typename Vec16<T>::type*>(in + src);
                *reinterpret_cast<typename Vec16<T>::type*>(out + dst) = v;
            } else {
                alignas(16) T v[V];
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const int kj = k + j, tj = kj / C, cj = kj - tj * C;
                    const int yj = y + tj / 3 - 1, xj = x + tj % 3 - 1;
                    const bool lj = tj < 9 && yj >= 0 && yj < H && xj >= 0 && xj < W;
                    v[j] = lj ? in[(int64_t)b * bstride + (int64_t)(yj * W + xj) * ldc + cj] : T(0);
                }
                *reinterpret_cast<typename Vec16<T>::type*>(out + dst) = *reinterpret_cast<const typename Vec16<T>::type*>(v);
            }
        }
    }
}

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16_t v) { return bf2f(v); }

// 16 bytes of channels as scalars
template <typename T> struct Lanes {
    typename Vec16<T>::type v;
    __device__ __forceinline__ T& operator[](int i) { return reinterpret_cast<T*>(&v)[i]; }
};

// tf.nn.max_pool ksize 2, strides 2, padding SAME: out = ceil(H/2); the pad (bottom / right, odd H) never
// wins.  One thread = one window x 16 bytes of channels.
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ in, int H, int W, int ld, int64_t n_win, T* __restrict__ out) {
    constexpr int V = Vec16<T>::N;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, cv = ld / V;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_win * cv; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % cv) * V;
        const int64_t q = e / cv;
        const int xo = (int)(q % Wo), yo = (int)((q / Wo) % Ho);
        const int64_t b = q / ((int64_t)Wo * Ho);
        const T* base = in + (b * H * W) * ld + c;
        Lanes<T> best;
        best.v = *reinterpret_cast<const typename Vec16<T>::type*>(base + ((int64_t)(2 * yo) * W + 2 * xo) * ld);
#pragma unroll
        for (int t = 1; t < 4; ++t) {
            const int yy = 2 * yo + (t >> 1), xx = 2 * xo + (t & 1);
            if (yy < H && xx < W) {
                Lanes<T> v;
                v.v = *reinterpret_cast<const typename Vec16<T>::type*>(base + ((int64_t)yy * W + xx) * ld);
#pragma unroll
                for (int j = 0; j < V; ++j)
                    if (to_f(v[j]) > to_f(best[j])) best[j] = v[j];
            }
        }
        *reinterpret_cast<typename Vec16<T>::type*>(out + q * ld + c) = best.v;
    }
}

// Gradient of ReLU -> max-pool: a pixel receives its window's gradient iff it is the FIRST maximum of the
// window in row-major order (TF's MaxPoolGrad) and its own value is positive (the ReLU in front of the
// pool).  One thread = one window x 16 bytes of channels: reads the window once, writes its four pixels.
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_relu_kernel(const T* __restrict__ in, const T* __restrict__ dout, int H, int W, int ld,
                                                                int64_t n_win, T* __restrict__ din) {
    constexpr int V = Vec16<T>::N;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2, cv = ld / V;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_win * cv; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % cv) * V;
        const int64_t q = e / cv;
        const int xo = (int)(q % Wo), yo = (int)((q / Wo) % Ho);
        const int64_t b = q / ((int64_t)Wo * Ho);
        const int64_t img = (b * H * W) * ld + c;
        Lanes<T> x[4], g;
        g.v = *reinterpret_cast<const typename Vec16<T>::type*>(dout + q * ld + c);
        bool live[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int yy = 2 * yo + (t >> 1), xx = 2 * xo + (t & 1);
            live[t] = yy < H && xx < W;
            if (live[t]) x[t].v = *reinterpret_cast<const typename Vec16<T>::type*>(in + img + ((int64_t)yy * W + xx) * ld);
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
            const int yy = 2 * yo + (t >> 1), xx = 2 * xo + (t & 1);
            if (live[t]) *reinterpret_cast<typename Vec16<T>::type*>(din + img + ((int64_t)yy * W + xx) * ld) = o[t].v;
        }
    }
}

// Kernel of the input-gradient convolution: Wt[ci][(jy*3+jx)*Cout + co] = W[((2-jy)*3 + (2-jx))*Cin + ci][co]
// (taps flipped, channels transposed), zero in the pad rows ci >= Cin and pad columns.
template <typename T>
__global__ __launch_bounds__(256) void conv_wflip_kernel(const T* __restrict__ W, int ldw, int Cin, int Cout, T* __restrict__ Wt, int rows_pad, int Ktpad) {
    const int total = rows_pad * Ktpad;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
        const int ci = e / Ktpad, k = e - ci * Ktpad;
        const int tap = k / Cout, co = k - tap * Cout;
        T v = T(0);
        if (ci < Cin && tap < 9) v = W[(int64_t)((8 - tap) * Cin + ci) * ldw + co];     // (2-jy)*3 + (2-jx) = 8 - tap
        Wt[e] = v;
    }
}

static int grid_for(int64_t n) {
    int64_t nb = (n + 255) / 256;
    return (int)(nb < 1 ? 1 : (nb > 65536 ? 65536 : nb));
}

int im2col3x3_launch(hipStream_t s, int dtype, const void* in, int64_t bstride, int ldc, int C, int H, int W, int64_t n_img, void* out, int Kpad) {
    const int64_t n_pix = n_img * H * W;
    const int V = dtype == DMVAE_BF16 ? 8 : 4;
    const bool vec = C % V == 0 && ldc % V == 0 && bstride % V == 0;
    if (!vec && C != 1) { set_error("im2col3x3: %d channels: a multiple of %d or 1", C, V); return DMVAE_EUNSUPPORTED; }
    if (Kpad % 64 || Kpad < 9 * C) { set_error("im2col3x3: Kpad=%d must be a multiple of 64 >= 9*C", Kpad); return DMVAE_EINVAL; }
    const double bytes = (double)n_pix * Kpad * (dtype == DMVAE_BF16 ? 2 : 4) * 1.12;
    ProfScope ps(s, "im2col3x3", 0.0, bytes);
    if (n_pix >= (1ll << 31) - 64) { set_error("im2col3x3: %lld pixels exceed the 32-bit pixel index", (long long)n_pix); return DMVAE_EUNSUPPORTED; }
    const int nb = (int)std::min<int64_t>((n_pix + 63) / 64, 65536);
    const int np = (int)n_pix;
    if (dtype == DMVAE_BF16) {
        if (vec) hipLaunchKernelGGL((im2col3x3_kernel<bf16_t, true>), dim3(nb), dim3(256), 0, s, (const bf16_t*)in, bstride, ldc, C, H, W, np, (bf16_t*)out, Kpad);
        else hipLaunchKernelGGL((im2col3x3_kernel<bf16_t, false>), dim3(nb), dim3(256), 0, s, (const bf16_t*)in, bstride, ldc, C, H, W, np, (bf16_t*)out, Kpad);
    } else {
        if (vec) hipLaunchKernelGGL((im2col3x3_kernel<float, true>), dim3(nb), dim3(256), 0, s, (const float*)in, bstride, ldc, C, H, W, np, (float*)out, Kpad);
        else hipLaunchKernelGGL((im2col3x3_kernel<float, false>), dim3(nb), dim3(256), 0, s, (const float*)in, bstride, ldc, C, H, W, np, (float*)out, Kpad);
    }
    return check_launch("im2col3x3");
}

int maxpool2_fwd_launch(hipStream_t s, int dtype, const void* in, int H, int W, int ld, int64_t n_img, void* out) {
    const int64_t n_win = n_img * ((H + 1) / 2) * ((W + 1) / 2);
    const int es = dtype == DMVAE_BF16 ? 2 : 4;
    if (ld % 64) { set_error("maxpool2: channel stride %d must be a multiple of 64", ld); return DMVAE_EINVAL; }
    ProfScope ps(s, "maxpool2_fwd", 0.0, ((double)n_img * H * W + n_win) * ld * es);
    const int nb = grid_for(n_win * (ld / (16 / es)));
    if (dtype == DMVAE_BF16) hipLaunchKernelGGL((maxpool2_fwd_kernel<bf16_t>), dim3(nb), dim3(256), 0, s, (const bf16_t*)in, H, W, ld, n_win, (bf16_t*)out);
    else hipLaunchKernelGGL((maxpool2_fwd_kernel<float>), dim3(nb), dim3(256), 0, s, (const float*)in, H, W, ld, n_win, (float*)out);
    return check_launch("maxpool2_fwd");
}

int maxpool2_bwd_relu_launch(hipStream_t s, int dtype, const void* in, const void* dout, int H, int W, int ld, int64_t n_img, void* din) {
    const int64_t n_win = n_img * ((H + 1) / 2) * ((W + 1) / 2);
    const int es = dtype == DMVAE_BF16 ? 2 : 4;
    if (ld % 64) { set_error("maxpool2: channel stride %d must be a multiple of 64", ld); return DMVAE_EINVAL; }
    ProfScope ps(s, "maxpool2_bwd_relu", 0.0, (2.0 * n_img * H * W + n_win) * ld * es);
    const int nb = grid_for(n_win * (ld / (16 / es)));
    if (dtype == DMVAE_BF16) hipLaunchKernelGGL((maxpool2_bwd_relu_kernel<bf16_t>), dim3(nb), dim3(256), 0, s, (const bf16_t*)in, (const bf16_t*)dout, H, W, ld, n_win, (bf16_t*)din);
    else hipLaunchKernelGGL((maxpool2_bwd_relu_kernel<float>), dim3(nb), dim3(256), 0, s, (const float*)in, (const float*)dout, H, W, ld, n_win, (float*)din);
    return check_launch("maxpool2_bwd_relu");
}

int conv_wflip_launch(hipStream_t s, int dtype, const void* W, int ldw, int Cin, int Cout, void* Wt, int rows_pad, int Ktpad) {
    if (Ktpad < 9 * Cout || rows_pad < Cin) { set_error("conv_wflip: pads smaller than the kernel"); return DMVAE_EINVAL; }
    ProfScope ps(s, "conv_wflip", 0.0, 2.0 * rows_pad * Ktpad * (dtype == DMVAE_BF16 ? 2 : 4));
    const int nb = grid_for((int64_t)rows_pad * Ktpad);
    if (dtype == DMVAE_BF16) hipLaunchKernelGGL((conv_wflip_kernel<bf16_t>), dim3(nb), dim3(256), 0, s, (const bf16_t*)W, ldw, Cin, Cout, (bf16_t*)Wt, rows_pad, Ktpad);
    else hipLaunchKernelGGL((conv_wflip_kernel<float>), dim3(nb), dim3(256), 0, s, (const float*)W, ldw, Cin, Cout, (float*)Wt, rows_pad, Ktpad);
    return check_launch("conv_wflip");
}

}  // namespace dmvae
