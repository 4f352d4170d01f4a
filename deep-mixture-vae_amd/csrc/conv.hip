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
                                                        int64_t n_pix, T* __restrict__ out, int Kpad) {
    constexpr int V = VEC ? Vec16<T>::N : 1;
    const int kv = Kpad / V;
    const int64_t total = n_pix * kv;
    const int HW = H * W;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t pix = e / kv;
        const int k = (int)(e - pix * kv) * V;
        const int tap = k / C, c = k - tap * C;
        const int64_t b = pix / HW;
        const int r = (int)(pix - b * HW), y = r / W, x = r - y * W;
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        const bool live = tap < 9 && yy >= 0 && yy < H && xx >= 0 && xx < W;
        if constexpr (VEC) {
            typename Vec16<T>::type v = {};
            if (live) v = *reinterpret_cast<const typename Vec16<T>::type*>(in + b * bstride + ((int64_t)yy * W + xx) * ldc + c);
            *reinterpret_cast<typename Vec16<T>::type*>(out + pix * Kpad + k) = v;
        } else {
            out[pix * Kpad + k] = live ? in[b * bstride + ((int64_t)yy * W + xx) * ldc + c] : T(0);
        }
    }
}

__device__ __forceinline__ float to_f(float v) { return v; }
__device__ __forceinline__ float to_f(bf16_t v) { return bf2f(v); }

// tf.nn.max_pool ksize 2, strides 2, padding SAME: out = ceil(H/2); the pad (bottom / right, odd H) never wins
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ in, int H, int W, int ld, int64_t n_out, T* __restrict__ out) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_out; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % ld);
        const int64_t q = e / ld;
        const int xo = (int)(q % Wo), yo = (int)((q / Wo) % Ho);
        const int64_t b = q / ((int64_t)Wo * Ho);
        const T* base = in + (b * H * W) * ld + c;
        T best = base[((int64_t)(2 * yo) * W + 2 * xo) * ld];
        float bf = to_f(best);
#pragma unroll
        for (int t = 1; t < 4; ++t) {
            const int yy = 2 * yo + (t >> 1), xx = 2 * xo + (t & 1);
            if (yy < H && xx < W) {
                const T v = base[((int64_t)yy * W + xx) * ld];
                if (to_f(v) > bf) { bf = to_f(v); best = v; }
            }
        }
        out[e] = best;
    }
}

// Gradient of max-pool followed (in the forward graph: preceded) by ReLU: every input pixel belongs to one
// window; it receives the window's gradient iff it is the FIRST maximum in row-major window order (TF's
// MaxPoolGrad) and its own value is positive (the ReLU in front of the pool).  No atomics.
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_relu_kernel(const T* __restrict__ in, const T* __restrict__ dout, int H, int W, int ld,
                                                                int64_t n_in, T* __restrict__ din) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < n_in; e += (int64_t)gridDim.x * 256) {
        const int c = (int)(e % ld);
        const int64_t q = e / ld;
        const int x = (int)(q % W), y = (int)((q / W) % H);
        const int64_t b = q / ((int64_t)W * H);
        const int yo = y >> 1, xo = x >> 1, me = ((y & 1) << 1) | (x & 1);
        const T* base = in + (b * H * W) * ld + c;
        float bf = 0.f;
        int first = -1;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int yy = 2 * yo + (t >> 1), xx = 2 * xo + (t & 1);
            if (yy < H && xx < W) {
                const float v = to_f(base[((int64_t)yy * W + xx) * ld]);
                if (first < 0 || v > bf) { bf = v; first = t; }
            }
        }
        const float mine = to_f(in[e]);
        T g = T(0);
        if (first == me && mine > 0.f) g = dout[((b * Ho + yo) * Wo + xo) * ld + c];
        din[e] = g;
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
    const int nb = grid_for(n_pix * (Kpad / (vec ? V : 1)));
    if (dtype == DMVAE_BF16) {
        if (vec) hipLaunchKernelGGL((im2col3x3_kernel<bf16_t, true>), dim3(nb), dim3(256), 0, s, (const bf16_t*)in, bstride, ldc, C, H, W, n_pix, (bf16_t*)out, Kpad);
        else hipLaunchKernelGGL((im2col3x3_kernel<bf16_t, false>), dim3(nb), dim3(256), 0, s, (const bf16_t*)in, bstride, ldc, C, H, W, n_pix, (bf16_t*)out, Kpad);
    } else {
        if (vec) hipLaunchKernelGGL((im2col3x3_kernel<float, true>), dim3(nb), dim3(256), 0, s, (const float*)in, bstride, ldc, C, H, W, n_pix, (float*)out, Kpad);
        else hipLaunchKernelGGL((im2col3x3_kernel<float, false>), dim3(nb), dim3(256), 0, s, (const float*)in, bstride, ldc, C, H, W, n_pix, (float*)out, Kpad);
    }
    return check_launch("im2col3x3");
}

int maxpool2_fwd_launch(hipStream_t s, int dtype, const void* in, int H, int W, int ld, int64_t n_img, void* out) {
    const int64_t n_out = n_img * ((H + 1) / 2) * ((W + 1) / 2) * ld;
    ProfScope ps(s, "maxpool2_fwd", 0.0, (double)n_out * 5 * (dtype == DMVAE_BF16 ? 2 : 4));
    if (dtype == DMVAE_BF16) hipLaunchKernelGGL((maxpool2_fwd_kernel<bf16_t>), dim3(grid_for(n_out)), dim3(256), 0, s, (const bf16_t*)in, H, W, ld, n_out, (bf16_t*)out);
    else hipLaunchKernelGGL((maxpool2_fwd_kernel<float>), dim3(grid_for(n_out)), dim3(256), 0, s, (const float*)in, H, W, ld, n_out, (float*)out);
    return check_launch("maxpool2_fwd");
}

int maxpool2_bwd_relu_launch(hipStream_t s, int dtype, const void* in, const void* dout, int H, int W, int ld, int64_t n_img, void* din) {
    const int64_t n_in = n_img * H * W * ld;
    ProfScope ps(s, "maxpool2_bwd_relu", 0.0, (double)n_in * 2.25 * (dtype == DMVAE_BF16 ? 2 : 4));
    if (dtype == DMVAE_BF16) hipLaunchKernelGGL((maxpool2_bwd_relu_kernel<bf16_t>), dim3(grid_for(n_in)), dim3(256), 0, s, (const bf16_t*)in, (const bf16_t*)dout, H, W, ld, n_in, (bf16_t*)din);
    else hipLaunchKernelGGL((maxpool2_bwd_relu_kernel<float>), dim3(grid_for(n_in)), dim3(256), 0, s, (const float*)in, (const float*)dout, H, W, ld, n_in, (float*)din);
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
