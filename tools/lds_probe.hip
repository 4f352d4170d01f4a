// LDS read bandwidth per CU by instruction (tools/README.md):  hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o deep-mixture-vae_amd/build/lds_probe
// 8 waves per workgroup, one workgroup per CU; every wave re-reads its own 8 KB of LDS `reps` times with
//   mode 0: ds_read_b128 (16 B per lane: what k-contiguous operands use)     mode 1: ds_read_b64_tr_b16 (8 B per lane: the transposing read)
//   mode 2: ds_read_b64                                                        mode 3: ds_read_b32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

template <int MODE>
__global__ __launch_bounds__(512) void lds_kernel(int reps, unsigned long long* ticks, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    char* mine = lds + wave * 8192;
    float acc = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if constexpr (MODE == 0) {
                const float4 v = *reinterpret_cast<const float4*>(mine + ((j * 1024 + lane * 16 + r * 1024) & 8191));
                acc += v.x + v.w;
            } else if constexpr (MODE == 1) {
                const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(mine + ((j * 512 + lane * 8 + r * 512) & 8191)));
                acc += (float)v[0] + (float)v[3];
            } else if constexpr (MODE == 2) {
                const float2 v = *reinterpret_cast<const float2*>(mine + ((j * 512 + lane * 8 + r * 512) & 8191));
                acc += v.x + v.y;
            } else {
                acc += *reinterpret_cast<const float*>(mine + ((j * 256 + lane * 4 + r * 256) & 8191));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
    if (acc == 123.456f) sink[0] = acc;
}

template <int MODE>
static void run(const char* name, int bytes_per_lane, int reps, unsigned long long* dt, float* sink) {
    std::vector<unsigned long long> h(256);
    for (int it = 0; it < 3; ++it) { hipLaunchKernelGGL(lds_kernel<MODE>, dim3(256), dim3(512), 0, 0, reps, dt, sink); CK(hipDeviceSynchronize()); }
    CK(hipMemcpy(h.data(), dt, 8 * 256, hipMemcpyDeviceToHost));
    double sum = 0; for (auto t : h) sum += (double)t;
    const double bytes = (double)reps * 8 * 64 * bytes_per_lane * 8;     // 8 instructions x 64 lanes x bytes x 8 waves
    const double us = sum / 256 / 100.0;
    printf("%-24s %8.1f GB/s per CU = %6.1f B/clk at 2.1 GHz   (%d reps, %.1f us)\n", name, bytes / (us * 1e-6) / 1e9, bytes / (us * 1e-6) / 2.1e9, reps, us);
}

int main() {
    unsigned long long* dt; float* sink;
    CK(hipMalloc(&dt, 8 * 256)); CK(hipMalloc(&sink, 64));
    run<0>("ds_read_b128", 16, 4000, dt, sink);
    run<1>("ds_read_b64_tr_b16", 8, 4000, dt, sink);
    run<2>("ds_read_b64", 8, 4000, dt, sink);
    run<3>("ds_read_b32", 4, 4000, dt, sink);
    return 0;
}
