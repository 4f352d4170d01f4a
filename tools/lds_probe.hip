// LDS read rate per CU by instruction (tools/README.md):
//   hipcc --offload-arch=gfx950 -O3 tools/lds_probe.hip -o deep-mixture-vae_amd/build/lds_probe && deep-mixture-vae_amd/build/lds_probe
// Round 4 rewrite.  The first version consumed every result with VALU work (conversions, adds, per-iteration address
// arithmetic) between the reads and converted ticks at an assumed 2.1 GHz: it printed ds_read_b32 at 182 B/clk -- above the 128
// the instruction can do (MI355X_MICROARCH.md, LDS table) -- and ds_read_b64_tr_b16 at 60 % of a plain read, the figure DESIGN.md's
// "the dW K loop is LDS-bound" rested on.  Here: reads issued from inline asm with loop-invariant addresses and NO consumer in the
// loop (destinations kept live to one final wait), a drain every DRAIN reads or never, 1 or 2 waves per SIMD, and cycles taken
// from the in-kernel clock (s_memtime), so B/clk needs no clock assumption.
//   mode 0 ds_read_b128   1 ds_read_b64_tr_b16   2 ds_read_b64   3 ds_read_b32
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// 16 reads per group, destinations d[0..15]; every address = base VGPR + immediate (conflict-free: lane-linear)
template <int MODE>
__device__ __forceinline__ void group16(unsigned a, f32x4 (&q)[16], f32x2 (&h)[16], float (&w)[16]) {
#define RD128(i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q[i]) : "v"(a), "n"((i) * 1024))
#define RDTR(i)  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(h[i]) : "v"(a), "n"((i) * 512))
#define RD64(i)  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(h[i]) : "v"(a), "n"((i) * 512))
#define RD32(i)  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(w[i]) : "v"(a), "n"((i) * 256))
#define ALL16(M) M(0); M(1); M(2); M(3); M(4); M(5); M(6); M(7); M(8); M(9); M(10); M(11); M(12); M(13); M(14); M(15)
    if constexpr (MODE == 0) { ALL16(RD128); }
    else if constexpr (MODE == 1) { ALL16(RDTR); }
    else if constexpr (MODE == 2) { ALL16(RD64); }
    else { ALL16(RD32); }
}

template <int MODE, bool DRAIN>
__global__ void lds_kernel(int reps, unsigned long long* out, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 65536 / 4; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = (float)i;
    __syncthreads();
    // each wave reads inside its own 16 KiB window (mode 0: 16 x 1 KiB; others less): offsets stay < 64 KiB for 4 or 8 waves
    const unsigned bytes = MODE == 0 ? 16u : MODE == 3 ? 4u : 8u;
    const unsigned a = (unsigned)(size_t)((__attribute__((address_space(3))) char*)lds) + (unsigned)(wave & 3) * 16384u + (unsigned)lane * bytes;
    f32x4 q[16]; f32x2 h[16]; float w[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { q[i] = f32x4{0, 0, 0, 0}; h[i] = f32x2{0, 0}; w[i] = 0.f; }
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int r = 0; r < reps; ++r) {
        group16<MODE>(a, q, h, w);
        if constexpr (DRAIN) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float acc = 0.f;          // consume every destination AFTER the final wait (keeps them live across the loop)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc += q[i][0] + q[i][3] + h[i][0] + h[i][1] + w[i];
    // the workgroup's span: earliest wave start -> latest wave end (with two waves per SIMD the arbiter favours the older wave, which
    // then finishes early: ONE wave's elapsed time against ALL waves' bytes over-reads the rate -- the first round-4 run did that)
    __shared__ unsigned long long span[4];
    if (tid == 0) { span[0] = ~0ull; span[1] = 0; span[2] = ~0ull; span[3] = 0; }
    __syncthreads();
    if (lane == 0) {
        atomicMin(&span[0], c0); atomicMax(&span[1], c1);
        atomicMin(&span[2], r0); atomicMax(&span[3], r1);
    }
    __syncthreads();
    if (tid == 0) { out[2 * blockIdx.x] = span[1] - span[0]; out[2 * blockIdx.x + 1] = span[3] - span[2]; }
    if (acc == 123.456f) sink[0] = acc;
}

template <int MODE, bool DRAIN>
static void run(const char* name, int bytes_per_lane, int waves, int reps, unsigned long long* dt, float* sink) {
    std::vector<unsigned long long> h(512);
    for (int it = 0; it < 3; ++it) { hipLaunchKernelGGL((lds_kernel<MODE, DRAIN>), dim3(256), dim3(64 * waves), 0, 0, reps, dt, sink); CK(hipDeviceSynchronize()); }
    CK(hipMemcpy(h.data(), dt, 8 * 512, hipMemcpyDeviceToHost));
    std::vector<double> cyc, ghz;
    for (int b = 0; b < 256; ++b) { cyc.push_back((double)h[2 * b]); ghz.push_back((double)h[2 * b] / ((double)h[2 * b + 1] * 10.0) ); }   // realtime tick = 10 ns
    std::sort(cyc.begin(), cyc.end()); std::sort(ghz.begin(), ghz.end());
    const double bytes = (double)reps * 16 * 64 * bytes_per_lane * waves;
    printf("%-20s %d wave(s)/SIMD  %-22s %7.1f B/clk/CU   (median CU: %.0f cycles, clock %.2f GHz -> %.0f GB/s per CU)\n", name, waves / 4,
           DRAIN ? "lgkmcnt(0) per 16 reads" : "no wait inside the loop", bytes / cyc[128], cyc[128], ghz[128], bytes / cyc[128] * ghz[128]);
}

int main() {
    unsigned long long* dt; float* sink;
    CK(hipMalloc(&dt, 8 * 512)); CK(hipMalloc(&sink, 64));
    const int reps = 4000;
    for (int waves : {4, 8}) {
        run<0, false>("ds_read_b128", 16, waves, reps, dt, sink);       run<0, true>("ds_read_b128", 16, waves, reps, dt, sink);
        run<1, false>("ds_read_b64_tr_b16", 8, waves, reps, dt, sink);  run<1, true>("ds_read_b64_tr_b16", 8, waves, reps, dt, sink);
        run<2, false>("ds_read_b64", 8, waves, reps, dt, sink);         run<2, true>("ds_read_b64", 8, waves, reps, dt, sink);
        run<3, false>("ds_read_b32", 4, waves, reps, dt, sink);         run<3, true>("ds_read_b32", 4, waves, reps, dt, sink);
    }
    return 0;
}
