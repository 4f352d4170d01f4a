// Per-CU intake on MI355X by load path (tools/README.md):  hipcc --offload-arch=gfx950 -O3 tools/intake_probe.hip -o deep-mixture-vae_amd/build/intake_probe
//   mode 0  global_load_lds_dwordx4 (LDS-DMA, what the GEMM rings use)
//   mode 1  global_load_dwordx4 into VGPRs, discarded
//   mode 2  global_load_dwordx4 into VGPRs + ds_write_b128 (register-staged ring)
//   mode 3  LDS-DMA with the GEMM's k-contiguous tile pattern: a wave instruction = 8 rows x 128 B, row stride `stride` bytes; a
//           workgroup's chunk = 64 rows x 128 B, successive chunks advance 128 B along the rows (the K loop), `region` = 64 rows x stride
// Every workgroup streams its own region `reps` times (region small: L2-resident after the first pass; large: from Infinity Cache / HBM).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE, int DEPTH, int SEG = 128>
__global__ __launch_bounds__(512) void intake_kernel(const char* base, size_t region, int reps, unsigned long long* ticks, float* sink, unsigned stride = 0, unsigned seg = 128) {      // MODE 0: stride = G > 0 -> workgroups b and b + G read the SAME region (panel sharing)
    __shared__ __attribute__((aligned(16))) char lds[DEPTH * 8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* mine = base + (size_t)((MODE == 0 && stride) ? blockIdx.x % stride : blockIdx.x) * region;
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)lds) + 1024u * wave);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    const int chunks = (int)(region / 8192);          // one chunk = 512 threads x 16 B
    for (int r = 0; r < reps; ++r) {
        for (int c0 = 0; c0 < chunks; c0 += DEPTH) {
            if constexpr (MODE == 3) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    constexpr unsigned STRIDE = 1024, lps = SEG / 16, rpi = 1024 / SEG;      // lanes per segment, rows per instruction
                    const int c = c0 + d, kt = c % (int)(STRIDE / SEG), rb = c / (int)(STRIDE / SEG);      // chunk c = row block rb (8 waves x rpi rows), k tile kt
                    const char* src = mine + (size_t)kt * SEG + ((size_t)rb * 8 + wave) * rpi * STRIDE;      // wave-uniform
                    const unsigned off = (unsigned)(lane / lps) * STRIDE + (unsigned)(lane % lps) * 16, dst = lds_w + d * 8192;
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "s"(src), "v"(off), "s"(dst) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if constexpr (MODE == 0) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    const char* src = mine + (size_t)(c0 + d) * 8192 + wave * 1024;      // wave-uniform
                    const unsigned off = lane * 16, dst = lds_w + d * 8192;
                    unsigned keep;
                    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                                 : "=&s"(keep) : "s"(src), "v"(off), "s"(dst) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                float4 v[DEPTH];
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) v[d] = *reinterpret_cast<const float4*>(mine + (size_t)(c0 + d) * 8192 + tid * 16);
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
                    if constexpr (MODE == 2) *reinterpret_cast<float4*>(lds + d * 8192 + tid * 16) = v[d];
                    else { acc.x += v[d].x; acc.y += v[d].y; acc.z += v[d].z; acc.w += v[d].w; }
                }
            }
        }
    }
    if constexpr (MODE != 1) { __syncthreads(); acc = *reinterpret_cast<float4*>(lds + tid * 16); }
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
    if (acc.x == 123.456f) sink[0] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE, int DEPTH, int SEG = 128>
static void run(const char* name, const char* buf, size_t region, int wgs, int reps, unsigned long long* dticks, float* sink, unsigned stride = 0, unsigned seg = 128) {
    std::vector<unsigned long long> h(wgs);
    for (int it = 0; it < 3; ++it) {
        hipLaunchKernelGGL((intake_kernel<MODE, DEPTH, SEG>), dim3(wgs), dim3(512), 0, 0, buf, region, reps, dticks, sink, stride, seg);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), dticks, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost));
    double worst = 0, sum = 0;
    for (auto t : h) { worst = t > worst ? (double)t : worst; sum += (double)t; }
    const double bytes = (double)region * reps;
    printf("%-34s depth %2d  region %7zu KB x %3d reps, %4d WGs : per-WG %6.1f GB/s (mean)  %6.1f GB/s (slowest)  chip %6.2f TB/s\n", name, DEPTH, region >> 10, reps, wgs,
           bytes / (sum / wgs * 10e-9) / 1e9, bytes / (worst * 10e-9) / 1e9, bytes * wgs / (worst * 10e-9) / 1e12);
}

// mode 4: the GEMM K loop's skeleton without the arithmetic: a ring of NST stages of IPW x 8 KB per workgroup; per iteration every wave
// issues its IPW instructions of the next stage, waits until its share of the oldest stage has landed (counted vmcnt), s_barrier.
template <int NST, int IPW, bool BARRIER>
__global__ __launch_bounds__(512) void ring_kernel(const char* base, size_t region, int iters, unsigned long long* ticks, float* sink) {
    __shared__ __attribute__((aligned(16))) char lds[NST * IPW * 8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char* mine = base + (size_t)blockIdx.x * region;
    const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)((__attribute__((address_space(3))) char*)lds) + 1024u * wave);
    const int nchunk = (int)(region / (IPW * 8192));
    auto issue = [&](int it, int slot) {
#pragma unroll
        for (int i = 0; i < IPW; ++i) {
            const char* src = mine + (size_t)(it % nchunk) * (IPW * 8192) + i * 8192 + wave * 1024;
            const unsigned off = lane * 16, dst = lds_w + (slot * IPW + i) * 8192;
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "s"(src), "v"(off), "s"(dst) : "memory");
        }
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int s = 0; s < NST; ++s) issue(s, s);
    for (int it = 0; it < iters; it += NST) {
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IPW * (NST - 1)) : "memory");       // stage it + s has landed (this wave's share)
            if (BARRIER) __builtin_amdgcn_s_barrier();
            issue(it + s + NST, s);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
    if (*reinterpret_cast<float*>(lds + tid * 16) == 123.456f) sink[0] = 1.f;
}

template <int NST, int IPW, bool BARRIER>
static void run_ring(const char* buf, size_t region, int wgs, int iters, unsigned long long* dticks, float* sink) {
    std::vector<unsigned long long> h(wgs);
    for (int it = 0; it < 3; ++it) {
        hipLaunchKernelGGL((ring_kernel<NST, IPW, BARRIER>), dim3(wgs), dim3(512), 0, 0, buf, region, iters, dticks, sink);
        CK(hipDeviceSynchronize());
    }
    CK(hipMemcpy(h.data(), dticks, sizeof(unsigned long long) * wgs, hipMemcpyDeviceToHost));
    double worst = 0, sum = 0;
    for (auto t : h) { worst = t > worst ? (double)t : worst; sum += (double)t; }
    const double bytes = (double)(iters + NST) * IPW * 8192;
    printf("ring: %d stages x %2d KB, %s, %4d WGs, region %4zu KB : per-WG %6.1f GB/s (mean)  %6.1f (slowest)  %.3f us per stage\n", NST, IPW * 8, BARRIER ? "barrier" : "no barrier", wgs,
           region >> 10, bytes / (sum / wgs * 10e-9) / 1e9, bytes / (worst * 10e-9) / 1e9, sum / wgs / 100.0 / (iters + NST));
}

int main() {
    const size_t total = (size_t)1 << 30;
    char* buf; unsigned long long* dt; float* sink;
    CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total)); CK(hipMalloc(&dt, 8 * 4096)); CK(hipMalloc(&sink, 64));
    for (int wgs : {256, 512}) {
        for (size_t region : {(size_t)64 << 10, (size_t)2 << 20}) {       // 64 KB per WG: L2-resident; 2 MB per WG: 0.5-1 GB in all, Infinity Cache / HBM
            const int reps = region == ((size_t)64 << 10) ? 64 : 2;
            run<0, 2>("LDS-DMA global_load_lds_dwordx4", buf, region, wgs, reps, dt, sink);
            run<0, 4>("LDS-DMA global_load_lds_dwordx4", buf, region, wgs, reps, dt, sink);
            run<0, 8>("LDS-DMA global_load_lds_dwordx4", buf, region, wgs, reps, dt, sink);
            run<1, 4>("global_load_dwordx4 -> VGPR", buf, region, wgs, reps, dt, sink);
            run<1, 8>("global_load_dwordx4 -> VGPR", buf, region, wgs, reps, dt, sink);
            run<2, 4>("global_load_dwordx4 -> ds_write", buf, region, wgs, reps, dt, sink);
            run<2, 8>("global_load_dwordx4 -> ds_write", buf, region, wgs, reps, dt, sink);
        }
    }
    // the GEMM's tile pattern: 64 rows x (stride / 128) K tiles per workgroup = stride * 64 bytes, re-read `reps` times (L2-resident)
    for (size_t kb : {64, 1024}) {       // 64 KB per WG: L2-resident; 1 MB per WG (256 MB): Infinity Cache / HBM
        const size_t region = kb << 10; const int iters = 1200;
        run_ring<3, 3, true>(buf, region, 256, iters, dt, sink);
        run_ring<3, 3, false>(buf, region, 256, iters, dt, sink);
        run_ring<6, 3, true>(buf, region, 256, iters, dt, sink);
        run_ring<2, 4, true>(buf, region, 256, iters, dt, sink);
        run_ring<3, 6, true>(buf, region, 256, iters, dt, sink);
        run_ring<3, 2, true>(buf, region, 256, iters, dt, sink);
    }
    // one pass over data that sits in the Infinity Cache but not in L2 (a launch boundary invalidates the L2s): what a GEMM's first touch sees
    for (size_t kb : {64, 128, 256, 512}) {
        run<0, 4>("LDS-DMA, ONE pass, Infinity-Cache", buf, kb << 10, 256, 1, dt, sink);
        run<0, 8>("LDS-DMA, ONE pass, Infinity-Cache", buf, kb << 10, 256, 1, dt, sink);
    }
    for (unsigned G : {256u, 64u, 32u, 8u}) {        // 256 / G workgroups (round-robin: b and b + G, G % 8 == 0 -> the same XCD) stream the same 64 KB in lockstep
        char nm[64]; snprintf(nm, sizeof nm, "LDS-DMA, %u WGs share a region", 256 / G);
        run<0, 4>(nm, buf, (size_t)64 << 10, 256, 64, dt, sink, G);
    }
    {
        const size_t region = (size_t)64 << 10;      // 64 rows x 1 KB: the same 64 KB (L2-resident, larger than L1) for every instruction shape
        const int reps = 64;
        run<3, 4, 128>("LDS-DMA 8 rows x 128 B per instr", buf, region, 256, reps, dt, sink);
        run<3, 8, 128>("LDS-DMA 8 rows x 128 B per instr", buf, region, 256, reps, dt, sink);
        run<3, 4, 256>("LDS-DMA 4 rows x 256 B per instr", buf, region, 256, reps, dt, sink);
        run<3, 8, 256>("LDS-DMA 4 rows x 256 B per instr", buf, region, 256, reps, dt, sink);
        run<3, 4, 512>("LDS-DMA 2 rows x 512 B per instr", buf, region, 256, reps, dt, sink);
        run<3, 8, 512>("LDS-DMA 2 rows x 512 B per instr", buf, region, 256, reps, dt, sink);
        run<3, 4, 1024>("LDS-DMA 1 row x 1 KB per instr", buf, region, 256, reps, dt, sink);
        run<3, 8, 1024>("LDS-DMA 1 row x 1 KB per instr", buf, region, 256, reps, dt, sink);
    }
    return 0;
}
