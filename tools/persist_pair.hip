// Measurement tool (not product): what does keeping two dependent GEMMs of the head inside ONE launch cost or save on this code?
// The head's w1 (N = 15360, K = 5120, fused SwiGLU, 2 K slices reduced in the launch) feeds w2 (N = 5120, K = 7680, 3 slabs): an
// all-to-all seam (every workgroup of w2 reads the whole activation tensor w1 produced).  Both use 240 workgroups of 256 threads, so
// the product kernels' bodies (bd_gemm_kernel.h gemm_body) can run back to back in one persistent launch with a grid barrier
// between them -- release fence, XCD-hierarchical arrival counters, acquire fence -- and be compared, bit for bit and in time, with
// the two launches the engine issues.  Optionally the second body's first weight stages are requested before the barrier.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I bitdance_amd/csrc tools/persist_pair.hip -o tools/persist_pair
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "bd_gemm_kernel.h"

int bdk_gemm_tile(const GemmP&, int, hipStream_t) { return -1; }   // (declared by the header; not used here)

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Bar {                // one cache line per word: XCD counters [8], top counter, generation
    unsigned* xcd;          // [8 * 32]
    unsigned* top;          // [32]
    unsigned* gen;          // [32]
    int nxcd;
};

// XCD-hierarchical grid barrier (MI355X_MICROARCH "barrier-xcd"): the last arriver of an XCD carries that XCD into the top
// counter, the last XCD bumps the generation every workgroup polls.  Release before arriving, acquire after leaving.
__device__ __forceinline__ void grid_barrier(const Bar& b, unsigned per_xcd_expected[8], unsigned epoch) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                          // every wave's stores have reached the L2 before lane 0 writes it back
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned x = blockIdx.x % 8;                                   // the XCD this block runs on (dispatch order; speed only)
        const unsigned a = __hip_atomic_fetch_add(b.xcd + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
        if (a == per_xcd_expected[x] * epoch) {
            const unsigned t = __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1;
            if (t == 8u * epoch) __hip_atomic_store(b.gen, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        long long spins = 0;
        while (__hip_atomic_load(b.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch && ++spins < (1ll << 26)) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

struct PairArgs { GemmP p1, p2; Bar bar; unsigned per_xcd[8]; unsigned epoch; int prefetch; };

__global__ __launch_bounds__(256) void pair_kernel(PairArgs a) {
    gemm_body<4, 1, 4, BD_EPI_SWIGLU, 3, true, 0, 0>(a.p1);
    if (a.prefetch) {
        // ask for the first two weight stages of this workgroup's w2 slice before waiting (the lines land in the XCD's L2 / the
        // memory-side cache while the barrier runs): the "prefetch credit" of a run-ahead loader, without an LDS ring
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int pw = wave % 2, kg = wave / 2, S = a.p2.S, s = blockIdx.x % S, nt = blockIdx.x / S;
        const int nst_total = a.p2.K / 128, q = (nst_total + S - 1) / S, st0 = s * q;
        const u32x4* Wp = a.p2.W + (size_t)(nt * 2 + pw) * a.p2.PS + (size_t)(st0 * 2 + kg) * a.p2.SS + lane;
        u32x4 f = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) f ^= Wp[(size_t)i * a.p2.SS * 2 + j * 64];
        asm volatile("" :: "v"(f));
    }
    unsigned ex[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) ex[i] = a.per_xcd[i];
    grid_barrier(a.bar, ex, a.epoch);
    gemm_body<2, 2, 4, BD_EPI_PARTIAL, 2, false, 0, 0>(a.p2);
}

__global__ void fill_kernel(unsigned short* p, size_t n, unsigned seed) {      // small bf16 values with varied mantissas
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)(i * 2654435761u) ^ seed;
        h ^= h >> 13; h *= 0x5bd1e995u; h ^= h >> 15;
        const float v = ((int)(h & 0xff) - 128) * (1.0f / 2048.0f);
        p[i] = (unsigned short)(__float_as_uint(v) >> 16);
    }
}

int main() {
    const int M = 128, RB = 4, D = 5120, H = 7680;
    const size_t w1n = (size_t)2 * H * D, w2n = (size_t)D * H;
    const int rot = 3;
    unsigned short *W1, *W2, *A, *act;
    float *slab1, *slab2;
    int* cnt;
    unsigned* bar;
    CK(hipMalloc(&W1, w1n * 2 * rot)); CK(hipMalloc(&W2, w2n * 2 * rot));
    CK(hipMalloc(&A, (size_t)M * D * 2)); CK(hipMalloc(&act, (size_t)M * H * 2));
    CK(hipMalloc(&slab1, (size_t)2 * M * 2 * H * 4)); CK(hipMalloc(&slab2, (size_t)3 * M * D * 4));
    CK(hipMalloc(&cnt, 16384 * 4)); CK(hipMemset(cnt, 0, 16384 * 4));
    CK(hipMalloc(&bar, 10 * 32 * 4)); CK(hipMemset(bar, 0, 10 * 32 * 4));
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, W1, w1n * rot, 1u);
    hipLaunchKernelGGL(fill_kernel, dim3(2048), dim3(256), 0, 0, W2, w2n * rot, 2u);
    hipLaunchKernelGGL(fill_kernel, dim3(256), dim3(256), 0, 0, A, (size_t)M * D, 3u);
    CK(hipDeviceSynchronize());
    auto k1 = gemm_kernel<4, 1, 4, BD_EPI_SWIGLU, 3, true, 0, 0>;
    auto k2 = gemm_kernel<2, 2, 4, BD_EPI_PARTIAL, 2, false, 0, 0>;
    CK(hipFuncSetAttribute((const void*)pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    CK(hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    auto mk = [&](int r, GemmP& p1, GemmP& p2) {
        p1 = GemmP{(const u32x4*)A, (const u32x4*)(W1 + (size_t)r * w1n), slab1, act, nullptr, cnt, nullptr, RB, 2 * H, D, 2, M, (size_t)(D >> 4) * 64, 256};
        p2 = GemmP{(const u32x4*)act, (const u32x4*)(W2 + (size_t)r * w2n), slab2, nullptr, nullptr, nullptr, nullptr, RB, D, H, 3, M, (size_t)(H >> 4) * 64, 256};
    };
    PairArgs pa;
    pa.bar = Bar{bar, bar + 8 * 32, bar + 9 * 32, 8};
    for (int i = 0; i < 8; ++i) pa.per_xcd[i] = 240 / 8;                     // 240 blocks dealt round-robin to 8 XCDs
    unsigned epoch = 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> ref_act((size_t)M * H / 2), ref_slab((size_t)3 * M * D), got((size_t)3 * M * D);
    // reference: two launches
    GemmP p1, p2;
    mk(0, p1, p2);
    CK(hipMemset(act, 0, (size_t)M * H * 2)); CK(hipMemset(slab2, 0, (size_t)3 * M * D * 4));
    hipLaunchKernelGGL(k1, dim3(240), dim3(256), 32768, 0, p1);
    hipLaunchKernelGGL(k2, dim3(240), dim3(256), 65536, 0, p2);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(ref_slab.data(), slab2, ref_slab.size() * 4, hipMemcpyDeviceToHost));
    for (int prefetch = 0; prefetch < 2; ++prefetch) {
        CK(hipMemset(act, 0, (size_t)M * H * 2)); CK(hipMemset(slab2, 0, (size_t)3 * M * D * 4));
        pa.p1 = p1; pa.p2 = p2; pa.epoch = ++epoch; pa.prefetch = prefetch;
        hipLaunchKernelGGL(pair_kernel, dim3(240), dim3(256), 65536, 0, pa);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(got.data(), slab2, got.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        double s = 0;
        for (size_t i = 0; i < got.size(); ++i) { bad += got[i] != ref_slab[i]; s += ref_slab[i]; }
        printf("one launch (prefetch %d): w2 slabs vs two launches: %zu of %zu words differ (checksum %.6g)\n", prefetch, bad, got.size(), s);
    }
    auto timeit = [&](int mode) {                                            // 0 two launches, 1 one launch, 2 one launch + prefetch
        const int reps = 60;                                                 // back to back on the stream, as a graph replay issues them
        auto burst = [&](int n) {
            for (int r = 0; r < n; ++r) {
                mk(r % rot, p1, p2);
                if (mode == 0) {
                    hipLaunchKernelGGL(k1, dim3(240), dim3(256), 32768, 0, p1);
                    hipLaunchKernelGGL(k2, dim3(240), dim3(256), 65536, 0, p2);
                } else {
                    pa.p1 = p1; pa.p2 = p2; pa.epoch = ++epoch; pa.prefetch = mode == 2;
                    hipLaunchKernelGGL(pair_kernel, dim3(240), dim3(256), 65536, 0, pa);
                }
            }
        };
        burst(4);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        burst(reps);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / reps * 1e3;
    };
    for (int round = 0; round < 2; ++round) {
        const double t0 = timeit(0), t1 = timeit(1), t2 = timeit(2);
        printf("w1 -> w2 (236 MB of weights): two launches %.1f us | one launch + grid barrier %.1f us | + w2 weight prefetch before the barrier %.1f us\n", t0, t1, t2);
    }
    return 0;
}
