// Micro-benchmark (measurement tool, not product): what is this box's SUSTAINED dense bf16 MFMA rate, and what clock does the chip hold
// while it delivers it?  VERDICT r05 item 7: every >= 512-row GEMM of the repo sits at mfma_busy x effective clock ~= 1.0-1.1 GHz whatever
// the kernel -- a power / clock governor, or operand starvation?  Legs (one 256- or 512-thread workgroup per CU, all 256 CUs):
//   regs      v_mfma_f32_32x32x16_bf16 back to back on 8 independent accumulators, operands never change: no LDS, no memory
//   lds       the tiled GEMM's operand diet: per 8 MFMAs 6 ds_read_b128 (4 A + 2 W fragments), lane-linear, conflict free
//   lds+glb   the same plus the tile kernel's global traffic (4 x 16 B loads per wave and 32-deep stage, L2-resident source)
// Each leg runs ~`ms` of wall time per launch, `reps` launches back to back; per launch: TFLOP/s from HIP events, shader clock from
// s_memtime / s_memrealtime (100 MHz) deltas of wave 0 of every workgroup (median).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma.hip -o tools/probe_mfma && tools/probe_mfma
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

struct Args { int iters; const u32x4* src; unsigned long long* clk; float* sink; int random; };

__device__ __forceinline__ unsigned rnd_bf16x2(unsigned& st) {       // two bf16 in (-0.5, 0.5): xorshift bits under a fixed exponent
    st ^= st << 13; st ^= st >> 17; st ^= st << 5;
    const unsigned m0 = st & 0x7f, m1 = (st >> 7) & 0x7f, s0 = (st >> 14) & 1, s1 = (st >> 15) & 1, e0 = 0x7a + ((st >> 16) & 3), e1 = 0x7a + ((st >> 18) & 3);
    return ((s0 << 15) | (e0 << 7) | m0) | (((s1 << 15) | (e1 << 7) | m1) << 16);
}

// MODE 0 regs: operands never change.  MODE 1: LDS-fed, wave tile 128 x 64 (8 accumulators, 6 ds_read_b128 per 8 MFMAs -- the tiled GEMM's
// diet), fragments of step i + 1 read while step i multiplies.  MODE 2: LDS-fed, wave tile 128 x 128 (16 accumulators = 256 registers, 8 reads
// per 16 MFMAs), same pipelining; one wave per SIMD only.
template <int MODE, int NT>
__global__ __launch_bounds__(NT) void mfma_kernel(Args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* const lds = reinterpret_cast<u32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned st = 0x9E3779B9u * (blockIdx.x * NT + tid + 1);
    for (int i = tid; i < 64 * 64; i += NT)                      // 64 KiB: 64 fragment chunks
        lds[i] = a.random ? (u32x4){rnd_bf16x2(st), rnd_bf16x2(st), rnd_bf16x2(st), rnd_bf16x2(st)} : (u32x4){0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};
    __syncthreads();
    constexpr int NA = 4, NB = (MODE == 2) ? 4 : 2;
    f32x16 acc[NA * NB];
#pragma unroll
    for (int m = 0; m < NA * NB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    u32x4 af[2][NA], wf[2][NB];
    auto rd = [&](int buf, int it) {
        const int base = (it * 7 + wave * 3) & 31;
#pragma unroll
        for (int m = 0; m < NA; ++m) af[buf][m] = lds[((base + m) & 31) * 64 + lane];
#pragma unroll
        for (int n = 0; n < NB; ++n) wf[buf][n] = lds[(32 + ((base + n * 5) & 31)) * 64 + lane];
    };
    rd(0, 0);
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < a.iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (MODE >= 1) rd(h ^ 1, it + h + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < NA; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[m * NB + n] = mfma(af[MODE ? h : 0][m], wf[MODE ? h : 0][n], acc[m * NB + n]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < NA * NB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[m][r];
    if (s == 123.456f) a.sink[0] = s;
    if (tid == 0) { a.clk[blockIdx.x * 2] = c1 - c0; a.clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int MODE, int NT>
static void run(const char* name, int iters, int reps, int random, unsigned long long* clk, float* sink, int blocks) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Args a{iters, nullptr, clk, sink, random};
    CK(hipFuncSetAttribute((const void*)mfma_kernel<MODE, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    hipLaunchKernelGGL((mfma_kernel<MODE, NT>), dim3(blocks), dim3(NT), 64 * 1024, 0, a);   // warm
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(blocks * 2);
    constexpr int per_it = 4 * ((MODE == 2) ? 4 : 2);
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((mfma_kernel<MODE, NT>), dim3(blocks), dim3(NT), 64 * 1024, 0, a);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), clk, blocks * 16, hipMemcpyDeviceToHost));
        std::vector<double> ghz;
        for (int b = 0; b < blocks; ++b) ghz.push_back((double)h[b * 2] / ((double)h[b * 2 + 1] * 10.0));   // cycles per ns (realtime: 100 MHz)
        std::sort(ghz.begin(), ghz.end());
        const double flop = (double)blocks * (NT / 64) * iters * per_it * 32.0 * 32 * 16 * 2;
        const double cyc_per_mfma = (double)h[blocks & ~1] / ((double)iters * per_it) * (NT / 256);   // SIMD cycles per MFMA issued on that SIMD
        if (r == 0 || r == reps - 1)
            printf("%-34s %-6s waves/SIMD %d  rep %3d  %7.3f ms  %7.1f TFLOP/s (%.3f of 2500)  clock %.2f GHz (%.2f..%.2f)  %5.1f cyc/MFMA/SIMD\n",
                   name, random ? "random" : "const", NT / 256, r, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 2500.0, ghz[blocks / 2], ghz.front(), ghz.back(), cyc_per_mfma);
    }
}

int main(int argc, char** argv) {
    const int reps = argc > 1 ? atoi(argv[1]) : 60;
    int dev_cus = 0;
    CK(hipDeviceGetAttribute(&dev_cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("CUs: %d   (each leg: %d launches of ~5-8 ms back to back; first and last shown)\n", dev_cus, reps);
    unsigned long long* clk; float* sink;
    CK(hipMalloc(&clk, 256 * 16)); CK(hipMalloc(&sink, 16));
    for (int random = 0; random < 2; ++random) {
        run<0, 256>("regs (no operand traffic)", 40000, reps, random, clk, sink, dev_cus);
        run<0, 512>("regs (no operand traffic)", 20000, reps, random, clk, sink, dev_cus);
        run<1, 256>("lds 128x64 wave tile (6 rd / 8)", 40000, reps, random, clk, sink, dev_cus);
        run<1, 512>("lds 128x64 wave tile (6 rd / 8)", 20000, reps, random, clk, sink, dev_cus);
        run<2, 256>("lds 128x128 wave tile (8 rd / 16)", 20000, reps, random, clk, sink, dev_cus);
    }
    return 0;
}
