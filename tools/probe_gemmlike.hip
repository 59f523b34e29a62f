// Micro-benchmark (measurement tool, not product): starting from a free-running mixed stream (W from HBM + A from L2, which
// reaches ~5.9 TB/s of W beside an equal A stream), add the 128-row GEMM's structure one piece at a time and see which piece costs
// the bandwidth: fewer waves, the per-stage barrier, the panel-strided W pattern, the LDS staging of A, the fragment reads, the MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_gemmlike.hip -o tools/probe_gemmlike && tools/probe_gemmlike
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
    const u32x4* W;        // "weights": N/32 panels x (K/16) k-steps x 64 lanes x 16 B, panel-major (the product layout)
    const u32x4* A;        // activations: K/64 stages x 1024 units (16 KiB per stage: 128 rows)
    int nst;               // 64-deep K stages per workgroup (K / 64 / S)
    int S;                 // K slices
    size_t PS;             // panel stride in 16 B units
    int flags;             // 1 barrier per stage | 2 A through LDS (ds_write) | 4 fragment reads | 8 MFMAs | 16 W contiguous per workgroup
    int apol;              // cache policy of the A loads: 0 plain global load, else raw buffer load with aux = apol - 1 (1 sc0, 2 nt, 16 sc1)
    unsigned* sink;
};

// NW waves; each wave owns one 32-column panel: per stage 4 W loads (4 KiB) and 1024 / (NW * 64) A loads
template <int NW, int R, int APOL = 0>
__global__ __launch_bounds__(NW * 64) void gl_kernel(Args a) {
    constexpr int XL = 1024 / (NW * 64);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* lds = reinterpret_cast<u32x4*>(smem);          // 2 x 1024 units
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int s = blockIdx.x % a.S, nt = blockIdx.x / a.S;
    const u32x4* Wp;
    size_t wst;                                            // W stage stride (units)
    if (a.flags & 16) { Wp = a.W + ((size_t)blockIdx.x * a.nst) * (NW * 256) + wave * 256 + lane; wst = NW * 256; }
    else { Wp = a.W + (size_t)(nt * NW + wave) * a.PS + (size_t)s * a.nst * 256 + lane; wst = 256; }
    const u32x4* Ap = a.A + (size_t)s * a.nst * 1024 + tid;
    const __amdgpu_buffer_rsrc_t arsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(a.A), 0, 128 * 5120 * 2, 0x00020000);
    u32x4 w[R][4], x[R][XL];
    f32x16 acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    u32x4 fold = {0, 0, 0, 0};
    auto ld = [&](int slot, int st) {
        st = st < a.nst ? st : a.nst - 1;
#pragma unroll
        for (int j = 0; j < XL; ++j) {
            const unsigned off = (unsigned)(((size_t)s * a.nst * 1024 + tid + (size_t)st * 1024 + j * NW * 64) * 16);
            if (APOL == 0) x[slot][j] = Ap[(size_t)st * 1024 + j * NW * 64];
            else x[slot][j] = __builtin_amdgcn_raw_buffer_load_b128(arsrc, off, 0, APOL - 1);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) w[slot][j] = __builtin_nontemporal_load(Wp + (size_t)st * wst + j * 64);
    };
#pragma unroll
    for (int r = 0; r < R; ++r) ld(r, r);
    auto phase = [&](auto SL, int j) {
        constexpr int slot = decltype(SL)::value;
        u32x4* buf = lds + (j & 1) * 1024;
        if (a.flags & 2) {
#pragma unroll
            for (int q = 0; q < XL; ++q) buf[tid + q * NW * 64] = x[slot][q];
        } else {
#pragma unroll
            for (int q = 0; q < XL; ++q) fold ^= x[slot][q];
        }
        if (a.flags & 1) __syncthreads();
        if (a.flags & 4) {
            u32x4 xf[4][4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int m = 0; m < 4; ++m) xf[kk][m] = buf[(kk * 4 + m) * 64 + lane];
            if (a.flags & 8) {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xf[kk][m]), __builtin_bit_cast(bf16x8, w[slot][kk]), acc[m], 0, 0, 0);
            } else {
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int m = 0; m < 4; ++m) fold ^= xf[kk][m];
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) fold ^= w[slot][kk];
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) fold ^= w[slot][kk];
        }
        ld(slot, j + R);
    };
    int j = 0;
    for (; j + R <= a.nst; j += R) {
        if constexpr (R >= 1) phase(std::integral_constant<int, 0>{}, j);
        if constexpr (R >= 2) phase(std::integral_constant<int, 1>{}, j + 1);
        if constexpr (R >= 3) phase(std::integral_constant<int, 2>{}, j + 2);
        if constexpr (R >= 4) phase(std::integral_constant<int, 3>{}, j + 3);
    }
    unsigned r = fold[0] ^ fold[1] ^ fold[2] ^ fold[3];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int q = 0; q < 16; ++q) r ^= __float_as_uint(acc[m][q]);
    if (r == 0x12345678u) a.sink[0] = r;
}

int main() {
    const int N = 15360, K = 5120;
    const size_t WB = (size_t)N * K * 2;
    const int rot = 6;
    char *W, *A;
    unsigned* sink;
    CK(hipMalloc(&W, WB * rot));
    CK(hipMemset(W, 0x3c, WB * rot));
    CK(hipMalloc(&A, (size_t)128 * K * 2));
    CK(hipMemset(A, 0x3c, (size_t)128 * K * 2));
    CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* tag, auto kern, int NW, int S, int flags) {
        const int nst = K / 64 / S;
        const int blocks = N / 32 / NW * S;
        Args a{nullptr, (const u32x4*)A, nst, S, (size_t)(K / 16) * 64, flags, 0, sink};
        const int reps = 12;
        float sum = 0.f, best = 1e9f;
        for (int r = 0; r < reps + 2; ++r) {
            a.W = (const u32x4*)(W + (size_t)(r % rot) * WB);
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(NW * 64), 32768, 0, a);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        const double us = sum / reps * 1e3;
        printf("%-64s blocks %3d  %7.1f us (best %6.1f)  W %6.0f GB/s\n", tag, blocks, us, best * 1e3, (double)WB / us / 1e3);
        fflush(stdout);
    };
    // 4 waves x 128 columns x 2 K slices = 240 workgroups (the product's qkv form)
    run("4w R3 free-running, W contiguous per WG", gl_kernel<4, 3>, 4, 2, 16);
    run("4w R3 free-running, W panel-strided", gl_kernel<4, 3>, 4, 2, 0);
    run("4w R3 + barrier", gl_kernel<4, 3>, 4, 2, 1);
    run("4w R3 + barrier + A via LDS", gl_kernel<4, 3>, 4, 2, 1 | 2);
    run("4w R3 + barrier + A via LDS + frag reads", gl_kernel<4, 3>, 4, 2, 1 | 2 | 4);
    run("4w R3 + barrier + A via LDS + frag reads + MFMA", gl_kernel<4, 3>, 4, 2, 1 | 2 | 4 | 8);
    run("4w R4 + barrier + A via LDS + frag reads + MFMA", gl_kernel<4, 4>, 4, 2, 1 | 2 | 4 | 8);
    run("4w R2 + barrier + A via LDS + frag reads + MFMA", gl_kernel<4, 2>, 4, 2, 1 | 2 | 4 | 8);
    run("4w R3 all, W contiguous per WG", gl_kernel<4, 3>, 4, 2, 1 | 2 | 4 | 8 | 16);
    run("4w R3 free-running, A buffer_load aux 0", gl_kernel<4, 3, 1>, 4, 2, 0);
    run("4w R3 free-running, A sc0", gl_kernel<4, 3, 2>, 4, 2, 0);
    run("4w R3 free-running, A nt", gl_kernel<4, 3, 3>, 4, 2, 0);
    run("4w R3 free-running, A sc1", gl_kernel<4, 3, 17>, 4, 2, 0);
    run("4w R3 free-running, A sc0 sc1", gl_kernel<4, 3, 18>, 4, 2, 0);
    run("4w R3 all, A nt", gl_kernel<4, 3, 3>, 4, 2, 1 | 2 | 4 | 8);
    run("4w R3 all, A sc1", gl_kernel<4, 3, 17>, 4, 2, 1 | 2 | 4 | 8);
    // 8 waves x 256 columns x 4 K slices = 240 workgroups
    run("8w R2 all, A nt", gl_kernel<8, 2, 3>, 8, 4, 1 | 2 | 4 | 8);
    run("8w R2 all, A sc1", gl_kernel<8, 2, 17>, 8, 4, 1 | 2 | 4 | 8);
    run("8w R2 free-running, W panel-strided", gl_kernel<8, 2>, 8, 4, 0);
    run("8w R2 + barrier + A via LDS + frag reads + MFMA", gl_kernel<8, 2>, 8, 4, 1 | 2 | 4 | 8);
    // 8 waves x 256 columns x 2 K slices = 120 workgroups (per-CU rate with half the chip)
    run("8w R2 free-running, 120 workgroups", gl_kernel<8, 2>, 8, 2, 0);
    run("8w R2 all, 120 workgroups", gl_kernel<8, 2>, 8, 2, 1 | 2 | 4 | 8);
    // 8 waves x 256 columns x 8 / 6 K slices (480 / 360 workgroups: two per CU where they fit)
    run("8w R2 all, 480 workgroups", gl_kernel<8, 2>, 8, 8, 1 | 2 | 4 | 8);
    run("4w R3 all, 480 workgroups (S = 4)", gl_kernel<4, 3>, 4, 4, 1 | 2 | 4 | 8);
    return 0;
}
