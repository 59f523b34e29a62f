// Micro-benchmark (measurement tool, not product): what limits a CU that streams weights from HBM while it re-reads a small
// activation operand from L2?  One 512-thread workgroup per CU; waves [0, hw) stream a private HBM region with 16 B/lane
// non-temporal loads, waves [hw, 8) re-read a shared 1.25 MiB buffer (the A operand of a 128-row GEMM) in 16 KiB stages,
// either all workgroups in lockstep (the GEMM's access pattern) or each workgroup rotated to its own starting stage.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mix.hip -o tools/probe_mix && tools/probe_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args {
    const u32x4* hbm;      // streamed region, n_hbm16 units of 16 B per workgroup
    size_t n_hbm16;
    const u32x4* l2;       // shared buffer, nst stages of 1024 units (16 KiB)
    int nst;               // stages in the shared buffer
    int l2_stages;         // stages each workgroup reads (loops over the buffer)
    int hw;                // waves that stream HBM (the other 8 - hw re-read L2)
    int rotate;            // 1: workgroup b starts at stage (b / 8 * 11) % nst
    int same_wave;         // 1: every wave does both (alternating), like the GEMM loop
    unsigned* sink;
};

__global__ __launch_bounds__(512) void mix_kernel(Args a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32x4 acc = {0, 0, 0, 0};
    const int rot = a.rotate ? (int)(((blockIdx.x >> 3) * 11) % a.nst) : 0;
    if (a.same_wave) {
        // every wave: per "stage" 4 HBM loads (4 KiB per wave) + 2 L2 loads (its eighth of the 16 KiB stage), 3 stages in flight
        const u32x4* h = a.hbm + (size_t)blockIdx.x * a.n_hbm16 + wave * 64 + lane;
        const size_t nstage = a.n_hbm16 / (8 * 256);         // 8 waves x 4 x 64 units per stage
        for (size_t s = 0; s < nstage; s += 2) {
            u32x4 v[12];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const size_t ss = s + q;
                int st = (int)((ss + rot) % a.nst);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[q * 6 + j] = __builtin_nontemporal_load(h + ss * 2048 + j * 512);
                if (a.l2_stages) {
                    v[q * 6 + 4] = a.l2[(size_t)st * 1024 + wave * 128 + lane];
                    v[q * 6 + 5] = a.l2[(size_t)st * 1024 + wave * 128 + 64 + lane];
                } else { v[q * 6 + 4] = v[q * 6 + 5] = (u32x4){0, 0, 0, 0}; }
            }
#pragma unroll
            for (int j = 0; j < 12; ++j) acc ^= v[j];
        }
    } else if (wave < a.hw) {
        const u32x4* h = a.hbm + (size_t)blockIdx.x * a.n_hbm16 + wave * 64 + lane;
        const size_t stride = (size_t)a.hw * 64;
        const size_t n = a.n_hbm16;
        size_t i = 0;
        for (; i + 7 * stride + wave * 64 + lane < n; i += 8 * stride) {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(h + i + j * stride);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc ^= v[j];
        }
    } else {
        const int lw = wave - a.hw, nl = 8 - a.hw;            // this wave's share of each 16 KiB stage: 1024 / nl units
        const int per = 1024 / nl / 64;                       // loads per lane per stage (nl in {1, 2, 4, 8}: 16, 8, 4, 2)
        const int psh = 31 - __builtin_clz(per);
        const int total = a.l2_stages * per;                  // multiple of 8 for every configuration used below
        const u32x4* base = a.l2 + lw * (1024 / nl) + lane;
        for (int i = 0; i < total; i += 8) {
            u32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int idx = i + j;
                int st = (idx >> psh) + rot;
                st %= a.nst;
                v[j] = base[(size_t)st * 1024 + (idx & (per - 1)) * 64];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc ^= v[j];
        }
    }
    const unsigned r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (r == 0x12345678u) a.sink[0] = r;
}

// Round 5: the L2-only leg with more requests in flight.  Every one of the workgroup's 8 waves re-reads its eighth (2 x 64 units) of
// each 16 KiB stage of the shared buffer with INF 16 B loads outstanding per lane (INF / 2 stages at once): 8 waves x 12 loads x 1 KiB
// = 96 KiB in flight per workgroup; launched with 1 or 2 workgroups per CU.  Is the 11.7 - 15.3 TB/s of the 4-wave x 8-load leg above
// a limit of the L2 -> CU path, or of the concurrency that leg offered?
template <int INF>
__global__ __launch_bounds__(512) void l2deep_kernel(Args a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    u32x4 acc = {0, 0, 0, 0};
    const int rot = a.rotate ? (int)(((blockIdx.x >> 3) * 11) % a.nst) : 0;
    const u32x4* base = a.l2 + wave * 128 + lane;
    constexpr int SPI = INF / 2;                               // stages per iteration
    for (int s0 = 0; s0 + SPI <= a.l2_stages; s0 += SPI) {
        u32x4 v[INF];
#pragma unroll
        for (int q = 0; q < SPI; ++q) {
            const int st = (s0 + q + rot) % a.nst;
            v[2 * q] = base[(size_t)st * 1024];
            v[2 * q + 1] = base[(size_t)st * 1024 + 64];
        }
#pragma unroll
        for (int j = 0; j < INF; ++j) acc ^= v[j];
    }
    const unsigned r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (r == 0x12345678u) a.sink[0] = r;
}

int main() {
    const size_t HB = (size_t)3 << 30;                         // streamed pool, rotated so the Infinity Cache never helps
    char *hbm, *l2;
    unsigned* sink;
    CK(hipMalloc(&hbm, HB));
    CK(hipMemset(hbm, 1, HB));
    const int nst = 80;                                        // 80 x 16 KiB = 1.25 MiB = 128 rows x 5120 bf16
    CK(hipMalloc(&l2, (size_t)nst * 16384));
    CK(hipMemset(l2, 2, (size_t)nst * 16384));
    CK(hipMalloc(&sink, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int blocks = 256;
    const size_t per_block = 655360;                           // bytes of HBM per workgroup per launch (157 MB / 240)
    auto run = [&](const char* tag, int hw, size_t hbm_bytes, int l2_stages, int rotate, int same_wave) {
        Args a{nullptr, hbm_bytes / 16, (const u32x4*)l2, nst, l2_stages, hw, rotate, same_wave, sink};
        const int reps = 12;
        float best = 1e9f, sum = 0.f;
        for (int r = 0; r < reps + 2; ++r) {
            a.hbm = (const u32x4*)(hbm + ((size_t)r * blocks * per_block) % (HB - blocks * per_block));
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(mix_kernel, dim3(blocks), dim3(512), 0, 0, a);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        const double us = sum / reps * 1e3;
        const double hb = same_wave || hw > 0 ? (double)blocks * hbm_bytes : 0.0, lb = (double)blocks * l2_stages * 16384.0;
        printf("%-46s %7.1f us (best %6.1f)  HBM %6.0f GB/s  L2 %6.0f GB/s  per CU %5.1f + %5.1f GB/s\n", tag, us, best * 1e3,
               hb / us / 1e3, lb / us / 1e3, hb / us / 1e3 / blocks, lb / us / 1e3 / blocks);
        fflush(stdout);
    };
    const int LS = 40;                                         // 40 stages x 16 KiB = 655 KB of A per workgroup (A : W = 1 : 1)
    run("HBM only, 4 waves", 4, per_block, 0, 0, 0);
    run("HBM only, 8 waves", 8, per_block, 0, 0, 0);
    run("HBM only, 2 waves", 2, per_block, 0, 0, 0);
    run("L2 only, 4 waves lockstep", 4, 0, LS, 0, 0);
    run("L2 only, 4 waves rotated", 4, 0, LS, 1, 0);
    run("L2 only x4 work, 4 waves lockstep", 4, 0, 4 * LS, 0, 0);
    run("L2 only x4 work, 4 waves rotated", 4, 0, 4 * LS, 1, 0);
    run("HBM 4 waves + L2 4 waves lockstep", 4, per_block, LS, 0, 0);
    run("HBM 4 waves + L2 4 waves rotated", 4, per_block, LS, 1, 0);
    run("HBM 6 waves + L2 2 waves lockstep", 6, per_block, LS, 0, 0);
    run("HBM 6 waves + L2 2 waves rotated", 6, per_block, LS, 1, 0);
    run("HBM 7 waves + L2 1 wave rotated", 7, per_block, LS, 1, 0);
    run("HBM 4 waves + L2 4 waves rotated, A:W 2:1", 4, per_block, 2 * LS, 1, 0);
    run("same wave, HBM only", 8, per_block, 0, 0, 1);
    run("same wave, HBM + L2 lockstep", 8, per_block, LS, 0, 1);
    run("same wave, HBM + L2 rotated", 8, per_block, LS, 1, 1);
    // ---- round 5: deep L2-only legs
    auto run_deep = [&](const char* tag, int inf, int nblocks, int l2_stages, int rotate) {
        Args a{nullptr, 0, (const u32x4*)l2, nst, l2_stages, 0, rotate, 0, sink};
        const int reps = 12;
        float best = 1e9f, sum = 0.f;
        for (int r = 0; r < reps + 2; ++r) {
            CK(hipEventRecord(e0, 0));
            if (inf == 8) hipLaunchKernelGGL(l2deep_kernel<8>, dim3(nblocks), dim3(512), 0, 0, a);
            else if (inf == 12) hipLaunchKernelGGL(l2deep_kernel<12>, dim3(nblocks), dim3(512), 0, 0, a);
            else hipLaunchKernelGGL(l2deep_kernel<16>, dim3(nblocks), dim3(512), 0, 0, a);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        const double us = sum / reps * 1e3, lb = (double)nblocks * l2_stages * 16384.0;
        printf("%-58s %7.1f us (best %6.1f)  L2 %6.0f GB/s (best %6.0f)  per CU %5.1f GB/s\n", tag, us, best * 1e3, lb / us / 1e3, lb / (best * 1e3) / 1e3,
               lb / us / 1e3 / 256);
        fflush(stdout);
    };
    const int DS = 480;                                        // 480 stages x 16 KiB = 7.9 MB of re-reads per workgroup: long enough to leave the launch ramp behind
    run_deep("L2 only, 8 waves x  8 loads, 1 WG/CU, lockstep", 8, 256, DS, 0);
    run_deep("L2 only, 8 waves x 12 loads, 1 WG/CU, lockstep", 12, 256, DS, 0);
    run_deep("L2 only, 8 waves x 12 loads, 1 WG/CU, rotated", 12, 256, DS, 1);
    run_deep("L2 only, 8 waves x 16 loads, 1 WG/CU, rotated", 16, 256, DS, 1);
    run_deep("L2 only, 8 waves x 12 loads, 2 WG/CU, lockstep", 12, 512, DS, 0);
    run_deep("L2 only, 8 waves x 12 loads, 2 WG/CU, rotated", 12, 512, DS, 1);
    run_deep("L2 only, 8 waves x 16 loads, 2 WG/CU, rotated", 16, 512, DS, 1);
    run_deep("L2 only, 8 waves x 12 loads, 2 WG/CU, rotated, short (40 st)", 12, 512, 40 / 6 * 6, 1);
    return 0;
}
