// Measurement tool (not product): what would overlapping DEPENDENT launches buy on this chip and runtime?  
//
// The launch anatomy (profiles/r06_launch_anatomy.log) says a 128-row GEMM launch spends ~37 % of its span outside its K loop: the dependent
// boundary (2.1 us), the ramp to the first landed weight stage (2.3-2.8 us) and a straggler tail (2-7 us) -- all serialised, although launch
// k + 1 depends on launch k only through its ACTIVATION operand.  This probe prices the alternative on a skeleton of the head's chain:
//   R  row-kernel-like   128 workgroups x 640 threads: reads 3 fp32 slabs of its row, two block reductions, writes one bf16 row
//   G  GEMM-like         240 workgroups x 256 threads: streams its slice of a 157 MB weight matrix (16 B non-temporal loads, 3 stages of
//                        4 KB per wave in flight) and re-reads the 1.3 MB operand R wrote; parks a 32 KB fp32 slab per workgroup
// chain R -> G -> R -> G ... as one hipGraph, in two forms:
//   serial    every node depends on its predecessor (what the engine captures today); plain stores
//   overlap   node n depends on node n - 2 only; the dependence on node n - 1 is in-kernel: the producer stores write-through (sc0 sc1) and
//             raises per-row flags / a done counter after s_waitcnt vmcnt(0); the consumer first requests what does not depend on the
//             producer (G: its first weight stages), then polls, invalidates (buffer_inv sc0 sc1) and loads the operand
// Prints us per (R, G) pair for both forms and checks a checksum of the last operand (the dependence is real: a wrong order changes it).
//   hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.hip -o tools/overlap_probe && tools/overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ROWS = 128, D = 5120, NWG_G = 240;
constexpr size_t W_BYTES = (size_t)15360 * 5120 * 2;          // 157 MB

struct RArgs { const float* slabs; unsigned short* out; int* rowflag; const int* done_prev; int done_target; int epoch; int overlap; float bias; int naps; };
struct GArgs { const u32x4* W; const u32x4* A; float* slabs; const int* rowflag; int* done; int epoch; int overlap; unsigned* sink; int naps; };

__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(void* p, size_t bytes) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, (int)bytes, 0x00020000); }

// one workgroup per row: 640 threads x 8 columns
__global__ __launch_bounds__(640) void row_kernel(RArgs a) {
    __shared__ float red[16];
    __shared__ int ok;
    const int m = blockIdx.x, d0 = threadIdx.x * 8;
    if (a.overlap && a.done_prev) {                            // the producing G launch: every workgroup has drained its slab
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(a.done_prev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < a.done_target && wall_clock64() - t0 < 100000000LL) for (int q = 0; q < a.naps; ++q) __builtin_amdgcn_s_sleep(16);
            ok = 1;
        }
        __syncthreads();
        asm volatile("buffer_inv sc0 sc1" ::: "memory");
    }
    float x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const float4* p = reinterpret_cast<const float4*>(a.slabs + ((size_t)s * ROWS + m) * D + d0);
        const float4 u = p[0], v = p[1];
        x[0] += u.x; x[1] += u.y; x[2] += u.z; x[3] += u.w; x[4] += v.x; x[5] += v.y; x[6] += v.z; x[7] += v.w;
    }
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) ss += x[j] * x[j];
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < 10; ++i) t += red[i];
    const float rs = rsqrtf(t / D + 1e-6f);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned lo = __float_as_uint(x[2 * j] * rs + a.bias) >> 16, hi = __float_as_uint(x[2 * j + 1] * rs + a.bias) >> 16;
        o[j] = lo | (hi << 16);
    }
    if (a.overlap) {
        __builtin_amdgcn_raw_buffer_store_b128(o, rsrc(a.out, (size_t)ROWS * D * 2), (unsigned)(((size_t)m * D + d0) * 2), 0, 17);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(a.rowflag + m, a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *reinterpret_cast<u32x4*>(a.out + (size_t)m * D + d0) = o;
    }
}

// 240 workgroups x 4 waves; wave streams W in 4 KB chunks, 3 in flight, and re-reads the operand
__global__ __launch_bounds__(256) void gemm_like_kernel(GArgs g) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t per_wg = W_BYTES / 16 / NWG_G;                // 16 B units per workgroup
    const u32x4* w = g.W + (size_t)blockIdx.x * per_wg + wave * 64 + lane;
    const int nst = (int)(per_wg / (4 * 256));                  // stages of 4 x 16 B per lane per wave
    u32x4 r[3][4], acc = {0, 0, 0, 0};
    auto ldw = [&](u32x4(&x)[4], int st) {
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = __builtin_nontemporal_load(w + (size_t)st * 1024 + j * 256);
    };
    ldw(r[0], 0); ldw(r[1], 1); ldw(r[2], 2);                    // what does not depend on the producer
    if (g.overlap) {
        if (tid < 64) {                                          // ONE wave polls: two flags per lane
            const long long t0 = wall_clock64();
            while ((__hip_atomic_load(g.rowflag + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.epoch ||
                    __hip_atomic_load(g.rowflag + 64 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < g.epoch) && wall_clock64() - t0 < 100000000LL)
                for (int q = 0; q < g.naps; ++q) __builtin_amdgcn_s_sleep(16);
        }
        if (tid < 64) asm volatile("buffer_inv sc0 sc1" ::: "memory");
        __syncthreads();
    }
    const int na = ROWS * D * 2 / 16 / 256;                      // operand units per thread (the whole operand per workgroup)
    for (int st = 0; st < nst; st += 3) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
#pragma unroll
            for (int j = 0; j < 4; ++j) acc ^= r[q][j];
            if (st + q + 3 < nst) ldw(r[q], st + q + 3);
            const int ia = ((st + q) * 4) % na;                   // 4 operand loads per stage (L2)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc ^= g.A[(size_t)((ia + j) % na) * 256 + tid];
        }
    }
    // park a 32 KB slab per workgroup (3 slabs x 128 rows x 5120 fp32 = 7.8 MB over 240 workgroups): 8 x 16 B per thread
    const float v = (float)((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) & 0xff) * (1.0f / 256.0f);
    const size_t base = (size_t)blockIdx.x * 8192;               // floats
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const size_t off = base + (size_t)j * 1024 + tid * 4;
        if (off + 4 <= (size_t)3 * ROWS * D) {
            const u32x4 o = {__float_as_uint(v + j), __float_as_uint(v), __float_as_uint(v), __float_as_uint(v)};
            if (g.overlap) __builtin_amdgcn_raw_buffer_store_b128(o, rsrc(g.slabs, (size_t)3 * ROWS * D * 4), (unsigned)(off * 4), 0, 17);
            else *reinterpret_cast<u32x4*>(g.slabs + off) = o;
        }
    }
    if (g.overlap) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(g.done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (acc[0] == 0x12345678u) g.sink[0] = acc[1];
}

static double run_graph(bool overlap, int naps, int pairs, int reps, int dep_gap, int eager, float* slabs, unsigned short* A, const u32x4* W, int* rowflag, int* done, unsigned* sink, unsigned* checksum) {
    CK(hipMemset(rowflag, 0, ROWS * 4));
    CK(hipMemset(done, 0, 4 * (pairs + 1)));
    CK(hipMemset(slabs, 0, (size_t)3 * ROWS * D * 4));
    hipGraph_t graph;
    CK(hipGraphCreate(&graph, 0));
    std::vector<hipGraphNode_t> nodes;
    std::vector<RArgs> ra(pairs);
    std::vector<GArgs> ga(pairs);
    std::vector<void*> keep;
    for (int p = 0; p < pairs; ++p) {
        ra[p] = RArgs{slabs, A, rowflag, p ? done + (p - 1) : nullptr, NWG_G, p + 1, overlap ? 1 : 0, 0.001f * p, naps};
        ga[p] = GArgs{W + (size_t)(p % 3) * (W_BYTES / 16), reinterpret_cast<const u32x4*>(A), slabs, rowflag, done + p, p + 1, overlap ? 1 : 0, sink, naps};
    }
    for (int n = 0; n < 2 * pairs; ++n) {
        hipKernelNodeParams kp = {};
        void** args = (void**)malloc(sizeof(void*));
        keep.push_back(args);
        if (n % 2 == 0) { args[0] = &ra[n / 2]; kp.func = (void*)row_kernel; kp.gridDim = dim3(ROWS); kp.blockDim = dim3(640); }
        else { args[0] = &ga[n / 2]; kp.func = (void*)gemm_like_kernel; kp.gridDim = dim3(NWG_G); kp.blockDim = dim3(256); }
        kp.kernelParams = args;
        hipGraphNode_t node;
        std::vector<hipGraphNode_t> deps;
        if (n >= dep_gap) deps.push_back(nodes[n - dep_gap]);
        CK(hipGraphAddKernelNode(&node, graph, deps.data(), deps.size(), &kp));
        nodes.push_back(node);
    }
    hipGraphExec_t exec;
    CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double best = 1e30;
    hipStream_t st2;
    CK(hipStreamCreate(&st2));
    hipEvent_t ej;
    CK(hipEventCreate(&ej));
    for (int r = 0; r < reps; ++r) {
        CK(hipMemsetAsync(rowflag, 0, ROWS * 4, st));
        CK(hipMemsetAsync(done, 0, 4 * (pairs + 1), st));
        CK(hipEventRecord(e0, st));
        if (!eager) CK(hipGraphLaunch(exec, st));
        else {                                                   // two plain streams: R chain on st, G chain on st2 (the flags order them)
            CK(hipEventRecord(ej, st)); CK(hipStreamWaitEvent(st2, ej, 0));
            for (int p = 0; p < pairs; ++p) {
                hipLaunchKernelGGL(row_kernel, dim3(ROWS), dim3(640), 0, st, ra[p]);
                hipLaunchKernelGGL(gemm_like_kernel, dim3(NWG_G), dim3(256), 0, st2, ga[p]);
            }
            CK(hipEventRecord(ej, st2)); CK(hipStreamWaitEvent(st, ej, 0));
        }
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    std::vector<unsigned short> h((size_t)ROWS * D);
    CK(hipMemcpy(h.data(), A, h.size() * 2, hipMemcpyDeviceToHost));
    unsigned cs = 0;
    for (size_t i = 0; i < h.size(); ++i) cs = cs * 31u + h[i];
    *checksum = cs;
    CK(hipGraphExecDestroy(exec)); CK(hipGraphDestroy(graph)); CK(hipStreamDestroy(st));
    for (void* p : keep) free(p);
    return best * 1e3 / pairs;
}

int main(int argc, char** argv) {
    const int pairs = argc > 1 ? atoi(argv[1]) : 48, reps = argc > 2 ? atoi(argv[2]) : 8;
    float* slabs; unsigned short* A; u32x4* W; int *rowflag, *done; unsigned* sink;
    CK(hipMalloc(&slabs, (size_t)3 * ROWS * D * 4));
    CK(hipMalloc(&A, (size_t)ROWS * D * 2));
    CK(hipMalloc(&W, 3 * W_BYTES));                              // three weight matrices in rotation: 472 MB > the 256 MB Infinity Cache
    CK(hipMemset(W, 1, 3 * W_BYTES));
    CK(hipMalloc(&rowflag, ROWS * 4)); CK(hipMalloc(&done, 4 * 4096)); CK(hipMalloc(&sink, 16));
    unsigned c0 = 0, c1 = 0;
    auto serial = [&]() { return run_graph(false, 1, pairs, reps, 1, 0, slabs, A, W, rowflag, done, sink, &c0); };
    printf("chain of %d (row kernel 128 x 640, GEMM-like 240 x 256 streaming 157 MB) pairs; serial hipGraph (node n after n-1, plain stores): %.2f us per pair\n", pairs, serial());
    double t;
    t = run_graph(true, 4, pairs, reps, 1, 0, slabs, A, W, rowflag, done, sink, &c1);
    printf("A  serial graph, flag-protocol kernels (write-through producers, poll + buffer_inv consumers): %.2f us per pair (serial plain %.2f)  checksum %s\n", t, serial(), c0 == c1 ? "equal" : "DIFFERENT");
    t = run_graph(false, 4, pairs, reps, 2, 0, slabs, A, W, rowflag, done, sink, &c1);
    printf("B  graph with node n after n-2 (two independent chains), PLAIN kernels (no flags: values meaningless): %.2f us per pair (serial plain %.2f)\n", t, serial());
    for (int naps : {1, 4, 16}) {
        t = run_graph(true, naps, pairs, reps, 2, 0, slabs, A, W, rowflag, done, sink, &c1);
        printf("C  graph with node n after n-2 + flag-protocol kernels, poll period %5d cycles: %.2f us per pair (serial plain %.2f)  checksum %s\n", naps * 1024, t, serial(), c0 == c1 ? "equal" : "DIFFERENT");
    }
    t = run_graph(true, 4, pairs, reps, 2, 1, slabs, A, W, rowflag, done, sink, &c1);
    printf("D  two plain streams, eager launches, flag-protocol kernels: %.2f us per pair (serial plain %.2f)  checksum %s\n", t, serial(), c0 == c1 ? "equal" : "DIFFERENT");
    return 0;
}
