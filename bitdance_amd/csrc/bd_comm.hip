// Tensor-parallel exchange for the BitDance step on one xGMI node: the all-reduce after every
// row-split Linear (head wo / w2, LLM o_proj / down_proj) as ONE hand-written kernel per exchange, captured inside the
// AR-step hipGraph with the GEMMs around it.
//
// One process per GPU.  Every rank owns one exported allocation {staging [tp][slice] fp32 | result [rows][N] bf16} and a
// small flag block; the peers' allocations are mapped with hipIpcOpenMemHandle (or handed over as plain pointers when
// the "ranks" are contexts of one process -- the single-GPU protocol test).  xGMI is point to point, so the exchange is
// two-shot and PUSH-only (posted writes, no read round trips over the links):
//
//   phase 1 (reduce-scatter by push)  rank r writes, for every peer p, the slice p of its fp32 partial [rows][N] into
//                                     p's staging area [r]                -> one link per peer, (tp-1)/tp of 4*rows*N B out
//   phase 2 (all-gather by push)      rank r sums the tp staged copies of slice r IN RANK ORDER (every rank computes each
//                                     element exactly once, so the replicated result is bit-identical everywhere), adds
//                                     the Linear's bias, rounds once to bf16 (what F.linear returns under autocast) and
//                                     writes the rows into every rank's result buffer
//
// Synchronisation is per (block b of rank r) <-> (block b of every peer): after its pushes a block drains its stores,
// fences at system scope and writes an epoch number into the peers' flag blocks; it polls its OWN flag block (remote
// writes, local polls).  Epochs only grow, so there is no reset and the same kernel node replays in a graph.  All remote
// traffic and all reads of remotely written memory are system-scope (sc0 sc1) accesses: the XCD L2s would otherwise serve
// stale lines.  Every spin is bounded by a wall-clock budget; on expiry the kernel sets an error word and returns, the
// host raises (bd_comm_error).
//
// RCCL (the library `torch.distributed`'s "nccl" backend drives) is used for bootstrap (handle exchange), for the
// once-per-image prefill, and as a drop-in fallback / baseline for this kernel: mode 1 calls ncclAllReduce on the fp32
// partial through a function pointer handed over by the host (bd_comm_set_rccl), also graph-capturable.
#include <cstdio>
#include <cstring>
#include <string>

#include "bd_common.h"
#include "bd_kernels.h"
#include "../../include/bitdance_hip.h"

#define BD_TP_MAX 8
#define BD_TP_GMAX 64

struct bd_comm {
    int rank = 0, size = 1, mode = 0;
    long long max_elems = 0;                 // rows * N capacity of one exchange
    long long gather_bytes = 0;              // capacity of the all-gather region behind the exchange buffers (bd_comm_create2)
    char* data = nullptr;                    // local: staging fp32 [max_elems] | result bf16 [max_elems] | gather region [gather_bytes]
    int* flags = nullptr;                    // local: A [tp][GMAX] | B [tp][GMAX] | err | epoch [GMAX] | C [tp][GMAX] | epoch of C [GMAX]
    bool own = false;
    bool data_uncached = false, flags_uncached = false;   // allocation kinds actually obtained (bd_comm_info): the cross-GPU
                                                          // coherence argument of the hand-written exchange needs both
    char* peer_data[BD_TP_MAX] = {};
    int* peer_flags[BD_TP_MAX] = {};
    bool ipc_open[BD_TP_MAX] = {};
    void* nccl_comm = nullptr;
    int (*nccl_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
    double timeout_s = 20.0;
    int fences = 0;                          // 1: system-scope fences around every flag (bd_comm_set_fences): belt and braces on a node whose
                                             // self-test fails without them; the payload is write-through sc0 sc1 stores drained by vmcnt(0) and
                                             // read back with sc0 sc1 loads, which needs no fence (MI355X_MICROARCH.md, hand-off recipe R1)
    long long n_exchanges = 0;               // launches issued (graph captures count once): reporting only
    long long n_gathers = 0;
    long long n_prepushed = 0;               // exchanges whose phase 1 ran in the producing GEMM's epilogue (tests assert the fusion engaged)
    int prepushed_next = 0;                  // set by bdk_tp_push_target: the next exchange finds its staging rows already pushed
    // sequence-parallel row kernels (bd_sp.hip): a second, CACHEABLE exported allocation where the peers land the bf16 operand rows the
    // consuming GEMM re-reads from L2 (an uncached buffer would make each of its 240 workgroups fetch the 1.3 MB operand from memory),
    // a small landing area for the final latent rows behind the gather region, and the flag block BD_SP_* behind the exchange's flags
    char* hbuf = nullptr;
    long long hbuf_bytes = 0, aux_bytes = 0;
    char* peer_hbuf[BD_TP_MAX] = {};
    bool ipc_open_h[BD_TP_MAX] = {};
    int sp_seq = 0;                          // host-side sequence number of the next hand-off since the last bdk_sp_begin
    // loop-back: ONE rank of a `size`-rank group alone on a GPU (tools/head_sweep.py --tp-shard): the peers' buffers are scratch copies,
    // every flag a peer would write is written locally -- the rank's launches, traffic and waits without the links
    int loopback = 0;
    char* scratch_data[BD_TP_MAX] = {};
    char* scratch_hbuf[BD_TP_MAX] = {};
};
#define BD_TP_FLAG_INTS (3 * BD_TP_MAX * BD_TP_GMAX + 1 + 2 * BD_TP_GMAX)
#define BD_SP_AUX_BYTES 65536

void bdk_set_error(const std::string& m);    // bd_api.hip

static int cfail(const std::string& m) { bdk_set_error(m); return -1; }

struct ArArgs {
    const float* part;          // this rank's partial [rows][N] fp32
    const bf16_t* bias;         // [N] or null
    char* peer_data[BD_TP_MAX];
    int* peer_flags[BD_TP_MAX];
    int* flags;                 // local flag block
    long long stage_bytes;      // offset of the result buffer inside a data allocation
    long long data_bytes;       // size of a data allocation (staging + result)
    long long timeout_ticks;    // wall_clock64 ticks (100 MHz)
    int rank, size, G, U, Us, N8;
    int fences = 0;
    int prepushed = 0;          // 1: the producing GEMM's epilogue already pushed every peer's slice into its staging row (bdk_tp_push_target)
    int loopback = 0;           // 1: no peers (bd_comm_set_loopback): the flag a peer t would write here is written by this rank itself
    BD_STAMP_FIELD
};

// The fence-less hand-off (payload: sc0 sc1 write-through stores drained by vmcnt(0); flag: relaxed system-scope store; reader: sc0 sc1
// loads) relies on gfx950's cache-policy semantics, not on a release / acquire edge of the memory model -- like the in-launch
// split-K reduction (bd_gemm_kernel.h) it must not be compiled for another target unnoticed.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "bd_comm.hip: the sc0 sc1 / vmcnt(0) flag hand-off is validated for gfx950 only"
#endif
// all of this block's pushes are at their destination, then the epoch goes to every peer's flag row of this rank
BD_DEV void tp_signal(const ArArgs& a, int base, int b, int e) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int t = bd_spread_lane(a.size);                     // one flag per wave: parallel fabric writes (bd_common.h)
    if (t >= 0 && t != a.rank) {
        if (a.fences) __threadfence_system();
        int* const dst = a.loopback ? a.flags + base + t * BD_TP_GMAX + b : a.peer_flags[t] + base + a.rank * BD_TP_GMAX + b;
        __hip_atomic_store(dst, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// returns false (block-uniform) when the exchange is dead: a wait of THIS rank ran out of its budget now or earlier, or a
// peer reported that one of its waits did (the error word travels: a rank that times out ORs its bit into every peer's error
// word as well, so all ranks stop pushing and all of them raise at the next host check instead of one raising while the
// others return a silently corrupted image)
BD_DEV bool tp_wait(const ArArgs& a, int base, int b, int e, int err_index) {
    __shared__ int alive_sh;
    if (threadIdx.x == 0) alive_sh = 1;
    __syncthreads();
    const int t = bd_spread_lane(a.size);
    if (t >= 0 && t != a.rank) {
        const int* f = a.flags + base + t * BD_TP_GMAX + b;
        const long long t0 = wall_clock64();
        bool dead = __hip_atomic_load(a.flags + err_index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
        while (!dead && bd_epoch_before(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), e)) {
            __builtin_amdgcn_s_sleep(2);
            if (wall_clock64() - t0 > a.timeout_ticks) {
                __hip_atomic_fetch_or(a.flags + err_index, 1 << t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                for (int p = 0; p < a.size; ++p)             // tell everyone: bit 8 + rank = "rank r gave up"
                    if (p != a.rank)
                        __hip_atomic_fetch_or(a.peer_flags[p] + err_index, 1 << (8 + a.rank), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                dead = true;
                break;
            }
            dead = __hip_atomic_load(a.flags + err_index, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
        }
        if (dead) alive_sh = 0;
        if (a.fences) __threadfence_system();
    }
    __syncthreads();
    return alive_sh != 0;
}

// 16 B system-scope (sc0 sc1) accesses through buffer instructions: the compiler tracks their vmcnt like any other load, and
// a 16 B write is one fabric transaction where the 8 B atomics of the first version were two (MI355X_MICROARCH: dwordx2 stores
// cost 2.7x the dwordx4 time per byte on the fabric)
#define BD_SYS_AUX 17                     /* cache policy bits: sc0 | sc1 */
BD_DEV __amdgpu_buffer_rsrc_t sys_rsrc(void* base, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
}

__global__ __launch_bounds__(256) void tp_allreduce_kernel(ArArgs a) {
    BD_KSTAMP(a.stamp, 0);
    const int b = blockIdx.x, tid = threadIdx.x;
    const int FA = 0, FB = BD_TP_MAX * BD_TP_GMAX, ERR = 2 * BD_TP_MAX * BD_TP_GMAX, EP = ERR + 1;
    const int e = (int)((unsigned)a.flags[EP + b] + 1u);     // only this block ever writes its epoch word (unsigned: wraps, bd_common.h)
    const int Ub = (a.Us + a.G - 1) / a.G;
    const int c0 = b * Ub, c1 = min(a.Us, c0 + Ub);
    // ---- phase 1: push my partial of every peer's slice into that peer's staging row [rank]
    for (int p = 0; p < a.size && !a.prepushed; ++p) {
        if (p == a.rank) continue;
        const __amdgpu_buffer_rsrc_t dst = sys_rsrc(a.peer_data[p], a.data_bytes);
        for (int u = c0 + tid; u < c1; u += 256) {
            const int gu = p * a.Us + u;
            if (gu >= a.U) break;
            const u32x4* src = reinterpret_cast<const u32x4*>(a.part + (size_t)gu * 8);
            const u32x4 v0 = src[0], v1 = src[1];
            const unsigned off = (unsigned)(((size_t)a.rank * a.Us + u) * 32);
            __builtin_amdgcn_raw_buffer_store_b128(v0, dst, off, 0, BD_SYS_AUX);
            __builtin_amdgcn_raw_buffer_store_b128(v1, dst, off + 16, 0, BD_SYS_AUX);
        }
    }
    tp_signal(a, FA, b, e);
    // a dead exchange pushes nothing further: the staged copies may be stale, and a result written now would be consumed by
    // peers whose own waits still succeed (the host check after the next sync raises on every rank)
    if (!tp_wait(a, FA, b, e, ERR)) { if (tid == 0) a.flags[EP + b] = e; return; }
    // ---- phase 2: reduce my slice in rank order, bias, one rounding to bf16, push the rows to every rank
    const __amdgpu_buffer_rsrc_t stage = sys_rsrc(a.peer_data[a.rank], a.data_bytes);
    for (int u = c0 + tid; u < c1; u += 256) {
        const int gu = a.rank * a.Us + u;
        if (gu >= a.U) break;
        u32x4 v[BD_TP_MAX][2];
#pragma unroll
        for (int p = 0; p < BD_TP_MAX; ++p) {                    // every staged copy in flight before the first add
            if (p >= a.size) break;
            if (p == a.rank) {
                const u32x4* src = reinterpret_cast<const u32x4*>(a.part + (size_t)gu * 8);
                v[p][0] = src[0]; v[p][1] = src[1];
            } else {
                const unsigned off = (unsigned)(((size_t)p * a.Us + u) * 32);
                v[p][0] = __builtin_amdgcn_raw_buffer_load_b128(stage, off, 0, BD_SYS_AUX);
                v[p][1] = __builtin_amdgcn_raw_buffer_load_b128(stage, off + 16, 0, BD_SYS_AUX);
            }
        }
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int p = 0; p < BD_TP_MAX; ++p) {                    // rank order: identical bits on every rank
            if (p >= a.size) break;
#pragma unroll
            for (int j = 0; j < 4; ++j) { acc[j] += __uint_as_float(v[p][0][j]); acc[4 + j] += __uint_as_float(v[p][1][j]); }
        }
        if (a.bias) {
            const int col = (gu % a.N8) * 8;
            const u32x4 bq = *reinterpret_cast<const u32x4*>(a.bias + col);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[2 * j] += bf2f((bf16_t)(bq[j] & 0xffff));
                acc[2 * j + 1] += bf2f((bf16_t)(bq[j] >> 16));
            }
        }
        const u32x4 o = {pack2(acc[0], acc[1]), pack2(acc[2], acc[3]), pack2(acc[4], acc[5]), pack2(acc[6], acc[7])};
        const unsigned ooff = (unsigned)(a.stage_bytes + (long long)gu * 16);
        for (int q = 0; q < a.size; ++q)
            __builtin_amdgcn_raw_buffer_store_b128(o, sys_rsrc(a.peer_data[q], a.data_bytes), ooff, 0, BD_SYS_AUX);
    }
    tp_signal(a, FB, b, e);
    (void)tp_wait(a, FB, b, e, ERR);
    if (tid == 0) a.flags[EP + b] = e;
    BD_KSTAMP_END(a.stamp);
}

// after an RCCL all-reduce the fp32 sums sit in the staging area: the consumer adds bias and rounds (Partial with S = 1)
int bdk_tp_allreduce(bd_comm* c, const float* part, const void* bias, int rows, int N, Partial* res, hipStream_t st) {
    if (!c || (c->size < 2 && c->mode != 1)) return -2;            // a single rank only through RCCL (plumbing tests)
    if (N % 8) return -3;
    {
        const long long U = (long long)rows * (N / 8), Us = (U + c->size - 1) / c->size;
        if (Us * c->size * 8 > c->max_elems) return -3;         // staging rows [size][Us] must fit in front of the result
    }
    const int Mpad = rows <= 32 ? 32 : (rows <= 64 ? 64 : ((rows + 127) / 128) * 128);
    c->n_exchanges++;
    if (c->mode == 1) {
        if (!c->nccl_allreduce || !c->nccl_comm) return -4;
        float* sum = reinterpret_cast<float*>(c->data);
        const int rc = c->nccl_allreduce(part, sum, (size_t)rows * N, /*ncclFloat32*/ 7, /*ncclSum*/ 0, c->nccl_comm, st);
        if (rc != 0) return -5;
        *res = Partial{sum, bias, 1, N, Mpad};
        return 0;
    }
    for (int p = 0; p < c->size; ++p)
        if (!c->peer_data[p] || !c->peer_flags[p]) return -6;
    ArArgs a;
    a.part = part; a.bias = (const bf16_t*)bias; a.flags = c->flags;
    for (int p = 0; p < BD_TP_MAX; ++p) { a.peer_data[p] = c->peer_data[p]; a.peer_flags[p] = c->peer_flags[p]; }
    a.stage_bytes = c->max_elems * 4;
    a.data_bytes = c->max_elems * 6;
    a.timeout_ticks = (long long)(c->timeout_s * 1e8);
    a.rank = c->rank; a.size = c->size; a.fences = c->fences; a.prepushed = c->prepushed_next; a.loopback = c->loopback;
    c->n_prepushed += c->prepushed_next;
    c->prepushed_next = 0;
    a.U = rows * (N / 8); a.Us = (a.U + c->size - 1) / c->size; a.N8 = N / 8;
    a.G = (a.Us + 255) / 256;
    if (a.G > BD_TP_GMAX) a.G = BD_TP_GMAX;
    if (a.G < 1) a.G = 1;
#ifdef BD_GEMM_STAMP
    a.stamp = bdk_stamp_next("tp_allreduce", a.G);
#endif
    BD_LAUNCH(tp_allreduce_kernel, dim3(a.G), dim3(256), 0, st, a);
    if (bd_launch_status() != 0) return -1;
    Partial r{reinterpret_cast<const float*>(c->data + a.stage_bytes), nullptr, 0, N, Mpad};
    r.sys = 1;                                               // remotely written: system-scope loads in the consumer
    *res = r;
    return 0;
}

// ---- all-gather of a COLUMN-split Linear's output (the adaLN projection under tensor parallelism, flow_head_parallel_x.py:331):
// rank r holds columns [r * Nl, (r + 1) * Nl) of the bf16 tensor [rows][N] and pushes them into every rank's copy (its own
// included) in the gather region; one flag round (block b of rank r <-> block b of every peer, its own epoch counters) tells a
// rank that every slice of ITS copy has landed.  Push-only, 16 B system-scope stores, bounded waits, the same error word.
struct AgArgs {
    const u32x4* src;           // this rank's slice [rows][Nl] bf16 row-major
    char* peer_data[BD_TP_MAX];
    int* peer_flags[BD_TP_MAX];
    int* flags;
    long long gather_off, gather_bytes, timeout_ticks;
    int rank, size, G, rows, Nl8, N8;          // Nl8 / N8: 16 B units per slice row / per full row
    int fences, loopback;
};
__global__ __launch_bounds__(256) void tp_allgather_kernel(AgArgs g) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int FC = 2 * BD_TP_MAX * BD_TP_GMAX + 1 + BD_TP_GMAX, ERR = 2 * BD_TP_MAX * BD_TP_GMAX, EPC = FC + BD_TP_MAX * BD_TP_GMAX;
    const int e = (int)((unsigned)g.flags[EPC + b] + 1u);
    const long long U = (long long)g.rows * g.Nl8, Ub = (U + g.G - 1) / g.G;
    const long long u0 = b * Ub, u1 = min(U, u0 + Ub);
    __amdgpu_buffer_rsrc_t dst[BD_TP_MAX];
#pragma unroll
    for (int q = 0; q < BD_TP_MAX; ++q) dst[q] = sys_rsrc(g.peer_data[q < g.size ? q : 0] + g.gather_off, g.gather_bytes);
    for (long long u = u0 + tid; u < u1; u += 256) {
        const int row = (int)(u / g.Nl8), cu = (int)(u % g.Nl8);
        const u32x4 v = g.src[u];
        const unsigned off = (unsigned)(((long long)row * g.N8 + (long long)g.rank * g.Nl8 + cu) * 16);
#pragma unroll
        for (int q = 0; q < BD_TP_MAX; ++q)
            if (q < g.size) __builtin_amdgcn_raw_buffer_store_b128(v, dst[q], off, 0, BD_SYS_AUX);
    }
    ArArgs a;                                                    // the signalling helpers take the exchange's argument block
    for (int p = 0; p < BD_TP_MAX; ++p) a.peer_flags[p] = g.peer_flags[p];
    a.flags = g.flags; a.rank = g.rank; a.size = g.size; a.timeout_ticks = g.timeout_ticks; a.fences = g.fences; a.loopback = g.loopback;
    tp_signal(a, FC, b, e);
    (void)tp_wait(a, FC, b, e, ERR);
    if (tid == 0) g.flags[EPC + b] = e;
}

// dst_local: where this rank's copy of the gathered tensor lives -- must be bd_comm_gather_ptr(c) (+ an offset every rank uses alike)
int bdk_tp_allgather(bd_comm* c, const void* slice, void* dst_local, int rows, int Nl, int N, hipStream_t st) {
    if (!c || c->size < 2 || c->mode == 1) return -2;             // (RCCL mode: the caller keeps the projection replicated)
    if (Nl % 8 || N != Nl * c->size) return -3;
    const long long off = (char*)dst_local - c->data, need = (long long)rows * N * 2;
    const long long g0 = c->max_elems * 6;
    if (off < g0 || off + need > g0 + c->gather_bytes || off % 16 || need >= (1LL << 31)) return -3;
    for (int p = 0; p < c->size; ++p)
        if (!c->peer_data[p] || !c->peer_flags[p]) return -6;
    AgArgs g;
    g.src = (const u32x4*)slice; g.flags = c->flags;
    for (int p = 0; p < BD_TP_MAX; ++p) { g.peer_data[p] = c->peer_data[p]; g.peer_flags[p] = c->peer_flags[p]; }
    g.gather_off = off; g.gather_bytes = need; g.timeout_ticks = (long long)(c->timeout_s * 1e8);
    g.rank = c->rank; g.size = c->size; g.rows = rows; g.Nl8 = Nl / 8; g.N8 = N / 8; g.fences = c->fences; g.loopback = c->loopback;
    const long long U = (long long)rows * g.Nl8;
    g.G = (int)((U + 2047) / 2048);                              // >= 8 units per thread, up to BD_TP_GMAX blocks
    if (g.G > BD_TP_GMAX) g.G = BD_TP_GMAX;
    if (g.G < 1) g.G = 1;
    c->n_gathers++;
    BD_LAUNCH(tp_allgather_kernel, dim3(g.G), dim3(256), 0, st, g);
    return bd_launch_status();
}
bool bdk_tp_push_target(bd_comm* c, int rows, int N, BdTpPush* out) {
    if (!c || c->size < 2 || c->mode != 0 || N % 8 || rows % c->size || (rows / c->size) % 8) return false;
    const long long U = (long long)rows * (N / 8), Us = U / c->size;          // whole rows per slice: U % size == 0
    if (Us * c->size * 8 > c->max_elems) return false;
    for (int p = 0; p < c->size; ++p)
        if (!c->peer_data[p]) return false;
    BdTpPush t;
    for (int p = 0; p < c->size; ++p) t.stage[p] = c->peer_data[p];
    t.Us = Us; t.rank = c->rank; t.size = c->size; t.rows_per_rank = rows / c->size;
    *out = t;
    return true;
}
void bdk_tp_mark_prepushed(bd_comm* c) { if (c) c->prepushed_next = 1; }
// the same push target with the sequence-parallel row ownership (8-row groups dealt round-robin to the ranks, BdTpPush::il)
bool bdk_tp_push_target_sp(bd_comm* c, int rows, int N, int seq, BdTpPush* out) {
    if (!bdk_tp_push_target(c, rows, N, out) || (rows / 8) % c->size) return false;
    out->il = 1;
    if (seq > 0) {
        int* const spf = c->flags + BD_TP_FLAG_INTS;
        out->done_cnt = spf + BD_SP_DONE;
        out->rc = spf + BD_SP_RC;
        out->seq = seq;
        for (int q = 0; q < c->size; ++q)
            out->sig[q] = c->loopback ? spf + BD_SP_P + q : c->peer_flags[q] + BD_TP_FLAG_INTS + BD_SP_P + c->rank;
    }
    return true;
}

void* bdk_comm_gather_ptr(const bd_comm* c) { return (c && c->gather_bytes > 0) ? c->data + c->max_elems * 6 : nullptr; }
long long bdk_comm_gather_bytes(const bd_comm* c) { return c ? c->gather_bytes : 0; }
int bdk_comm_mode(const bd_comm* c) { return c ? c->mode : 0; }

int bdk_comm_rank(const bd_comm* c) { return c ? c->rank : 0; }
int bdk_comm_size(const bd_comm* c) { return c ? c->size : 1; }


// ---- sequence-parallel exchange: host side (kernels: bd_sp.hip; the consumer's wait: bd_gemm_kernel.h)
bool bdk_sp_link(bd_comm* c, BdSpLink* out) {
    if (!c || c->size < 2 || c->mode != 0 || !c->hbuf || c->size > 8) return false;
    BdSpLink L;
    const long long aux_off = c->max_elems * 6 + c->gather_bytes;
    for (int p = 0; p < c->size; ++p) {
        if (!c->peer_data[p] || !c->peer_flags[p] || !c->peer_hbuf[p]) return false;
        L.stage[p] = c->peer_data[p];
        L.hbuf[p] = c->peer_hbuf[p];
        L.aux[p] = c->peer_data[p] + aux_off;
        L.spf[p] = c->peer_flags[p] + BD_TP_FLAG_INTS;
    }
    L.spf_local = c->flags + BD_TP_FLAG_INTS;
    L.err = c->flags + 2 * BD_TP_MAX * BD_TP_GMAX;
    L.stage_bytes = c->max_elems * 4; L.hbuf_bytes = c->hbuf_bytes; L.aux_bytes = c->aux_bytes;
    L.timeout_ticks = (long long)(c->timeout_s * 1e8);
    L.rank = c->rank; L.size = c->size; L.loopback = c->loopback;
    *out = L;
    return true;
}
long long bdk_sp_hbuf_bytes(const bd_comm* c) { return c ? c->hbuf_bytes : 0; }
void* bdk_sp_hbuf(const bd_comm* c) { return c ? c->hbuf : nullptr; }
int bdk_sp_next_seq(bd_comm* c) { return (c && c->sp_seq < BD_SP_SEQ_MAX) ? ++c->sp_seq : -1; }
void bdk_comm_count_exchange(bd_comm* c) { if (c) { c->n_exchanges++; c->n_prepushed++; } }
bool bdk_sp_hwait(bd_comm* c, int seq, int rows, BdHWait* out) {
    if (!c || !c->hbuf || rows > BD_SP_MAXROWS) return false;
    BdHWait w;
    w.flags = c->flags + BD_TP_FLAG_INTS + BD_SP_H;
    w.rc = c->flags + BD_TP_FLAG_INTS + BD_SP_RC;
    w.err = c->flags + 2 * BD_TP_MAX * BD_TP_GMAX;
    w.timeout_ticks = (long long)(c->timeout_s * 1e8);
    w.seq = seq; w.n = rows;
    *out = w;
    return true;
}
__global__ void sp_begin_kernel(int* spf) {
    if (threadIdx.x == 0) __hip_atomic_store(spf + BD_SP_RC, (int)((unsigned)__hip_atomic_load(spf + BD_SP_RC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + 1u),
                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
int bdk_sp_begin(bd_comm* c, hipStream_t st) {
    if (!c || !c->hbuf) return -2;
    c->sp_seq = 0;
    BD_LAUNCH(sp_begin_kernel, dim3(1), dim3(64), 0, st, c->flags + BD_TP_FLAG_INTS);
    return bd_launch_status();
}
// the consumer's wait as its own one-workgroup kernel: the following GEMM then starts behind a kernel boundary (whose acquire
// invalidates the caches) instead of polling in 240 workgroups -- the form for several ranks sharing ONE GPU, where a chip full of
// polling GEMM workgroups would starve the peer rank's row kernel they are waiting for (tests), and an A/B for the in-GEMM wait
__global__ void sp_wait_rows_kernel(BdHWait w) {
    const int e = bd_sp_epoch_of(__hip_atomic_load(w.rc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), w.seq);
    const long long t0 = wall_clock64();
    for (int i = threadIdx.x; i < w.n; i += blockDim.x) {
        while (bd_epoch_before(__hip_atomic_load(w.flags + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), e)) {
            if (__hip_atomic_load(w.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return;
            if (wall_clock64() - t0 > w.timeout_ticks) { __hip_atomic_fetch_or(w.err, 1 << 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
            __builtin_amdgcn_s_sleep(2);
        }
    }
}
int bdk_sp_wait_rows(bd_comm* c, int seq, int rows, hipStream_t st) {
    BdHWait w;
    if (!bdk_sp_hwait(c, seq, rows, &w)) return -2;
    BD_LAUNCH(sp_wait_rows_kernel, dim3(1), dim3(256), 0, st, w);
    return bd_launch_status();
}

extern "C" {

bd_comm* bd_comm_create(int rank, int size, long long max_elems) { return bd_comm_create3(rank, size, max_elems, 0, 0); }
bd_comm* bd_comm_create2(int rank, int size, long long max_elems, long long gather_bytes) { return bd_comm_create3(rank, size, max_elems, gather_bytes, 0); }

/* ... with an all-gather region of `gather_bytes` behind the exchange buffers (same allocation, same IPC handle): where the
 * column-split adaLN projection's output is assembled on every rank (bd_comm_gather_ptr) */
/* ... and, with hbuf_bytes > 0, the sequence-parallel exchange's buffers (bd_sp.hip): a cacheable exported landing buffer of
 * hbuf_bytes (rows x D x 2: the bf16 operand rows every rank pushes to every rank) and a 64 KiB landing area for the final latent rows */
bd_comm* bd_comm_create3(int rank, int size, long long max_elems, long long gather_bytes, long long hbuf_bytes) {
    if (size < 1 || size > BD_TP_MAX || rank < 0 || rank >= size || max_elems < 8 || gather_bytes < 0 || hbuf_bytes < 0 || hbuf_bytes >= (1LL << 31)) { bdk_set_error("bd_comm_create: bad rank/size/capacity"); return nullptr; }
    bd_comm* c = new bd_comm();
    c->rank = rank; c->size = size;
    c->max_elems = (max_elems + 127) / 128 * 128 + 128 * BD_TP_MAX;   // slices are ceil(units / size): slack per rank; keeps the gather region 256 B aligned
    c->gather_bytes = (gather_bytes + 255) / 256 * 256;
    // the kernel addresses an allocation through ONE buffer resource: 32-bit size and offsets
    if (c->max_elems * 6 >= (1LL << 31)) { delete c; bdk_set_error("bd_comm_create: capacity too large (rows * N * 6 must stay below 2^31 bytes)"); return nullptr; }
    c->hbuf_bytes = (hbuf_bytes + 255) / 256 * 256;
    c->aux_bytes = c->hbuf_bytes > 0 ? BD_SP_AUX_BYTES : 0;
    const size_t dbytes = (size_t)c->max_elems * 6 + (size_t)c->gather_bytes + (size_t)c->aux_bytes;
    const size_t fbytes = (size_t)(BD_TP_FLAG_INTS + BD_SP_FLAG_INTS) * sizeof(int);
    // staging / result buffer: written by the PEERS over xGMI and read here inside the same kernel.  Ordinary (coarse-grained)
    // device memory is cached in this GPU's L2 as device-coherent only -- a line kept from the previous exchange could be
    // served instead of what a peer has pushed since (the one-GPU tests cannot show this: all "ranks" share one L2).  Uncached
    // (fine-grained) memory is the coherence contract RCCL's own exchange buffers rely on; the same allocation kind as the flag
    // block below, whose IPC export is covered by the two-process test.  Plain hipMalloc only if the runtime refuses the flag.
    bool uncached = hipExtMallocWithFlags((void**)&c->data, dbytes, hipDeviceMallocUncached) == hipSuccess;
    if (uncached) {                                              // it has to be exportable: probe now, while falling back is still possible
        hipIpcMemHandle_t probe;
        if (hipIpcGetMemHandle(&probe, c->data) != hipSuccess) { (void)hipGetLastError(); hipFree(c->data); c->data = nullptr; uncached = false; }
    } else {
        (void)hipGetLastError();
    }
    if (!uncached && hipMalloc((void**)&c->data, dbytes) != hipSuccess) { delete c; bdk_set_error("bd_comm_create: hipMalloc failed"); return nullptr; }
    c->data_uncached = uncached;                                 // reported by bd_comm_info: the host routes cross-device exchanges
                                                                 // through RCCL when this is false (tp.py)
    // flags: uncached (fine-grained) so a peer's write is seen by the polling loads; plain device memory + system-scope
    // atomics if this runtime refuses the flag
    c->flags_uncached = true;
    if (hipExtMallocWithFlags((void**)&c->flags, fbytes, hipDeviceMallocUncached) != hipSuccess) {
        c->flags_uncached = false;
        (void)hipGetLastError();
        if (hipMalloc((void**)&c->flags, fbytes) != hipSuccess) { hipFree(c->data); delete c; bdk_set_error("bd_comm_create: flag allocation failed"); return nullptr; }
    }
    if (c->hbuf_bytes > 0) {
        // ordinary (cacheable) device memory, exportable like any hipMalloc: the consuming GEMM invalidates its caches after the flag
        // wait and then re-reads the operand from L2 like any other activation (bd_gemm_kernel.h)
        if (hipMalloc((void**)&c->hbuf, (size_t)c->hbuf_bytes) != hipSuccess) {
            hipFree(c->data); hipFree(c->flags); delete c; bdk_set_error("bd_comm_create: operand landing buffer allocation failed"); return nullptr;
        }
        hipMemset(c->hbuf, 0, (size_t)c->hbuf_bytes);
    }
    hipMemset(c->data, 0, dbytes);
    hipMemset(c->flags, 0, fbytes);
    hipDeviceSynchronize();
    c->own = true;
    c->peer_data[rank] = c->data;
    c->peer_flags[rank] = c->flags;
    c->peer_hbuf[rank] = c->hbuf;
    return c;
}

void bd_comm_destroy(bd_comm* c) {
    if (!c) return;
    for (int p = 0; p < BD_TP_MAX; ++p) {
        if (c->ipc_open[p]) { hipIpcCloseMemHandle(c->peer_data[p]); hipIpcCloseMemHandle(c->peer_flags[p]); }
        if (c->ipc_open_h[p]) hipIpcCloseMemHandle(c->peer_hbuf[p]);
        if (c->scratch_data[p]) hipFree(c->scratch_data[p]);
        if (c->scratch_hbuf[p]) hipFree(c->scratch_hbuf[p]);
    }
    if (c->own) { hipFree(c->data); hipFree(c->flags); if (c->hbuf) hipFree(c->hbuf); }
    delete c;
}

/* 2 x hipIpcMemHandle_t (64 B each): data, flags */
int bd_comm_ipc_handles(bd_comm* c, void* out128) {
    hipIpcMemHandle_t h[2];
    if (hipIpcGetMemHandle(&h[0], c->data) != hipSuccess) return cfail(std::string("hipIpcGetMemHandle(data): ") + hipGetErrorString(hipGetLastError()));
    if (hipIpcGetMemHandle(&h[1], c->flags) != hipSuccess) return cfail(std::string("hipIpcGetMemHandle(flags): ") + hipGetErrorString(hipGetLastError()));
    std::memcpy(out128, h, sizeof(h));
    return 0;
}

int bd_comm_open_peer(bd_comm* c, int peer, const void* handles128) {
    if (peer < 0 || peer >= c->size || peer == c->rank) return cfail("bd_comm_open_peer: bad peer");
    hipIpcMemHandle_t h[2];
    std::memcpy(h, handles128, sizeof(h));
    void *d = nullptr, *f = nullptr;
    if (hipIpcOpenMemHandle(&d, h[0], hipIpcMemLazyEnablePeerAccess) != hipSuccess)
        return cfail(std::string("hipIpcOpenMemHandle(data): ") + hipGetErrorString(hipGetLastError()));
    if (hipIpcOpenMemHandle(&f, h[1], hipIpcMemLazyEnablePeerAccess) != hipSuccess)
        return cfail(std::string("hipIpcOpenMemHandle(flags): ") + hipGetErrorString(hipGetLastError()));
    c->peer_data[peer] = (char*)d; c->peer_flags[peer] = (int*)f; c->ipc_open[peer] = true;
    return 0;
}

/* peers that live in this process (several contexts on one device: the protocol test) */
int bd_comm_set_peer_ptrs(bd_comm* c, int peer, void* data, void* flags) {
    if (peer < 0 || peer >= c->size || peer == c->rank) return cfail("bd_comm_set_peer_ptrs: bad peer");
    c->peer_data[peer] = (char*)data; c->peer_flags[peer] = (int*)flags;
    return 0;
}
void* bd_comm_local_data(bd_comm* c) { return c->data; }
void* bd_comm_gather_ptr(bd_comm* c) { return bdk_comm_gather_ptr(c); }   /* this rank's copy of the all-gather region (null: none) */
long long bd_comm_gather_bytes(bd_comm* c) { return bdk_comm_gather_bytes(c); }
void* bd_comm_local_flags(bd_comm* c) { return c->flags; }

int bd_comm_set_rccl(bd_comm* c, void* nccl_comm, void* nccl_allreduce_fn) {
    c->nccl_comm = nccl_comm;
    c->nccl_allreduce = reinterpret_cast<int (*)(const void*, void*, size_t, int, int, void*, hipStream_t)>(nccl_allreduce_fn);
    c->mode = (nccl_comm && nccl_allreduce_fn) ? 1 : 0;
    return 0;
}
int bd_comm_set_timeout(bd_comm* c, double seconds) { c->timeout_s = seconds; return 0; }
int bd_comm_set_fences(bd_comm* c, int on) { c->fences = on != 0; return 0; }
/* the NEXT exchange finds its staging rows already written (the producing GEMM's epilogue pushed them: BdTpPush); standalone use:
 * protocol micro-benchmarks of phase 2 alone */
int bd_comm_mark_prepushed(bd_comm* c) { bdk_tp_mark_prepushed(c); return 0; }
/* after a failed exchange (all ranks, between two host barriers): clear flags, epochs and the error word */
int bd_comm_reset(bd_comm* c) {
    const size_t fbytes = (size_t)(BD_TP_FLAG_INTS + BD_SP_FLAG_INTS) * sizeof(int);
    if (hipDeviceSynchronize() != hipSuccess || hipMemset(c->flags, 0, fbytes) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
        return cfail("bd_comm_reset failed");
    return 0;
}
long long bd_comm_exchanges(bd_comm* c) { return c->n_exchanges; }
/* exchanges whose reduce-scatter phase ran in the producing GEMM's epilogue (fused push): tests assert that the fusion engaged */
long long bd_comm_prepushed(bd_comm* c) { return c->n_prepushed; }
/* out[0] = exchange buffer is uncached (fine-grained) device memory, out[1] = flag block is, out[2] = mode (0 hand-written
 * exchange, 1 ncclAllReduce), out[3] = capacity in elements */
int bd_comm_info(bd_comm* c, long long* out4) {
    if (!c || !out4) return cfail("bd_comm_info: null");
    out4[0] = c->data_uncached; out4[1] = c->flags_uncached; out4[2] = c->mode; out4[3] = c->max_elems;
    return 0;
}

/* host-side check after a stream sync: bit p set = this rank's wait for peer p ran out of its time budget; bit 8 + r set =
 * rank r reported that one of ITS waits did (every rank then raises together) */
int bd_comm_error(bd_comm* c) {
    int e = 0;
    if (hipMemcpy(&e, c->flags + 2 * BD_TP_MAX * BD_TP_GMAX, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return e;
}

/* standalone exchange (tests, micro-benchmarks): out = bf16(sum over ranks of part + bias), [rows][N];
 * *out_ptr receives the address of the result (this rank's copy), *out_is_fp32 says whether the consumer still has to
 * add the bias and round (RCCL mode) */
int bd_comm_allreduce(bd_comm* c, const float* part, const void* bias_bf16, int rows, int N, void** out_ptr, int* out_is_fp32,
                      void* stream) {
    Partial r;
    const int rc = bdk_tp_allreduce(c, part, bias_bf16, rows, N, &r, (hipStream_t)stream);
    if (rc != 0) return cfail("bd_comm_allreduce failed with " + std::to_string(rc));
    *out_ptr = const_cast<float*>(r.p);
    *out_is_fp32 = r.S != 0;
    return 0;
}

/* standalone all-gather of column slices (tests): slice [rows][Nl] bf16 -> this rank's gather region [rows][Nl * size] */
int bd_comm_allgather(bd_comm* c, const void* slice_bf16, int rows, int Nl, void* stream) {
    const int rc = bdk_tp_allgather(c, slice_bf16, bdk_comm_gather_ptr(c), rows, Nl, Nl * (c ? c->size : 1), (hipStream_t)stream);
    if (rc != 0) return cfail("bd_comm_allgather failed with " + std::to_string(rc));
    return 0;
}

/* copy `bytes` of this rank's exchange buffer (offset from its base: 0 = fp32 staging, max_elems*4 = bf16 result) into a
 * caller-owned device buffer -- how tests read a standalone exchange back */
int bd_comm_copy_out(bd_comm* c, void* dst, long long bytes, int from_result, void* stream) {
    const long long off = from_result == 2 ? c->max_elems * 6 : (from_result ? c->max_elems * 4 : 0);   // 2: the all-gather region
    if (bytes < 0 || off + bytes > c->max_elems * 6 + c->gather_bytes + c->aux_bytes) return cfail("bd_comm_copy_out: range");
    return hipMemcpyAsync(dst, c->data + off, (size_t)bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream) == hipSuccess
               ? 0 : cfail("bd_comm_copy_out: hipMemcpyAsync failed");
}


/* 3 x hipIpcMemHandle_t (64 B each): data, flags, operand landing buffer (zeros when there is none) */
int bd_comm_ipc_handles3(bd_comm* c, void* out192) {
    std::memset(out192, 0, 192);
    if (bd_comm_ipc_handles(c, out192) != 0) return -1;
    if (c->hbuf) {
        hipIpcMemHandle_t h;
        if (hipIpcGetMemHandle(&h, c->hbuf) != hipSuccess) return cfail(std::string("hipIpcGetMemHandle(hbuf): ") + hipGetErrorString(hipGetLastError()));
        std::memcpy((char*)out192 + 128, &h, sizeof(h));
    }
    return 0;
}
int bd_comm_open_peer3(bd_comm* c, int peer, const void* handles192) {
    if (bd_comm_open_peer(c, peer, handles192) != 0) return -1;
    if (c->hbuf) {
        hipIpcMemHandle_t h;
        std::memcpy(&h, (const char*)handles192 + 128, sizeof(h));
        void* d = nullptr;
        if (hipIpcOpenMemHandle(&d, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess)
            return cfail(std::string("hipIpcOpenMemHandle(hbuf): ") + hipGetErrorString(hipGetLastError()));
        c->peer_hbuf[peer] = (char*)d; c->ipc_open_h[peer] = true;
    }
    return 0;
}
int bd_comm_set_peer_ptrs3(bd_comm* c, int peer, void* data, void* flags, void* hbuf) {
    if (bd_comm_set_peer_ptrs(c, peer, data, flags) != 0) return -1;
    c->peer_hbuf[peer] = (char*)hbuf;
    return 0;
}
void* bd_comm_local_hbuf(bd_comm* c) { return c->hbuf; }
long long bd_comm_hbuf_bytes(bd_comm* c) { return c->hbuf_bytes; }

/* one round of the sequence-parallel hand-off's self-test (bd_sp.hip): this rank pushes its rows of pattern `round` into every rank's operand
 * landing buffer, then checks every row of its own buffer behind the GEMM prologue's wait; *bad_dev (device int) += mismatching 16 B units */
int bd_comm_sp_selftest(bd_comm* c, int round, int rows, int D, int wait_in_check, int* bad_dev, void* stream) {
    const int rc = bdk_sp_selftest(c, round, rows, D, wait_in_check, bad_dev, (hipStream_t)stream);
    return rc == 0 ? 0 : cfail("bd_comm_sp_selftest failed with " + std::to_string(rc));
}

/* ONE rank of a `size`-rank group alone on this GPU: every peer's buffers become scratch allocations of this process, every flag a
 * peer would write is written locally by the block that plays the same role.  The rank's kernels then launch, stream, push and wait
 * exactly as on a node -- minus the links -- so that its critical path can be timed on one GPU (tools/head_sweep.py --tp-shard).
 * Results are meaningless (the peers' contributions are zeros). */
int bd_comm_set_loopback(bd_comm* c) {
    if (!c || c->size < 2 || c->mode != 0) return cfail("bd_comm_set_loopback: a multi-rank communicator on the hand-written exchange");
    const size_t dbytes = (size_t)c->max_elems * 6 + (size_t)c->gather_bytes + (size_t)c->aux_bytes;
    for (int p = 0; p < c->size; ++p) {
        if (p == c->rank) continue;
        if (!c->scratch_data[p] && hipMalloc((void**)&c->scratch_data[p], dbytes) != hipSuccess) return cfail("bd_comm_set_loopback: hipMalloc failed");
        hipMemset(c->scratch_data[p], 0, dbytes);
        if (c->hbuf_bytes > 0) {
            if (!c->scratch_hbuf[p] && hipMalloc((void**)&c->scratch_hbuf[p], (size_t)c->hbuf_bytes) != hipSuccess) return cfail("bd_comm_set_loopback: hipMalloc failed");
            hipMemset(c->scratch_hbuf[p], 0, (size_t)c->hbuf_bytes);
        }
        c->peer_data[p] = c->scratch_data[p];
        c->peer_hbuf[p] = c->scratch_hbuf[p];
        c->peer_flags[p] = c->flags;
    }
    hipDeviceSynchronize();
    c->loopback = 1;
    return 0;
}
}  // extern "C"
