// Weight-streaming skinny-M GEMM for gfx950:  out[M<=128*j, N] = A[M, K] (bf16) x W[N, K]^T (bf16), fp32 accumulate.
//
// This is the kernel that bounds the whole generation loop: at M = 128 rows (one 64-token patch, cond+uncond)
// every Linear of the diffusion head (flow_head_parallel_x.py:325-342) and of the Qwen3 decode step
// (HF modeling_qwen3.py:81-83,252-279) is HBM weight-streaming bound.
//
// Design (MI355X-first, not a tiled-GEMM port):
//  * W is re-packed ONCE at load time into MFMA-operand order: for every 32-row panel nb and k-step ks a
//    1 KiB chunk whose 64 lanes' 16 B are exactly the v_mfma_f32_32x32x16_bf16 B operand.  A wave streams its
//    panel with perfectly coalesced 1 KiB non-temporal loads straight into VGPRs -- W never touches LDS,
//    nothing is shared between waves, no bank conflicts, no transposes.
//  * A (the 128 activation rows) is produced by the upstream kernels directly in the same fragment-major
//    layout (bd_common.h afrag_off), so a 64-deep K stage is one contiguous 16 KiB copy into LDS and every
//    ds_read_b128 is lane-linear (conflict free).
//  * each wave owns 32 output columns x all rows of the tile (MB x 32x32 fp32 accumulators); NW waves per
//    block share the A stage.  Split-K over the grid fills 256 CUs for the N = 5120 shapes; partial sums go
//    to fp32 slabs that the consumer row-kernels reduce in their prologue (bd_rows.hip) -- no extra launch.
//  * fused SwiGLU epilogue (gate/up rows interleaved 16/16 inside each packed panel so the partner value is
//    one cross-lane exchange away) writes the bf16 activation in fragment-major order for the next GEMM.
#include <string>
#include "bd_gemm_kernel.h"

// Packed-weight order in HBM.  0 = panel-major  [panel][K/64 stages][4 k-steps][64 lanes]: every wave walks its own
// contiguous 32-column panel.  1 = stage-major [K/64 stages][panel][4 k-steps][64 lanes]: the waves of the whole grid,
// which advance through K in lockstep, read ONE contiguous moving window of (N/32) x 4 KiB per stage, so the stream
// is spread over every HBM channel the way a flat copy is, instead of N/32 streams 2K bytes apart.
static int g_w_layout = 0;   // measured identical on MI355X (profiles/r01_gemm_fixed_cost.log): panel-major keeps N-slices contiguous
void bdk_set_w_layout(int v) { g_w_layout = v; }
int bdk_get_w_layout() { return g_w_layout; }
void bdk_w_strides(int panels_total, int K, size_t* PS, size_t* SS) {
    if (g_w_layout == 0) { *PS = (size_t)(K >> 4) * 64; *SS = 256; }
    else { *PS = 256; *SS = (size_t)panels_total * 256; }
}

// ---------------------------------------------------------------------------------------------------
// weight packing: src [rows][K] bf16 row-major  ->  dst panels [nb0 + rows/32][K/16][64 lanes][8 bf16]
// mode 0: packed row r <- src row r.   mode 1 (SwiGLU pair): panel p, row i<16 <- gate[p*16+i], i>=16 <- up[p*16+i-16]
// ---------------------------------------------------------------------------------------------------
// dst unit (16 B) of (panel pn, k-step ks, lane l) = pn * PS + (ks >> 2) * SS + (ks & 3) * 64 + l  (see bdk_w_strides)
__global__ void pack_w_kernel(u32x4* __restrict__ dst, const bf16_t* __restrict__ src, const bf16_t* __restrict__ src2,
                              int panels, int K, int nb0, int mode, size_t PS, size_t SS) {
    const int KS = K >> 4;
    const size_t total = (size_t)panels * KS * 64;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(u & 63);
        const size_t c = u >> 6;
        const int ks = (int)(c % KS);
        const int pn = (int)(c / KS);
        const int i = l & 31;
        const bf16_t* row;
        if (mode == 0) row = src + ((size_t)pn * 32 + i) * K;
        else row = (i < 16) ? src + ((size_t)pn * 16 + i) * K : src2 + ((size_t)pn * 16 + (i - 16)) * K;
        const u32x4 v = *reinterpret_cast<const u32x4*>(row + ks * 16 + (l >> 5) * 8);
        dst[(size_t)(nb0 + pn) * PS + (size_t)(ks >> 2) * SS + (ks & 3) * 64 + l] = v;
    }
}

// rows fp32/bf16 row-major [M][K] -> bf16 fragment-major (pad rows untouched)
__global__ void rows_to_afrag_kernel(bf16_t* __restrict__ dst, const float* __restrict__ src32,
                                     const bf16_t* __restrict__ src16, int M, int K, int RB) {
    const size_t total = (size_t)M * (K >> 3);
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (size_t)gridDim.x * blockDim.x) {
        const int m = (int)(u / (K >> 3));
        const int k0 = (int)(u % (K >> 3)) * 8;
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (src32) w[j] = pack2(src32[(size_t)m * K + k0 + 2 * j], src32[(size_t)m * K + k0 + 2 * j + 1]);
            else w[j] = (unsigned)src16[(size_t)m * K + k0 + 2 * j] | ((unsigned)src16[(size_t)m * K + k0 + 2 * j + 1] << 16);
        }
        *reinterpret_cast<u32x4*>(dst + afrag_off(m, k0, RB)) = (u32x4){w[0], w[1], w[2], w[3]};
    }
}

// ---------------------------------------------------------------------------------------------------
// the GEMM
// ---------------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------------
// 256-row passes (two images with CFG, or 4 as two row tiles): the matrix pipe, not HBM, is the scarce unit here
// (256 FLOP per weight byte), so this variant is organised around keeping MFMAs issuing back to back:
//   * 4 waves, one per SIMD, each 256 rows x 64 columns = 16 accumulators (256 AGPRs); an A fragment read from
//     LDS feeds two MFMAs;
//   * the A stage is TRIPLE buffered in LDS and loaded three stages ahead, so that the first fragments of the next
//     stage are already in registers when the per-stage barrier falls (with two buffers every stage began with an
//     exposed ds_read latency on an otherwise idle SIMD);
//   * fragment reads, the ds_write of the stage after next and the global loads are issued one per MFMA pair inside
//     the MFMA stream (sched_barrier pins the order), not in a block between stages;
//   * loads past the last stage are clamped to it instead of branched around: every phase issues the same number of
//     loads, which keeps hipcc's s_waitcnt counts exact (the redundant lines are L2 hits).
//   * WR = W stages a wave keeps in flight in registers.  A W stage is consumed one 2048-cycle phase after the previous one,
//     so WR = 2 hides only ~0.85 us of HBM latency behind the MFMAs; WR = 3 doubles that for 32 more VGPRs.
//   * XCD = 1: the row tiles that stream the same weight slice are placed on the SAME XCD (blocks b and b + 8): the second
//     reader hits that XCD's L2 instead of making the fabric deliver the slice to two L2s.
//   * RED (round 5): two K slices reduced INSIDE the launch -- each slice parks its accumulators (accumulator order, 16 B `sc1`
//     write-through stores), takes a ticket on the tile's counter, and the slice that arrives second adds the other's slab to its own
//     registers (own + other == other + own: the result does not depend on the arrival order) and runs the bf16 / SwiGLU epilogue.
//     At num_images = 4 the consumers of the 2-slice GEMMs were reading 63 MB of fp32 slabs per launch (finalize_rows + head_attn
//     behind qkv, swiglu_rows behind w1); with RED they read 16 / 8 MB of bf16 written once.
template <int EPI, int WR, bool XCD, bool RED = false>
__global__ __launch_bounds__(256) void gemm_wide_kernel(GemmP p) {
    constexpr int MB = 8, NPW = 2, NT = 256, UNITS = MB * 256, XL = UNITS / NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* const lds = reinterpret_cast<u32x4*>(smem);    // three A-stage buffers of UNITS each (96 KiB)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int S = p.S;
    // row tiles fastest: the workgroups that stream the SAME weight slice (one per 256-row tile) are dispatched back to back,
    // so the slice comes from HBM once and from L2 / Infinity Cache for the others (512 rows = num_images 4: W used to be
    // streamed twice; 12 288 rows = the ImageNet batch: 48 consumers per slice)
    const int RT = p.RB / 8;
    int mt, rest;
    if constexpr (XCD) {
        // blocks are dealt round-robin to the 8 XCDs: within a group of 8 * RT blocks, block j runs on XCD j % 8; give the
        // RT blocks of one XCD the RT row tiles of one weight slice
        const int nrest = gridDim.x / RT, full = (nrest / 8) * 8;       // weight slices in whole groups of 8
        if ((int)blockIdx.x < full * RT) {
            const int grp = blockIdx.x / (8 * RT), j = blockIdx.x % (8 * RT);
            mt = j / 8;
            rest = grp * 8 + (j % 8);
        } else {                                        // the last, partial group: plain order over what is left
            const int left = nrest - full, jj = blockIdx.x - full * RT;
            mt = jj / left;
            rest = full + jj % left;
        }
    } else {
        mt = blockIdx.x % RT; rest = blockIdx.x / RT;
    }
    // K slice of this workgroup.  Under the XCD placement `rest` % 8 is the XCD: with S = 2 / 4 / 8 the plain rest % S would pin every
    // XCD to ONE K slice (its L2 then holds one half of the activations and the weight stream of a tile's slices never meets in one
    // L2); rotating the slice by the group index (rest / 8: the same for all slices of a tile when S divides 8, so the map stays a
    // bijection) lets every XCD walk through all slices
    const int nt = rest / S;
    const int s = (XCD && (8 % S) == 0) ? (rest % S + rest / 8) % S : rest % S;
    const int nb = (nt * 4 + wave) * NPW;
    const int nst_total = p.K >> 6;
    const int q = (nst_total + S - 1) / S;
    const int st0 = s * q;
    const int nst = min(q, nst_total - st0);
    const int last = nst - 1;

    const u32x4* Wp = p.W + (size_t)nb * p.PS + (size_t)st0 * p.SS + lane;
    const size_t w_stage = p.SS;
    size_t a_off[XL];
#pragma unroll
    for (int j = 0; j < XL; ++j) {
        const int u = tid + j * NT;
        const int c = u >> 6;
        a_off[j] = (((size_t)(st0 * 4 + c / MB) * p.RB) + mt * MB + (c % MB)) * 64 + (u & 63);
    }
    const size_t a_stage = (size_t)4 * p.RB * 64;

    u32x4 w[WR][NPW * 4], xr[XL], xf[2][MB];
    f32x16 acc[MB * NPW];
#pragma unroll
    for (int m = 0; m < MB * NPW; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    // one row tile per weight slice: non-temporal (the slice is read once, keep it out of L2's way).  Several row tiles: the SAME
    // slice is read by RT workgroups a few hundred ns apart -- a non-temporal line is gone again by then (PMC, 512 rows, adaLN:
    // 2.16 x the weight bytes fetched, profiles/r03_pmc_gemm_traffic.json), a default-policy line is still in that XCD's L2
    const bool keep = p.w_keep != 0;
    auto load_w = [&](u32x4(&wr)[NPW * 4], int i) {
        i = min(i, last);
        if (keep) {
#pragma unroll
            for (int pn = 0; pn < NPW; ++pn)
#pragma unroll
                for (int j = 0; j < 4; ++j) wr[pn * 4 + j] = Wp[(size_t)pn * p.PS + (size_t)i * w_stage + j * 64];
        } else {
#pragma unroll
            for (int pn = 0; pn < NPW; ++pn)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    wr[pn * 4 + j] = __builtin_nontemporal_load(Wp + (size_t)pn * p.PS + (size_t)i * w_stage + j * 64);
        }
    };
    auto load_x = [&](int i) {
        i = min(i, last);
#pragma unroll
        for (int j = 0; j < XL; ++j) xr[j] = p.A[a_off[j] + (size_t)i * a_stage];
    };
    auto store_x = [&](u32x4* buf) {
#pragma unroll
        for (int j = 0; j < XL; ++j) buf[tid + j * NT] = xr[j];
    };

    // prologue: A stages 0 and 1 into LDS, stage 2 in registers; W stages 0 .. WR-1 in registers
    u32x4 *cur = lds, *nxt = lds + UNITS, *wr3 = lds + 2 * UNITS;
    load_x(0);
    load_w(w[0], 0);
    store_x(cur);
    load_x(1);
    store_x(nxt);
    if constexpr (WR == 3) load_w(w[1], 1);
    load_x(2);                                // issue order A(j+2), W(j+WR-1) as in the steady state: the waitcnt states
    load_w(w[WR - 1], WR - 1);                // merged at the loop header then agree and stay exact
    __syncthreads();
#pragma unroll
    for (int m = 0; m < MB; ++m) xf[0][m] = cur[m * 64 + lane];

    // one phase = one 64-deep K stage j: 4 k-steps x 16 MFMAs.  PAR = j % WR selects the W register slot.
#ifdef BD_GEMM_STAMP
    // measurement build: shader-clock cycles wave 0 spends (1) waiting for this phase's W stage, (2) waiting for the A stage it writes to
    // LDS, (3) at the per-phase barrier -- explicit waits in front of the compiler's own, bracketed by s_memtime (bd_common.h stamps:
    // word 1 = W wait, 2 = A wait, 4 = barrier, 5 = whole loop, all in shader cycles; 0 / 3 / 6 = realtime start / loop end / drained)
    unsigned long long st_w = 0, st_a = 0, st_b = 0;
    const unsigned long long st_loop0 = __builtin_readcyclecounter();
    BD_KSTAMP(p.stamp, 0);
#endif
    auto phase = [&](auto PAR, int j) {
        constexpr int P = decltype(PAR)::value;
#ifdef BD_GEMM_STAMP
        {
            const unsigned long long t0 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 + 8 * (WR - 1)) : "memory");
            st_w += __builtin_readcyclecounter() - t0;
        }
#endif
        // k-step 0 (xf[0]) | read k-step 1 -> xf[1]
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            xf[1][m] = cur[(1 * MB + m) * 64 + lane];
            acc[m * 2 + 0] = mfma32(xf[0][m], w[P][0], acc[m * 2 + 0]);
            acc[m * 2 + 1] = mfma32(xf[0][m], w[P][4], acc[m * 2 + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // k-step 1 (xf[1]) | read k-step 2 -> xf[0] | write A stage j+2 (loaded during phase j-1)
#ifdef BD_GEMM_STAMP
        {
            const unsigned long long t0 = __builtin_readcyclecounter();
            asm volatile("s_waitcnt vmcnt(%0)" :: "n"(8 * (WR - 1)) : "memory");
            st_a += __builtin_readcyclecounter() - t0;
        }
#endif
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            xf[0][m] = cur[(2 * MB + m) * 64 + lane];
            wr3[tid + m * NT] = xr[m];
            acc[m * 2 + 0] = mfma32(xf[1][m], w[P][1], acc[m * 2 + 0]);
            acc[m * 2 + 1] = mfma32(xf[1][m], w[P][5], acc[m * 2 + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // k-step 2 (xf[0]) | read k-step 3 -> xf[1] | load A stage j+3
        const int ix = min(j + 3, last);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            xf[1][m] = cur[(3 * MB + m) * 64 + lane];
            xr[m] = p.A[a_off[m] + (size_t)ix * a_stage];
            acc[m * 2 + 0] = mfma32(xf[0][m], w[P][2], acc[m * 2 + 0]);
            acc[m * 2 + 1] = mfma32(xf[0][m], w[P][6], acc[m * 2 + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // k-step 3 (xf[1]) | read k-step 0 of stage j+1 -> xf[0] (visible since the previous barrier)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            xf[0][m] = nxt[m * 64 + lane];
            acc[m * 2 + 0] = mfma32(xf[1][m], w[P][3], acc[m * 2 + 0]);
            acc[m * 2 + 1] = mfma32(xf[1][m], w[P][7], acc[m * 2 + 1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        load_w(w[P], j + WR);                 // this slot's MFMAs have all issued
#ifdef BD_GEMM_STAMP
        const unsigned long long tb0 = __builtin_readcyclecounter();
        __syncthreads();
        st_b += __builtin_readcyclecounter() - tb0;
#else
        __syncthreads();
#endif
        u32x4* t = cur; cur = nxt; nxt = wr3; wr3 = t;
        if (j == last) BD_MFMA_DRAIN();       // the last phase is followed, across a branch, by the accumulator reads (bd_common.h)
    };

    int j = 0;
    if constexpr (WR == 2) {
        for (; j + 1 < nst; j += 2) {
            phase(std::integral_constant<int, 0>{}, j);
            phase(std::integral_constant<int, 1>{}, j + 1);
        }
        if (j < nst) phase(std::integral_constant<int, 0>{}, j);
    } else {
        for (; j + 2 < nst; j += 3) {
            phase(std::integral_constant<int, 0>{}, j);
            phase(std::integral_constant<int, 1>{}, j + 1);
            phase(std::integral_constant<int, 2>{}, j + 2);
        }
        if (j < nst) phase(std::integral_constant<int, 0>{}, j);
        if (j + 1 < nst) phase(std::integral_constant<int, 1>{}, j + 1);
    }

#ifdef BD_GEMM_STAMP
    BD_KSTAMP(p.stamp, 3);
    bd_kstamp_val(p.stamp, 1, st_w); bd_kstamp_val(p.stamp, 2, st_a); bd_kstamp_val(p.stamp, 4, st_b);
    bd_kstamp_val(p.stamp, 5, __builtin_readcyclecounter() - st_loop0);
    bd_kstamp_val(p.stamp, 7, (unsigned long long)nst);
#endif
    if constexpr (RED) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
        static_assert(!RED, "the relaxed sc1 slab hand-off is validated for gfx950 only");
#endif
        // S == 2 (the launcher guarantees it).  Region (tile, wave, slice) = [accumulator][r4][lane] x 16 B.
        const __amdgpu_buffer_rsrc_t sl = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)2 * p.Mpad * p.N * 4), 0x00020000);
        const int tile = mt * (p.N >> 8) + nt;
        const size_t region0 = ((size_t)tile * 4 + wave) * 2;
        auto slab_off = [&](int s_, int a, int r4) -> unsigned {
            return (unsigned)((((region0 + s_) * (MB * NPW) + a) * 4 + r4) * 1024 + lane * 16);
        };
        constexpr int SC1 = 16;
        // TICKET FIRST (round 6): only the slice that arrives first parks its accumulators; the second keeps its own in registers, waits
        // for the first one's "slab drained" mark and adds that slab (own + other == other + own bit for bit).  Before, both slices parked
        // and drained 256 KiB per workgroup ahead of the ticket: twice the slab traffic, and ~5 us of store drain on the critical path of
        // the workgroup that finishes the tile (profiles/r06_launch_anatomy_b4.log: loop end -> last store 14.6 us median with the
        // reduction against 6 us with plain slabs).  Counter: +1 per arrival, +2 when the first slice's stores have drained; the second
        // arriver re-arms it.  The first arriver is running by construction when the second waits for it -- no residency assumption.
        int* const flag = reinterpret_cast<int*>(smem);
        int* const ticket = p.cnt + tile;
        if (!p.red_first) {                                                // ("red.first" = 0, the round-5 form: both slices park, then the ticket)
#pragma unroll
            for (int a = 0; a < MB * NPW; ++a)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(acc[a][4 * r4]), __float_as_uint(acc[a][4 * r4 + 1]),
                                                                   __float_as_uint(acc[a][4 * r4 + 2]), __float_as_uint(acc[a][4 * r4 + 3])},
                                                           sl, slab_off(s, a, r4), 0, SC1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (tid == 0) flag[0] = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (!p.red_first && flag[0] == 0) { BD_KSTAMP_END(p.stamp); return; }
        if (p.red_first && flag[0] == 0) {                                 // first slice of the tile: park, drain, mark, leave
#pragma unroll
            for (int a = 0; a < MB * NPW; ++a)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    __builtin_amdgcn_raw_buffer_store_b128((u32x4){__float_as_uint(acc[a][4 * r4]), __float_as_uint(acc[a][4 * r4 + 1]),
                                                                   __float_as_uint(acc[a][4 * r4 + 2]), __float_as_uint(acc[a][4 * r4 + 3])},
                                                           sl, slab_off(s, a, r4), 0, SC1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(ticket, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            BD_KSTAMP_END(p.stamp);
            return;
        }
        if (p.red_first && tid == 0) {                                     // second slice: the other slab must have drained (bounded: ~2 s)
            const long long t0 = wall_clock64();
            while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 3 && wall_clock64() - t0 < 200000000LL) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
        // the other slice's slab in batches of 8 accumulators (32 loads of 16 B in flight per lane: the operand registers of the K loop are dead
        // here); one accumulator at a time was 16 dependent round trips to write-through lines of another XCD
constexpr int RB_ = 4;
#pragma unroll
        for (int a0 = 0; a0 < MB * NPW; a0 += RB_) {
            u32x4 v[RB_][4];
#pragma unroll
            for (int a = 0; a < RB_; ++a)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) v[a][r4] = __builtin_amdgcn_raw_buffer_load_b128(sl, slab_off(1 - s, a0 + a, r4), 0, SC1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int a = 0; a < RB_; ++a)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[a0 + a][4 * r4 + j] += __uint_as_float(v[a][r4][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tid == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
    }
    // ---- epilogue (the values of gemm_kernel's forms).  bf16 / SwiGLU outputs leave through a per-wave LDS patch as 16 B per lane
    // (round 6): straight from the accumulators a store instruction writes 2 bytes per lane -- 256 partial-line store instructions per
    // wave, which is what made the fused SwiGLU behind the in-launch reduction 25 us slower per launch than slabs + swiglu_rows at
    // num_images = 4 (profiles/r05_head_sweep_b4.log).  The A tiles are dead here (every wave passed the loop's last barrier after its
    // last fragment read); the first 1 KiB stays the reduction's flag word.
    const int col = nb * 32 + (lane & 31);
    float bias_pn[NPW];
#pragma unroll
    for (int pn = 0; pn < NPW; ++pn) bias_pn[pn] = (EPI != BD_EPI_PARTIAL && p.bias) ? bf2f(p.bias[col + pn * 32]) : 0.f;
    if constexpr (EPI == BD_EPI_BF16) {
        // row-major bf16: two alternating patches of [32 rows][64 columns (+ 8 pad)] per wave; read back as 8 rows x 128 contiguous bytes
        constexpr int PITCH = 72;
        bf16_t* const patch = reinterpret_cast<bf16_t*>(smem + 1024) + wave * (2 * 32 * PITCH);
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            bf16_t* const pt = patch + (m & 1) * (32 * PITCH);
#pragma unroll
            for (int pn = 0; pn < NPW; ++pn)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    pt[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * PITCH + pn * 32 + (lane & 31)] = f2bf(acc[m * NPW + pn][r] + bias_pn[pn]);
            __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0): this wave's own writes (no other wave touches the patch)
            bf16_t* const o = p.act + (size_t)(mt * MB + m) * 32 * p.N + (size_t)nb * 32;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int row = it * 8 + (lane >> 3), seg = lane & 7;
                *reinterpret_cast<u32x4*>(o + (size_t)row * p.N + seg * 8) = *reinterpret_cast<const u32x4*>(pt + row * PITCH + seg * 8);
            }
        }
        BD_KSTAMP_END(p.stamp);
        return;
    }
    if constexpr (EPI == BD_EPI_SWIGLU) {
        // one accumulator (32 rows x 32 packed columns = 16 gate + 16 up features) = exactly ONE 1 KiB chunk of the fragment-major
        // operand: (row block, k-step = packed panel).  Every lane ends up with feature l & 15 of 8 rows (swiglu_pairs): 8 two-byte LDS
        // writes into the chunk's layout, then the chunk goes out as one coalesced 16 B-per-lane store.
        bf16_t* const patch = reinterpret_cast<bf16_t*>(smem + 1024) + wave * (MB * NPW * 512);     // 16 chunks of 1 KiB per wave
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int pn = 0; pn < NPW; ++pn) {
                bf16_t* const ch = patch + (m * NPW + pn) * 512;
                const int f = lane & 15;
                bf16_t o8[8];
                swiglu_pairs(acc[m * NPW + pn], bias_pn[pn], lane, o8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = 2 * j + ((lane >> 4) & 1);
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    ch[(row + 32 * (f >> 3)) * 8 + (f & 7)] = o8[j];
                }
            }
        __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int pn = 0; pn < NPW; ++pn) {
                const u32x4 v = reinterpret_cast<const u32x4*>(patch + (m * NPW + pn) * 512)[lane];
                reinterpret_cast<u32x4*>(p.act)[((size_t)(nb + pn) * p.RB + (mt * MB + m)) * 64 + lane] = v;
            }
        BD_KSTAMP_END(p.stamp);
        return;
    }
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int pn = 0; pn < NPW; ++pn) {
            const f32x16& a = acc[m * NPW + pn];
            float* o = p.out + ((size_t)s * p.Mpad + (size_t)(mt * MB + m) * 32) * p.N + col + pn * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * p.N] = a[r];
        }
    BD_KSTAMP_END(p.stamp);
}

// measurement switches of the 256-row kernel (process-wide; bd_set_gemm_option): W register ring depth and XCD placement
static int g_wide_keep = -1;                              // weight loads of the 256-row kernel: -1 = default policy when > 1 row tile, 0 = always nt, 1 = never nt
static int g_wide_ring = 2, g_wide_xcd = -1;            // xcd: -1 = by shape (on when the weights outweigh the rows), 0 / 1 forced
void bdk_gemm_tile_debug(int v);
void bdk_gemm_half_form(int v);
// tensor parallelism: the push target of the NEXT bdk_gemm call with the fp32-partial epilogue on the 128-row bf16 kernel (set by
// bd_api.hip linear_rowsplit through bd_comm.hip bdk_tp_push_target; consumed and reported by bdk_gemm_push_used)
static thread_local BdTpPush g_push;
static thread_local bool g_push_set = false, g_push_used = false;
void bdk_gemm_set_push(const BdTpPush* t) { g_push_set = t != nullptr; if (t) g_push = *t; g_push_used = false; }
// sequence-parallel tensor parallelism: the NEXT bdk_gemm / bdk_gemm8 / bdk_gemm8a call on the 128-row kernel waits for its operand rows
static thread_local BdHWait g_hwait;
static thread_local bool g_hwait_set = false;
void bdk_gemm_set_hwait(const BdHWait* w) { g_hwait_set = w != nullptr; if (w) g_hwait = *w; }
bool bdk_gemm_take_hwait(BdHWait* out) { const bool s = g_hwait_set; if (s) *out = g_hwait; g_hwait_set = false; return s; }
// the pending push target, if this launch can take it (fp32-partial epilogue of the 128-row kernel, whole 8-row groups per rank)
bool bdk_gemm_claim_push(int epi, int RB, int N, BdTpPush* out) {
    const bool ok = g_push_set && epi == BD_EPI_F32 && RB % 4 == 0 && RB < 8 && g_push.size > 1 && g_push.rows_per_rank % 8 == 0 && N % 32 == 0;
    if (ok) { *out = g_push; g_push_used = true; }
    g_push_set = false;
    return ok;
}
bool bdk_gemm_push_used() { const bool u = g_push_used; g_push_used = false; g_push_set = false; return u; }
static int g_red_first = 1;
int bdk_red_first() { return g_red_first; }
static int g_tile_minrb = 32;                            // row blocks from which the tiled kernel takes over ("tile.minrb": 16 = from 512 rows)
static int g_half = 1;                                   // 512-768 rows, one K slice, bf16 / SwiGLU output: the 256 x 128-tile kernel (bd_gemm_half.hip); 0 = off, 2 = any N
static int g_tile = 1;                                   // >= 512 rows: the LDS-tiled MFMA-bound kernel (bd_gemm_tile.hip); 0 = 256-row kernel
int bdk_set_gemm_option(const char* name, int v) {
    const std::string n(name);
    // 0: 256-row kernel; 1: tiled kernel, operand fetch by shape; 2 / 3 / 4: tiled kernel, register-staged / LDS-DMA / W-in-registers fetch forced
    if (n == "tile" && v >= 0 && v <= 4) { g_tile = v ? 1 : 0; bdk_gemm_tile_stg(v == 2 ? 1 : (v == 3 ? 0 : (v == 4 ? 2 : -1))); return 0; }
    if (n == "tile.debug" && v >= 0 && v <= 3) { bdk_gemm_tile_debug(v); return 0; }
    if (n == "tile.minrb" && v >= 8 && v % 8 == 0) { g_tile_minrb = v; return 0; }
    if (n == "wide.ring" && (v == 2 || v == 3)) { g_wide_ring = v; return 0; }
    if (n == "wide.xcd" && v >= -1 && v <= 1) { g_wide_xcd = v; return 0; }
    if (n == "wide.keep" && v >= -1 && v <= 1) { g_wide_keep = v; return 0; }
    if (n == "red.first" && v >= 0 && v <= 1) { g_red_first = v; return 0; }
    if (n == "half" && v >= 0 && v <= 2) { g_half = v; return 0; }
    if (n == "half.form" && (v == 0 || v == 1 || v == 4 || v == 12)) { bdk_gemm_half_form(v); return 0; }
    if (n.rfind("rows.", 0) == 0) return bdk_set_rows_option(name, v);
    return -1;
}

template <int WR, bool XCD>
static int launch_gemm_wide_v(const GemmP& p, int epi, hipStream_t st) {
    const int ntiles = p.N / 256;
    dim3 grid(ntiles * p.S * (p.RB / 8));
    if (p.S > 1 && epi != BD_EPI_PARTIAL) {                       // two slices reduced in the launch (RED)
        if (p.S != 2 || !p.cnt || !p.out || (long long)ntiles * (p.RB / 8) > 16383) return -4;
        static unsigned long long optin_r[2] = {0, 0};
        const int nr = 3 * 8 * 256 * 16;
        if (!bd_lds_optin((const void*)gemm_wide_kernel<BD_EPI_BF16, WR, XCD, true>, nr, &optin_r[0]) ||
            !bd_lds_optin((const void*)gemm_wide_kernel<BD_EPI_SWIGLU, WR, XCD, true>, nr, &optin_r[1])) return -8;
        if (epi == BD_EPI_BF16) BD_LAUNCH((gemm_wide_kernel<BD_EPI_BF16, WR, XCD, true>), grid, dim3(256), (size_t)nr, st, p);
        else if (epi == BD_EPI_SWIGLU) BD_LAUNCH((gemm_wide_kernel<BD_EPI_SWIGLU, WR, XCD, true>), grid, dim3(256), (size_t)nr, st, p);
        else return -4;
        return bd_launch_status();
    }
    const size_t lds = (size_t)3 * 8 * 256 * 16;
    static unsigned long long optin[3] = {0, 0, 0};
    const int n = 3 * 8 * 256 * 16;                                // 96 KiB of dynamic LDS needs the opt-in, per device
    const bool lds_ok = bd_lds_optin((const void*)gemm_wide_kernel<BD_EPI_PARTIAL, WR, XCD>, n, &optin[0]) &&
                        bd_lds_optin((const void*)gemm_wide_kernel<BD_EPI_BF16, WR, XCD>, n, &optin[1]) &&
                        bd_lds_optin((const void*)gemm_wide_kernel<BD_EPI_SWIGLU, WR, XCD>, n, &optin[2]);
    if (!lds_ok) return -8;
    if (epi == BD_EPI_PARTIAL) BD_LAUNCH((gemm_wide_kernel<BD_EPI_PARTIAL, WR, XCD>), grid, dim3(256), lds, st, p);
    else if (epi == BD_EPI_BF16) BD_LAUNCH((gemm_wide_kernel<BD_EPI_BF16, WR, XCD>), grid, dim3(256), lds, st, p);
    else BD_LAUNCH((gemm_wide_kernel<BD_EPI_SWIGLU, WR, XCD>), grid, dim3(256), lds, st, p);
    return bd_launch_status();
}

static int launch_gemm_wide(const GemmP& p0, int epi, hipStream_t st) {
    GemmP p = p0;
    p.red_first = g_red_first;
#ifdef BD_GEMM_STAMP
    p.stamp = bdk_stamp_next((std::string("wide:") + bdk_stamp_current_label()).c_str(), (p.N / 256) * p.S * (p.RB / 8));
#endif
    p.w_keep = g_wide_keep < 0 ? (p.RB > 8) : g_wide_keep;
    // several row tiles per weight slice: keep them on one XCD when the slice is what dominates the traffic (N columns of
    // weights against RB * 32 rows of activations per K); a large batch (ImageNet: 12 288 rows) is the other way round.
    // Measured at 512 rows (profiles/r02_gemm_sweep3.log): adaLN 311 vs 352 us, gate/up 196 vs 207, wo / w2 (5 slices) 34.5 / 40.9
    // vs 36.0 / 43.5 -- but the 2-slice qkv / w1 shapes LOSE (73 vs 68.5 us: with an even slice count every XCD then works on
    // one K half only, and the two readers of a slice contend for the same L2 channels in lockstep), so: odd slice counts only.
    // round 5: the 2-slice shapes join -- with the slice rotated by the group index (gemm_wide_kernel) every XCD walks both slices
    // and the placement wins there as well (num_images = 4, in situ: 2347.5 vs 2387.8 us per evaluation, profiles/r05_head_sweep_b4.log)
    const bool xcd = (p.RB > 8) && (g_wide_xcd < 0 ? (p.N > p.RB * 32 && ((p.S & 1) || p.S == 2)) : g_wide_xcd == 1);
    if (g_wide_ring == 3) return xcd ? launch_gemm_wide_v<3, true>(p, epi, st) : launch_gemm_wide_v<3, false>(p, epi, st);
    return xcd ? launch_gemm_wide_v<2, true>(p, epi, st) : launch_gemm_wide_v<2, false>(p, epi, st);
}

// `nw_ring` = waves per workgroup (2, 4, 8, 10) + 16 * ring + 256 * (kw - 1) + 2048 * pipe: ring in {0 (=2), 3, 4} = stages of
// W/A a wave keeps in flight, kw in {1, 2} = waves that share one 32-column panel and split each K stage (NP = waves / kw
// panels per tile), pipe = LDS reads one stage ahead of the MFMAs (falls back to the plain loop where not instantiated).
int bdk_gemm(const void* A, int RB, const void* W, int N, int K, int S, int nw_ring, int epi,
             float* out_partial, void* out_act, const void* bias, int* cnt, hipStream_t st, const float* wscale) {
    if (wscale) return bdk_gemm8(A, RB, W, wscale, N, K, S, nw_ring, epi, out_partial, out_act, bias, cnt, st);   // fp8 weights: bd_gemm8.hip
    // the armed push target / operand wait belong to THIS call whatever happens to it: taken (and cleared) before any early return, so a
    // rejected shape cannot leave them attached to the next, unrelated GEMM of the thread
    BdTpPush pend_push; const bool have_push = bdk_gemm_claim_push(epi, RB, N, &pend_push);
    BdHWait pend_hw; const bool have_hw = bdk_gemm_take_hwait(&pend_hw);
    const int nw = nw_ring & 15;
    int ring = (nw_ring >> 4) & 15;
    const int kw = ((nw_ring >> 8) & 3) + 1;
    const bool pipe = (nw_ring >> 11) & 1;             // + 2048: software-pipelined LDS reads (4-wave workgroups, ring 2)
    if (ring == 0) ring = 2;
    if (ring < 2 || ring > 4 || kw > 2 || nw % kw) return -7;
    const int np = nw / kw;
    // a ragged last tile (N/32 not a multiple of the panels per workgroup): the waves past the last panel repeat it and store nothing
    if (K % (64 * kw) || N % 32 || S < 1) return -2;
    const int nst_total = K / (64 * kw), q = (nst_total + S - 1) / S;
    if ((S - 1) * q >= nst_total) return -3;                       // an empty split
    if (epi != BD_EPI_PARTIAL && S != 1 && (out_partial == nullptr || cnt == nullptr || (nw >= 9 && kw == 1))) return -4;   // needs slab scratch + counters
    size_t PS, SS;
    bdk_w_strides(N / 32, K, &PS, &SS);
    GemmP p{(const u32x4*)A, (const u32x4*)W, out_partial, (bf16_t*)out_act, (const bf16_t*)bias, cnt, nullptr, RB, N, K, S, RB * 32, PS, SS};
    if (have_push) p.push = pend_push;                 // the 128-row kernel's epilogue pushes the peers' slices itself (bd_gemm_kernel.h)
    if (have_hw) { if (RB % 8 == 0 || pipe) return -10; p.hw = pend_hw; }     // (the 128-row plain-loop kernel only)
    // rows per pass over the weights: 256 (two images with CFG: W streamed once for both) when the row count allows,
    // else 128 / 64 / 32
    const int MB = (RB % 8 == 0 && nw >= 4 && kw == 1) ? 8 : ((RB % 4 == 0) ? 4 : RB);
    if (MB != 8 && MB != 4 && MB != 2 && MB != 1) return -5;
    if (MB == 8 || nw == 2 || nw >= 9 || (kw == 2 && !(nw == 4 && MB == 4 && ring == 3))) ring = 2;        // register budget
    // 256-row passes over 256-column tiles: 4 waves x 2 panels (MFMA-friendly) instead of 8 waves x 1 panel;
    // same grid.  BD_GEMM_WIDE=0 keeps the 8-wave form (A/B switch for measurements).
    static const bool wide = [] { const char* e = getenv("BD_GEMM_WIDE"); return !(e && e[0] == '0'); }();
    // (a reduced epilogue -- S > 1 with bf16 / SwiGLU output -- exists for exactly two slices, round 5; more slices: the generic kernel)
    // 512 / 768 rows at ONE K slice with a rounded output (bd_api.hip choose_cfg "tune.half"): 256 x 128 tiles, split-K inside the workgroup
    // (nw + 16: the caller asks for this kernel -- bd_api.hip GemmCfg::half; "half" = 2 routes every fitting shape, tests)
    if (g_half && MB == 8 && nw == 8 && RB >= 16 && RB < g_tile_minrb && g_w_layout == 0 && N % 128 == 0 && K % 64 == 0 && !have_hw &&
        ((S == 1 && (epi == BD_EPI_BF16 || epi == BD_EPI_SWIGLU)) || epi == BD_EPI_PARTIAL) && (K / 64) / S >= 2 &&
        (g_half == 2 || ((nw_ring >> 12) & 1)))
        return bdk_gemm_half(p, epi, st);
    if (wide && MB == 8 && nw == 8 && N % 256 == 0 && epi != BD_EPI_F32 && !(S > 1 && epi != BD_EPI_PARTIAL && (S != 2 || RB < 16))) {
        // >= 512 rows: the matrix pipe is the roofline -> both operands through LDS, 256 x 256 tiles (bd_gemm_tile.hip)
        // (measured, profiles/r03_gemm_tile_v4.log: ahead of the 256-row kernel from 1024 rows on wide N -- adaLN x8 693 vs 786 us,
        // ImageNet w1 89 vs 114 us -- behind it at 512 rows and on narrow N, where it has too few tiles per CU)
        if (g_tile && RB >= g_tile_minrb && N >= 4096 && g_w_layout == 0 && !(S > 1 && epi != BD_EPI_PARTIAL)) return bdk_gemm_tile(p, epi, st);
        return launch_gemm_wide(p, epi, st);
    }
#define BD_CASE(NPV, KWV, MBV, RV) if (np == NPV && kw == KWV && MB == MBV && ring == RV) return launch_gemm<NPV, KWV, MBV, RV>(p, epi, st);
#define BD_CASE_PIPE(NPV, KWV, MBV) if (pipe && np == NPV && kw == KWV && MB == MBV) return launch_gemm<NPV, KWV, MBV, 2, 1>(p, epi, st);
    BD_CASE_PIPE(4, 1, 4) BD_CASE_PIPE(2, 1, 4) BD_CASE_PIPE(2, 2, 4) BD_CASE_PIPE(4, 1, 2) BD_CASE_PIPE(4, 1, 1)
#undef BD_CASE_PIPE
    if (MB <= 2 && ring == 3) ring = 4;
    if (MB == 1) ring = 2;
    BD_CASE(4, 1, 8, 2) BD_CASE(8, 1, 8, 2) BD_CASE(10, 1, 4, 2) BD_CASE(9, 1, 4, 2) BD_CASE(5, 1, 4, 2)
    BD_CASE(2, 1, 4, 2) BD_CASE(4, 1, 4, 2) BD_CASE(8, 1, 4, 2) BD_CASE(4, 1, 4, 3) BD_CASE(8, 1, 4, 3) BD_CASE(4, 1, 4, 4) BD_CASE(8, 1, 4, 4)
    BD_CASE(2, 1, 2, 2) BD_CASE(4, 1, 2, 2) BD_CASE(8, 1, 2, 2) BD_CASE(4, 1, 2, 4) BD_CASE(8, 1, 2, 4)
    BD_CASE(2, 1, 1, 2) BD_CASE(4, 1, 1, 2) BD_CASE(8, 1, 1, 2)
    BD_CASE(1, 2, 4, 2) BD_CASE(2, 2, 4, 2) BD_CASE(4, 2, 4, 2) BD_CASE(5, 2, 4, 2) BD_CASE(2, 2, 4, 3)
    BD_CASE(1, 2, 2, 2) BD_CASE(2, 2, 2, 2) BD_CASE(4, 2, 2, 2)
    BD_CASE(1, 2, 1, 2) BD_CASE(2, 2, 1, 2) BD_CASE(4, 2, 1, 2)
#undef BD_CASE
    return -6;
}

// measurement support: pure HBM read stream with the GEMM's access pattern (16 B/lane non-temporal loads), result
// folded into one word per thread so the loads cannot be elided.  Gives the practical read roofline of this chip.
__global__ __launch_bounds__(256) void probe_read_kernel(const u32x4* __restrict__ src, size_t n16, unsigned* sink) {
    u32x4 acc = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const u32x4 a = __builtin_nontemporal_load(src + i), b = __builtin_nontemporal_load(src + i + stride);
        const u32x4 c = __builtin_nontemporal_load(src + i + 2 * stride), d = __builtin_nontemporal_load(src + i + 3 * stride);
        acc ^= a ^ b ^ c ^ d;
    }
    for (; i < n16; i += stride) acc ^= __builtin_nontemporal_load(src + i);
    const unsigned r = acc[0] ^ acc[1] ^ acc[2] ^ acc[3];
    if (r == 0x12345678u) sink[0] = r;
}
int bdk_probe_read(const void* src, size_t bytes, int blocks, void* sink, hipStream_t st) {
    BD_LAUNCH(probe_read_kernel, dim3(blocks), dim3(256), 0, st, (const u32x4*)src, bytes / 16, (unsigned*)sink);
    return bd_launch_status();
}

int bdk_pack_w(void* dst, const void* src, const void* src2, int panels, int K, int nb0, int panels_total, int mode, hipStream_t st) {
    size_t PS, SS;
    bdk_w_strides(panels_total, K, &PS, &SS);
    if (K % 16 || nb0 + panels > panels_total) return -2;          // whole k-steps; the GEMM / conv launchers check their own stage depth
    const size_t total = (size_t)panels * (K / 16) * 64;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    BD_LAUNCH(pack_w_kernel, dim3(blocks), dim3(256), 0, st, (u32x4*)dst, (const bf16_t*)src,
                       (const bf16_t*)src2, panels, K, nb0, mode, PS, SS);
    return bd_launch_status();
}

int bdk_rows_to_afrag(void* dst, const float* src32, const void* src16, int M, int K, int RB, hipStream_t st) {
    if (K % 8) return -2;
    const size_t total = (size_t)M * (K / 8);
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    BD_LAUNCH(rows_to_afrag_kernel, dim3(blocks), dim3(256), 0, st, (bf16_t*)dst, src32,
                       (const bf16_t*)src16, M, K, RB);
    return bd_launch_status();
}
