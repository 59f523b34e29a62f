// MFMA-bound GEMM for MANY rows (>= 512): out[M, N] = A[M, K] (bf16) x W[N, K]^T (bf16), fp32 accumulate.
//
// Who launches it: the adaLN projections of a whole group of evaluations (bd_api.hip head_ada_group: G x 128 rows per pass
// over the 734 MB adaLN matrix, flow_head_parallel_x.py:331), the eval batch (num_images = 4: 512 rows, eval/eval_dpg.py:44)
// and the ImageNet batch (384 classes with CFG: 12 288 rows, imagenet_gen/sample_ddp_parallel.py:199-214).  At these row
// counts the matrix pipe, not HBM, is the roofline (arithmetic intensity = rows FLOP per weight byte >> the ~310 FLOP/B
// ridge), so unlike the weight-streaming kernels of bd_gemm.hip BOTH operands are staged through LDS and shared by the whole
// workgroup:
//
//   * workgroup tile 256 rows x 256 columns, 8 waves as 2 (rows) x 4 (columns), wave tile 128 x 64 = 8 accumulators of
//     32x32 (128 AGPR/VGPRs); per k-step of 16 a wave reads 4 A + 2 W fragments (6 ds_read_b128, lane-linear, conflict free:
//     both operands are ALREADY stored in MFMA-operand order in HBM -- bd_common.h afrag_off, bd_gemm.hip pack_w_kernel -- so
//     a 1 KiB chunk is one fragment for all 64 lanes) for 8 v_mfma_f32_32x32x16_bf16: 128 FLOP per LDS byte;
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4: one instruction moves one 1 KiB fragment chunk, no VGPR round trip,
//     no ds_write issue slots), 4 chunks per wave per 32-deep K stage, into a ring of four 32 KiB stages; the DMA of stage
//     j + 3 is issued right behind the barrier that opens stage j, so three stages (~1.3 us of MFMA time) of L2 / HBM latency
//     are covered.  The DMAs are inline asm with hand-counted s_waitcnt vmcnt: hipcc counts an LDS-DMA builtin as a store to
//     all of LDS and drains the queue before the next ds_read (no pipelining at all);
//     (second form, STG = true, default from 64 stages per slice: the same chunks by plain 16 B global loads into three rotating
//     register sets, written to their slot one stage ahead -- 2-6 % faster on long K, see the main-loop comment);
//   * the two waves of every SIMD run half a stage apart (ping-pong): while one issues its 16 MFMAs of a stage from registers
//     the other reads its next fragments from LDS and issues / awaits its DMAs, one s_barrier per half-step (main-loop comment);
//   * workgroup -> tile map: block b runs on XCD b % 8 (observed placement, used for speed only).  The 32 blocks an XCD runs
//     side by side form an 8 x 4 rectangle of tiles that walks K in step, so an A stage is fetched into that L2 once per 4
//     column tiles and a W stage once per 8 row tiles: HBM / Infinity-Cache traffic (1/4 + 1/8) of the operand bytes per tile.
//
// Each accumulator sees its K steps in ascending order through the same MFMA as in the 128-row kernel (bd_gemm_kernel.h), so
// for S = 1 the results are bit-identical to that kernel's (tests/test_gpu_parity.py::test_gemm_tile_kernel_both_fetch_forms).
#include "bd_gemm_kernel.h"

namespace {

constexpr int TS_STAGE_UNITS = 32 * 64;        // 16 B units per 32-deep stage: 16 A chunks + 16 W chunks of 64 units (1 KiB)
constexpr int TS_SLOTS = 4;

// one 1 KiB fragment chunk, global -> LDS; `lds_byte` wave-uniform.  M0 is saved / restored: it is compiler-reserved.
BD_DEV void dma_chunk(const u32x4* gsrc, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}

}  // namespace

template <int EPI, int STG = 0>
__global__ __launch_bounds__(512) void gemm_tile_kernel(GemmP p, int dbg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const u32x4* const lds = reinterpret_cast<const u32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;                    // 128-row half, 64-column quarter of the tile

    // ---- tile of this workgroup (header comment: 8 x 4 rectangles per XCD)
    const int RT = p.RB >> 3, NTS = (p.N >> 8) * p.S;           // row tiles, (column tile, K slice) pairs
    const int nRm = (RT + 7) >> 3, nRn = (NTS + 3) >> 2;
    const int b = blockIdx.x, x = b & 7, i = b >> 3;
    const int R = (i >> 5) * 8 + x, within = i & 31;
    if (R >= nRm * nRn) return;
    const int mt = (R % nRm) * 8 + (within & 7), cs = (R / nRm) * 4 + (within >> 3);
    if (mt >= RT || cs >= NTS) return;
    const int s = cs % p.S, nt = cs / p.S;
    const int nst_total = p.K >> 5;                             // 32-deep stages
    const int q = (nst_total + p.S - 1) / p.S;
    const int st0 = s * q;
    const int nst = min(q, nst_total - st0);

    // ---- this wave's 4 DMA chunks per stage: waves 0-3 fetch A (chunk c = 4 wave + i: k-step c >> 3, row block c & 7),
    //      waves 4-7 fetch W (c' = 4 (wave - 4) + i: panel c' >> 1, k-step c' & 1).  LDS stage = [16 A chunks | 16 W chunks].
    const u32x4* src[4];
    unsigned dst[4];
    size_t stride;                                              // 16 B units per stage
    if (wave < 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = wave * 4 + j, ks = c >> 3, rb = c & 7;
            src[j] = p.A + ((size_t)(st0 * 2 + ks) * p.RB + mt * 8 + rb) * 64 + lane;
            dst[j] = (unsigned)c * 1024u;
        }
        stride = (size_t)2 * p.RB * 64;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = (wave - 4) * 4 + j, pn = c >> 1, ks = c & 1;
            src[j] = p.W + (size_t)(nt * 8 + pn) * p.PS + (size_t)(st0 * 2 + ks) * 64 + lane;
            dst[j] = (unsigned)(16 + c) * 1024u;
        }
        stride = 128;
    }
    auto issue = [&](int st) {                                  // stage st (relative) -> slot st % 4
        const unsigned slot = (unsigned)(st & (TS_SLOTS - 1)) * (TS_STAGE_UNITS * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_chunk(src[j] + (size_t)st * stride, slot + dst[j]);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // ---- main loop: PING-PONG between the two waves of each SIMD.  Waves w and w + 4 share a SIMD (waves are dealt to the SIMDs
    // cyclically), so the workgroup is two groups, X = waves 0-3 and Y = waves 4-7, half a stage apart: in every half-step one
    // group issues nothing but its 16 MFMAs of a stage (fragments already in registers) while the other reads its 12 fragments
    // of the next stage from LDS, issues its 4 DMA chunks three stages ahead and waits for them -- the matrix pipe of every SIMD
    // always has one wave feeding it, and LDS / DMA latency sits under the partner's MFMA segment (MI355X_MICROARCH "two waves per
    // SIMD").  One s_barrier per half-step; X fetches A chunks, Y fetches W chunks.
    //   half-step 2j    :  X  LOAD(j)   |  Y  MFMA(j - 1)
    //   half-step 2j + 1:  X  MFMA(j)   |  Y  LOAD(j)
    // LOAD(j) = read this wave's fragments of stage j, then wait until the own DMAs of stage j + 1 have landed (X: A(j + 1), needed
    // from half-step 2j + 2; Y: W(j + 1), likewise) with one younger stage still in flight; MFMA(j) also issues the own DMAs of
    // stage j + 3.
    u32x4 af[2][4], wf[2][2];
    const bool isX = wave < 4;
    auto load_seg = [&](int j) {
        const u32x4* a = lds + (size_t)(j & (TS_SLOTS - 1)) * TS_STAGE_UNITS + lane;
        const u32x4* w = a + 16 * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int m = 0; m < 4; ++m) af[ks][m] = a[(ks * 8 + (wr * 4 + m)) * 64];
#pragma unroll
            for (int n = 0; n < 2; ++n) wf[ks][n] = w[((wc * 2 + n) * 2 + ks) * 64];
        }
        // own DMAs of stage j + 1 landed (issued during MFMA(j - 2)); those of stage j + 2 (MFMA(j - 1)) may stay in flight
        if ((dbg & 2) || j + 2 >= nst) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    };
    // 16 MFMAs of stage j from registers, with this wave's 4 DMA chunks of stage j + 3 issued BETWEEN them (one per 4 MFMAs): a
    // DMA issued beside ds_reads in the load segment costs the wave 100-185 cycles and made that segment longer than the MFMA
    // segment it is supposed to hide under; between MFMAs the wave is waiting on the matrix pipe anyway.  The slot is that of
    // stage j - 1, whose last readers (the partner's LOAD(j - 1)) finished before the barrier that opened this half-step.
    auto mfma_seg = [&](int j) {
        const bool dma = (j + 3 < nst) && !(dbg & 2);
        const unsigned slot = (unsigned)((j + 3) & (TS_SLOTS - 1)) * (TS_STAGE_UNITS * 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = mfma32(af[ks][m], wf[ks][n], acc[m][n]);
                if ((m & 1) == 1) {
                    const int c = ks * 2 + (m >> 1);
                    if (dma) dma_chunk(src[c] + (size_t)(j + 3) * stride, slot + dst[c]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    };

    if constexpr (STG == 2) {
        // W STRAIGHT INTO REGISTERS (tile option 4, round 6; an A/B form, not the default).  A wave's two W fragments of a k-step are whole 1 KiB
        // chunks in HBM already and only the wave of the other row half shares them, so every wave loads its own W (the partner's copy is an
        // L1 / L2 hit) and only A goes through LDS: per wave and stage 2 LDS stores instead of 4, 8 fragment reads instead of 12, 6 global loads
        // instead of 4.  A: every wave fetches 2 of the stage's 16 chunks three stages ahead (three register sets) and parks them one stage
        // before they are read; W of stage j + 2 is loaded during MFMA(j) over the set it is reading, each load behind the last MFMA that reads
        // its register.  Same MFMAs in the same order as the other two forms: bit-identical results.
        // Why it exists and what it showed: in bd_gemm_half.hip the LDS stores are the largest single cost of the loop, and THIS loop with its
        // stores removed runs at 0.70 of the matrix peak instead of 0.50 -- but with half of them removed it runs at 0.48 (adaLN x16 1252 vs
        // 1224 us, qkv at 2048 / 1024 rows 266 / 134 vs 268 / 137, ImageNet w1 81 vs 83): the 0.70 was the power governor's answer to operands
        // that stop changing (an LDS nobody writes), not a faster loop.  This kernel sits at the LDS-fed power ceiling of the chip
        // (tools/probe_mfma.hip: 0.62 on random data), whichever way its operands arrive.
        u32x4 sa[3][2], wr_[2][2][2];
        u32x4* const ldsw = reinterpret_cast<u32x4*>(smem);
        // this wave's two A chunks of a stage are adjacent (chunks 2 wave, 2 wave + 1: one k-step, consecutive row blocks); its four W fragments
        // are two panels x two k-steps.  Every address is wave-uniform + lane * 16: buffer loads with the uniform part in SGPRs -- one VGPR of
        // addressing for all six loads of a stage (64-bit per-lane pointers for each stream did not fit beside 216 data registers: hipcc spilled
        // them and reloaded them inside the loop behind s_waitcnt vmcnt(0))
        const int ca = wave * 2;
        const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p.A), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<u32x4*>(p.W), 0, 0x7fffffff, 0x00020000);
        const unsigned lane16 = (unsigned)lane * 16u;
        const unsigned a_base = (unsigned)((((size_t)(st0 * 2 + (ca >> 3)) * p.RB + mt * 8 + (ca & 7)) * 64) * 16);      // bytes (A < 2 GiB, W < 2 GiB: launcher)
        const unsigned a_stride = (unsigned)((size_t)2 * p.RB * 64 * 16);
        const unsigned w_base0 = (unsigned)(((size_t)(nt * 8 + wc * 2) * p.PS + (size_t)(st0 * 2) * 64) * 16);
        const unsigned w_base1 = w_base0 + (unsigned)(p.PS * 16);
        const unsigned adst0 = (unsigned)ca * 64u;
        auto ldb = [&](const __amdgpu_buffer_rsrc_t& r, unsigned soff) -> u32x4 {
            return __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane16, (int)soff, 0);
        };
        auto fetch_a = [&](auto SET, int st) {
            constexpr int Q = decltype(SET)::value;
            const unsigned o = a_base + (unsigned)min(st, nst - 1) * a_stride;
            sa[Q][0] = ldb(ra, o);
            sa[Q][1] = ldb(ra, o + 1024u);
        };
        auto fetch_w = [&](auto SET, int st) {
            constexpr int Q = decltype(SET)::value;
            const unsigned o = (unsigned)min(st, nst - 1) * 2048u;
            wr_[Q][0][0] = ldb(rw, w_base0 + o); wr_[Q][0][1] = ldb(rw, w_base1 + o);
            wr_[Q][1][0] = ldb(rw, w_base0 + o + 1024u); wr_[Q][1][1] = ldb(rw, w_base1 + o + 1024u);
        };
        auto park_a = [&](auto SET, int st) {                // stage st's two chunks of this wave -> slot st & 3 (16 KiB slots: A only)
            constexpr int Q = decltype(SET)::value;
            u32x4* const w = ldsw + (size_t)(st & (TS_SLOTS - 1)) * (16 * 64) + adst0 + lane;
            w[0] = sa[Q][0];
            w[64] = sa[Q][1];
        };
        // LOAD(j), j % 3 == Q: A fragments of stage j, then stage j + 1 (set (Q + 1) % 3) into its slot
        auto load3 = [&](auto SET, int jj) {
            constexpr int Q = decltype(SET)::value;
            const u32x4* a = lds + (size_t)(jj & (TS_SLOTS - 1)) * (16 * 64) + lane;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int m = 0; m < 4; ++m) af[ks][m] = a[(ks * 8 + (wr * 4 + m)) * 64];
            park_a(std::integral_constant<int, (Q + 1) % 3>{}, jj + 1);
        };
        // MFMA(j), j % 3 == QA, j % 2 == QW: 16 MFMAs on wr_[QW]; the A loads of stage j + 3 (set QA, parked since LOAD(j - 1)) behind the first
        // two MFMA pairs; the W loads of stage j + 2 into the SAME set, each behind the last MFMA that reads the register it overwrites
        // (k-step 0 after pair 3, k-step 1 after pair 7): two W sets are enough and a stage is ~3.5 half-steps ahead
        auto mfma3 = [&](auto SETA, auto SETW, int jj) {
            constexpr int QA = decltype(SETA)::value, QW = decltype(SETW)::value;
            const unsigned oa = a_base + (unsigned)min(jj + 3, nst - 1) * a_stride;
            const unsigned ow = (unsigned)min(jj + 2, nst - 1) * 2048u;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = mfma32(af[ks][m], wr_[QW][ks][n], acc[m][n]);
                    const int pr = ks * 4 + m;
                    if (pr < 2) sa[QA][pr] = ldb(ra, oa + pr * 1024u);
                    else if (pr == 4) wr_[QW][0][0] = ldb(rw, w_base0 + ow);
                    else if (pr == 5) wr_[QW][0][1] = ldb(rw, w_base1 + ow);
                    else if (pr == 7) {
                        wr_[QW][1][0] = ldb(rw, w_base0 + ow + 1024u);
                        wr_[QW][1][1] = ldb(rw, w_base1 + ow + 1024u);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
#define TILE_LD(QA, jj) do { __syncthreads(); load3(QA{}, jj); } while (0)
#define TILE_MF(QA, QW, jj) do { __syncthreads(); mfma3(QA{}, QW{}, jj); } while (0)
        fetch_a(I0{}, 0); fetch_w(I0{}, 0); fetch_a(I1{}, 1); fetch_w(I1{}, 1); fetch_a(I2{}, 2);
        park_a(I0{}, 0);
        if (isX) {
            int j = 0;
            for (; j + 5 < nst; j += 6) {
                TILE_LD(I0, j); TILE_MF(I0, I0, j);         TILE_LD(I1, j + 1); TILE_MF(I1, I1, j + 1); TILE_LD(I2, j + 2); TILE_MF(I2, I0, j + 2);
                TILE_LD(I0, j + 3); TILE_MF(I0, I1, j + 3); TILE_LD(I1, j + 4); TILE_MF(I1, I0, j + 4); TILE_LD(I2, j + 5); TILE_MF(I2, I1, j + 5);
            }
            if (j < nst) { TILE_LD(I0, j); TILE_MF(I0, I0, j); }
            if (j + 1 < nst) { TILE_LD(I1, j + 1); TILE_MF(I1, I1, j + 1); }
            if (j + 2 < nst) { TILE_LD(I2, j + 2); TILE_MF(I2, I0, j + 2); }
            if (j + 3 < nst) { TILE_LD(I0, j + 3); TILE_MF(I0, I1, j + 3); }
            if (j + 4 < nst) { TILE_LD(I1, j + 4); TILE_MF(I1, I0, j + 4); }
            __syncthreads();                                    // Y has read its last fragments: LDS is free for the epilogue
            BD_MFMA_DRAIN();
        } else {
            __syncthreads();
            TILE_LD(I0, 0);
            int j = 0;
            for (; j + 6 < nst; j += 6) {
                TILE_MF(I0, I0, j);     TILE_LD(I1, j + 1); TILE_MF(I1, I1, j + 1); TILE_LD(I2, j + 2); TILE_MF(I2, I0, j + 2); TILE_LD(I0, j + 3);
                TILE_MF(I0, I1, j + 3); TILE_LD(I1, j + 4); TILE_MF(I1, I0, j + 4); TILE_LD(I2, j + 5); TILE_MF(I2, I1, j + 5); TILE_LD(I0, j + 6);
            }
            TILE_MF(I0, I0, j);                                 // 1-6 stages left, the fragments of stage j are loaded
            if (j + 1 < nst) {
                TILE_LD(I1, j + 1); TILE_MF(I1, I1, j + 1);
                if (j + 2 < nst) {
                    TILE_LD(I2, j + 2); TILE_MF(I2, I0, j + 2);
                    if (j + 3 < nst) {
                        TILE_LD(I0, j + 3); TILE_MF(I0, I1, j + 3);
                        if (j + 4 < nst) {
                            TILE_LD(I1, j + 4); TILE_MF(I1, I0, j + 4);
                            if (j + 5 < nst) { TILE_LD(I2, j + 5); TILE_MF(I2, I1, j + 5); }
                        }
                    }
                }
            }
            BD_MFMA_DRAIN();
        }
#undef TILE_LD
#undef TILE_MF
    } else if constexpr (STG == 1) {
        // REGISTER-STAGED operand fetch (tile option 2): the wave's 4 chunks of a stage come in by plain 16 B global loads, three stages
        // ahead, into one of three register sets (set = stage % 3), and are written to their LDS slot in the LOAD segment one stage
        // before they are read.  An LDS-DMA piece costs the issuing wave 60-185 cycles beside ds_reads or between MFMAs (MI355X_MICROARCH
        // "LDS-DMA piece issue cost"); a global load costs its issue slot.  Price: 48 VGPRs and 4 ds_write_b128 per wave and stage.
        // Loads past the last stage are clamped to it (every segment issues the same loads: hipcc's vmcnt stays exact); their data
        // lands in a slot nobody reads again.
        u32x4 stg[3][4];
        u32x4* const ldsw = reinterpret_cast<u32x4*>(smem);
        auto fetch = [&](auto SET, int st) {
            constexpr int Q = decltype(SET)::value;
            const int sc = min(st, nst - 1);
#pragma unroll
            for (int c = 0; c < 4; ++c) stg[Q][c] = src[c][(size_t)sc * stride];
        };
        auto park = [&](auto SET, int st) {                  // stage st's chunks of this wave -> slot st & 3
            constexpr int Q = decltype(SET)::value;
            u32x4* const w = ldsw + (size_t)(st & (TS_SLOTS - 1)) * TS_STAGE_UNITS + lane;
#pragma unroll
            for (int c = 0; c < 4; ++c) w[dst[c] >> 4] = stg[Q][c];
        };
        auto frags = [&](int jj) {
            const u32x4* a = lds + (size_t)(jj & (TS_SLOTS - 1)) * TS_STAGE_UNITS + lane;
            const u32x4* w = a + 16 * 64;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                for (int m = 0; m < 4; ++m) af[ks][m] = a[(ks * 8 + (wr * 4 + m)) * 64];
#pragma unroll
                for (int n = 0; n < 2; ++n) wf[ks][n] = w[((wc * 2 + n) * 2 + ks) * 64];
            }
        };
        // LOAD(j), j % 3 == Q: fragments of stage j, then stage j + 1 (set (Q + 1) % 3) into its slot
        auto load2 = [&](auto SET, int jj) {
            constexpr int Q = decltype(SET)::value;
            frags(jj);
            park(std::integral_constant<int, (Q + 1) % 3>{}, jj + 1);
        };
        // MFMA(j), j % 3 == Q: 16 MFMAs from registers, the 4 loads of stage j + 3 (set Q, free since LOAD(j - 1)) between them
        auto mfma2 = [&](auto SET, int jj) {
            constexpr int Q = decltype(SET)::value;
            const int sc = min(jj + 3, nst - 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = mfma32(af[ks][m], wf[ks][n], acc[m][n]);
                    if ((m & 1) == 1) {
                        const int c = ks * 2 + (m >> 1);
                        stg[Q][c] = src[c][(size_t)sc * stride];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        fetch(I0{}, 0); fetch(I1{}, 1); fetch(I2{}, 2);
        park(I0{}, 0);
        // whole triples of stages without a branch inside (the outstanding-load state at the back edge equals the one at entry --
        // stages j + 1 and j + 2 in flight -- so the compiler's vmcnt before each park stays exact: the younger stage keeps flying);
        // the last 1-3 stages behind conditions
        if (isX) {
            int j = 0;
            for (; j + 2 < nst; j += 3) {
                __syncthreads(); load2(I0{}, j); __syncthreads(); mfma2(I0{}, j);
                __syncthreads(); load2(I1{}, j + 1); __syncthreads(); mfma2(I1{}, j + 1);
                __syncthreads(); load2(I2{}, j + 2); __syncthreads(); mfma2(I2{}, j + 2);
            }
            if (j < nst) { __syncthreads(); load2(I0{}, j); __syncthreads(); mfma2(I0{}, j); }
            if (j + 1 < nst) { __syncthreads(); load2(I1{}, j + 1); __syncthreads(); mfma2(I1{}, j + 1); }
            __syncthreads();                                    // Y has read its last fragments: LDS is free for the epilogue
            BD_MFMA_DRAIN();
        } else {
            __syncthreads();
            __syncthreads();
            load2(I0{}, 0);
            int j = 0;
            for (; j + 3 < nst; j += 3) {
                __syncthreads(); mfma2(I0{}, j);
                __syncthreads(); load2(I1{}, j + 1); __syncthreads(); mfma2(I1{}, j + 1);
                __syncthreads(); load2(I2{}, j + 2); __syncthreads(); mfma2(I2{}, j + 2);
                __syncthreads(); load2(I0{}, j + 3);
            }
            __syncthreads(); mfma2(I0{}, j);                    // 1-3 stages left, the fragments of stage j are loaded
            if (j + 1 < nst) {
                __syncthreads(); load2(I1{}, j + 1); __syncthreads(); mfma2(I1{}, j + 1);
                if (j + 2 < nst) { __syncthreads(); load2(I2{}, j + 2); __syncthreads(); mfma2(I2{}, j + 2); }
            }
            BD_MFMA_DRAIN();
        }
    } else {
        // ---- prologue: three stages in flight; stage 0 must have landed for everybody before the first LOAD
        issue(0);
        if (1 < nst) issue(1);
        if (2 < nst) issue(2);
        {
            const int younger = min(nst - 1, 2);
            if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // two loops, one per group: the same number of barriers on both sides (2 per stage)
        if (isX) {
            for (int j = 0; j < nst; ++j) {
                __syncthreads();                                    // opens half-step 2j
                load_seg(j);
                __syncthreads();                                    // opens half-step 2j + 1
                if (!(dbg & 1)) mfma_seg(j);
            }
            __syncthreads();                                        // Y has read its last fragments: LDS is free for the epilogue
            BD_MFMA_DRAIN();                                        // (bd_common.h: drain in the block that holds the last MFMAs)
        } else {
            __syncthreads();
            __syncthreads();
            load_seg(0);
            for (int j = 1; j < nst; ++j) {
                __syncthreads();                                    // opens half-step 2j
                if (!(dbg & 1)) mfma_seg(j - 1);
                __syncthreads();                                    // opens half-step 2j + 1
                load_seg(j);
            }
            __syncthreads();
            if (!(dbg & 1)) mfma_seg(nst - 1);                      // Y's last stage
            BD_MFMA_DRAIN();
        }

    }

    // ---- epilogue.  D layout of the 32x32 MFMA: lane -> column lane & 31, reg r -> row (r&3)+8(r>>2)+4(lane>>5).
    if constexpr (EPI == BD_EPI_BF16) {
        // bf16(+bias) row-major: straight from the accumulators every store instruction would write 2 bytes per lane (23 us of a
        // 140 us tile).  Each wave instead lays its 128 x 64 sub-tile down in its own 16 KiB of LDS (free since the barrier
        // above) and writes it out as 16 B per lane, 8 rows x 128 contiguous bytes per instruction.
        bf16_t* const mine = reinterpret_cast<bf16_t*>(smem) + (size_t)wave * (128 * 64);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int lc = n * 32 + (lane & 31);
            const float bias_col = p.bias ? bf2f(p.bias[(nt * 8 + wc * 2 + n) * 32 + (lane & 31)]) : 0.f;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    mine[(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 64 + lc] = f2bf(acc[m][n][r] + bias_col);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                      // lgkmcnt(0): the wave's own LDS writes (no other wave reads them)
        const u32x4* const rd = reinterpret_cast<const u32x4*>(mine);
        bf16_t* const o = p.act + (size_t)((mt * 8 + wr * 4) * 32) * p.N + (size_t)(nt * 8 + wc * 2) * 32;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int row = it * 8 + (lane >> 3), seg = lane & 7;
            *reinterpret_cast<u32x4*>(o + (size_t)row * p.N + seg * 8) = rd[row * 8 + seg];
        }
        return;
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int panel = nt * 8 + wc * 2 + n;
        const int col = panel * 32 + (lane & 31);
        const float bias_col = (EPI != BD_EPI_PARTIAL && p.bias) ? bf2f(p.bias[col]) : 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const f32x16& a = acc[m][n];
            const int row0 = (mt * 8 + wr * 4 + m) * 32;
            if (EPI == BD_EPI_PARTIAL) {
                float* o = p.out + ((size_t)s * p.Mpad + row0) * p.N + col;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[(size_t)((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * p.N] = a[r];
            } else {                                             // BD_EPI_SWIGLU: lanes (l & 16) == 0 hold gate f, the others the matching up
                const int f = panel * 16 + (lane & 15);
                bf16_t o8[8];
                swiglu_pairs(a, bias_col, lane, o8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = 2 * j + ((lane >> 4) & 1);
                    const int row = row0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                    p.act[afrag_off(row, f, p.RB)] = o8[j];
                }
            }
        }
    }
}

// RB % 8 == 0 (256-row tiles), N % 256 == 0, K % 32 == 0, panel-major weights; S > 1 only with fp32 slabs (BD_EPI_PARTIAL)
static int g_tile_dbg = 0;                                       // measurement only: 1 = no MFMA work, 2 = no DMA after the prologue
static int g_tile_wreg = 0;                                      // by-shape choice: 1 = the W-in-registers form ("tile" = 1 keeps the by-shape rule)
static int g_tile_stg = -1;                                      // operand fetch: 1 register-staged, 0 LDS-DMA, -1 by shape (option "tile" = 2 / 3 / 1)
void bdk_gemm_tile_debug(int v) { g_tile_dbg = v; }
void bdk_gemm_tile_stg(int v) { g_tile_stg = v; }
template <int EPI, int STG>
static int launch_tile(const GemmP& p, int blocks, hipStream_t st) {
    constexpr int lds = TS_SLOTS * TS_STAGE_UNITS * 16;            // 128 KiB: one workgroup per CU
    static unsigned long long optin = 0;
    if (!bd_lds_optin((const void*)gemm_tile_kernel<EPI, STG>, lds, &optin)) return -8;
    BD_LAUNCH((gemm_tile_kernel<EPI, STG>), dim3(blocks), dim3(512), lds, st, p, g_tile_dbg);
    return bd_launch_status();
}
int bdk_gemm_tile(const GemmP& p, int epi, hipStream_t st) {
    if (p.RB % 8 || p.N % 256 || p.K % 32 || p.S < 1 || (p.S > 1 && epi != BD_EPI_PARTIAL) || epi == BD_EPI_F32) return -2;
    const int nst_total = p.K / 32, q = (nst_total + p.S - 1) / p.S;
    if ((p.S - 1) * q >= nst_total) return -3;
    const int RT = p.RB / 8, NTS = (p.N / 256) * p.S;
    const int nR = ((RT + 7) / 8) * ((NTS + 3) / 4);
    const int blocks = ((nR + 7) / 8) * 8 * 32;
    // measured (profiles/r04_gemm_tile_regstaged.log): adaLN x8 665 vs 707 us, x16 1218 vs 1236, x52 4506 vs 4579; ImageNet w1 (K = 768) 94.0 vs 92.7
    // operand fetch: 2 = W straight into registers (round 6), 1 = both operands register-staged through LDS, 0 = LDS-DMA
    const int stg = g_tile_stg < 0 ? (g_tile_wreg ? 2 : (nst_total / p.S >= 64 ? 1 : 0)) : g_tile_stg;
    if (stg == 2) {
        if (epi == BD_EPI_PARTIAL) return launch_tile<BD_EPI_PARTIAL, 2>(p, blocks, st);
        if (epi == BD_EPI_BF16) return launch_tile<BD_EPI_BF16, 2>(p, blocks, st);
        return launch_tile<BD_EPI_SWIGLU, 2>(p, blocks, st);
    }
    if (stg == 1) {
        if (epi == BD_EPI_PARTIAL) return launch_tile<BD_EPI_PARTIAL, 1>(p, blocks, st);
        if (epi == BD_EPI_BF16) return launch_tile<BD_EPI_BF16, 1>(p, blocks, st);
        return launch_tile<BD_EPI_SWIGLU, 1>(p, blocks, st);
    }
    if (epi == BD_EPI_PARTIAL) return launch_tile<BD_EPI_PARTIAL, 0>(p, blocks, st);
    if (epi == BD_EPI_BF16) return launch_tile<BD_EPI_BF16, 0>(p, blocks, st);
    return launch_tile<BD_EPI_SWIGLU, 0>(p, blocks, st);
}
