// The weight-streaming GEMM kernel template and its launchers, shared by the bf16-weight (bd_gemm.hip) and the fp8-weight
// (bd_gemm8.hip) translation units.  Design notes: bd_gemm.hip.
#pragma once
#include <cstdlib>
#include <type_traits>
#include "bd_common.h"
#include "bd_kernels.h"
#include "bd_hwait.h"

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;

BD_DEV f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                   __builtin_bit_cast(mfma_bf16x8, b), c, 0, 0, 0);
}


// 8 fp8-e4m3 (OCP, gfx950) weights of one lane -> the 8 bf16 of an MFMA B operand.  Every e4m3 value is exactly
// representable in bf16, so the conversion is exact; the per-output-channel scale is applied to the accumulator.
typedef __attribute__((ext_vector_type(2))) __bf16 bd_bf16x2;
BD_DEV u32x4 fp8x8_to_bf16x8(unsigned lo, unsigned hi) {
    return (u32x4){__builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, false)),
                   __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(lo, 1.0f, true)),
                   __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, false)),
                   __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(hi, 1.0f, true))};
}

// fp8 x fp8 on the block-scaled matrix pipe (the only low-precision MFMA at 2x the bf16 rate, MI355X_MICROARCH): 32x32x64,
// A / B = 32 e4m3 bytes per lane (row / column lane & 31, k = (lane >> 5) * 32 + byte), all block scales = 2^0 (E8M0 127): the real
// scales are one fp32 per activation ROW and one per weight CHANNEL, applied to the fp32 accumulator after the K loop
typedef __attribute__((ext_vector_type(8))) int bd_i32x8;
BD_DEV f32x16 mfma32_f8(u32x4 a0, u32x4 a1, u32x4 b0, u32x4 b1, f32x16 c) {
    const bd_i32x8 a = {(int)a0[0], (int)a0[1], (int)a0[2], (int)a0[3], (int)a1[0], (int)a1[1], (int)a1[2], (int)a1[3]};
    const bd_i32x8 b = {(int)b0[0], (int)b0[1], (int)b0[2], (int)b0[3], (int)b1[0], (int)b1[1], (int)b1[2], (int)b1[3]};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

struct GemmP {
    const u32x4* A;      // fragment-major activations, RB row-blocks
    const u32x4* W;      // packed weights
    float* out;          // EPI_PARTIAL: [S][Mpad][N] fp32
    bf16_t* act;         // EPI_SWIGLU : fragment-major bf16 [Mpad][N/2];  EPI_BF16: row-major bf16 [Mpad][N]
    const bf16_t* bias;  // EPI_SWIGLU : [N] in PACKED row order (or null); EPI_BF16: [N] (or null)
    int* cnt;            // EPI_BF16 / EPI_SWIGLU with S > 1: one arrival counter per output tile (zero between launches)
    const float* wscale; // fp8 weights: per packed output row [N] fp32 dequantisation scale (null for bf16 weights)
    int RB, N, K, S, Mpad;
    size_t PS, SS;       // packed-W strides in 16 B units: panel stride, 64-deep-K-stage stride
    const float* ascale = nullptr;   // fp8 ACTIVATIONS (WT = 2): per row [Mpad] fp32 dequantisation scale
    int w_keep = 0;                  // 256-row kernel: 1 = default-policy weight loads (several row tiles read each slice: let L2 keep it)
    BdTpPush push;                   // BD_EPI_F32 under tensor parallelism: the epilogue pushes the peers' slices (size > 1)
    BdHWait hw;                      // sequence-parallel tensor parallelism: the A operand is pushed by the peers' row kernels -- poll its row flags
                                     // after the first weight stages are in flight, then invalidate and load it (flags == nullptr: no wait)
    int red_first = 1;               // two-slice in-launch reduction: 1 = ticket first, only the first arriver parks its slab (round 6); 0 = both park
    BD_STAMP_FIELD                   // measurement builds (-DBD_GEMM_STAMP): the launch's stamp region (bd_common.h)
};
int bdk_red_first();                 // process-wide A/B switch (bd_set_gemm_option "red.first")

// MFMA-bound form for >= 512 rows: both operands through LDS, 256 x 256 workgroup tiles (bd_gemm_tile.hip)
int bdk_gemm_tile(const GemmP& p, int epi, hipStream_t st);
void bdk_gemm_tile_stg(int v);
// 512-row form: 256 x 128 tiles at one K slice, split-K inside the workgroup, bf16 / SwiGLU epilogues (bd_gemm_half.hip)
int bdk_gemm_half(const GemmP& p, int epi, hipStream_t st);

// R = depth of the per-wave W register ring = number of K stages a wave keeps in flight.  The A stage is
// prefetched equally far ahead (R-1 register slots, then one ds_write into the double-buffered LDS tile): vmcnt retires
// in order, so an A load issued late would force every older W load to complete with it.
// hipcc's s_waitcnt placement is exact inside a straight-line body but drains the whole queue at the first use after
// a loop back-edge; U (8 or 12) phases per iteration make that one drain in U.
// NP x KW waves per workgroup: NP 32-column panels, each streamed by KW waves that split every (64*KW)-deep K stage between
// them (wave kg takes the kg-th 64-deep part) and add their accumulators through LDS at the end -- split-K INSIDE the
// workgroup.  KW = 2 halves the tile width at the same number of waves and bytes in flight per CU, so the N = 15360 shapes
// fill 240 CUs with no cross-workgroup split at all (no slabs, no tickets) and the N = 5120 shapes need 3 slices
// instead of 6 (half the slab traffic, in-launch reduction applies).
// PIPE: the LDS reads of stage j+1 are issued BEFORE the MFMAs of stage j (whose fragments were read one phase earlier), so
// the matrix pipe never waits on ds_read latency and a phase costs max(MFMA, LDS, HBM) instead of their sum; the A tile of
// stage j+2 is written over stage j's buffer in the same phase (its reads drained at the previous barrier).  One more A
// stage of lookahead, 64 more VGPRs for the second fragment set (4-wave workgroups: one wave per SIMD, 512 registers).
// WT = weight storage: 0 bf16 (1 KiB chunk per (panel, k-step of 16)), 1 fp8-e4m3 with per-output-channel scales (1 KiB
// chunk per (panel, PAIR of k-steps): a lane's 16 B = its 8 weights of k-step 2j, then of 2j+1; converted to bf16 in registers
// right before the MFMA, the scale multiplied into the accumulator after the K loop): half the bytes per weight.
// WT = 2: fp8 weights AND fp8 activations on the fp8 matrix pipe (mfma32_f8): a 64-deep stage is ONE MFMA per row block instead
// of four bf16 ones at twice the rate, and half the LDS bytes for the activations.  Weights: 2 KiB per (panel, 64-deep stage) as
// two lane-linear 1 KiB halves (bytes 0-15 / 16-31 of every lane's 32); activations: the same chunk shape per (stage, row block),
// produced by the row kernels together with one fp32 scale per row (bd_rows.hip quant8_store).
// The kernel's body as a device function (the __global__ wrapper follows it): tools/persist_pair.hip runs two bodies inside ONE launch
// with a grid barrier between them, to price a persistent chain against two launches on this code.
template <int NP, int KW, int MB, int EPI, int R, bool RED, int MODE = 0, int WT = 0>
BD_DEV void gemm_body(const GemmP& p) {
    constexpr int WL = (WT != 0) ? 2 : 4;                 // 16 B loads per lane and 64-deep stage
    constexpr bool PIPE = (MODE == 1);
    constexpr int NW = NP * KW, NT = NW * 64;
    constexpr int UNITS = MB * (WT == 2 ? 128 : 256) * KW; // 16 B units per (64*KW)-deep A stage
    constexpr int XL = (UNITS + NT - 1) / NT;             // A loads per thread per stage
    constexpr int XR = R - 1;                             // A register-ring slots
    constexpr int U = (R == 2) ? 8 : 12;
    static_assert(U % R == 0 && U % XR == 0 && U % 2 == 0, "static ring/buffer indices");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    u32x4* const lds = reinterpret_cast<u32x4*>(smem);    // two A-stage buffers of UNITS each

    BD_KSTAMP(p.stamp, 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pw = wave % NP, kg = wave / NP;              // panel / K-part of this wave
    const int S = p.S;
    const int s = blockIdx.x % S, nt = blockIdx.x / S, mt = blockIdx.y;
    const int nb = nt * NP + pw;                           // panel of this wave
    // ragged last tile (N/32 not a multiple of NP): the waves past the last panel stream the last panel again (L2 hits, no
    // branches in the loop) and store nothing.  Lets a GEMM pick the NP that puts ceil(panels / NP) just under 256 workgroups.
    const int npan = p.N >> 5;
    const bool pvalid = nb < npan;
    const int nbl = pvalid ? nb : npan - 1;
    const int nst_total = p.K / (64 * KW);
    const int q = (nst_total + S - 1) / S;
    const int st0 = s * q;
    const int nst = min(q, nst_total - st0);

    const u32x4* Wp = p.W + (size_t)nbl * p.PS + (size_t)(st0 * KW + kg) * p.SS + lane;
    const size_t w_stage = p.SS * KW;
    // A: unit u of a stage = chunk (ksl = (u>>6)/MB in [0, 4*KW), mb = (u>>6)%MB), lane u&63
    // (WT = 2: half-chunk c = u >> 6 of a stage = (K part c / (2 MB), row block (c >> 1) % MB, half c & 1); global order [stage][rb][half])
    size_t a_off[XL];
#pragma unroll
    for (int j = 0; j < XL; ++j) {
        const int u = tid + j * NT;
        const int c = u >> 6;
        if constexpr (WT == 2)
            a_off[j] = ((((size_t)(st0 * KW + c / (2 * MB)) * p.RB) + mt * MB + ((c >> 1) % MB)) * 2 + (c & 1)) * 64 + (u & 63);
        else
            a_off[j] = (((size_t)(st0 * 4 * KW + c / MB) * p.RB) + mt * MB + (c % MB)) * 64 + (u & 63);
    }
    const size_t a_stage = (WT == 2) ? (size_t)KW * p.RB * 128 : (size_t)4 * KW * p.RB * 64;

    u32x4 w[R][WL], xr[XR][XL];
    f32x16 acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    auto load_w = [&](u32x4(&wr)[WL], int i) {
#pragma unroll
        for (int j = 0; j < WL; ++j) wr[j] = __builtin_nontemporal_load(Wp + (size_t)i * w_stage + j * 64);
    };
    auto wop = [&](const u32x4(&wr)[WL], int kk) -> u32x4 {      // B operand of k-step kk of the stage
        if constexpr (WT == 1) return fp8x8_to_bf16x8(wr[kk >> 1][(kk & 1) * 2], wr[kk >> 1][(kk & 1) * 2 + 1]);
        else return wr[kk];
    };
    auto load_x = [&](u32x4(&x)[XL], int i) {
#pragma unroll
        for (int j = 0; j < XL; ++j)
            if (UNITS % NT == 0 || tid + j * NT < UNITS) x[j] = p.A[a_off[j] + (size_t)i * a_stage];
    };
    auto store_x = [&](u32x4* buf, const u32x4(&x)[XL]) {
#pragma unroll
        for (int j = 0; j < XL; ++j)
            if (UNITS % NT == 0 || tid + j * NT < UNITS) buf[tid + j * NT] = x[j];
    };
    // This wave's 64-deep part of the A stage goes LDS -> registers in one burst (16 ds_read_b128 for 128 rows), then the
    // MFMAs issue back to back: with the reads interleaved two-at-a-time the matrix pipe idled on LDS latency (the loop
    // was bound by the ds_read -> MFMA chain, not by HBM).
    constexpr int KG = (MB <= 4 && NW <= 8) ? 4 : 1;      // k-steps whose A fragments are resident at once (VGPR budget)
    auto compute = [&](const u32x4* stage, const u32x4(&wr)[WL]) {
        if constexpr (WT == 2) {                                          // one fp8 MFMA (K = 64) per row block
            const u32x4* buf8 = stage + kg * MB * 128;
            if constexpr (NW > 8 || MB > 4) {                            // 9 / 10 waves (168 registers per wave) or 256 rows (128 accumulator registers): one row block's fragments at a time
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const u32x4 x0 = buf8[(m * 2) * 64 + lane], x1 = buf8[(m * 2 + 1) * 64 + lane];
                    acc[m] = mfma32_f8(x0, x1, wr[0], wr[1], acc[m]);
                }
            } else {
                u32x4 x0[MB], x1[MB];
#pragma unroll
                for (int m = 0; m < MB; ++m) { x0[m] = buf8[(m * 2) * 64 + lane]; x1[m] = buf8[(m * 2 + 1) * 64 + lane]; }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int m = 0; m < MB; ++m) acc[m] = mfma32_f8(x0[m], x1[m], wr[0], wr[1], acc[m]);
            }
            return;
        }
        const u32x4* buf = stage + kg * MB * 256;
#pragma unroll
        for (int k0 = 0; k0 < 4; k0 += KG) {
            u32x4 xf[KG][MB];
#pragma unroll
            for (int kk = 0; kk < KG; ++kk)
#pragma unroll
                for (int m = 0; m < MB; ++m) xf[kk][m] = buf[((k0 + kk) * MB + m) * 64 + lane];
            if constexpr (KG > 1) __builtin_amdgcn_sched_barrier(0);   // keep the burst: hipcc otherwise re-interleaves 2 reads / 2 MFMAs
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) {
                const u32x4 b = wop(wr, k0 + kk);
#pragma unroll
                for (int m = 0; m < MB; ++m) acc[m] = mfma32(xf[kk][m], b, acc[m]);
            }
        }
    };

  if constexpr (PIPE) {
    static_assert(R == 2 && KG == 4 && WT != 2, "pipelined loop: two W stages in flight, whole-stage fragment sets (bf16 MFMA forms)");
    u32x4 xf[2][4][MB];
    auto read_stage = [&](u32x4(&x)[4][MB], const u32x4* stage) {
        const u32x4* buf = stage + kg * MB * 256;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int m = 0; m < MB; ++m) x[kk][m] = buf[(kk * MB + m) * 64 + lane];
    };
    auto mma_stage = [&](const u32x4(&x)[4][MB], const u32x4(&wr)[WL]) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const u32x4 b = wop(wr, kk);
#pragma unroll
            for (int m = 0; m < MB; ++m) acc[m] = mfma32(x[kk][m], b, acc[m]);
        }
    };
    // prologue: A stages 0 and 1 in LDS, stage 2 in registers; W stages 0 and 1 in registers; fragments of stage 0 read
    load_x(xr[0], 0);
    load_w(w[0], 0);
    store_x(lds, xr[0]);
    if (1 < nst) { load_x(xr[0], 1); load_w(w[1], 1); store_x(lds + UNITS, xr[0]); }
    if (2 < nst) load_x(xr[0], 2);
    __syncthreads();
    read_stage(xf[0], lds);
    int i = 0;
    for (; i + U + 2 < nst; i += U) {
#pragma unroll
        for (int ph = 0; ph < U; ++ph) {
            read_stage(xf[(ph + 1) & 1], lds + ((ph + 1) & 1) * UNITS);       // stage j+1 (stored during phase ph-1)
            __builtin_amdgcn_sched_barrier(0);
            mma_stage(xf[ph & 1], w[ph & 1]);                                 // stage j
            store_x(lds + (ph & 1) * UNITS, xr[0]);                           // stage j+2 over stage j's tile
            load_x(xr[0], i + ph + 3);
            load_w(w[ph & 1], i + ph + 2);
            __syncthreads();
        }
    }
#pragma unroll
    for (int ph = 0; ph < U + 2; ++ph) {
        const int j = i + ph;
        if (j < nst) {
            if (j + 1 < nst) read_stage(xf[(ph + 1) & 1], lds + ((ph + 1) & 1) * UNITS);
            __builtin_amdgcn_sched_barrier(0);
            mma_stage(xf[ph & 1], w[ph & 1]);
            if (j + 2 < nst) {
                store_x(lds + (ph & 1) * UNITS, xr[0]);
                if (j + 3 < nst) load_x(xr[0], j + 3);
                load_w(w[ph & 1], j + 2);
            }
            if (j + 1 < nst) __syncthreads();
            else BD_MFMA_DRAIN();                          // the last executed phase branches straight to the accumulator reads (bd_common.h)
        }
    }
  } else {
    // prologue: stages 0..R-1 of A and W in flight; stage q of A lives in ring slot q % XR
    if (p.hw.flags) {                                          // block-uniform: the operand comes from the peers (bd_sp.hip)
        load_w(w[0], 0);
#pragma unroll
        for (int r = 1; r < R; ++r)
            if (r < nst) load_w(w[r], r);                      // the weight stream starts before the wait ...
        gemm_hwait(p.hw, tid, NT);                             // ... which ends when every operand row has landed
        BD_KSTAMP(p.stamp, 1);
        load_x(xr[0], 0);
        store_x(lds, xr[0]);
#pragma unroll
        for (int r = 1; r < R; ++r)
            if (r < nst) load_x(xr[r % XR], r);
    } else {
        load_x(xr[0], 0);
        load_w(w[0], 0);
        store_x(lds, xr[0]);
#ifdef BD_GEMM_STAMP
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WL) : "memory");            // the A stage has landed (W stage 0 still in flight)
        BD_KSTAMP(p.stamp, 1);
#endif
#pragma unroll
        for (int r = 1; r < R; ++r)
            if (r < nst) { load_x(xr[r % XR], r); load_w(w[r], r); }
    }
    __syncthreads();
#ifdef BD_GEMM_STAMP
    if (nst >= R) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((R - 1) * (XL + WL)) : "memory");   // W stage 0 has landed, the younger stages fly on
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BD_KSTAMP(p.stamp, 2);
#endif

    int i = 0;
    // steady state: stage j = i + ph; every ring / buffer index below is a compile-time constant
    for (; i + U + R - 1 < nst; i += U) {
#pragma unroll
        for (int ph = 0; ph < U; ++ph) {
            compute(lds + (ph & 1) * UNITS, w[ph % R]);
            store_x(lds + ((ph + 1) & 1) * UNITS, xr[(ph + 1) % XR]);
            load_x(xr[(ph + 1) % XR], i + ph + R);          // same slot: (ph + R) % XR == (ph + 1) % XR
            load_w(w[ph % R], i + ph + R);
            __syncthreads();
        }
    }
    // tail: at most U + R - 1 stages, guarded (block-uniform conditions)
#pragma unroll
    for (int ph = 0; ph < U + R - 1; ++ph) {
        const int j = i + ph;
        if (j < nst) {
            compute(lds + (ph & 1) * UNITS, w[ph % R]);
            if (j + 1 < nst) {
                store_x(lds + ((ph + 1) & 1) * UNITS, xr[(ph + 1) % XR]);
                if (j + R < nst) { load_x(xr[(ph + 1) % XR], j + R); load_w(w[ph % R], j + R); }
                __syncthreads();
            } else {
                BD_MFMA_DRAIN();                           // the last executed phase branches straight to the accumulator reads (bd_common.h)
            }
        }
    }
  }

    BD_KSTAMP(p.stamp, 3);
    // ---- K parts of one panel meet in LDS: parts 1..KW-1 park their accumulators, part 0 adds them in order
    if constexpr (KW > 1) {
        __syncthreads();                                                  // every wave is done with the A tiles
        f32x4* const red = reinterpret_cast<f32x4*>(smem);                // [(kg-1)*NP + pw][m][r4][lane]: lane-linear 16 B
        if (kg > 0) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    red[((((kg - 1) * NP + pw) * MB + m) * 4 + r4) * 64 + lane] =
                        (f32x4){acc[m][4 * r4], acc[m][4 * r4 + 1], acc[m][4 * r4 + 2], acc[m][4 * r4 + 3]};
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int g = 1; g < KW; ++g)
#pragma unroll
                for (int m = 0; m < MB; ++m)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        const f32x4 v = red[((((g - 1) * NP + pw) * MB + m) * 4 + r4) * 64 + lane];
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[m][4 * r4 + j] += v[j];
                    }
        }
    }
    BD_KSTAMP(p.stamp, 4);
    const bool owner = (kg == 0) && pvalid;                                // the wave that holds the tile's sums
    if constexpr (WT != 0) {                                               // dequantisation scale of this lane's output column
        const float sc = p.wscale[nbl * 32 + (lane & 31)];
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] *= sc;
    }
    if constexpr (WT == 2) {                                               // ... and of the activation rows (reg r -> row (r&3)+8(r>>2)+4(lane>>5))
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const float* as = p.ascale + (mt * MB + m) * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] *= as[(r & 3) + 8 * (r >> 2)];
        }
    }

    // ---- epilogue.  D layout of the 32x32 MFMA: lane -> column (lane&31), reg r -> row (r&3)+8(r>>2)+4(lane>>5)
    const int col = nbl * 32 + (lane & 31);
    const float bias_col = ((EPI == BD_EPI_BF16 || EPI == BD_EPI_SWIGLU) && p.bias) ? bf2f(p.bias[col]) : 0.f;
    auto finalize = [&](int m) {
        const f32x16& a = acc[m];
        if (EPI == BD_EPI_PARTIAL) {
            float* o = p.out + ((size_t)s * p.Mpad + (size_t)(mt * MB + m) * 32) * p.N + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * p.N] = a[r];
        } else if (EPI == BD_EPI_F32) {    // the finished fp32 sum (no bias, no rounding): one rank's partial of a row-split Linear
            if (p.push.size > 1) {
                // the accumulator holds a COLUMN per lane; a store over the fabric should be 16 B of ONE row: turn the 32 x 32 block
                // through this wave's LDS patch ([32 rows][36 floats]: conflict-free writes, 16 B aligned reads), then lane L holds
                // columns 4 (L & 7) .. + 3 of rows (L >> 3) + 8 j -- eight consecutive rows per store, all reduced by ONE rank
                // (rows_per_rank % 8 == 0, checked by the launcher)
                float* const tb = reinterpret_cast<float*>(smem + 256) + pw * (32 * 36);
#pragma unroll
                for (int r = 0; r < 16; ++r) tb[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = a[r];
                const int c4 = nbl * 32 + (lane & 7) * 4;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int rr = (lane >> 3) + 8 * j;
                    const f32x4 v = *reinterpret_cast<const f32x4*>(tb + rr * 36 + (lane & 7) * 4);
                    const int row = (mt * MB + m) * 32 + rr;
                    const int g8 = (mt * MB + m) * 4 + j;                  // the 8-row group of this store (rr >> 3 == j)
                    // owner of the group: contiguous slices, or (sequence parallel, push.il) groups dealt round-robin to the ranks
                    const int q = __builtin_amdgcn_readfirstlane(p.push.il ? g8 % p.push.size : (g8 * 8) / p.push.rows_per_rank);
                    const int lrow = p.push.il ? (g8 / p.push.size) * 8 + (rr & 7) : row - q * p.push.rows_per_rank;
                    if (q == p.push.rank || q >= p.push.size) {            // mine (or a pad row past the last rank's rows): local
                        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.act) + (size_t)row * p.N + c4) = v;
                    } else {
                        const __amdgpu_buffer_rsrc_t dst = __builtin_amdgcn_make_buffer_rsrc(p.push.stage[q], 0, (int)(p.push.Us * p.push.size * 32), 0x00020000);
                        const unsigned off = (unsigned)((p.push.rank * p.push.Us + (size_t)lrow * (p.N >> 3)) * 32 + (size_t)c4 * 4);
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), dst, off, 0, 17 /* sc0 sc1: system scope */);
                    }
                }
                return;
            }
            float* o = reinterpret_cast<float*>(p.act) + (size_t)(mt * MB + m) * 32 * p.N + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * p.N] = a[r];
        } else if (EPI == BD_EPI_BF16) {   // Linear output rounded once to bf16 (what autocast's F.linear returns)
            bf16_t* o = p.act + (size_t)(mt * MB + m) * 32 * p.N + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * p.N] = f2bf(a[r] + bias_col);
        } else {  // BD_EPI_SWIGLU: lanes (l&16)==0 hold gate feature f, lanes (l&16)!=0 the matching up feature
            const int f = nbl * 16 + (lane & 15);
            bf16_t o8[8];
            swiglu_pairs(a, bias_col, lane, o8);                              // (bd_common.h: one exchange per pair of rows, no branch)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = 2 * j + ((lane >> 4) & 1);
                const int row = (mt * MB + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                p.act[afrag_off(row, f, p.RB)] = o8[j];
            }
        }
    };
    if constexpr (RED) {
        // In-launch split-K reduction ("last arriver reduces"): every K-slice parks its fp32 slab and takes a ticket on
        // the tile's counter; the slice that draws S-1 re-reads ALL slabs in slice order (a fixed summation order: the
        // result does not depend on which slice happened to arrive last) and runs the real epilogue, so the consumers
        // read ONE finished tensor instead of S fp32 slabs.
        // The XCD L2s are not coherent with each other, so the slab traffic is agent-scope relaxed atomics: `sc1`
        // write-through stores, drained by vmcnt(0) before the ticket, and `sc1` loads on the reducing side
        // (MI355X_MICROARCH.md "publish-large": 3.0 vs 8.2 us for plain stores + release fence; the fence form also
        // writes back / invalidates the whole L2 under the other workgroups' A-operand reuse).  This relies on gfx950's
        // sc1 semantics (write-through to the memory side, L2-bypassing loads) rather than on a release/acquire edge of the
        // memory model: the static_assert below keeps it from being compiled for any other target, and
        // tests/test_gpu_parity.py::test_gemm_in_launch_splitk_reduction checks determinism and values on the hardware.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
        static_assert(!RED, "the relaxed sc1 slab hand-off is validated for gfx950 only");
#endif
        // The slabs of this path are scratch between the slices of ONE tile, read back by the same lane mapping, so they are
        // kept in ACCUMULATOR order: region (tile, panel wave, slice) = [m][r4][lane] x 16 B, a lane's four consecutive rows of
        // its column in one dwordx4.  16 stores / loads of 16 B per lane and row block instead of 64 scalar ones: a 16 B `sc1`
        // store costs what a plain one does, a dword `sc1` store about six times as much per byte (MI355X_MICROARCH, stores).
        const __amdgpu_buffer_rsrc_t sl = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)((size_t)S * p.Mpad * p.N * 4), 0x00020000);
        const int ntl = (npan + NP - 1) / NP;
        const size_t region0 = ((size_t)(mt * ntl + nt) * NP + pw) * S;          // + slice
        auto slab_off = [&](int s_, int m, int r4) -> unsigned {
            return (unsigned)((((region0 + s_) * MB + m) * 4 + r4) * 1024 + lane * 16);
        };
        constexpr int SC1 = 16;                                                  // buffer cache-policy bit: sc1 (agent scope)
        if (S == 2 && p.red_first) {
            // TWO slices, ticket first (round 6): only the slice that arrives FIRST parks its accumulators (and marks the tile's counter
            // +2 once its stores have drained); the second keeps its own in registers, waits for that mark and adds the parked slab --
            // own + other == other + own bit for bit, so the result does not depend on the arrival order.  Before, both slices parked and
            // drained ahead of the ticket (profiles/r06_launch_anatomy.log: "slabs drained + ticket" 1.7 us on every workgroup of qkv / w1,
            // then 2-5 us for the last arriver): half the slab traffic, and the finishing workgroup skips its own store drain.  The first
            // arriver is running by construction when the second waits for it; the wait is bounded all the same.
            int* const flag = reinterpret_cast<int*>(smem);
            int* const ticket = p.cnt + (mt * ntl + nt);
            __syncthreads();                                              // all waves are done with the LDS tiles / the K-part sums
            if (tid == 0) flag[0] = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            const bool first = flag[0] == 0;
#ifdef BD_GEMM_STAMP
            BD_KSTAMP(p.stamp, 5);
            bd_kstamp_val(p.stamp, 7, ((unsigned long long)s << 8) | (first ? 0u : 1u));
#endif
            if (first) {
                if (owner) {
#pragma unroll
                    for (int m = 0; m < MB; ++m)
#pragma unroll
                        for (int r4 = 0; r4 < 4; ++r4)
                            __builtin_amdgcn_raw_buffer_store_b128(
                                (u32x4){__float_as_uint(acc[m][4 * r4]), __float_as_uint(acc[m][4 * r4 + 1]), __float_as_uint(acc[m][4 * r4 + 2]),
                                        __float_as_uint(acc[m][4 * r4 + 3])}, sl, slab_off(s, m, r4), 0, SC1);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave drains its write-throughs
                __syncthreads();
                if (tid == 0) __hip_atomic_fetch_add(ticket, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                BD_KSTAMP(p.stamp, 6);
                return;
            }
            if (tid == 0) {
                const long long t0 = wall_clock64();
                while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 3 && wall_clock64() - t0 < 200000000LL) __builtin_amdgcn_s_sleep(1);
            }
            __syncthreads();
            if (!owner) return;
            u32x4 v[MB][4];
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) v[m][r4] = __builtin_amdgcn_raw_buffer_load_b128(sl, slab_off(1 - s, m, r4), 0, SC1);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[m][4 * r4 + j] += __uint_as_float(v[m][r4][j]);
#pragma unroll
            for (int m = 0; m < MB; ++m) finalize(m);
            if (EPI == BD_EPI_F32 && p.push.size > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
            BD_KSTAMP_END(p.stamp);
            return;
        }
        if (owner) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
                    __builtin_amdgcn_raw_buffer_store_b128(
                        (u32x4){__float_as_uint(acc[m][4 * r4]), __float_as_uint(acc[m][4 * r4 + 1]), __float_as_uint(acc[m][4 * r4 + 2]),
                                __float_as_uint(acc[m][4 * r4 + 3])}, sl, slab_off(s, m, r4), 0, SC1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // every storing wave drains its write-throughs
        __syncthreads();                                                  // (also: all waves are done with the LDS tiles)
        int* const flag = reinterpret_cast<int*>(smem);
        int* const ticket = p.cnt + (mt * ntl + nt);
        if (tid == 0) flag[0] = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
#ifdef BD_GEMM_STAMP
        BD_KSTAMP(p.stamp, 5);
        bd_kstamp_val(p.stamp, 7, ((unsigned long long)s << 8) | (flag[0] == S - 1 ? 1u : 0u));
        if (flag[0] != S - 1) BD_KSTAMP(p.stamp, 6);
#endif
        if (flag[0] != S - 1 || !owner) return;                           // not the last slice of this tile / nothing to store
        if (S == 2) {                                                     // ("red.first" = 0: both slices parked; the last arriver adds the other slab)
            u32x4 v[MB][4];
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) v[m][r4] = __builtin_amdgcn_raw_buffer_load_b128(sl, slab_off(1 - s, m, r4), 0, SC1);
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[m][4 * r4 + j] += __uint_as_float(v[m][r4][j]);
#pragma unroll
            for (int m = 0; m < MB; ++m) finalize(m);
            if (EPI == BD_EPI_F32 && p.push.size > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (tid == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            BD_KSTAMP_END(p.stamp);
            return;
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
            for (int s2 = 0; s2 < S; ++s2) {                               // slice order: the sum does not depend on who arrived last
                u32x4 v[4];
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) v[r4] = __builtin_amdgcn_raw_buffer_load_b128(sl, slab_off(s2, m, r4), 0, SC1);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[m][4 * r4 + j] += __uint_as_float(v[r4][j]);
            }
            finalize(m);                                                  // row-block by row-block: short live ranges
            __builtin_amdgcn_sched_barrier(0);
        }
        if (EPI == BD_EPI_F32 && p.push.size > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-arm for the next launch
        BD_KSTAMP_END(p.stamp);
        return;
    }
    if (EPI == BD_EPI_F32 && p.push.size > 1) __syncthreads();           // every wave is done with the A tiles (the patches overlay them)
    if (owner) {
#pragma unroll
        for (int m = 0; m < MB; ++m) finalize(m);
    }
    if (EPI == BD_EPI_F32 && p.push.size > 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the pushes are at their destinations before the kernel ends
#ifdef BD_GEMM_STAMP
    bd_kstamp_val(p.stamp, 5, 0);
    bd_kstamp_val(p.stamp, 7, (unsigned long long)s << 8);
    BD_KSTAMP_END(p.stamp);
#endif
}

template <int NP, int KW, int MB, int EPI, int R, bool RED, int MODE = 0, int WT = 0>
__global__ __launch_bounds__(NP * KW * 64) void gemm_kernel(GemmP p) {
    gemm_body<NP, KW, MB, EPI, R, RED, MODE, WT>(p);
    if constexpr (EPI == BD_EPI_F32) {
        // sequence-parallel tensor parallelism: every wave has drained its pushes (vmcnt(0) at the end of the body); the workgroup that
        // arrives LAST in the whole launch writes the hand-off's epoch into every owner's flag word -- the owners' row kernels start
        // behind a flag that is already on its way instead of behind a signal their own first block would send
        if (p.push.done_cnt) {                                  // block-uniform
            __syncthreads();
            if (threadIdx.x == 0) {
                const int total = (int)(gridDim.x * gridDim.y);
                if (__hip_atomic_fetch_add(p.push.done_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == total - 1) {
                    __hip_atomic_store(p.push.done_cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // re-arm for the next launch
                    const int e = bd_sp_epoch_of(__hip_atomic_load(p.push.rc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), p.push.seq);
                    for (int q = 0; q < p.push.size; ++q)
                        if (q != p.push.rank) __hip_atomic_store(p.push.sig[q], e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
}

template <int NP, int KW, int MB, int EPI, int R, bool RED, int MODE = 0, int WT = 0>
static int launch_one(const GemmP& p, hipStream_t st) {
    const int ntiles = (p.N / 32 + NP - 1) / NP;          // the last tile may be ragged
    dim3 grid(ntiles * p.S, p.RB / MB);
    if constexpr (RED) {
        // the in-launch reduction indexes its slab regions by (row tile, N tile, panel wave): with a ragged last tile that runs
        // past the S * Mpad * N * 4 scratch (out-of-range buffer accesses are dropped silently -> wrong sums), and the tickets
        // live in a fixed block of 16384 counters
        if ((p.N / 32) % NP != 0) return -9;
        if ((long long)(p.RB / MB) * ntiles > 16383) return -9;      // (the last word is the tensor-parallel arrival counter, bd_api.hip)
    }
    // two A-stage buffers; with KW > 1 the same LDS is re-used for the accumulators of K parts 1..KW-1
    constexpr size_t lds_a = (size_t)2 * MB * 256 * KW * 16, lds_r = (size_t)(KW - 1) * NP * MB * 4096;
    constexpr size_t lds_e = (EPI == BD_EPI_F32) ? 256 + (size_t)NP * 32 * 36 * 4 : 0;   // the pushing epilogue's transposition patches, one per panel wave
    constexpr size_t lds_ar = lds_a > lds_r ? lds_a : lds_r;
    constexpr size_t lds = lds_ar > lds_e ? lds_ar : lds_e;
    if constexpr (lds > 64 * 1024) {                               // beyond 64 KiB of dynamic LDS needs the opt-in
        static unsigned long long optin = 0;                      // per device (bd_kernels.h)
        if (!bd_lds_optin((const void*)gemm_kernel<NP, KW, MB, EPI, R, RED, MODE, WT>, (int)lds, &optin)) return -8;
    }
#ifdef BD_GEMM_STAMP
    GemmP q = p;
    q.red_first = bdk_red_first();
    q.stamp = bdk_stamp_next(bdk_stamp_current_label(), (int)(grid.x * grid.y));
    BD_LAUNCH((gemm_kernel<NP, KW, MB, EPI, R, RED, MODE, WT>), grid, dim3(NP * KW * 64), lds, st, q);
#else
    GemmP q = p;
    q.red_first = bdk_red_first();
    BD_LAUNCH((gemm_kernel<NP, KW, MB, EPI, R, RED, MODE, WT>), grid, dim3(NP * KW * 64), lds, st, q);
#endif
    return bd_launch_status();
}

template <int NP, int KW, int MB, int R, bool RED, int MODE = 0, int WT = 0>
static int launch_gemm_r(const GemmP& p, int epi, hipStream_t st) {
    if (epi == BD_EPI_PARTIAL) return launch_one<NP, KW, MB, BD_EPI_PARTIAL, R, false, MODE, WT>(p, st);
    if (epi == BD_EPI_BF16) return launch_one<NP, KW, MB, BD_EPI_BF16, R, RED, MODE, WT>(p, st);
    if (epi == BD_EPI_F32) return launch_one<NP, KW, MB, BD_EPI_F32, R, RED, MODE, WT>(p, st);
    return launch_one<NP, KW, MB, BD_EPI_SWIGLU, R, RED, MODE, WT>(p, st);
}

// A: fragment-major bf16, RB row-blocks (RB must be 1, 2 or a multiple of 4).  W: packed.  N % (32*NP) == 0, K % (64*KW) == 0.
template <int NP, int KW, int MB, int R, int MODE = 0, int WT = 0>
static int launch_gemm(const GemmP& p, int epi, hipStream_t st) {
    if constexpr (NP * KW >= 9 && KW == 1) return launch_gemm_r<NP, KW, MB, R, false, 0, WT>(p, epi, st);   // single-slice tiles only
    else return (p.S > 1 && epi != BD_EPI_PARTIAL) ? launch_gemm_r<NP, KW, MB, R, true, MODE, WT>(p, epi, st)
                                                   : launch_gemm_r<NP, KW, MB, R, false, MODE, WT>(p, epi, st);
}

