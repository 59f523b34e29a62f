// fp8-e4m3 weight path of the weight-streaming GEMM (BASELINE config 5): every Linear whose weights are streamed from HBM
// stores them as OCP e4m3 with one fp32 scale per output channel; a wave converts its 16 B/lane (two k-steps) to bf16 in
// registers right before the bf16 MFMA and multiplies the scale into the fp32 accumulator after the K loop, so activations,
// accumulation and every rounding point of the bf16 path are unchanged and only the weight bytes halve.  A separate
// precision mode (never the bf16 headline): oracle policy "fp8w", tests/test_gpu_fp8.py.
//
// Packed layout: 1 KiB chunk per (32-row panel, PAIR of k-steps); lane l holds 8 weights of W[panel*32 + (l&31)][.] for
// k-step 2j (its low 8 B) and 2j+1 (high 8 B), each group = k offsets ks*16 + (l>>5)*8 + 0..7.  Chunks of a panel are
// contiguous in K: a 64-deep stage is 2 KiB per panel, two 16 B/lane loads.
#include "bd_gemm_kernel.h"

__global__ void pack_w8_kernel(u32x4* __restrict__ dst, const unsigned char* __restrict__ src, const unsigned char* __restrict__ src2,
                               int panels, int K, int nb0, int mode) {
    const int KP = K >> 5;                                   // k-step pairs
    const size_t total = (size_t)panels * KP * 64;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(u & 63);
        const size_t c = u >> 6;
        const int kp = (int)(c % KP);
        const int pn = (int)(c / KP);
        const int i = l & 31;
        const unsigned char* row;
        if (mode == 0) row = src + ((size_t)pn * 32 + i) * K;
        else row = (i < 16) ? src + ((size_t)pn * 16 + i) * K : src2 + ((size_t)pn * 16 + (i - 16)) * K;
        const uint2 lo = *reinterpret_cast<const uint2*>(row + (2 * kp) * 16 + (l >> 5) * 8);
        const uint2 hi = *reinterpret_cast<const uint2*>(row + (2 * kp + 1) * 16 + (l >> 5) * 8);
        dst[((size_t)(nb0 + pn) * KP + kp) * 64 + l] = (u32x4){lo.x, lo.y, hi.x, hi.y};
    }
}

int bdk_pack_w8(void* dst, const void* src, const void* src2, int panels, int K, int nb0, int panels_total, int mode, hipStream_t st) {
    if (K % 64 || nb0 + panels > panels_total) return -2;
    const size_t total = (size_t)panels * (K / 32) * 64;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    BD_LAUNCH(pack_w8_kernel, dim3(blocks), dim3(256), 0, st, (u32x4*)dst, (const unsigned char*)src, (const unsigned char*)src2,
              panels, K, nb0, mode);
    return bd_launch_status();
}

// ---- fp8 weights for the fp8 x fp8 matrix pipe (WT = 2, bd_gemm_kernel.h mfma32_f8): 2 KiB per (panel, 64-deep stage) as two
// lane-linear 1 KiB halves; lane l, half h = the 16 weights W[panel*32 + (l&31)][kk*64 + (l>>5)*32 + 16 h + 0..15]
__global__ void pack_w8k_kernel(u32x4* __restrict__ dst, const unsigned char* __restrict__ src, const unsigned char* __restrict__ src2,
                                int panels, int K, int nb0, int mode) {
    const int KS = K >> 6;                                   // 64-deep stages
    const size_t total = (size_t)panels * KS * 128;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (size_t)gridDim.x * blockDim.x) {
        const int l = (int)(u & 63), h = (int)((u >> 6) & 1);
        const size_t c = u >> 7;
        const int kk = (int)(c % KS);
        const int pn = (int)(c / KS);
        const int i = l & 31;
        const unsigned char* row;
        if (mode == 0) row = src + ((size_t)pn * 32 + i) * K;
        else row = (i < 16) ? src + ((size_t)pn * 16 + i) * K : src2 + ((size_t)pn * 16 + (i - 16)) * K;
        dst[(((size_t)(nb0 + pn) * KS + kk) * 2 + h) * 64 + l] = *reinterpret_cast<const u32x4*>(row + kk * 64 + (l >> 5) * 32 + 16 * h);
    }
}

int bdk_pack_w8k(void* dst, const void* src, const void* src2, int panels, int K, int nb0, int panels_total, int mode, hipStream_t st) {
    if (K % 64 || nb0 + panels > panels_total) return -2;
    const size_t total = (size_t)panels * (K / 64) * 128;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    BD_LAUNCH(pack_w8k_kernel, dim3(blocks), dim3(256), 0, st, (u32x4*)dst, (const unsigned char*)src, (const unsigned char*)src2,
              panels, K, nb0, mode);
    return bd_launch_status();
}

// fp8 weights AND fp8 activations (A: the A8 layout + `ascale` [rows] fp32): the launch forms of the 128-row kernel
int bdk_gemm8a(const void* A8, const float* ascale, int RB, const void* W8k, const float* wscale, int N, int K, int S, int nw_ring, int epi,
               float* out_partial, void* out_act, const void* bias, int* cnt, hipStream_t st) {
    // (armed push target / operand wait: taken before any early return, bdk_gemm.  The sequence-parallel hand-off carries bf16 rows only)
    { BdTpPush none_p; (void)bdk_gemm_claim_push(-1, RB, N, &none_p); }
    { BdHWait none; if (bdk_gemm_take_hwait(&none)) return -10; }
    const int nw = nw_ring & 15;
    const int kw = ((nw_ring >> 8) & 3) + 1;
    if (kw > 2 || nw % kw || !ascale || !wscale) return -7;
    const int np = nw / kw;
    if (K % (64 * kw) || N % (32 * np) || S < 1) return -2;
    const int nst_total = K / (64 * kw), q = (nst_total + S - 1) / S;
    if ((S - 1) * q >= nst_total) return -3;
    if (epi != BD_EPI_PARTIAL && S != 1 && (out_partial == nullptr || cnt == nullptr)) return -4;
    const size_t PS = (size_t)(K >> 6) * 128, SS = 128;
    GemmP p{(const u32x4*)A8, (const u32x4*)W8k, out_partial, (bf16_t*)out_act, (const bf16_t*)bias, cnt, wscale, RB, N, K, S, RB * 32, PS, SS};
    p.ascale = ascale;
    // 256-row passes (the adaLN projection of a group of evaluations, bd_api.hip head_ada_group): one pass over the weights per 256
    // rows, same MFMA and K order per row as the 128-row form -> bit-identical rows.  Only that call shape (one slice, bf16 epilogue):
    // everything else keeps the 128-row forms the parity tests cover
    const int MB = (RB % 8 == 0 && kw == 1 && np == 4 && S == 1 && epi == BD_EPI_BF16) ? 8 : ((RB % 4 == 0) ? 4 : RB);
    if (MB != 8 && MB != 4 && MB != 2 && MB != 1) return -5;
    // ring 4: a 64-deep fp8 stage is only 2 KiB per wave, and the fp8 MFMA leaves the loop latency-bound on bytes in flight (ring 2:
    // qkv 33.0 us = 2.4 TB/s of fp8 bytes, profiles/r03_bench_fp8a_v1.json).  No 9 / 10-wave tiles: the loop needs ~190 registers.
#define BD_CASE8A(NPV, KWV, MBV) if (np == NPV && kw == KWV && MB == MBV) return launch_gemm<NPV, KWV, MBV, 4, 0, 2>(p, epi, st);
    BD_CASE8A(4, 1, 8)                                     // (8 waves x 256 rows spills: 128 accumulator registers in a 256-register wave)
    BD_CASE8A(4, 1, 4) BD_CASE8A(8, 1, 4) BD_CASE8A(2, 1, 4) BD_CASE8A(4, 2, 4) BD_CASE8A(2, 2, 4)
    BD_CASE8A(4, 1, 2) BD_CASE8A(8, 1, 2) BD_CASE8A(2, 1, 2) BD_CASE8A(4, 2, 2) BD_CASE8A(2, 2, 2)
    BD_CASE8A(4, 1, 1) BD_CASE8A(8, 1, 1) BD_CASE8A(2, 1, 1) BD_CASE8A(4, 2, 1) BD_CASE8A(2, 2, 1)
#undef BD_CASE8A
    return -6;
}

int bdk_gemm8(const void* A, int RB, const void* W8, const float* wscale, int N, int K, int S, int nw_ring, int epi,
              float* out_partial, void* out_act, const void* bias, int* cnt, hipStream_t st) {
    // the armed push target / operand wait belong to THIS call: taken (and cleared) before any early return (bdk_gemm)
    BdTpPush pend_push; const bool have_push = bdk_gemm_claim_push(epi, RB, N, &pend_push);   // tensor parallelism: the fp32-partial epilogue pushes the peers' rows
    BdHWait pend_hw; const bool have_hw = bdk_gemm_take_hwait(&pend_hw);                      // sequence-parallel: the operand comes from the peers' row kernels
    const int nw = nw_ring & 15;
    const int kw = ((nw_ring >> 8) & 3) + 1;
    if (kw > 2 || nw % kw) return -7;
    const int np = nw / kw;
    if (K % (64 * kw) || N % (32 * np) || S < 1) return -2;
    const int nst_total = K / (64 * kw), q = (nst_total + S - 1) / S;
    if ((S - 1) * q >= nst_total) return -3;
    if (epi != BD_EPI_PARTIAL && S != 1 && (out_partial == nullptr || cnt == nullptr || (nw == 10 && kw == 1))) return -4;
    const size_t PS = (size_t)(K >> 5) * 64, SS = 128;        // 16 B units: panel stride, 64-deep stage stride
    GemmP p{(const u32x4*)A, (const u32x4*)W8, out_partial, (bf16_t*)out_act, (const bf16_t*)bias, cnt, wscale, RB, N, K, S, RB * 32, PS, SS};
    const int MB = (RB % 4 == 0) ? 4 : RB;                     // 128-row passes (row blocks beyond 4: grid.y)
    if (MB != 4 && MB != 2 && MB != 1) return -5;
    if (have_push) p.push = pend_push;
    if (have_hw) p.hw = pend_hw;
    // ring 2 (two 2 KiB stages per wave in flight).  Ring 4 was measured SLOWER (adaLN 125 vs 97 us, profiles/r02_bench_fp8_v2.json):
    // at 128 rows and half the bytes per weight the workgroup is bound by its LDS-read + MFMA work per stage (256 FLOP per weight
    // byte, at the ridge), not by bytes in flight.
#define BD_CASE8(NPV, KWV, MBV) if (np == NPV && kw == KWV && MB == MBV) return launch_gemm<NPV, KWV, MBV, 2, 0, 1>(p, epi, st);
    BD_CASE8(4, 1, 4) BD_CASE8(8, 1, 4) BD_CASE8(10, 1, 4) BD_CASE8(2, 1, 4) BD_CASE8(4, 2, 4) BD_CASE8(2, 2, 4)
    BD_CASE8(4, 1, 2) BD_CASE8(8, 1, 2) BD_CASE8(2, 1, 2) BD_CASE8(4, 2, 2) BD_CASE8(2, 2, 2)
    BD_CASE8(4, 1, 1) BD_CASE8(8, 1, 1) BD_CASE8(2, 1, 1) BD_CASE8(4, 2, 1) BD_CASE8(2, 2, 1)
#undef BD_CASE8
    return -6;
}
