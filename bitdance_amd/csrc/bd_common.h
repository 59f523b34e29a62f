// Shared device helpers for the BitDance gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;                                     // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) short bf16x8;          // one MFMA A/B operand (4 VGPRs)
typedef __attribute__((ext_vector_type(16))) float f32x16;         // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define BD_DEV __device__ __forceinline__

BD_DEV float bf2f(bf16_t x) { return __uint_as_float(((unsigned)x) << 16); }

// fp32 -> bf16 round-to-nearest-even (what torch's .to(bfloat16) does): gfx950 has it in hardware (v_cvt_pk_bf16_f32).
BD_DEV bf16_t f2bf(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
BD_DEV float bfr(float f) { return bf2f(f2bf(f)); }                 // round through bf16
BD_DEV unsigned pack2(float lo, float hi) { return (unsigned)f2bf(lo) | ((unsigned)f2bf(hi) << 16); }

// individually rounded fp32 ops (no FMA contraction) where the reference runs separate torch kernels
BD_DEV float fmul(float a, float b) { return __fmul_rn(a, b); }
BD_DEV float fadd(float a, float b) { return __fadd_rn(a, b); }
BD_DEV float fsub(float a, float b) { return __fsub_rn(a, b); }
BD_DEV float fdiv(float a, float b) { return __fdiv_rn(a, b); }

// torch silu opmath x / (1 + exp(-x)) in fp32; every caller rounds the result to bf16, so the hardware exp2 / rcp
// (1 ulp each) are indistinguishable from libm's expf and an IEEE divide after that rounding, at a fifth of the VALU work
BD_DEV float silu_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x)); }
BD_DEV float silu_bf(float x_bf) { return bfr(silu_f(x_bf)); }       // bf16 tensor in -> bf16 out

// MFMA-operand ("fragment-major") activation layout.
// A [rows][K] bf16 matrix is stored as 1 KiB chunks, one per (k-step of 16, row-block of 32):
//   chunk(ks, rb) holds, for lane l (0..63), the 8 bf16  A[rb*32 + (l&31)][ks*16 + (l>>5)*8 + 0..7]
// chunks ordered [ks][rb] (rb fastest), RB = padded_rows/32.  A wave reads one chunk with one
// 16 B/lane load and feeds it to v_mfma_f32_32x32x16_bf16 as either operand.
BD_DEV size_t afrag_off(int row, int k, int RB) {                    // element (bf16) offset
    return ((((size_t)(k >> 4) * RB + (row >> 5)) * 64) + ((row & 31) + (((k >> 3) & 1) << 5))) * 8 + (k & 7);
}

BD_DEV float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
BD_DEV float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// block-wide sum over <=16 waves; `red` is >=16 floats of LDS. All threads get the result.
BD_DEV float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// Body of a prefetch workgroup (PfDesc, bd_kernels.h): j = its index among the `nblk` extra workgroups of the launch (extra
// workgroup 0 has a blockIdx that is a multiple of 8, so that j % 8 is this workgroup's XCD).  The loads are LDS-DMA
// (global_load_lds_dwordx4: 1 KiB per wave instruction, default cache policy so the lines allocate in L2): they have NO
// register destination -- a register load whose result is never read would leave the compiler free to reuse the destination
// registers while the data is still in flight -- and every wave drops its kilobyte onto the same scratch in LDS.
template <class PF>
BD_DEV void bd_prefetch_run(const PF& d, int j, int nthreads) {
    __shared__ __attribute__((aligned(16))) unsigned pf_sink[256];
    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    const int per = d.nblk >> 3;                                   // prefetch workgroups per XCD
    const int upp = d.bytes >> 4;                                  // 16 B units per stream
    const int units = d.NP * upp;
    const char* const W = reinterpret_cast<const char*>(d.W);
    for (int b = (j & 7) + 8 * (j >> 3); b < d.nwg; b += 8 * per) {
        const int s = b % d.S, nt = b / d.S;
        // whole waves only (LDS-DMA writes lane-linear from an M0 base): units is a multiple of 64 (bytes % 1024 == 0)
        for (int u = threadIdx.x; u < units; u += nthreads) {
            const int pn = u / upp, o = u - pn * upp;
            const int panel = min(nt * d.NP + pn, d.npan - 1);     // ragged last tile: re-touch the last panel
            const char* src = W + (size_t)panel * d.panel_bytes + (size_t)s * d.slice_bytes + (size_t)o * 16;
            __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)pf_sink, 16, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Device-resident state of the autoregressive loop, read by every step-dependent kernel so that
// one captured hipGraph can be replayed for every AR step.
struct BdStepState {
    int step;          // AR step index (0-based)
    int kv_len[16];    // per sequence (branch-major: cond b0.., uncond b0..): tokens already in the KV cache
};
