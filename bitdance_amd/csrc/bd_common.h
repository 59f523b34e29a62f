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

// hipcc (ROCm 7.2, gfx950) under-counts the MFMA -> accumulator-read wait states across a TAKEN BRANCH: a guarded loop tail whose last
// executed phase ends in MFMAs jumps to a join that reads the last accumulator registers 6 instructions later (one `s_nop 0` inserted,
// >= 11 needed for an 8-pass MFMA).  Observed: acc[MB-1][15] stale -- rows 27 / 31 of the last row block wrong -- exactly when the K
// stages fill the whole tail (found by a ring-depth sweep in round 3: gemm_kernel<2,2,4,..,R=3> at 14 stages, <4,1,4,..,R=4> at 15 / 27).  The forms the
// engine launches never hit that stage count, but nothing guaranteed it.  The accumulator copies are emitted at the very top of the
// join block, ahead of anything the source can place there, so the drain (32 wait states: a 16-pass MFMA needs 19) goes at the END of
// the block that holds the last MFMAs: the final guarded phase of a K loop, the last statement of an MFMA loop body.
#define BD_MFMA_DRAIN() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 15\n\ts_nop 15" ::: "memory"); \
                             __builtin_amdgcn_sched_barrier(0); } while (0)

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

// Fused SwiGLU on a 32 x 32 accumulator whose columns are a packed (16 gate | 16 up) panel: lanes (l & 16) == 0 hold gate feature
// l & 15, lane l ^ 16 the matching up feature, 16 rows each (register r -> row (r&3) + 8 (r>>2) + 4 (l>>5)).
//     out = bf16( bf16(silu(bf16(h1 + b1))) * bf16(h2 + b2) )                                       flow_head_parallel_x.py:250-251
// One cross-lane exchange per PAIR of rows, no branch: the gate lane keeps row 2j and receives up[2j], the up lane takes row 2j + 1
// and receives gate[2j + 1] -- every lane evaluates 8 SiLUs instead of the gate lanes 16 behind a per-element exec mask (round 6: the
// per-element form was a 200-cycle serial chain per value on a one-wave-per-SIMD workgroup: 3 us of a 128-row launch's tail, 25 us of a
// 256-row one's).  out[j] belongs to register r = 2j + ((lane >> 4) & 1), feature lane & 15; the bits are those of the per-element form.
BD_DEV void swiglu_pairs(const f32x16& a, float bias_col, int lane, bf16_t (&out)[8]) {
    const bool up = (lane & 16) != 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v0 = bfr(a[2 * j] + bias_col), v1 = bfr(a[2 * j + 1] + bias_col);      // Linear outputs rounded to bf16
        const float got = __shfl_xor(up ? v0 : v1, 16);
        const float g = up ? got : v0, u = up ? v1 : got;
        out[j] = f2bf(silu_bf(g) * u);                                                    // silu -> bf16, product -> bf16
    }
}

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

// block-wide max over <=16 waves (non-negative values); `red` is >=16 floats of LDS.  All threads get the result.
BD_DEV float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// fp8-e4m3 activation operand of the fp8 matrix pipe ("A8", bd_gemm_kernel.h WT = 2): 2 KiB chunks per (64-deep stage, row block of
// 32), each two lane-linear 1 KiB halves; lane l = row (l & 31), k = stage * 64 + (l >> 5) * 32 + byte.  BYTE offset of (row, k):
BD_DEV size_t a8_off(int row, int k, int RB) {
    return (((((size_t)(k >> 6) * RB + (row >> 5)) * 2 + ((k >> 4) & 1)) * 64) + ((row & 31) + (((k >> 5) & 1) << 5))) * 16 + (k & 15);
}
// one row's per-row quantisation: amax over the row -> scale = amax / 448 (what the GEMM multiplies back), inv = 448 / amax;
// a thread's 8 consecutive k: q = e4m3(v * inv) (round to nearest even), 8 bytes in one store.  amax == 0: scale 0, bytes 0.
BD_DEV void quant8_store(unsigned char* a8, int row, int k0, int RB, const float* v, float inv) {
    int w0 = 0, w1 = 0;
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(__fmul_rn(v[0], inv), __fmul_rn(v[1], inv), w0, false);
    w0 = __builtin_amdgcn_cvt_pk_fp8_f32(__fmul_rn(v[2], inv), __fmul_rn(v[3], inv), w0, true);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(__fmul_rn(v[4], inv), __fmul_rn(v[5], inv), w1, false);
    w1 = __builtin_amdgcn_cvt_pk_fp8_f32(__fmul_rn(v[6], inv), __fmul_rn(v[7], inv), w1, true);
    *reinterpret_cast<uint2*>(a8 + a8_off(row, k0, RB)) = make_uint2((unsigned)w0, (unsigned)w1);
}
// the row's (scale, inv) from the block-wide amax of |v|; thread 0 publishes the scale
BD_DEV float row_quant_scale(const float* v, bool active, float* red, float* scale_out, int row) {
    float am = 0.f;
    if (active) {
#pragma unroll
        for (int j = 0; j < 8; ++j) am = fmaxf(am, fabsf(v[j]));
    }
    am = block_max(am, red);
    if (threadIdx.x == 0) scale_out[row] = __fdiv_rn(am, 448.0f);
    return am > 0.f ? __fdiv_rn(448.0f, am) : 0.f;
}

// Epochs of the flag hand-offs (bd_comm.hip, bd_sp.hip, the GEMM prologue wait): values only grow, MODULO 2^32 -- all epoch arithmetic is
// unsigned (a signed `flag - e < 0` is undefined behaviour at the wrap and clang folds it into `flag < e`, which passes stale flags after
// 2^31: ADVICE r05).  Sequence-parallel epochs = replay counter * 2^16 + the hand-off's sequence number inside the replayed graph
// (1 .. BD_SP_SEQ_MAX: 65535 hand-offs per sampling run = ~2700 sampling steps at 6 blocks; the replay counter wraps harmlessly).
#define BD_SP_SEQ_BITS 16
#define BD_SP_SEQ_MAX ((1 << BD_SP_SEQ_BITS) - 1)
BD_DEV int bd_sp_epoch_of(int rc, int seq) { return (int)(((unsigned)rc << BD_SP_SEQ_BITS) + (unsigned)seq); }
BD_DEV bool bd_epoch_before(int flag, int e) { return (int)((unsigned)flag - (unsigned)e) < 0; }     // flag has not reached e yet

// Launch anatomy (measurement builds only: -DBD_GEMM_STAMP, tools/launch_anatomy.py builds libbitdance_hip_stamp.so): thread 0 of every
// workgroup writes s_memrealtime (the chip-wide 100 MHz clock: comparable across CUs and kernels) into 8 words per workgroup of the
// launch's stamp region.  GEMM: 0 workgroup start, 1 first A stage in LDS, 2 first W stage landed, 3 K loop done, 4 K parts reduced in
// LDS, 5 slabs drained + ticket taken, 6 last store drained, 7 (slice << 8) | last-arriver bit.  Row kernels: 0 start, 6 end.
// The product build compiles none of it.
#ifdef BD_GEMM_STAMP
#define BD_STAMP_FIELD unsigned long long* stamp = nullptr;
BD_DEV unsigned long long bd_realtime() {          // inline asm: stays where it is written, between the waits around it
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
BD_DEV void bd_kstamp(unsigned long long* s, int k) {
    if (s && threadIdx.x == 0) s[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + k] = bd_realtime();
}
BD_DEV void bd_kstamp_val(unsigned long long* s, int k, unsigned long long v) {
    if (s && threadIdx.x == 0) s[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 8 + k] = v;
}
#define BD_KSTAMP(s, k) bd_kstamp(s, k)
#define BD_KSTAMP_END(s) do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); bd_kstamp(s, 6); } while (0)
unsigned long long* bdk_stamp_next(const char* name, int nwg);      // bd_api.hip: the next launch's region (null: stamping off)
void bdk_stamp_label(const char* name);                             // names the GEMM launches that follow
const char* bdk_stamp_current_label();
#define BD_STAMPED(ARGS_T, a, name, nwg) ARGS_T a##_st = a; a##_st.stamp = bdk_stamp_next(name, nwg); const ARGS_T& a##_l = a##_st
#else
#define BD_STAMP_FIELD
#define BD_KSTAMP(s, k) do {} while (0)
#define BD_KSTAMP_END(s) do {} while (0)
#define BD_STAMPED(ARGS_T, a, name, nwg) const ARGS_T& a##_l = a
#endif

// n one-lane system-scope flag accesses (one per destination rank) spread over the workgroup's WAVES: lanes 0..n-1 of ONE wave storing to n
// uncached words leave as n serialized fabric writes (~0.5 us each: ln_mod_sp's "flags raised" phase was 4.1 us at 8 ranks against 1.5 at
// 2, profiles/r06_launch_anatomy.log) -- lane 0 of wave q stores flag q instead (more destinations than waves: lanes 0, 1, ... of each).
// Returns the destination this thread handles, or -1.
BD_DEV int bd_spread_lane(int n) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
    const int q = wave + lane * nw;
    return (q < n && lane < (n + nw - 1) / nw) ? q : -1;
}

// Device-resident state of the autoregressive loop, read by every step-dependent kernel so that
// one captured hipGraph can be replayed for every AR step.
#define BD_MAX_SEQ 64  // per-sequence KV-length slots: num_images <= 32 with CFG on the Qwen3 path (imagenet sequences share slot 0)
struct BdStepState {
    int step;                  // AR step index (0-based)
    int kv_len[BD_MAX_SEQ];    // per sequence (branch-major: cond b0.., uncond b0..): tokens already in the KV cache
};
