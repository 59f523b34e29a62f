// Attention kernels of the BitDance hot path (gfx950, wave64, v_mfma_f32_32x32x16_bf16).
//
//  * head_attn   : the DiT block's non-causal attention over one 64-token patch (flow_head:192-220, the
//                  flash_attn_func branch).  One workgroup per (sequence, head); K and V^T of the patch are
//                  staged once in LDS, each wave owns 32 query rows.
//  * llm_attn    : block-bidirectional decode attention (t2i_pipeline.py:256-268: an all-True mask, i.e. every
//                  one of the P new queries sees all past+P keys) over a static, pre-allocated KV cache --
//                  K as [seq][kvh][pos][128], V TRANSPOSED as [seq][kvh][128][pos] so that both MFMA operands are
//                  16-byte contiguous reads.  GQA-aware: one workgroup = one KV head x all G query heads x 64
//                  queries (G*2 waves share every LDS-staged K/V tile), split over the key range (flash-decode),
//                  merged by llm_attn_combine.
//
// Both use the "swapped" score product S^T = K Q^T so that a lane owns one query row's scores in registers
// (row max / sum are register reductions plus one lane^32 exchange), softmax statistics in fp32, the
// un-normalised P rounded to bf16 for P.V, fp32 output accumulation, one normalisation + rounding at the end
// (the FlashAttention-2 schedule the reference's kernels follow).
#include "bd_common.h"
#include "bd_kernels.h"

typedef __attribute__((ext_vector_type(8))) __bf16 mfma_bf16x8;
BD_DEV f32x16 mfma32(u32x4 a, u32x4 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, a),
                                                   __builtin_bit_cast(mfma_bf16x8, b), c, 0, 0, 0);
}

BD_DEV void ld_bf16x8_attn(const bf16_t* p, float* v) {
    const u32x4 q = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = bf2f((bf16_t)(q[j] & 0xffff)); v[2 * j + 1] = bf2f((bf16_t)(q[j] >> 16)); }
}
#define KSTR 136   // K tile row stride (bf16): 128 + 8 -> conflict-free ds_read_b128 across 16 rows
#define VSTR 72    // V^T tile row stride (bf16): 64 + 8

BD_DEV int mfma_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// P (fp32, this lane's 16 scores of a 32-key block: rows mfma_row(r)) -> the two bf16 A operands
// (keys 0..15 and 16..31 of the block) for O += P V.
BD_DEV void p_to_afrags(const float* p, int lane, u32x4& a_lo, u32x4& a_hi) {
    const bool up = lane >= 32;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
        const int r0 = hf * 8;
        const unsigned X0 = pack2(p[r0 + 0], p[r0 + 1]), X1 = pack2(p[r0 + 2], p[r0 + 3]);
        const unsigned Y0 = pack2(p[r0 + 4], p[r0 + 5]), Y1 = pack2(p[r0 + 6], p[r0 + 7]);
        const unsigned tX0 = __shfl_xor(X0, 32), tX1 = __shfl_xor(X1, 32);
        const unsigned tY0 = __shfl_xor(Y0, 32), tY1 = __shfl_xor(Y1, 32);
        // lanes <32 need keys base+0..7 = own(0-3) | partner(4-7); lanes >=32 need base+8..15 = partner(8-11) | own(12-15)
        const u32x4 f = up ? (u32x4){tY0, tY1, Y0, Y1} : (u32x4){X0, X1, tX0, tX1};
        if (hf == 0) a_lo = f; else a_hi = f;
    }
}

// ------------------------------------------------------------------------------------------------
// DiT head attention: seq = 64
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_attn_kernel(HeadAttnArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * KSTR];
    __shared__ __attribute__((aligned(16))) bf16_t Vs[128 * VSTR];
    BD_KSTAMP(a.stamp, 0);
    const int seq = blockIdx.x / a.nhead, h = blockIdx.x % a.nhead;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int D = a.D;
    const Partial& q = a.qkv;
    const bf16_t* bias = (const bf16_t*)q.bias;
    const size_t slab = (size_t)q.Mpad * q.N;

    auto load8 = [&](int row, int col, float* v) {          // Linear output (sum of slabs + bias), bf16-rounded
        if (q.S == 0) {                                     // finished bf16 tensor: one 16 B load
            const u32x4 w = *reinterpret_cast<const u32x4*>((const bf16_t*)q.p + (size_t)row * q.N + col);
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[2 * j] = bf2f((bf16_t)(w[j] & 0xffff)); v[2 * j + 1] = bf2f((bf16_t)(w[j] >> 16)); }
            return;
        }
        const float* p = q.p + (size_t)row * q.N + col;
        f32x4 lo = *reinterpret_cast<const f32x4*>(p), hi = *reinterpret_cast<const f32x4*>(p + 4);
        for (int s = 1; s < q.S; ++s) {
            lo += *reinterpret_cast<const f32x4*>(p + s * slab);
            hi += *reinterpret_cast<const f32x4*>(p + s * slab + 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = bfr(lo[j] + (bias ? bf2f(bias[col + j]) : 0.f));
            v[4 + j] = bfr(hi[j] + (bias ? bf2f(bias[col + 4 + j]) : 0.f));
        }
    };

    // stage K [key][d] and V^T [d][key] of this (seq, head): all 4 waves; waves 0/1 then own 32 queries each.
    u32x4 qf[8];
    const int qrow = seq * 64 + (wave & 1) * 32 + (lane & 31);
    if (q.S == 0) {
        // finished bf16 qkv (the GEMM reduced its K-slices): K rows are straight 16 B copies, Q fragments 16 B loads
        const bf16_t* base = (const bf16_t*)q.p;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int u = tid + it * 256;
            const int key = u >> 4, dp = (u & 15) * 8;
            const bf16_t* row = base + (size_t)(seq * 64 + key) * q.N + h * 128 + dp;
            *reinterpret_cast<u32x4*>(&Ks[key * KSTR + dp]) = *reinterpret_cast<const u32x4*>(row + D);
            const u32x4 vv = *reinterpret_cast<const u32x4*>(row + 2 * D);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                Vs[(dp + 2 * j) * VSTR + key] = (bf16_t)(vv[j] & 0xffff);
                Vs[(dp + 2 * j + 1) * VSTR + key] = (bf16_t)(vv[j] >> 16);
            }
        }
        if (wave < 2) {
            const bf16_t* qp = base + (size_t)qrow * q.N + h * 128 + (lane >> 5) * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) qf[ks] = *reinterpret_cast<const u32x4*>(qp + ks * 16);
        }
    } else {
        // split-K slabs: fully unrolled so that the 4 x (K + V) x S slab loads of a thread are all in flight together
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int u = tid + it * 256;
            const int key = u >> 4, dp = (u & 15) * 8;
            float kv[8], vv[8];
            load8(seq * 64 + key, D + h * 128 + dp, kv);
            load8(seq * 64 + key, 2 * D + h * 128 + dp, vv);
            *reinterpret_cast<u32x4*>(&Ks[key * KSTR + dp]) =
                (u32x4){pack2(kv[0], kv[1]), pack2(kv[2], kv[3]), pack2(kv[4], kv[5]), pack2(kv[6], kv[7])};
#pragma unroll
            for (int j = 0; j < 8; ++j) Vs[(dp + j) * VSTR + key] = f2bf(vv[j]);
        }
        if (wave < 2) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                float v[8];
                load8(qrow, h * 128 + ks * 16 + (lane >> 5) * 8, v);
                qf[ks] = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
            }
        }
    }
    __syncthreads();
    if (wave >= 2) return;

    f32x16 sacc[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const u32x4 kf = *reinterpret_cast<const u32x4*>(&Ks[(kb * 32 + (lane & 31)) * KSTR + ks * 16 + (lane >> 5) * 8]);
            sacc[kb] = mfma32(kf, qf[ks], sacc[kb]);
        }
    }
    const float scale = 0.08838834764831845f;                // 128^-0.5
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sacc[kb][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float p[2][16], lsum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { p[kb][r] = __expf((sacc[kb][r] - mx) * scale); lsum += p[kb][r]; }
    lsum += __shfl_xor(lsum, 32);

    // O^T = V^T P^T: D[row = d][col = query] -> a lane keeps ONE query (its own row sum normalises it, no shuffles) and
    // four consecutive d per register quad (8-byte bf16 stores straight into the fragment-major operand of `wo`)
    f32x16 oacc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[nb][r] = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
        u32x4 pa[2];
        p_to_afrags(p[kb], lane, pa[0], pa[1]);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int key0 = kb * 32 + hf * 16 + (lane >> 5) * 8;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const u32x4 vf = *reinterpret_cast<const u32x4*>(&Vs[(nb * 32 + (lane & 31)) * VSTR + key0]);
                oacc[nb] = mfma32(vf, pa[hf], oacc[nb]);
            }
        }
    }
    bf16_t* O = (bf16_t*)a.o_frag;
    const float inv = 1.0f / lsum;
    const int row = seq * 64 + wave * 32 + (lane & 31);
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int d0 = h * 128 + nb * 32 + 8 * qd + 4 * (lane >> 5);
            *reinterpret_cast<uint2*>(O + afrag_off(row, d0, a.RB)) =
                make_uint2(pack2(oacc[nb][4 * qd] / lsum, oacc[nb][4 * qd + 1] / lsum),
                           pack2(oacc[nb][4 * qd + 2] / lsum, oacc[nb][4 * qd + 3] / lsum));
        }
    (void)inv;
    BD_KSTAMP_END(a.stamp);
}

// seq <= 32 branch of the reference (flow_head:203-208), here P = 16 (the 16x models): explicit softmax with the
// reference's bf16 rounding points -- q*scale (bf16), scores = q k^T (bf16 matmul output), softmax in fp32, P cast to
// bf16 by the second matmul, output bf16.  16x16 scores per head: plain VALU, one thread per (query, key) then per
// (query, 8 channels).
__global__ __launch_bounds__(256) void head_attn16_kernel(HeadAttnArgs a) {
    __shared__ float qs[16 * 132], ks[16 * 132], vs[16 * 132], sc[16 * 17];
    const int seq = blockIdx.x / a.nhead, h = blockIdx.x % a.nhead;
    const int tid = threadIdx.x, D = a.D, dh = a.dh;           // dh = 128 (T2I heads) or 64 (imagenet diff_head_parallel.py:207)
    const int P = a.P;                                         // tokens per sequence: 16 (16x models) or 4 (ImageNet 4x)
    const Partial& q = a.qkv;
    const bf16_t* bias = (const bf16_t*)q.bias;
    const float scale = (dh == 64) ? 0.125f : 0.08838834764831845f;     // head_dim ** -0.5
    const bool act = (tid & 15) * 8 < dh && (tid >> 4) < P;    // thread -> (row i, 8 channels d0..d0+7)
    if (act) {   // Linear outputs (sum of slabs + bias) rounded to bf16
        const int i = tid >> 4, d0 = (tid & 15) * 8;
        for (int which = 0; which < 3; ++which) {
            const int col = which * D + h * dh + d0;
            const float* p = q.p + (size_t)(seq * P + i) * q.N + col;
            float* dst = (which == 0 ? qs : (which == 1 ? ks : vs)) + i * 132 + d0;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float v = 0.f;
                if (q.S == 0) {
                    v = bf2f(((const bf16_t*)q.p)[(size_t)(seq * P + i) * q.N + col + j]);
                } else {
                    for (int s_ = 0; s_ < q.S; ++s_) v += p[(size_t)s_ * q.Mpad * q.N + j];
                    v = bfr(v + (bias ? bf2f(bias[col + j]) : 0.f));
                }
                dst[j] = (which == 0) ? bfr(v * scale) : v;        // xq = xq * scale (bf16)
            }
        }
    }
    __syncthreads();
    {   // scores[i][j] = bf16( sum_d q_i[d] k_j[d] )
        const int i = tid >> 4, j = tid & 15;
        if (i < P && j < P) {
            float acc = 0.f;
            for (int d = 0; d < dh; ++d) acc += qs[i * 132 + d] * ks[j * 132 + d];
            sc[i * 17 + j] = bfr(acc);
        }
    }
    __syncthreads();
    if (act) {   // softmax over j in fp32, then out[i][d0..d0+7] = bf16( sum_j bf16(p_ij) v_j[d] )
        const int i = tid >> 4, d0 = (tid & 15) * 8;
        float m = -INFINITY;
        for (int j = 0; j < P; ++j) m = fmaxf(m, sc[i * 17 + j]);
        float e[16], sum = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) { e[j] = (j < P) ? expf(sc[i * 17 + j] - m) : 0.f; sum += e[j]; }
        float o[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) o[t] = 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (j < P) {
                const float pj = bfr(e[j] / sum);
#pragma unroll
                for (int t = 0; t < 8; ++t) o[t] += pj * vs[j * 132 + d0 + t];
            }
        }
        bf16_t* O = (bf16_t*)a.o_frag;
        *reinterpret_cast<u32x4*>(O + afrag_off(seq * P + i, h * dh + d0, a.RB)) =
            (u32x4){pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7])};
    }
}


// The same branch for a FINISHED bf16 qkv tensor (the GEMM rounded its own output): one wave per (sequence, head), four
// heads per workgroup.  q k^T on the matrix pipe straight from global memory (v_mfma_f32_16x16x32_bf16: a lane's 16 B of Q /
// K are exactly its A / B operand), softmax across the 16 lanes that hold one query row, P and the V tile through a few KiB
// of LDS, P V on the VALU (16 keys: 256 FMAs per lane).  Rounding points as above.  At the ImageNet batch (768 sequences x
// 12 heads) the element-wise kernel above took 1.48 ms per call (profiles/r02_kernel_stats_imagenet_b16x_batch384_10steps.md).
typedef __attribute__((ext_vector_type(8))) __bf16 bd_bf16x8v;
typedef __attribute__((ext_vector_type(4))) float f32x4v;
template <int DH>
__global__ __launch_bounds__(256) void head_attn16_mfma_kernel(HeadAttnArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t Vs[4][16 * DH];
    __shared__ float Ps[4][16 * 17];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + wave;                       // (sequence, head)
    if (pair >= a.nseq * a.nhead) return;                         // whole waves only: no barrier below is block-wide
    const int seq = pair / a.nhead, h = pair % a.nhead;
    const bf16_t* base = (const bf16_t*)a.qkv.p + (size_t)(seq * 16) * a.qkv.N + h * DH;
    const float scale = (DH == 64) ? 0.125f : 0.08838834764831845f;
    // V tile -> LDS (bf16, row-major [key][DH]); lane l: key l >> 2, 16 B pieces (l & 3) + 4 t
    {
        const bf16_t* vrow = base + 2 * a.D + (size_t)(lane >> 2) * a.qkv.N;
#pragma unroll
        for (int t = 0; t < DH / 32; ++t) {
            const int pc = (lane & 3) + 4 * t;
            *reinterpret_cast<u32x4*>(&Vs[wave][(lane >> 2) * DH + pc * 8]) = *reinterpret_cast<const u32x4*>(vrow + pc * 8);
        }
    }
    // scores: S[i][j] = bf16( sum_d bf16(q_i[d] * scale) k_j[d] ); this lane: i = 4 (lane >> 4) + r, j = lane & 15
    f32x4v sacc = {0.f, 0.f, 0.f, 0.f};
    {
        const bf16_t* qrow = base + (size_t)(lane & 15) * a.qkv.N + (lane >> 4) * 8;
        const bf16_t* krow = qrow + a.D;
#pragma unroll
        for (int kk = 0; kk < DH / 32; ++kk) {
            const u32x4 qraw = *reinterpret_cast<const u32x4*>(qrow + kk * 32);
            const u32x4 kraw = *reinterpret_cast<const u32x4*>(krow + kk * 32);
            u32x4 qs;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                qs[j] = pack2(bf2f((bf16_t)(qraw[j] & 0xffff)) * scale, bf2f((bf16_t)(qraw[j] >> 16)) * scale);   // xq * scale (bf16)
            sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bd_bf16x8v, qs), __builtin_bit_cast(bd_bf16x8v, kraw), sacc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const float sv = bfr(sacc[r]);                               // matmul output rounded to bf16
        float m = sv;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        const float e = expf(sv - m);
        float sum = e;
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        Ps[wave][((lane >> 4) * 4 + r) * 17 + (lane & 15)] = bfr(e / sum);        // P cast to bf16 by the second matmul
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");          // the wave's own LDS writes before its own reads
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // out[i][d0 .. d0 + DH/4) = bf16( sum_j p_ij v_j[d] ); lane: i = lane >> 2, quarter lane & 3
    constexpr int CH = DH / 4;
    const int i = lane >> 2, d0 = (lane & 3) * CH;
    float o[CH];
#pragma unroll
    for (int t = 0; t < CH; ++t) o[t] = 0.f;
#pragma unroll 4
    for (int j = 0; j < 16; ++j) {
        const float pj = Ps[wave][i * 17 + j];
#pragma unroll
        for (int g8 = 0; g8 < CH / 8; ++g8) {
            const u32x4 vq = *reinterpret_cast<const u32x4*>(&Vs[wave][j * DH + d0 + g8 * 8]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                o[g8 * 8 + 2 * t] += pj * bf2f((bf16_t)(vq[t] & 0xffff));
                o[g8 * 8 + 2 * t + 1] += pj * bf2f((bf16_t)(vq[t] >> 16));
            }
        }
    }
    bf16_t* O = (bf16_t*)a.o_frag;
#pragma unroll
    for (int g8 = 0; g8 < CH / 8; ++g8)
        *reinterpret_cast<u32x4*>(O + afrag_off(seq * 16 + i, h * DH + d0 + g8 * 8, a.RB)) =
            (u32x4){pack2(o[g8 * 8], o[g8 * 8 + 1]), pack2(o[g8 * 8 + 2], o[g8 * 8 + 3]), pack2(o[g8 * 8 + 4], o[g8 * 8 + 5]),
                    pack2(o[g8 * 8 + 6], o[g8 * 8 + 7])};
}

int bdk_head_attn(const HeadAttnArgs& a, hipStream_t st) {
    if (a.dh != 128 && !(a.dh == 64 && a.P <= 16)) return -2;
    BD_STAMPED(HeadAttnArgs, a, "head_attn", a.nseq * a.nhead);
    if (a.P == 64) BD_LAUNCH(head_attn_kernel, dim3(a.nseq * a.nhead), dim3(256), 0, st, a_l);
    else if (a.P == 16 && a.qkv.S == 0 && a.qkv.N % 8 == 0) {       // finished bf16 qkv: matrix-pipe scores, 4 heads per workgroup
        const int blocks = (a.nseq * a.nhead + 3) / 4;
        if (a.dh == 64) BD_LAUNCH(head_attn16_mfma_kernel<64>, dim3(blocks), dim3(256), 0, st, a);
        else BD_LAUNCH(head_attn16_mfma_kernel<128>, dim3(blocks), dim3(256), 0, st, a);
    }
    else if (a.P <= 16 && a.P >= 1) BD_LAUNCH(head_attn16_kernel, dim3(a.nseq * a.nhead), dim3(256), 0, st, a);   // 16x from slabs; 4x; 1 (out = v)
    else return -2;
    return bd_launch_status();
}

// ------------------------------------------------------------------------------------------------
// ImageNet transformer decode attention = the reference's naive_attention (layers_parallel.py:120-133) with its rounding
// points: scores = bf16(q_scaled k^T) (autocast matmul output), + mask (all zeros for a decode block: every cached key
// and the block itself are visible), softmax in fp32, P cast to bf16 by the second matmul, output bf16.
// One workgroup per (sequence, head): 16 queries x L <= 1152 keys x 64 dims -- plain VALU through LDS, the whole
// transformer is 4 % of this model's FLOPs.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void in_attn_kernel(InAttnArgs a) {
    extern __shared__ float smem_f[];
    const int seq = blockIdx.x / a.nh, h = blockIdx.x % a.nh, tid = threadIdx.x;
    const int L = a.state->kv_len[0] + a.P;
    const int LS = ((L + 63) & ~63) + 1;
    float* qs = smem_f;                  // [16][64]
    float* kt = qs + 16 * 64;            // [64][65]
    float* sc = kt + 64 * 65;            // [16][LS]
    const int D = a.nh * 64;
    const bf16_t* Q = (const bf16_t*)a.q + (size_t)seq * a.P * D + h * 64;
    const bf16_t* Kc = a.k_cache + ((size_t)seq * a.nh + h) * a.Lmax * 64;
    const bf16_t* Vc = a.v_cache + ((size_t)seq * a.nh + h) * a.Lmax * 64;
    for (int e = tid; e < 16 * 64; e += 256) qs[e] = (e >> 6) < a.P ? bf2f(Q[(size_t)(e >> 6) * D + (e & 63)]) : 0.f;
    const int i = tid >> 4, jb = tid & 15;
    for (int t0 = 0; t0 < L; t0 += 64) {
        __syncthreads();
        for (int e = tid; e < 64 * 64; e += 256) {
            const int j = e >> 6, d = e & 63;
            kt[j * 65 + d] = (t0 + j < L) ? bf2f(Kc[(size_t)(t0 + j) * 64 + d]) : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = jb + 16 * jj;
            float acc = 0.f;
            for (int d = 0; d < 64; ++d) acc += qs[i * 64 + d] * kt[j * 65 + d];
            if (t0 + j < L) sc[i * LS + t0 + j] = (a.causal && t0 + j > L - a.P + i) ? -INFINITY : bfr(acc);   // att + mask
        }
    }
    __syncthreads();
    {   // softmax over the row in fp32; 16 threads per row
        float mx = -INFINITY;
        for (int j = jb; j < L; j += 16) mx = fmaxf(mx, sc[i * LS + j]);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (int j = jb; j < L; j += 16) { const float e = expf(sc[i * LS + j] - mx); sc[i * LS + j] = e; sum += e; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        for (int j = jb; j < L; j += 16) sc[i * LS + j] = bfr(sc[i * LS + j] / sum);
    }
    float o4[4] = {0.f, 0.f, 0.f, 0.f};
    const int d0 = jb * 4;
    for (int t0 = 0; t0 < L; t0 += 64) {
        __syncthreads();
        for (int e = tid; e < 64 * 64; e += 256) {
            const int j = e >> 6, d = e & 63;
            kt[j * 65 + d] = (t0 + j < L) ? bf2f(Vc[(size_t)(t0 + j) * 64 + d]) : 0.f;
        }
        __syncthreads();
        const int nj = min(64, L - t0);
        for (int j = 0; j < nj; ++j) {
            const float pj = sc[i * LS + t0 + j];
#pragma unroll
            for (int t = 0; t < 4; ++t) o4[t] += pj * kt[j * 65 + d0 + t];
        }
    }
    if (i < a.P) {
        bf16_t* O = (bf16_t*)a.o_frag;
        *reinterpret_cast<uint2*>(O + afrag_off(seq * a.P + i, h * 64 + d0, a.RB)) =
            make_uint2(pack2(o4[0], o4[1]), pack2(o4[2], o4[3]));
    }
}

// The same arithmetic on the matrix pipe, one wave per (sequence, head), four pairs per workgroup, all LDS wave-private:
//   scores: v_mfma_f32_16x16x32_bf16 with Q (16 queries x 64) resident in registers and 16 keys per step read straight from
//           the cache (a lane's 16 B = its B operand), four steps of loads in flight; bf16(score) parked in LDS;
//   softmax over the whole row (the reference rounds P = e / sum to bf16 only after the full-row sum): max, sum, then
//           p = bf16(e / sum) overwrites the score;
//   P V on the VALU from 64-key V tiles staged in LDS (16 x L x 64 MACs per pair).
// 12 + 8 KiB of LDS per wave at L <= 384: two workgroups per CU.  ImageNet batch 384: 1027 -> see profiles/.
__global__ __launch_bounds__(256) void in_attn_mfma_kernel(InAttnArgs a, int LS) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= a.nseq * a.nh) return;                              // wave-private LDS: no block-wide barrier below
    bf16_t* sc = reinterpret_cast<bf16_t*>(smem_raw) + (size_t)wave * (16 * LS + 64 * 64);   // [16][LS] scores, then P
    bf16_t* Vs = sc + 16 * LS;                                      // [64 keys][64] V tile
    const int seq = pair / a.nh, h = pair % a.nh;
    const int L = a.state->kv_len[0] + a.P;
    const int D = a.nh * 64;
    const bf16_t* Q = (const bf16_t*)a.q + (size_t)seq * 16 * D + h * 64;
    const bf16_t* Kc = a.k_cache + ((size_t)seq * a.nh + h) * a.Lmax * 64;
    const bf16_t* Vc = a.v_cache + ((size_t)seq * a.nh + h) * a.Lmax * 64;
    const int r16 = lane & 15, kg = (lane >> 4) * 8;
    u32x4 qa[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) qa[kk] = *reinterpret_cast<const u32x4*>(Q + (size_t)r16 * D + kk * 32 + kg);
    // ---- scores
    for (int t0 = 0; t0 < L; t0 += 64) {
        u32x4 kb[4][2];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = t0 + u * 16 + r16;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                kb[u][kk] = (key < L) ? *reinterpret_cast<const u32x4*>(Kc + (size_t)key * 64 + kk * 32 + kg) : (u32x4){0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x4v sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
                sacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bd_bf16x8v, qa[kk]), __builtin_bit_cast(bd_bf16x8v, kb[u][kk]), sacc, 0, 0, 0);
            const int key = t0 + u * 16 + r16;                      // D layout: row (query) 4 (lane >> 4) + r, column (key) lane & 15
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool vis = key < L && !(a.causal && key > L - a.P + (lane >> 4) * 4 + r);   // causal: keys <= past + query index
                sc[((lane >> 4) * 4 + r) * LS + key] = vis ? f2bf(sacc[r]) : (bf16_t)0xff80;        // -inf: pad / masked
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- softmax: 4 lanes per query row
    const int i = lane >> 2, sub = lane & 3;
    const int Lp = (L + 63) & ~63;
    {
        bf16_t* row = sc + i * LS;
        float mx = -INFINITY;
        for (int j = sub; j < L; j += 4) mx = fmaxf(mx, bf2f(row[j]));
        mx = fmaxf(mx, __shfl_xor(mx, 1)); mx = fmaxf(mx, __shfl_xor(mx, 2));
        float sum = 0.f;
        for (int j = sub; j < L; j += 4) sum += expf(bf2f(row[j]) - mx);
        sum += __shfl_xor(sum, 1); sum += __shfl_xor(sum, 2);
        for (int j = sub; j < Lp; j += 4) row[j] = (j < L) ? f2bf(expf(bf2f(row[j]) - mx) / sum) : (bf16_t)0;   // P as bf16; pads 0
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    // ---- P V: lane -> query i, 16 channels d0..
    const int d0 = sub * 16;
    float o[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) o[t] = 0.f;
    for (int t0 = 0; t0 < L; t0 += 64) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {                               // 64 keys x 8 pieces of 16 B
            const int u = lane + 64 * t, key = u >> 3, pc = u & 7;
            *reinterpret_cast<u32x4*>(Vs + key * 64 + pc * 8) =
                (t0 + key < L) ? *reinterpret_cast<const u32x4*>(Vc + (size_t)(t0 + key) * 64 + pc * 8) : (u32x4){0, 0, 0, 0};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll 2
        for (int j8 = 0; j8 < 8; ++j8) {
            const u32x4 pq = *reinterpret_cast<const u32x4*>(sc + i * LS + t0 + j8 * 8);
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const float pj = bf2f((bf16_t)((jj & 1) ? (pq[jj >> 1] >> 16) : (pq[jj >> 1] & 0xffff)));
                const bf16_t* vrow = Vs + (j8 * 8 + jj) * 64 + d0;
                const u32x4 v0 = *reinterpret_cast<const u32x4*>(vrow), v1 = *reinterpret_cast<const u32x4*>(vrow + 8);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    o[2 * t] += pj * bf2f((bf16_t)(v0[t] & 0xffff));
                    o[2 * t + 1] += pj * bf2f((bf16_t)(v0[t] >> 16));
                    o[8 + 2 * t] += pj * bf2f((bf16_t)(v1[t] & 0xffff));
                    o[8 + 2 * t + 1] += pj * bf2f((bf16_t)(v1[t] >> 16));
                }
            }
        }
        __builtin_amdgcn_wave_barrier();                            // the tile is consumed before the next one overwrites it
    }
    bf16_t* O = (bf16_t*)a.o_frag;
#pragma unroll
    for (int g8 = 0; g8 < 2; ++g8)
        *reinterpret_cast<u32x4*>(O + afrag_off(seq * 16 + i, h * 64 + d0 + g8 * 8, a.RB)) =
            (u32x4){pack2(o[g8 * 8], o[g8 * 8 + 1]), pack2(o[g8 * 8 + 2], o[g8 * 8 + 3]), pack2(o[g8 * 8 + 4], o[g8 * 8 + 5]),
                    pack2(o[g8 * 8 + 6], o[g8 * 8 + 7])};
}

// The 1x / 4x checkpoints (one or four queries per sequence and step): the 16-query forms above spend a 256-thread workgroup and
// scalar bf16 loads on a row or four (795 us per call at the ImageNet batch of B-1x: a third of its AR step).  Here a WAVE owns a
// (sequence, head) and every global access is one contiguous KiB: lane = (key of a group of 8, 16 B piece of its 128 B row).
//   scores: a lane's 8-channel partial of q . k (fp32, separate multiply and add), summed over the row's 8 lanes by three xor
//           steps, rounded to bf16 (the autocast matmul output), masked, parked in wave-private LDS;
//   softmax over the whole row in fp32, P = bf16(e / sum) (the reference rounds P only after the full-row sum);
//   P V:    the same lane map on V -- a lane accumulates its 8 channels over the keys of its residue class, the 8 classes are
//           summed by three xor steps, the class-0 lanes store 16 B of the fragment-major output.
// Rounding points as in in_attn_kernel; the fp32 sums run in a different (tree) order.
template <int P>
__global__ __launch_bounds__(256) void in_attn_small_kernel(InAttnArgs a, int LS) {
    extern __shared__ float smem_f[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= a.nseq * a.nh) return;                              // wave-private LDS: no block-wide barrier below
    const int seq = pair / a.nh, h = pair % a.nh;
    const int L = a.state->kv_len[0] + P;
    float* sc = smem_f + (size_t)wave * (P * LS);                   // [P][LS]
    const int D = a.nh * 64;
    const int g = lane >> 3, pc = lane & 7;
    const bf16_t* Q = (const bf16_t*)a.q + (size_t)seq * P * D + h * 64 + pc * 8;
    const bf16_t* Kc = a.k_cache + ((size_t)seq * a.nh + h) * a.Lmax * 64 + pc * 8;
    const bf16_t* Vc = a.v_cache + ((size_t)seq * a.nh + h) * a.Lmax * 64 + pc * 8;
    float q[P][8];
#pragma unroll
    for (int i = 0; i < P; ++i) ld_bf16x8_attn(Q + (size_t)i * D, q[i]);
    for (int t0 = 0; t0 < L; t0 += 32) {                            // 4 groups of 8 keys: four loads in flight per lane
        u32x4 kr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = t0 + u * 8 + g;
            kr[u] = (j < L) ? *reinterpret_cast<const u32x4*>(Kc + (size_t)j * 64) : (u32x4){0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = t0 + u * 8 + g;
            float kf[8];
#pragma unroll
            for (int t = 0; t < 4; ++t) { kf[2 * t] = bf2f((bf16_t)(kr[u][t] & 0xffff)); kf[2 * t + 1] = bf2f((bf16_t)(kr[u][t] >> 16)); }
#pragma unroll
            for (int i = 0; i < P; ++i) {
                float acc = 0.f;
#pragma unroll
                for (int t = 0; t < 8; ++t) acc = fadd(acc, fmul(q[i][t], kf[t]));
                acc = fadd(acc, __shfl_xor(acc, 1));
                acc = fadd(acc, __shfl_xor(acc, 2));
                acc = fadd(acc, __shfl_xor(acc, 4));
                if (pc == 0 && j < L) sc[i * LS + j] = (a.causal && j > L - P + i) ? -INFINITY : bfr(acc);   // att + mask
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < P; ++i) {                                   // softmax over the row in fp32, P = bf16(e / sum)
        float mx = -INFINITY;
        for (int j = lane; j < L; j += 64) mx = fmaxf(mx, sc[i * LS + j]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < L; j += 64) { const float e = expf(sc[i * LS + j] - mx); sc[i * LS + j] = e; sum += e; }
        sum = wave_sum(sum);
        for (int j = lane; j < L; j += 64) sc[i * LS + j] = bfr(sc[i * LS + j] / sum);
    }
    __builtin_amdgcn_wave_barrier();
    float o[P][8];
#pragma unroll
    for (int i = 0; i < P; ++i)
#pragma unroll
        for (int t = 0; t < 8; ++t) o[i][t] = 0.f;
    for (int t0 = 0; t0 < L; t0 += 32) {
        u32x4 vr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = t0 + u * 8 + g;
            vr[u] = (j < L) ? *reinterpret_cast<const u32x4*>(Vc + (size_t)j * 64) : (u32x4){0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = t0 + u * 8 + g;
            if (j < L) {
                float vf[8];
#pragma unroll
                for (int t = 0; t < 4; ++t) { vf[2 * t] = bf2f((bf16_t)(vr[u][t] & 0xffff)); vf[2 * t + 1] = bf2f((bf16_t)(vr[u][t] >> 16)); }
#pragma unroll
                for (int i = 0; i < P; ++i) {
                    const float pj = sc[i * LS + j];
#pragma unroll
                    for (int t = 0; t < 8; ++t) o[i][t] = fadd(o[i][t], fmul(pj, vf[t]));
                }
            }
        }
    }
    bf16_t* O = (bf16_t*)a.o_frag;
#pragma unroll
    for (int i = 0; i < P; ++i) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            float v = o[i][t];
            v = fadd(v, __shfl_xor(v, 8));
            v = fadd(v, __shfl_xor(v, 16));
            v = fadd(v, __shfl_xor(v, 32));
            o[i][t] = v;
        }
        if (g == 0)
            *reinterpret_cast<u32x4*>(O + afrag_off(seq * P + i, h * 64 + pc * 8, a.RB)) =
                (u32x4){pack2(o[i][0], o[i][1]), pack2(o[i][2], o[i][3]), pack2(o[i][4], o[i][5]), pack2(o[i][6], o[i][7])};
    }
}

int bdk_in_attn(const InAttnArgs& a, hipStream_t st) {
    if (a.P > 16) return -2;
    if (a.P == 1 || a.P == 4) {                                    // the 1x / 4x checkpoints: a wave per (sequence, head)
        const int LS = ((a.Lmax + 64 + 63) & ~63) + 4;             // row stride in floats
        const size_t lds_s = (size_t)4 * (a.P * LS) * sizeof(float);
        if (lds_s <= 64 * 1024) {
            const dim3 grid((a.nseq * a.nh + 3) / 4);
            if (a.P == 1) BD_LAUNCH(in_attn_small_kernel<1>, grid, dim3(256), lds_s, st, a, LS);
            else BD_LAUNCH(in_attn_small_kernel<4>, grid, dim3(256), lds_s, st, a, LS);
            return bd_launch_status();
        }
    }
    {   // matrix-pipe form: whole 16-token blocks, score rows that fit the per-wave LDS budget
        const int LS = ((a.Lmax + 64 + 63) & ~63) + 8;             // row stride in bf16: multiple of 8 (16 B reads), off the bank period
        const size_t lds_m = (size_t)4 * (16 * LS + 64 * 64) * sizeof(bf16_t);
        static unsigned long long optin_m = 0;
        const bool ok = bd_lds_optin((const void*)in_attn_mfma_kernel, 160 * 1024, &optin_m);
        if (a.P == 16 && ok && lds_m <= 160 * 1024) {
            BD_LAUNCH(in_attn_mfma_kernel, dim3((a.nseq * a.nh + 3) / 4), dim3(256), lds_m, st, a, LS);
            return bd_launch_status();
        }
    }
    const int Lcap = a.Lmax + 64;
    const size_t lds = (size_t)(16 * 64 + 64 * 65 + 16 * (Lcap + 1)) * sizeof(float);
    if (lds > 160 * 1024) return -3;
    static unsigned long long optin = 0;
    if (!bd_lds_optin((const void*)in_attn_kernel, 160 * 1024, &optin)) return -8;
    BD_LAUNCH(in_attn_kernel, dim3(a.nseq * a.nh), dim3(256), lds, st, a);
    return bd_launch_status();
}

// ------------------------------------------------------------------------------------------------
// LLM decode attention (flash-decode over the static KV cache) + combine
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(640) void llm_attn_kernel(LlmAttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_la[];
    bf16_t* const Ks = reinterpret_cast<bf16_t*>(smem_la);            // [64 keys][KSTR]
    bf16_t* const Vs = Ks + 64 * KSTR;                                // [128 d][VSTR]
    const int split = blockIdx.x, kvh = blockIdx.y, seq = blockIdx.z;
    const int G = a.nh / a.nkv;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, NT = blockDim.x;
    const int halves = (a.P + 31) / 32;                       // P = 64: two 32-query halves per head; P = 16: one, padded
    const int qh = wave / halves, half = wave % halves;
    const int head = kvh * G + qh;
    const bool qvalid = half * 32 + (lane & 31) < a.P;
    const int past = a.state->kv_len[seq];
    const int L = past + a.P;                                // keys visible to this block of queries
    const int qlim = a.causal ? past + half * 32 + (lane & 31) : L - 1;   // last key this lane's query may see
    const int ntiles = (L + 63) >> 6;
    const int per = (ntiles + a.splits - 1) / a.splits;
    const int t_beg = split * per, t_end = min(ntiles, t_beg + per);

    const bf16_t* Kc = (const bf16_t*)a.k_cache + ((size_t)seq * a.nkv + kvh) * a.Lmax * 128;
    const bf16_t* Vc = (const bf16_t*)a.vt_cache + ((size_t)seq * a.nkv + kvh) * 128 * a.Lmax;

    // the wave's Q fragments (8 k-steps x 16 B per lane) live in LDS, each lane reading back exactly what it parked: 32 registers
    // freed for the K / V prefetch below (the kernel sits at the 168-register line of 3 waves per SIMD)
    u32x4* const Qs = reinterpret_cast<u32x4*>(Vs + 128 * VSTR) + (size_t)wave * 512 + lane;
    {
        const int qrow = qvalid ? half * 32 + (lane & 31) : 0;   // padded lanes recompute row 0, never stored
        const bf16_t* qp = (const bf16_t*)a.q + ((size_t)(seq * a.P + qrow) * a.nh + head) * 128 + (lane >> 5) * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) Qs[ks * 64] = *reinterpret_cast<const u32x4*>(qp + ks * 16);
    }
    const float scale = 0.08838834764831845f;
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 oacc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[nb][r] = 0.f;

    // One 64-key tile = a K phase (scores + online softmax, from Ks) and a V phase (P V, from Vs).  With two 16 B units per thread
    // and tile (NT >= 512: the 14B model's 10 waves) the tile's V^T is requested before its K phase and the NEXT tile's K before its
    // V phase, each parked in the same 8 registers and written to LDS behind the phase that hides it: the loop used to load K and V,
    // wait, and only then compute -- 2 exposed HBM / L2 round trips per tile of a kernel that runs 3-5 tiles.
    float p[2][16];
    auto k_phase = [&](int key_base) {
        f32x16 sacc[2];
        float tmax = -INFINITY;
        int qi = 0;
        asm volatile("" : "+v"(qi));                          // opaque zero: keeps the Q fragment reads inside the tile loop (not hoisted back into registers)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sacc[kb][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(&Ks[(kb * 32 + (lane & 31)) * KSTR + ks * 16 + (lane >> 5) * 8]);
                sacc[kb] = mfma32(kf, Qs[ks * 64 + qi], sacc[kb]);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = key_base + kb * 32 + mfma_row(r, lane);
                const float s = (key < L && key <= qlim) ? sacc[kb][r] : -INFINITY;
                p[kb][r] = s;
                tmax = fmaxf(tmax, s);
            }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float m_new = fmaxf(m_run, tmax);               // -inf only while a causal query has not met a visible key yet
        const bool none = (m_new == -INFINITY);
        const float alpha = none ? 1.f : __expf((m_run - m_new) * scale);
        float tsum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[kb][r] = none ? 0.f : __expf((p[kb][r] - m_new) * scale); tsum += p[kb][r]; }
        tsum += __shfl_xor(tsum, 32);
        l_run = l_run * alpha + tsum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) oacc[nb][r] *= alpha;     // O^T layout: the lane's own query
    };
    auto v_phase = [&]() {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            u32x4 pa[2];
            p_to_afrags(p[kb], lane, pa[0], pa[1]);
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int key0 = kb * 32 + hf * 16 + (lane >> 5) * 8;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const u32x4 vf = *reinterpret_cast<const u32x4*>(&Vs[(nb * 32 + (lane & 31)) * VSTR + key0]);
                    oacc[nb] = mfma32(vf, pa[hf], oacc[nb]);        // O^T[d][query]
                }
            }
        }
        BD_MFMA_DRAIN();                                     // the accumulators are read (scaled) right after, across a loop edge (bd_common.h)
    };
    if (NT >= 512) {
        const int u0 = tid, u1 = tid + NT;
        const bool two = u1 < 1024;
        u32x4 r0, r1 = {0, 0, 0, 0};
        auto ld_k = [&](int kb_) {
            r0 = *reinterpret_cast<const u32x4*>(Kc + (size_t)(kb_ + (u0 >> 4)) * 128 + (u0 & 15) * 8);
            if (two) r1 = *reinterpret_cast<const u32x4*>(Kc + (size_t)(kb_ + (u1 >> 4)) * 128 + (u1 & 15) * 8);
        };
        auto st_k = [&]() {
            *reinterpret_cast<u32x4*>(&Ks[(u0 >> 4) * KSTR + (u0 & 15) * 8]) = r0;
            if (two) *reinterpret_cast<u32x4*>(&Ks[(u1 >> 4) * KSTR + (u1 & 15) * 8]) = r1;
        };
        auto ld_v = [&](int kb_) {
            r0 = *reinterpret_cast<const u32x4*>(Vc + (size_t)(u0 >> 3) * a.Lmax + kb_ + (u0 & 7) * 8);
            if (two) r1 = *reinterpret_cast<const u32x4*>(Vc + (size_t)(u1 >> 3) * a.Lmax + kb_ + (u1 & 7) * 8);
        };
        auto st_v = [&]() {
            *reinterpret_cast<u32x4*>(&Vs[(u0 >> 3) * VSTR + (u0 & 7) * 8]) = r0;
            if (two) *reinterpret_cast<u32x4*>(&Vs[(u1 >> 3) * VSTR + (u1 & 7) * 8]) = r1;
        };
        if (t_beg < t_end) { ld_k(t_beg * 64); st_k(); }
        for (int t = t_beg; t < t_end; ++t) {
            const int key_base = t * 64;
            ld_v(key_base);                                  // in flight under the K phase
            __syncthreads();                                 // Ks(t) visible; Vs(t - 1) consumed
            k_phase(key_base);
            st_v();
            __syncthreads();                                 // Vs(t) visible; Ks(t) consumed
            if (t + 1 < t_end) ld_k(key_base + 64);          // in flight under the V phase
            v_phase();
            if (t + 1 < t_end) st_k();
        }
    } else {
        for (int t = t_beg; t < t_end; ++t) {
            const int key_base = t * 64;
            __syncthreads();
            for (int u = tid; u < 1024; u += NT) {               // K tile: 64 keys x 16 pieces of 8 d
                const int key = u >> 4, dp = (u & 15) * 8;
                *reinterpret_cast<u32x4*>(&Ks[key * KSTR + dp]) =
                    *reinterpret_cast<const u32x4*>(Kc + (size_t)(key_base + key) * 128 + dp);
            }
            for (int u = tid; u < 1024; u += NT) {               // V^T tile: 128 d x 8 pieces of 8 keys
                const int d = u >> 3, kp = (u & 7) * 8;
                *reinterpret_cast<u32x4*>(&Vs[d * VSTR + kp]) =
                    *reinterpret_cast<const u32x4*>(Vc + (size_t)d * a.Lmax + key_base + kp);
            }
            __syncthreads();
            k_phase(key_base);
            v_phase();
        }
    }
    // partial results: [seq][kvh][split][G*P rows][128] and (m, l) per row
    const size_t blk = ((size_t)seq * a.nkv + kvh) * a.splits + split;
    const int rows = G * a.P;
    float* op = a.o_part + blk * rows * 128;
    float* ml = a.ml_part + blk * rows * 2;
    const int rbase = qh * a.P + half * 32;
    if (qvalid) {
        float* orow = op + (size_t)(rbase + (lane & 31)) * 128 + 4 * (lane >> 5);
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
                *reinterpret_cast<f32x4*>(orow + nb * 32 + 8 * qd) =
                    (f32x4){oacc[nb][4 * qd], oacc[nb][4 * qd + 1], oacc[nb][4 * qd + 2], oacc[nb][4 * qd + 3]};
    }
    if (lane < 32 && qvalid) { ml[(rbase + lane) * 2] = m_run; ml[(rbase + lane) * 2 + 1] = l_run; }
}

__global__ __launch_bounds__(256) void llm_attn_combine_kernel(LlmAttnArgs a) {
    const int m = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int G = a.nh / a.nkv;
    const int seq = m / a.P, pq = m % a.P;
    const float scale = 0.08838834764831845f;
    bf16_t* O = (bf16_t*)a.o_frag;
    for (int head = blockIdx.y * 4 + (threadIdx.x >> 6); head < a.nh; head += gridDim.y * 4) {
        const int kvh = head / G, qh = head % G;
        const int rows = G * a.P, row = qh * a.P + pq;
        const size_t blk0 = ((size_t)seq * a.nkv + kvh) * a.splits;
        if (a.splits <= 16) {
            // every load of the wave issued before the first use (the loop below was three dependent round trips per split: 12.7 us
            // for 40 KB of work); lane j holds split j's (m, l); the sums run over the splits in ascending order exactly as below
            float mj = -INFINITY, lj = 0.f;
            if (lane < a.splits) {
                mj = a.ml_part[((blk0 + lane) * rows + row) * 2];
                lj = a.ml_part[((blk0 + lane) * rows + row) * 2 + 1];
            }
            float p0[16], p1[16];
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < a.splits) {
                    const float* op = a.o_part + ((blk0 + j) * rows + row) * 128;
                    p0[j] = op[lane];
                    p1[j] = op[lane + 64];
                }
            const float M = wave_max(mj);
            const float wv = (mj == -INFINITY) ? 0.f : __expf((mj - M) * scale);
            float l = 0.f, o0 = 0.f, o1 = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < a.splits) {
                    const float mjj = __shfl(mj, j);
                    if (mjj == -INFINITY) continue;                // split without keys (its partial rows were never written)
                    const float w = __shfl(wv, j);
                    l += w * __shfl(lj, j);
                    o0 += w * p0[j];
                    o1 += w * p1[j];
                }
            O[afrag_off(m, head * 128 + lane, a.RB)] = f2bf(o0 / l);
            O[afrag_off(m, head * 128 + lane + 64, a.RB)] = f2bf(o1 / l);
            continue;
        }
        float M = -INFINITY;
        for (int j = 0; j < a.splits; ++j) M = fmaxf(M, a.ml_part[((blk0 + j) * rows + row) * 2]);
        float l = 0.f, o0 = 0.f, o1 = 0.f;
        for (int j = 0; j < a.splits; ++j) {
            const float mj = a.ml_part[((blk0 + j) * rows + row) * 2];
            if (mj == -INFINITY) continue;                     // split without keys
            const float w = __expf((mj - M) * scale);
            l += w * a.ml_part[((blk0 + j) * rows + row) * 2 + 1];
            const float* op = a.o_part + ((blk0 + j) * rows + row) * 128;
            o0 += w * op[lane];
            o1 += w * op[lane + 64];
        }
        O[afrag_off(m, head * 128 + lane, a.RB)] = f2bf(o0 / l);
        O[afrag_off(m, head * 128 + lane + 64, a.RB)] = f2bf(o1 / l);
    }
}

int bdk_llm_attn(const LlmAttnArgs& a, hipStream_t st) {
    const int G = a.nh / a.nkv;
    const int halves = (a.P + 31) / 32;
    if (a.P < 1 || a.P > 64 || G * halves * 64 > 640 || a.nh % a.nkv) return -2;   // G <= 5 (Qwen3-14B: 40/8); P = 1 / 4: a mostly padded half
    const int lds = (64 * KSTR + 128 * VSTR) * (int)sizeof(bf16_t) + G * halves * 8192;     // K / V^T tiles + 8 KiB of Q fragments per wave
    static unsigned long long optin = 0;
    if (!bd_lds_optin((const void*)llm_attn_kernel, 160 * 1024, &optin)) return -8;
    BD_LAUNCH(llm_attn_kernel, dim3(a.splits, a.nkv, a.nseq), dim3(G * halves * 64), lds, st, a);
    BD_LAUNCH(llm_attn_combine_kernel, dim3(a.nseq * a.P, (a.nh + 3) / 4), dim3(256), 0, st, a);
    return bd_launch_status();
}
