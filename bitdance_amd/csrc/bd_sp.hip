// Sequence-parallel row kernels of the diffusion head under tensor parallelism (round 5).
//
// What one DiT block computes between its Linears (flow_head_parallel_x.py:242-252) is row-wise:
//     x = x + gate * branch ;  h = LN(x) * (1 + scale) + shift
// In the all-reduce form (bd_comm.hip tp_allreduce_kernel + bd_rows.hip ln_mod_kernel) every rank receives the whole reduced
// branch and repeats that for all rows: per row-split Linear two launches and two flag round trips that do not shrink with the
// tensor-parallel size.  Here a rank OWNS rows / size rows of the residual stream (8-row groups dealt round-robin, so that the cond
// row bp and the uncond row BP + bp of a patch position meet on one rank):
//
//   row-split GEMM (wo / w2)  epilogue pushes every owner its rows of the fp32 partial          (bd_gemm_kernel.h, BdTpPush::il)
//   ln_mod_sp  (this file)    owner: signal / wait "partials pushed", sum the partials IN RANK ORDER + bias, one bf16 rounding (the
//                             value the all-reduce form computes, bit for bit), gate, residual, LayerNorm, modulate -- for its rows --
//                             and push the bf16 operand rows into EVERY rank's landing buffer, then one flag per row
//   column-split GEMM (qkv / w1)  issues its first weight stages, polls the row flags, invalidates, loads the operand (bd_gemm_kernel.h)
//
// and the evaluation ends in head_final_sp: the final layer, x_hat, the CFG mix and the sampler step for the rank's own patch
// positions -- nothing is exchanged until the LAST evaluation of an AR step, whose latent rows go to every rank (tok_finish).
// No stand-alone exchange kernel is left in the evaluation: 12 row kernels with one flag round each, instead of 12 exchanges with two
// flag rounds plus 12 replicated row kernels.
//
// Flags: epoch = replay counter * 2^16 + the hand-off's sequence number inside the replayed graph (BD_SP_*, bd_kernels.h; bd_common.h
// bd_sp_epoch_of): values only grow modulo 2^32 and are compared in unsigned arithmetic, so a captured graph replays without resets.  Payload and flags follow bd_comm.hip's hand-off: sc0 sc1 write-through 16 B stores,
// s_waitcnt vmcnt(0), relaxed system-scope flag store; readers of remotely written staging rows use sc0 sc1 loads.  Every wait is
// bounded by a wall-clock budget and reports through the communicator's error word (bd_comm_error).
#include "bd_rowhelp.h"
#include "bd_hwait.h"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "bd_sp.hip: the sc0 sc1 / vmcnt(0) flag hand-off is validated for gfx950 only"
#endif

#define BD_SYS_AUX 17                     /* cache policy bits: sc0 | sc1 */
BD_DEV __amdgpu_buffer_rsrc_t sp_rsrc(void* base, long long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)bytes, 0x00020000);
}
BD_DEV int sp_epoch(const BdSpLink& L, int seq) {
    return bd_sp_epoch_of(__hip_atomic_load(L.spf_local + BD_SP_RC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), seq);
}
// global row of this rank's local row lr: 8-row group (lr >> 3) * size + rank
BD_DEV int sp_row(int rank, int size, int lr) { return (((lr >> 3) * size + rank) << 3) + (lr & 7); }

// "every push of my row-split GEMM is at its destination" (the GEMM drained its stores before it ended; this kernel starts behind the
// kernel boundary): one block tells every peer
BD_DEV void sp_signal_p(const BdSpLink& L, int e) {
    const int t = bd_spread_lane(L.size);                     // one flag per wave: parallel fabric writes (bd_common.h)
    if (t >= 0 && t != L.rank) {
        int* const dst = L.loopback ? L.spf_local + BD_SP_P + t : L.spf[t] + BD_SP_P + L.rank;
        __hip_atomic_store(dst, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// all threads call; false (block-uniform): the exchange is dead (a wait ran out of its budget here or on a peer)
BD_DEV bool sp_wait_p(const BdSpLink& L, int e, int* alive_sh) {
    if (threadIdx.x == 0) *alive_sh = 1;
    __syncthreads();
    const int t = bd_spread_lane(L.size);
    if (t >= 0 && t != L.rank) {
        const int* f = L.spf_local + BD_SP_P + t;
        const long long t0 = wall_clock64();
        bool dead = __hip_atomic_load(L.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
        while (!dead && bd_epoch_before(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), e)) {
            __builtin_amdgcn_s_sleep(1);
            if (wall_clock64() - t0 > L.timeout_ticks) {
                __hip_atomic_fetch_or(L.err, 1 << t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (!L.loopback)
                    for (int p = 0; p < L.size; ++p)
                        if (p != L.rank) __hip_atomic_fetch_or(L.spf[p] - (L.spf_local - L.err), 1 << (8 + L.rank), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                dead = true;
                break;
            }
            dead = __hip_atomic_load(L.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
        }
        if (dead) *alive_sh = 0;
    }
    __syncthreads();
    return *alive_sh != 0;
}

// the reduced pending branch of 8 consecutive columns of local row lr: sum of the ranks' fp32 partials in rank order (+ bias), rounded
// once to bf16 -- exactly tp_allreduce_kernel's phase 2 (bd_comm.hip), so the sequence-parallel and the all-reduce forms agree bit for
// bit.  `own` = this rank's partial of the row (already loaded); the peers' copies sit in the local staging area [src][lr][N].
BD_DEV void sp_reduce8(const BdSpLink& L, const __amdgpu_buffer_rsrc_t stage, int rows_local, int N, int lr, int d0, const u32x4 own0, const u32x4 own1,
                       const bf16_t* bias, float* o) {
    u32x4 v[8][2];
#pragma unroll
    for (int p = 0; p < 8; ++p) {                              // every staged copy in flight before the first add
        if (p >= L.size) break;
        if (p == L.rank) { v[p][0] = own0; v[p][1] = own1; }
        else {
            const unsigned off = (unsigned)((((size_t)p * rows_local + lr) * N + d0) * 4);
            v[p][0] = __builtin_amdgcn_raw_buffer_load_b128(stage, off, 0, BD_SYS_AUX);
            v[p][1] = __builtin_amdgcn_raw_buffer_load_b128(stage, off + 16, 0, BD_SYS_AUX);
        }
    }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) {                              // rank order: identical bits whoever reduces
        if (p >= L.size) break;
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j] += __uint_as_float(v[p][0][j]); acc[4 + j] += __uint_as_float(v[p][1][j]); }
    }
    if (bias) {
        float b[8];
        ld_bf16x8(bias + d0, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += b[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = bfr(acc[j]);
}

// ------------------------------------------------------------------------------------------------
// x (+= pending branch * gate) ; h = LN(x)*(1+scale)+shift for THIS RANK'S rows; h rows to every rank      flow_head:242-252
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MAX_ROW_THREADS) void ln_mod_sp_kernel(LnModSpArgs a) {
    __shared__ float red[32];
    __shared__ int alive_sh;
    BD_KSTAMP(a.ln.stamp, 0);
    const BdSpLink& L = a.L;
    const int lr = blockIdx.x, d0 = threadIdx.x * 8, D = a.ln.D;
    const int m = sp_row(L.rank, L.size, lr);
    const bool active = d0 < D;
    const bf16_t* ada = (const bf16_t*)a.ln.ada + (size_t)m * a.ln.ada_ld;
    float x[8], w[8], b[8];
    u32x4 xr = {0, 0, 0, 0}, gr = xr, scr = xr, sfr = xr, own0 = xr, own1 = xr;
    if (active) {                                              // everything that does not depend on the peers is in flight before the wait
        xr = ld_raw8((const bf16_t*)a.ln.X + (size_t)m * D + d0);
        scr = ld_raw8(ada + a.ln.scale_off + d0);
        sfr = ld_raw8(ada + a.ln.shift_off + d0);
        if (a.ln.ln_w) { ld_f32x8(a.ln.ln_w + d0, w); ld_f32x8(a.ln.ln_b + d0, b); }
        if (a.part) {
            gr = ld_raw8(ada + a.ln.gate_off + d0);
            const u32x4* src = reinterpret_cast<const u32x4*>(a.part + (size_t)m * D + d0);
            own0 = src[0]; own1 = src[1];
        }
    }
#ifdef BD_GEMM_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BD_KSTAMP(a.ln.stamp, 1);
#endif
    if (a.part) {
        const int e = sp_epoch(L, a.seq_p);
        if (blockIdx.x == 0 && a.signal_p) sp_signal_p(L, e);
        if (!sp_wait_p(L, e, &alive_sh)) return;               // a dead exchange pushes nothing further; the host check raises on every rank
    }
    BD_KSTAMP(a.ln.stamp, 2);
    if (active) {
        unpack8(xr, x);
        if (a.part) {
            float o[8], g[8];
            sp_reduce8(L, sp_rsrc(L.stage[L.rank], L.stage_bytes), a.rows_local, D, lr, d0, own0, own1, (const bf16_t*)a.bias, o);
            unpack8(gr, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = bfr(x[j] + bfr(o[j] * g[j]));       // x = bf16(x + bf16(bf16(branch) * gate))
            *reinterpret_cast<u32x4*>((bf16_t*)a.ln.X + (size_t)m * D + d0) = pack8(x);
        }
    }
#ifdef BD_GEMM_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    BD_KSTAMP(a.ln.stamp, 3);
#endif
    float mean, rstd;
    ln_stats(x, active, D, a.ln.eps, red, mean, rstd);
    BD_KSTAMP(a.ln.stamp, 4);
    if (active) {
        float h[8], sc[8], sf[8];
        unpack8(scr, sc);
        unpack8(sfr, sf);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float ln = (x[j] - mean) * rstd;
            if (a.ln.ln_w) ln = ln * w[j] + b[j];
            h[j] = fadd(fmul(ln, bfr(1.0f + sc[j])), sf[j]);   // fp32 * bf16 + bf16 -> fp32, separate ops (ln_mod_kernel's arithmetic)
        }
        const u32x4 hv = pack8(h);                             // cast by the next Linear
        const unsigned off = (unsigned)(afrag_off(m, d0, a.ln.RB) * 2);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q < L.size) __builtin_amdgcn_raw_buffer_store_b128(hv, sp_rsrc(L.hbuf[q], L.hbuf_bytes), off, 0, BD_SYS_AUX);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // this row is at every destination ...
    BD_KSTAMP(a.ln.stamp, 5);
    __syncthreads();
    const int t = bd_spread_lane(L.size);
    if (t >= 0) {                                              // ... then its flag, on every rank (this one included), one store per wave
        const int e = sp_epoch(L, a.seq_h);
        int* const dst = L.loopback ? L.spf_local + BD_SP_H + sp_row(t, L.size, lr) : L.spf[t] + BD_SP_H + m;
        __hip_atomic_store(dst, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    BD_KSTAMP_END(a.ln.stamp);
}
int bdk_ln_mod_sp(const LnModSpArgs& a, hipStream_t st) {
    const int t = row_threads(a.ln.D);
    if (t < 0 || a.ln.D % 8 || a.ln.a8_scale || a.rows_local < 1 || a.rows_local % 8 || a.ln.M > BD_SP_MAXROWS) return -2;
#ifdef BD_GEMM_STAMP
    LnModSpArgs a2 = a; a2.ln.stamp = bdk_stamp_next("ln_mod_sp", a.rows_local);
    BD_LAUNCH(ln_mod_sp_kernel, dim3(a.rows_local), dim3(t), 0, st, a2);
    return bd_launch_status();
#endif
    BD_LAUNCH(ln_mod_sp_kernel, dim3(a.rows_local), dim3(t), 0, st, a);
    return bd_launch_status();
}

// ------------------------------------------------------------------------------------------------
// final layer + sampler step for this rank's patch positions (bd_rows.hip head_final_kernel with the pending w2 branch taken
// from the peers' pushes).  One workgroup per OWNED (image, patch position): its cond row and (CFG) its uncond row.
//   flow_head:169-173,342 ; sampling_x.py:77-95 (+ :6-41) ; t2i_pipeline.py:248 (sign)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(640) void head_final_sp_kernel(HeadFinalSpArgs s) {
    __shared__ float red[32];
    __shared__ float wsum[16][64];
    __shared__ float xh[64];
    __shared__ float xnext[32];
    __shared__ float xfin[32];
    __shared__ int alive_sh;
    const HeadFinalArgs& a = s.f;
    const BdSpLink& L = s.L;
    const int lb = blockIdx.x, d0 = threadIdx.x * 8, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwav = blockDim.x >> 6;
    const bool active = d0 < a.D;
    const bool two = a.sc.cfg_mult == 2;                         // block-uniform
    const bf16_t* ada = (const bf16_t*)a.ada;
    const int bp = sp_row(L.rank, L.size, lb);                   // the cond row's index = the patch position; local rows lb / bp_local + lb
    const int m0 = bp, m1 = a.BP + bp;
    const int lr0 = lb, lr1 = s.bp_local + lb;
    float x0[8], x1[8], h0[8], h1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { x0[j] = x1[j] = h0[j] = h1[j] = 0.f; }
    constexpr int PRE = 8;
    u32x4 z4 = {0, 0, 0, 0};
    u32x4 sc0r = z4, sf0r = z4, sc1r = z4, sf1r = z4, wpre[PRE], xr0 = z4, xr1 = z4, g0r = z4, g1r = z4, o00 = z4, o01 = z4, o10 = z4, o11 = z4;
    if (active) {
        sc0r = ld_raw8(ada + (size_t)m0 * a.ada_ld + a.scale_off + d0);
        sf0r = ld_raw8(ada + (size_t)m0 * a.ada_ld + a.shift_off + d0);
        g0r = ld_raw8(ada + (size_t)m0 * a.ada_ld + a.gate_off + d0);
        xr0 = ld_raw8((const bf16_t*)a.X + (size_t)m0 * a.D + d0);
        { const u32x4* src = reinterpret_cast<const u32x4*>(s.part + (size_t)m0 * a.D + d0); o00 = src[0]; o01 = src[1]; }
        if (two) {
            sc1r = ld_raw8(ada + (size_t)m1 * a.ada_ld + a.scale_off + d0);
            sf1r = ld_raw8(ada + (size_t)m1 * a.ada_ld + a.shift_off + d0);
            g1r = ld_raw8(ada + (size_t)m1 * a.ada_ld + a.gate_off + d0);
            xr1 = ld_raw8((const bf16_t*)a.X + (size_t)m1 * a.D + d0);
            const u32x4* src = reinterpret_cast<const u32x4*>(s.part + (size_t)m1 * a.D + d0); o10 = src[0]; o11 = src[1];
        }
#pragma unroll
        for (int c = 0; c < PRE; ++c) wpre[c] = ld_raw8((const bf16_t*)a.lin_w + (size_t)min(c, a.C - 1) * a.D + d0);
    }
    const int e = sp_epoch(L, s.seq_p);
    if (blockIdx.x == 0 && s.signal_p) sp_signal_p(L, e);
    if (!sp_wait_p(L, e, &alive_sh)) return;
    if (active) {
        const __amdgpu_buffer_rsrc_t stage = sp_rsrc(L.stage[L.rank], L.stage_bytes);
        const int rows_local = s.bp_local * a.sc.cfg_mult;
        float o[8], g[8];
        unpack8(xr0, x0); unpack8(g0r, g);
        sp_reduce8(L, stage, rows_local, a.D, lr0, d0, o00, o01, (const bf16_t*)s.bias, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) x0[j] = bfr(x0[j] + bfr(o[j] * g[j]));
        if (two) {
            unpack8(xr1, x1); unpack8(g1r, g);
            sp_reduce8(L, stage, rows_local, a.D, lr1, d0, o10, o11, (const bf16_t*)s.bias, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) x1[j] = bfr(x1[j] + bfr(o[j] * g[j]));
        }
    }
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0 += x0[j]; s1 += x1[j]; }
    block_sum2(s0, s1, red);
    const float mean0 = s0 / (float)a.D, mean1 = s1 / (float)a.D;
    float v0 = 0.f, v1 = 0.f;
    if (active) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float c0 = x0[j] - mean0, c1 = x1[j] - mean1; v0 += c0 * c0; v1 += c1 * c1; }
    }
    block_sum2(v0, v1, red);
    const float rstd0 = rsqrtf(v0 / (float)a.D + a.eps_ln), rstd1 = rsqrtf(v1 / (float)a.D + a.eps_ln);
    if (active) {
        float sc[8], sf[8];
        unpack8(sc0r, sc); unpack8(sf0r, sf);
#pragma unroll
        for (int j = 0; j < 8; ++j) h0[j] = bfr(fadd(fmul((x0[j] - mean0) * rstd0, bfr(1.0f + sc[j])), sf[j]));
        if (two) {
            unpack8(sc1r, sc); unpack8(sf1r, sf);
#pragma unroll
            for (int j = 0; j < 8; ++j) h1[j] = bfr(fadd(fmul((x1[j] - mean1) * rstd1, bfr(1.0f + sc[j])), sf[j]));
        }
    }
    float pv[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) pv[i] = 0.f;
    if (active) {
#pragma unroll
        for (int c0 = 0; c0 < 32; c0 += PRE) {
            if (c0 < a.C) {
                u32x4 wr[PRE];
#pragma unroll
                for (int c = 0; c < PRE; ++c)
                    wr[c] = (c0 == 0) ? wpre[c] : ld_raw8((const bf16_t*)a.lin_w + (size_t)min(c0 + c, a.C - 1) * a.D + d0);
#pragma unroll
                for (int c = 0; c < PRE; ++c) {
                    float w[8];
                    unpack8(wr[c], w);
                    float p0 = 0.f, p1 = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { p0 += h0[j] * w[j]; p1 += h1[j] * w[j]; }
                    pv[c0 + c] = p0; pv[32 + c0 + c] = p1;
                }
            }
        }
    }
#pragma unroll
    for (int half = 32; half >= 1; half >>= 1) {
        const bool up = (lane & half) != 0;
#pragma unroll
        for (int i = 0; i < half; ++i) {
            const float mine = up ? pv[i + half] : pv[i];
            const float other = up ? pv[i] : pv[i + half];
            pv[i] = mine + __shfl_xor(other, half);
        }
    }
    wsum[wave][lane] = pv[0];
    __syncthreads();
    if (threadIdx.x < 64) {
        float tot = 0.f;
        for (int w = 0; w < nwav; ++w) tot += wsum[w][threadIdx.x];
        const int r = threadIdx.x >> 5, c = threadIdx.x & 31;
        if (c < a.C && r < a.sc.cfg_mult) {
            const float o = bfr(tot + bf2f(((const bf16_t*)a.lin_b)[c]));
            float xv = o;
            if (a.sigmoid) {
                const float sg = bfr(1.0f / (1.0f + expf(-o)));
                xv = bfr(fsub(bfr(2.0f * sg), 1.0f));
            }
            xh[threadIdx.x] = xv;
            if (a.xhat_out) a.xhat_out[(size_t)(r * a.BP + bp) * a.C + c] = xv;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < a.C) {
        const int c = threadIdx.x;
        const SamplerScalars& q = a.sc;
        const size_t idx = (size_t)bp * a.C + c;
        const float x = a.xt[idx];
        float v = fdiv(fsub(xh[c], x), q.den);
        if (q.cfg_mult == 2) {
            const float vu = fdiv(fsub(xh[32 + c], x), q.den);
            const float cfg = a.cfg_table ? a.cfg_table[a.state->step] : q.cfg;
            v = fadd(vu, fmul(cfg, fsub(v, vu)));
        }
        float xn;
        if (!q.is_final) {
            const float score = fdiv(fsub(fmul(q.t, v), x), q.var);
            const float drift = fadd(v, fmul(q.omt, score));
            const float* eps = a.noise + (size_t)a.state->step * a.noise_step_stride + (size_t)(a.eval_index + 1) * a.BP * a.C;
            xn = fadd(fadd(x, fmul(drift, q.dt)), fmul(q.noise_scale, eps[idx]));
        } else {
            xn = fadd(x, fmul(v, q.dt));
            xfin[c] = xn;                                        // the finished latent row: every rank gets it (below), tok_finish binarises
        }
        a.xt[idx] = xn;
        xnext[c] = bfr(xn);
    }
    if (a.X_next && !a.sc.is_final) {
        __syncthreads();
        if (active) {
            float x0n[8], b[8];
            ld_bf16x8((const bf16_t*)a.in_b + d0, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) x0n[j] = small_dot(xnext, (const bf16_t*)a.in_w + (size_t)(d0 + j) * a.C, a.C) + b[j];
            const u32x4 qv = pack8(x0n);
            *reinterpret_cast<u32x4*>((bf16_t*)a.X_next + (size_t)m0 * a.D + d0) = qv;
            if (two) *reinterpret_cast<u32x4*>((bf16_t*)a.X_next + (size_t)m1 * a.D + d0) = qv;
        }
    }
    if (a.sc.is_final && s.seq_f) {                              // block-uniform: the sampled latent row of this patch position to every rank
        __syncthreads();
        const int t = threadIdx.x;
        if (t < (a.C + 3) / 4) {
            const u32x4 v = {__float_as_uint(xfin[4 * t]), __float_as_uint(4 * t + 1 < a.C ? xfin[4 * t + 1] : 0.f),
                             __float_as_uint(4 * t + 2 < a.C ? xfin[4 * t + 2] : 0.f), __float_as_uint(4 * t + 3 < a.C ? xfin[4 * t + 3] : 0.f)};
            const unsigned off = (unsigned)(((size_t)bp * 32 + 4 * t) * 4);      // rows of 32 floats whatever C is
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < L.size) __builtin_amdgcn_raw_buffer_store_b128(v, sp_rsrc(L.aux[q], L.aux_bytes), off, 0, BD_SYS_AUX);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int tq = bd_spread_lane(L.size);
        if (tq >= 0) {
            const int ef = sp_epoch(L, s.seq_f);
            int* const dst = L.loopback ? L.spf_local + BD_SP_H + sp_row(tq, L.size, lb) : L.spf[tq] + BD_SP_H + bp;
            __hip_atomic_store(dst, ef, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
int bdk_head_final_sp(const HeadFinalSpArgs& a, hipStream_t st) {
    const int t = row_threads(a.f.D);
    if (t < 0 || t > 640 || a.f.D % 8 || a.f.C > 32 || a.bp_local < 1 || a.bp_local % 8 || !a.part || a.f.BP * 32 * 4 > a.L.aux_bytes) return -2;
    BD_LAUNCH(head_final_sp_kernel, dim3(a.bp_local), dim3(t), 0, st, a);
    return bd_launch_status();
}

// ------------------------------------------------------------------------------------------------
// after the last evaluation of an AR step: the sampled latent of EVERY patch position (pushed by its owner) -> pred, sign tokens
// (t2i_pipeline.py:248), the latent state -- on every rank, so that the projector / LLM phase stays replicated
// ------------------------------------------------------------------------------------------------
__global__ void tok_finish_kernel(TokFinishArgs a) {
    const BdSpLink& L = a.L;
    const int bp = blockIdx.x, c = threadIdx.x;
    __shared__ int ok;
    if (c == 0) {
        const int e = sp_epoch(L, a.seq_f);
        const int* f = L.spf_local + BD_SP_H + bp;
        const long long t0 = wall_clock64();
        int good = 1;
        while (bd_epoch_before(__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM), e)) {
            if (__hip_atomic_load(L.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) { good = 0; break; }
            if (wall_clock64() - t0 > L.timeout_ticks) { __hip_atomic_fetch_or(L.err, 1 << 17, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); good = 0; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        ok = good;
    }
    __syncthreads();
    if (!ok || c >= a.C) return;
    const float xn = __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned*>(L.aux[L.rank]) + (size_t)bp * 32 + c, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_SYSTEM));
    const size_t idx = (size_t)bp * a.C + c;
    a.xt[idx] = xn;
    if (a.pred_out) a.pred_out[idx] = xn;
    const float sg = (xn > 0.f) ? 1.f : ((xn < 0.f) ? -1.f : xn);                    // torch.sign (0 -> 0)
    if (a.tok_cur)
        for (int r = 0; r < a.tok_branches; ++r) a.tok_cur[(size_t)r * a.BP * a.C + idx] = sg;
    if (a.tok_all) {
        const int b = bp / a.P, pp = bp % a.P;
        a.tok_all[((size_t)b * a.T + (size_t)a.state->step * a.P + pp) * a.C + c] = sg;
    }
}
int bdk_tok_finish(const TokFinishArgs& a, hipStream_t st) {
    if (a.C > 32 || a.BP > BD_SP_MAXROWS) return -2;
    BD_LAUNCH(tok_finish_kernel, dim3(a.BP), dim3(64), 0, st, a);
    return bd_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Qwen3 decode step, sequence-parallel (round 6): R (+= bf16(sum of the ranks' o_proj / down_proj partials)) and
// RMSNorm for the rows THIS RANK owns, operand rows to every rank                         HF modeling_qwen3.py:59-64, 294-323
// The arithmetic is rms_kernel's (bd_rows.hip) on the value tp_allreduce_kernel would have produced: partials summed in rank order,
// rounded once to bf16, added to the fp32 residual -- the two forms agree bit for bit.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(MAX_ROW_THREADS) void rms_sp_kernel(RmsSpArgs a) {
    __shared__ float red[32];
    __shared__ int alive_sh;
    const BdSpLink& L = a.L;
    const int lr = blockIdx.x, d0 = threadIdx.x * 8, D = a.r.D;
    const int m = sp_row(L.rank, L.size, lr);
    const bool active = d0 < D;
    float r[8];
    u32x4 wr = {0, 0, 0, 0}, own0 = wr, own1 = wr;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = 0.f;
    if (active) {                                              // everything local is in flight before the wait
        wr = ld_raw8((const bf16_t*)a.r.w + d0);
        ld_f32x8(a.r.R + (size_t)m * D + d0, r);
        if (a.part) {
            const u32x4* src = reinterpret_cast<const u32x4*>(a.part + (size_t)m * D + d0);
            own0 = src[0]; own1 = src[1];
        }
    }
    if (a.part) {
        const int e = sp_epoch(L, a.seq_p);
        if (blockIdx.x == 0 && a.signal_p) sp_signal_p(L, e);
        if (!sp_wait_p(L, e, &alive_sh)) return;
    }
    float ss = 0.f;
    if (active) {
        if (a.part) {
            float o[8];
            sp_reduce8(L, sp_rsrc(L.stage[L.rank], L.stage_bytes), a.rows_local, D, lr, d0, own0, own1, nullptr, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = r[j] + o[j];            // fp32 residual + bf16 branch (decode; the prefill keeps the all-reduce form)
            st_f32x8(a.r.R + (size_t)m * D + d0, r);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += r[j] * r[j];
    }
    const float rs = rsqrtf(block_sum(ss, red) / (float)D + a.r.eps);
    if (active) {
        float w[8], n[8];
        unpack8(wr, w);
#pragma unroll
        for (int j = 0; j < 8; ++j) n[j] = fmul(w[j], fmul(r[j], rs));
        if (a.final_rows) {                                            // fp32 rows, row-major, into every rank's landing buffer
            const u32x4 v0 = {__float_as_uint(n[0]), __float_as_uint(n[1]), __float_as_uint(n[2]), __float_as_uint(n[3])};
            const u32x4 v1 = {__float_as_uint(n[4]), __float_as_uint(n[5]), __float_as_uint(n[6]), __float_as_uint(n[7])};
            const unsigned off = (unsigned)(a.final_off + ((size_t)m * D + d0) * 4);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < L.size) {
                    __builtin_amdgcn_raw_buffer_store_b128(v0, sp_rsrc(L.hbuf[q], L.hbuf_bytes), off, 0, BD_SYS_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128(v1, sp_rsrc(L.hbuf[q], L.hbuf_bytes), off + 16, 0, BD_SYS_AUX);
                }
        } else {
            const u32x4 hv = pack8(n);                                 // cast by the next Linear
            const unsigned off = (unsigned)(afrag_off(m, d0, a.r.RB) * 2);
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < L.size) __builtin_amdgcn_raw_buffer_store_b128(hv, sp_rsrc(L.hbuf[q], L.hbuf_bytes), off, 0, BD_SYS_AUX);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int t = bd_spread_lane(L.size);
    if (t >= 0) {
        const int e = sp_epoch(L, a.seq_h);
        int* const dst = L.loopback ? L.spf_local + BD_SP_H + sp_row(t, L.size, lr) : L.spf[t] + BD_SP_H + m;
        __hip_atomic_store(dst, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
int bdk_rms_sp(const RmsSpArgs& a, hipStream_t st) {
    const int t = row_threads(a.r.D);
    if (t < 0 || a.r.D % 8 || a.r.a8_scale || a.r.bf16_stream || a.rows_local < 1 || a.rows_local % 8 || a.r.M > BD_SP_MAXROWS) return -2;
    if ((a.final_rows ? a.final_off + (long long)a.r.M * a.r.D * 4 : (long long)a.r.M * a.r.D * 2) > a.L.hbuf_bytes) return -3;
    BD_LAUNCH(rms_sp_kernel, dim3(a.rows_local), dim3(t), 0, st, a);
    return bd_launch_status();
}

// every rank: the final-norm rows of ALL sequences have landed (per-row flags, the GEMM prologue's wait + invalidate) -> hidden state
// and the next patch's condition, exactly rms_kernel's tail
__global__ __launch_bounds__(MAX_ROW_THREADS) void sp_final_rows_kernel(SpFinalRowsArgs a) {
    const int m = blockIdx.x, d0 = threadIdx.x * 8;
    BdHWait w = a.w;
    w.flags = a.w.flags + m; w.n = 1;                              // this workgroup's row only
    gemm_hwait(w, threadIdx.x, blockDim.x);
    if (d0 >= a.D) return;
    float n[8];
    ld_f32x8(a.rows + (size_t)m * a.D + d0, n);
    if (a.hidden_out) st_f32x8(a.hidden_out + (size_t)m * a.D + d0, n);
    if (a.cond_frag) {
        float p[8];
        ld_f32x8(a.pos + ((size_t)a.state->step * a.P + (m % a.P)) * a.D + d0, p);
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = fadd(n[j], p[j]);
        *reinterpret_cast<u32x4*>((bf16_t*)a.cond_frag + afrag_off(m, d0, a.RB)) = pack8(p);
    }
}
int bdk_sp_final_rows(const SpFinalRowsArgs& a, hipStream_t st) {
    const int t = row_threads(a.D);
    if (t < 0 || a.D % 8 || a.M > BD_SP_MAXROWS || !a.w.flags) return -2;
    BD_LAUNCH(sp_final_rows_kernel, dim3(a.M), dim3(t), 0, st, a);
    return bd_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Construction-time self-test of THIS hand-off (the all-reduce self-test does not reach it).  What is specific to the
// sequence-parallel form is the operand landing buffer: ordinary CACHEABLE device memory that other GPUs write with sc0 sc1 stores and
// that this GPU re-reads through its L2 after one `buffer_inv sc0 sc1` (bd_hwait.h) -- one-GPU tests cannot show a stale line there.
// Round r: every rank pushes its OWN rows (the product's row ownership and fragment-major addresses) of a pattern that depends on
// (row, 16 B unit, r) into every rank's buffer and raises the row flags; 64 workgroups per rank then run the GEMM prologue's wait
// (the same device function) and compare every unit of every row.  The host runs several rounds back to back, each behind a barrier
// of the ranks (so that round r + 1 finds round r's lines warm in the consumers' caches -- the stale-line case), and keeps the
// all-reduce form if any rank counts a mismatch (tp.py TPComm._sp_self_test; Engine: "tp.seq" needs comm.sp_ok).
// ------------------------------------------------------------------------------------------------
struct SpTestArgs { BdSpLink L; BdHWait w; int seq, rows, D, RB, round; int* bad; };
BD_DEV u32x4 sp_test_pattern(int m, int u, int round) {
    const unsigned h = (unsigned)round * 0x9E3779B1u ^ (unsigned)m * 0x85EBCA6Bu ^ (unsigned)u * 0xC2B2AE35u;
    return (u32x4){h, h ^ 0x11111111u, h ^ 0x22222222u, h ^ 0x33333333u};
}
__global__ __launch_bounds__(256) void sp_test_push_kernel(SpTestArgs a) {
    const BdSpLink& L = a.L;
    const int lr = blockIdx.x, m = sp_row(L.rank, L.size, lr);
    for (int u = threadIdx.x; u < a.D / 8; u += 256) {
        const u32x4 v = sp_test_pattern(m, u, a.round);
        const unsigned off = (unsigned)(afrag_off(m, u * 8, a.RB) * 2);
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (q < L.size) __builtin_amdgcn_raw_buffer_store_b128(v, sp_rsrc(L.hbuf[q], L.hbuf_bytes), off, 0, BD_SYS_AUX);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int t = bd_spread_lane(L.size);
    if (t >= 0) {
        const int e = sp_epoch(L, a.seq);
        int* const dst = L.loopback ? L.spf_local + BD_SP_H + sp_row(t, L.size, lr) : L.spf[t] + BD_SP_H + m;
        __hip_atomic_store(dst, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
__global__ __launch_bounds__(256) void sp_test_check_kernel(SpTestArgs a) {
    if (a.w.flags) gemm_hwait(a.w, threadIdx.x, 256);           // (null: a one-workgroup wait kernel ran in front, "tune.sp_wait" = 0)
    const u32x4* h = reinterpret_cast<const u32x4*>(a.L.hbuf[a.L.rank]);
    const int upr = a.D / 8;
    int bad = 0;
    for (int i = threadIdx.x; i < a.rows * upr; i += 256) {      // EVERY workgroup reads every unit: each CU / XCD sees the whole operand
        const int m = i / upr, u = i % upr;
        const u32x4 got = h[afrag_off(m, u * 8, a.RB) / 8], want = sp_test_pattern(m, u, a.round);
        bad += (got[0] != want[0]) | (got[1] != want[1]) | (got[2] != want[2]) | (got[3] != want[3]);
    }
    if (bad) atomicAdd(a.bad, bad);
}
int bdk_sp_selftest(bd_comm* c, int round, int rows, int D, int wait_in_check, int* bad_dev, hipStream_t st) {
    SpTestArgs a;
    if (!bdk_sp_link(c, &a.L) || rows % 32 || rows > BD_SP_MAXROWS || (rows / 8) % a.L.size || D % 8 || (long long)rows * D * 2 > a.L.hbuf_bytes) return -2;
    if (bdk_sp_begin(c, st) != 0) return -1;
    a.seq = bdk_sp_next_seq(c);
    if (!bdk_sp_hwait(c, a.seq, rows, &a.w)) return -2;
    a.rows = rows; a.D = D; a.RB = rows / 32; a.round = round; a.bad = bad_dev;
    BD_LAUNCH(sp_test_push_kernel, dim3(rows / a.L.size), dim3(256), 0, st, a);
    if (bd_launch_status() != 0) return -1;
    if (!wait_in_check) {
        if (bdk_sp_wait_rows(c, a.seq, rows, st) != 0) return -1;
        a.w.flags = nullptr;
    }
    BD_LAUNCH(sp_test_check_kernel, dim3(64), dim3(256), 0, st, a);
    return bd_launch_status();
}
