// Row-wise kernels of the BitDance hot path (gfx950): one workgroup per activation row, ONE THREAD PER 8
// CONSECUTIVE CHANNELS.  Eight consecutive k of one row are exactly one lane's 16 bytes of an MFMA operand
// chunk (bd_common.h afrag_off), so every kernel here ends in a single 16 B store per thread that lays down the
// next GEMM's A operand; all inputs come in as 16 B loads (8 bf16, or 2 x float4 per split-K slab).  The
// kernels reduce the split-K slabs of the preceding GEMM in their prologue and apply exactly the bf16/fp32
// rounding points of the reference's autocast flow, so between two weight-streaming GEMMs there is one small
// launch and no standalone elementwise pass.
#include "bd_rowhelp.h"
#include <string>

// A/B switches (bd_set_gemm_option "rows.*"): registers-per-thread bound of ln_mod (waves per SIMD: 4 = one 640-thread workgroup per CU,
// 5 = two), thread cap of swiglu_rows (512 = three workgroups per CU, 1024 = one 960-thread workgroup)
static int g_ln_occ = 5, g_swiglu_t = 512;   // measured (profiles/r06_head_sweep_b4.log, same box): ln_occ 5 (83 registers, the affine fetched after the reductions) -0.6 % per evaluation at 128 and 2048 rows, +-0 at 512; swiglu_t 512 -0.2 % at 512 rows
int bdk_set_rows_option(const char* name, int v) {
    const std::string n(name);
    if (n == "rows.ln_occ" && (v == 4 || v == 5)) { g_ln_occ = v; return 0; }
    if (n == "rows.swiglu_t" && (v == 512 || v == 1024)) { g_swiglu_t = v; return 0; }
    return -1;
}

// scalar helpers (small kernels)
BD_DEV float slab_sum(const Partial& q, int row, int col) {
    if (q.S == 0) return bf2f(((const bf16_t*)q.p)[(size_t)row * q.N + col]);
    float a = 0.f;
    const float* p = q.p + (size_t)row * q.N + col;
    for (int s = 0; s < q.S; ++s) a += p[(size_t)s * q.Mpad * q.N];
    if (q.bias) a += bf2f(((const bf16_t*)q.bias)[col]);
    return a;
}
BD_DEV float slab_bf(const Partial& q, int row, int col) { return bfr(slab_sum(q, row, col)); }

// ------------------------------------------------------------------------------------------------
// bf16 row-major finalisation of a split-K Linear output (cond_embed, once per AR step)
// ------------------------------------------------------------------------------------------------
__global__ void finalize_rows_kernel(FinalizeRowsArgs a) {
    BD_KSTAMP(a.stamp, 0);
    const int m = blockIdx.x;
    for (int c = threadIdx.x; c < a.N / 8; c += blockDim.x) {
        float v[8];
        slab8(a.in, m, c * 8, v);
        *reinterpret_cast<u32x4*>((bf16_t*)a.out + (size_t)m * a.N + c * 8) = pack8(v);
    }
    BD_KSTAMP_END(a.stamp);
}
int bdk_finalize_rows(const FinalizeRowsArgs& a, hipStream_t st) {
    if (a.N % 8) return -2;
    BD_STAMPED(FinalizeRowsArgs, a, "finalize_rows", a.M);
    BD_LAUNCH(finalize_rows_kernel, dim3(a.M), dim3(256), 0, st, a_l);
    return bd_launch_status();
}

// ------------------------------------------------------------------------------------------------
// standalone per-row e4m3 quantisation into the A8 layout (tests; the row kernels do this in their epilogues)
// ------------------------------------------------------------------------------------------------
__global__ void quant_rows8_kernel(unsigned char* a8, float* ascale, const float* src, int M, int K, int RB) {
    __shared__ float red[32];
    const int m = blockIdx.x;
    float am = 0.f;
    for (int k = threadIdx.x; k < K; k += blockDim.x) am = fmaxf(am, fabsf(src[(size_t)m * K + k]));
    am = block_max(am, red);
    if (threadIdx.x == 0) ascale[m] = __fdiv_rn(am, 448.0f);
    const float inv = am > 0.f ? __fdiv_rn(448.0f, am) : 0.f;
    for (int k0 = threadIdx.x * 8; k0 < K; k0 += blockDim.x * 8) quant8_store(a8, m, k0, RB, src + (size_t)m * K + k0, inv);
}
int bdk_quant_rows8(void* a8, float* ascale, const float* src, int M, int K, int RB, hipStream_t st) {
    if (K % 64) return -2;
    BD_LAUNCH(quant_rows8_kernel, dim3(M), dim3(256), 0, st, (unsigned char*)a8, ascale, src, M, K, RB);
    return bd_launch_status();
}

// ------------------------------------------------------------------------------------------------
// head prologue:  y = silu(time_embed(t) + cond_embed(c)),  x0 = input_proj(x_t)     flow_head:326-330
// ------------------------------------------------------------------------------------------------
__global__ void head_prologue_kernel(HeadPrologueArgs a) {
    extern __shared__ float sh[];                    // C floats of the latent row (bf16-rounded)
    __shared__ float red8[32];
    const int m = blockIdx.x;
    const int src = m % a.BP;                        // cond / uncond rows share the latent (sampling_x.py:71)
    if (a.X) {
        for (int k = threadIdx.x; k < a.C; k += blockDim.x) sh[k] = bfr(a.xt[(size_t)src * a.C + k]);
        __syncthreads();
    }
    const int d0 = threadIdx.x * 8;
    const bool act_ = d0 < a.D;
    // either half may be switched off (null output): y depends on (t_i, cond) only and can run ahead of the chain on a
    // second stream; x0 depends on the latent the previous evaluation produced
    if (a.y_frag) {
        float ce[8], te[8], y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = 0.f;
        if (act_) {
            ld_bf16x8((const bf16_t*)a.cemb + (size_t)m * a.D + d0, ce);
            ld_bf16x8((const bf16_t*)a.temb + d0, te);
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = silu_f(bfr(te[j] + ce[j]));       // bf16 + bf16 -> bf16 ; silu -> bf16 (rounded by pack8)
        }
        if (a.a8_scale) {                                          // fp8 x fp8 adaLN GEMM: per-row e4m3 of the bf16 silu output
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = bfr(y[j]);
            const float inv = row_quant_scale(y, act_, red8, a.a8_scale, m);
            if (act_) quant8_store((unsigned char*)a.y_frag, m, d0, a.RB, y, inv);
        } else if (act_) {
            *reinterpret_cast<u32x4*>((bf16_t*)a.y_frag + afrag_off(m, d0, a.RB)) = pack8(y);
        }
    }
    if (!act_) return;
    if (a.X) {
        float x0[8], b[8];
        ld_bf16x8((const bf16_t*)a.in_b + d0, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) x0[j] = small_dot(sh, (const bf16_t*)a.in_w + (size_t)(d0 + j) * a.C, a.C) + b[j];
        *reinterpret_cast<u32x4*>((bf16_t*)a.X + (size_t)m * a.D + d0) = pack8(x0);
    }
}
int bdk_head_prologue(const HeadPrologueArgs& a, hipStream_t st) {
    const int t = row_threads(a.D);
    if (t < 0 || a.D % 8) return -2;
    BD_LAUNCH(head_prologue_kernel, dim3(a.M), dim3(t), a.C * sizeof(float), st, a);
    return bd_launch_status();
}

// ------------------------------------------------------------------------------------------------
// x += branch*gate (pending) ; h = LN(x)*(1+scale)+shift                            flow_head:242-252
// ------------------------------------------------------------------------------------------------
// x[8] of this thread with the pending gated branch applied:  x = bf16(x + bf16(bf16(branch) * gate))
BD_DEV void load_x_pending(float* x, const bf16_t* X, const Partial& pend, const bf16_t* ada, int ada_ld, int gate_off,
                           int m, int D, int d0) {
    ld_bf16x8(X + (size_t)m * D + d0, x);
    if (pend.p) {
        float o[8], g[8];
        slab8(pend, m, d0, o);
        ld_bf16x8(ada + (size_t)m * ada_ld + gate_off + d0, g);
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = bfr(x[j] + bfr(o[j] * g[j]));
    }
}

// h = LN(x) * bf16(1 + scale) + shift   (fp32; the caller rounds when the following Linear casts)
BD_DEV void modulate8(const float* x, float mean, float rstd, const float* lw, const float* lb, const bf16_t* ada_row,
                      int scale_off, int shift_off, int d0, float* h) {
    float sc[8], sf[8];
    ld_bf16x8(ada_row + scale_off + d0, sc);
    ld_bf16x8(ada_row + shift_off + d0, sf);
    float w[8], b[8];
    if (lw) { ld_f32x8(lw + d0, w); ld_f32x8(lb + d0, b); }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float ln = (x[j] - mean) * rstd;
        if (lw) ln = ln * w[j] + b[j];
        h[j] = fadd(fmul(ln, bfr(1.0f + sc[j])), sf[j]);           // fp32 * bf16 + bf16 -> fp32, separate ops
    }
}

// OCC = waves per SIMD the register allocation must admit (launch bound): 4 = the up-front loads of everything (98 registers, ONE 640-thread
// workgroup per CU); 5 = the LayerNorm affine fetched after the reductions (83 registers, two workgroups per CU; round 6).  Same-box A/B
// ("rows.ln_occ"): 5 is 0.6 % faster per evaluation at 128 and at 2048 rows and equal at 512 (the 512-row kernel runs two waves of
// workgroups either way: its rows are bound by the slab reads, not by residency) -- default 5.
template <int OCC>
__global__ __launch_bounds__(MAX_ROW_THREADS, OCC) void ln_mod_kernel(LnModArgs a) {
    __shared__ float red[32];
    BD_KSTAMP(a.stamp, 0);
    const int m = blockIdx.x, d0 = threadIdx.x * 8;
    const bool active = d0 < a.D;
    const bf16_t* ada = (const bf16_t*)a.ada + (size_t)m * a.ada_ld;
    float x[8], w[8], b[8];
    u32x4 xr, gr, scr, sfr;
    if (active) {
        // every load of the kernel is issued up front (the row is one latency chain otherwise: slabs, then X, then
        // gate, then -- after two block reductions -- scale, shift and the LN affine); unpacked where needed
        xr = ld_raw8((const bf16_t*)a.X + (size_t)m * a.D + d0);
        scr = ld_raw8(ada + a.scale_off + d0);
        sfr = ld_raw8(ada + a.shift_off + d0);
        if (OCC == 4 && a.ln_w) { ld_f32x8(a.ln_w + d0, w); ld_f32x8(a.ln_b + d0, b); }
        if (a.pend.p) {
            gr = ld_raw8(ada + a.gate_off + d0);
            float o[8], g[8];
            slab8(a.pend, m, d0, o);
            unpack8(xr, x);
            unpack8(gr, g);
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = bfr(x[j] + bfr(o[j] * g[j]));       // x = bf16(x + bf16(bf16(branch) * gate))
            *reinterpret_cast<u32x4*>((bf16_t*)a.X + (size_t)m * a.D + d0) = pack8(x);
        } else {
            unpack8(xr, x);
        }
    }
    float mean, rstd;
    ln_stats(x, active, a.D, a.eps, red, mean, rstd);
    if (!active && !a.a8_scale) return;
    float h[8], sc[8], sf[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = 0.f;
    if (active) {
        if (OCC != 4 && a.ln_w) { ld_f32x8(a.ln_w + d0, w); ld_f32x8(a.ln_b + d0, b); }   // (the two-per-CU form fetches the affine late: 16 registers fewer across the reductions; L2-hot)
        unpack8(scr, sc);
        unpack8(sfr, sf);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float ln = (x[j] - mean) * rstd;
            if (a.ln_w) ln = ln * w[j] + b[j];
            h[j] = fadd(fmul(ln, bfr(1.0f + sc[j])), sf[j]);           // fp32 * bf16 + bf16 -> fp32, separate ops
        }
    }
    if (a.a8_scale) {                                              // fp8 x fp8 GEMMs behind this row kernel: per-row e4m3 of the fp32 h
        const float inv = row_quant_scale(h, active, red, a.a8_scale, m);
        if (active) quant8_store((unsigned char*)a.h_frag, m, d0, a.RB, h, inv);
        BD_KSTAMP_END(a.stamp);
        return;
    }
    *reinterpret_cast<u32x4*>((bf16_t*)a.h_frag + afrag_off(m, d0, a.RB)) = pack8(h);   // cast by the next Linear
    BD_KSTAMP_END(a.stamp);
}
// Many rows of a narrow model (the ImageNet batch: 12 288 rows of D = 768): one workgroup per row is 12 288 two-wave workgroups with
// two LDS block reductions each, and the eight rows that share every 128 B line of the fragment-major output are written from
// eight workgroups on eight XCDs (eight partial lines per L2).  Here a WAVE owns a row -- statistics by cross-lane sums only, no
// LDS, no barrier -- and the eight waves of a workgroup own the eight consecutive rows of one output line group.  Same arithmetic
// per element as ln_mod_kernel; the LayerNorm sums are taken in a different order (fp32, last-bit differences in mean / rstd).
template <int NPASS>
__global__ __launch_bounds__(512) void ln_mod_rows_kernel(LnModArgs a) {
    const int lane = threadIdx.x & 63, m = blockIdx.x * 8 + (threadIdx.x >> 6);
    if (m >= a.M) return;
    const bf16_t* ada = (const bf16_t*)a.ada + (size_t)m * a.ada_ld;
    bf16_t* const Xr = (bf16_t*)a.X + (size_t)m * a.D;
    float x[NPASS][8];
    u32x4 scr[NPASS], sfr[NPASS];
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int d0 = (p * 64 + lane) * 8;
        if (d0 < a.D) {
            const u32x4 xr = ld_raw8(Xr + d0);
            scr[p] = ld_raw8(ada + a.scale_off + d0);
            sfr[p] = ld_raw8(ada + a.shift_off + d0);
            unpack8(xr, x[p]);
            if (a.pend.p) {
                float o[8], g[8];
                const u32x4 gr = ld_raw8(ada + a.gate_off + d0);
                slab8(a.pend, m, d0, o);
                unpack8(gr, g);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[p][j] = bfr(x[p][j] + bfr(o[j] * g[j]));
                *reinterpret_cast<u32x4*>(Xr + d0) = pack8(x[p]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += x[p][j];
        }
    }
    const float mean = wave_sum(s) / (float)a.D;
    float v = 0.f;
#pragma unroll
    for (int p = 0; p < NPASS; ++p)
        if ((p * 64 + lane) * 8 < a.D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float c = x[p][j] - mean; v += c * c; }
        }
    const float rstd = rsqrtf(wave_sum(v) / (float)a.D + a.eps);
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
        const int d0 = (p * 64 + lane) * 8;
        if (d0 < a.D) {
            float sc[8], sf[8], w[8], b[8], h[8];
            unpack8(scr[p], sc);
            unpack8(sfr[p], sf);
            if (a.ln_w) { ld_f32x8(a.ln_w + d0, w); ld_f32x8(a.ln_b + d0, b); }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float ln = (x[p][j] - mean) * rstd;
                if (a.ln_w) ln = ln * w[j] + b[j];
                h[j] = fadd(fmul(ln, bfr(1.0f + sc[j])), sf[j]);
            }
            *reinterpret_cast<u32x4*>((bf16_t*)a.h_frag + afrag_off(m, d0, a.RB)) = pack8(h);
        }
    }
}
int bdk_ln_mod(const LnModArgs& a, hipStream_t st) {
    const int t = row_threads(a.D);
    if (t < 0 || a.D % 8) return -2;
    if (a.wave_rows && !a.a8_scale && a.D <= 1024) {
        const dim3 grid((a.M + 7) / 8);
        if (a.D <= 512) BD_LAUNCH(ln_mod_rows_kernel<1>, grid, dim3(512), 0, st, a);
        else BD_LAUNCH(ln_mod_rows_kernel<2>, grid, dim3(512), 0, st, a);
        return bd_launch_status();
    }
    BD_STAMPED(LnModArgs, a, "ln_mod", a.M);
    if (g_ln_occ == 5) BD_LAUNCH(ln_mod_kernel<5>, dim3(a.M), dim3(t), 0, st, a_l);
    else BD_LAUNCH(ln_mod_kernel<4>, dim3(a.M), dim3(t), 0, st, a_l);
    return bd_launch_status();
}

// ------------------------------------------------------------------------------------------------
// final layer + sampler step.  One workgroup per (image, patch position): its cond row and (CFG) uncond row.
//   flow_head:169-173,342 ; sampling_x.py:77-95 (+ :6-41) ; t2i_pipeline.py:248 (sign)
// ------------------------------------------------------------------------------------------------
__global__ void head_y_all_kernel(HeadYAllArgs a) {
    __shared__ float red8[32];
    const int m = blockIdx.x, i = blockIdx.y, d0 = threadIdx.x * 8;
    const bool act_ = d0 < a.D;
    if (!act_ && !a.a8_scale) return;
    float ce[8], te[8], y[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) y[j] = 0.f;
    if (act_) {
        ld_bf16x8((const bf16_t*)a.cemb + (size_t)m * a.D + d0, ce);
        ld_bf16x8((const bf16_t*)a.temb + (size_t)i * a.D + d0, te);
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = silu_f(bfr(te[j] + ce[j]));           // bf16 + bf16 -> bf16 ; silu -> bf16 (rounded by pack8)
    }
    const int g = i / a.G, left = a.n_evals - g * a.G;              // evaluations in this group's matrix
    const int rbg = (left >= a.G || a.G == 1) ? a.RB * a.G : ((a.RB * left + 7) & ~7);
    bf16_t* const base = (bf16_t*)a.y_all + (size_t)g * a.G * a.Mpad * a.D;      // (fp8: the group's operand uses half of this span)
    const int row = m + (i % a.G) * a.Mpad;
    if (a.a8_scale) {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = bfr(y[j]);
        const float inv = row_quant_scale(y, act_, red8, a.a8_scale + (size_t)i * a.Mpad, m);   // scales [eval][Mpad]: a group's rows are contiguous
        if (act_) quant8_store((unsigned char*)base, row, d0, rbg, y, inv);
        return;
    }
    *reinterpret_cast<u32x4*>(base + afrag_off(row, d0, rbg)) = pack8(y);
}
int bdk_head_y_all(const HeadYAllArgs& a, hipStream_t st) {
    const int t = row_threads(a.D);
    if (t < 0 || a.D % 8) return -2;
    if (a.G > 1 && a.n_evals % a.G) {
        // the last group is short: its matrix keeps a rounded-up row-block count whose pad rows nobody writes below -- zero them
        // (and their fp8 row scales) here instead of relying on how the caller allocated the buffers
        const int g = a.n_evals / a.G;
        bf16_t* base = (bf16_t*)a.y_all + (size_t)g * a.G * a.Mpad * a.D;
        if (hipMemsetAsync(base, 0, (size_t)a.G * a.Mpad * a.D * sizeof(bf16_t), st) != hipSuccess) return -1;
        if (a.a8_scale && hipMemsetAsync(a.a8_scale + (size_t)g * a.G * a.Mpad, 0, (size_t)a.G * a.Mpad * sizeof(float), st) != hipSuccess) return -1;
    }
    BD_LAUNCH(head_y_all_kernel, dim3(a.M, a.n_evals), dim3(t), 0, st, a);
    return bd_launch_status();
}

__global__ __launch_bounds__(640) void head_final_kernel(HeadFinalArgs a) {
    __shared__ float red[32];
    __shared__ float wsum[16][64];
    __shared__ float xh[64];
    __shared__ float xnext[32];
    BD_KSTAMP(a.stamp, 0);
    const int bp = blockIdx.x, d0 = threadIdx.x * 8, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nwav = blockDim.x >> 6;
    const bool active = d0 < a.D;
    const bool two = a.sc.cfg_mult == 2;                         // block-uniform
    const bf16_t* ada = (const bf16_t*)a.ada;
    const int m0 = bp, m1 = a.BP + bp;
    // both rows' loads are issued together, their LayerNorm statistics reduced together; everything the kernel reads that does
    // not depend on the statistics (scale / shift of the final modulation, the first channels of the output Linear) is in
    // flight before the first block reduction -- the kernel is one latency chain, not a bandwidth problem
    float x0[8], x1[8], h0[8], h1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { x0[j] = x1[j] = h0[j] = h1[j] = 0.f; }
    constexpr int PRE = 8;                                     // output channels whose weights are loaded ahead
    u32x4 sc0r = {0, 0, 0, 0}, sf0r = sc0r, sc1r = sc0r, sf1r = sc0r, wpre[PRE];
    if (active) {
        sc0r = ld_raw8(ada + (size_t)m0 * a.ada_ld + a.scale_off + d0);
        sf0r = ld_raw8(ada + (size_t)m0 * a.ada_ld + a.shift_off + d0);
        if (two) {
            sc1r = ld_raw8(ada + (size_t)m1 * a.ada_ld + a.scale_off + d0);
            sf1r = ld_raw8(ada + (size_t)m1 * a.ada_ld + a.shift_off + d0);
        }
#pragma unroll
        for (int c = 0; c < PRE; ++c) wpre[c] = ld_raw8((const bf16_t*)a.lin_w + (size_t)min(c, a.C - 1) * a.D + d0);
        load_x_pending(x0, (const bf16_t*)a.X, a.pend, ada, a.ada_ld, a.gate_off, m0, a.D, d0);
        if (two) load_x_pending(x1, (const bf16_t*)a.X, a.pend, ada, a.ada_ld, a.gate_off, m1, a.D, d0);
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
        for (int j = 0; j < 8; ++j)                                // LN (no affine) * bf16(1 + scale) + shift, then the Linear's input cast
            h0[j] = bfr(fadd(fmul((x0[j] - mean0) * rstd0, bfr(1.0f + sc[j])), sf[j]));
        if (two) {
            unpack8(sc1r, sc); unpack8(sf1r, sf);
#pragma unroll
            for (int j = 0; j < 8; ++j) h1[j] = bfr(fadd(fmul((x1[j] - mean1) * rstd1, bfr(1.0f + sc[j])), sf[j]));
        }
    }
    // D -> C Linear: per channel one 16 B weight load serves both rows.  Every thread keeps its 2 x 32 partial dot products;
    // the 64 lanes of a wave then reduce all 64 of them TOGETHER by recursive halving (lane l ends up with the wave total of
    // value l: 63 cross-lane exchanges instead of 64 x 6), and each lane parks its total.
    float pv[64];
#pragma unroll
    for (int i = 0; i < 64; ++i) pv[i] = 0.f;
    if (active) {
#pragma unroll
        for (int c0 = 0; c0 < 32; c0 += PRE) {
            if (c0 < a.C) {                                        // block-uniform
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
    wsum[wave][lane] = pv[0];                                      // lane = row * 32 + channel (channels >= C: unused)
    __syncthreads();
    if (threadIdx.x < 64) {
        float tot = 0.f;
        for (int w = 0; w < nwav; ++w) tot += wsum[w][threadIdx.x];
        const int r = threadIdx.x >> 5, c = threadIdx.x & 31;
        if (c < a.C && r < a.sc.cfg_mult) {
            const float o = bfr(tot + bf2f(((const bf16_t*)a.lin_b)[c]));   // Linear output bf16
            float xv = o;
            if (a.sigmoid) {
                const float sg = bfr(1.0f / (1.0f + expf(-o)));             // sigmoid (bf16)
                xv = bfr(fsub(bfr(2.0f * sg), 1.0f));                       // 2*sigmoid - 1 (bf16 ops)
            }
            xh[threadIdx.x] = xv;
            if (a.xhat_out) a.xhat_out[(size_t)(r * a.BP + bp) * a.C + c] = xv;
        }
    }
    __syncthreads();
    if ((int)threadIdx.x < a.C) {
        const int c = threadIdx.x;
        const SamplerScalars& s = a.sc;
        const size_t idx = (size_t)bp * a.C + c;
        const float x = a.xt[idx];
        float v = fdiv(fsub(xh[c], x), s.den);                    // (x_hat - x) / clamp_min(1-t, .05)
        if (s.cfg_mult == 2) {
            const float vu = fdiv(fsub(xh[32 + c], x), s.den);
            const float cfg = a.cfg_table ? a.cfg_table[a.state->step] : s.cfg;
            v = fadd(vu, fmul(cfg, fsub(v, vu)));                  // v_u + cfg (v_c - v_u)
        }
        float xn;
        if (!s.is_final) {
            const float score = fdiv(fsub(fmul(s.t, v), x), s.var);
            const float drift = fadd(v, fmul(s.omt, score));
            const float* eps = a.noise + (size_t)a.state->step * a.noise_step_stride + (size_t)(a.eval_index + 1) * a.BP * a.C;
            xn = fadd(fadd(x, fmul(drift, s.dt)), fmul(s.noise_scale, eps[idx]));
        } else {
            xn = fadd(x, fmul(v, s.dt));
            if (a.pred_out) a.pred_out[idx] = xn;
            const float sg = (xn > 0.f) ? 1.f : ((xn < 0.f) ? -1.f : xn);                    // torch.sign (0 -> 0)
            if (a.tok_cur)
                for (int r = 0; r < a.tok_branches; ++r) a.tok_cur[(size_t)r * a.BP * a.C + idx] = sg;
            if (a.tok_all) {
                const int b = bp / a.P, pp = bp % a.P;
                a.tok_all[((size_t)b * a.T + (size_t)a.state->step * a.P + pp) * a.C + c] = sg;
            }
        }
        a.xt[idx] = xn;
        xnext[c] = bfr(xn);                                          // what input_proj's autocast cast sees
    }
    if (a.X_next && !a.sc.is_final) {                                // block-uniform
        __syncthreads();
        if (active) {
            float x0n[8], b[8];
            ld_bf16x8((const bf16_t*)a.in_b + d0, b);
#pragma unroll
            for (int j = 0; j < 8; ++j) x0n[j] = small_dot(xnext, (const bf16_t*)a.in_w + (size_t)(d0 + j) * a.C, a.C) + b[j];
            const u32x4 q = pack8(x0n);
            *reinterpret_cast<u32x4*>((bf16_t*)a.X_next + (size_t)m0 * a.D + d0) = q;       // cond / uncond rows share the latent
            if (two) *reinterpret_cast<u32x4*>((bf16_t*)a.X_next + (size_t)m1 * a.D + d0) = q;
        }
    }
    BD_KSTAMP_END(a.stamp);
}
int bdk_head_final(const HeadFinalArgs& a, hipStream_t st) {
    const int t = row_threads(a.D);
    if (t < 0 || t > 640 || a.D % 8 || a.C > 32) return -2;      // 64 accumulators/thread: register budget of 10 waves
    BD_STAMPED(HeadFinalArgs, a, "head_final", a.BP);
    BD_LAUNCH(head_final_kernel, dim3(a.BP), dim3(t), 0, st, a_l);
    return bd_launch_status();
}

__global__ void init_latent_kernel(InitLatentArgs a) {
    const float* src = a.noise + (size_t)a.state->step * a.noise_step_stride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) a.xt[i] = src[i];
}
int bdk_init_latent(const InitLatentArgs& a, hipStream_t st) {
    BD_LAUNCH(init_latent_kernel, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
    return bd_launch_status();
}

// ------------------------------------------------------------------------------------------------
// SwiGLU from split-K slabs (when the fused GEMM epilogue would leave the chip under-filled): act = silu(h1)*h2
// ------------------------------------------------------------------------------------------------
__global__ void swiglu_rows_kernel(SwigluArgs a) {
    BD_KSTAMP(a.stamp, 0);
    const int m = blockIdx.x;
    for (int c = threadIdx.x; c < a.F / 8; c += blockDim.x) {
        const int f0 = c * 8;
        const int cg = a.interleaved ? ((f0 >> 4) * 32 + (f0 & 15)) : f0;
        const int cu = a.interleaved ? cg + 16 : a.F + f0;
        float g[8], u[8], o[8];
        slab8(a.up, m, cg, g);
        slab8(a.up, m, cu, u);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = silu_bf(g[j]) * u[j];
        *reinterpret_cast<u32x4*>((bf16_t*)a.act_frag + afrag_off(m, f0, a.RB)) = pack8(o);
    }
    BD_KSTAMP_END(a.stamp);
}
int bdk_swiglu_rows(const SwigluArgs& a, hipStream_t st) {
    if (a.F % 8) return -2;
    // at most 512 threads: three 8-wave workgroups fit a CU at this kernel's 80 registers (960 threads = 15 waves was one per CU: the 512
    // workgroups of num_images = 4 ran as two waves of workgroups, profiles/r06_launch_anatomy_b4.log); the row loop covers the rest
    int t = ((a.F / 8 + 63) / 64) * 64;
    if (t > g_swiglu_t) t = g_swiglu_t;
    BD_STAMPED(SwigluArgs, a, "swiglu_rows", a.M);
    BD_LAUNCH(swiglu_rows_kernel, dim3(a.M), dim3(t), 0, st, a_l);
    return bd_launch_status();
}

// ------------------------------------------------------------------------------------------------
// projector fc1 + gelu(tanh)                                                   modeling/utils.py:16-19
// ------------------------------------------------------------------------------------------------
BD_DEV float gelu_tanh_f(float x) {            // torch GeluCUDAKernelImpl, approximate='tanh'
    const float kBeta = 0.7978845608028654f;   // sqrt(2/pi)
    const float inner = kBeta * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.0f + tanhf(inner));
}
__global__ void proj_fc1_kernel(ProjFc1Args a) {
    extern __shared__ float sh[];
    const int m = blockIdx.x;
    for (int k = threadIdx.x; k < a.C; k += blockDim.x) sh[k] = bfr(a.tok[(size_t)m * a.C + k]);
    __syncthreads();
    const int d0 = threadIdx.x * 8;
    if (d0 >= a.D) return;
    float b[8], o[8];
    ld_bf16x8((const bf16_t*)a.b + d0, b);
#pragma unroll
    for (int j = 0; j < 8; ++j)
        o[j] = gelu_tanh_f(bfr(small_dot(sh, (const bf16_t*)a.w + (size_t)(d0 + j) * a.C, a.C) + b[j]));
    *reinterpret_cast<u32x4*>((bf16_t*)a.h_frag + afrag_off(m, d0, a.RB)) = pack8(o);
}
int bdk_proj_fc1(const ProjFc1Args& a, hipStream_t st) {
    const int t = row_threads(a.D);
    if (t < 0 || a.D % 8) return -2;
    BD_LAUNCH(proj_fc1_kernel, dim3(a.BP), dim3(t), a.C * sizeof(float), st, a);
    return bd_launch_status();
}

// model_input = fc2 output (bf16) + 2-D pos-embed (fp32) -> fp32, same rows for every CFG branch
// (t2i_pipeline.py:249-253; both halves of curr_tokens are identical copies)
__global__ void embed_finalize_kernel(EmbedFinalizeArgs a) {
    const int m = blockIdx.x, d0 = threadIdx.x * 8;   // m: 0..BP-1
    if (d0 >= a.D) return;
    const float* pos = a.pos + ((size_t)a.state->step * a.P + (m % a.P)) * a.D + d0;
    float e[8], p[8];
    slab8(a.fc2, m, d0, e);
    ld_f32x8(pos, p);
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] += p[j];
    for (int br = 0; br < a.branches; ++br) st_f32x8(a.R + ((size_t)br * a.BP + m) * a.D + d0, e);
}
int bdk_embed_finalize(const EmbedFinalizeArgs& a, hipStream_t st) {
    const int t = row_threads(a.D);
    if (t < 0 || a.D % 8) return -2;
    BD_LAUNCH(embed_finalize_kernel, dim3(a.BP), dim3(t), 0, st, a);
    return bd_launch_status();
}

// ------------------------------------------------------------------------------------------------
// LLM: residual add of the pending branch + RMSNorm                             HF:59-64, 294-323
// ------------------------------------------------------------------------------------------------
__global__ void rms_kernel(RmsArgs a) {
    __shared__ float red[32];
    const int m = blockIdx.x, d0 = threadIdx.x * 8;
    const bool active = d0 < a.D;
    float r[8];
    float ss = 0.f;
    u32x4 wr;
    if (active) {
        wr = ld_raw8((const bf16_t*)a.w + d0);                     // in flight across the reduction
        ld_f32x8(a.R + (size_t)m * a.D + d0, r);
        if (a.pend.p) {
            float o[8];
            slab8(a.pend, m, d0, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = a.bf16_stream ? bfr(r[j] + o[j]) : r[j] + o[j];   // fp32 (decode) / bf16 (prefill) residual + bf16 branch
            st_f32x8(a.R + (size_t)m * a.D + d0, r);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += r[j] * r[j];
    }
    const float rs = rsqrtf(block_sum(ss, red) / (float)a.D + a.eps);
    if (!active && !a.a8_scale) return;
    float w[8], n[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) n[j] = 0.f;
    if (active) {
        unpack8(wr, w);
#pragma unroll
        for (int j = 0; j < 8; ++j)                                 // weight * (x * rsqrt(var+eps)).to(input dtype): fp32, or twice-rounded bf16
            n[j] = a.bf16_stream ? bfr(fmul(w[j], bfr(fmul(r[j], rs)))) : fmul(w[j], fmul(r[j], rs));
    }
    if (a.a8_scale && a.a_frag) {                                  // fp8 x fp8 q/k/v and gate/up GEMMs: per-row e4m3 of the normed row
        const float inv = row_quant_scale(n, active, red, a.a8_scale, m);
        if (active) quant8_store((unsigned char*)a.a_frag, m, d0, a.RB, n, inv);
    }
    if (!active) return;
    if (a.a_frag && !a.a8_scale) *reinterpret_cast<u32x4*>((bf16_t*)a.a_frag + afrag_off(m, d0, a.RB)) = pack8(n);   // cast by the next Linear
    if (a.hidden_out) st_f32x8(a.hidden_out + (size_t)m * a.D + d0, n);
    if (a.cond_frag) {                                             // cond = hidden + pos (t2i:244-245), cast by cond_embed
        float p[8];
        ld_f32x8(a.pos + ((size_t)a.state->step * a.P + (m % a.P)) * a.D + d0, p);
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = fadd(n[j], p[j]);
        *reinterpret_cast<u32x4*>((bf16_t*)a.cond_frag + afrag_off(m, d0, a.RB)) = pack8(p);
    }
}
int bdk_rms(const RmsArgs& a, hipStream_t st) {
    const int t = row_threads(a.D);
    if (t < 0 || a.D % 8) return -2;
    BD_LAUNCH(rms_kernel, dim3(a.M), dim3(t), 0, st, a);
    return bd_launch_status();
}

// ------------------------------------------------------------------------------------------------
// q/k head RMS-norm + RoPE + KV append.  One wave per (row, head slot); lane holds dims (l, l+64).
//   HF:59-64 (bf16 in -> bf16 out), :140-170 (fp32 cos/sin in decode => fp32 products), cache_utils update
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void qkv_post_kernel(QkvPostArgs a) {
    const int m = blockIdx.x;
    const int lane = threadIdx.x & 63;
    const int nslot = a.nh + 2 * a.nkv;
    const int seq = m / a.P, p = m % a.P;
    const int pos = a.state->kv_len[seq] + p;
    if (pos >= a.Lmax) return;                                     // a step past the end of the static cache (caller error): no out-of-range write
    const bf16_t* QW = (const bf16_t*)a.qn_w;
    const bf16_t* KW = (const bf16_t*)a.kn_w;
    for (int slot = blockIdx.y * 4 + (threadIdx.x >> 6); slot < nslot; slot += gridDim.y * 4) {
        const int col = slot * 128;
        float x0 = slab_bf(a.qkv, m, col + lane);
        float x1 = slab_bf(a.qkv, m, col + lane + 64);
        if (slot < a.nh + a.nkv) {
            const bf16_t* w = (slot < a.nh) ? QW : KW;
            const float var = wave_sum(x0 * x0 + x1 * x1) / 128.f;
            const float rs = rsqrtf(var + a.eps);
            x0 = bfr(bf2f(w[lane]) * bfr(x0 * rs));               // weight * normed.to(bf16)  (bf16*bf16 -> bf16)
            x1 = bfr(bf2f(w[lane + 64]) * bfr(x1 * rs));
            float c0 = a.cos[(size_t)pos * 128 + lane], s0 = a.sin[(size_t)pos * 128 + lane];
            float c1 = a.cos[(size_t)pos * 128 + lane + 64], s1 = a.sin[(size_t)pos * 128 + lane + 64];
            float y0, y1;
            if (a.rope_bf16) {                                     // bf16 hidden states: tables and every op in bf16
                c0 = bfr(c0); s0 = bfr(s0); c1 = bfr(c1); s1 = bfr(s1);
                y0 = bfr(bfr(x0 * c0) + bfr(-x1 * s0));
                y1 = bfr(bfr(x1 * c1) + bfr(x0 * s1));
            } else {
                y0 = fadd(fmul(x0, c0), fmul(-x1, s0));            // q*cos + rotate_half(q)*sin
                y1 = fadd(fmul(x1, c1), fmul(x0, s1));
            }
            x0 = y0; x1 = y1;
        }
        if (slot < a.nh) {
            bf16_t* q = (bf16_t*)a.q_out + (size_t)m * a.nh * 128 + slot * 128;
            q[lane] = f2bf(x0); q[lane + 64] = f2bf(x1);           // SDPA (autocast) casts q to bf16
        } else if (slot < a.nh + a.nkv) {
            const int kvh = slot - a.nh;
            bf16_t* k = (bf16_t*)a.k_cache + (((size_t)seq * a.nkv + kvh) * a.Lmax + pos) * 128;
            k[lane] = f2bf(x0); k[lane + 64] = f2bf(x1);
        } else {
            const int kvh = slot - a.nh - a.nkv;
            bf16_t* v = (bf16_t*)a.vt_cache + ((size_t)seq * a.nkv + kvh) * 128 * a.Lmax + pos;
            v[(size_t)lane * a.Lmax] = f2bf(x0); v[(size_t)(lane + 64) * a.Lmax] = f2bf(x1);
        }
    }
}
int bdk_qkv_post(const QkvPostArgs& a, hipStream_t st) {
    const int nslot = a.nh + 2 * a.nkv;
    BD_LAUNCH(qkv_post_kernel, dim3(a.M, (nslot + 3) / 4), dim3(256), 0, st, a);
    return bd_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Class-conditional ImageNet transformer rows (imagenet_gen/src): bf16 residual stream, fp32 norm weights
// ------------------------------------------------------------------------------------------------
__global__ void in_proj_fc1_kernel(InProjFc1Args a) {
    extern __shared__ float sh[];
    const int m = blockIdx.x;
    for (int k = threadIdx.x; k < a.C; k += blockDim.x) sh[k] = bfr(a.tok[(size_t)m * a.C + k]);
    __syncthreads();
    const int d0 = threadIdx.x * 8;
    if (d0 >= a.hid) return;
    const bf16_t* W = (const bf16_t*)a.w;
    float b1[8], b2[8], o[8];
    ld_bf16x8((const bf16_t*)a.b + d0, b1);
    ld_bf16x8((const bf16_t*)a.b + a.hid + d0, b2);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float h1 = bfr(small_dot(sh, W + (size_t)(d0 + j) * a.C, a.C) + b1[j]);            // Linear -> bf16
        const float h2 = bfr(small_dot(sh, W + (size_t)(a.hid + d0 + j) * a.C, a.C) + b2[j]);
        o[j] = silu_bf(h1) * h2;                                                                 // silu -> bf16; product rounded by pack8
    }
    *reinterpret_cast<u32x4*>((bf16_t*)a.act_frag + afrag_off(m, d0, a.RB)) = pack8(o);
}
int bdk_in_proj_fc1(const InProjFc1Args& a, hipStream_t st) {
    const int t = row_threads(a.hid);
    if (t < 0 || a.hid % 8) return -2;
    BD_LAUNCH(in_proj_fc1_kernel, dim3(a.rows), dim3(t), a.C * sizeof(float), st, a);
    return bd_launch_status();
}

__global__ void in_rms_kernel(InRmsArgs a) {
    __shared__ float red[32];
    const int m = blockIdx.x, d0 = threadIdx.x * 8;
    const bool active = d0 < a.D;
    float x[8];
    float ss = 0.f;
    if (active) {
        if (a.init_from_pend) {
            slab8(a.pend, m, d0, x);                               // proj_in output: bf16(sum + bias)
        } else {
            ld_f32x8(a.R + (size_t)m * a.D + d0, x);
            if (a.pend.p) {
                float o[8];
                slab8(a.pend, m, d0, o);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = a.f32_stream ? x[j] + o[j] : bfr(x[j] + o[j]);   // bf16 residual + bf16 branch -> bf16 (fp32 + bf16 -> fp32)
            }
        }
        if (!a.renorm_to_R && (a.pend.p || a.init_from_pend)) st_f32x8(a.R + (size_t)m * a.D + d0, x);
#pragma unroll
        for (int j = 0; j < 8; ++j) ss += x[j] * x[j];
    }
    const float rs = rsqrtf(block_sum(ss, red) / (float)a.D + a.eps);
    if (!active) return;
    float w[8], n[8];
    ld_f32x8(a.w + d0, w);
#pragma unroll
    for (int j = 0; j < 8; ++j) {                                  // rms_norm composite: one rounding to the input dtype (none for fp32)
        const float v = fmul(fmul(x[j], rs), w[j]);
        n[j] = a.f32_stream ? v : bfr(v);
    }
    if (a.renorm_to_R) st_f32x8(a.R + (size_t)m * a.D + d0, n);
    if (a.a_frag) *reinterpret_cast<u32x4*>((bf16_t*)a.a_frag + afrag_off(m, d0, a.RB)) = pack8(n);
    if (a.hidden_out) st_f32x8(a.hidden_out + (size_t)m * a.D + d0, n);
    if (a.cond_frag) {                                             // z = norm(x) + pos_for_diff (fp32), cast by cond_embed
        float p[8];
        ld_f32x8(a.pos + ((size_t)a.state->step * a.P + (m % a.P)) * a.D + d0, p);
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] = fadd(n[j], p[j]);
        *reinterpret_cast<u32x4*>((bf16_t*)a.cond_frag + afrag_off(m, d0, a.RB)) = pack8(p);
    }
}
int bdk_in_rms(const InRmsArgs& a, hipStream_t st) {
    const int t = row_threads(a.D);
    if (t < 0 || a.D % 8) return -2;
    BD_LAUNCH(in_rms_kernel, dim3(a.M), dim3(t), 0, st, a);
    return bd_launch_status();
}

// one wave per (row, q/k/v head slot of 64 dims): lane = dim; the RoPE partner of dim 2i is lane 2i+1
__global__ __launch_bounds__(256) void in_qkv_post_kernel(InQkvPostArgs a) {
    const int m = blockIdx.x, lane = threadIdx.x & 63;
    const int D = a.nh * 64, nslot = 3 * a.nh;
    const int seq = m / a.P, p = m % a.P;
    const int pos = a.state->kv_len[0] + p;                        // every sequence has the same length here
    if (pos >= a.Lmax) return;                                     // a step past the end of the static cache (caller error): no out-of-range write
    for (int slot = blockIdx.y * 4 + (threadIdx.x >> 6); slot < nslot; slot += gridDim.y * 4) {
        const int which = slot / a.nh, h = slot % a.nh;
        float x = slab_bf(a.qkv, m, which * D + h * 64 + lane);
        if (which < 2) {
            const float other = __shfl_xor(x, 1);
            const float c = a.rope[((size_t)pos * 32 + (lane >> 1)) * 2], s = a.rope[((size_t)pos * 32 + (lane >> 1)) * 2 + 1];
            const float y = (lane & 1) ? fadd(fmul(x, c), fmul(other, s))          // x1*cos + x0*sin
                                       : fsub(fmul(x, c), fmul(other, s));         // x0*cos - x1*sin
            x = bfr(y);                                                             // .type_as(x)
        }
        if (which == 0) {
            ((bf16_t*)a.q_out)[(size_t)m * D + h * 64 + lane] = f2bf(x * 0.125f);   // xq * head_dim**-0.5 (bf16, exact)
        } else {
            bf16_t* dst = (which == 1 ? a.k_cache : a.v_cache) + (((size_t)seq * a.nh + h) * a.Lmax + pos) * 64;
            dst[lane] = f2bf(x);
        }
    }
}
int bdk_in_qkv_post(const InQkvPostArgs& a, hipStream_t st) {
    BD_LAUNCH(in_qkv_post_kernel, dim3(a.M, (3 * a.nh + 3) / 4), dim3(256), 0, st, a);
    return bd_launch_status();
}

__global__ void step_advance_kernel(StepAdvanceArgs a) {
    if (threadIdx.x == 0) a.state->step += 1;
    if ((int)threadIdx.x < a.nseq) a.state->kv_len[threadIdx.x] += a.P;
}
int bdk_step_advance(const StepAdvanceArgs& a, hipStream_t st) {
    BD_LAUNCH(step_advance_kernel, dim3(1), dim3(64), 0, st, a);
    return bd_launch_status();
}

// ------------------------------------------------------------------------------------------------
// Group-wise lookup-free quantiser index math (imagenet_gen/src/gfq.py:217-239, :152-160): integer, bit exact.
//   quantise: idx[t][c] = sum_k (z[t][c*bits + k] > 0) << k ;   codes: code[t][c*bits + k] = bit k of idx ? +1 : -1
// ------------------------------------------------------------------------------------------------
__global__ void gfq_indices_kernel(const float* __restrict__ z, int* __restrict__ idx, int ntok, int ncb, int bits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntok * ncb) return;
    const float* p = z + (size_t)(i / ncb) * ncb * bits + (size_t)(i % ncb) * bits;
    int v = 0;
    for (int k = 0; k < bits; ++k) v |= (p[k] > 0.f) ? (1 << k) : 0;
    idx[i] = v;
}
__global__ void gfq_codes_kernel(const int* __restrict__ idx, float* __restrict__ code, int ntok, int ncb, int bits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntok * ncb * bits) return;
    const int k = i % bits, tc = i / bits;
    code[i] = ((idx[tc] >> k) & 1) ? 1.f : -1.f;
}
int bdk_gfq_indices(const float* z, int* idx, int ntok, int ncb, int bits, hipStream_t st) {
    if (bits < 1 || bits > 30) return -2;
    BD_LAUNCH(gfq_indices_kernel, dim3((ntok * ncb + 255) / 256), dim3(256), 0, st, z, idx, ntok, ncb, bits);
    return bd_launch_status();
}
int bdk_gfq_codes(const int* idx, float* code, int ntok, int ncb, int bits, hipStream_t st) {
    if (bits < 1 || bits > 30) return -2;
    BD_LAUNCH(gfq_codes_kernel, dim3((ntok * ncb * bits + 255) / 256), dim3(256), 0, st, idx, code, ntok, ncb, bits);
    return bd_launch_status();
}
