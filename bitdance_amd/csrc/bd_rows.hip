// Row-wise kernels of the BitDance hot path (gfx950): one workgroup per activation row.
// They reduce the split-K slabs of the preceding GEMM in their prologue, apply exactly the bf16/fp32
// rounding points of the reference's autocast flow, and emit the next GEMM's A operand directly in
// MFMA fragment-major order (bd_common.h) -- so between two weight-streaming GEMMs there is one small
// launch and no standalone elementwise pass.
#include "bd_common.h"
#include "bd_kernels.h"

#define ROW_THREADS 256

// sum_s P[s][row][col] (+ bias) -> fp32
BD_DEV float slab_sum(const Partial& q, int row, int col) {
    float a = 0.f;
    const float* p = q.p + (size_t)row * q.N + col;
    for (int s = 0; s < q.S; ++s) a += p[(size_t)s * q.Mpad * q.N];
    if (q.bias) a += bf2f(((const bf16_t*)q.bias)[col]);
    return a;
}
BD_DEV float slab_bf(const Partial& q, int row, int col) { return bfr(slab_sum(q, row, col)); }  // Linear output (bf16)

// dot of an LDS fp32 vector with one bf16 weight row, K small (latent channels); 16 B loads when K % 8 == 0
BD_DEV float small_dot(const float* x, const bf16_t* w, int K) {
    float acc = 0.f;
    if ((K & 7) == 0) {
        for (int k = 0; k < K; k += 8) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(w + k);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc += x[k + 2 * j] * bf2f((bf16_t)(v[j] & 0xffff));
                acc += x[k + 2 * j + 1] * bf2f((bf16_t)(v[j] >> 16));
            }
        }
    } else {
        for (int k = 0; k < K; ++k) acc += x[k] * bf2f(w[k]);
    }
    return acc;
}

// ------------------------------------------------------------------------------------------------
// head prologue:  y = silu(time_embed(t) + cond_embed(c)),  x0 = input_proj(x_t)     flow_head:326-330
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ROW_THREADS) void head_prologue_kernel(HeadPrologueArgs a) {
    extern __shared__ float sh[];                    // C floats of the latent row (bf16-rounded)
    const int m = blockIdx.x;
    const int src = m % a.BP;                        // cond / uncond rows share the latent (sampling_x.py:71)
    for (int k = threadIdx.x; k < a.C; k += blockDim.x) sh[k] = bfr(a.xt[(size_t)src * a.C + k]);
    __syncthreads();
    const bf16_t* temb = (const bf16_t*)a.temb;
    const bf16_t* inw = (const bf16_t*)a.in_w;
    const bf16_t* inb = (const bf16_t*)a.in_b;
    bf16_t* X = (bf16_t*)a.X;
    bf16_t* Y = (bf16_t*)a.y_frag;
    for (int d = threadIdx.x; d < a.D; d += blockDim.x) {
        const float ce = slab_bf(a.cond, m, d);
        const float s = bfr(bf2f(temb[d]) + ce);     // bf16 + bf16 -> bf16
        Y[afrag_off(m, d, a.RB)] = f2bf(silu_f(s));
        X[(size_t)m * a.D + d] = f2bf(small_dot(sh, inw + (size_t)d * a.C, a.C) + bf2f(inb[d]));
    }
}

int bdk_head_prologue(const HeadPrologueArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(head_prologue_kernel, dim3(a.M), dim3(ROW_THREADS), a.C * sizeof(float), st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------
// x += branch*gate (pending) ; h = LN(x)*(1+scale)+shift                            flow_head:242-252
// ------------------------------------------------------------------------------------------------
BD_DEV float ada_val(const Partial& ada, int row, int col) { return slab_bf(ada, row, col); }

// loads the row (applying a pending gated-branch update) into LDS as fp32; returns nothing
BD_DEV void load_row_with_pending(float* row, const bf16_t* X, bf16_t* Xw, const Partial& pend, const Partial& ada,
                                  int gate_off, int m, int D) {
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        float x = bf2f(X[(size_t)m * D + d]);
        if (pend.p) {
            const float o = slab_bf(pend, m, d);                 // wo / w2 output, bf16
            const float g = ada_val(ada, m, gate_off + d);
            const float hg = bfr(o * g);                          // h * gate   (bf16*bf16 -> bf16)
            x = bfr(x + hg);                                      // x + ...    (bf16+bf16 -> bf16)
            if (Xw) Xw[(size_t)m * D + d] = f2bf(x);
        }
        row[d] = x;
    }
}

BD_DEV void row_stats(const float* row, int D, float eps, float* red, float& mean, float& rstd) {
    float s = 0.f;
    for (int d = threadIdx.x; d < D; d += blockDim.x) s += row[d];
    mean = block_sum(s, red) / (float)D;
    float v = 0.f;
    for (int d = threadIdx.x; d < D; d += blockDim.x) { const float c = row[d] - mean; v += c * c; }
    const float var = block_sum(v, red) / (float)D;
    rstd = rsqrtf(var + eps);
}

__global__ __launch_bounds__(ROW_THREADS) void ln_mod_kernel(LnModArgs a) {
    extern __shared__ float sh[];
    float* row = sh;
    float* red = sh + a.D;
    const int m = blockIdx.x;
    load_row_with_pending(row, (const bf16_t*)a.X, (bf16_t*)a.X, a.pend, a.ada, a.gate_off, m, a.D);
    __syncthreads();
    float mean, rstd;
    row_stats(row, a.D, a.eps, red, mean, rstd);
    bf16_t* H = (bf16_t*)a.h_frag;
    for (int d = threadIdx.x; d < a.D; d += blockDim.x) {
        float ln = (row[d] - mean) * rstd;
        if (a.ln_w) ln = ln * a.ln_w[d] + a.ln_b[d];
        const float sc = ada_val(a.ada, m, a.scale_off + d);
        const float onep = bfr(1.0f + sc);                        // (1 + scale) is a bf16 tensor
        const float sft = ada_val(a.ada, m, a.shift_off + d);
        const float t = fadd(fmul(ln, onep), sft);                // fp32 * bf16 + bf16 -> fp32, separate ops
        H[afrag_off(m, d, a.RB)] = f2bf(t);                        // cast by the following Linear
    }
}

int bdk_ln_mod(const LnModArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(ln_mod_kernel, dim3(a.M), dim3(ROW_THREADS), (a.D + 32) * sizeof(float), st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------
// final layer + sampler step.  One workgroup per (image, patch position): its cond row and (CFG) uncond row.
//   flow_head:169-173,342 ; sampling_x.py:77-95 (+ :6-41) ; t2i_pipeline.py:248 (sign)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ROW_THREADS) void head_final_kernel(HeadFinalArgs a) {
    extern __shared__ float sh[];
    float* row = sh;                       // D
    float* red = sh + a.D;                 // 32
    float* part = red + 32;                // (ROW_THREADS/64) * C partial dot products
    float* xh = part + (ROW_THREADS / 64) * a.C;   // 2*C : x_hat of the cond / uncond row
    const int bp = blockIdx.x;
    const int nrows = a.sc.cfg_mult;
    const bf16_t* W = (const bf16_t*)a.lin_w;
    const bf16_t* LB = (const bf16_t*)a.lin_b;
    for (int r = 0; r < nrows; ++r) {
        const int m = r * a.BP + bp;
        __syncthreads();
        load_row_with_pending(row, (const bf16_t*)a.X, nullptr, a.pend, a.ada, a.gate_off, m, a.D);
        __syncthreads();
        float mean, rstd;
        row_stats(row, a.D, a.eps_ln, red, mean, rstd);
        __syncthreads();
        for (int d = threadIdx.x; d < a.D; d += blockDim.x) {
            const float ln = (row[d] - mean) * rstd;
            const float onep = bfr(1.0f + ada_val(a.ada, m, a.scale_off + d));
            const float sft = ada_val(a.ada, m, a.shift_off + d);
            row[d] = bfr(fadd(fmul(ln, onep), sft));              // Linear input, bf16
        }
        __syncthreads();
        // out[c] = sum_d row[d] * W[c][d]: wave w takes channels c = w, w+4, ...; lanes stride over d
        const int wv = threadIdx.x >> 6, ln_ = threadIdx.x & 63;
        for (int c = wv; c < a.C; c += ROW_THREADS / 64) {
            float acc = 0.f;
            const bf16_t* w = W + (size_t)c * a.D;
            for (int d = ln_ * 2; d < a.D; d += 128) {
                const unsigned pr = *reinterpret_cast<const unsigned*>(w + d);
                acc += row[d] * bf2f((bf16_t)(pr & 0xffff)) + row[d + 1] * bf2f((bf16_t)(pr >> 16));
            }
            acc = wave_sum(acc);
            if (ln_ == 0) {
                const float o = bfr(acc + bf2f(LB[c]));            // Linear output bf16
                const float sg = bfr(1.0f / (1.0f + expf(-o)));    // sigmoid (bf16)
                const float xv = bfr(fsub(bfr(2.0f * sg), 1.0f));  // 2*sigmoid - 1 (bf16 ops)
                xh[r * a.C + c] = xv;
                if (a.xhat_out) a.xhat_out[(size_t)m * a.C + c] = xv;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < a.C) {
        const int c = threadIdx.x;
        const SamplerScalars& s = a.sc;
        const size_t idx = (size_t)bp * a.C + c;
        const float x = a.xt[idx];
        float v = fdiv(fsub(xh[c], x), s.den);                    // (x_hat - x) / clamp_min(1-t, .05)
        if (s.cfg_mult == 2) {
            const float vu = fdiv(fsub(xh[a.C + c], x), s.den);
            v = fadd(vu, fmul(s.cfg, fsub(v, vu)));                // v_u + cfg (v_c - v_u)
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
            if (a.tok_cur) a.tok_cur[idx] = sg;
            if (a.tok_all) {
                const int b = bp / a.P, pp = bp % a.P;
                a.tok_all[((size_t)b * a.T + (size_t)a.state->step * a.P + pp) * a.C + c] = sg;
            }
        }
        a.xt[idx] = xn;
    }
}

int bdk_head_final(const HeadFinalArgs& a, hipStream_t st) {
    const size_t lds = (a.D + 32 + (ROW_THREADS / 64) * a.C + 2 * a.C) * sizeof(float);
    hipLaunchKernelGGL(head_final_kernel, dim3(a.BP), dim3(ROW_THREADS), lds, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

__global__ void init_latent_kernel(InitLatentArgs a) {
    const float* src = a.noise + (size_t)a.state->step * a.noise_step_stride;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += gridDim.x * blockDim.x) a.xt[i] = src[i];
}
int bdk_init_latent(const InitLatentArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(init_latent_kernel, dim3((a.n + 255) / 256), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------
// SwiGLU from split-K slabs (fallback when the fused GEMM epilogue cannot be used): act = silu(h1)*h2
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ROW_THREADS) void swiglu_rows_kernel(SwigluArgs a) {
    const int m = blockIdx.x;
    bf16_t* A = (bf16_t*)a.act_frag;
    for (int f = threadIdx.x; f < a.F; f += blockDim.x) {
        const int cg = a.interleaved ? ((f >> 4) * 32 + (f & 15)) : f;
        const int cu = a.interleaved ? cg + 16 : a.F + f;
        const float g = slab_bf(a.up, m, cg);
        const float u = slab_bf(a.up, m, cu);
        A[afrag_off(m, f, a.RB)] = f2bf(silu_bf(g) * u);
    }
}
int bdk_swiglu_rows(const SwigluArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(swiglu_rows_kernel, dim3(a.M), dim3(ROW_THREADS), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------
// projector fc1 + gelu(tanh)                                                   modeling/utils.py:16-19
// ------------------------------------------------------------------------------------------------
BD_DEV float gelu_tanh_f(float x) {            // torch GeluCUDAKernelImpl, approximate='tanh'
    const float kBeta = 0.7978845608028654f;   // sqrt(2/pi)
    const float inner = kBeta * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.0f + tanhf(inner));
}
__global__ __launch_bounds__(ROW_THREADS) void proj_fc1_kernel(ProjFc1Args a) {
    extern __shared__ float sh[];
    const int m = blockIdx.x;
    for (int k = threadIdx.x; k < a.C; k += blockDim.x) sh[k] = bfr(a.tok[(size_t)m * a.C + k]);
    __syncthreads();
    const bf16_t* W = (const bf16_t*)a.w;
    const bf16_t* B = (const bf16_t*)a.b;
    bf16_t* H = (bf16_t*)a.h_frag;
    for (int d = threadIdx.x; d < a.D; d += blockDim.x) {
        const float o = bfr(small_dot(sh, W + (size_t)d * a.C, a.C) + bf2f(B[d]));
        H[afrag_off(m, d, a.RB)] = f2bf(gelu_tanh_f(o));
    }
}
int bdk_proj_fc1(const ProjFc1Args& a, hipStream_t st) {
    hipLaunchKernelGGL(proj_fc1_kernel, dim3(a.BP), dim3(ROW_THREADS), a.C * sizeof(float), st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// model_input = fc2 output (bf16) + 2-D pos-embed (fp32) -> fp32, same rows for every CFG branch
// (t2i_pipeline.py:249-253; both halves of curr_tokens are identical copies)
__global__ __launch_bounds__(ROW_THREADS) void embed_finalize_kernel(EmbedFinalizeArgs a) {
    const int m = blockIdx.x;                    // 0..BP-1
    const int p = m % a.P;
    const int step = a.state->step;
    const float* pos = a.pos + ((size_t)step * a.P + p) * a.D;
    for (int d = threadIdx.x; d < a.D; d += blockDim.x) {
        const float e = slab_bf(a.fc2, m, d) + pos[d];
        for (int br = 0; br < a.branches; ++br) a.R[((size_t)br * a.BP + m) * a.D + d] = e;
    }
}
int bdk_embed_finalize(const EmbedFinalizeArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(embed_finalize_kernel, dim3(a.BP), dim3(ROW_THREADS), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

// ------------------------------------------------------------------------------------------------
// LLM: residual add of the pending branch + RMSNorm                             HF:59-64, 294-323
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(ROW_THREADS) void rms_kernel(RmsArgs a) {
    extern __shared__ float sh[];
    float* row = sh;
    float* red = sh + a.D;
    const int m = blockIdx.x;
    float ss = 0.f;
    for (int d = threadIdx.x; d < a.D; d += blockDim.x) {
        float r = a.R[(size_t)m * a.D + d];
        if (a.pend.p) {
            r = r + slab_bf(a.pend, m, d);                         // fp32 residual + bf16 branch -> fp32
            a.R[(size_t)m * a.D + d] = r;
        }
        row[d] = r;
        ss += r * r;
    }
    const float var = block_sum(ss, red) / (float)a.D;
    const float rs = rsqrtf(var + a.eps);
    const bf16_t* W = (const bf16_t*)a.w;
    bf16_t* A = (bf16_t*)a.a_frag;
    bf16_t* Cf = (bf16_t*)a.cond_frag;
    const float* pos = nullptr;
    if (a.pos) pos = a.pos + ((size_t)a.state->step * a.P + (m % a.P)) * a.D;
    for (int d = threadIdx.x; d < a.D; d += blockDim.x) {
        const float n = fmul(bf2f(W[d]), fmul(row[d], rs));        // weight * (x * rsqrt(var+eps)), fp32
        if (A) A[afrag_off(m, d, a.RB)] = f2bf(n);                  // cast by the following Linear
        if (a.hidden_out) a.hidden_out[(size_t)m * a.D + d] = n;
        if (Cf) Cf[afrag_off(m, d, a.RB)] = f2bf(fadd(n, pos[d]));  // cond = hidden + pos (t2i:244-245), cast by cond_embed
    }
}
int bdk_rms(const RmsArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(rms_kernel, dim3(a.M), dim3(ROW_THREADS), (a.D + 32) * sizeof(float), st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
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
            const float c0 = a.cos[(size_t)pos * 128 + lane], s0 = a.sin[(size_t)pos * 128 + lane];
            const float c1 = a.cos[(size_t)pos * 128 + lane + 64], s1 = a.sin[(size_t)pos * 128 + lane + 64];
            const float y0 = fadd(fmul(x0, c0), fmul(-x1, s0));    // q*cos + rotate_half(q)*sin
            const float y1 = fadd(fmul(x1, c1), fmul(x0, s1));
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
    hipLaunchKernelGGL(qkv_post_kernel, dim3(a.M, (nslot + 3) / 4), dim3(256), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

__global__ void step_advance_kernel(StepAdvanceArgs a) {
    if (threadIdx.x == 0) a.state->step += 1;
    if ((int)threadIdx.x < a.nseq) a.state->kv_len[threadIdx.x] += a.P;
}
int bdk_step_advance(const StepAdvanceArgs& a, hipStream_t st) {
    hipLaunchKernelGGL(step_advance_kernel, dim3(1), dim3(64), 0, st, a);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
