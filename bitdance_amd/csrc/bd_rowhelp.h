// Row-kernel helpers shared by bd_rows.hip and bd_sp.hip: 16 B loads / stores of eight consecutive channels of one row, the
// split-K slab reduction of a Linear output, block-wide LayerNorm statistics.
#pragma once
#include "bd_common.h"
#include "bd_kernels.h"

#define MAX_ROW_THREADS 1024

static inline int row_threads(int D) {                    // D/8 threads rounded up to whole waves
    int t = ((D / 8 + 63) / 64) * 64;
    return t > MAX_ROW_THREADS ? -1 : t;
}

BD_DEV void ld_bf16x8(const bf16_t* p, float* v) {
    const u32x4 q = *reinterpret_cast<const u32x4*>(p);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = bf2f((bf16_t)(q[j] & 0xffff)); v[2 * j + 1] = bf2f((bf16_t)(q[j] >> 16)); }
}
BD_DEV u32x4 ld_raw8(const bf16_t* p) { return *reinterpret_cast<const u32x4*>(p); }   // issue now, unpack later
BD_DEV void unpack8(const u32x4 q, float* v) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[2 * j] = bf2f((bf16_t)(q[j] & 0xffff)); v[2 * j + 1] = bf2f((bf16_t)(q[j] >> 16)); }
}
BD_DEV void ld_f32x8(const float* p, float* v) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
}
BD_DEV void st_f32x8(float* p, const float* v) {
    *reinterpret_cast<f32x4*>(p) = (f32x4){v[0], v[1], v[2], v[3]};
    *reinterpret_cast<f32x4*>(p + 4) = (f32x4){v[4], v[5], v[6], v[7]};
}
BD_DEV u32x4 pack8(const float* v) {
    return (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
}
// Linear output of 8 consecutive columns: bf16( sum of split-K slabs + bias )
BD_DEV void slab8(const Partial& q, int row, int col, float* v) {
    if (q.S == 0) {                                     // finished bf16 tensor (the GEMM reduced its own K-slices)
        const bf16_t* src = (const bf16_t*)q.p + (size_t)row * q.N + col;
        if (q.sys) {                                    // pushed into this GPU's memory by the tensor-parallel peers
            const unsigned long long lo = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(src), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const unsigned long long hi = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(src) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            unpack8((u32x4){(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)}, v);
        } else {
            ld_bf16x8(src, v);
        }
        return;
    }
    const float* p = q.p + (size_t)row * q.N + col;
    const size_t slab = (size_t)q.Mpad * q.N;
    u32x4 braw = {0, 0, 0, 0};
    if (q.bias) braw = ld_raw8((const bf16_t*)q.bias + col);
    // all slab loads of a batch in flight before the first add: a `for (s < S)` loop with a runtime S serialises S
    // dependent L2/MALL round trips (the whole row kernel is that latency chain); same summation order as before
    for (int s0 = 0; s0 < q.S; s0 += 6) {
        float t[6][8];
#pragma unroll
        for (int s = 0; s < 6; ++s)
            if (s0 + s < q.S) ld_f32x8(p + (size_t)(s0 + s) * slab, t[s]);
#pragma unroll
        for (int s = 0; s < 6; ++s)
            if (s0 + s < q.S) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (s0 + s == 0) ? t[s][j] : v[j] + t[s][j];
            }
    }
    if (q.bias) {
        float b[8];
        unpack8(braw, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] += b[j];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = bfr(v[j]);
}

// LayerNorm statistics over the row; inactive threads contribute zeros.  Two-pass (mean, then centred variance).
BD_DEV void ln_stats(const float* x, bool active, int D, float eps, float* red, float& mean, float& rstd) {
    float s = 0.f;
    if (active) {
#pragma unroll
        for (int j = 0; j < 8; ++j) s += x[j];
    }
    mean = block_sum(s, red) / (float)D;
    float v = 0.f;
    if (active) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float c = x[j] - mean; v += c * c; }
    }
    rstd = rsqrtf(block_sum(v, red) / (float)D + eps);
}

// block-wide sum of TWO values at once (cond / uncond row): halves the number of barrier round trips
BD_DEV void block_sum2(float& v0, float& v1, float* red) {
    v0 = wave_sum(v0); v1 = wave_sum(v1);
    const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[w] = v0; red[16 + w] = v1; }
    __syncthreads();
    float t0 = 0.f, t1 = 0.f;
    for (int i = 0; i < nw; ++i) { t0 += red[i]; t1 += red[16 + i]; }
    v0 = t0; v1 = t1;
}

// dot of an LDS fp32 vector with one bf16 weight row, K small (latent channels); 16 B loads when K % 8 == 0
BD_DEV float small_dot(const float* x, const bf16_t* w, int K) {
    float acc = 0.f;
    if ((K & 7) == 0) {
        for (int k = 0; k < K; k += 8) {
            float wv[8];
            ld_bf16x8(w + k, wv);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += x[k + j] * wv[j];
        }
    } else {
        for (int k = 0; k < K; ++k) acc += x[k] * bf2f(w[k]);
    }
    return acc;
}

