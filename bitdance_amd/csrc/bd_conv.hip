// Native kernels of the tokenizer's conv decoder (/root/reference/modeling/vision_encoder/autoencoder.py:129-277):
//   * conv_tile_kernel   3x3 (and 1x1) convolution as an implicit GEMM on the matrix pipe;
//   * gn_stats_kernel / gn_finalize_kernel   GroupNorm(32) statistics, deterministic two-level reduction;
//   * gn_apply_kernel    normalise (+affine / AdaptiveGroupNorm scale-bias) (+swish) and lay the result down as the next
//                        convolution's input;
//   * tokens_to_padded_kernel   [B, C, h, w] latent -> that input layout.
// Host orchestration: bitdance_amd/ae_native.py (same module / checkpoint surface as the torch decoder it replaces).
//
// Layout.  Activations are NHWC.  A convolution INPUT is bf16 with a one-pixel zero border, [n][H + 2][W + 2][C] ("padded"):
// the nine taps of a 3x3 then are nine plain shifted views, no boundary branches in the kernel.  The residual stream between
// the blocks is unpadded [n][H][W][C], fp32 or bf16 exactly where the reference's autocast flow has it (bf16 conv output +
// fp32 AdaGN output -> fp32; bf16 + bf16 -> bf16).
//
// The convolution is GEMM  out[pixel, co] = sum_{tap, ci} in[pixel + tap, ci] * w[co, tap, ci]:  M = pixels, N = C_out,
// K = taps * C_in ordered (tap, ci), so the WEIGHTS are an ordinary [C_out][9 C_in] matrix in the packed MFMA-operand order of
// bd_gemm.hip (pack_w_kernel) and the kernel is the LDS-tiled GEMM of bd_gemm_tile.hip -- 256 pixels x 256 channels per
// workgroup, 8 waves as two ping-pong groups, LDS-DMA ring of four 32-deep stages -- with one difference: an A-operand chunk
// (32 pixels x 16 channels) is GATHERED, every lane's 16 bytes coming from its own pixel's channel run (global_load_lds takes a
// per-lane address and still writes the LDS chunk lane-linearly, i.e. in fragment order).  A 32-deep stage = 64 B per pixel,
// two stages = one 128 B line, and the 9 taps re-read the same lines from L2.
// A pixel tile is 256 / TW rows x TW columns of the image (TW = 32, or 16 for 16-pixel-wide maps).
#include <string>
#include "bd_gemm_kernel.h"

namespace {

constexpr int CV_STAGE_UNITS = 32 * 64;
constexpr int CV_SLOTS = 4;

BD_DEV void cv_dma(const void* gsrc, unsigned lds_byte) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}

}  // namespace

struct ConvP {
    const bf16_t* in;        // padded bf16 NHWC [n][H+2][W+2][Cin] (taps = 9) or unpadded [n][H][W][Cin] (taps = 1)
    const u32x4* Wt;         // packed [Npad][taps * Cin] (bd_pack_weight), Npad = Cout rounded up to 256
    const bf16_t* bias;      // [Cout] or null
    const void* res;         // residual, unpadded NHWC [n][H][W][Cout], fp32 or bf16, or null
    void* out;
    int n, H, W, Cin, Cout, taps, TW;   // H, W: OUTPUT size; the input is [stride * H (+2)][stride * W (+2)]
    int stride;              // 1, or 2 (3x3, padding 1: the encoder's down-sampling convolution) -- output pixel (y, x) gathers taps around (2y, 2x)
    int res_f32, out_f32;    // dtypes of the residual / of the output (out_mode 0)
    int out_mode;            // 0: unpadded NHWC [n][H][W][Cout]   1: depth-to-space, bf16 NHWC [n][2H][2W][Cout/4], channel = (dy, dx, c)
                             // 2: image, fp32 NCHW [n][Cout][H][W]  3: padded bf16 NHWC [n][H+2][W+2][Cout] (interior)
    size_t PS;               // packed-weight panel stride in 16 B units
};

__global__ __launch_bounds__(512) void conv_tile_kernel(ConvP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const u32x4* const lds = reinterpret_cast<const u32x4*>(smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;

    // ---- tile: blockIdx.x -> (pixel tile, channel tile); pixel tiles fastest within groups of 8 so that the workgroups that share a
    //      weight tile are dispatched together (weights from L2), and neighbouring pixel tiles share their halo rows
    const int TW = p.TW, TH = 256 / TW;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH, tiles_img = tiles_x * tiles_y;   // edge tiles may be partial
    const int MT = p.n * tiles_img, NT = (p.Cout + 255) >> 8;
    const int b = blockIdx.x;
    const int grp = b / (8 * NT), rem = b - grp * 8 * NT;        // groups of 8 pixel tiles x all channel tiles
    const int pt0 = grp * 8, np = min(8, MT - pt0);
    const int nt = rem / np, pt = pt0 + rem % np;
    if (nt >= NT) return;                                       // the last, short group
    const int img = pt / tiles_img, t2 = pt - img * tiles_img;
    const int y0 = (t2 / tiles_x) * TH, x0 = (t2 % tiles_x) * TW;

    const int spt = p.Cin >> 5;                                 // 32-deep stages per tap
    const int nst = p.taps * spt;
    const int sd = p.stride;
    const int Wp = p.taps == 9 ? p.W * sd + 2 : p.W * sd, Hp = p.taps == 9 ? p.H * sd + 2 : p.H * sd;

    // ---- DMA sources: waves 0-3 gather A (chunk c = 4 wave + j: k-step c >> 3, row block c & 7), waves 4-7 stream W
    const char* src[4];
    unsigned dst[4];
    if (wave < 4) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = wave * 4 + j, ks = c >> 3, rb = c & 7;
            const int pix = rb * 32 + (lane & 31), ty = pix / TW, tx = pix - ty * TW;
            const int yy = min(y0 + ty, p.H - 1), xx = min(x0 + tx, p.W - 1);      // pixels past the edge of a partial tile: re-read the edge
            src[j] = reinterpret_cast<const char*>(p.in + (((size_t)img * Hp + yy * sd) * Wp + xx * sd) * p.Cin + ks * 16 + (lane >> 5) * 8);
            dst[j] = (unsigned)c * 1024u;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = (wave - 4) * 4 + j, pn = c >> 1, ks = c & 1;
            src[j] = reinterpret_cast<const char*>(p.Wt + (size_t)(nt * 8 + pn) * p.PS + (size_t)ks * 64 + lane);
            dst[j] = (unsigned)(16 + c) * 1024u;
        }
    }
    const bool isX = wave < 4;
    // byte offset of stage st from stage 0: A: tap (ky, kx) shifts the pixel, then 32 channels per stage; W: 2 k-steps of 1 KiB
    auto stage_off = [&](int st) -> size_t {
        if (!isX) return (size_t)st * 2048;
        const int tap = st / spt, cs = st - tap * spt;
        const int ky = tap / 3, kx = tap - ky * 3;
        return ((size_t)(ky * Wp + kx) * p.Cin + (size_t)cs * 32) * 2;
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    u32x4 af[2][4], wf[2][2];
    auto issue = [&](int st) {
        const unsigned slot = (unsigned)(st & (CV_SLOTS - 1)) * (CV_STAGE_UNITS * 16);
        const size_t off = stage_off(st);
#pragma unroll
        for (int j = 0; j < 4; ++j) cv_dma(src[j] + off, slot + dst[j]);
    };
    auto load_seg = [&](int j) {
        const u32x4* a = lds + (size_t)(j & (CV_SLOTS - 1)) * CV_STAGE_UNITS + lane;
        const u32x4* w = a + 16 * 64;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int m = 0; m < 4; ++m) af[ks][m] = a[(ks * 8 + (wr * 4 + m)) * 64];
#pragma unroll
            for (int n = 0; n < 2; ++n) wf[ks][n] = w[((wc * 2 + n) * 2 + ks) * 64];
        }
        if (j + 2 >= nst) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    };
    auto mfma_seg = [&](int j) {
        const bool dma = j + 3 < nst;
        const unsigned slot = (unsigned)((j + 3) & (CV_SLOTS - 1)) * (CV_STAGE_UNITS * 16);
        const size_t off = dma ? stage_off(j + 3) : 0;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = mfma32(af[ks][m], wf[ks][n], acc[m][n]);
                if ((m & 1) == 1) {
                    const int c = ks * 2 + (m >> 1);
                    if (dma) cv_dma(src[c] + off, slot + dst[c]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        __builtin_amdgcn_sched_barrier(0);
    };

    issue(0);
    if (1 < nst) issue(1);
    if (2 < nst) issue(2);
    {
        const int younger = min(nst - 1, 2);
        if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (isX) {
        for (int j = 0; j < nst; ++j) {
            __syncthreads();
            load_seg(j);
            __syncthreads();
            mfma_seg(j);
        }
        __syncthreads();
        BD_MFMA_DRAIN();                                        // (bd_common.h: drain in the block that holds the last MFMAs)
    } else {
        __syncthreads();
        __syncthreads();
        load_seg(0);
        for (int j = 1; j < nst; ++j) {
            __syncthreads();
            mfma_seg(j - 1);
            __syncthreads();
            load_seg(j);
        }
        __syncthreads();
        mfma_seg(nst - 1);
        BD_MFMA_DRAIN();
    }

    // ---- epilogue: conv output = bf16(acc + bias) (what F.conv2d returns under autocast), parked per wave in LDS as
    //      [128 pixels][64 channels], then written 8 channels (16 B of bf16) per lane with the residual / layout of out_mode
    bf16_t* const mine = reinterpret_cast<bf16_t*>(smem) + (size_t)wave * (128 * 64);
    const int co_w = nt * 256 + wc * 64;                        // first output channel of this wave
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int lc = n * 32 + (lane & 31), co = co_w + lc;
        const float bias_col = (p.bias && co < p.Cout) ? bf2f(p.bias[co]) : 0.f;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                mine[(m * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 64 + lc] = f2bf(acc[m][n][r] + bias_col);
    }
    const u32x4* const rd = reinterpret_cast<const u32x4*>(mine);
    const int seg = lane & 7, co = co_w + seg * 8;
    if (co >= p.Cout) return;                                   // channel padding of the last tile (Cout % 8 == 0)
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int lp = it * 8 + (lane >> 3);                    // pixel of this wave's 128
        const int pix = wr * 128 + lp, ty = pix / TW, tx = pix - ty * TW;
        const int y = y0 + ty, x = x0 + tx;
        if (y >= p.H || x >= p.W) continue;                     // partial edge tile
        const u32x4 q = rd[lp * 8 + seg];
        float v[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[2 * j] = bf2f((bf16_t)(q[j] & 0xffff)); v[2 * j + 1] = bf2f((bf16_t)(q[j] >> 16)); }
        if (p.out_mode == 0) {
            const size_t o = (((size_t)img * p.H + y) * p.W + x) * p.Cout + co;
            if (p.res) {
                if (p.res_f32) {
                    const f32x4 r0 = *reinterpret_cast<const f32x4*>((const float*)p.res + o), r1 = *reinterpret_cast<const f32x4*>((const float*)p.res + o + 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] += r0[j]; v[4 + j] += r1[j]; }
                } else {
                    const u32x4 r = *reinterpret_cast<const u32x4*>((const bf16_t*)p.res + o);
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[2 * j] += bf2f((bf16_t)(r[j] & 0xffff)); v[2 * j + 1] += bf2f((bf16_t)(r[j] >> 16)); }
                }
            }
            if (p.out_f32) {
                *reinterpret_cast<f32x4*>((float*)p.out + o) = (f32x4){v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>((float*)p.out + o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
            } else {
                *reinterpret_cast<u32x4*>((bf16_t*)p.out + o) = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
            }
        } else if (p.out_mode == 1) {                           // depth to space (DCR): channel co = (dy, dx, c)
            const int Cq = p.Cout >> 2, quad = co / Cq, c = co - quad * Cq;
            const int dy = quad >> 1, dx = quad & 1;
            const size_t o = (((size_t)img * (2 * p.H) + 2 * y + dy) * (2 * p.W) + 2 * x + dx) * Cq + c;
            *reinterpret_cast<u32x4*>((bf16_t*)p.out + o) = q;
        } else if (p.out_mode == 3) {
            const size_t o = (((size_t)img * (p.H + 2) + y + 1) * (p.W + 2) + x + 1) * p.Cout + co;
            *reinterpret_cast<u32x4*>((bf16_t*)p.out + o) = q;
        } else {                                                // image: fp32 NCHW
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (co + j < p.Cout) ((float*)p.out)[(((size_t)img * p.Cout + co + j) * p.H + y) * p.W + x] = v[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// GroupNorm(32 groups) statistics of an unpadded NHWC tensor (fp32 or bf16): level 1 = one workgroup per (image, chunk of 256
// pixels): per-group sum and sum of squares of its chunk; level 2 = one thread per (image, group) adds the chunks IN ORDER in
// double precision: deterministic, no atomics.  out: [n][32][2] = (mean, rstd).
// ------------------------------------------------------------------------------------------------------------------
struct GnStatsP { const void* x; int x_f32; float* partial; int n, HW, C, chunks; };

__global__ __launch_bounds__(256) void gn_stats_kernel(GnStatsP p) {
    const int chunk = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
    const int C8 = p.C >> 3, cpg = p.C >> 5;                    // channel octets per pixel, channels per group
    const int px0 = chunk * 256, px1 = min(p.HW, px0 + 256);
    // thread -> fixed channel octet (so its group(s) are fixed), striding over the chunk's pixels
    const int oct = tid % C8, lane_px = tid / C8, px_step = 256 / C8;     // C8 in {4 .. 128}: divides 256 for power-of-two C
    float a1[8], a2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a1[j] = a2[j] = 0.f;
    if (px_step > 0 && lane_px < px_step) {
        for (int px = px0 + lane_px; px < px1; px += px_step) {
            const size_t o = ((size_t)img * p.HW + px) * p.C + oct * 8;
            float v[8];
            if (p.x_f32) {
                const f32x4 r0 = *reinterpret_cast<const f32x4*>((const float*)p.x + o), r1 = *reinterpret_cast<const f32x4*>((const float*)p.x + o + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = r0[j]; v[4 + j] = r1[j]; }
            } else {
                const u32x4 q = *reinterpret_cast<const u32x4*>((const bf16_t*)p.x + o);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[2 * j] = bf2f((bf16_t)(q[j] & 0xffff)); v[2 * j + 1] = bf2f((bf16_t)(q[j] >> 16)); }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) { a1[j] += v[j]; a2[j] += v[j] * v[j]; }
        }
    }
    // every thread parks its sums; thread g < 32 then adds the threads of group g in a FIXED order (deterministic)
    __shared__ float t1[256][2], t2[256][2];
    if (cpg >= 8) {                                             // an octet lies inside one group
        float u1 = 0.f, u2 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { u1 += a1[j]; u2 += a2[j]; }
        t1[tid][0] = u1; t2[tid][0] = u2;
        __syncthreads();
        if (tid < 32) {                                          // group g: octets g * cpg/8 .. ; every pixel lane, in order
            const int opg = cpg >> 3;
            float r1 = 0.f, r2 = 0.f;
            for (int lp = 0; lp < px_step; ++lp)
                for (int o = 0; o < opg; ++o) { const int t = lp * C8 + tid * opg + o; r1 += t1[t][0]; r2 += t2[t][0]; }
            p.partial[(((size_t)img * p.chunks + chunk) * 32 + tid) * 2] = r1;
            p.partial[(((size_t)img * p.chunks + chunk) * 32 + tid) * 2 + 1] = r2;
        }
    } else {                                                    // fewer than 8 channels per group (narrow test models): per channel
        __shared__ float c1[256 * 8], c2[256 * 8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { c1[tid * 8 + j] = a1[j]; c2[tid * 8 + j] = a2[j]; }
        __syncthreads();
        if (tid < 32) {
            float r1 = 0.f, r2 = 0.f;
            for (int lp = 0; lp < px_step; ++lp)
                for (int cc = 0; cc < cpg; ++cc) {
                    const int ch = tid * cpg + cc, t = lp * C8 + (ch >> 3);
                    r1 += c1[t * 8 + (ch & 7)]; r2 += c2[t * 8 + (ch & 7)];
                }
            p.partial[(((size_t)img * p.chunks + chunk) * 32 + tid) * 2] = r1;
            p.partial[(((size_t)img * p.chunks + chunk) * 32 + tid) * 2 + 1] = r2;
        }
    }
}

struct GnFinalP { const float* partial; float* stats; int n, HW, C, chunks; float eps; };
// one workgroup per (image, group): thread t adds chunks t, t + 256, ... in double, then a fixed-shape tree over the 256 threads
__global__ __launch_bounds__(256) void gn_finalize_kernel(GnFinalP p) {
    __shared__ double r1[256], r2[256];
    const int i = blockIdx.x, img = i >> 5, g = i & 31, tid = threadIdx.x;
    double s1 = 0.0, s2 = 0.0;
    for (int c = tid; c < p.chunks; c += 256) {
        const float* q = p.partial + (((size_t)img * p.chunks + c) * 32 + g) * 2;
        s1 += (double)q[0];
        s2 += (double)q[1];
    }
    r1[tid] = s1; r2[tid] = s2;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if (tid < h) { r1[tid] += r1[tid + h]; r2[tid] += r2[tid + h]; }
        __syncthreads();
    }
    if (tid == 0) {
        const double cnt = (double)p.HW * (p.C >> 5);
        const double mean = r1[0] / cnt;
        double var = r2[0] / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        p.stats[i * 2] = (float)mean;
        p.stats[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
}

// y = (x - mean) * rstd [* gamma[c] + beta[c]] [then scale[n][c] * y + bias[n][c]] [then y * sigmoid(y)]
// -> out_mode 0: padded bf16 NHWC (interior; the border stays zero)   1: unpadded fp32   2: unpadded bf16
struct GnApplyP {
    const void* x; int x_f32;
    const float* stats;        // [n][32][2] or null (identity: y = x)
    const float* gamma; const float* beta;        // [C] or null
    const float* scale; const float* bias;        // AdaptiveGroupNorm: [n][C] or null
    void* out; int out_mode; int swish;
    int n, H, W, C;
};
// one workgroup per image row; a thread keeps ONE channel octet (C / 8 divides 256 or is a multiple of it), so its affine
// parameters, group statistics and AdaGN scale / bias are loaded once, and all index math is 32-bit shifts
__global__ __launch_bounds__(256) void gn_apply_kernel(GnApplyP p) {
    const int C8 = p.C >> 3, cpg = p.C >> 5;
    const int row = blockIdx.x, img = row / p.H, y = row - img * p.H;
    const int tid = threadIdx.x;
    for (int oct0 = tid % C8; oct0 < C8; oct0 += 256) {          // one pass unless C8 > 256
        float ga[8], be[8], mu[8], rs[8], sc[8], bi[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = oct0 * 8 + j;
            ga[j] = p.gamma ? p.gamma[c] : 1.f;
            be[j] = p.beta ? p.beta[c] : 0.f;
            if (p.stats) { const float* st = p.stats + ((size_t)img * 32 + c / cpg) * 2; mu[j] = st[0]; rs[j] = st[1]; }
            else { mu[j] = 0.f; rs[j] = 1.f; }
            sc[j] = p.scale ? p.scale[(size_t)img * p.C + c] : 1.f;
            bi[j] = p.bias ? p.bias[(size_t)img * p.C + c] : 0.f;
        }
        // units with this octet: v = x * C8 + oct0, visited by this thread for x = x_first, x_first + 256 / C8 (or every x when C8 >= 256)
        const int xstep = C8 >= 256 ? 1 : 256 / C8, xfirst = C8 >= 256 ? 0 : tid / C8;
        for (int x = xfirst; x < p.W; x += xstep) {
            const size_t o = (((size_t)img * p.H + y) * p.W + x) * p.C + oct0 * 8;
            float v[8];
            if (p.x_f32) {
                const f32x4 r0 = *reinterpret_cast<const f32x4*>((const float*)p.x + o), r1 = *reinterpret_cast<const f32x4*>((const float*)p.x + o + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] = r0[j]; v[4 + j] = r1[j]; }
            } else {
                const u32x4 q = *reinterpret_cast<const u32x4*>((const bf16_t*)p.x + o);
#pragma unroll
                for (int j = 0; j < 4; ++j) { v[2 * j] = bf2f((bf16_t)(q[j] & 0xffff)); v[2 * j + 1] = bf2f((bf16_t)(q[j] >> 16)); }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float t = (v[j] - mu[j]) * rs[j];
                if (p.gamma) t = t * ga[j] + be[j];
                if (p.scale) t = sc[j] * t + bi[j];
                if (p.swish) t = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * t));
                v[j] = t;
            }
            if (p.out_mode == 0) {
                const size_t po = (((size_t)img * (p.H + 2) + y + 1) * (p.W + 2) + x + 1) * p.C + oct0 * 8;
                *reinterpret_cast<u32x4*>((bf16_t*)p.out + po) = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
            } else if (p.out_mode == 1) {
                *reinterpret_cast<f32x4*>((float*)p.out + o) = (f32x4){v[0], v[1], v[2], v[3]};
                *reinterpret_cast<f32x4*>((float*)p.out + o + 4) = (f32x4){v[4], v[5], v[6], v[7]};
            } else {
                *reinterpret_cast<u32x4*>((bf16_t*)p.out + o) = (u32x4){pack2(v[0], v[1]), pack2(v[2], v[3]), pack2(v[4], v[5]), pack2(v[6], v[7])};
            }
        }
    }
}

// latent tokens [n][C][h][w] fp32 (NCHW) -> padded bf16 NHWC interior
__global__ void tokens_to_padded_kernel(const float* z, bf16_t* out, int n, int C, int H, int W) {
    const size_t total = (size_t)n * H * W * C;
    for (size_t u = (size_t)blockIdx.x * blockDim.x + threadIdx.x; u < total; u += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(u % C);
        const size_t pxl = u / C;
        const int x = (int)(pxl % W), y = (int)((pxl / W) % H), img = (int)(pxl / ((size_t)W * H));
        out[(((size_t)img * (H + 2) + y + 1) * (W + 2) + x + 1) * C + c] = f2bf(z[(((size_t)img * C + c) * H + y) * W + x]);
    }
}

void bdk_set_error(const std::string& m);                        // bd_api.hip
static int bd_last_error_set(const char* m) { bdk_set_error(m); return -1; }
// a shape / mode the kernels do not cover, refused BEFORE anything is launched (include/bitdance_hip.h BD_ERR_UNSUPPORTED): the one
// error a caller may answer by taking another path; launch / runtime failures keep -1 and must propagate
static int bd_unsupported(const char* m) { bdk_set_error(m); return -22; }

extern "C" {

/* 3x3 (taps = 9, `in` padded) or 1x1 (taps = 1, `in` unpadded) convolution; w = bd_pack_weight of [Cout padded to 256][taps * Cin]
 * with K ordered (ky, kx, ci).  The image is covered by 256-pixel tiles of (256 / TW) rows x TW columns, TW in {32, 16, 8} chosen for
 * the least padding (the released sizes have latent grids of any multiple of 8: 16 .. 128); edge tiles may be partial. */
int bd_conv_strided(const void* in, const void* w_packed, const void* bias, const void* res, int res_f32, void* out, int out_mode, int out_f32,
                    int n, int H, int W, int Cin, int Cout, int taps, int stride, void* stream) {
    if (stride != 1 && !(stride == 2 && taps == 9)) return bd_unsupported("bd_conv_strided: stride 1, or 2 with a 3x3 kernel");
    int TW = 32;
    {
        long long best = -1;
        for (int tw : {32, 16, 8}) {
            const int th = 256 / tw;
            const long long area = (long long)((W + tw - 1) / tw) * tw * ((H + th - 1) / th) * th;
            if (best < 0 || area < best) { best = area; TW = tw; }
        }
    }
    if ((taps != 9 && taps != 1) || Cin % 32 || (Cout % 8 && out_mode != 2) || H < 1 || W < 1 || out_mode < 0 || out_mode > 3 ||
        (out_mode == 1 && (Cout % 4 || (Cout / 4) % 8)))
        return bd_unsupported("bd_conv: taps 9 / 1, Cin % 32, Cout % 8 (any Cout for the image output), depth-to-space needs Cout / 4 % 8 == 0");
    ConvP p;
    p.in = (const bf16_t*)in; p.Wt = (const u32x4*)w_packed; p.bias = (const bf16_t*)bias; p.res = res; p.out = out;
    p.n = n; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.taps = taps; p.TW = TW; p.stride = stride;
    p.res_f32 = res_f32; p.out_f32 = out_f32; p.out_mode = out_mode;
    p.PS = (size_t)((taps * Cin) >> 4) * 64;
    const int MT = n * ((H + 256 / TW - 1) / (256 / TW)) * ((W + TW - 1) / TW), NT = (Cout + 255) / 256;
    const int blocks = ((MT + 7) / 8) * 8 * NT;
    constexpr int lds = CV_SLOTS * CV_STAGE_UNITS * 16;
    static unsigned long long optin = 0;
    if (!bd_lds_optin((const void*)conv_tile_kernel, lds, &optin)) return bd_last_error_set("bd_conv: LDS opt-in failed");
    BD_LAUNCH(conv_tile_kernel, dim3(blocks), dim3(512), lds, (hipStream_t)stream, p);
    return bd_launch_status() == 0 ? 0 : bd_last_error_set("bd_conv: launch failed");
}

int bd_conv(const void* in, const void* w_packed, const void* bias, const void* res, int res_f32, void* out, int out_mode, int out_f32,
            int n, int H, int W, int Cin, int Cout, int taps, void* stream) {
    return bd_conv_strided(in, w_packed, bias, res, res_f32, out, out_mode, out_f32, n, H, W, Cin, Cout, taps, 1, stream);
}

/* GroupNorm(32) statistics of an unpadded NHWC tensor -> stats [n][32][2] = (mean, rstd); partial: scratch [n][ceil(HW / 256)][32][2] fp32 */
int bd_gn_stats(const void* x, int x_f32, float* partial, float* stats, int n, int HW, int C, float eps, void* stream) {
    if (C % 32 || C < 32 || C > 2048 || (256 % (C / 8)) != 0) return bd_unsupported("bd_gn_stats: C must be a power-of-two multiple of 32, <= 2048");
    GnStatsP p{x, x_f32, partial, n, HW, C, (HW + 255) / 256};
    BD_LAUNCH(gn_stats_kernel, dim3(p.chunks, n), dim3(256), 0, (hipStream_t)stream, p);
    GnFinalP f{partial, stats, n, HW, C, p.chunks, eps};
    BD_LAUNCH(gn_finalize_kernel, dim3(n * 32), dim3(256), 0, (hipStream_t)stream, f);
    return bd_launch_status() == 0 ? 0 : bd_last_error_set("bd_gn_stats: launch failed");
}

int bd_gn_apply(const void* x, int x_f32, const float* stats, const float* gamma, const float* beta, const float* scale, const float* bias,
                void* out, int out_mode, int swish, int n, int H, int W, int C, void* stream) {
    if (C % 32) return bd_unsupported("bd_gn_apply: C % 32");
    GnApplyP p{x, x_f32, stats, gamma, beta, scale, bias, out, out_mode, swish, n, H, W, C};
    if ((C / 8) < 256 ? (256 % (C / 8)) != 0 : ((C / 8) % 256) != 0) return bd_unsupported("bd_gn_apply: C / 8 must divide, or be a multiple of, 256");
    BD_LAUNCH(gn_apply_kernel, dim3(n * H), dim3(256), 0, (hipStream_t)stream, p);
    return bd_launch_status() == 0 ? 0 : bd_last_error_set("bd_gn_apply: launch failed");
}

int bd_tokens_to_padded(const float* z, void* out, int n, int C, int H, int W, void* stream) {
    const size_t total = (size_t)n * C * H * W;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    BD_LAUNCH(tokens_to_padded_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z, (bf16_t*)out, n, C, H, W);
    return bd_launch_status() == 0 ? 0 : bd_last_error_set("bd_tokens_to_padded: launch failed");
}

}  // extern "C"
