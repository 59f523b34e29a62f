// Internal launcher declarations (C++ side of libbitdance_hip.so). Public C ABI: include/bitdance_hip.h
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bd_common.h"          // (BD_STAMP_FIELD: every translation unit must see the same struct layouts)

// hipGetLastError() is sticky per thread: another library's benign failure (torch probing peers, querying an
// unfinished event ...) would otherwise be blamed on our launch.  Clear before, check after.
#define BD_LAUNCH(...) do { (void)hipGetLastError(); hipLaunchKernelGGL(__VA_ARGS__); } while (0)
static inline int bd_launch_status() { return hipGetLastError() == hipSuccess ? 0 : -1; }

// Opt a kernel in to more than 64 KiB of dynamic LDS.  The attribute is PER DEVICE: a process that drives a second GPU later
// (replicas in one process, tests that move between devices) must set it there too, so the "done" state is a bit per device id
// in `*done_mask` (one static word per call site / kernel instantiation), not a process-wide once-flag.
static inline bool bd_lds_optin(const void* fn, int bytes, unsigned long long* done_mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
    const unsigned long long bit = 1ull << dev;
    if (__atomic_load_n(done_mask, __ATOMIC_RELAXED) & bit) return true;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) { (void)hipGetLastError(); return false; }
    __atomic_fetch_or(done_mask, bit, __ATOMIC_RELAXED);
    return true;
}

#define BD_EPI_PARTIAL 0
#define BD_EPI_SWIGLU 1
#define BD_EPI_BF16 2       // out bf16 row-major [Mpad][N] = bf16(acc + bias); split-K > 1: reduced inside the launch
#define BD_EPI_F32 3        // out fp32 row-major [Mpad][N] = the finished K sum, no bias / rounding (a tensor-parallel rank's partial)

struct BdStepState;

// ---- bd_gemm.hip
// wscale != nullptr: W holds fp8-e4m3 weights (bdk_pack_w8) with per-packed-row fp32 scales
int bdk_gemm(const void* A, int RB, const void* W, int N, int K, int S, int nw, int epi,
             float* out_partial, void* out_act, const void* bias, int* tile_counters, hipStream_t st, const float* wscale = nullptr);
// ---- bd_gemm8.hip : the same GEMM on fp8-e4m3 weights
int bdk_gemm8(const void* A, int RB, const void* W8, const float* wscale, int N, int K, int S, int nw, int epi,
              float* out_partial, void* out_act, const void* bias, int* tile_counters, hipStream_t st);
// fp8 weights + fp8 activations on the fp8 matrix pipe (bd_gemm_kernel.h WT = 2)
int bdk_gemm8a(const void* A8, const float* ascale, int RB, const void* W8k, const float* wscale, int N, int K, int S, int nw, int epi,
               float* out_partial, void* out_act, const void* bias, int* tile_counters, hipStream_t st);
int bdk_pack_w8k(void* dst, const void* src_fp8, const void* src2_fp8, int panels, int K, int nb0, int panels_total, int mode, hipStream_t st);
int bdk_pack_w8(void* dst, const void* src_fp8, const void* src2_fp8, int panels, int K, int nb0, int panels_total, int mode, hipStream_t st);
int bdk_pack_w(void* dst, const void* src, const void* src2, int panels, int K, int nb0, int panels_total, int mode, hipStream_t st);
void bdk_set_w_layout(int v);
int bdk_set_gemm_option(const char* name, int v);   // process-wide measurement switches (bd_gemm.hip)
int bdk_get_w_layout();
void bdk_w_strides(int panels_total, int K, size_t* PS, size_t* SS);
int bdk_probe_read(const void* src, size_t bytes, int blocks, void* sink, hipStream_t st);
int bdk_quant_rows8(void* a8, float* ascale, const float* src, int M, int K, int RB, hipStream_t st);   // bd_rows.hip
int bdk_rows_to_afrag(void* dst, const float* src32, const void* src16, int M, int K, int RB, hipStream_t st);

// ---- bd_rows.hip : row-wise kernels (one workgroup per activation row)
struct Partial {            // a Linear output as the consumer sees it:
    const float* p;         //   S >= 1: split-K slabs [S][Mpad][N] fp32 (+ optional bf16 bias[N]) to be summed and rounded
    const void* bias;       //   S == 0: p is a FINISHED bf16 row-major [Mpad][N] tensor (bias added, rounded) -- the
    int S, N, Mpad;         //           GEMM reduced its own K-slices (last-arriver epilogue)
    int sys = 0;            //   1 (with S == 0): the tensor was written by OTHER GPUs over xGMI (tensor-parallel all-reduce,
};                          //           bd_comm.hip): read it with system-scope loads, the L2 may hold stale lines

// diffusion head
struct HeadPrologueArgs {   // y = silu(t_emb + cond_embed(c)) ; x0 = input_proj(x_t)      flow_head:326-330
    const void* cemb;       // cond_embed(c) finalised once per AR step: bf16 row-major [Mpad][D]
    const void* temb;       // [D] bf16: time_embed(t_i) for this eval
    const float* xt;        // [B*P][C] fp32 latent
    const void* in_w;       // [D][C] bf16
    const void* in_b;       // [D] bf16
    void* y_frag;           // out: fragment-major bf16 [Mpad][D]
    void* X;                // out: bf16 row-major [Mpad][D]
    int M, BP, D, C, RB;
    float* a8_scale = nullptr;   // not null: y goes out as fp8-e4m3 (A8 layout) + per-row scale
};
int bdk_head_prologue(const HeadPrologueArgs& a, hipStream_t st);

struct LnModArgs {          // x (+= pending branch * gate) ; h = LN(x)*(1+scale)+shift        flow_head:242-252
    void* X;                // in/out bf16 row-major [Mpad][D]
    Partial pend;           // pending branch output (wo / w2 slabs + bias); p == nullptr -> none
    const void* ada;        // adaLN Linear output, bf16 row-major [Mpad][ada_ld]
    int ada_ld;
    int gate_off, scale_off, shift_off;   // column offsets inside the adaLN output
    const float* ln_w;      // [D] fp32 or null (no affine)
    const float* ln_b;
    void* h_frag;           // out: fragment-major bf16 (or the fp8 operand when a8_scale is set)
    int M, D, RB;
    float eps;
    float* a8_scale = nullptr;   // not null: h goes out as fp8-e4m3 in the A8 layout + one fp32 scale per row (fp8 x fp8 GEMMs)
    int wave_rows = 0;           // 1: one wave per row, eight rows per workgroup (many rows of a narrow model; D <= 1024, bf16 output)
    BD_STAMP_FIELD
};
int bdk_ln_mod(const LnModArgs& a, hipStream_t st);

struct SamplerScalars {     // data-independent scalars of one sampling step (sampling_x.py:33-41,77-95)
    float t, dt, den, var, omt, noise_scale, cfg;
    int is_final, cfg_mult;
};
struct HeadFinalArgs {      // pending update, final LN-mod, Linear(D->C), 2*sigmoid-1, SDE/ODE step, sign
    const void* X;          // bf16 row-major
    Partial pend;           // w2 slabs of the last block (+bias)
    const void* ada;        // bf16 row-major [Mpad][ada_ld]
    int ada_ld;
    int gate_off, scale_off, shift_off;
    const void* lin_w;      // [C][D] bf16
    const void* lin_b;      // [C] bf16
    float* xt;              // in/out latent [B*P][C] fp32
    const float* noise;     // pre-drawn normals [AR steps][N+1][B*P][C] in the reference's RNG call order
    long long noise_step_stride;   // (N+1)*B*P*C
    int eval_index;         // i: this eval consumes noise[step][i+1] (ignored on the final step)
    const BdStepState* state;
    float* pred_out;        // final step: sampled latent x [B*P][C] (pre-sign), may be null
    float* tok_cur;         // final step: sign(x) [B*P][C] fp32 for the projector, may be null
    float* tok_all;         // final step: sign(x) scattered to [B][T][C] at token step*P+p, may be null
    int T, P;
    float* xhat_out;        // debug: x_hat rows [M][C] fp32, may be null
    SamplerScalars sc;
    int BP, D, C, M;
    float eps_ln;
    int sigmoid;            // 1: x_hat = 2*sigmoid(out)-1 (flow_head_parallel_x.py:342); 0: x_hat = out (diff_head_parallel.py:310)
    const float* cfg_table = nullptr;   // not null: the guidance scale of THIS AR step = cfg_table[state->step] (the ImageNet sampler's
                            // linear ramp, model_parallel.py:356-365, as device data so that one captured graph serves every step)
    int tok_branches = 1;   // final step: tok_cur gets this many copies, rows r * BP + bp (the imagenet projector feeds every CFG branch)
    void* X_next = nullptr; // not null (and not the final step): x0 = input_proj(x_t) of the NEXT evaluation, written over X's rows of this
    const void* in_w = nullptr;   // workgroup (flow_head:326) -- saves the next evaluation's prologue launch
    const void* in_b = nullptr;
    BD_STAMP_FIELD
};
int bdk_head_final(const HeadFinalArgs& a, hipStream_t st);

struct HeadYAllArgs {       // y_i = silu(time_embed(t_i) + cond_embed(c)) for EVERY evaluation of the schedule at once: depends on
    const void* cemb;       // (t_i, cond) only (flow_head:328-330), so it is computed once per AR step, not once per evaluation
    const void* temb;       // [n_evals][D] bf16
    void* y_all;            // out: [ceil(n_evals / G)] fragment-major bf16 matrices, one every G * Mpad * D elements: evaluation i = rows
                            //      (i % G) * Mpad .. of matrix i / G -- the A operand of ONE adaLN GEMM over G evaluations.  The last
                            //      matrix holds the n_evals % G left-over evaluations only, its row-block count rounded up to 8
    int M, D, RB, Mpad, n_evals;
    int G = 1;              // evaluations per adaLN GEMM (1: one [Mpad][D] matrix per evaluation)
    float* a8_scale = nullptr;   // not null: fp8-e4m3 operands (A8 layout, any G) + scales [ceil(n_evals / G) * G][Mpad] (a group's rows
                                 // contiguous).  A short last group's pad rows (row blocks past its evaluations, up to the rounded-up count)
                                 // are zero-filled by bdk_head_y_all itself, data and scales: the GEMM reads them
};
int bdk_head_y_all(const HeadYAllArgs& a, hipStream_t st);

struct FinalizeRowsArgs { Partial in; void* out; int M, N; BD_STAMP_FIELD };   // bf16 row-major out = bf16(sum of slabs + bias)
int bdk_finalize_rows(const FinalizeRowsArgs& a, hipStream_t st);
int bdk_set_rows_option(const char* name, int v);    // bd_rows.hip: "rows.ln_occ" 4|5, "rows.swiglu_t" 512|1024 (A/B switches)

struct InitLatentArgs { float* xt; const float* noise; long long noise_step_stride; const BdStepState* state; int n; };
int bdk_init_latent(const InitLatentArgs& a, hipStream_t st);    // x_0 = first draw of this AR step (sampling_x.py:60)

struct SwigluArgs {         // split-K fallback of the fused epilogue: act = silu(h1)*h2 from slabs
    Partial up;             // [.,Mpad,2F]; column order: interleaved ? packed pairs (16 gate | 16 up per 32) : [gate F | up F]
    void* act_frag;
    int M, F, RB, interleaved;
    BD_STAMP_FIELD
};
int bdk_swiglu_rows(const SwigluArgs& a, hipStream_t st);

// projector (modeling/utils.py:16-20)
struct ProjFc1Args {
    const float* tok;       // [BP][C] fp32 in {-1,0,1}
    const void* w;          // [D][C] bf16
    const void* b;          // [D]
    void* h_frag;           // out fragment-major bf16 [BPpad][D]
    int BP, D, C, RB;
};
int bdk_proj_fc1(const ProjFc1Args& a, hipStream_t st);

struct EmbedFinalizeArgs {  // model_input = fc2(h)+b (bf16) + pos (fp32), replicated to both CFG branches
    Partial fc2;            // [.,BPpad,D]
    const float* pos;       // [tokens][D] fp32 2-D sincos table in patch-raster order
    float* R;               // out: residual stream fp32 [Mpad][D]
    const BdStepState* state;
    int BP, P, D, branches;
};
int bdk_embed_finalize(const EmbedFinalizeArgs& a, hipStream_t st);

// LLM
struct RmsArgs {            // R (+= bf16(pending)) ; a = bf16(w * R*rsqrt(mean(R^2)+eps))          HF:59-64,294-323
    float* R;               // in/out fp32 [Mpad][D]
    Partial pend;           // p == nullptr -> none
    const void* w;          // [D] bf16
    void* a_frag;           // out fragment-major bf16 (may be null)
    float* hidden_out;      // final norm: w * normed, fp32 [M][D] (may be null)
    void* cond_frag;        // final norm: bf16(hidden + pos[step]) fragment-major (may be null)
    const float* pos;
    const BdStepState* state;
    int M, D, RB, P;
    float eps;
    float* a8_scale = nullptr;   // not null: a_frag goes out as fp8-e4m3 (A8 layout) + per-row scale
    int bf16_stream = 0;    // 1: prefill -- the residual stream is bf16 (bf16 embeds, no fp32 position table added): the branch add
                            //    and both RMSNorm products round to bf16 (HF:59-64 with a bf16 input)
};
int bdk_rms(const RmsArgs& a, hipStream_t st);

struct QkvPostArgs {        // q/k RMS-norm + RoPE + KV-cache append                               HF:241-262
    Partial qkv;            // [.,Mpad,(nh+2nkv)*128]
    const void* qn_w;       // [128] bf16
    const void* kn_w;
    const float* cos;       // [maxpos][128] fp32
    const float* sin;
    void* q_out;            // bf16 [Mpad][nh*128]
    void* k_cache;          // bf16 [nseq][nkv][Lmax][128]
    void* vt_cache;         // bf16 [nseq][nkv][128][Lmax]
    const BdStepState* state;
    int M, P, nh, nkv, Lmax;
    float eps;
    int rope_bf16 = 0;      // 1: prefill -- cos / sin cast to the bf16 hidden dtype (HF:137), every RoPE op rounds to bf16
};
int bdk_qkv_post(const QkvPostArgs& a, hipStream_t st);

// ---- class-conditional ImageNet transformer (imagenet_gen/src/layers_parallel.py, model_parallel.py) -------------
struct InProjFc1Args {      // MLPConnector.forward first half: w1 -> chunk -> silu(h1)*h2          model_parallel.py:73-75
    const float* tok;       // [rows][C] fp32 in {-1,0,1}
    const void* w;          // [2*hid][C] bf16 row-major (h1 rows, then h2 rows)
    const void* b;          // [2*hid] bf16
    void* act_frag;         // out fragment-major bf16 [rows_pad][hid]
    int rows, hid, C, RB;
};
int bdk_in_proj_fc1(const InProjFc1Args& a, hipStream_t st);

struct InRmsArgs {          // nn.RMSNorm on a bf16 residual stream: bf16( x * rsqrt(mean(x^2)+eps) * w ), w fp32
    float* R;               // residual stream, fp32 storage of bf16 values [Mpad][D]
    Partial pend;           // branch output to add first (bf16 + bf16 -> bf16); p == nullptr -> none
    int init_from_pend;     // 1: x = pend (proj_in output) instead of R + pend
    int renorm_to_R;        // 1: the normalised value becomes the residual stream (emb_norm, model_parallel.py:344)
    const float* w;         // [D] fp32
    void* a_frag;           // out fragment-major bf16 (may be null)
    float* hidden_out;      // fp32 [M][D] (may be null)
    void* cond_frag;        // bf16(normed + pos[step]) fragment-major (may be null)
    const float* pos;
    const BdStepState* state;
    int M, D, RB, P;
    float eps;
    int f32_stream = 0;     // 1: the FIRST forward_model call (class + query tokens, model_parallel.py:386-388): the class embedding is
                            // fp32, so the residual stream stays fp32 (fp32 + bf16 branch -> fp32, rms_norm of an fp32 tensor returns fp32)
};
int bdk_in_rms(const InRmsArgs& a, hipStream_t st);

struct InQkvPostArgs {      // interleaved 2-D RoPE on q,k + KV append, head_dim 64, MHA       layers_parallel.py:135-160,273-290
    Partial qkv;            // [.,Mpad,3*D], q | k | v thirds
    const float* rope;      // [tokens][32][2] (cos, sin)
    void* q_out;            // bf16 [Mpad][D], already multiplied by head_dim^-0.5 (exact in bf16)
    bf16_t* k_cache;        // [seq][nh][Lmax][64]
    bf16_t* v_cache;
    const BdStepState* state;
    int M, P, nh, Lmax;
};
int bdk_in_qkv_post(const InQkvPostArgs& a, hipStream_t st);

struct InAttnArgs {         // naive_attention with the reference's rounding points          layers_parallel.py:120-133
    const void* q;
    const bf16_t* k_cache;
    const bf16_t* v_cache;
    void* o_frag;           // out fragment-major bf16 [Mpad][D]
    const BdStepState* state;
    int nseq, P, nh, Lmax, RB;
    int causal = 0;         // 1: query i of the block sees keys <= past + i only (the causal part of attn_mask, model_parallel.py:90-101);
                            // 0: every cached key and the whole block (a decode block, and the first call's last, bidirectional block)
};
int bdk_in_attn(const InAttnArgs& a, hipStream_t st);

struct StepAdvanceArgs { BdStepState* state; int nseq, P; };
int bdk_step_advance(const StepAdvanceArgs& a, hipStream_t st);

int bdk_gfq_indices(const float* z, int* idx, int ntok, int ncb, int bits, hipStream_t st);
int bdk_gfq_codes(const int* idx, float* code, int ntok, int ncb, int bits, hipStream_t st);

// ---- bd_comm.hip : tensor-parallel exchange (all-reduce of a row-split Linear's fp32 partials, bias, one bf16 rounding)
struct bd_comm;
int bdk_tp_allreduce(bd_comm* c, const float* part, const void* bias, int rows, int N, Partial* res, hipStream_t st);
int bdk_comm_rank(const bd_comm* c);
int bdk_comm_size(const bd_comm* c);
// all-gather of a column-split Linear's bf16 output [rows][Nl] into every rank's [rows][Nl * size] copy at `dst_local` (inside the
// gather region, the same offset on every rank); hand-written exchange only (mode 0)
int bdk_tp_allgather(bd_comm* c, const void* slice, void* dst_local, int rows, int Nl, int N, hipStream_t st);
void* bdk_comm_gather_ptr(const bd_comm* c);
long long bdk_comm_gather_bytes(const bd_comm* c);
int bdk_comm_mode(const bd_comm* c);            // 0 hand-written exchange, 1 ncclAllReduce
// the exchange of a [rows][N] partial with phase 1 fused into the producing GEMM: fills the GEMM's push target (false: this
// shape / mode keeps the unfused form) and marks the NEXT bdk_tp_allreduce as pre-pushed once the GEMM confirms it took the target
// Tensor parallelism: where the finished fp32 partial of a ROW-split Linear goes (BD_EPI_F32).  With size > 1 the epilogue itself
// is phase 1 of the all-reduce (bd_comm.hip): the rows rank q reduces are written straight into q's staging row [rank] over the
// fabric (16 B system-scope stores), the rows this rank reduces into its own partial buffer as before -- the exchange kernel
// that follows only signals, waits and reduces.
struct BdTpPush {
    char* stage[8] = {};        // every rank's staging area (peer_data[q]); [rank] unused
    long long Us = 0;           // 32 B units per rank slice = rows_per_rank * N / 8
    int rank = 0, size = 0, rows_per_rank = 0;
    // sequence-parallel form: the LAST workgroup of the launch (device-wide arrival counter) tells every owner "this rank's partial
    // rows have landed" -- the owner's row kernel then finds the flag set instead of waiting for a signal its own first block sends
    int* done_cnt = nullptr;    // arrival counter (zero between launches); null: the consumer kernel signals (all-reduce form)
    int* sig[8] = {};           // where to write the epoch for owner q (q's BD_SP_P word of this rank; loop-back: the local word of q)
    const int* rc = nullptr;    // local replay counter
    int seq = 0;                // epoch = bd_sp_epoch_of(*rc, seq)
    int il = 0;                 // 1: sequence-parallel row ownership -- 8-row group g of the tensor belongs to rank g % size (its local
                                //    row (g / size) * 8 + row % 8), so that a rank owns the cond row AND the uncond row of the same patch
                                //    position (rows bp and BP + bp) and the final layer / sampler step need no exchange; 0: contiguous slices
};

// ---- sequence-parallel row kernels under tensor parallelism (bd_sp.hip, bd_comm.hip): one rank's view of the exchange.  A rank owns
// rows / size rows of the residual stream.  The row-split GEMM's epilogue pushes every owner its rows of the fp32 partial (BdTpPush);
// the row kernel of the owner sums them, applies gate / residual / LayerNorm / modulation for ITS rows only and pushes the bf16
// operand rows to every rank's landing buffer; the consuming GEMM polls the per-row flags before its first activation load.
#define BD_SP_MAXROWS 512
#define BD_SP_RC 0            /* replay counter: epoch = RC * 2^16 + sequence number of the hand-off inside the replayed graph (bd_common.h) */
#define BD_SP_P 8             /* [8]  "the partials of rank p are pushed" epochs */
#define BD_SP_G 16            /* [8]  reserved: "the adaLN columns of rank p are pushed" epochs for an all-gather pushed a group ahead and waited for
                                        by its first consumer (not built: needs links to measure) */
#define BD_SP_DONE 24         /* arrival counter of the pushing GEMM's workgroups (BdTpPush::done_cnt) */
#define BD_SP_H 32            /* [BD_SP_MAXROWS] "operand row m is pushed" epochs (also: final latent row bp) */
#define BD_SP_FLAG_INTS (32 + BD_SP_MAXROWS)
struct BdSpLink {
    char* stage[8] = {};       // every rank's staging area [src rank][local row][N] fp32 (uncached)
    char* hbuf[8] = {};        // every rank's operand landing buffer, fragment-major bf16 [Mpad][D] (cacheable: the GEMM re-reads it from L2)
    char* aux[8] = {};         // every rank's landing area of the final latent rows [BP][C] fp32 (uncached)
    int* spf[8] = {};          // every rank's sequence-parallel flag block
    int* spf_local = nullptr;
    int* err = nullptr;        // local error word (shared with the all-reduce exchange)
    long long stage_bytes = 0, hbuf_bytes = 0, aux_bytes = 0, timeout_ticks = 0;
    int rank = 0, size = 0;
    int loopback = 0;          // 1: no peers exist (one rank's critical path timed on one GPU): pushes go to scratch copies, every flag a
                               //    peer would write is written locally by the block that plays the same role
};
struct BdHWait {               // the consumer GEMM's wait for the pushed operand rows (bd_gemm_kernel.h prologue)
    const int* flags = nullptr;   // local BD_SP_H block; null: no wait
    const int* rc = nullptr;      // local replay counter
    int* err = nullptr;
    long long timeout_ticks = 0;
    int seq = 0, n = 0;           // epoch = bd_sp_epoch_of(*rc, seq); rows to wait for
    int inv = 0;                  // 0: wave 0 invalidates (buffer_inv sc0 sc1) before the barrier; 1: every wave after it
};
int bdk_sp_begin(bd_comm* c, hipStream_t st);          // once per replayed graph / eager sequence: RC += 1, sequence numbers restart
int bdk_sp_next_seq(bd_comm* c);                        // the next hand-off's sequence number (1 .. BD_SP_SEQ_MAX); -1: exhausted
bool bdk_sp_link(bd_comm* c, BdSpLink* out);            // false: no sequence-parallel exchange on this communicator
bool bdk_sp_hwait(bd_comm* c, int seq, int rows, BdHWait* out);
int bdk_sp_wait_rows(bd_comm* c, int seq, int rows, hipStream_t st);   // the wait as its own tiny kernel ("tune.sp_wait" = 0)
int bdk_sp_selftest(bd_comm* c, int round, int rows, int D, int wait_in_check, int* bad_dev, hipStream_t st);   // bd_sp.hip
long long bdk_sp_hbuf_bytes(const bd_comm* c);
void* bdk_sp_hbuf(const bd_comm* c);
void bdk_gemm_set_hwait(const BdHWait* w);              // bd_gemm.hip: the NEXT bdk_gemm / bdk_gemm8 call waits in its prologue
bool bdk_gemm_take_hwait(BdHWait* out);
void bdk_comm_count_exchange(bd_comm* c);

struct LnModSpArgs {           // ln_mod for this rank's rows only, fed by the peers' partial pushes, feeding every rank's operand buffer
    LnModArgs ln;              // X, ada, offsets, LayerNorm affine, M, D, RB, eps (pend / h_frag / a8_scale unused)
    BdSpLink L;
    const float* part = nullptr;   // this rank's own fp32 partial [Mpad][D] of the pending row-split Linear (rows at their global index); null: no pending branch
    const void* bias = nullptr;    // its bias [D] bf16 (added once, by the reducing rank) or null
    int seq_p = 0, seq_h = 0;      // sequence numbers of the partial hand-off (0: none) and of this kernel's operand rows
    int rows_local = 0;
    int signal_p = 1;              // 0: the pushing GEMM's last workgroup already told the owners (BdTpPush::done_cnt)
};
int bdk_ln_mod_sp(const LnModSpArgs& a, hipStream_t st);
struct HeadFinalSpArgs {
    HeadFinalArgs f;           // pend unused: the last block's w2 partials come through the staging area
    BdSpLink L;
    const float* part = nullptr;
    const void* bias = nullptr;
    int seq_p = 0, seq_f = 0;  // seq_f: the final evaluation's latent rows (pushed to every rank's aux area); 0 otherwise
    int bp_local = 0;          // patch positions this rank owns (BP / size)
    int signal_p = 1;
};
int bdk_head_final_sp(const HeadFinalSpArgs& a, hipStream_t st);
// The Qwen3 decode step on the same hand-off (round 6): residual stream rows OWNED by a rank, RMSNorm on the owner, bf16 operand rows to
// every rank (rms_sp_kernel); the step's final-norm rows travel as fp32 and every rank finishes hidden state / next condition itself.
struct RmsSpArgs {
    RmsArgs r;                 // R (fp32 residual, rows at their global index), w, eps, M, D, RB, P, pos, state; pend / a_frag unused;
                               // hidden_out / cond_frag are written by sp_final_rows_kernel, not here
    BdSpLink L;
    const float* part = nullptr;   // this rank's own fp32 partial [Mpad][D] of the pending row-split Linear (o_proj / down_proj); null: none
    int seq_p = 0, seq_h = 0;      // hand-off of the partials (0: none) / of this kernel's rows
    int rows_local = 0;
    int signal_p = 1;
    int final_rows = 0;            // 1: the step's final norm -- fp32 rows w * normed, row-major [M][D], instead of bf16 operand fragments
    long long final_off = 0;       //    ... at this byte offset of the landing buffers (behind the operand region)
};
int bdk_rms_sp(const RmsSpArgs& a, hipStream_t st);
struct SpFinalRowsArgs {       // every rank, after the final rms_sp: hidden_out = row, cond_frag = bf16(row + pos)   (rms_kernel's tail, HF:59-64, t2i:244-245)
    BdHWait w;                 // per-row flags of the final hand-off
    const float* rows;         // local landing buffer, fp32 row-major [M][D]
    float* hidden_out; void* cond_frag; const float* pos; const BdStepState* state;
    int M, D, RB, P;
};
int bdk_sp_final_rows(const SpFinalRowsArgs& a, hipStream_t st);

struct TokFinishArgs {         // after the final evaluation: every rank assembles pred / tokens of ALL patch positions from the gathered latent rows
    BdSpLink L;
    int seq_f = 0;
    float* xt; float* pred_out; float* tok_cur; float* tok_all;
    const BdStepState* state;
    int BP, C, T, P, tok_branches;
};
int bdk_tok_finish(const TokFinishArgs& a, hipStream_t st);

bool bdk_tp_push_target(bd_comm* c, int rows, int N, BdTpPush* out);
bool bdk_tp_push_target_sp(bd_comm* c, int rows, int N, int seq, BdTpPush* out);   // seq > 0: the GEMM's last workgroup signals the owners
void bdk_tp_mark_prepushed(bd_comm* c);
void bdk_gemm_set_push(const BdTpPush* t);      // bd_gemm.hip
bool bdk_gemm_push_used();
bool bdk_gemm_claim_push(int epi, int RB, int N, BdTpPush* out);

// ---- bd_attn.hip
struct HeadAttnArgs {       // DiT attention over one patch (seq = P = 64 or 16), non-causal      flow_head:192-220
    Partial qkv;            // [.,Mpad,3D]
    void* o_frag;           // out fragment-major bf16 [Mpad][D]
    int nseq, nhead, D, RB, P;
    int dh;                 // head dim: 128, or 64 with P = 16 (imagenet head)
    BD_STAMP_FIELD
};
int bdk_head_attn(const HeadAttnArgs& a, hipStream_t st);

struct LlmAttnArgs {        // block-bidirectional decode attention over the static KV cache
    const void* q;          // bf16 [Mpad][nh*128]
    const void* k_cache;
    const void* vt_cache;
    float* o_part;          // [nseq][nkv][splits][G*P][128] fp32 (un-normalised)
    float* ml_part;         // [nseq][nkv][splits][G*P][2]
    void* o_frag;           // out fragment-major bf16 [Mpad][nh*128]
    const BdStepState* state;
    int nseq, P, nh, nkv, Lmax, splits, RB;
    int causal = 0;         // 1: query p sees keys <= kv_len + p (the prompt call, t2i_pipeline.py:199-203); 0: all kv_len + P keys
};
int bdk_llm_attn(const LlmAttnArgs& a, hipStream_t st);
