/* libbitdance_hip.so -- C ABI of the MI355X-native BitDance generation hot path.
 *
 * The reference (shallowdream204/BitDance) is pure Python/PyTorch and has no FFI; its "operator API" for the
 * hot path is the set of Python seams that BitDanceT2IPipeline.gen_image calls
 * (modeling/t2i_pipeline.py:170,246,249,261-268).  Each entry point below names the reference interface it
 * replaces.  Conventions: plain C, device pointers owned by the caller (torch tensors kept alive by the
 * Python wrapper), stream-ordered and asynchronous, no allocation and no host sync inside a call (so every
 * call is hipGraph-capturable), int return 0 = ok / negative = error with text in bd_last_error(), not
 * thread-safe per context.  `stream` is a hipStream_t passed as void*.
 */
#ifndef BITDANCE_HIP_H
#define BITDANCE_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

typedef struct bd_ctx bd_ctx;
typedef struct bd_comm bd_comm;      /* tensor-parallel exchange state of one rank (below) */

int bd_version(void);
const char* bd_last_error(void);
/* Return codes: 0 = ok; BD_ERR_UNSUPPORTED = a shape / mode the kernels do not cover, refused by host-side validation BEFORE anything
 * was launched (the convolution / GroupNorm entries: the caller -- autoencoder.VQModel -- may answer it, and only it, by running the
 * torch module, which is what the reference runs: autoencoder.py:129-277); any other negative value = a failed launch or a misuse,
 * which must propagate. */
#define BD_ERR_UNSUPPORTED (-22)

/* ---- weight / activation layout conversion (load time; replaces nothing in the reference: the reference
 *      keeps nn.Linear weights [N][K] row-major, t2i_pipeline.py:50-74 -- we re-pack once into MFMA order) */
int bd_pack_weight(void* dst_packed, const void* src_bf16, int rows, int K, int dst_row0, int dst_rows_total, void* stream);
/* packed order in HBM: 0 (default) = panel-major; 1 = stage-major, the whole grid reads one contiguous window per
 * 64-deep K stage (an experiment: measured identical on MI355X).  Process-wide; set before packing, weights packed under one setting must be used under it. */
int bd_set_weight_layout(int stage_major);
/* process-wide A/B switches of the GEMM kernels, for measurement (tools/, bench.py --gemm-opt); every setting computes the same
 * values.  "wide.ring" 2|3 = weight stages a wave of the 256-row kernel keeps in flight; "wide.xcd" -1|0|1 = row tiles of one
 * weight slice on one XCD (by shape / off / on); "wide.keep" -1|0|1 = default-policy instead of non-temporal weight loads when
 * several row tiles read a slice; "tile" 0|1|2|3|4 = from 1024 rows on N >= 4096: the 256-row kernel / the LDS-tiled 256 x 256
 * kernel with its operand fetch chosen by shape / register-staged fetch forced / LDS-DMA fetch forced / W straight into registers
 * forced (bd_gemm_tile.hip);
 * "tile.minrb" 8, 16, ... = row blocks from which the tiled kernel takes over (default 32 = 1024 rows);
 * "red.first" 0|1 = two-slice in-launch reduction with the ticket taken first (1, default: only the first arriver parks its
 * accumulators) or both slices parking (0); "rows.ln_occ" 4|5 = ln_mod's register bound (one / two 640-thread workgroups per CU,
 * default 5); "rows.swiglu_t" 512|1024 = thread cap of swiglu_rows (default 512); "half" 0|1|2 = the 256 x 128-tile kernel of the
 * 512-row passes (bd_gemm_half.hip): off / where the launch code asks for it (nwaves + 4096; default) / for every shape it can run;
 * "half.form" 1|0 = its weight fragments straight into registers (default) / both operands through LDS. */
int bd_set_gemm_option(const char* name, int value);
int bd_pack_weight_swiglu(void* dst_packed, const void* gate_bf16, const void* up_bf16, int F, int K, void* stream);
int bd_rows_to_frag(void* dst_frag, const void* src, int src_is_fp32, int M, int K, int row_blocks, void* stream);

/* ---- F.linear under bf16 autocast (flow_head_parallel_x.py:326-339, HF modeling_qwen3.py:81-83,252-279).
 *      out_partial: [splitk][row_blocks*32][N] fp32 slabs, summed (+bias, bf16 rounding) by the consumer.
 *      nwaves = waves per workgroup (2, 4, 8) [+ 16 * ring, ring in {2,3,4} = K stages a wave keeps in flight]
 *      [+ 4096: >= 512 rows, < 1024 rows: the 256 x 128-tile kernel, one K slice with a rounded output or any number of slabs]. */
int bd_gemm_partial(const void* a_frag, int row_blocks, const void* w_packed, int N, int K, int splitk, int nwaves,
                    float* out_partial, void* stream);
/* The same Linear with the split-K slices reduced INSIDE the launch (the last-arriving slice of each tile sums the
 * others' slabs, adds the bias and rounds once): out_bf16 [row_blocks*32][N] row-major is exactly what F.linear returns
 * under autocast.  scratch: splitk x rows x N fp32 of slab space (layout private to the kernel); counters: one int per output tile, zero on entry, zero on exit. */
int bd_gemm_bf16(const void* a_frag, int row_blocks, const void* w_packed, const void* bias_bf16, int N, int K, int splitk,
                 int nwaves, float* scratch, int* counters, void* out_bf16, void* stream);
/* The same with a finished fp32 result and no bias / rounding: one tensor-parallel rank's partial of a row-split Linear
 * (the reference has no tensor parallelism; this is the Megatron-style split SURVEY.md 8e prescribes for wo / w2 / o_proj /
 * down_proj).  nwaves may carry + 256 for the 2-panels x 2-K-parts workgroup shape (bd_gemm.hip). */
int bd_gemm_f32(const void* a_frag, int row_blocks, const void* w_packed, int N, int K, int splitk, int nwaves, float* scratch,
                int* counters, float* out_f32, void* stream);
/* Linear -> chunk(2) -> silu(h1)*h2 (flow_head:250-251) / down_proj input act_fn(gate)*up (HF:82) */
int bd_gemm_swiglu(const void* a_frag, int row_blocks, const void* w_packed_pairs, const void* bias_packed, int N2, int K,
                   int nwaves, void* act_frag, void* stream);
/* the same over `splitk` K slices reduced inside the launch (the head's w1 at 128 rows: 2 slices); scratch [splitk][rows][N2] fp32,
 * counters: zeroed ints, one per output tile (left zero) */
int bd_gemm_swiglu_splitk(const void* a_frag, int row_blocks, const void* w_packed_pairs, const void* bias_packed, int N2, int K,
                          int splitk, int nwaves, float* scratch, int* counters, void* act_frag, void* stream);

/* ---- fp8-e4m3 weight storage (BASELINE config 5; a separate precision mode).  src: OCP e4m3 bytes [rows][K] row-major,
 *      already divided by the per-output-channel scale; the GEMM converts to bf16 in registers and multiplies the fp32 scale
 *      (wscale[N], packed row order) into the accumulator.  epi: 0 fp32 slabs [splitk][rows][N] into out, 1 fused SwiGLU
 *      (out = activation fragments), 2 bf16(+bias), 3 finished fp32 sum.  Context form: int "wdtype" = 1 and a "<weight key>_s"
 *      scale pointer next to every streamed weight. */
int bd_pack_weight8(void* dst_packed, const void* src_fp8, int rows, int K, int dst_row0, int dst_rows_total, void* stream);
int bd_pack_weight8_swiglu(void* dst_packed, const void* gate_fp8, const void* up_fp8, int F, int K, void* stream);
int bd_gemm_w8(const void* a_frag, int row_blocks, const void* w8_packed, const float* wscale, const void* bias_bf16, int N, int K,
               int splitk, int nwaves, int epi, float* scratch, int* counters, void* out, void* stream);
/* fp8 x fp8 on the block-scaled fp8 matrix pipe ("wdtype" 2): the GEMMs a row kernel feeds (head adaLN / qkv / w1, LLM q/k/v and
 * gate/up) take their activations as e4m3 with one fp32 scale per ROW (emitted by that row kernel) next to the e4m3 weights with one
 * scale per output channel; their weights are packed in the K = 64 operand order by these entry points.  bd_quant_rows8 +
 * bd_gemm_w8a8: the same arithmetic standalone (tests).  A separate precision mode with its own oracle policy ("fp8wa"). */
int bd_pack_weight8k(void* dst, const void* src_fp8, int rows, int K, int dst_row0, int dst_rows_total, void* stream);
int bd_pack_weight8k_swiglu(void* dst, const void* gate_fp8, const void* up_fp8, int F, int K, void* stream);
int bd_quant_rows8(void* a8, float* ascale, const float* src_rows_f32, int M, int K, int RB, void* stream);
int bd_gemm_w8a8(const void* a8, const float* ascale, int RB, const void* w8k, const float* wscale, const void* bias, int N, int K, int S, int nw,
                 int epi, float* scratch, int* counters, void* out, void* stream);

/* ---- context: named ints / floats / device pointers, then finalize.  Keys are listed in INTEGRATION.md (appendix) / bd_api.hip kIntKeys, kPtrKeys; an unknown key is an
 *      error (-1, text in bd_last_error()), never a silent default. */
bd_ctx* bd_ctx_create(void);
void bd_ctx_destroy(bd_ctx* c);
int bd_ctx_set_int(bd_ctx* c, const char* key, long long v);
int bd_ctx_set_float(bd_ctx* c, const char* key, double v);
int bd_ctx_set_ptr(bd_ctx* c, const char* key, const void* device_ptr);
int bd_ctx_set_comm(bd_ctx* c, bd_comm* comm);        /* before finalize: this context is rank comm.rank of comm.size;
                                                         weights handed over are this rank's slices (engine.py) */
int bd_ctx_set_tp(bd_ctx* c, int rank, int size);     /* plan a rank's context WITHOUT a communicator (inspection, host tests) */
int bd_ctx_finalize(bd_ctx* c);                       /* validates dims, plans the workspaces */
int bd_ctx_ws_count(bd_ctx* c);                       /* workspaces the caller must allocate (zero-filled) ... */
const char* bd_ctx_ws_name(bd_ctx* c, int i);         /* ... and hand back with bd_ctx_set_ptr(name, ptr) */
long long bd_ctx_ws_bytes(bd_ctx* c, int i);
int bd_ctx_bind(bd_ctx* c);                           /* after all workspace pointers are set */

/* ---- DiffHead.sample (flow_head_parallel_x.py:107-120 -> sampling_x.py:44-97).
 * scalars: [n_steps+1][6] = {t, dt, den=clamp_min(1-t,.05), var, 1-t, noise_scale} per eval, computed by the
 * host exactly as the reference computes its 0-dim tensors (last row: t=1-last_step, dt=last_step). */
int bd_head_set_schedule(bd_ctx* c, int n_steps, const float* scalars, float cfg);
int bd_head_set_cfg(bd_ctx* c, float cfg);            /* the guidance scale alone (imagenet linear ramp, model_parallel.py:356-365):
                                                         eager launches use it at once; captured graphs keep theirs */
int bd_head_sample(bd_ctx* c, void* stream);          /* cond (ws head.cond_frag) + noise -> head.pred / tokens */
int bd_head_cond(bd_ctx* c, void* stream);            /* cond_embed(c) once per AR step (value-identical hoist) */
int bd_head_eval(bd_ctx* c, int i, void* stream);     /* one TransEncoder.forward (:325-342) + sampler step i */

/* ---- MLPconnector.forward (modeling/utils.py:16-20) + "+ pos_embed" (t2i_pipeline.py:249-253) */
int bd_projector(bd_ctx* c, void* stream);

/* ---- Qwen3Model.forward for one P-token block against the KV cache, cond+uncond batched
 *      (t2i_pipeline.py:261-268; HF modeling_qwen3.py:367-427) */
int bd_llm_step(bd_ctx* c, void* stream);

/* ---- the AR step as hipGraphs: phase 0 = head sample (+sign/tokens), phase 1 = projector + LLM step + advance */
int bd_graph_capture(bd_ctx* c, int phase, void* stream);
int bd_graph_launch(bd_ctx* c, int phase, void* stream);
int bd_step_reset(bd_ctx* c, const int* kv_len, int nseq, void* stream);   /* step = 0, kv_len[] after prefill */

/* ---- tensor parallelism over xGMI (no counterpart in the reference, whose multi-GPU mode is replicas,
 *      eval/eval_dpg.py:25-29).  One bd_comm per process/GPU.  bd_comm_create allocates (once, not on the hot path) the
 *      exported staging/result buffer for exchanges of up to max_elems = rows*N elements and an uncached flag block;
 *      handles travel through the host's process group (128 bytes per rank); peers are mapped with bd_comm_open_peer.
 *      The exchange itself -- all-reduce of the ranks' fp32 partials [rows][N], + bias, one rounding to bf16, result
 *      replicated bit-identically on every rank -- is one kernel launch inside bd_head_eval / bd_llm_step (captured in the
 *      step graphs), or standalone through bd_comm_allreduce.  bd_comm_set_rccl switches the exchange to ncclAllReduce of
 *      the same fp32 partials (host passes the RCCL communicator and the address of ncclAllReduce): fallback and baseline. */
bd_comm* bd_comm_create(int rank, int size, long long max_elems);
/* ... plus an all-gather region of gather_bytes in the same exported allocation: where the COLUMN-split adaLN projection
 * (flow_head_parallel_x.py:331: each rank computes N / size of its 71 680 output columns) is assembled on every rank by pushes
 * (bd_comm_allgather standalone; inside the step when the context finds "head.ada_w_l" and the region is large enough) */
bd_comm* bd_comm_create2(int rank, int size, long long max_elems, long long gather_bytes);
/* ... plus, with hbuf_bytes > 0, the buffers of the SEQUENCE-PARALLEL form of the exchange (round 5; csrc/bd_sp.hip; "tp.seq" in the
 * context): what one DiT block computes (flow_head_parallel_x.py:242-252: x += gate * branch; h = LN(x) * (1 + scale) + shift) is done
 * for rows / size rows per rank -- the row-split GEMM's epilogue pushes each owner its rows of the fp32 partial, the owner's row kernel
 * sums them in rank order (+ bias, one bf16 rounding: the same value the all-reduce form computes), normalises / modulates and pushes
 * the bf16 operand rows into EVERY rank's landing buffer (hbuf_bytes = rows x D x 2, a second, cacheable exported allocation: the
 * consuming GEMM re-reads it from L2), and the consuming GEMM polls per-row flags after issuing its first weight loads.  No stand-alone
 * exchange kernel is left in the evaluation.  A 64 KiB landing area for the final latent rows sits behind the gather region. */
bd_comm* bd_comm_create3(int rank, int size, long long max_elems, long long gather_bytes, long long hbuf_bytes);
int bd_comm_ipc_handles3(bd_comm* c, void* out192);                  /* 3 x hipIpcMemHandle_t: data, flags, operand landing buffer */
int bd_comm_open_peer3(bd_comm* c, int peer, const void* handles192);
int bd_comm_set_peer_ptrs3(bd_comm* c, int peer, void* data, void* flags, void* hbuf);   /* peers inside this process */
void* bd_comm_local_hbuf(bd_comm* c);
long long bd_comm_hbuf_bytes(bd_comm* c);
/* Self-test of the sequence-parallel hand-off itself (cacheable landing buffer written by the peers' sc0 sc1 stores, read back after the
 * GEMM prologue's flag wait + invalidate): one round = this rank pushes its own rows of a (row, unit, round)-dependent pattern into every
 * rank's buffer and raises the row flags, then 64 workgroups wait like the consuming GEMM and compare every 16 B unit of every row;
 * *bad_dev (a device int the caller zeroes) += mismatches.  Every rank calls it with the same arguments, rounds separated by a barrier
 * of the ranks; wait_in_check = 0 puts the wait into a one-workgroup kernel (ranks sharing one GPU).  No reference counterpart
 * (the reference has no tensor parallelism: eval/eval_dpg.py:25-29 runs replicas). */
int bd_comm_sp_selftest(bd_comm* c, int round, int rows, int D, int wait_in_check, int* bad_dev, void* stream);
/* ONE rank of a `size`-rank group alone on this GPU: the peers' buffers become scratch copies, every flag a peer would write is written
 * locally -- the rank's launches, weight shards, pushes and waits minus the links, for timing its critical path on one GPU
 * (tools/head_sweep.py --tp-shard).  The results are meaningless (the peers contribute zeros). */
int bd_comm_set_loopback(bd_comm* c);
long long bd_comm_prepushed(bd_comm* c);                             /* exchanges whose reduce-scatter push ran in the producing GEMM's epilogue */
void* bd_comm_gather_ptr(bd_comm* c);                                /* this rank's copy of the region (null: none) */
long long bd_comm_gather_bytes(bd_comm* c);
int bd_comm_allgather(bd_comm* c, const void* slice_bf16, int rows, int Nl, void* stream);   /* [rows][Nl] of every rank -> [rows][Nl * size] */
void bd_comm_destroy(bd_comm* c);
int bd_comm_ipc_handles(bd_comm* c, void* out128);                   /* 2 x hipIpcMemHandle_t: data, flags */
int bd_comm_open_peer(bd_comm* c, int peer, const void* handles128);
int bd_comm_set_peer_ptrs(bd_comm* c, int peer, void* data, void* flags);   /* peers inside this process (protocol tests) */
void* bd_comm_local_data(bd_comm* c);
void* bd_comm_local_flags(bd_comm* c);
int bd_comm_set_rccl(bd_comm* c, void* nccl_comm, void* nccl_allreduce_fn);
int bd_comm_set_timeout(bd_comm* c, double seconds);                 /* budget of every in-kernel wait (default 20 s) */
int bd_comm_set_fences(bd_comm* c, int on);                          /* 1: system-scope fences around every flag (default 0: the payload is
                                                                        write-through stores drained before the flag, read with system-scope loads) */
int bd_comm_mark_prepushed(bd_comm* c);                              /* the next exchange skips its push phase: the producing GEMM's epilogue
                                                                        wrote every peer's slice of the partial into that peer's staging row */
int bd_comm_reset(bd_comm* c);                                       /* all ranks, between host barriers: clear flags / epochs / error */
int bd_comm_error(bd_comm* c);                                       /* after a sync: bit p = this rank's wait for peer p timed out;
                                                                        bit 8+r = rank r reported a timed-out wait (all ranks raise together) */
int bd_comm_info(bd_comm* c, long long* out4);                       /* {exchange buffer uncached, flag block uncached, mode (0 hand-written,
                                                                        1 ncclAllReduce), capacity in elements}: the host must not run the
                                                                        hand-written exchange ACROSS devices on cached (coarse-grained) memory */
long long bd_comm_exchanges(bd_comm* c);                             /* exchange launches issued so far (reporting) */
int bd_comm_allreduce(bd_comm* c, const float* part, const void* bias_bf16, int rows, int N, void** out_ptr, int* out_is_fp32,
                      void* stream);

int bd_comm_copy_out(bd_comm* c, void* dst, long long bytes, int from_result, void* stream);   /* read a standalone exchange back:
                                                     from_result 0 fp32 staging, 1 bf16 result, 2 the all-gather region */

/* ---- GFQ bit <-> index math of the ImageNet tokenizer (imagenet_gen/src/gfq.py:152-160,217-239): integer, bit exact.
 *      z/codes: [ntok][ncodebooks*bits] fp32 channels-last; idx: [ntok][ncodebooks] int32 (LSB = first channel). */
int bd_gfq_indices(const float* z, int* idx, int ntok, int ncodebooks, int bits, void* stream);
int bd_gfq_codes(const int* idx, float* codes, int ntok, int ncodebooks, int bits, void* stream);

/* ---- conv decoder of the binary tokenizer (/root/reference/modeling/vision_encoder/autoencoder.py:129-277:
 *      Decoder.forward = conv_in, ResBlocks (GroupNorm -> swish -> conv3x3, x2, + shortcut), AdaptiveGroupNorm, depth-to-space
 *      upsamplers, norm_out -> swish -> conv_out).  Kernel-level entry points; the module that sequences them and carries the
 *      checkpoint surface is bitdance_amd/ae_native.py.  Activations NHWC; a convolution input is bf16 with a one-pixel zero
 *      border [n][H+2][W+2][C] ("padded").
 *      bd_conv: 3x3 (taps 9, padded input) or 1x1 (taps 1, unpadded input) as an implicit GEMM on the matrix pipe; w_packed =
 *      bd_pack_weight of the [Cout rounded up to 256][taps * Cin] matrix with K ordered (ky, kx, ci); conv output = bf16(acc + bias)
 *      (F.conv2d under bf16 autocast), then out_mode 0: unpadded NHWC (+ residual fp32 / bf16 -> fp32 / bf16), 1: depth-to-space
 *      (autoencoder.py:198-230, DCR) bf16 NHWC [n][2H][2W][Cout/4], 2: fp32 NCHW image, 3: padded bf16 NHWC. */
int bd_conv(const void* in, const void* w_packed, const void* bias_bf16, const void* res, int res_f32, void* out, int out_mode, int out_f32,
            int n, int H, int W, int Cin, int Cout, int taps, void* stream);
/* the same with a stride: 1, or 2 for a 3x3 / padding-1 kernel -- the Encoder's down-sampling convolution (autoencoder.py:59-127:
 * nn.Conv2d(c, c, 3, stride=2, padding=1)); H, W are the OUTPUT size, the padded input is [n][2H+2][2W+2][Cin] */
int bd_conv_strided(const void* in, const void* w_packed, const void* bias_bf16, const void* res, int res_f32, void* out, int out_mode,
                    int out_f32, int n, int H, int W, int Cin, int Cout, int taps, int stride, void* stream);
/* GroupNorm(32) statistics (nn.GroupNorm(32, C, eps), autoencoder.py:9-11) of an unpadded NHWC tensor: stats [n][32][2] = (mean, rstd);
 * `partial` = scratch [n][ceil(HW / 256)][32][2] fp32; deterministic (no atomics) */
int bd_gn_stats(const void* x, int x_f32, float* partial, float* stats, int n, int HW, int C, float eps, void* stream);
/* y = (x - mean) rstd [gamma, beta] [AdaptiveGroupNorm scale, bias per (image, channel): autoencoder.py:251-277] [swish], all fp32,
 * out_mode 0: padded bf16 NHWC, 1: unpadded fp32, 2: unpadded bf16; stats == NULL: y = x (a cast / re-layout) */
int bd_gn_apply(const void* x, int x_f32, const float* stats, const float* gamma, const float* beta, const float* scale, const float* bias,
                void* out, int out_mode, int swish, int n, int H, int W, int C, void* stream);
int bd_tokens_to_padded(const float* z_nchw, void* out_padded_bf16, int n, int C, int H, int W, void* stream);

/* ---- measurement support (bench.py): in-situ HIP-event timing of every weight-streaming GEMM launch (eager mode) */
int bd_prof_enable(bd_ctx* c, int on);
int bd_prof_count(bd_ctx* c);
int bd_prof_get(bd_ctx* c, int i, char* name64, float* ms, double* weight_bytes);
int bd_gemm_config(bd_ctx* c, const char* gemm_name, int* splitk, int* nwaves);
/* pure HBM read stream with the GEMM's load pattern (16 B/lane non-temporal): the practical read roofline */
int bd_probe_read(const void* src, long long bytes, int blocks, void* sink, void* stream);

#ifdef __cplusplus
}
#endif
#endif
