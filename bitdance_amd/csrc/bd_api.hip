// C ABI + native orchestration of the BitDance AR step (see include/bitdance_hip.h).
// Everything here is host code: it sequences the kernels of bd_gemm/bd_rows/bd_attn on the caller's stream and
// captures the two phases of an AR step into hipGraphs.  No allocation, no sync inside the step.
#include <algorithm>
#include <map>
#include <string>
#include <vector>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <stdexcept>

#include "bd_common.h"
#include "bd_kernels.h"
#include "../../include/bitdance_hip.h"

static thread_local std::string g_err;
static int fail(const std::string& m) { g_err = m; return -1; }

#ifdef BD_GEMM_STAMP
// ---- launch anatomy (measurement build only, tools/launch_anatomy.py): every stamped launch gets 8 words per workgroup of the
// caller's device buffer; the host keeps (name, offset, workgroups) per launch.  Not part of the C ABI of the product library.
namespace {
struct StampRec { std::string name; long long off; int nwg; };
unsigned long long* g_stamp_buf = nullptr;
long long g_stamp_cap = 0, g_stamp_used = 0;                // in 8-byte words
std::vector<StampRec> g_stamp_recs;
thread_local std::string g_stamp_label = "gemm";
}
unsigned long long* bdk_stamp_next(const char* name, int nwg) {
    if (!g_stamp_buf || g_stamp_used + (long long)nwg * 8 > g_stamp_cap) return nullptr;
    unsigned long long* r = g_stamp_buf + g_stamp_used;
    g_stamp_recs.push_back({name, g_stamp_used, nwg});
    g_stamp_used += (long long)nwg * 8;
    return r;
}
void bdk_stamp_label(const char* name) { g_stamp_label = name; }
const char* bdk_stamp_current_label() { return g_stamp_label.c_str(); }
extern "C" {
int anatomy_begin(void* dev_buf, long long bytes) { g_stamp_buf = (unsigned long long*)dev_buf; g_stamp_cap = bytes / 8; g_stamp_used = 0; g_stamp_recs.clear(); return 0; }
int anatomy_count(void) { return (int)g_stamp_recs.size(); }
int anatomy_get(int i, char* name64, long long* off_words, int* nwg) {
    if (i < 0 || i >= (int)g_stamp_recs.size()) return -1;
    std::strncpy(name64, g_stamp_recs[i].name.c_str(), 63); name64[63] = 0;
    *off_words = g_stamp_recs[i].off; *nwg = g_stamp_recs[i].nwg;
    return 0;
}
}
#endif
void bdk_set_error(const std::string& m) { g_err = m; }      // bd_comm.hip reports through the same bd_last_error()

#define BD_TRY(expr)                                                                     \
    do {                                                                                 \
        int _r = (expr);                                                                 \
        if (_r != 0) return fail(std::string(#expr) + " failed with " + std::to_string(_r)); \
    } while (0)

// nw = waves per workgroup, kw = waves sharing one 32-column panel (split-K inside the workgroup, bd_gemm.hip), ring = K
// stages in flight per wave, S = split-K over the grid
struct GemmCfg { int S = 1, nw = 4, ring = 2, kw = 1, half = 0; int code() const { return nw + 16 * ring + 256 * (kw - 1) + 4096 * half; } };   // half: the 256 x 128-tile kernel (bd_gemm_half.hip)

struct bd_ctx {
    std::map<std::string, long long> I;
    std::map<std::string, double> F;
    std::map<std::string, const void*> P;
    std::vector<std::pair<std::string, long long>> ws;
    std::map<std::string, GemmCfg> g;
    std::vector<SamplerScalars> sched;
    bool finalized = false, bound = false;
    bool prof_on = false;
    struct ProfRec { std::string name; hipEvent_t e0, e1; double bytes; };
    std::vector<ProfRec> prof;
    hipGraphExec_t gexec[2] = {nullptr, nullptr};
    hipGraph_t graph[2] = {nullptr, nullptr};
    bool y_ready = false;                 // head.y_all holds y_i of every evaluation for the current cond (set by head_cond)

    // derived
    int B = 1, branches = 2, Pn = 64, BP = 64, M = 128, RB = 4, RBp = 2, Mpad = 128, BPpad = 64;
    int prows = 64;                       // rows the projector runs over: BP, or M with "proj.rows_all" (imagenet: every CFG branch's tokens)
    int hD = 0, hC = 0, hDz = 0, hH = 0, hNB = 0, hNA = 0, hNada = 0, hT = 0;
    int hMlp = 0;                         // "head.variant" = 1: MLP head of the 1x ImageNet models (imagenet_gen/src/diff_head.py), no attention
    int lD = 0, lL = 0, lnh = 0, lnkv = 0, lF = 0, lLmax = 0, lsplits = 8, lNqkv = 0;
    int ldh = 128, lvariant = 0;          // lvariant 1: imagenet transformer (head_dim 64, MHA, 2-D RoPE, bf16 residual)
    bool has_head = false, has_llm = false, has_proj = false;
    // tensor parallelism: weights arrive pre-sliced (engine.py); every dim below with an `l` suffix is this rank's
    bd_comm* comm = nullptr;
    bool wfp8 = false;                    // "wdtype" >= 1: every streamed weight is fp8-e4m3 + "<key>_s" scales (bd_gemm8.hip)
    bool fp8a = false;                    // "wdtype" = 2: ALSO fp8 activations, on the fp8 matrix pipe, for the GEMMs a row kernel feeds
                                          //   (head.ada / qkv / w1, llm.qkv / gu: their weights are packed for it, bd_pack_weight8k)
    int tp = 1, tpr = 0;
    int hDl = 0, hHl = 0, lnhl = 0, lnkvl = 0, lFl = 0;
    // "tune.ada_group": evaluations whose adaLN projections run as ONE GEMM (head_ada_group); 1 = one GEMM per evaluation
    int adaG = 1;
    // "tp.seq": sequence-parallel row kernels under tensor parallelism (bd_sp.hip): a rank owns rows / tp rows of the head's residual
    // stream; no stand-alone exchange kernel inside an evaluation.  sp_fseq: sequence number of the final latent hand-off of the
    // sampling run being issued (tok_finish waits for it)
    bool sp = false;
    bool sp_llm = false;                  // "tp.llm_seq": the Qwen3 decode step on the same hand-off (rms_sp_kernel; round 6)
    int sp_fseq = 0;
    int ada_slots = 1;                    // column-split adaLN: the gathered modulation tensor is double-buffered by group parity

    const GemmCfg& cfg(const char* name) const {
        auto it = g.find(name);
        if (it == g.end()) throw std::runtime_error(std::string("no launch config for GEMM '") + name + "'");
        return it->second;
    }

    long long geti(const std::string& k) const {
        auto it = I.find(k);
        if (it == I.end()) throw std::runtime_error("missing int '" + k + "'");
        return it->second;
    }
    long long geti(const std::string& k, long long d) const { auto it = I.find(k); return it == I.end() ? d : it->second; }
    double getf(const std::string& k, double d) const { auto it = F.find(k); return it == F.end() ? d : it->second; }
    const void* ptr(const std::string& k) const {
        auto it = P.find(k);
        if (it == P.end() || it->second == nullptr) throw std::runtime_error("missing pointer '" + k + "'");
        return it->second;
    }
    void* wptr(const std::string& k) const { return const_cast<void*>(ptr(k)); }
    const void* optr(const std::string& k) const { auto it = P.find(k); return it == P.end() ? nullptr : it->second; }
};

static int pad_rows(int m) { return m <= 32 ? 32 : (m <= 64 ? 64 : ((m + 127) / 128) * 128); }

// Launch heuristics from the (nwaves, ring, split-K) sweeps on MI355X (tools/gemm_sweep.py, profiles/r0*_gemm_sweep*):
//  * 8 waves (256 output columns per workgroup: the A tile is re-read half as often) for wide N or deep K, else 4;
//  * split-K so that ~180-240 workgroups exist (<= one wave of workgroups over 256 CUs);
//  * SwiGLU: fused epilogue (S = 1) when the grid fills the chip, otherwise split-K slabs + swiglu_rows;
//  * 128-row passes with K % 128 == 0: 2 panels x 2 K-parts per workgroup (64-column tiles at 4 waves, bd_gemm.hip): the
//    N = 15360 shapes become 240 single-slice tiles, the N = 5120 shapes 80 tiles x 3 slices reduced in the launch.
// `reduce3`: the caller needs ONE finished tensor (a tensor-parallel rank's fp32 partial): at most 3 grid slices.
static GemmCfg choose_cfg(const bd_ctx* c, const std::string& name, int N, int K, bool swiglu, bool reduce3 = false) {
    GemmCfg g;
    const bool two_images = (c->Mpad % 256 == 0);      // 256-row passes (MB = 8) need the 8-wave variant for occupancy
    g.nw = (N % 256 == 0 && (N >= 8192 || K >= 16384 || two_images)) ? 8 : ((N % 128 == 0) ? 4 : 2);
    // one workgroup per CU and a single wave of workgroups: 10-wave tiles when that lands N/320 just under 256 tiles
    // (not with fp8 activations: the fp8 x fp8 loop needs ~190 registers, a 10-wave workgroup has 168 per wave)
    if (!two_images && c->Mpad % 128 == 0 && N % 320 == 0 && N / 320 > 200 && N / 320 <= 256 && !c->fp8a) g.nw = 10;
    // ~120 tiles of 128 columns: two splits give 240 workgroups and only TWO slabs for the consumer to re-read
    if (!two_images && N % 128 == 0 && N / 128 >= 100 && N / 128 <= 128 && K <= 8192) g.nw = 4;
    // one wave of workgroups with a RAGGED last tile: N/32 panels over ceil(panels / 9) workgroups of 9 waves when that lands just
    // under 256 (adaLN: 2240 panels -> 249 workgroups instead of 224 x 10 waves: 131.7 vs 137.5 us isolated, profiles/
    // r02_gemm_sweep3.log).  The 5-wave form (gate/up: 1088 panels -> 218 workgroups instead of 136 x 8) measured SLOWER (88.1 vs
    // 82.8 us: fewer bytes in flight per CU), so it stays a tune option.  bf16 weights, 128-row passes (the instantiated shapes).
    bool ragged = false;
    if (!two_images && c->Mpad % 128 == 0 && !c->wfp8 && N % 32 == 0 && c->geti("tune.ragged", 1) != 0) {
        const int npmin = (N / 32 + 255) / 256;
        if (npmin == 9) { g.nw = npmin; ragged = true; }
        // 5 panels x 2 K-parts = 10 waves per workgroup over ceil(panels / 5) workgroups (gate/up: 218 CUs instead of 136) also
        // measured slower (95.8 vs 82.2 us isolated, 100 vs 93.5 us in situ): tune.ragged52 = 1 selects it
        if (npmin == 5 && K % 128 == 0 && c->geti("tune.ragged52", 0) != 0) { g.nw = 10; g.kw = 2; ragged = true; }
    }
    int ntiles = (N / 32 + g.nw / g.kw - 1) / (g.nw / g.kw);
    // row tiles of the grid (256-row passes): a large batch (ImageNet: 12288 rows = 48 row tiles) already fills the chip
    // with N tiles x row tiles -- splitting K there only multiplies fp32 slab traffic
    const int row_tiles = two_images ? c->Mpad / 256 : 1;
    int S = (int)std::lround((g.nw >= 8 ? 180.0 : 240.0) / ((double)ntiles * row_tiles));
    if (S < 1) S = 1;
    if (two_images && row_tiles >= 8) {
        // every K slice parks an fp32 slab of rows x N that the consumer re-reads: with enough row tiles to spread the work, do
        // not split beyond the point where the slabs outweigh the operands themselves.  The 14B shapes are unaffected (few row
        // tiles, W 157 MB vs 31 MB per slab at 512 rows).  ImageNet B-4x (3072 rows = 12 row tiles, W 1.8-3.5 MB vs 9-28 MB per
        // slab): one slice, the GEMM rounds / applies SwiGLU in its epilogue instead of handing 2-5 slabs to ln_mod / swiglu_rows:
        // 34.9 -> 38.1 img/s although every GEMM launch got slower.  B-1x (768 rows = 3 row tiles) keeps its splits: at one slice
        // only 9-27 workgroups exist and the GEMMs double in time (20.9 -> 18.3 img/s measured with the cap applied there too).
        const double operands = 2.0 * N * K + 2.0 * c->Mpad * K, slab = 4.0 * c->Mpad * N;
        const int cap = (int)std::floor(operands / slab);
        if (S > std::max(1, cap) && c->geti("tune.slab_cap", 1) != 0) S = std::max(1, cap);
    }
    // >= 1024 rows and FEW 256 x 256 tiles (the N = 5120 Linears of the 14B models: 80 tiles at 8 images, 160 at 16 -- 31 / 62 % of the CUs at
    // one K slice): pick the slice count that fills whole waves of 256 workgroups, charging 3 % per extra fp32 slab.  Measured (round 6,
    // tools/head_sweep.py, us per evaluation): 8 images wo / w2 at 3 slices 3613 vs 3819 at 2 (4: 3940, 1: 4305); 16 images 3 slices 6929 vs
    // 7185 at 1 (2: 7332).  512 rows keep their 5 slices (6 measured slower: 13-stage loops are all ramp), small weights their cap above.
    if (two_images && row_tiles >= 4 && !swiglu && g.nw >= 8 && (double)N * K * 2 > 12e6 && ntiles * row_tiles < 1024 &&
        (double)((ntiles * row_tiles + 255) / 256 * 256) / (ntiles * row_tiles) > 1.2 && c->geti("tune.fill_waves", 1) != 0) {
        const int W0 = ntiles * row_tiles;                     // (a last wave of workgroups less than ~83 % full at one slice)
        double best = 1e30;
        for (int s_ = 1; s_ <= 6; ++s_) {
            const int wgs = W0 * s_;
            const double cost = (double)((wgs + 255) / 256 * 256) / wgs * (1.0 + 0.03 * (s_ - 1));
            if (cost < best - 1e-9) { best = cost; S = s_; }
        }
    }
    if (ntiles >= 260 && !swiglu && row_tiles == 1) S = 3;   // > 1 wave of workgroups: split for tail balance
    if (swiglu && ntiles >= 130) S = 1;
    const bool kw2_shape = !two_images && K % 128 == 0 && N % 64 == 0 && g.nw != 10;
    const bool kw2_ok = kw2_shape && c->geti("tune.kw2", 0) != 0;          // measured slower at tp = 1 (profiles/r02_gemm_sweep2.log)
    if (kw2_ok) {
        const int t64 = N / 64;                        // 64-column tiles: 2 panels x 2 K-parts
        int s64 = (int)std::lround(240.0 / t64);
        if (s64 < 1) s64 = 1;
        if (t64 * s64 <= 256 && t64 * s64 >= 200 && s64 <= 3) { g.nw = 4; g.kw = 2; S = s64; ntiles = t64; }
    }
    // The N = 5120 shapes (wo, w2, o_proj, cond, fc2: 40 tiles of 128 columns would need 6 K slices = 6 fp32 slabs of 2.6 MB that the
    // consumer row kernel re-reads): 64-column tiles of 2 panels x 2 K-parts instead -- 80 tiles x 3 slices, HALF the slab traffic.
    // The GEMM itself is no faster (wo 20.9 vs 20.5 us, w2 25.0 vs 22.5) but everything around it is: ln_mod reads 3 slabs instead of
    // 6 and the next GEMM starts behind less dirty data -- in situ on one box (profiles/r03_bench_b1_s3slabs.json vs _v4.json): AR loop
    // 3695 vs 3794 ms, image 0.2638 vs 0.2569 /s.  3 slices stay slabs (tune.reduce_max_s = 2: the in-launch reduction of 3 costs 8 us).
    bool slab3 = false;
    if (!two_images && !reduce3 && g.nw == 4 && g.kw == 1 && S >= 4 && K % 128 == 0 && N % 64 == 0 && K <= 8192 && !c->wfp8 &&
        c->geti("tune.slab3", 1) != 0) {     // (fp8 weights: measured 2 % SLOWER, profiles/r03_head_sweep5_fp8a.log -- the 26-39 MB streams are all fill / drain)
        const int t64 = N / 64, s64 = std::min(3, (int)std::lround(240.0 / t64));
        // (exactly 3: the 2-slice case, llm.qkv N = 7168, would turn 4 slabs into an in-launch reduction: 27.9 vs 23.9 us measured)
        if (s64 == 3 && t64 * s64 <= 256 && t64 * s64 >= 200) { g.nw = 4; g.kw = 2; S = s64; ntiles = t64; slab3 = true; }
    }
    // otherwise, 128-column tiles whose slices stay with the consumer (S >= 4): 8 waves as 4 panels x 2 K-parts -- two waves per SIMD
    // overlap each other's LDS / MFMA latencies at the same tile, grid and slab count (profiles/r02_gemm_sweep2_pipe.log)
    if (!slab3 && !two_images && g.nw == 4 && g.kw == 1 && S >= 4 && K % 128 == 0 && K <= 8192 && c->geti("tune.kparts8", 1) != 0) {
        g.nw = 8; g.kw = 2;
    }
    if (reduce3 && S > 3) {
        // narrower tiles instead of more slices: 64 columns (2 waves, or 2 x 2 when K allows), then 32
        if (N % 64 == 0) { g.nw = (kw2_shape ? 4 : 2); g.kw = kw2_shape ? 2 : 1; ntiles = N / 64; }
        S = (int)std::lround(240.0 / ntiles);
        if (S > 3) S = 3;
        if (S < 1) S = 1;
    }
    // Tensor-parallel shards of the 128-row passes, measured on ONE rank in loop-back (tools/head_sweep.py --tp-shard,
    // profiles/r05_tp_shard_sweep.log): what bounds a small GEMM is a CU's vector-memory path (~47 GB/s) over the weights PLUS
    // the 128 activation rows every workgroup re-reads -- so the work has to sit on ~240 CUs in SHORT K slices, and what a slice costs in
    // fp32 slabs is paid to a row-parallel consumer (finalize_rows in front of the attention, swiglu_rows behind w1), not to an in-launch
    // reduction by the last arriver.  At the tp = 8 shard (per evaluation): the tp = 1 rules 1221 us -> these 705 us.
    if (c->tp > 1 && !two_images && c->Mpad == 128 && c->geti("tune.tp_shapes", 1) != 0) {
        if (reduce3 && K <= 1024 && !c->wfp8) {
            // row-split with a short local K (wo / w2 at tp = 8: 640 / 960): ONE slice -- no ticket, no slab round trip before the push
            // epilogue -- on 32-column tiles x 2 K parts (160 workgroups), or 64-column tiles when K is not a multiple of 128
            g.nw = 2; g.kw = (K % 128 == 0) ? 2 : 1;
            S = 1;
        } else if (!reduce3 && N < 7680 && K % 128 == 0 && N % 64 == 0) {
            // column-split with few columns per rank (qkv / w1 from tp = 4 up): 64-column tiles x up to 8 K slices
            const int t64 = N / 64;
            g.nw = 4; g.kw = 2;
            S = std::max(1, std::min(8, (int)std::lround(240.0 / t64)));
        }
    }
    // Small weights under a few thousand rows (the ImageNet 1x / 4x batches: 768 / 3072 rows, 1.2 - 3.5 MB per Linear): 256 x 256
    // tiles leave most CUs idle at one K slice (B-4x qkv: 108 workgroups; B-1x w1: 27) and splitting K to fill the chip parks
    // 12 - 18 fp32 slabs (B-1x w1: 85 MB of slabs for 3.5 MB of weights, then a separate swiglu_rows pass).  Smaller tiles
    // instead, re-reading the tiny weight matrix from L2 per row tile: 256 x 128 (4 waves) when that fills the chip, else
    // 128 x 64 (2 waves), one slice with the fused epilogue wherever that gives ~200 workgroups.  Measured (same box, no decode):
    // B-1x 36.3 -> 49.9 images/s (w1 13.2 + 12.5 (swiglu_rows) -> 17.3 us, w2 15.5 (18 slabs) -> 11.8 (3)), B-4x 50.6 -> 58.8.
    if (two_images && c->Mpad <= (int)c->geti("tune.small_tiles_rows", 4096) && (double)N * K * 2 <= 12e6 && !c->wfp8 && c->tp <= 1 &&
        N % 64 == 0 && !reduce3) {
        int nw = 4, wgs = (N % 128 == 0) ? (N / 128) * (c->Mpad / 256) : 0;
        if (wgs < 200) { nw = 2; wgs = (N / 64) * (c->Mpad / 128); }
        g.nw = nw; g.kw = 1;
        S = swiglu ? 1 : std::max(1, std::min(4, (int)std::lround(240.0 / wgs)));
    }
    // 512 rows on the big weights of the 14B models (bd_gemm_half.hip: 256 x 128 tiles, split-K inside the workgroup; "tune.half" bit mask):
    //   1: N = 15360 (qkv, gate / up of the head): 240 tiles at ONE slice -- rounded / SwiGLU output, no slabs (was 120 tiles x 2 slices)
    //   2: N = 5120 .. 7168 (wo, w2, o, down, llm.qkv): 80 .. 112 tiles x 3 / 2 slices = 3 / 2 slabs (was 40 x 5 / 56 x 3 on 256 x 256 tiles)
    //   4: N > 16384 (gate / up of the LLM: 544 tiles, three waves of workgroups instead of two at twice the length)
    if (two_images && row_tiles == 2 && g.nw >= 8 && g.kw == 1 && N % 128 == 0 && K % 64 == 0 && (double)N * K * 2 > 12e6 && !c->wfp8 && c->tp <= 1 && !reduce3) {
        const int hm = (int)c->geti("tune.half", 7), t128 = (N / 128) * row_tiles;
        if (N >= 8192 && N <= 16384 && (hm & 1)) { S = 1; g.half = 1; }
        else if (N > 16384 && (hm & 4)) { S = 1; g.half = 1; }
        else if (N < 8192 && (hm & 2) && !swiglu) { S = std::max(1, std::min(4, (int)std::lround(240.0 / t128))); g.half = 1; }
    }
    g.S = S;
    // K stages in flight per wave: 2; 3 for the 4-wave tiles of the 128-row passes (qkv / w1: only 32 KiB per CU in flight at ring 2;
    // in situ on one box, profiles/r03_bench_b1_ring*.json: qkv 39.4 vs 41.6 us, w1 42.5 vs 44.8, image 0.2597 vs 0.2552 /s; ring 4 and
    // the 8-wave tiles: no gain)
    g.ring = (!two_images && c->Mpad % 128 == 0 && g.nw == 4 && g.kw == 1 && !c->wfp8) ? 3 : 2;
    g.S = (int)c->geti("tune." + name + ".S", g.S);
    g.nw = (int)c->geti("tune." + name + ".nw", g.nw);
    g.kw = (int)c->geti("tune." + name + ".kw", g.kw);
    g.ring = (int)c->geti("tune." + name + ".ring", g.ring);
    if (g.kw < 1 || g.kw > 2 || g.nw % g.kw || K % (64 * g.kw)) g.kw = 1;
    ragged = ragged || (((g.kw == 1 && (g.nw == 5 || g.nw == 9)) || (g.kw == 2 && g.nw == 10)) && !c->wfp8 && c->Mpad % 128 == 0 && !two_images);
    if (N % (32 * (g.nw / g.kw)) && !ragged) { g.kw = 1; g.nw = (N % 128 == 0) ? 4 : 2; }
    const int nst = K / (64 * g.kw);
    if (g.S > nst) g.S = nst;
    while (g.S > 1 && (g.S - 1) * ((nst + g.S - 1) / g.S) >= nst) --g.S;      // no empty split
    return g;
}

static const char* const kIntKeys[] = {
    "B", "branches", "P", "wdtype", "head.D", "head.C", "head.Dz", "head.H", "head.nblocks", "head.nada", "head.T", "head.dh", "head.sigmoid", "head.y_evals", "head.variant",
    "proj.D", "proj.C", "proj.hid", "proj.variant", "proj.rows_all", "llm.D", "llm.L", "llm.nh", "llm.nkv", "llm.F", "llm.Lmax", "llm.splits",
    "llm.head_dim", "llm.variant", "rt.dump_xhat", "rt.emit_cond", "rt.chain", "rt.llm_causal", "rt.llm_bf16", "rt.no_advance", "rt.in_first", "tune.reduce_max_s", "tune.w1_fused", "tune.kw2", "tune.kparts8", "tune.fill_waves", "tune.half", "tune.ragged", "tune.ragged52", "tune.slab_cap", "tune.slab3",
    "tune.ada_group", "tune.ada_group_nw", "tune.tp_fuse", "tune.ln_rows", "tune.small_tiles_rows", "tp.ada_split", "tp.seq", "tp.llm_seq", "tune.sp_wait", "tune.sp_inv", "tune.finalize_s", "tune.sp_gsig", "tune.tp_shapes"};
static const char* const kGemmNames[] = {"head.cond", "head.ada", "head.qkv", "head.wo", "head.w1", "head.w2", "proj.fc2",
                                         "llm.qkv", "llm.o", "llm.gu", "llm.down"};
static bool known_int_key(const std::string& k) {
    for (const char* n : kIntKeys) if (k == n) return true;
    if (k.rfind("tune.", 0) == 0) {                    // tune.<gemm>.{S,nw,kw,ring}
        for (const char* gname : kGemmNames)
            for (const char* f : {".S", ".nw", ".kw", ".ring"})
                if (k == std::string("tune.") + gname + f) return true;
    }
    return false;
}
static const char* const kPtrKeys[] = {
    "head.cond_w", "head.cond_b", "head.in_w", "head.in_b", "head.ada_w", "head.ada_b", "head.ada_w_l", "head.ada_b_l", "head.ada_loc", "head.lin_w", "head.lin_b", "head.temb",
    "head.noise", "head.tok_all", "head.y_all", "head.y_scale_all", "head.cfg_table", "proj.w1", "proj.b1", "proj.w2", "proj.b2", "llm.final_norm", "llm.emb_norm", "llm.rope2d",
    "llm.cos", "llm.sin", "pos",
    // workspaces (the caller allocates them after bd_ctx_finalize; a head-/projector-only context may borrow another's)
    "state", "gemm.cnt", "head.cond_frag", "head.cond_part", "head.xt", "head.y_frag", "head.X", "head.ada_bf", "head.cemb",
    "head.h_frag", "head.h_scale", "head.y_scale", "head.qkv_part", "head.qkv_bf", "head.br_bf", "head.attn_frag", "head.br_part", "head.act_frag", "head.w1_part",
    "head.pred", "head.tok_cur", "head.xhat", "head.tp_part", "proj.h_frag", "proj.part", "proj.out_bf", "llm.R", "llm.a_frag", "llm.a_scale", "llm.qkv_part",
    "llm.qkv_bf", "llm.br_bf", "llm.gu_part", "llm.q", "llm.k_cache", "llm.vt_cache", "llm.attn_opart", "llm.attn_ml",
    "llm.attn_frag", "llm.br_part", "llm.act_frag", "llm.hidden", "llm.tp_part"};
static bool known_ptr_key(const std::string& k) {
    for (const char* n : kPtrKeys) if (k == n) return true;
    auto indexed = [&](const char* prefix, std::initializer_list<const char*> fields) {
        const size_t L = std::strlen(prefix);
        if (k.rfind(prefix, 0) != 0) return false;
        size_t i = L;
        while (i < k.size() && k[i] >= '0' && k[i] <= '9') ++i;
        if (i == L || i >= k.size() || k[i] != '.') return false;
        for (const char* f : fields) if (k.compare(i + 1, std::string::npos, f) == 0) return true;
        return false;
    };
    for (const char* n : {"head.cond_w_s", "head.ada_w_s", "head.ada_w_l_s", "proj.w2_s"}) if (k == n) return true;      // fp8 scales
    return indexed("head.blk", {"ln1_w", "ln1_b", "ln2_w", "ln2_b", "wqkv", "bqkv", "wo", "bo", "w1", "b1", "w2", "b2",
                                "wqkv_s", "wo_s", "w1_s", "w2_s"}) ||
           indexed("llm.l", {"in_norm", "post_norm", "q_norm", "k_norm", "wqkv", "wo", "wgu", "wdown",
                             "wqkv_s", "wo_s", "wgu_s", "wdown_s"});
}

extern "C" {

int bd_version(void) { return 1; }
const char* bd_last_error(void) { return g_err.c_str(); }

int bd_pack_weight(void* dst, const void* src, int rows, int K, int dst_row0, int dst_rows_total, void* stream) {
    if (rows % 32 || dst_row0 % 32 || dst_rows_total % 32 || dst_row0 + rows > dst_rows_total)
        return fail("bd_pack_weight: rows, dst_row0, dst_rows_total must be multiples of 32 and nest");
    BD_TRY(bdk_pack_w(dst, src, nullptr, rows / 32, K, dst_row0 / 32, dst_rows_total / 32, 0, (hipStream_t)stream));
    return 0;
}
int bd_set_weight_layout(int stage_major) {
    if (stage_major != 0 && stage_major != 1) return fail("bd_set_weight_layout: 0 (panel-major) or 1 (stage-major)");
    bdk_set_w_layout(stage_major);
    return 0;
}
int bd_set_gemm_option(const char* name, int value) {
    if (!name || bdk_set_gemm_option(name, value) != 0)
        return fail(std::string("bd_set_gemm_option: unknown option or value: ") + (name ? name : "(null)"));
    return 0;
}
int bd_pack_weight_swiglu(void* dst, const void* gate, const void* up, int F, int K, void* stream) {
    if (F % 16) return fail("bd_pack_weight_swiglu: F must be a multiple of 16");
    BD_TRY(bdk_pack_w(dst, gate, up, F / 16, K, 0, F / 16, 1, (hipStream_t)stream));
    return 0;
}
int bd_rows_to_frag(void* dst, const void* src, int src_is_fp32, int M, int K, int RB, void* stream) {
    BD_TRY(bdk_rows_to_afrag(dst, src_is_fp32 ? (const float*)src : nullptr, src_is_fp32 ? nullptr : src, M, K, RB,
                             (hipStream_t)stream));
    return 0;
}
int bd_gfq_indices(const float* z, int* idx, int ntok, int ncodebooks, int bits, void* stream) {
    BD_TRY(bdk_gfq_indices(z, idx, ntok, ncodebooks, bits, (hipStream_t)stream));
    return 0;
}
int bd_gfq_codes(const int* idx, float* codes, int ntok, int ncodebooks, int bits, void* stream) {
    BD_TRY(bdk_gfq_codes(idx, codes, ntok, ncodebooks, bits, (hipStream_t)stream));
    return 0;
}
int bd_probe_read(const void* src, long long bytes, int blocks, void* sink, void* stream) {
    BD_TRY(bdk_probe_read(src, (size_t)bytes, blocks, sink, (hipStream_t)stream));
    return 0;
}
int bd_gemm_partial(const void* a, int RB, const void* w, int N, int K, int S, int nw, float* out, void* stream) {
    BD_TRY(bdk_gemm(a, RB, w, N, K, S, nw, BD_EPI_PARTIAL, out, nullptr, nullptr, nullptr, (hipStream_t)stream));
    return 0;
}
int bd_gemm_bf16(const void* a, int RB, const void* w, const void* bias, int N, int K, int S, int nw, float* scratch,
                 int* counters, void* out_bf16, void* stream) {
    BD_TRY(bdk_gemm(a, RB, w, N, K, S, nw, BD_EPI_BF16, scratch, out_bf16, bias, counters, (hipStream_t)stream));
    return 0;
}
int bd_gemm_f32(const void* a, int RB, const void* w, int N, int K, int S, int nw, float* scratch, int* counters, float* out_f32,
                void* stream) {
    BD_TRY(bdk_gemm(a, RB, w, N, K, S, nw, BD_EPI_F32, scratch, out_f32, nullptr, counters, (hipStream_t)stream));
    return 0;
}
int bd_pack_weight8(void* dst, const void* src_fp8, int rows, int K, int dst_row0, int dst_rows_total, void* stream) {
    if (rows % 32 || dst_row0 % 32 || dst_rows_total % 32 || dst_row0 + rows > dst_rows_total)
        return fail("bd_pack_weight8: rows, dst_row0, dst_rows_total must be multiples of 32 and nest");
    BD_TRY(bdk_pack_w8(dst, src_fp8, nullptr, rows / 32, K, dst_row0 / 32, dst_rows_total / 32, 0, (hipStream_t)stream));
    return 0;
}
int bd_pack_weight8_swiglu(void* dst, const void* gate_fp8, const void* up_fp8, int F, int K, void* stream) {
    if (F % 16) return fail("bd_pack_weight8_swiglu: F must be a multiple of 16");
    BD_TRY(bdk_pack_w8(dst, gate_fp8, up_fp8, F / 16, K, 0, F / 16, 1, (hipStream_t)stream));
    return 0;
}
/* fp8 weights for the fp8 x fp8 GEMMs (wdtype 2: head.ada / qkv / w1, llm.qkv / gu): e4m3 bytes in the K = 64 operand order */
int bd_pack_weight8k(void* dst, const void* src_fp8, int rows, int K, int dst_row0, int dst_rows_total, void* stream) {
    if (rows % 32 || dst_row0 % 32 || dst_rows_total % 32 || dst_row0 + rows > dst_rows_total)
        return fail("bd_pack_weight8k: rows, dst_row0, dst_rows_total must be multiples of 32 and nest");
    BD_TRY(bdk_pack_w8k(dst, src_fp8, nullptr, rows / 32, K, dst_row0 / 32, dst_rows_total / 32, 0, (hipStream_t)stream));
    return 0;
}
int bd_pack_weight8k_swiglu(void* dst, const void* gate_fp8, const void* up_fp8, int F, int K, void* stream) {
    if (F % 16) return fail("bd_pack_weight8k_swiglu: F must be a multiple of 16");
    BD_TRY(bdk_pack_w8k(dst, gate_fp8, up_fp8, F / 16, K, 0, F / 16, 1, (hipStream_t)stream));
    return 0;
}
/* standalone fp8 x fp8 GEMM (tests): a8 / ascale = fp8 activations in the A8 layout + per-row scales (bd_quant_rows8) */
int bd_gemm_w8a8(const void* a8, const float* ascale, int RB, const void* w8k, const float* wscale, const void* bias, int N, int K, int S, int nw,
                 int epi, float* scratch, int* counters, void* out, void* stream) {
    if (!wscale || !ascale) return fail("bd_gemm_w8a8: scales required");
    if (epi < 0 || epi > 3) return fail("bd_gemm_w8a8: epi 0 = fp32 slabs, 1 = SwiGLU, 2 = bf16(+bias), 3 = fp32 sum");
    BD_TRY(bdk_gemm8a(a8, ascale, RB, w8k, wscale, N, K, S, nw, epi, epi == BD_EPI_PARTIAL ? (float*)out : scratch,
                      epi == BD_EPI_PARTIAL ? nullptr : out, bias, counters, (hipStream_t)stream));
    return 0;
}
/* rows [M][K] fp32 -> fp8-e4m3 activations in the A8 layout + per-row scales (what the row kernels emit in wdtype 2) */
int bd_quant_rows8(void* a8, float* ascale, const float* src, int M, int K, int RB, void* stream) {
    BD_TRY(bdk_quant_rows8(a8, ascale, src, M, K, RB, (hipStream_t)stream));
    return 0;
}
int bd_gemm_w8(const void* a, int RB, const void* w8, const float* wscale, const void* bias, int N, int K, int S, int nw, int epi,
               float* scratch, int* counters, void* out, void* stream) {
    if (!wscale) return fail("bd_gemm_w8: scales required");
    if (epi < 0 || epi > 3) return fail("bd_gemm_w8: epi 0 = fp32 slabs, 1 = SwiGLU, 2 = bf16(+bias), 3 = fp32 sum");
    BD_TRY(bdk_gemm(a, RB, w8, N, K, S, nw, epi, epi == BD_EPI_PARTIAL ? (float*)out : scratch, epi == BD_EPI_PARTIAL ? nullptr : out,
                    bias, counters, (hipStream_t)stream, wscale));
    return 0;
}
int bd_gemm_swiglu(const void* a, int RB, const void* w, const void* bias, int N2, int K, int nw, void* act, void* stream) {
    BD_TRY(bdk_gemm(a, RB, w, N2, K, 1, nw, BD_EPI_SWIGLU, nullptr, act, bias, nullptr, (hipStream_t)stream));
    return 0;
}

int bd_gemm_swiglu_splitk(const void* a, int RB, const void* w, const void* bias, int N2, int K, int S, int nw, float* scratch, int* counters,
                          void* act, void* stream) {
    BD_TRY(bdk_gemm(a, RB, w, N2, K, S, nw, BD_EPI_SWIGLU, scratch, act, bias, counters, (hipStream_t)stream));
    return 0;
}

bd_ctx* bd_ctx_create(void) { return new bd_ctx(); }
void bd_ctx_destroy(bd_ctx* c) {
    if (!c) return;
    for (int i = 0; i < 2; ++i) {
        if (c->gexec[i]) hipGraphExecDestroy(c->gexec[i]);
        if (c->graph[i]) hipGraphDestroy(c->graph[i]);
    }
    delete c;
}
// unknown keys are rejected: a typo must not silently fall back to a default (keys: DESIGN.md / the tables above)
int bd_ctx_set_int(bd_ctx* c, const char* k, long long v) {
    if (!c || !k || !known_int_key(k)) return fail(std::string("bd_ctx_set_int: unknown key '") + (k ? k : "(null)") + "'");
    c->I[k] = v;
    return 0;
}
int bd_ctx_set_float(bd_ctx* c, const char* k, double v) {
    if (!c || !k || std::string(k) != "llm.eps") return fail(std::string("bd_ctx_set_float: unknown key '") + (k ? k : "(null)") + "'");
    c->F[k] = v;
    return 0;
}
int bd_ctx_set_ptr(bd_ctx* c, const char* k, const void* p) {
    if (!c || !k || !known_ptr_key(k)) return fail(std::string("bd_ctx_set_ptr: unknown key '") + (k ? k : "(null)") + "'");
    c->P[k] = p;
    return 0;
}
/* tensor parallelism: this context is rank bd_comm.rank of bd_comm.size; call before bd_ctx_finalize, weights pre-sliced */
/* plan a rank's context without a communicator (inspection / host-side tests); a step then fails loudly */
int bd_ctx_set_tp(bd_ctx* c, int rank, int size) {
    if (!c || c->finalized || size < 1 || size > 8 || rank < 0 || rank >= size) return fail("bd_ctx_set_tp: before finalize, 0 <= rank < size <= 8");
    c->comm = nullptr; c->tp = size; c->tpr = rank;
    return 0;
}
int bd_ctx_set_comm(bd_ctx* c, bd_comm* comm) {
    if (!c || c->finalized) return fail("bd_ctx_set_comm: call before bd_ctx_finalize");
    c->comm = comm;
    c->tp = bdk_comm_size(comm);
    c->tpr = bdk_comm_rank(comm);
    return 0;
}

int bd_ctx_finalize(bd_ctx* c) {
    try {
        c->B = (int)c->geti("B");
        c->branches = (int)c->geti("branches");
        c->Pn = (int)c->geti("P");
        // tokens per AR step: 64 / 16 (T2I 64x / 16x, ImageNet 16x), 4 (ImageNet 4x), 1 (ImageNet 1x: MLP head, causal transformer)
        if (c->Pn != 64 && c->Pn != 16 && c->Pn != 4 && c->Pn != 1) return fail("parallel_num must be 64, 16, 4 or 1");
        c->BP = c->B * c->Pn;
        c->M = c->branches * c->BP;
        c->Mpad = pad_rows(c->M);
        c->BPpad = pad_rows(c->BP);
        c->RB = c->Mpad / 32;
        c->RBp = c->BPpad / 32;
        c->prows = c->BP;
        if (c->geti("proj.rows_all", 0)) {                 // the projector feeds all branches * B sequences from their own token rows
            c->prows = c->M; c->BPpad = c->Mpad; c->RBp = c->RB;
        }
        c->has_head = c->I.count("head.D") > 0;
        c->has_llm = c->I.count("llm.D") > 0;
        c->has_proj = c->I.count("proj.D") > 0;
        c->wfp8 = c->geti("wdtype", 0) >= 1;
        c->fp8a = c->geti("wdtype", 0) == 2;
        if (c->fp8a && c->has_llm && c->geti("llm.variant", 0) != 0) return fail("wdtype 2 (fp8 activations): the T2I paths only");
        // the step state has BD_MAX_SEQ (64) per-sequence KV-length slots: a limit of the Qwen3 decode path only (prompts differ in
        // length); head-only contexts read just the step counter, imagenet sequences all share slot 0
        if (c->branches * c->B > BD_MAX_SEQ && c->has_llm && c->geti("llm.variant", 0) == 0)
            return fail("too many sequences for the Qwen3 decode path (max 64: num_images <= 32 with CFG)");
        c->hMlp = c->has_head ? (int)c->geti("head.variant", 0) : 0;
        if (c->hMlp != 0 && c->hMlp != 1) return fail("head.variant: 0 (transformer blocks) or 1 (MLP blocks)");
        c->ws.clear();
        auto add = [&](const std::string& n, long long bytes) { c->ws.push_back({n, bytes}); };
        add("state", sizeof(BdStepState));
        add("gemm.cnt", 16384 * sizeof(int));          // split-K arrival counters (one per output tile), zero between launches
        const long long Mp = c->Mpad;
        if (c->has_head) {
            c->hD = (int)c->geti("head.D"); c->hC = (int)c->geti("head.C"); c->hDz = (int)c->geti("head.Dz");
            c->hH = (int)c->geti("head.H"); c->hNB = (int)c->geti("head.nblocks"); c->hNA = (int)c->geti("head.nada");
            c->hT = (int)c->geti("head.T", c->Pn);
            if (c->hD % 128 || c->hH % 64 || c->hDz % 64) return fail("head dims must be multiples of 128/64");
            if (c->hNB % c->hNA) return fail("head.nblocks must be divisible by head.nada");
            // stacked adaLN projections: per adaLN block 6 chunks (scale1, shift1, gate1, scale2, shift2, gate2; flow_head:331)
            // or 3 for the MLP head (scale, shift, gate; imagenet diff_head.py:237), then the final layer's (scale, shift)
            c->hNada = c->hNA * (c->hMlp ? 3 : 6) * c->hD + 2 * c->hD;
            if (c->hMlp && c->tp > 1) return fail("head.variant = 1 (MLP head) has no tensor-parallel form");
            const int dh = (int)c->geti("head.dh", 128), tp = c->tp;
            if ((c->hD / dh) % tp || (c->hH / tp) % 64 || (c->hD / tp) % 64)
                return fail("head: attention heads and the SwiGLU width must divide by the tensor-parallel size (64-column units)");
            c->hDl = c->hD / tp; c->hHl = c->hH / tp;      // this rank's attention columns / SwiGLU features
            c->g["head.cond"] = choose_cfg(c, "head.cond", c->hD, c->hDz, false);
            c->g["head.ada"] = choose_cfg(c, "head.ada", c->hNada, c->hD, false);
            {
                GemmCfg& ga = c->g["head.ada"];
                ga.S = 1;                                      // bf16(+bias) epilogue: 13 consumers read 2 B, not S x 4 B
                // one wave per panel walks K front to back: the SAME summation order as the 256-row kernel that computes the
                // projections of a whole group of evaluations (head_ada_group), so the grouped and the per-evaluation forms
                // are bit-identical at every size (K-parts inside a workgroup would add two half sums instead)
                if (ga.kw > 1) { ga.nw /= ga.kw; ga.kw = 1; }
            }
            c->g["head.qkv"] = choose_cfg(c, "head.qkv", 3 * c->hDl, c->hD, false);
            c->g["head.wo"] = choose_cfg(c, "head.wo", c->hD, c->hDl, false, tp > 1);
            c->g["head.w1"] = choose_cfg(c, "head.w1", 2 * c->hHl, c->hD, true);
            c->g["head.w2"] = choose_cfg(c, "head.w2", c->hD, c->hHl, false, tp > 1);
            const int sbr = std::max(c->cfg("head.wo").S, c->cfg("head.w2").S);
            // The adaLN projection of evaluation i depends on (t_i, cond) only, not on the latent: the projections of G
            // consecutive evaluations are ONE GEMM over G * Mpad rows (the 256-row kernel: weights streamed once per G
            // evaluations instead of once per evaluation -- 21 % of the head's weight bytes; every row's K sum runs in the same
            // order through the same MFMA, so the result is bit-identical to G separate launches).  Default: 512 rows per GEMM
            // (4 evaluations at 128 rows; round 3, profiles/r03_head_sweep2.log: 974 us per evaluation at G = 4, 8 and 16 against
            // 1006 at G = 1), 2048 rows for the single-GPU 14B head since round 4 (below).  The last group of a schedule is short.
            {
                long long g = c->geti("tune.ada_group", -1);
                // (fp8 weights with bf16 activations have no 256-row form; fp8 weights + activations do: bd_gemm8.hip)
                const bool can = (!c->wfp8 || c->fp8a) && c->hNada % 256 == 0;
                // 128 rows and fewer: 512 rows per GEMM; 256 / 512 rows (num_images 2 / 4): 1024 rows per GEMM, where the LDS-tiled
                // MFMA-bound kernel takes over (bd_gemm_tile.hip: adaLN at 1024 rows 694 vs 786 us on the 256-row kernel)
                if (g < 0) g = !can ? 1 : (Mp <= 128 ? 512 / Mp : (Mp <= 512 && 1024 % Mp == 0 ? 1024 / Mp : 1));
                // round 4: one image on one GPU, bf16, a wide projection (the 14B head): 2048 rows per GEMM -- the tiled kernel with
                // register-staged operand fetch (bd_gemm_tile.hip) runs 16 evaluations' projections in 1204 us (75 us each) against 92-96 us
                // each for 4 on the 256-row kernel; in situ on one box (profiles/r04_head_sweep_ada_group.log) 987.9 us per evaluation at
                // G = 16 against 1003.7 / 1007.0 at G = 4, 998.3 at 8, 990.8 at 26.  293 MB of modulation tensor instead of 73.
                // Tensor-parallel contexts keep 512 rows (the gather region and the all-gather payload scale with G).
                // (round 5: the same 2048 rows per GEMM at 256 / 512 rows per evaluation -- num_images = 4: 2318.6 vs 2336-2339 us per
                // evaluation at 4 evaluations per GEMM against 2, profiles/r05_head_sweep_b4.log)
                if (c->geti("tune.ada_group", -1) < 0 && can && (Mp == 128 || Mp == 256 || Mp == 512) && tp <= 1 && !c->wfp8 && c->hNada >= 4096 && c->hD >= 2048)
                    g = 2048 / Mp;
                if (g < 1 || g > 64 || (g > 1 && ((c->RB * g) % 8 != 0 || !can)))
                    return fail("tune.ada_group: 1..64 evaluations, rows a multiple of 256, adaLN width a multiple of 256, bf16 weights or fp8 weights + activations");
                c->adaG = (int)g;
            }
            add("head.cond_frag", Mp * c->hDz * 2);
            add("head.cond_part", (long long)c->cfg("head.cond").S * Mp * c->hD * 4);
            add("head.xt", (long long)c->BP * c->hC * 4);
            add("head.y_frag", Mp * c->hD * 2);
            add("head.X", Mp * c->hD * 2);
            // column-split projection: the peers push group g + 1 while a slower rank may still read group g's gate / scale / shift in its
            // last head_final -> two slots, selected by the group's parity (a rank can only be TWO groups ahead of another after a whole
            // group of exchanges with it, i.e. after that rank has left the slot)
            c->ada_slots = c->geti("tp.ada_split", 0) ? 2 : 1;
            add("head.ada_bf", Mp * c->hNada * 2 * c->adaG * c->ada_slots);
            if (c->geti("tp.ada_split", 0)) {
                // COLUMN-split adaLN projection (reference layout flow_head_parallel_x.py:331): this rank computes hNada / tp
                // of the output columns of a whole group of evaluations into head.ada_loc and pushes them into every rank's head.ada_bf
                // (which the caller places in the communicator's gather region) -- instead of every rank streaming all 0.73 GB
                if (tp < 2 || !c->comm || bdk_comm_mode(c->comm) != 0) return fail("tp.ada_split: a tensor-parallel context on the hand-written exchange only");
                if (c->adaG < 2) return fail("tp.ada_split: needs the grouped adaLN projection (tune.ada_group > 1)");
                if (c->hNada % tp || (c->hNada / tp) % 256) return fail("tp.ada_split: the adaLN width must split into multiples of 256 columns");
                add("head.ada_loc", Mp * (c->hNada / tp) * 2 * c->adaG);
            }
            add("head.cemb", Mp * c->hD * 2);
            add("head.h_frag", Mp * c->hD * 2);
            add("head.h_scale", Mp * 4);                       // fp8 activations: per-row scales of h / y (wdtype 2)
            add("head.y_scale", Mp * 4);
            add("head.qkv_part", (long long)c->cfg("head.qkv").S * Mp * 3 * c->hDl * 4);
            add("head.qkv_bf", Mp * 3 * c->hDl * 2);
            add("head.br_bf", Mp * c->hD * 2);
            add("head.attn_frag", Mp * c->hDl * 2);
            add("head.br_part", (long long)sbr * Mp * c->hD * 4);
            add("head.act_frag", Mp * c->hHl * 2);
            add("head.w1_part", (long long)c->cfg("head.w1").S * Mp * 2 * c->hHl * 4);
            add("head.pred", (long long)c->BP * c->hC * 4);
            add("head.tok_cur", (long long)(c->geti("proj.rows_all", 0) ? c->M : c->BP) * c->hC * 4);
            add("head.xhat", Mp * c->hC * 4);
            if (tp > 1) add("head.tp_part", Mp * c->hD * 4);   // this rank's fp32 partial of a row-split Linear (wo / w2)
            c->sp = false;
            if (c->geti("tp.seq", 0)) {
                // sequence-parallel row kernels: 128 real rows (one image, 64-token patches, the headline shape: the fused push exists
                // for the 128-row kernel), whole 8-row groups per rank with the cond / uncond rows of a patch position on one rank,
                // bf16 operand rows (bf16 or fp8 weights, not the fp8-activation mode), the hand-written exchange with a landing buffer
                BdSpLink L;
                if (tp < 2 || !c->comm || !bdk_sp_link(c->comm, &L)) return fail("tp.seq: needs a tensor-parallel communicator created with an operand landing buffer (bd_comm_create3)");
                if (c->M != 128 || Mp != 128 || c->hMlp || c->fp8a || (c->BP / 8) % tp || c->BP % 8 || c->branches > 2)
                    return fail("tp.seq: 128 rows (one image, parallel_num 64), patch positions in whole 8-row groups per rank, bf16 activations");
                if (bdk_sp_hbuf_bytes(c->comm) < Mp * c->hD * 2) return fail("tp.seq: the communicator's operand landing buffer is smaller than rows x head.D x 2");
                c->sp = true;
            }
        }
        if (c->has_proj) {
            const int D = (int)c->geti("proj.D");
            const int hk = (int)c->geti("proj.variant", 0) ? (int)c->geti("proj.hid") : D;     // K of the second Linear
            if (hk % 64) return fail("proj hidden width must be a multiple of 64");
            c->g["proj.fc2"] = choose_cfg(c, "proj.fc2", D, hk, false);
            add("proj.h_frag", (long long)c->BPpad * hk * 2);
            add("proj.part", (long long)c->cfg("proj.fc2").S * c->BPpad * D * 4);
            add("proj.out_bf", (long long)c->BPpad * D * 2);
        }
        if (c->has_llm) {
            c->lD = (int)c->geti("llm.D"); c->lL = (int)c->geti("llm.L"); c->lnh = (int)c->geti("llm.nh");
            c->lnkv = (int)c->geti("llm.nkv"); c->lF = (int)c->geti("llm.F"); c->lLmax = (int)c->geti("llm.Lmax");
            c->lsplits = (int)c->geti("llm.splits", 8);
            c->ldh = (int)c->geti("llm.head_dim", 128);
            c->lvariant = (int)c->geti("llm.variant", 0);
            if (!((c->ldh == 128 && c->lvariant == 0) || (c->ldh == 64 && c->lvariant == 1 && c->lnkv == c->lnh && c->Pn <= 16)))
                return fail("llm: head_dim 128 (Qwen3) or head_dim 64 + variant 1 (imagenet transformer, MHA, P <= 16)");
            if (c->lLmax % 64) return fail("llm.Lmax must be a multiple of 64");
            const int tp = c->tp;
            if (tp > 1 && (c->lvariant != 0 || c->lnh % tp || c->lnkv % tp || (c->lF / tp) % 64 || c->lF % tp))
                return fail("llm: q heads, kv heads and the FFN width must divide by the tensor-parallel size (Qwen3 path only)");
            c->lnhl = c->lnh / tp; c->lnkvl = c->lnkv / tp; c->lFl = c->lF / tp;
            c->lNqkv = (c->lnhl + 2 * c->lnkvl) * c->ldh;
            if (c->lD % 64 || c->lF % 64) return fail("llm dims must be multiples of 64");
            c->g["llm.qkv"] = choose_cfg(c, "llm.qkv", c->lNqkv, c->lD, false);
            c->g["llm.o"] = choose_cfg(c, "llm.o", c->lD, c->lnhl * c->ldh, false, tp > 1);
            c->g["llm.gu"] = choose_cfg(c, "llm.gu", 2 * c->lFl, c->lD, true);
            c->g["llm.down"] = choose_cfg(c, "llm.down", c->lD, c->lFl, false, tp > 1);
            const int nseq = c->branches * c->B, G = c->lnh / c->lnkv;
            const int sbr = std::max(c->cfg("llm.o").S, c->cfg("llm.down").S);
            add("llm.R", Mp * c->lD * 4);
            add("llm.a_frag", Mp * c->lD * 2);
            add("llm.a_scale", Mp * 4);
            add("llm.qkv_part", (long long)c->cfg("llm.qkv").S * Mp * c->lNqkv * 4);
            add("llm.qkv_bf", Mp * c->lNqkv * 2);
            add("llm.br_bf", Mp * c->lD * 2);
            add("llm.gu_part", (long long)c->cfg("llm.gu").S * Mp * 2 * c->lFl * 4);
            add("llm.q", Mp * c->lnhl * c->ldh * 2);
            add("llm.k_cache", (long long)c->lL * nseq * c->lnkvl * c->lLmax * c->ldh * 2);
            add("llm.vt_cache", (long long)c->lL * nseq * c->lnkvl * c->lLmax * c->ldh * 2);
            add("llm.attn_opart", (long long)nseq * c->lnkvl * c->lsplits * G * c->Pn * 128 * 4);
            add("llm.attn_ml", (long long)nseq * c->lnkvl * c->lsplits * G * c->Pn * 2 * 4);
            add("llm.attn_frag", Mp * c->lnhl * c->ldh * 2);
            add("llm.br_part", (long long)sbr * Mp * c->lD * 4);
            add("llm.act_frag", Mp * c->lFl * 2);
            add("llm.hidden", Mp * c->lD * 4);
            if (tp > 1) add("llm.tp_part", Mp * c->lD * 4);
            c->sp_llm = false;
            if (c->geti("tp.llm_seq", 0)) {
                // the decode step's row kernels sequence-parallel: 128 rows in whole 8-row groups per rank, bf16 operand rows, a landing
                // buffer that holds the bf16 operand region AND the fp32 final-rows region (tp.seq_hbuf_bytes)
                BdSpLink L;
                if (tp < 2 || !c->comm || !bdk_sp_link(c->comm, &L)) return fail("tp.llm_seq: needs a tensor-parallel communicator created with an operand landing buffer (bd_comm_create3)");
                if (c->M != 128 || Mp != 128 || c->fp8a || (c->M / 8) % tp || c->lvariant != 0)
                    return fail("tp.llm_seq: 128 rows (one image with CFG, parallel_num 64) in whole 8-row groups per rank, bf16 activations, the Qwen3 path");
                if (bdk_sp_hbuf_bytes(c->comm) < Mp * c->lD * 6) return fail("tp.llm_seq: the communicator's operand landing buffer is smaller than rows x llm.D x 6");
                c->sp_llm = true;
            }
        }
        c->finalized = true;
        return 0;
    } catch (const std::exception& e) { return fail(e.what()); }
}
int bd_ctx_ws_count(bd_ctx* c) { return (int)c->ws.size(); }
const char* bd_ctx_ws_name(bd_ctx* c, int i) {
    if (!c || i < 0 || i >= (int)c->ws.size()) { fail("bd_ctx_ws_name: index out of range"); return nullptr; }
    return c->ws[i].first.c_str();
}
long long bd_ctx_ws_bytes(bd_ctx* c, int i) {
    if (!c || i < 0 || i >= (int)c->ws.size()) { fail("bd_ctx_ws_bytes: index out of range"); return -1; }
    return c->ws[i].second;
}
int bd_ctx_bind(bd_ctx* c) {
    if (!c->finalized) return fail("bd_ctx_bind before bd_ctx_finalize");
    for (auto& w : c->ws)
        if (!c->optr(w.first)) return fail("workspace '" + w.first + "' not set");
    c->bound = true;
    return 0;
}

int bd_head_set_schedule(bd_ctx* c, int n_steps, const float* s, float cfg) {
    // sequence-parallel form: one sampling run issues 4 hand-offs per block and evaluation (+ the final latent rows); the sequence
    // numbers of a run must fit the epoch's low bits (bd_common.h) -- refuse here, before anything is launched or captured
    if (c->sp && ((long long)(n_steps + 1) * 4 * c->hNB + 1 > BD_SP_SEQ_MAX))
        return fail("bd_head_set_schedule: too many sampling steps for the sequence-parallel hand-off (tp.seq = 0 keeps the all-reduce form)");
    c->sched.clear();
    for (int i = 0; i <= n_steps; ++i) {
        SamplerScalars q;
        q.t = s[i * 6 + 0]; q.dt = s[i * 6 + 1]; q.den = s[i * 6 + 2]; q.var = s[i * 6 + 3];
        q.omt = s[i * 6 + 4]; q.noise_scale = s[i * 6 + 5]; q.cfg = cfg;
        q.is_final = (i == n_steps); q.cfg_mult = c->branches;
        c->sched.push_back(q);
    }
    return 0;
}

/* the guidance scale alone (the ImageNet sampler's linear ramp changes it every AR step, model_parallel.py:356-365): a host-side
 * scalar of the schedule -- eager launches pick it up at once; captured graphs hold the value they were captured with */
int bd_head_set_cfg(bd_ctx* c, float cfg) {
    if (!c || c->sched.empty()) return fail("bd_head_set_cfg: no schedule set");
    for (auto& q : c->sched) q.cfg = cfg;
    return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// every weight-streaming GEMM of the step goes through here; with profiling on (eager mode only) each launch is
// bracketed by HIP events on the launch stream so bench.py can report in-situ per-launch durations.
// a streamed weight: packed bf16, or packed fp8-e4m3 + per-output-channel fp32 scales ("<key>_s") when the context is fp8
struct WRef { const void* w; const float* s; const float* a = nullptr; };   // a: per-row scales of an fp8 ACTIVATION operand (wdtype 2)
static WRef wref(const bd_ctx* c, const std::string& key, const char* ascale_ws = nullptr) {
    WRef r{c->ptr(key), nullptr};
    if (c->wfp8) r.s = (const float*)c->ptr(key + "_s");
    if (c->fp8a && ascale_ws) r.a = (const float*)c->ptr(ascale_ws);
    return r;
}

static int gemm(bd_ctx* c, const char* name, const void* A, int RB, WRef W, int N, int K, int S, int nw, int epi,
                float* out, void* act, const void* bias, hipStream_t st) {
    bd_ctx::ProfRec r;
    if (c->prof_on) {
        r.name = name; r.bytes = (double)N * K * (W.s ? 1 : 2);
        hipEventCreate(&r.e0); hipEventCreate(&r.e1);
        hipEventRecord(r.e0, st);
    }
    int* cnt = (epi != BD_EPI_PARTIAL && S > 1) ? (int*)c->wptr("gemm.cnt") : nullptr;   // in-launch reduction tickets
#ifdef BD_GEMM_STAMP
    bdk_stamp_label(name);
#endif
    const int rc = W.a ? bdk_gemm8a(A, W.a, RB, W.w, W.s, N, K, S, nw, epi, out, act, bias, cnt, st)
                       : bdk_gemm(A, RB, W.w, N, K, S, nw, epi, out, act, bias, cnt, st, W.s);
    if (c->prof_on) { hipEventRecord(r.e1, st); c->prof.push_back(r); }
    return rc;
}

static Partial part(const bd_ctx* c, const std::string& ws, const void* bias, int S, int N, int Mpad) {
    return Partial{(const float*)c->ptr(ws), bias, S, N, Mpad};
}

static Partial done(const bd_ctx* c, const std::string& ws, int N, int Mpad) {   // finished bf16 Linear output
    return Partial{(const float*)c->ptr(ws), nullptr, 0, N, Mpad};
}

// Many K slices in front of a consumer with FEW workgroups (the attention of a tensor-parallel rank: 2 x heads / tp workgroups, each
// reading its q / k / v through every slab): one row-parallel pass (a workgroup per row, every slab load in flight) sums them into the
// finished bf16 tensor first.  "tune.finalize_s": slab count from which this runs (default: 3 under tensor parallelism, off on one GPU).
static int finalize_if_many(bd_ctx* c, Partial* q, const char* out_ws, int M, hipStream_t st) {
    // default: from 3 slabs under tensor parallelism; from 2 at 512 rows and more on one GPU (num_images = 4: 320 attention workgroups each
    // walking two 31 MB fp32 slabs -- 2286 vs 2318 us per evaluation with the pass in front, profiles/r05_head_sweep_b4.log); off otherwise
    const int from_s = (int)c->geti("tune.finalize_s", c->tp > 1 ? 3 : (c->Mpad >= 512 ? 2 : 0));
    if (from_s <= 0 || q->S < from_s) return 0;
    FinalizeRowsArgs fr{*q, c->wptr(out_ws), M, q->N};
    BD_TRY(bdk_finalize_rows(fr, st));
    *q = Partial{(const float*)c->ptr(out_ws), nullptr, 0, q->N, q->Mpad};
    return 0;
}

// A Linear whose output the consumer reads as bf16(sum + bias).  Few K-slices: the GEMM reduces them in the launch
// (last-arriver epilogue) and the consumer reads one bf16 tensor.  Many K-slices: the serial tail of a single reducing
// workgroup (S x 64-128 KiB through one CU) costs more than it saves, so the slabs stay and the consumer sums them.
static int linear(bd_ctx* c, const char* name, const void* A, int RB, WRef W, int N, int K, const GemmCfg& g,
                  const char* scratch_ws, const char* out_ws, const void* bias, int Mpad, Partial* res, hipStream_t st,
                  bool force_reduce = false) {
    // 256-row passes run the 4-wave x 2-panel kernel, which has no in-launch reduction: slabs for the consumer there
    // (round 5: at 512 rows and more the 256-row kernel reduces TWO slices in the launch too -- the consumers stop reading fp32 slabs)
    // -- for the wide weights of the 14B models on the 8-wave 256-row kernel, the forms measured; the ImageNet batches keep their rules)
    // (round 6, same box, 512 rows: slabs + finalize_rows 2339 us per evaluation against 2374 with the two slices reduced in the launch,
    //  even with the ticket-first form and the 16 B epilogues -- the reducing workgroup's tail (wait for the partner's drain, 256 KiB of
    //  write-through slab back through one CU, epilogue) is 17 us median against 6 for plain slabs, profiles/r06_launch_anatomy.log;
    //  "tune.reduce_max_s" = 2 selects the in-launch reduction there)
    const bool wide_red = false && c->Mpad >= 512 && (double)N * K * 2 > 12e6 && (g.nw & 15) == 8 && g.kw == 1 && !c->wfp8;
    const int max_s = (int)c->geti("tune.reduce_max_s", (c->Mpad % 256 == 0) ? (wide_red ? 2 : 0) : 2);
    if (g.S <= max_s || g.S == 1 || force_reduce) {            // a single slice needs no reduction: bias + rounding in the epilogue
        BD_TRY(gemm(c, name, A, RB, W, N, K, g.S, g.code(), BD_EPI_BF16, (float*)c->wptr(scratch_ws), c->wptr(out_ws), bias, st));
        *res = Partial{(const float*)c->ptr(out_ws), nullptr, 0, N, Mpad};
    } else {
        BD_TRY(gemm(c, name, A, RB, W, N, K, g.S, g.code(), BD_EPI_PARTIAL, (float*)c->wptr(scratch_ws), nullptr, nullptr, st));
        *res = Partial{(const float*)c->ptr(scratch_ws), bias, g.S, N, Mpad};
    }
    return 0;
}

// A ROW-split Linear under tensor parallelism (wo / w2 / o_proj / down_proj): this rank multiplies its K-slice into ONE
// finished fp32 partial (grid slices, if any, reduced inside the launch), then the exchange kernel sums the ranks' partials,
// adds the bias and rounds once (bd_comm.hip).  With one rank it is the plain Linear above.
static int linear_rowsplit(bd_ctx* c, const char* name, const void* A, int RB, WRef W, int N, int Klocal, const GemmCfg& g,
                           const char* scratch_ws, const char* out_ws, const char* tp_ws, const void* bias, int Mpad, int rows,
                           Partial* res, hipStream_t st) {
    if (c->tp <= 1) return linear(c, name, A, RB, W, N, Klocal, g, scratch_ws, out_ws, bias, Mpad, res, st);
    if (!c->comm) return fail(std::string(name) + ": tensor-parallel context without a communicator (bd_ctx_set_comm)");
    if (g.S > 3) return fail(std::string(name) + ": a tensor-parallel partial needs at most 3 grid slices");
    // phase 1 of the exchange (every peer's slice of the partial into that peer's staging row) fused into the GEMM's epilogue where
    // the shape allows it: the exchange kernel then only signals, waits, reduces and pushes the result ("tune.tp_fuse" = 0: unfused)
    BdTpPush push;
    const bool want = c->geti("tune.tp_fuse", 1) != 0 && bdk_tp_push_target(c->comm, rows, N, &push);
    bdk_gemm_set_push(want ? &push : nullptr);
    BD_TRY(gemm(c, name, A, RB, W, N, Klocal, g.S, g.code(), BD_EPI_F32, (float*)c->wptr(scratch_ws), c->wptr(tp_ws), nullptr, st));
    if (bdk_gemm_push_used()) bdk_tp_mark_prepushed(c->comm);
    BD_TRY(bdk_tp_allreduce(c->comm, (const float*)c->ptr(tp_ws), bias, rows, N, res, st));
    return 0;
}

// Sequence-parallel form of a ROW-split Linear (wo / w2): the GEMM's epilogue pushes every owner its rows of the fp32 partial (8-row
// groups round-robin, BdTpPush::il); nothing else is launched -- the owner's row kernel (ln_mod_sp / head_final_sp) reduces.
// Returns the sequence number of the hand-off in *seq.
static int linear_rowsplit_sp(bd_ctx* c, const char* name, const void* A, int RB, WRef W, int N, int Klocal, const GemmCfg& g,
                              const char* scratch_ws, const char* tp_ws, int rows, int* seq, hipStream_t st) {
    if (g.S > 3) return fail(std::string(name) + ": a tensor-parallel partial needs at most 3 grid slices");
    *seq = bdk_sp_next_seq(c->comm);
    if (*seq < 0) return fail("sequence-parallel exchange: more than 65535 hand-offs since the last bd_head_cond / bd_head_sample");
    BdTpPush push;
    // "tune.sp_gsig" = 1: the GEMM's last workgroup signals the owners; 0 (default): the owner's row kernel's first block does -- the
    // arrival counter + barrier at the end of every GEMM workgroup cost more than the flag's head start returns (loop-back, per
    // evaluation: tp 8 728.6 vs 702.1 us, tp 4 761.0 vs 743.4, tp 2 918.4 vs 909.5; profiles/r05_tp_rank_critical_path.log)
    if (!bdk_tp_push_target_sp(c->comm, rows, N, c->geti("tune.sp_gsig", 0) ? *seq : 0, &push))
        return fail(std::string(name) + ": no sequence-parallel push target for this shape");
    if (push.done_cnt) push.done_cnt = (int*)c->wptr("gemm.cnt") + 16383;   // arrival counter in ordinary (L2-served) memory: the last counter word, never a tile's
    bdk_gemm_set_push(&push);
    BD_TRY(gemm(c, name, A, RB, W, N, Klocal, g.S, g.code(), BD_EPI_F32, (float*)c->wptr(scratch_ws), c->wptr(tp_ws), nullptr, st));
    if (!bdk_gemm_push_used()) return fail(std::string(name) + ": the GEMM did not take the push target");
    bdk_comm_count_exchange(c->comm);
    return 0;
}
// the consumer GEMM of the operand rows pushed with sequence number `seq`: waits in its prologue ("tune.sp_wait" = 0: a one-workgroup
// wait kernel in front of it instead)
static int sp_arm_wait(bd_ctx* c, int seq, hipStream_t st) {
    if (c->geti("tune.sp_wait", 1) == 0) { BD_TRY(bdk_sp_wait_rows(c->comm, seq, c->M, st)); return 0; }
    BdHWait w;
    if (!bdk_sp_hwait(c->comm, seq, c->M, &w)) return fail("sequence-parallel exchange: no flag block");
    w.inv = (int)c->geti("tune.sp_inv", 0);
    bdk_gemm_set_hwait(&w);
    return 0;
}

static int head_cond(bd_ctx* c, hipStream_t st) {   // cond_embed(c) is constant over the N+1 evals of this AR step
    if (c->sp) BD_TRY(bdk_sp_begin(c->comm, st));   // sequence numbers of the hand-offs restart with every sampling run
    const GemmCfg& g = c->cfg("head.cond");
    Partial unused;
    BD_TRY(linear(c, "head.cond", c->ptr("head.cond_frag"), c->RB, wref(c, "head.cond_w"), c->hD, c->hDz, g, "head.cond_part",
                  "head.cemb", c->ptr("head.cond_b"), c->Mpad, &unused, st, /*force_reduce=*/true));
    // y_i = silu(time_embed(t_i) + cond_embed(c)) for the whole schedule while cond_embed is hot (the caller provides
    // head.y_all [head.y_evals][Mpad][D] when it fits; otherwise every evaluation computes its own y)
    c->y_ready = false;
    const int n_evals = (int)c->sched.size();
    const int G = c->adaG;
    if (n_evals > 0 && c->optr("head.y_all") && c->geti("head.y_evals", 0) >= (n_evals + G - 1) / G * G) {
        HeadYAllArgs ya{c->ptr("head.cemb"), c->ptr("head.temb"), c->wptr("head.y_all"), c->M, c->hD, c->RB, c->Mpad, n_evals, G};
        if (c->fp8a) ya.a8_scale = (float*)c->wptr("head.y_scale_all");
        BD_TRY(bdk_head_y_all(ya, st));
        c->y_ready = true;
    }
    return 0;
}

// y = silu(time_embed(t_i) + cond_embed(c)) and the stacked adaLN projection of evaluation i into half `buf` of head.ada_bf
static int head_ada(bd_ctx* c, int i, int buf, hipStream_t st) {
    const int D = c->hD, RB = c->RB;
    const void* y = c->ptr("head.y_frag");
    const float* ysc = c->fp8a ? (const float*)c->ptr("head.y_scale") : nullptr;
    if (c->y_ready && c->adaG == 1) {                          // every y_i of this AR step was produced with cond_embed (head_cond)
        y = (const bf16_t*)c->ptr("head.y_all") + (size_t)i * c->Mpad * D;
        if (c->fp8a) ysc = (const float*)c->ptr("head.y_scale_all") + (size_t)i * c->Mpad;
    } else {
        HeadPrologueArgs pa;
        pa.cemb = c->ptr("head.cemb");
        pa.temb = (const bf16_t*)c->ptr("head.temb") + (size_t)i * D;
        pa.xt = nullptr; pa.in_w = nullptr; pa.in_b = nullptr; pa.X = nullptr;
        pa.y_frag = c->wptr("head.y_frag");
        pa.M = c->M; pa.BP = c->BP; pa.D = D; pa.C = c->hC; pa.RB = RB;
        if (c->fp8a) pa.a8_scale = (float*)c->wptr("head.y_scale");
        BD_TRY(bdk_head_prologue(pa, st));
    }
    GemmCfg ga = c->cfg("head.ada");
    bf16_t* out = (bf16_t*)c->wptr("head.ada_bf") + (size_t)buf * c->Mpad * c->hNada;
    WRef wa = wref(c, "head.ada_w");
    wa.a = ysc;
    BD_TRY(gemm(c, "head.ada", y, RB, wa, c->hNada, D, 1, ga.code(), BD_EPI_BF16,
                nullptr, out, c->ptr("head.ada_b"), st));
    return 0;
}

// The adaLN projections of evaluations g * G .. g * G + G - 1 as one GEMM over G * Mpad rows (head.y_all holds the operand in
// that layout, head_cond); evaluation i then reads rows (i % G) * Mpad .. of head.ada_bf.
static int head_ada_group(bd_ctx* c, int g, hipStream_t st) {
    const int G = c->adaG, D = c->hD;
    const int left = (int)c->sched.size() - g * G, Gg = left < G ? left : G;      // the last group may be short
    const int rbg = Gg == G ? c->RB * G : ((c->RB * Gg + 7) & ~7);                // (head_y_all_kernel lays it out with this row-block count)
    const bf16_t* y = (const bf16_t*)c->ptr("head.y_all") + (size_t)g * G * c->Mpad * D;
    char name[32];
    std::snprintf(name, sizeof(name), "head.ada[x%d]", Gg);     // profiling: Gg evaluations' worth of rows per weight pass
    const bool split = c->tp > 1 && c->geti("tp.ada_split", 0) != 0;
    WRef wa = wref(c, split ? "head.ada_w_l" : "head.ada_w");
    if (c->fp8a) wa.a = (const float*)c->ptr("head.y_scale_all") + (size_t)g * G * c->Mpad;
    const int nw = (int)c->geti("tune.ada_group_nw", c->fp8a ? 4 : 8) + 16 * 2;   // 8 waves, ring 2: the 256-row / tiled kernels
    if (split) {
        // this rank's columns of the group's modulation tensor, then one push all-gather into every rank's head.ada_bf
        const int Nl = c->hNada / c->tp;
        BD_TRY(gemm(c, name, y, rbg, wa, Nl, D, 1, nw, BD_EPI_BF16, nullptr, c->wptr("head.ada_loc"), c->ptr("head.ada_b_l"), st));
        bf16_t* dst = (bf16_t*)c->wptr("head.ada_bf") + (size_t)(g & 1) * G * c->Mpad * c->hNada;      // the group's slot (double-buffered by parity)
        BD_TRY(bdk_tp_allgather(c->comm, c->ptr("head.ada_loc"), dst, rbg * 32, Nl, c->hNada, st));
        return 0;
    }
    BD_TRY(gemm(c, name, y, rbg, wa, c->hNada, D, 1, nw, BD_EPI_BF16, nullptr, c->wptr("head.ada_bf"), c->ptr("head.ada_b"), st));
    return 0;
}

// `ada_buf` < 0: compute y and the adaLN projection here, in line (the plain path); >= 0: this evaluation's rows of a grouped
// projection (head_ada_group) are slot `ada_buf` of head.ada_bf
static int head_eval(bd_ctx* c, int i, hipStream_t st, int ada_buf = -1, bool x0_ready = false, bool chain_next = false) {
    if (i < 0 || i >= (int)c->sched.size()) return fail("bd_head_eval: eval index outside the schedule");
    const int D = c->hD, Mp = c->Mpad, RB = c->RB, M = c->M;
    const int Dl = c->hDl, Hl = c->hHl;                       // this rank's attention columns / SwiGLU features (== D, H at tp = 1)
    const int n_steps = (int)c->sched.size() - 1;
    const BdStepState* state = (const BdStepState*)c->ptr("state");
    if (ada_buf < 0) {
        BD_TRY(head_ada(c, i, 0, st));
        ada_buf = 0;
    }
    if (!x0_ready) {                                           // x0 = input_proj(x_t): the previous evaluation's final kernel wrote it
        HeadPrologueArgs pa;                                   // when the evaluations run as a chain (head_sample)
        pa.cemb = nullptr; pa.temb = nullptr; pa.y_frag = nullptr;
        pa.xt = (const float*)c->ptr("head.xt");
        pa.in_w = c->ptr("head.in_w"); pa.in_b = c->ptr("head.in_b");
        pa.X = c->wptr("head.X");
        pa.M = M; pa.BP = c->BP; pa.D = D; pa.C = c->hC; pa.RB = RB;
        BD_TRY(bdk_head_prologue(pa, st));
    }

    const void* ada = (const bf16_t*)c->ptr("head.ada_bf") + (size_t)ada_buf * Mp * c->hNada;
    const int sw = c->hNB / c->hNA;
    const bool mlp = c->hMlp != 0;
    const int nc = mlp ? 3 : 6;                               // adaLN chunks per adaLN block; the block's last gate is chunk nc - 1
    const GemmCfg &gq = c->cfg("head.qkv"), &go = c->cfg("head.wo"), &g1 = c->cfg("head.w1"), &g2 = c->cfg("head.w2");
    if (c->sp) {
        // ---- sequence-parallel evaluation (bd_sp.hip): per block  ln_mod_sp -> qkv (waits for the rows) -> attention -> wo (pushes
        // partial rows) -> ln_mod_sp -> w1 (waits) -> w2 (pushes) ; head_final_sp.  12 GEMMs + 6 attention + 13 row kernels, no exchange kernel.
        BdSpLink L;
        if (!bdk_sp_link(c->comm, &L)) return fail("sequence-parallel exchange: the communicator lost its peers");
        const int rows_local = M / c->tp;
        int seq_p = 0;                                         // pending row-split Linear's hand-off (0: none)
        const void* pend_bias = nullptr;
        for (int b = 0; b < c->hNB; ++b) {
            const std::string pre = "head.blk" + std::to_string(b) + ".";
            const int base = (b / sw) * nc * D;
            LnModSpArgs l1;
            l1.ln.X = c->wptr("head.X");
            l1.ln.ada = ada; l1.ln.ada_ld = c->hNada;
            l1.ln.gate_off = ((b - 1 < 0 ? 0 : b - 1) / sw) * nc * D + (nc - 1) * D;
            l1.ln.scale_off = base; l1.ln.shift_off = base + D;
            l1.ln.h_frag = nullptr; l1.ln.M = M; l1.ln.D = D; l1.ln.RB = RB; l1.ln.eps = 1e-6f;
            l1.ln.ln_w = (const float*)c->ptr(pre + "ln1_w"); l1.ln.ln_b = (const float*)c->ptr(pre + "ln1_b");
            l1.L = L; l1.rows_local = rows_local;
            l1.part = seq_p ? (const float*)c->ptr("head.tp_part") : nullptr; l1.bias = pend_bias; l1.seq_p = seq_p;
            l1.signal_p = c->geti("tune.sp_gsig", 0) ? 0 : 1;
            l1.seq_h = bdk_sp_next_seq(c->comm);
            if (l1.seq_h < 0) return fail("sequence-parallel exchange: more than 65535 hand-offs since the last bd_head_cond / bd_head_sample");
            BD_TRY(bdk_ln_mod_sp(l1, st));
            HeadAttnArgs at;
            BD_TRY(sp_arm_wait(c, l1.seq_h, st));
            BD_TRY(linear(c, "head.qkv", bdk_sp_hbuf(c->comm), RB, wref(c, pre + "wqkv"), 3 * Dl, D, gq, "head.qkv_part", "head.qkv_bf",
                          c->ptr(pre + "bqkv"), Mp, &at.qkv, st));
            BD_TRY(finalize_if_many(c, &at.qkv, "head.qkv_bf", M, st));
            at.o_frag = c->wptr("head.attn_frag"); at.nseq = M / c->Pn; at.dh = (int)c->geti("head.dh", 128); at.nhead = Dl / at.dh; at.D = Dl; at.RB = RB; at.P = c->Pn;
            BD_TRY(bdk_head_attn(at, st));
            BD_TRY(linear_rowsplit_sp(c, "head.wo", c->ptr("head.attn_frag"), RB, wref(c, pre + "wo"), D, Dl, go, "head.br_part", "head.tp_part", M, &seq_p, st));
            LnModSpArgs l2 = l1;
            l2.ln.gate_off = base + 2 * D; l2.ln.scale_off = base + 3 * D; l2.ln.shift_off = base + 4 * D;
            l2.ln.ln_w = (const float*)c->ptr(pre + "ln2_w"); l2.ln.ln_b = (const float*)c->ptr(pre + "ln2_b");
            l2.part = (const float*)c->ptr("head.tp_part"); l2.bias = c->ptr(pre + "bo"); l2.seq_p = seq_p;
            l2.seq_h = bdk_sp_next_seq(c->comm);
            if (l2.seq_h < 0) return fail("sequence-parallel exchange: more than 65535 hand-offs since the last bd_head_cond / bd_head_sample");
            BD_TRY(bdk_ln_mod_sp(l2, st));
            BD_TRY(sp_arm_wait(c, l2.seq_h, st));
            if (g1.S == 1 || c->geti("tune.w1_fused", g1.S > 2 ? 0 : 1)) {
                BD_TRY(gemm(c, "head.w1", bdk_sp_hbuf(c->comm), RB, wref(c, pre + "w1"), 2 * Hl, D, g1.S, g1.code(), BD_EPI_SWIGLU,
                            (float*)c->wptr("head.w1_part"), c->wptr("head.act_frag"), c->ptr(pre + "b1"), st));
            } else {                                           // many short K slices + a row-parallel SwiGLU pass over the slabs
                BD_TRY(gemm(c, "head.w1", bdk_sp_hbuf(c->comm), RB, wref(c, pre + "w1"), 2 * Hl, D, g1.S, g1.code(), BD_EPI_PARTIAL,
                            (float*)c->wptr("head.w1_part"), nullptr, nullptr, st));
                SwigluArgs sw_;
                sw_.up = part(c, "head.w1_part", c->ptr(pre + "b1"), g1.S, 2 * Hl, Mp);
                sw_.act_frag = c->wptr("head.act_frag"); sw_.M = M; sw_.F = Hl; sw_.RB = RB; sw_.interleaved = 1;
                BD_TRY(bdk_swiglu_rows(sw_, st));
            }
            BD_TRY(linear_rowsplit_sp(c, "head.w2", c->ptr("head.act_frag"), RB, wref(c, pre + "w2"), D, Hl, g2, "head.br_part", "head.tp_part", M, &seq_p, st));
            pend_bias = c->ptr(pre + "b2");
        }
        HeadFinalSpArgs fs;
        HeadFinalArgs& fa = fs.f;
        fa.X = c->ptr("head.X");
        fa.pend = Partial{nullptr, nullptr, 0, 0, 0};
        fa.ada = ada; fa.ada_ld = c->hNada;
        fa.gate_off = ((c->hNB - 1) / sw) * nc * D + (nc - 1) * D;
        fa.scale_off = c->hNA * nc * D; fa.shift_off = c->hNA * nc * D + D;
        fa.lin_w = c->ptr("head.lin_w"); fa.lin_b = c->ptr("head.lin_b");
        fa.xt = (float*)c->wptr("head.xt");
        fa.noise = (const float*)c->ptr("head.noise");
        fa.noise_step_stride = (long long)(n_steps + 1) * c->BP * c->hC;
        fa.eval_index = i; fa.state = state;
        fa.pred_out = nullptr; fa.tok_cur = nullptr; fa.tok_all = nullptr;       // written by tok_finish, on every rank
        fa.T = c->hT; fa.P = c->Pn;
        fa.xhat_out = c->geti("rt.dump_xhat", 0) ? (float*)c->wptr("head.xhat") : nullptr;
        fa.sc = c->sched[i];
        fa.BP = c->BP; fa.D = D; fa.C = c->hC; fa.M = M; fa.eps_ln = 1e-6f; fa.sigmoid = (int)c->geti("head.sigmoid", 1);
        fa.cfg_table = (const float*)c->optr("head.cfg_table");
        fa.tok_branches = 1;
        if (chain_next) { fa.X_next = c->wptr("head.X"); fa.in_w = c->ptr("head.in_w"); fa.in_b = c->ptr("head.in_b"); }
        fs.L = L; fs.part = (const float*)c->ptr("head.tp_part"); fs.bias = pend_bias; fs.seq_p = seq_p; fs.bp_local = c->BP / c->tp;
        fs.seq_f = 0; fs.signal_p = c->geti("tune.sp_gsig", 0) ? 0 : 1;
        if (fa.sc.is_final) {
            fs.seq_f = bdk_sp_next_seq(c->comm);
            if (fs.seq_f < 0) return fail("sequence-parallel exchange: more than 65535 hand-offs since the last bd_head_cond / bd_head_sample");
        }
        BD_TRY(bdk_head_final_sp(fs, st));
        if (fa.sc.is_final) {
            TokFinishArgs tf;
            tf.L = L; tf.seq_f = fs.seq_f;
            tf.xt = (float*)c->wptr("head.xt"); tf.pred_out = (float*)c->wptr("head.pred"); tf.tok_cur = (float*)c->wptr("head.tok_cur");
            tf.tok_all = (float*)const_cast<void*>(c->optr("head.tok_all"));
            tf.state = state; tf.BP = c->BP; tf.C = c->hC; tf.T = c->hT; tf.P = c->Pn;
            tf.tok_branches = (c->geti("proj.rows_all", 0) && c->has_proj) ? c->branches : 1;
            BD_TRY(bdk_tok_finish(tf, st));
        }
        return 0;
    }
    Partial br{nullptr, nullptr, 0, 0, 0};                    // pending gated branch output (wo / w2)
    for (int b = 0; b < c->hNB; ++b) {
        const std::string pre = "head.blk" + std::to_string(b) + ".";
        const int base = (b / sw) * nc * D;
        LnModArgs l1;
        l1.X = c->wptr("head.X");
        l1.pend = br;                                      // w2 output of the previous block (none for block 0)
        l1.ada = ada; l1.ada_ld = c->hNada;
        l1.gate_off = ((b - 1 < 0 ? 0 : b - 1) / sw) * nc * D + (nc - 1) * D;
        l1.scale_off = base; l1.shift_off = base + D;
        l1.h_frag = c->wptr("head.h_frag"); l1.M = M; l1.D = D; l1.RB = RB; l1.eps = 1e-6f;
        if (c->fp8a) l1.a8_scale = (float*)c->wptr("head.h_scale");
        l1.wave_rows = (int)c->geti("tune.ln_rows", M >= 1024 ? 1 : 0);   // the ImageNet batch: a wave per row (bd_rows.hip)
        LnModArgs l2 = l1;
        if (!mlp) {
            l1.ln_w = (const float*)c->ptr(pre + "ln1_w"); l1.ln_b = (const float*)c->ptr(pre + "ln1_b");
            BD_TRY(bdk_ln_mod(l1, st));
            HeadAttnArgs at;
            BD_TRY(linear(c, "head.qkv", c->ptr("head.h_frag"), RB, wref(c, pre + "wqkv", "head.h_scale"), 3 * Dl, D, gq, "head.qkv_part", "head.qkv_bf",
                          c->ptr(pre + "bqkv"), Mp, &at.qkv, st));
            BD_TRY(finalize_if_many(c, &at.qkv, "head.qkv_bf", M, st));
            at.o_frag = c->wptr("head.attn_frag"); at.nseq = M / c->Pn; at.dh = (int)c->geti("head.dh", 128); at.nhead = Dl / at.dh; at.D = Dl; at.RB = RB; at.P = c->Pn;
            BD_TRY(bdk_head_attn(at, st));
            BD_TRY(linear_rowsplit(c, "head.wo", c->ptr("head.attn_frag"), RB, wref(c, pre + "wo"), D, Dl, go, "head.br_part", "head.br_bf",
                                   "head.tp_part", c->ptr(pre + "bo"), Mp, M, &l2.pend, st));
            l2.gate_off = base + 2 * D; l2.scale_off = base + 3 * D; l2.shift_off = base + 4 * D;
        }
        // MLP head (diff_head.py:133-137): the block IS the second half -- h = norm(x) * (1 + scale) + shift with the block's
        // (scale, shift) = chunks 0, 1 and the previous block's gated w2 output still pending, exactly l1's offsets above
        l2.ln_w = (const float*)c->ptr(pre + "ln2_w"); l2.ln_b = (const float*)c->ptr(pre + "ln2_b");
        BD_TRY(bdk_ln_mod(l2, st));
        // Linear -> chunk(2) -> silu(h1)*h2.  Fused epilogue (on the last-arriving K-slice when split) writes the next
        // operand: 46.9 + 22.9 us (w1 + w2) against 40.8 + 9.5 + 26.0 us for slabs + swiglu_rows on the same MI355X
        // (tune.w1_fused = 0 selects the latter).
        // (256-row passes: slabs + swiglu_rows.  The fused epilogue behind the 256-row kernel's in-launch reduction -- "tune.w1_fused" = 1 at
        // 512 rows -- measured 25 us SLOWER per launch: the last arriver of a tile writes the whole SwiGLU tile with 2-byte stores; 2390 vs 2250 us
        // per evaluation, profiles/r05_head_sweep_b4.log)
        if (g1.S == 1 || c->geti("tune.w1_fused", (c->Mpad % 256 == 0 || g1.S > 2) ? 0 : 1)) {
            BD_TRY(gemm(c, "head.w1", c->ptr("head.h_frag"), RB, wref(c, pre + "w1", "head.h_scale"), 2 * Hl, D, g1.S, g1.code(), BD_EPI_SWIGLU,
                        (float*)c->wptr("head.w1_part"), c->wptr("head.act_frag"), c->ptr(pre + "b1"), st));
        } else {
            BD_TRY(gemm(c, "head.w1", c->ptr("head.h_frag"), RB, wref(c, pre + "w1", "head.h_scale"), 2 * Hl, D, g1.S, g1.code(), BD_EPI_PARTIAL,
                        (float*)c->wptr("head.w1_part"), nullptr, nullptr, st));
            SwigluArgs sw_;
            sw_.up = part(c, "head.w1_part", c->ptr(pre + "b1"), g1.S, 2 * Hl, Mp);
            sw_.act_frag = c->wptr("head.act_frag"); sw_.M = M; sw_.F = Hl; sw_.RB = RB; sw_.interleaved = 1;
            BD_TRY(bdk_swiglu_rows(sw_, st));
        }
        BD_TRY(linear_rowsplit(c, "head.w2", c->ptr("head.act_frag"), RB, wref(c, pre + "w2"), D, Hl, g2, "head.br_part", "head.br_bf",
                               "head.tp_part", c->ptr(pre + "b2"), Mp, M, &br, st));
    }
    HeadFinalArgs fa;
    fa.X = c->ptr("head.X");
    fa.pend = br;
    fa.ada = ada; fa.ada_ld = c->hNada;
    fa.gate_off = ((c->hNB - 1) / sw) * nc * D + (nc - 1) * D;
    fa.scale_off = c->hNA * nc * D; fa.shift_off = c->hNA * nc * D + D;
    fa.lin_w = c->ptr("head.lin_w"); fa.lin_b = c->ptr("head.lin_b");
    fa.xt = (float*)c->wptr("head.xt");
    fa.noise = (const float*)c->ptr("head.noise");
    fa.noise_step_stride = (long long)(n_steps + 1) * c->BP * c->hC;
    fa.eval_index = i; fa.state = state;
    fa.pred_out = (float*)c->wptr("head.pred"); fa.tok_cur = (float*)c->wptr("head.tok_cur");
    fa.tok_all = (float*)const_cast<void*>(c->optr("head.tok_all"));
    fa.T = c->hT; fa.P = c->Pn;
    fa.xhat_out = c->geti("rt.dump_xhat", 0) ? (float*)c->wptr("head.xhat") : nullptr;
    fa.sc = c->sched[i];
    fa.BP = c->BP; fa.D = D; fa.C = c->hC; fa.M = M; fa.eps_ln = 1e-6f; fa.sigmoid = (int)c->geti("head.sigmoid", 1);
    fa.cfg_table = (const float*)c->optr("head.cfg_table");
    fa.tok_branches = (c->geti("proj.rows_all", 0) && c->has_proj) ? c->branches : 1;
    if (chain_next) { fa.X_next = c->wptr("head.X"); fa.in_w = c->ptr("head.in_w"); fa.in_b = c->ptr("head.in_b"); }
    BD_TRY(bdk_head_final(fa, st));
    return 0;
}

static int head_sample(bd_ctx* c, hipStream_t st) {
    if (c->sched.empty()) return fail("bd_head_sample: no schedule set");
    const int n_steps = (int)c->sched.size() - 1;
    InitLatentArgs ia{(float*)c->wptr("head.xt"), (const float*)c->ptr("head.noise"),
                      (long long)(n_steps + 1) * c->BP * c->hC, (const BdStepState*)c->ptr("state"), c->BP * c->hC};
    BD_TRY(bdk_init_latent(ia, st));
    BD_TRY(head_cond(c, st));
    const int G = c->adaG;
    for (int i = 0; i <= n_steps; ++i) {
        if (G > 1 && c->y_ready) {
            if (i % G == 0) BD_TRY(head_ada_group(c, i / G, st));
            // (column-split projection: the group's slot of the double-buffered modulation tensor)
            BD_TRY(head_eval(c, i, st, (c->ada_slots == 2 ? ((i / G) & 1) * G : 0) + i % G, /*x0_ready=*/i > 0, /*chain_next=*/true));
        } else {
            BD_TRY(head_eval(c, i, st, -1, /*x0_ready=*/i > 0, /*chain_next=*/true));
        }
    }
    return 0;
}

// imagenet MLPConnector (model_parallel.py:73-75) + emb_norm (:344): the normalised bf16 value starts the residual stream
static int projector_in(bd_ctx* c, hipStream_t st) {
    const int D = (int)c->geti("proj.D"), hid = (int)c->geti("proj.hid");
    InProjFc1Args f1{(const float*)c->ptr("head.tok_cur"), c->ptr("proj.w1"), c->ptr("proj.b1"), c->wptr("proj.h_frag"),
                     c->prows, hid, (int)c->geti("proj.C"), c->RBp};
    BD_TRY(bdk_in_proj_fc1(f1, st));
    const GemmCfg& g = c->cfg("proj.fc2");
    InRmsArgs e;
    BD_TRY(linear(c, "proj.fc2", c->ptr("proj.h_frag"), c->RBp, wref(c, "proj.w2"), D, hid, g, "proj.part", "proj.out_bf",
                  c->ptr("proj.b2"), c->BPpad, &e.pend, st));
    e.R = (float*)c->wptr("llm.R"); e.init_from_pend = 1; e.renorm_to_R = 1; e.w = (const float*)c->ptr("llm.emb_norm");
    e.a_frag = nullptr; e.hidden_out = nullptr; e.cond_frag = nullptr; e.pos = nullptr;
    e.state = (const BdStepState*)c->ptr("state"); e.M = c->prows; e.D = D; e.RB = c->RBp; e.P = c->Pn;
    e.eps = (float)c->getf("llm.eps", 1e-6);
    BD_TRY(bdk_in_rms(e, st));
    return 0;
}

static int projector(bd_ctx* c, hipStream_t st) {
    if (c->geti("proj.variant", 0)) return projector_in(c, st);
    const int D = (int)c->geti("proj.D");
    ProjFc1Args f1{(const float*)c->ptr("head.tok_cur"), c->ptr("proj.w1"), c->ptr("proj.b1"), c->wptr("proj.h_frag"),
                   c->BP, D, (int)c->geti("proj.C"), c->RBp};
    BD_TRY(bdk_proj_fc1(f1, st));
    const GemmCfg& g = c->cfg("proj.fc2");
    Partial fc2;
    BD_TRY(linear(c, "proj.fc2", c->ptr("proj.h_frag"), c->RBp, wref(c, "proj.w2"), D, D, g, "proj.part", "proj.out_bf",
                  c->ptr("proj.b2"), c->BPpad, &fc2, st));
    EmbedFinalizeArgs ef;
    ef.fc2 = fc2;
    ef.pos = (const float*)c->ptr("pos"); ef.R = (float*)c->wptr("llm.R");
    ef.state = (const BdStepState*)c->ptr("state"); ef.BP = c->BP; ef.P = c->Pn; ef.D = D; ef.branches = c->branches;
    BD_TRY(bdk_embed_finalize(ef, st));
    return 0;
}

// forward_model for one 16-token block of the imagenet transformer (model_parallel.py:342-350, layers_parallel.py:229-241)
static int llm_step_in(bd_ctx* c, hipStream_t st) {
    const int D = c->lD, F = c->lF, Mp = c->Mpad, RB = c->RB, M = c->M, nh = c->lnh;
    const int nseq = c->branches * c->B;
    const float eps = (float)c->getf("llm.eps", 1e-6);
    BdStepState* state = (BdStepState*)c->wptr("state");
    const GemmCfg &gq = c->cfg("llm.qkv"), &go = c->cfg("llm.o"), &gg = c->cfg("llm.gu"), &gd = c->cfg("llm.down");
    const size_t layer_elems = (size_t)nseq * nh * c->lLmax * 64;
    Partial br{nullptr, nullptr, 0, 0, 0};
    // "rt.in_first" = 1: a block of the FIRST forward_model call (model_parallel.py:386-388 / model.py:372-377): llm.R holds the raw
    // fp32 class-embedding / query-token rows, emb_norm runs here (the decode steps get it from projector_in), the residual stream
    // stays fp32; "rt.llm_causal": causal mask inside the block (the class tokens); "rt.no_advance": the host sets the cache lengths
    const int first = (int)c->geti("rt.in_first", 0), causal = (int)c->geti("rt.llm_causal", 0);
    InRmsArgs r1;
    r1.R = (float*)c->wptr("llm.R"); r1.init_from_pend = 0; r1.renorm_to_R = 0;
    r1.a_frag = c->wptr("llm.a_frag"); r1.hidden_out = nullptr; r1.cond_frag = nullptr; r1.pos = nullptr; r1.state = state;
    r1.M = M; r1.D = D; r1.RB = RB; r1.P = c->Pn; r1.eps = eps; r1.f32_stream = first;
    if (first) {
        InRmsArgs e = r1;
        e.pend = br; e.renorm_to_R = 1; e.a_frag = nullptr; e.w = (const float*)c->ptr("llm.emb_norm");
        BD_TRY(bdk_in_rms(e, st));
    }
    for (int l = 0; l < c->lL; ++l) {
        const std::string pre = "llm.l" + std::to_string(l) + ".";
        r1.pend = br;
        r1.w = (const float*)c->ptr(pre + "in_norm");
        BD_TRY(bdk_in_rms(r1, st));
        InQkvPostArgs qa;
        BD_TRY(linear(c, "llm.qkv", c->ptr("llm.a_frag"), RB, wref(c, pre + "wqkv"), c->lNqkv, D, gq, "llm.qkv_part", "llm.qkv_bf",
                      nullptr, Mp, &qa.qkv, st));
        qa.rope = (const float*)c->ptr("llm.rope2d"); qa.q_out = c->wptr("llm.q");
        qa.k_cache = (bf16_t*)c->wptr("llm.k_cache") + l * layer_elems;
        qa.v_cache = (bf16_t*)c->wptr("llm.vt_cache") + l * layer_elems;
        qa.state = state; qa.M = M; qa.P = c->Pn; qa.nh = nh; qa.Lmax = c->lLmax;
        BD_TRY(bdk_in_qkv_post(qa, st));
        InAttnArgs aa{c->ptr("llm.q"), qa.k_cache, qa.v_cache, c->wptr("llm.attn_frag"), state, nseq, c->Pn, nh, c->lLmax, RB};
        aa.causal = causal;
        BD_TRY(bdk_in_attn(aa, st));
        InRmsArgs r2 = r1;
        BD_TRY(linear(c, "llm.o", c->ptr("llm.attn_frag"), RB, wref(c, pre + "wo"), D, D, go, "llm.br_part", "llm.br_bf",
                      nullptr, Mp, &r2.pend, st));
        r2.w = (const float*)c->ptr(pre + "post_norm");
        BD_TRY(bdk_in_rms(r2, st));
        BD_TRY(gemm(c, "llm.gu", c->ptr("llm.a_frag"), RB, wref(c, pre + "wgu"), 2 * F, D, gg.S, gg.code(), BD_EPI_SWIGLU,
                    (float*)c->wptr("llm.gu_part"), c->wptr("llm.act_frag"), nullptr, st));
        BD_TRY(linear(c, "llm.down", c->ptr("llm.act_frag"), RB, wref(c, pre + "wdown"), D, F, gd, "llm.br_part", "llm.br_bf",
                      nullptr, Mp, &br, st));
    }
    if (!c->geti("rt.no_advance", 0)) {
        StepAdvanceArgs sa{state, nseq < BD_MAX_SEQ ? nseq : BD_MAX_SEQ, c->Pn};
        BD_TRY(bdk_step_advance(sa, st));
    }
    InRmsArgs rf = r1;
    rf.pend = br; rf.w = (const float*)c->ptr("llm.final_norm"); rf.a_frag = nullptr;
    rf.hidden_out = (float*)c->wptr("llm.hidden");
    const bool emit = c->geti("rt.emit_cond", 1) != 0 && c->has_head;
    rf.cond_frag = emit ? c->wptr("head.cond_frag") : nullptr;
    rf.pos = emit ? (const float*)c->ptr("pos") : nullptr;
    BD_TRY(bdk_in_rms(rf, st));
    return 0;
}

// The decode step with sequence-parallel row kernels (bd_sp.hip rms_sp_kernel): a rank owns rows / tp rows of the fp32 residual stream; per
// layer  rms_sp -> qkv (waits for the operand rows) -> q/k norm + RoPE + append -> attention -> o_proj (pushes partial rows to the owners)
// -> rms_sp -> gate/up (waits) -> down_proj (pushes) ; the final norm's rows travel as fp32 and every rank finishes hidden state and the
// next patch's condition itself (sp_final_rows_kernel).  No stand-alone exchange kernel: 2 L + 1 hand-offs of operand rows and 2 L of
// partial rows instead of 2 L all-reduce kernels with two flag rounds each + 2 L replicated RMSNorms.  Values: the all-reduce form's,
// bit for bit (the partials are summed in rank order and rounded once to bf16 by the owner).
static int llm_step_sp(bd_ctx* c, hipStream_t st) {
    const int D = c->lD, F = c->lFl, Mp = c->Mpad, RB = c->RB, M = c->M, nh = c->lnhl, nkv = c->lnkvl;
    const int nseq = c->branches * c->B;
    const float eps = (float)c->getf("llm.eps", 1e-6);
    BdStepState* state = (BdStepState*)c->wptr("state");
    const GemmCfg &gq = c->cfg("llm.qkv"), &go = c->cfg("llm.o"), &gg = c->cfg("llm.gu"), &gd = c->cfg("llm.down");
    const size_t layer_elems = (size_t)nseq * nkv * c->lLmax * 128;
    BdSpLink L;
    if (!bdk_sp_link(c->comm, &L)) return fail("sequence-parallel exchange: the communicator lost its peers");
    BD_TRY(bdk_sp_begin(c->comm, st));                      // sequence numbers restart with every replay of the step
    const char* const too_many = "sequence-parallel exchange: more than 65535 hand-offs in one Qwen3 step";
    int seq_p = 0;
    RmsSpArgs a1;
    a1.r.R = (float*)c->wptr("llm.R");
    a1.r.pend = Partial{nullptr, nullptr, 0, 0, 0};
    a1.r.a_frag = nullptr; a1.r.hidden_out = nullptr; a1.r.cond_frag = nullptr; a1.r.pos = nullptr; a1.r.state = state;
    a1.r.M = M; a1.r.D = D; a1.r.RB = RB; a1.r.P = c->Pn; a1.r.eps = eps; a1.r.bf16_stream = 0;
    a1.L = L; a1.rows_local = M / c->tp; a1.signal_p = c->geti("tune.sp_gsig", 0) ? 0 : 1;
    for (int l = 0; l < c->lL; ++l) {
        const std::string pre = "llm.l" + std::to_string(l) + ".";
        a1.r.w = c->ptr(pre + "in_norm");
        a1.part = seq_p ? (const float*)c->ptr("llm.tp_part") : nullptr; a1.seq_p = seq_p;
        if ((a1.seq_h = bdk_sp_next_seq(c->comm)) < 0) return fail(too_many);
        BD_TRY(bdk_rms_sp(a1, st));
        QkvPostArgs qa;
        BD_TRY(sp_arm_wait(c, a1.seq_h, st));
        BD_TRY(linear(c, "llm.qkv", bdk_sp_hbuf(c->comm), RB, wref(c, pre + "wqkv"), c->lNqkv, D, gq, "llm.qkv_part", "llm.qkv_bf", nullptr, Mp, &qa.qkv, st));
        qa.qn_w = c->ptr(pre + "q_norm"); qa.kn_w = c->ptr(pre + "k_norm");
        qa.cos = (const float*)c->ptr("llm.cos"); qa.sin = (const float*)c->ptr("llm.sin");
        qa.q_out = c->wptr("llm.q");
        qa.k_cache = (bf16_t*)c->wptr("llm.k_cache") + l * layer_elems;
        qa.vt_cache = (bf16_t*)c->wptr("llm.vt_cache") + l * layer_elems;
        qa.state = state; qa.M = M; qa.P = c->Pn; qa.nh = nh; qa.nkv = nkv; qa.Lmax = c->lLmax; qa.eps = eps; qa.rope_bf16 = 0;
        BD_TRY(bdk_qkv_post(qa, st));
        LlmAttnArgs aa;
        aa.q = c->ptr("llm.q"); aa.k_cache = qa.k_cache; aa.vt_cache = qa.vt_cache;
        aa.o_part = (float*)c->wptr("llm.attn_opart"); aa.ml_part = (float*)c->wptr("llm.attn_ml");
        aa.o_frag = c->wptr("llm.attn_frag"); aa.state = state;
        aa.nseq = nseq; aa.P = c->Pn; aa.nh = nh; aa.nkv = nkv; aa.Lmax = c->lLmax; aa.splits = c->lsplits; aa.RB = RB; aa.causal = 0;
        BD_TRY(bdk_llm_attn(aa, st));
        BD_TRY(linear_rowsplit_sp(c, "llm.o", c->ptr("llm.attn_frag"), RB, wref(c, pre + "wo"), D, nh * 128, go, "llm.br_part", "llm.tp_part", M, &seq_p, st));
        RmsSpArgs a2 = a1;
        a2.r.w = c->ptr(pre + "post_norm");
        a2.part = (const float*)c->ptr("llm.tp_part"); a2.seq_p = seq_p;
        if ((a2.seq_h = bdk_sp_next_seq(c->comm)) < 0) return fail(too_many);
        BD_TRY(bdk_rms_sp(a2, st));
        BD_TRY(sp_arm_wait(c, a2.seq_h, st));
        BD_TRY(gemm(c, "llm.gu", bdk_sp_hbuf(c->comm), RB, wref(c, pre + "wgu"), 2 * F, D, gg.S, gg.code(), BD_EPI_SWIGLU,
                    (float*)c->wptr("llm.gu_part"), c->wptr("llm.act_frag"), nullptr, st));
        BD_TRY(linear_rowsplit_sp(c, "llm.down", c->ptr("llm.act_frag"), RB, wref(c, pre + "wdown"), D, F, gd, "llm.br_part", "llm.tp_part", M, &seq_p, st));
    }
    if (!c->geti("rt.no_advance", 0)) {
        StepAdvanceArgs sa{state, nseq, c->Pn};
        BD_TRY(bdk_step_advance(sa, st));
    }
    RmsSpArgs af = a1;
    af.r.w = c->ptr("llm.final_norm");
    af.part = seq_p ? (const float*)c->ptr("llm.tp_part") : nullptr; af.seq_p = seq_p;
    if ((af.seq_h = bdk_sp_next_seq(c->comm)) < 0) return fail(too_many);
    af.final_rows = 1; af.final_off = (long long)Mp * D * 2;
    BD_TRY(bdk_rms_sp(af, st));
    SpFinalRowsArgs fr;
    if (!bdk_sp_hwait(c->comm, af.seq_h, M, &fr.w)) return fail("sequence-parallel exchange: no flag block");
    fr.rows = (const float*)((const char*)bdk_sp_hbuf(c->comm) + af.final_off);
    fr.hidden_out = (float*)c->wptr("llm.hidden");
    const bool emit = c->geti("rt.emit_cond", 1) != 0 && c->has_head;
    fr.cond_frag = emit ? c->wptr("head.cond_frag") : nullptr;
    fr.pos = emit ? (const float*)c->ptr("pos") : nullptr;
    fr.state = state; fr.M = M; fr.D = D; fr.RB = RB; fr.P = c->Pn;
    BD_TRY(bdk_sp_final_rows(fr, st));
    return 0;
}

static int llm_step(bd_ctx* c, hipStream_t st) {
    if (c->lvariant == 1) return llm_step_in(c, st);
    // (the once-per-image prefill -- causal blocks of prompt tokens, bf16 hidden states -- keeps the all-reduce form)
    if (c->sp_llm && !c->geti("rt.llm_causal", 0) && !c->geti("rt.llm_bf16", 0)) return llm_step_sp(c, st);
    // nh / nkv / F: this rank's q heads, kv heads and FFN features (the full counts at tp = 1)
    const int D = c->lD, F = c->lFl, Mp = c->Mpad, RB = c->RB, M = c->M, nh = c->lnhl, nkv = c->lnkvl;
    const int nseq = c->branches * c->B;
    const float eps = (float)c->getf("llm.eps", 1e-6);
    BdStepState* state = (BdStepState*)c->wptr("state");
    const GemmCfg &gq = c->cfg("llm.qkv"), &go = c->cfg("llm.o"), &gg = c->cfg("llm.gu"), &gd = c->cfg("llm.down");
    // prefill (once per image, eager): the same step over a block of PROMPT tokens -- causal mask, bf16 hidden states
    // (t2i_pipeline.py:199-217); the host sets the per-sequence cache lengths itself between blocks
    const int causal = (int)c->geti("rt.llm_causal", 0), bf16s = (int)c->geti("rt.llm_bf16", 0);
    const size_t layer_elems = (size_t)nseq * nkv * c->lLmax * 128;
    Partial br{nullptr, nullptr, 0, 0, 0};                    // pending branch output (o_proj / down_proj)
    for (int l = 0; l < c->lL; ++l) {
        const std::string pre = "llm.l" + std::to_string(l) + ".";
        RmsArgs r1;
        r1.R = (float*)c->wptr("llm.R");
        r1.pend = br;                                      // down_proj output of the previous layer (none for layer 0)
        r1.w = c->ptr(pre + "in_norm"); r1.a_frag = c->wptr("llm.a_frag");
        r1.hidden_out = nullptr; r1.cond_frag = nullptr; r1.pos = nullptr; r1.state = state;
        r1.M = M; r1.D = D; r1.RB = RB; r1.P = c->Pn; r1.eps = eps; r1.bf16_stream = bf16s;
        if (c->fp8a) r1.a8_scale = (float*)c->wptr("llm.a_scale");
        BD_TRY(bdk_rms(r1, st));

        QkvPostArgs qa;
        BD_TRY(linear(c, "llm.qkv", c->ptr("llm.a_frag"), RB, wref(c, pre + "wqkv", "llm.a_scale"), c->lNqkv, D, gq, "llm.qkv_part", "llm.qkv_bf",
                      nullptr, Mp, &qa.qkv, st));
        qa.qn_w = c->ptr(pre + "q_norm"); qa.kn_w = c->ptr(pre + "k_norm");
        qa.cos = (const float*)c->ptr("llm.cos"); qa.sin = (const float*)c->ptr("llm.sin");
        qa.q_out = c->wptr("llm.q");
        qa.k_cache = (bf16_t*)c->wptr("llm.k_cache") + l * layer_elems;
        qa.vt_cache = (bf16_t*)c->wptr("llm.vt_cache") + l * layer_elems;
        qa.state = state; qa.M = M; qa.P = c->Pn; qa.nh = nh; qa.nkv = nkv; qa.Lmax = c->lLmax; qa.eps = eps; qa.rope_bf16 = bf16s;
        BD_TRY(bdk_qkv_post(qa, st));
        LlmAttnArgs aa;
        aa.q = c->ptr("llm.q"); aa.k_cache = qa.k_cache; aa.vt_cache = qa.vt_cache;
        aa.o_part = (float*)c->wptr("llm.attn_opart"); aa.ml_part = (float*)c->wptr("llm.attn_ml");
        aa.o_frag = c->wptr("llm.attn_frag"); aa.state = state;
        aa.nseq = nseq; aa.P = c->Pn; aa.nh = nh; aa.nkv = nkv; aa.Lmax = c->lLmax; aa.splits = c->lsplits; aa.RB = RB; aa.causal = causal;
        BD_TRY(bdk_llm_attn(aa, st));

        RmsArgs r2 = r1;
        BD_TRY(linear_rowsplit(c, "llm.o", c->ptr("llm.attn_frag"), RB, wref(c, pre + "wo"), D, nh * 128, go, "llm.br_part", "llm.br_bf",
                               "llm.tp_part", nullptr, Mp, M, &r2.pend, st));
        r2.w = c->ptr(pre + "post_norm");
        BD_TRY(bdk_rms(r2, st));
        BD_TRY(gemm(c, "llm.gu", c->ptr("llm.a_frag"), RB, wref(c, pre + "wgu", "llm.a_scale"), 2 * F, D, gg.S, gg.code(), BD_EPI_SWIGLU,
                        (float*)c->wptr("llm.gu_part"), c->wptr("llm.act_frag"), nullptr, st));
        BD_TRY(linear_rowsplit(c, "llm.down", c->ptr("llm.act_frag"), RB, wref(c, pre + "wdown"), D, F, gd, "llm.br_part", "llm.br_bf",
                               "llm.tp_part", nullptr, Mp, M, &br, st));
    }
    if (!c->geti("rt.no_advance", 0)) {
        StepAdvanceArgs sa{state, nseq, c->Pn};
        BD_TRY(bdk_step_advance(sa, st));                   // step+1 / kv_len += P: the next patch's position
    }
    RmsArgs rf;
    rf.R = (float*)c->wptr("llm.R");
    rf.pend = br;
    rf.w = c->ptr("llm.final_norm"); rf.a_frag = nullptr;
    rf.hidden_out = (float*)c->wptr("llm.hidden");
    const bool emit = c->geti("rt.emit_cond", 1) != 0 && c->has_head;
    rf.cond_frag = emit ? c->wptr("head.cond_frag") : nullptr;
    rf.pos = emit ? (const float*)c->ptr("pos") : nullptr;
    rf.state = state; rf.M = M; rf.D = D; rf.RB = RB; rf.P = c->Pn; rf.eps = eps; rf.bf16_stream = bf16s;
    BD_TRY(bdk_rms(rf, st));
    return 0;
}

struct ResetArgs { BdStepState* state; int kv[BD_MAX_SEQ]; int nseq; };
__global__ void step_reset_kernel(ResetArgs a) {
    if (threadIdx.x == 0) a.state->step = 0;
    if ((int)threadIdx.x < BD_MAX_SEQ) a.state->kv_len[threadIdx.x] = ((int)threadIdx.x < a.nseq) ? a.kv[threadIdx.x] : 0;
}

extern "C" {

#define BD_GUARD(...)                                                        \
    if (!c->bound) return fail("context not bound (bd_ctx_bind)");           \
    try { __VA_ARGS__ } catch (const std::exception& e) { return fail(e.what()); }

int bd_head_cond(bd_ctx* c, void* s) { BD_GUARD(return head_cond(c, (hipStream_t)s);) }
int bd_head_eval(bd_ctx* c, int i, void* s) {
    BD_GUARD(const bool chain = c->geti("rt.chain", 0) != 0;      // debugging: drive head_sample's chained form one evaluation at a time
             return head_eval(c, i, (hipStream_t)s, -1, chain && i > 0, chain);)
}
int bd_head_sample(bd_ctx* c, void* s) { BD_GUARD(return head_sample(c, (hipStream_t)s);) }
int bd_projector(bd_ctx* c, void* s) { BD_GUARD(return projector(c, (hipStream_t)s);) }
int bd_llm_step(bd_ctx* c, void* s) { BD_GUARD(return llm_step(c, (hipStream_t)s);) }

int bd_step_reset(bd_ctx* c, const int* kv_len, int nseq, void* s) {
    BD_GUARD(
        if (nseq > BD_MAX_SEQ) return fail("bd_step_reset: nseq > 64");
        ResetArgs a; a.state = (BdStepState*)c->wptr("state"); a.nseq = nseq;
        for (int i = 0; i < BD_MAX_SEQ; ++i) a.kv[i] = i < nseq ? kv_len[i] : 0;
        BD_LAUNCH(step_reset_kernel, dim3(1), dim3(64), 0, (hipStream_t)s, a);
        return bd_launch_status() == 0 ? 0 : fail("step_reset launch failed");)
}

int bd_prof_enable(bd_ctx* c, int on) {
    for (auto& r : c->prof) { hipEventDestroy(r.e0); hipEventDestroy(r.e1); }
    c->prof.clear();
    c->prof_on = on != 0;
    return 0;
}
int bd_prof_count(bd_ctx* c) { return (int)c->prof.size(); }
/* after a stream sync: name (<=63 chars), elapsed ms and algorithmic weight bytes of the i-th profiled GEMM launch */
int bd_prof_get(bd_ctx* c, int i, char* name, float* ms, double* bytes) {
    if (i < 0 || i >= (int)c->prof.size()) return fail("bd_prof_get: index");
    const auto& r = c->prof[i];
    std::strncpy(name, r.name.c_str(), 63); name[63] = 0;
    if (hipEventElapsedTime(ms, r.e0, r.e1) != hipSuccess) return fail("hipEventElapsedTime failed (sync the stream first)");
    *bytes = r.bytes;
    return 0;
}
/* the (split-K, nwaves) launch config the engine chose for a named GEMM, e.g. "head.wo" */
int bd_gemm_config(bd_ctx* c, const char* name, int* splitk, int* nwaves) {
    auto it = c->g.find(name);
    if (it == c->g.end()) return fail(std::string("bd_gemm_config: unknown GEMM '") + name + "'");
    *splitk = it->second.S; *nwaves = it->second.code();   /* waves + 16 * ring depth */
    return 0;
}

int bd_graph_capture(bd_ctx* c, int phase, void* s) {
    BD_GUARD(
        if (phase < 0 || phase > 1) return fail("bd_graph_capture: phase must be 0 or 1");
        hipStream_t st = (hipStream_t)s;
        if (c->prof_on) return fail("bd_graph_capture: disable profiling first");
        if (c->gexec[phase]) { hipGraphExecDestroy(c->gexec[phase]); c->gexec[phase] = nullptr; }
        if (c->graph[phase]) { hipGraphDestroy(c->graph[phase]); c->graph[phase] = nullptr; }
        if (hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed) != hipSuccess) return fail("hipStreamBeginCapture failed");
        int r = (phase == 0) ? head_sample(c, st) : (projector(c, st) || llm_step(c, st));
        hipGraph_t g = nullptr;
        hipError_t e = hipStreamEndCapture(st, &g);
        if (r != 0) { if (g) hipGraphDestroy(g); return -1; }
        if (e != hipSuccess || !g) return fail(std::string("hipStreamEndCapture failed: ") + hipGetErrorString(e));
        e = hipGraphInstantiate(&c->gexec[phase], g, nullptr, nullptr, 0);
        if (e != hipSuccess) { hipGraphDestroy(g); return fail(std::string("hipGraphInstantiate failed: ") + hipGetErrorString(e)); }
        c->graph[phase] = g;
        return 0;)
}
int bd_graph_launch(bd_ctx* c, int phase, void* s) {
    BD_GUARD(
        if (phase < 0 || phase > 1 || !c->gexec[phase]) return fail("bd_graph_launch: phase not captured");
        hipError_t e = hipGraphLaunch(c->gexec[phase], (hipStream_t)s);
        return e == hipSuccess ? 0 : fail(std::string("hipGraphLaunch failed: ") + hipGetErrorString(e));)
}

}  // extern "C"
