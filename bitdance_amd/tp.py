"""Tensor parallelism of the BitDance step over the xGMI links of one node.

The reference's multi-GPU mode is replicas (eval/eval_dpg.py:25-29); splitting ONE image's generation over 2/4/8 GPUs is
this framework's own mode, Megatron-style:

  * column-split (by output features / attention heads): head ``wqkv`` and ``w1`` (the SwiGLU pair (h1_f, h2_f) stays on one
    rank), LLM ``q/k/v`` (by head; a rank's q heads use exactly its kv heads because nkv % tp == 0) and ``gate/up``;
  * row-split (by input features): head ``wo`` / ``w2``, LLM ``o_proj`` / ``down_proj``; their fp32 partials are summed
    across ranks by ONE hand-written exchange kernel per Linear (csrc/bd_comm.hip: push-only two-shot over IPC-mapped peer
    buffers, bias and the single bf16 rounding applied by the reducing rank, result replicated bit-identically), captured
    in the AR-step hipGraphs;
  * replicated: residual streams, LayerNorm / RMSNorm, the adaLN projection (its output feeds every rank's LayerNorm rows;
    it depends only on (t_i, cond) and is the natural candidate for the side stream), projector, sampler, ``sign``, RNG
    (same seed => same Philox draws on every rank), the conv decoder;
  * the KV cache is sharded by kv head.

``TPComm`` wraps one rank's ``bd_comm``.  Bootstrap (handle exchange, barriers, the once-per-image prefill's all-reduce)
goes through ``torch.distributed`` (RCCL on GPUs, gloo in the single-GPU two-process test); ``BD_TP_COMM=rccl`` also routes
the per-Linear exchange through ``ncclAllReduce`` of the same fp32 partials (fallback and baseline for the kernel).

The ``shard_*`` functions are pure tensor slicing (CPU-testable): ``sum_r rank_r(x) == unsharded(x)`` is checked in
tests/test_tp_cpu.py, including the q/k/v thirds and the SwiGLU pairs.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from ._lib import BitDanceHipError, check, lib

__all__ = ["TPComm", "ada_gather_bytes", "seq_hbuf_bytes", "shard_head_state", "shard_llm_state", "shard_rows", "shard_cols"]


# ----------------------------------------------------------------------------------------------- slicing
def shard_rows(w: torch.Tensor, rank: int, size: int, groups: int = 1) -> torch.Tensor:
    """Rows [rank/size) of each of ``groups`` equal row blocks, concatenated (q|k|v thirds, h1|h2 halves)."""
    n = w.shape[0] // groups
    if w.shape[0] % groups or n % size:
        raise BitDanceHipError(f"cannot split {tuple(w.shape)} rows in {groups} groups over {size} ranks")
    m = n // size
    return torch.cat([w[g * n + rank * m: g * n + (rank + 1) * m] for g in range(groups)], dim=0).contiguous()


def shard_cols(w: torch.Tensor, rank: int, size: int) -> torch.Tensor:
    k = w.shape[1]
    if k % size:
        raise BitDanceHipError(f"cannot split {tuple(w.shape)} columns over {size} ranks")
    m = k // size
    return w[:, rank * m:(rank + 1) * m].contiguous()


def shard_head_state(sd: dict, rank: int, size: int, head_dim: int = 128) -> dict:
    """vision_head.safetensors -> this rank's slices (same key names).  wqkv: q,k,v contiguous thirds
    (flow_head_parallel_x.py:197), split by attention head; w1: h1,h2 halves (:250); wo / w2: input columns; biases of the
    row-split Linears stay whole (added once, by the exchange)."""
    if size == 1:
        return dict(sd)
    out = {}
    D = sd["net.input_proj.weight"].shape[0]
    if (D // head_dim) % size:
        raise BitDanceHipError(f"{D // head_dim} attention heads do not divide over {size} ranks")
    for k, v in sd.items():
        if ".attn.wqkv." in k:
            out[k] = shard_rows(v, rank, size, groups=3)
        elif k.endswith(".w1.weight") or k.endswith(".w1.bias"):
            out[k] = shard_rows(v, rank, size, groups=2)
        elif k.endswith(".attn.wo.weight") or k.endswith(".w2.weight"):
            out[k] = shard_cols(v, rank, size)
        else:
            out[k] = v
    return out


def shard_llm_state(sd: dict, cfg: dict, rank: int, size: int) -> dict:
    """HF Qwen3 checkpoint -> this rank's slices: q/k/v rows by head, o_proj columns by q head, gate/up rows, down columns."""
    if size == 1:
        return dict(sd)
    nh, nkv = cfg["num_attention_heads"], cfg["num_key_value_heads"]
    if nh % size or nkv % size:
        raise BitDanceHipError(f"{nh} q heads / {nkv} kv heads do not divide over {size} ranks")
    out = {}
    for k, v in sd.items():
        if k.endswith(("q_proj.weight", "k_proj.weight", "v_proj.weight", "gate_proj.weight", "up_proj.weight")):
            out[k] = shard_rows(v, rank, size)
        elif k.endswith(("o_proj.weight", "down_proj.weight")):
            out[k] = shard_cols(v, rank, size)
        elif k == "lm_head.weight":
            continue
        else:
            out[k] = v
    return out


def ada_gather_bytes(rows: int, ada_cols: int, group: int | None = None) -> int:
    """Capacity a communicator's gather region needs for the column-split adaLN projection: one GROUP of evaluations' modulation
    tensors, [G * padded rows][ada_cols] bf16, with the engine's default grouping (512 rows per projection GEMM up to 128 rows per
    evaluation, 1024 up to 512: csrc/bd_api.hip ``tune.ada_group``).  rows = branches * num_images * parallel_num."""
    mp = 32 if rows <= 32 else (64 if rows <= 64 else (rows + 127) // 128 * 128)
    g = group or (512 // mp if mp <= 128 else (1024 // mp if (mp <= 512 and 1024 % mp == 0) else 1))
    # two slots: the gathered tensor is double-buffered by group parity (a peer may push group g + 1 while this rank still reads g)
    return 2 * g * mp * ada_cols * 2 if g >= 2 else 0


def seq_hbuf_bytes(rows: int, width: int) -> int:
    """Capacity of the operand landing buffer of the sequence-parallel exchange (csrc/bd_sp.hip): the bf16 operand rows every rank
    pushes to every rank, [padded rows][width], followed by the landing area of the one fp32 hand-off of the path, the Qwen3 step's
    final-norm rows (rms_sp_kernel -> sp_final_rows_kernel: hidden state and the next patch's condition are assembled on every rank).
    rows = branches * num_images * parallel_num; 0 where that form does not apply (it is built for 128-row passes: one image,
    64-token patches)."""
    return rows * width * 6 if rows == 128 else 0        # bf16 operand region | fp32 final-rows region (never overlaid: a rank may enter the
                                                         # next phase and push operand rows while a peer still reads the final rows)


# ----------------------------------------------------------------------------------------------- communicator
class _NcclUniqueId(C.Structure):
    _fields_ = [("internal", C.c_ubyte * 128)]          # c_ubyte: a c_char field would be read back truncated at the first NUL


class TPComm:
    """One rank's exchange state.  ``max_elems`` = rows * N of the largest exchanged tensor (e.g. 512 * 5120)."""

    def __init__(self, rank: int, size: int, max_elems: int, device=None, gather_bytes: int = 0, hbuf_bytes: int = 0):
        """``gather_bytes``: capacity of the all-gather region (the column-split adaLN projection's modulation tensor of one group of
        evaluations, two slots: 2 x G x rows x 71 680 x 2 B = 147 MB at one image); 0: none, the projection stays replicated.
        ``hbuf_bytes``: the operand landing buffer of the sequence-parallel row kernels (``seq_hbuf_bytes``); 0: the all-reduce form only."""
        self.l = lib()
        self.rank, self.size = rank, size
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.in_process_peers = False
        self.shares_gpu = False                          # some ranks of the group sit on the same physical GPU (set by the constructors)
        self.loopback = False
        # the sequence-parallel hand-off's own self-test (cacheable landing buffer written by the peers): None = not run (ranks inside
        # one process / loop-back: one L2 domain, covered by the GPU tests), True / False = from_process_group's verdict.  Engine turns
        # the sequence-parallel row kernels on by default only when this is not False -- and, across devices, only when it is True.
        self.sp_ok = None
        with torch.cuda.device(self.device):
            self.h = self.l.bd_comm_create3(rank, size, int(max_elems), int(gather_bytes), int(hbuf_bytes))
        if not self.h:
            raise BitDanceHipError(f"bd_comm_create failed: {self.l.bd_last_error().decode()}")
        self.group = None
        self.backend = "ipc"
        self.fences = 0
        self._nccl = None

    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.l.bd_comm_destroy(self.h)
                self.h = None
        except Exception:
            pass

    # -- construction ------------------------------------------------------------------------------------
    @classmethod
    def from_process_group(cls, max_elems: int, group=None, device=None, backend: str | None = None, gather_bytes: int = 0,
                           hbuf_bytes: int = 0) -> "TPComm":
        """One process per GPU: exchange the IPC handles through ``torch.distributed`` and map every peer."""
        import torch.distributed as dist
        rank, size = dist.get_rank(group), dist.get_world_size(group)
        self = cls(rank, size, max_elems, device, gather_bytes, hbuf_bytes)
        self.group = group
        backend = backend or os.environ.get("BD_TP_COMM", "ipc")
        if backend not in ("ipc", "rccl"):
            raise BitDanceHipError(f"exchange backend must be 'ipc' or 'rccl', not {backend!r}")
        self.fallback_reason = None
        if size > 1:
            # map the peers' buffers; if ANY rank cannot (IPC export / open refused on this node), every rank falls back to RCCL
            ok, why = True, ""
            buf = C.create_string_buffer(192)                # data, flags, operand landing buffer (zeros when there is none)
            if self.l.bd_comm_ipc_handles3(self.h, buf) != 0:
                ok, why = False, self.l.bd_last_error().decode()
            handles = [None] * size
            my_dev = torch.device(self.device).index if self.device is not None else None
            if my_dev is None:
                my_dev = torch.cuda.current_device()
            dist.all_gather_object(handles, (bytes(buf.raw), ok, why, my_dev, _device_uuid(self.device)), group=group)
            ok = all(h[1] for h in handles)
            if ok and torch.cuda.device_count() >= size:
                # ask the runtime before touching a peer's memory: a mapping that "opens" but faults on the first remote store
                # takes the process down, which no fall-back can catch (ranks that see only their own device cannot ask: skipped)
                for p in range(size):
                    if p != rank and handles[p][4] != handles[rank][4] and not torch.cuda.can_device_access_peer(my_dev, handles[p][3]):
                        ok, why = False, f"device {my_dev} cannot access device {handles[p][3]} (no peer-to-peer path)"
            if ok:
                for p in range(size):
                    if p != rank and self.l.bd_comm_open_peer3(self.h, p, C.create_string_buffer(handles[p][0], 192)) != 0:
                        ok, why = False, self.l.bd_last_error().decode()
            flags = [None] * size
            dist.all_gather_object(flags, (ok, why), group=group)
            if not all(f[0] for f in flags):
                self.fallback_reason = f"IPC mapping failed ({[f[1] for f in flags if not f[0]][:1]})"
                backend = "rccl"
            # the hand-written exchange reads memory the peers write over xGMI inside one kernel: that is only coherent on
            # uncached (fine-grained) allocations.  If ANY rank got plain device memory and the ranks sit on different devices,
            # the exchange goes through RCCL (ranks sharing one device -- the single-GPU functional run -- share one L2).
            infos = [None] * size
            dist.all_gather_object(infos, (self.info(), _device_uuid(self.device)), group=group)
            cross_device = len({i[1] for i in infos}) > 1
            # several ranks on ONE physical GPU (functional runs on a single-GPU box): a kernel that polls in every workgroup (the
            # sequence-parallel form's GEMM-side wait) could fill the chip and starve the very rank it waits for -- Engine then
            # puts the wait into a one-workgroup kernel in front of the GEMM ("tune.sp_wait" = 0)
            self.shares_gpu = len({i[1] for i in infos}) < size
            if backend == "ipc" and cross_device and not all(i[0]["data_uncached"] and i[0]["flags_uncached"] for i in infos):
                self.fallback_reason = "exchange buffers are not uncached (fine-grained) on every rank"
                backend = "rccl"
            # one-time cross-rank self-test of the hand-written exchange (known pattern, every rank checks every element):
            # a node on which the IPC mapping 'works' but remote writes are not seen must not find out 44 000 exchanges later
            if backend == "ipc":
                # first without system-scope fences around the flags (write-through payload, drained before the flag, read with
                # system-scope loads: no fence needed by the hand-off recipe), then -- every rank together -- with them, then RCCL
                for fences in (0, 1):
                    self.set_fences(fences)
                    ok = self._self_test()
                    oks = [None] * size
                    dist.all_gather_object(oks, ok, group=group)
                    if all(oks):
                        break
                    dist.barrier(group=group)
                    check(self.l.bd_comm_reset(self.h), "bd_comm_reset")
                    dist.barrier(group=group)
                if not all(oks):
                    self.fallback_reason = f"exchange self-test failed on ranks {[r for r, o in enumerate(oks) if not o]}"
                    backend = "rccl"
            if backend == "ipc" and self.hbuf_bytes > 0:
                # ... and of the sequence-parallel hand-off (its landing buffer is ordinary cacheable memory the peers write): six rounds of
                # changing patterns, every rank checks every 16 B unit behind the GEMM prologue's wait.  A failure keeps the all-reduce
                # form (Engine reads sp_ok); it does not touch the exchange backend
                ok = not self.fences and self._sp_self_test(lambda: dist.barrier(group=group))
                oks = [None] * size
                dist.all_gather_object(oks, bool(ok), group=group)
                self.sp_ok = all(oks)
                if not self.sp_ok:
                    dist.barrier(group=group)
                    check(self.l.bd_comm_reset(self.h), "bd_comm_reset")
                    dist.barrier(group=group)
                    if rank == 0:
                        print("[bitdance_amd.tp] the sequence-parallel hand-off's self-test failed on ranks "
                              f"{[r for r, o in enumerate(oks) if not o]}: the head keeps the all-reduce form", flush=True)
            if self.fallback_reason and rank == 0:
                print(f"[bitdance_amd.tp] {self.fallback_reason}: exchanges go through RCCL (ncclAllReduce)", flush=True)
            if backend == "rccl":
                self._init_rccl(dist, group)
            dist.barrier(group=group)
        self.backend = backend if size > 1 else "none"
        return self

    def info(self) -> dict:
        """Allocation kinds / mode of this rank's exchange state (bd_comm_info)."""
        out = (C.c_longlong * 4)()
        check(self.l.bd_comm_info(self.h, out), "bd_comm_info")
        return {"data_uncached": bool(out[0]), "flags_uncached": bool(out[1]), "mode": int(out[2]), "capacity": int(out[3])}

    def _self_test(self, shapes=((32, 256), (128, 1024), (64, 5120), (32, 256), (128, 5120), (8, 64))) -> bool:
        """Exchanges of known patterns over the sizes the step uses, back to back on re-used buffers (an ordering race is statistical: one
        small exchange would not show it): part_r[i] = (r + 1) * v[i] with v exactly representable, so the reduced bf16 result
        must equal v * size (size + 1) / 2 bit for bit on every rank -- and, where the communicator has a gather region, the push
        all-gather of per-rank column slices.  False on a timeout or any wrong element.  (The forms this cannot reach -- the push fused
        into the GEMM epilogues, the sequence-parallel row kernels -- are compared with the conservative form on the first warm-up
        image by bench.py, tokens bit for bit on every rank.)"""
        ok = True
        with torch.cuda.device(self.device):
            try:
                self.set_timeout(5.0)
                for rep, (rows, N) in enumerate(shapes):
                    if rows * N > int(self.info()["capacity"]) - 128 * 8:
                        continue
                    v = ((torch.arange(rows * N, device=self.device) % 61) - 30 + rep).float().view(rows, N) / 4.0
                    part = (v * (self.rank + 1)).contiguous()
                    got = self.allreduce(part)
                    torch.cuda.current_stream().synchronize()
                    ok = ok and self.l.bd_comm_error(self.h) == 0 and torch.equal(got.float(), v * (self.size * (self.size + 1) / 2))
                if ok and self.gather_bytes > 0:
                    for rep, (rows, nl) in enumerate(((64, 256), (128, 1024), (64, 256))):
                        if rows * nl * self.size * 2 > self.gather_bytes:
                            continue
                        cols = [(((torch.arange(rows * nl, device=self.device) % 53) - 26 + 3 * r + rep).float() / 8.0).view(rows, nl).to(torch.bfloat16)
                                for r in range(self.size)]
                        got = self.allgather(cols[self.rank].contiguous())
                        torch.cuda.current_stream().synchronize()
                        ok = ok and self.l.bd_comm_error(self.h) == 0 and torch.equal(got, torch.cat(cols, dim=1))
            except BitDanceHipError:
                ok = False
            finally:
                self.set_timeout(20.0)
        return bool(ok)

    def sp_selftest_round(self, rnd: int, rows: int = 128, D: int | None = None, bad: torch.Tensor | None = None) -> torch.Tensor:
        """One round of the sequence-parallel hand-off's self-test on the current stream (bd_comm_sp_selftest): returns the device int
        that accumulates mismatching 16 B units.  Every rank runs the same rounds, a barrier of the ranks between two rounds."""
        if D is None:
            D = min(5120, (self.hbuf_bytes // (rows * 2)) // 8 * 8)
        if bad is None:
            bad = torch.zeros(1, dtype=torch.int32, device=self.device)
        check(self.l.bd_comm_sp_selftest(self.h, int(rnd), int(rows), int(D), 0 if self.shares_gpu else 1, bad.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "bd_comm_sp_selftest")
        return bad

    def _sp_self_test(self, barrier, rounds: int = 6) -> bool:
        """False on a timeout or any wrong unit in any round.  ``barrier``: all ranks have finished the round (the next round's pushes
        overwrite the buffers the slowest rank may still be checking)."""
        ok = True
        with torch.cuda.device(self.device):
            try:
                self.set_timeout(5.0)
                bad = torch.zeros(1, dtype=torch.int32, device=self.device)
                for rnd in range(rounds):
                    self.sp_selftest_round(rnd, bad=bad)
                    torch.cuda.current_stream().synchronize()
                    ok = ok and self.l.bd_comm_error(self.h) == 0 and int(bad.item()) == 0
                    barrier()
            except BitDanceHipError:
                ok = False
            finally:
                self.set_timeout(20.0)
        return bool(ok)

    def use_rccl(self) -> None:
        """Switch the per-Linear exchange to ncclAllReduce (every rank must call it; engines / graphs built before are stale)."""
        import torch.distributed as dist
        if self._nccl is None:
            self._init_rccl(dist, self.group)
        else:
            check(self.l.bd_comm_set_rccl(self.h, self._nccl[1], C.cast(self._nccl[0].ncclAllReduce, C.c_void_p).value), "bd_comm_set_rccl")
        self.backend = "rccl"

    def reset(self) -> None:
        """After a failed exchange: every rank, between two barriers."""
        self.barrier()
        check(self.l.bd_comm_reset(self.h), "bd_comm_reset")
        self.barrier()

    @classmethod
    def in_process(cls, size: int, max_elems: int, device=None, gather_bytes: int = 0, hbuf_bytes: int = 0) -> list:
        """``size`` ranks as contexts of THIS process on one device, linked by plain pointers: the exchange protocol can then
        be exercised on a single GPU with one stream per rank (tests/test_gpu_tp.py)."""
        comms = [cls(r, size, max_elems, device, gather_bytes, hbuf_bytes) for r in range(size)]
        for a in comms:
            a.in_process_peers = True
            a.shares_gpu = True
            for b in comms:
                if a is not b:
                    check(a.l.bd_comm_set_peer_ptrs3(a.h, b.rank, a.l.bd_comm_local_data(b.h), a.l.bd_comm_local_flags(b.h),
                                                     a.l.bd_comm_local_hbuf(b.h)))
        return comms

    @classmethod
    def loopback_rank(cls, rank: int, size: int, max_elems: int, device=None, gather_bytes: int = 0, hbuf_bytes: int = 0) -> "TPComm":
        """ONE rank of a ``size``-rank group alone on this GPU (bd_comm_set_loopback): the peers' buffers are scratch copies and every
        flag a peer would write is written locally, so the rank's launches, weight shards, pushes and waits run as on a node minus
        the links -- for timing its critical path on one GPU (tools/head_sweep.py --tp-shard).  Results are meaningless."""
        self = cls(rank, size, max_elems, device, gather_bytes, hbuf_bytes)
        with torch.cuda.device(self.device):
            check(self.l.bd_comm_set_loopback(self.h), "bd_comm_set_loopback")
        self.loopback = True
        return self

    @property
    def hbuf_bytes(self) -> int:
        return int(self.l.bd_comm_hbuf_bytes(self.h))

    def prepushed(self) -> int:
        """Exchanges whose reduce-scatter push ran in the producing GEMM's epilogue."""
        return int(self.l.bd_comm_prepushed(self.h))

    def _init_rccl(self, dist, group) -> None:
        """ncclCommInitRank on the librccl torch itself uses; the per-Linear exchange then is ncclAllReduce(fp32 partials)."""
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        n = C.CDLL(path)
        uid = _NcclUniqueId()
        if self.rank == 0:
            n.ncclGetUniqueId.argtypes = [C.POINTER(_NcclUniqueId)]
            rc = n.ncclGetUniqueId(C.byref(uid))
            if rc != 0:
                raise BitDanceHipError(f"ncclGetUniqueId failed ({rc})")
        box = [C.string_at(C.byref(uid), 128)] if self.rank == 0 else [None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        C.memmove(C.byref(uid), box[0], 128)
        comm = C.c_void_p()
        n.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, _NcclUniqueId, C.c_int]
        with torch.cuda.device(self.device):
            rc = n.ncclCommInitRank(C.byref(comm), self.size, uid, self.rank)
            if rc != 0:
                raise BitDanceHipError(f"ncclCommInitRank failed ({rc})")
        fn = C.cast(n.ncclAllReduce, C.c_void_p).value
        check(self.l.bd_comm_set_rccl(self.h, comm, fn), "bd_comm_set_rccl")
        self._nccl = (n, comm)

    # -- host-side collectives for the once-per-image prefill -------------------------------------------------
    def all_reduce_(self, t: torch.Tensor) -> torch.Tensor:
        """In-place sum over the ranks on the current stream (RCCL); through the host when the group is gloo."""
        if self.size > 1:
            import torch.distributed as dist
            if dist.get_backend(self.group) == "gloo" and t.is_cuda:
                h = t.float().cpu()
                dist.all_reduce(h, group=self.group)
                t.copy_(h.to(t.dtype))
            else:
                dist.all_reduce(t, group=self.group)
        return t

    def barrier(self) -> None:
        if self.size > 1 and _dist_ready():
            import torch.distributed as dist
            dist.barrier(group=self.group)

    # -- status -------------------------------------------------------------------------------------------------
    def check(self) -> None:
        """After a stream sync: raise if any in-kernel wait of this rank ran out of its budget, or a peer reported that one of
        its waits did (the error word travels with the exchange, so every rank raises at the same check)."""
        e = self.l.bd_comm_error(self.h)
        if e != 0:
            peers = [p for p in range(self.size) if e & (1 << p)] if e > 0 else "?"
            gave_up = [p for p in range(self.size) if e > 0 and e & (1 << (8 + p))]
            raise BitDanceHipError(f"tensor-parallel exchange failed on rank {self.rank}: timed out waiting for peers {peers}"
                                   + (f"; ranks {gave_up} reported a timeout" if gave_up else ""))

    @property
    def gather_bytes(self) -> int:
        return int(self.l.bd_comm_gather_bytes(self.h))

    @property
    def gather_ptr(self) -> int:
        return int(self.l.bd_comm_gather_ptr(self.h) or 0)

    def allgather(self, slice_bf16: torch.Tensor) -> torch.Tensor:
        """Standalone all-gather of column slices (tests): this rank's [rows, Nl] bf16 -> a copy of the gathered [rows, Nl * size]."""
        rows, nl = slice_bf16.shape
        st = torch.cuda.current_stream().cuda_stream
        check(self.l.bd_comm_allgather(self.h, slice_bf16.contiguous().data_ptr(), rows, nl, st), "bd_comm_allgather")
        out = torch.empty(rows, nl * self.size, dtype=torch.bfloat16, device=slice_bf16.device)
        check(self.l.bd_comm_copy_out(self.h, out.data_ptr(), out.numel() * 2, 2, st), "bd_comm_copy_out")
        return out

    def exchanges(self) -> int:
        return int(self.l.bd_comm_exchanges(self.h))

    def set_fences(self, on: int) -> None:
        """System-scope fences around every flag of the hand-written exchange (default off; from_process_group turns them on when the
        self-test only passes with them)."""
        check(self.l.bd_comm_set_fences(self.h, int(on)))
        self.fences = int(on)

    def set_timeout(self, seconds: float) -> None:
        check(self.l.bd_comm_set_timeout(self.h, float(seconds)))

    def allreduce(self, part: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
        """Standalone exchange of one fp32 partial [rows, N] (tests / micro-benchmarks) on the current stream: returns this
        rank's copy of bf16(sum over ranks + bias)."""
        rows, N = part.shape
        out, is32 = C.c_void_p(), C.c_int()
        st = torch.cuda.current_stream().cuda_stream
        check(self.l.bd_comm_allreduce(self.h, part.data_ptr(), bias.data_ptr() if bias is not None else None, rows, N,
                                       C.byref(out), C.byref(is32), st), "bd_comm_allreduce")
        if is32.value:                                   # RCCL mode: fp32 sums; bias + rounding are the consumer's
            t = torch.empty(rows, N, dtype=torch.float32, device=part.device)
            check(self.l.bd_comm_copy_out(self.h, t.data_ptr(), t.numel() * 4, 0, st), "bd_comm_copy_out")
            if bias is not None:
                t = t + bias.float()
            return t.to(torch.bfloat16)
        t = torch.empty(rows, N, dtype=torch.bfloat16, device=part.device)
        check(self.l.bd_comm_copy_out(self.h, t.data_ptr(), t.numel() * 2, 1, st), "bd_comm_copy_out")
        return t


def _device_uuid(device) -> str:
    """Identifies the physical GPU behind a torch device (ranks that share one GPU share its L2).  Without a physical identifier
    (no ``uuid`` and no PCI address in this torch build) the answer is unique per process: ranks isolated with HIP_VISIBLE_DEVICES
    all see "device 0", so an index-based fallback would call eight GPUs one device and skip the peer-access and uncached-buffer
    checks -- unknown identity must read as DIFFERENT devices."""
    props = torch.cuda.get_device_properties(device)
    uuid = getattr(props, "uuid", None)
    if uuid is not None:
        return str(uuid)
    pci = [getattr(props, k, None) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")]
    if all(v is not None for v in pci):
        return "pci-%04x:%02x:%02x" % tuple(int(v) for v in pci)
    import os
    import socket
    return f"unknown-{socket.gethostname()}-{os.getpid()}"


def _dist_ready() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()
