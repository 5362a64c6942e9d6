"""ProcessGroup mirror: the TP collectives the hot path issues.

Reference: xllm::ProcessGroup over c10d::ProcessGroupNCCL (framework/parallel_state/process_group.h:40-125,
cuda_process_group.h:24-53 -- on ROCm torch "nccl" IS RCCL) and the helpers parallel_state::reduce / gather
(parallel_state.cpp:89-102, 183-192). One process per GPU; collectives run over xGMI.
`world_size == 1` is a no-op exactly as in the reference (parallel_state.cpp:184-186).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist


class ProcessGroup:
    def __init__(self, group: Optional[dist.ProcessGroup] = None, rank: int = 0, world_size: int = 1):
        self.group, self._rank, self._world = group, rank, world_size
        self.oneshot: Optional["OneShotAllReduce"] = None
        self.oneshot_note = "not requested"

    def enable_oneshot(self, device, max_bytes: int = 8 << 20, self_test: bool = True) -> Optional["OneShotAllReduce"]:
        """opt in to the one-shot xGMI all-reduce (csrc/allreduce.hip) for CUDA messages of at most max_bytes; larger
        messages, other dtypes and CPU tensors keep going through RCCL / gloo. Collective: every rank of the group calls it.
        The set-up ends with a SELF-TEST (peer mapping + one checked message per size class, also through the fused
        add + norm form); the verdict is agreed over the group, so either every rank uses the kernel or none does
        (`oneshot_note` says why not) -- the group's own all-reduce stays the fallback."""
        if self._world <= 1 or self.oneshot is not None:
            return self.oneshot
        # OneShotAllReduce() runs the same collectives on every rank whatever fails locally (its `refused` is already agreed over
        # the group); the self-test verdict is agreed here; close() -- a barrier + unmap -- is then run by EVERY rank or by none.
        ar = OneShotAllReduce(self, device, max_bytes)
        note = "ok"
        if ar.refused:
            note = ar.refused                      # identical on every rank: nobody enters the self-test
        elif self_test and not ar.self_test():     # (catches its own exceptions: returns False)
            note = "self-test failed (a checked message came back wrong or a wait timed out)"
        verdicts = [None] * self._world
        dist.all_gather_object(verdicts, note, group=self.group)
        bad = [f"rank {r}: {v}" for r, v in enumerate(verdicts) if v != "ok"]
        if bad:
            ar.close()                             # every rank takes this branch together
            self.oneshot, self.oneshot_note = None, "; ".join(sorted(set(bad)))
        else:
            self.oneshot, self.oneshot_note = ar, "ok"
        return self.oneshot

    def allreduce_kind(self) -> str:
        """what a small CUDA all-reduce of this group runs on (bench.py's config.allreduce)"""
        if self._world <= 1:
            return "none"
        if self.oneshot is not None:
            return "oneshot-xgmi"
        try:
            be = dist.get_backend(self.group)
        except Exception:   # noqa: BLE001
            be = "unknown"
        return "rccl" if be == "nccl" else str(be)

    def check(self) -> None:
        """host-side look at the one-shot kernel's status word; call it at a point where the step synchronises anyway (after the
        sampled tokens' D2H copy, after a timed region). Raises if a launch gave up waiting for a peer: its result was undefined."""
        if self.oneshot is not None:
            self.oneshot.check()

    def check_agreed(self) -> None:
        """collective form of check(): every rank of the group calls it, the status words are exchanged, and EVERY rank raises if
        any rank's launch timed out -- so a fall-back that re-times on RCCL is taken by all ranks together (a rank that was waited
        for sees no timeout of its own; round-3 advisor finding on bench.py)."""
        if self._world <= 1 or self.oneshot is None:
            return
        mine = 0 if int(self.oneshot.status.item()) == 0 else 1
        flags = [None] * self._world
        dist.all_gather_object(flags, mine, group=self.group)
        if any(flags):
            from . import _lib
            bad = [r for r, f in enumerate(flags) if f]
            raise _lib.Mi355Error(f"one-shot all-reduce: a wait for a peer's flag timed out on rank(s) {bad} (result undefined)")

    def rank(self) -> int:
        return self._rank

    def world_size(self) -> int:
        return self._world

    def allreduce(self, x: torch.Tensor) -> None:
        if self.oneshot is not None and self.oneshot.takes(x):
            self.oneshot.allreduce(x)     # a plain kernel: also inside a graph capture, no eager piece needed
            return
        _run_collective(lambda: dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group))

    def allreduce_add_rms_norm(self, partial: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, eps: float,
                               quantize: bool):
        """the tensor-parallel half-layer tail in ONE kernel when the one-shot path is on: SUM all-reduce of `partial` [M, H] ->
        residual <- r16(sum + residual) -> RMSNorm (-> per-token int8 quant). Returns (q int8, scale) or the 16-bit norm; None when
        the one-shot kernel does not take the message (the caller then runs allreduce + the row-wise operator)."""
        if self.oneshot is None or partial.dim() != 2 or not self.oneshot.takes(partial):
            return None
        if partial.size(1) % 8 != 0 or partial.size(1) > 16384:     # the fused kernel's envelope (H % 8, H <= 16384): not applicable
            return None
        return self.oneshot.allreduce_add_rms_norm(partial, residual, weight, eps, quantize)

    def matmul_allreduce_add_rms_norm(self, a_q, a_scale, w_packed, w_scale, bias, residual, weight, eps: float, quantize: bool):
        """row-parallel W8A8 linear -> SUM all-reduce -> residual add -> RMSNorm (-> int8 quant) as GEMM + ONE kernel: the one-shot
        kernel reads the GEMM's int32 K-slice sums directly (OneShotAllReduce.matmul_allreduce_add_rms_norm). None when the
        one-shot path is off or the shape is outside it (the caller runs linear + allreduce_add_rms_norm)."""
        if self.oneshot is None or w_packed is None or not self.oneshot._stream_ok():
            return None
        return self.oneshot.matmul_allreduce_add_rms_norm(a_q, a_scale, w_packed, w_scale, bias, residual, weight, eps, quantize)

    def allreduce_async(self, x: torch.Tensor):
        """ProcessGroup::allreduce_async (process_group.cpp:103-108): returns the c10d Work. With RCCL the collective runs on
        the process group's own HIP stream, fenced against the CURRENT stream by events on both sides (enqueue: the RCCL
        stream waits for what the current stream has queued; Work.wait(): the current stream waits for the collective), so
        every kernel launched between allreduce_async and wait() overlaps the transfer over xGMI."""
        return dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def allgather(self, x: torch.Tensor) -> torch.Tensor:
        out = torch.empty((self._world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        xc = x.contiguous()
        parts = list(out.unbind(0))
        # list form: supported by both RCCL and gloo (the CPU tests); the views alias `out`, no extra copy
        _run_collective(lambda: dist.all_gather(parts, xc, group=self.group))
        return out

    def alltoall(self, send: torch.Tensor, send_counts, recv_counts) -> torch.Tensor:
        """variable-size all-to-all over dim 0 (RCCL all-to-all over xGMI; gloo on the CPU tests): rows
        [sum(send_counts[:r]), +send_counts[r]) of `send` go to rank r; returns the received rows, rank-major"""
        if self._world == 1:
            return send
        out = send.new_empty((int(sum(recv_counts)),) + tuple(send.shape[1:]))
        sc = send.contiguous()
        _run_collective(lambda: dist.all_to_all_single(out, sc, list(recv_counts), list(send_counts), group=self.group))
        return out


class OneShotAllReduce:
    """One-shot SUM all-reduce over peer-mapped buffers (include/xllm_mi355.h, csrc/allreduce.hip): every rank copies its
    message into its own shared slot, raises a flag in every peer's buffer and sums all slots in rank order (fp32, one
    rounding), so the result is bit-identical on every rank and costs ONE xGMI hop instead of the ring's 2 (W - 1).
    The handles travel over the group's object all-gather (host side, once). Opt-in: ProcessGroup.enable_oneshot.

    Rules that keep the ranks in lock step (round-2 review):
      * `takes()` looks only at properties every rank shares (dtype, element count, the size limit, being on the GPU) and at the
        stream: a tensor that is not contiguous / 16-byte aligned on SOME rank is staged through an aligned copy instead of
        sending that rank alone down the RCCL path;
      * ONE stream: the epoch counter and the two data slots are per (rank, buffer), so two launches may never be in flight on
        two streams at once. The first launch binds the object to its stream; launches from any other stream decline
        (`takes()` is False there: RCCL) -- a graph capture counts as the stream it captures on;
      * plain hipMalloc memory (kind 2) is only accepted when every rank sits on the SAME device (the two-processes-one-GPU
        protocol test): across devices coarse-grained memory is not guaranteed visible to a peer while the kernel runs;
      * the status word is checked where the step synchronises anyway (ProcessGroup.check / DecodeEngine.step / bench.py)."""

    _DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}

    def __init__(self, pg: "ProcessGroup", device, max_bytes: int = 8 << 20, timeout_s: float = 2.0):
        import ctypes as C
        from . import _lib
        self._C, self._lib = C, _lib
        l = _lib.lib()
        self.pg, self.max_bytes, self.timeout_s = pg, int(max_bytes), float(timeout_s)
        self.device = torch.device(device)
        self.refused = None
        self._stream = None
        self._suspended = 0          # > 0 inside suspended(): every launch declines the kernel
        self.grid_limit = 0          # 0 = the library's 64 blocks; raised below when every rank has a GPU of its own
        world, rank = pg.world_size(), pg.rank()
        total = l.xllm_mi355_oneshot_allreduce_buffer_bytes(self.max_bytes)
        # Every collective below (object all-gather, barrier) runs UNCONDITIONALLY on every rank: a rank-local failure (no
        # exportable memory, a peer handle that does not open) is recorded in self.refused and travels with the next exchange,
        # it never makes this rank skip a collective its peers sit in (round-3 advisor finding: a rank that raised here skipped
        # the barrier and desynchronised the group instead of falling back to RCCL).
        handle, err = None, None
        self.own, self.kind, self._opened = None, -1, []
        self.peers = (C.c_void_p * world)()
        try:
            with torch.cuda.device(self.device):
                for first_kind in (0, 1, 2):            # fine-grained, uncached, plain: the first kind that can be exported
                    ptr, kind = C.c_void_p(), C.c_int(first_kind)
                    _lib.check(l.xllm_mi355_ipc_alloc(total, C.byref(ptr), C.byref(kind)), "ipc_alloc")
                    buf = C.create_string_buffer(64)
                    if l.xllm_mi355_ipc_get_handle(ptr, buf) == 0:
                        handle = bytes(buf.raw)
                        self.own, self.kind = ptr, kind.value
                        break
                    l.xllm_mi355_ipc_free(ptr)
                    if kind.value >= 2:
                        break
            if handle is None:
                err = "hipIpcGetMemHandle failed for every memory kind"
        except Exception as e:   # noqa: BLE001
            err = f"buffer set-up failed: {e!r}"
        try:
            dev_id = str(torch.cuda.get_device_properties(self.device).uuid)
        except Exception:   # noqa: BLE001
            dev_id = f"{os.uname().nodename}:{self.device.index}"
        everyone = [None] * world
        dist.all_gather_object(everyone, (os.getpid(), handle, self.kind, dev_id, err), group=pg.group)
        export_errs = [f"rank {r}: {e}" for r, (_p, _h, _k, _d, e) in enumerate(everyone) if e or _h is None]
        if export_errs:
            self.refused = "; ".join(export_errs)
        else:
            try:
                with torch.cuda.device(self.device):
                    for r, (pid, h, _k, _d, _e) in enumerate(everyone):
                        if r == rank:
                            self.peers[r] = self.own.value
                            continue
                        p = C.c_void_p()
                        _lib.check(l.xllm_mi355_ipc_open_handle(h, C.byref(p)), f"ipc_open_handle(rank {r})")
                        self.peers[r] = p.value
                        self._opened.append(p)
            except Exception as e:   # noqa: BLE001
                self.refused = f"peer mapping failed: {e!r}"      # rank-local: agreed below
            same_device = len({d for (_p, _h, _k, d, _e) in everyone}) == 1
            # one rank per GPU: the fused kernel may use 256 blocks (one row per block for a 256-row decode message); ranks that
            # share a GPU keep 64 -- every block waits for its peers' blocks, so all grids must be co-resident (csrc/allreduce.hip).
            # `everyone` is the same list on every rank, so is the choice.
            if len({d for (_p, _h, _k, d, _e) in everyone}) == world:
                self.grid_limit = 256
            if any(k >= 2 for (_p, _h, k, _d, _e) in everyone) and not same_device:   # the same verdict on every rank
                self.refused = self.refused or ("only plain hipMalloc memory could be exported on some rank: not guaranteed "
                                                "visible to a peer GPU while the kernel runs")
        self.state = self.status = None
        try:
            self.state = torch.zeros(2, dtype=torch.int32, device=self.device)     # epoch, blocks-done counter
            self.status = torch.zeros(1, dtype=torch.int32, device=self.device)
            torch.cuda.synchronize(self.device)
        except Exception as e:   # noqa: BLE001
            self.refused = self.refused or f"state allocation failed: {e!r}"
        # agree: either every rank has every buffer mapped before the first launch, or nobody launches
        verdicts = [None] * world
        dist.all_gather_object(verdicts, self.refused, group=pg.group)
        bad = [f"rank {r}: {v}" for r, v in enumerate(verdicts) if v]
        if bad:
            self.refused = "; ".join(bad)

    # ---- which messages ------------------------------------------------------------------------------------------------
    def _stream_ok(self) -> bool:
        if self._suspended:
            return False
        cur = torch.cuda.current_stream(self.device).cuda_stream
        if self._stream is None:
            return True
        return cur == self._stream or torch.cuda.is_current_stream_capturing()

    def suspended(self):
        """context manager: every launch inside declines the one-shot kernel (RCCL / gloo serves it), whatever the stream and
        also under graph capture. For callers that issue collectives from SEVERAL streams of one rank -- DualBatchDecoder's
        two half-streams: all launches of a rank share one epoch word, one flag area and one set of data slots, so two of them
        running concurrently (parallel graph branches!) and in a rank-dependent order corrupt each other or time out. `capturing`
        alone cannot tell a forked branch from the bound stream, hence the explicit switch (round-4 advisor). Rank-invariant as
        long as every rank runs the same code path."""
        import contextlib

        @contextlib.contextmanager
        def cm():
            self._suspended += 1
            try:
                yield
            finally:
                self._suspended -= 1
        return cm()

    def takes(self, x: torch.Tensor) -> bool:
        """rank-invariant: dtype, element count, size limit, device kind (+ the one-stream rule)"""
        n = x.numel() * x.element_size()
        return bool(x.is_cuda and x.dtype in self._DT and 0 < n <= self.max_bytes and n % 16 == 0 and self._stream_ok())

    def _bind(self) -> int:
        s = torch.cuda.current_stream(self.device).cuda_stream
        if self._stream is None and not torch.cuda.is_current_stream_capturing():
            self._stream = s
        return s

    def rebind_stream(self) -> None:
        """a caller that moves ALL further launches to another stream (e.g. a graph's capture stream) says so explicitly"""
        torch.cuda.synchronize(self.device)
        self._stream = torch.cuda.current_stream(self.device).cuda_stream

    # ---- the collectives -----------------------------------------------------------------------------------------------
    def allreduce(self, x: torch.Tensor) -> None:
        l = self._lib.lib()
        s = self._bind()
        staged = None
        if not x.is_contiguous() or x.data_ptr() % 16:
            staged = x.contiguous() if not x.is_contiguous() else x.clone()      # fresh allocations are 256-byte aligned
        t = staged if staged is not None else x
        self._lib.check(l.xllm_mi355_oneshot_allreduce(
            t.data_ptr(), t.numel(), self._DT[t.dtype], self.peers, self.pg.rank(), self.pg.world_size(), self.max_bytes,
            self.state.data_ptr(), self.status.data_ptr(), self.timeout_s, s), "oneshot_allreduce")
        if staged is not None:
            x.copy_(staged)

    def allreduce_add_rms_norm(self, partial, residual, weight, eps: float, quantize: bool, want_sum: bool = False):
        """xllm_mi355_oneshot_allreduce_add_rms_norm; returns (q, scale) | norm16 (and the reduced sum when want_sum)"""
        l = self._lib.lib()
        s = self._bind()
        M, H = partial.shape
        if residual.shape != partial.shape or not residual.is_contiguous() or residual.dtype != partial.dtype:
            raise self._lib.Mi355Error("allreduce_add_rms_norm: residual [M, H] contiguous, same dtype as the partial sums")
        pc = partial if (partial.is_contiguous() and partial.data_ptr() % 16 == 0) else partial.contiguous().clone()
        dev = partial.device
        q = qs = n16 = ysum = None
        if quantize:
            q = torch.empty(M, H, dtype=torch.int8, device=dev)
            qs = torch.empty(M, dtype=torch.float32, device=dev)
        else:
            n16 = torch.empty(M, H, dtype=partial.dtype, device=dev)
        if want_sum:
            ysum = torch.empty(M, H, dtype=partial.dtype, device=dev)
        P = lambda t: 0 if t is None else t.data_ptr()
        rc = l.xllm_mi355_oneshot_allreduce_add_rms_norm(
            pc.data_ptr(), residual.data_ptr(), weight.data_ptr(), float(eps), P(n16), P(q), P(qs), P(ysum), M, H,
            self._DT[partial.dtype], self.peers, self.pg.rank(), self.pg.world_size(), self.max_bytes, self.state.data_ptr(),
            self.status.data_ptr(), self.timeout_s, self.grid_limit, s)
        self._lib.check(rc, "oneshot_allreduce_add_rms_norm")
        out = (q, qs) if quantize else n16
        return (out, ysum) if want_sum else out

    def matmul_allreduce_add_rms_norm(self, a_q, a_scale, w_packed, w_scale, bias, residual, weight, eps: float,
                                      quantize: bool, want_sum: bool = False):
        """xllm_mi355_scaled_matmul_oneshot_allreduce_add_rms_norm: the row-parallel W8A8 linear on packed weights whose int32
        K-slice sums are dequantised by step 1 of the one-shot kernel (no dequant pass, no 16-bit partial in memory). Returns like
        allreduce_add_rms_norm, or None when the GEMM declines the shape (decided by M, N, K alone: the same on every rank;
        nothing has been written then)."""
        from . import ops
        l = self._lib.lib()
        s = self._bind()
        M, K = a_q.shape
        N = w_scale.numel()
        if residual.shape != (M, N) or not residual.is_contiguous() or not a_q.is_contiguous():
            raise self._lib.Mi355Error("matmul_allreduce_add_rms_norm: a [M, K] and residual [M, N] contiguous")
        if M * N * residual.element_size() > self.max_bytes or M > 512 or not ops._prefer_packed(M, N, K):
            return None
        if N % 8 != 0 or N > 16384:     # the fused consumer's envelope, checked BEFORE the GEMM is launched
            return None
        ws = ops._slab_workspace(a_q.device)
        dev = a_q.device
        q = qs = n16 = ysum = None
        if quantize:
            q = torch.empty(M, N, dtype=torch.int8, device=dev)
            qs = torch.empty(M, dtype=torch.float32, device=dev)
        else:
            n16 = torch.empty(M, N, dtype=residual.dtype, device=dev)
        if want_sum:
            ysum = torch.empty(M, N, dtype=residual.dtype, device=dev)
        P = lambda t: 0 if t is None else t.data_ptr()
        rc = l.xllm_mi355_scaled_matmul_oneshot_allreduce_add_rms_norm(
            a_q.data_ptr(), w_packed.data_ptr(), a_scale.data_ptr(), w_scale.data_ptr(), P(bias), residual.data_ptr(),
            weight.data_ptr(), float(eps), P(n16), P(q), P(qs), P(ysum), M, N, K, self._DT[residual.dtype], ws.data_ptr(),
            ws.numel(), self.peers, self.pg.rank(), self.pg.world_size(), self.max_bytes, self.state.data_ptr(),
            self.status.data_ptr(), self.timeout_s, self.grid_limit, s)
        if rc in (-2, -4):
            return None
        self._lib.check(rc, "scaled_matmul_oneshot_allreduce_add_rms_norm")
        out = (q, qs) if quantize else n16
        return (out, ysum) if want_sum else out

    # ---- health --------------------------------------------------------------------------------------------------------
    def self_test(self) -> bool:
        """one checked message per size class through both kernels (every rank calls it; the caller agrees on the verdict):
        rank r sends (r + 1) * pattern, the sum must be pattern * W (W + 1) / 2 exactly (small integers: exact in bf16)"""
        world, rank = self.pg.world_size(), self.pg.rank()
        tri = world * (world + 1) // 2
        ok = True
        saved = self.timeout_s
        self.timeout_s = min(saved, 5.0)
        try:
            for n in (8, 4096, 256 * 3584, self.max_bytes // 2):
                if n * 2 > self.max_bytes:
                    continue
                pat = (torch.arange(n, device=self.device) % 7).to(torch.bfloat16)
                x = pat * float(rank + 1)
                self.allreduce(x)
                ok = ok and bool(torch.equal(x, pat * float(tri)))
            M, H = 8, 512
            pat = (torch.arange(M * H, device=self.device).view(M, H) % 5).to(torch.bfloat16)
            res = torch.ones(M, H, dtype=torch.bfloat16, device=self.device)
            w = torch.ones(H, dtype=torch.bfloat16, device=self.device)
            (n16, ysum) = self.allreduce_add_rms_norm(pat * float(rank + 1), res, w, 1e-6, quantize=False, want_sum=True)
            ok = ok and bool(torch.equal(ysum, pat * float(tri))) and bool(torch.equal(res, pat * float(tri) + 1.0))
            ok = ok and bool(torch.isfinite(n16.float()).all())
            if self.grid_limit > 64 and 256 * H * 2 <= self.max_bytes:   # (a buffer too small for it is rank-invariant: skipped by all)
                # one row per block (grid_limit = 256, every rank on a GPU of its own): production decode sends M = 256 rows, which
                # uses flag rows 64 .. 255 and relies on 256 co-resident blocks per rank -- the 8-row message above never touches
                # either (round-4 advisor): send one message of that regime before the verdict is agreed
                M2 = 256
                pat2 = (torch.arange(M2 * H, device=self.device).view(M2, H) % 5).to(torch.bfloat16)
                res2 = torch.ones(M2, H, dtype=torch.bfloat16, device=self.device)
                (n16b, ysum2) = self.allreduce_add_rms_norm(pat2 * float(rank + 1), res2, w, 1e-6, quantize=False, want_sum=True)
                ok = ok and bool(torch.equal(ysum2, pat2 * float(tri))) and bool(torch.equal(res2, pat2 * float(tri) + 1.0))
                ok = ok and bool(torch.isfinite(n16b.float()).all())
            # the GEMM-fed form: every rank multiplies ones by ones (sum = K) with a_scale = (rank + 1) / K -> partial = rank + 1
            from . import ops
            M, K, N = 8, 512, 512
            a = torch.ones(M, K, dtype=torch.int8, device=self.device)
            wp = ops.pack_weight_i8(torch.ones(N, K, dtype=torch.int8, device=self.device))
            a_s = torch.full((M,), float(rank + 1) / K, dtype=torch.float32, device=self.device)
            w_s = torch.ones(N, dtype=torch.float32, device=self.device)
            res = torch.ones(M, N, dtype=torch.bfloat16, device=self.device)
            w = torch.ones(N, dtype=torch.bfloat16, device=self.device)
            got = self.matmul_allreduce_add_rms_norm(a, a_s, wp, w_s, None, res, w, 1e-6, quantize=False, want_sum=True)
            if got is not None:    # (None: the packed GEMM is switched off -- the same on every rank)
                ok = ok and bool((got[1] == float(tri)).all()) and bool((res == float(tri) + 1.0).all())
            torch.cuda.synchronize(self.device)
            ok = ok and int(self.status.item()) == 0
        except Exception:   # noqa: BLE001
            ok = False
        finally:
            self.timeout_s = saved
        return ok

    def check(self) -> None:
        """host-side look at the status word (synchronises): raises if any launch gave up waiting for a peer"""
        if int(self.status.item()) != 0:
            raise self._lib.Mi355Error("one-shot all-reduce: a wait for a peer's flag timed out (result undefined)")

    def close(self) -> None:
        """collective: every rank of the group calls it (nobody unmaps while a peer may still read)"""
        l = self._lib.lib()
        if self.state is not None:
            torch.cuda.synchronize(self.device)
        dist.barrier(group=self.pg.group)
        for p in self._opened:
            l.xllm_mi355_ipc_close_handle(p)
        self._opened = []
        if self.own is not None:
            l.xllm_mi355_ipc_free(self.own)
            self.own = None


# ---- all-to-all expert parallelism (cfg5 names an RCCL all-to-all) ---------------------------------------
# Reference: DeepEPImpl (layers/common/deep_ep.{h,cpp}: dispatch_step -> process_dispatch_result -> combine_step_*; MLU
# only -- the CUDA / DCU layer combines EP ranks by all-reduce, fused_moe.cpp:308-315, which FusedMoE also mirrors).
# Here every rank owns DIFFERENT tokens (data-parallel attention) and E / world experts:
#   dispatch: the (token, k) pairs are grouped by destination rank (stable: token order inside a rank), the per-rank counts
#             are exchanged, then the rows and their LOCAL expert ids travel in one all-to-all each;
#   combine:  the expert outputs travel back through the mirrored all-to-all and are put back in (token, k) order, where the
#             usual weighted combine (moe_combine_result) finishes the layer.
# The split sizes are read on the host (one sync per layer, as DeepEP's normal mode does), so this path is for prefill;
# decode under graph replay keeps the all-reduce form.
def ep_dispatch(x: torch.Tensor, expert_ids: torch.Tensor, n_experts: int, pg: ProcessGroup):
    world = pg.world_size()
    e_local = n_experts // world
    topk = expert_ids.size(1)
    flat = expert_ids.reshape(-1).long()
    dest = torch.div(flat, e_local, rounding_mode="floor")
    order = torch.argsort(dest, stable=True)
    send_counts = torch.bincount(dest, minlength=world)
    if world == 1:
        recv_counts = send_counts
    else:
        recv_counts = torch.empty_like(send_counts)
        _run_collective(lambda: dist.all_to_all_single(recv_counts, send_counts, group=pg.group))
    sc, rc = send_counts.tolist(), recv_counts.tolist()
    rows = x.index_select(0, torch.div(order, topk, rounding_mode="floor"))
    local_e = (flat[order] - dest[order] * e_local).to(torch.int32)
    return pg.alltoall(rows, sc, rc), pg.alltoall(local_e, sc, rc), (order, sc, rc)


def ep_combine(y: torch.Tensor, ctx, pg: ProcessGroup) -> torch.Tensor:
    """y: one output row per received row (arrival order) -> [T * topk, H] at the source, in (token, k) order"""
    order, sc, rc = ctx
    back = pg.alltoall(y, rc, sc)
    out = torch.empty_like(back)
    out[order] = back
    return out


_piecewise = None  # the PiecewiseGraph that is capturing, if any


def _run_collective(fn) -> None:
    if _piecewise is not None:
        _piecewise.collective(fn)
    else:
        fn()


class PiecewiseGraph:
    """HIP-graph replay of a step that contains collectives (N2; reference: the piecewise graphs of
    kernels/dcu/piecewise_graphs.cpp + runtime/dcu_graph_executor_impl.cpp): the kernels between two collectives
    are captured into one graph each (all graphs share one memory pool and are replayed in capture order, so the
    tensors that cross a collective keep their addresses), the collectives themselves are issued eagerly between
    the graph launches -- RCCL never runs inside a capture. A 28-layer TP step becomes 57 graph launches + 57
    collectives instead of ~400 kernel launches from Python."""

    def __init__(self):
        self.items = []      # CUDAGraph | callable, in replay order
        self._pool = None
        self._ctx = None
        self._graph = None
        self._stream = None

    def _begin(self):
        self._graph = torch.cuda.CUDAGraph()
        self._ctx = torch.cuda.graph(self._graph, pool=self._pool, stream=self._stream)
        self._ctx.__enter__()

    def _end(self):
        self._ctx.__exit__(None, None, None)
        if self._pool is None:
            self._pool = self._graph.pool()
        self.items.append(self._graph)
        self._graph = self._ctx = None

    def collective(self, fn) -> None:
        self._end()
        fn()
        self.items.append(fn)
        self._begin()

    def capture(self, step_fn):
        """runs step_fn once under capture and returns its (static) output"""
        global _piecewise
        cur = torch.cuda.current_stream()
        self._stream = torch.cuda.Stream()
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            _piecewise = self
            self._begin()
            try:
                out = step_fn()
            finally:
                self._end()
                _piecewise = None
        cur.wait_stream(self._stream)
        return out

    def replay(self, skip_collectives: bool = False) -> None:
        """skip_collectives: timing aid (bench.py's exposed-communication figure): the graphs alone, wrong results"""
        for it in self.items:
            if isinstance(it, torch.cuda.CUDAGraph):
                it.replay()
            elif not skip_collectives:
                it()


def reduce(x: torch.Tensor, pg: Optional[ProcessGroup]) -> torch.Tensor:
    """parallel_state::reduce (parallel_state.cpp:183-192): in-place SUM all-reduce after a row-parallel linear."""
    if pg is None or pg.world_size() == 1:
        return x
    pg.allreduce(x)
    return x


class ReduceAsyncCtx:
    """parallel_state.h:37-40: the tensor being reduced in place + the pending Work (None: nothing in flight)"""
    __slots__ = ("tensor", "work")

    def __init__(self, tensor, work=None):
        self.tensor, self.work = tensor, work


def launch_reduce(x: torch.Tensor, pg: Optional[ProcessGroup]) -> ReduceAsyncCtx:
    """parallel_state::launch_reduce (parallel_state_async.cpp:72-84): start the SUM all-reduce of `x` and return at once;
    the caller keeps launching independent work (the next GEMM of the other micro-batch, the shared experts) and calls
    finish_reduce where it needs the sum. Inside a piecewise-graph capture there is nothing to overlap with on the host
    side (the collective sits between two graph launches), so the collective is issued in place."""
    if pg is None or pg.world_size() == 1:
        return ReduceAsyncCtx(x)
    xc = x.contiguous()
    # in place (no Work to wait for): inside a piecewise capture (the collective sits between two graph launches), inside a
    # plain capture (RCCL's async form must not run there), and whenever the one-shot kernel takes the message (a kernel on the
    # current stream: the caller's independent work simply queues behind it)
    if _piecewise is not None or (xc.is_cuda and torch.cuda.is_current_stream_capturing()) or \
            (pg.oneshot is not None and pg.oneshot.takes(xc)):
        pg.allreduce(xc)
        return ReduceAsyncCtx(xc)
    return ReduceAsyncCtx(xc, pg.allreduce_async(xc))


def finish_reduce(ctx: ReduceAsyncCtx) -> torch.Tensor:
    """parallel_state::finish_reduce (parallel_state.cpp:176-181)"""
    if ctx.work is not None:
        ctx.work.wait()
        ctx.work = None
    return ctx.tensor


def argmax_merge(val: torch.Tensor, idx: torch.Tensor, pg: Optional[ProcessGroup]) -> torch.Tensor:
    """greedy sampling over a column-sharded lm_head without gathering the logits: every rank holds, per row, the maximum of ITS
    shard (val float32 [B]) and its GLOBAL column (idx int64 [B]); the ranks exchange the [B] pairs and pick, per row, the best
    value -- NaN above every number, the lowest global column among equals -- i.e. torch.argmax of the gathered [B, V] logits
    (parallel_state::gather + Sampler::greedy_sample, linear.cpp:712-714, sampler.cpp:160-168; argmax commutes with the gather)"""
    if pg is None or pg.world_size() == 1:
        return idx
    vals = pg.allgather(val)                                        # [W, B]
    idxs = pg.allgather(idx)
    key = torch.where(torch.isnan(vals), torch.full_like(vals, float("inf")), vals)
    nan_rows = torch.isnan(vals).any(0, keepdim=True)
    key = torch.where(nan_rows & ~torch.isnan(vals), torch.full_like(vals, float("-inf")), key)   # a NaN beats +inf too
    best = key.max(0, keepdim=True).values
    cand = torch.where(key == best, idxs, torch.full_like(idxs, torch.iinfo(torch.int64).max))
    return cand.min(0).values


def gather(x: torch.Tensor, pg: Optional[ProcessGroup], dim: int = -1) -> torch.Tensor:
    """parallel_state::gather (parallel_state.cpp:89-102): all-gather-base then cat along `dim`."""
    if pg is None or pg.world_size() == 1:
        return x
    parts = pg.allgather(x)  # [world, ...]
    return torch.cat(list(parts.unbind(0)), dim=dim)


def make_tp_dp_groups(world_size: int, rank: int, tp_size: int):
    """TP groups = consecutive ranks (the reference builds tp_group the same way, process_group.cpp:218-229);
    DP replicas need no exchange on this path."""
    assert world_size % tp_size == 0
    tp_pg = None
    for r0 in range(0, world_size, tp_size):
        ranks = list(range(r0, r0 + tp_size))
        g = dist.new_group(ranks) if tp_size > 1 else None
        if rank in ranks:
            tp_pg = ProcessGroup(g, rank - r0, tp_size)
    return tp_pg, rank // tp_size
