"""ProcessGroup mirror: the TP collectives the hot path issues.

Reference: xllm::ProcessGroup over c10d::ProcessGroupNCCL (framework/parallel_state/process_group.h:40-125,
cuda_process_group.h:24-53 -- on ROCm torch "nccl" IS RCCL) and the helpers parallel_state::reduce / gather
(parallel_state.cpp:89-102, 183-192). One process per GPU; collectives run over xGMI.
`world_size == 1` is a no-op exactly as in the reference (parallel_state.cpp:184-186).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist


class ProcessGroup:
    def __init__(self, group: Optional[dist.ProcessGroup] = None, rank: int = 0, world_size: int = 1):
        self.group, self._rank, self._world = group, rank, world_size

    def rank(self) -> int:
        return self._rank

    def world_size(self) -> int:
        return self._world

    def allreduce(self, x: torch.Tensor) -> None:
        _run_collective(lambda: dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group))

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
    if _piecewise is not None:
        pg.allreduce(xc)
        return ReduceAsyncCtx(xc)
    return ReduceAsyncCtx(xc, pg.allreduce_async(xc))


def finish_reduce(ctx: ReduceAsyncCtx) -> torch.Tensor:
    """parallel_state::finish_reduce (parallel_state.cpp:176-181)"""
    if ctx.work is not None:
        ctx.work.wait()
        ctx.work = None
    return ctx.tensor


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
