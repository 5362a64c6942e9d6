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
        dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.group)

    def allgather(self, x: torch.Tensor) -> torch.Tensor:
        out = torch.empty((self._world,) + tuple(x.shape), dtype=x.dtype, device=x.device)
        # list form: supported by both RCCL and gloo (the CPU tests); the views alias `out`, no extra copy
        dist.all_gather(list(out.unbind(0)), x.contiguous(), group=self.group)
        return out


def reduce(x: torch.Tensor, pg: Optional[ProcessGroup]) -> torch.Tensor:
    """parallel_state::reduce (parallel_state.cpp:183-192): in-place SUM all-reduce after a row-parallel linear."""
    if pg is None or pg.world_size() == 1:
        return x
    pg.allreduce(x)
    return x


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
