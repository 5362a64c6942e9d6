#!/usr/bin/env python3
"""round 6 (review next #5c): can an RCCL all-reduce issued from a FORKED stream be captured into a HIP graph next to kernels of another
branch -- the graph form of north_star's "RCCL all-reduce overlapped with the next GEMM on a side HIP stream" (reference:
parallel_state_async.cpp:72-84)? A one-GPU box cannot host two RCCL ranks (RCCL refuses two ranks on one device), so this probes
what one GPU can: a WORLD-SIZE-1 RCCL communicator (ncclAllReduce still runs, on c10d's own stream, fenced by events), captured
on a side stream under torch.cuda.graph with a GEMM on the other branch, replayed, checked. Prints one verdict line per step;
a failing call is reported with its exception text."""
import os
import sys
import traceback

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def step(name, fn):
    try:
        out = fn()
        print(f"[rccl-capture] {name}: ok" + (f" ({out})" if out is not None else ""), flush=True)
        return True
    except Exception as e:  # noqa: BLE001
        print(f"[rccl-capture] {name}: FAILED -- {type(e).__name__}: {str(e).splitlines()[0][:300]}", flush=True)
        traceback.print_exc(file=sys.stderr)
        return False


def main():
    torch.cuda.set_device(0)
    if not step("init_process_group(nccl, world_size=1)", lambda: dist.init_process_group("nccl", rank=0, world_size=1)):
        return
    x = torch.arange(1 << 20, device="cuda", dtype=torch.float32).bfloat16()
    a = torch.randn(2048, 2048, device="cuda", dtype=torch.bfloat16)
    step("eager all_reduce (warm-up, creates the communicator)", lambda: dist.all_reduce(x.clone()))
    _ = a @ a                                            # (the vendor GEMM initialises lazily: not inside a capture)
    torch.cuda.synchronize()
    y = x.clone()
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    cap = torch.cuda.Stream()

    def capture():
        with torch.cuda.stream(cap):
            with torch.cuda.graph(g, stream=cap):
                side.wait_stream(cap)                        # fork
                with torch.cuda.stream(side):
                    dist.all_reduce(y)                       # the collective on the side branch (c10d adds its own stream + events)
                b = a @ a                                    # "the next GEMM" on the main branch
                cap.wait_stream(side)                        # join
                z = b[:1, :1].float() + y[:1].float()        # consumes both branches
        return None

    if not step("capture: all_reduce on a forked stream + GEMM on the main branch", capture):
        return

    def replay():
        for _ in range(3):
            y.copy_(x)
            g.replay()
        torch.cuda.synchronize()
        return "sum over 1 rank == input: " + str(bool(torch.equal(y, x)))

    step("replay x3", replay)
    # the async form the overlap arm uses eagerly (Work.wait on the consumer stream) inside a capture
    g2 = torch.cuda.CUDAGraph()
    y2 = x.clone()

    def capture_async():
        with torch.cuda.stream(cap):
            with torch.cuda.graph(g2, stream=cap):
                w = dist.all_reduce(y2, async_op=True)
                b = a @ a
                w.wait()
                z = b[:1, :1].float() + y2[:1].float()
        return None

    if step("capture: all_reduce(async_op=True) ... GEMM ... Work.wait()", capture_async):
        step("replay x3 (async form)", lambda: [g2.replay() for _ in range(3)] and torch.cuda.synchronize())
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
