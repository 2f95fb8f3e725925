"""Multi-GPU plumbing of the hot path: one process per GPU, rows sharded by contiguous range, NO data-path collective
(every ★ transformer is a pure per-row function, `transformation.do` treats same-schema runs
independently — pkg/transformer/transformation.go:236-282 — and sharded snapshot parts already hand each worker its own
row range, pkg/worker/tasks/load_sharded_snapshot.go).  torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in the
CPU tests) is used for exactly two things: the barrier around the timed region and the MAX of the per-rank wall time.

PyTorch is plumbing here, not the product: nothing in this module touches row data.
"""
from __future__ import annotations

import os
from typing import Tuple


def env_rank() -> Tuple[int, int, int]:
    """(rank, local_rank, world) from the torchrun environment; (0, 0, 1) when launched plainly."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_range(total_rows: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [lo, hi) of rank `rank` when `total_rows` are cut into `world` contiguous ranges (strong scaling:
    GPU g gets rows [g·N/G, (g+1)·N/G)).  Concatenating the ranks' outputs in rank order gives the
    single-GPU row order."""
    if world <= 0 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    return total_rows * rank // world, total_rows * (rank + 1) // world


def weak_shard(rows_per_rank: int, rank: int) -> Tuple[int, int]:
    """Weak scaling (bench.py): every rank parses its own `rows_per_rank` rows; rank r owns [r·n, (r+1)·n)."""
    return rows_per_rank * rank, rows_per_rank * (rank + 1)


class Group:
    """Barrier + max-over-ranks timing; a no-op group when world == 1."""

    def __init__(self, backend: str = "nccl", device=None):
        self.rank, self.local_rank, self.world = env_rank()
        self.dist = None
        self.device = device
        if self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            kw = {}
            if backend == "nccl" and device is not None:
                kw["device_id"] = device
            dist.init_process_group(backend=backend, **kw)
            self.dist = dist

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def max_seconds(self, dt: float) -> float:
        """MAX over ranks of a wall-clock interval: the job is as slow as its slowest rank."""
        if self.dist is None:
            return dt
        import torch
        t = torch.tensor([dt], dtype=torch.float64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_int(self, v: int) -> int:
        if self.dist is None:
            return v
        import torch
        t = torch.tensor([v], dtype=torch.int64, device=self.device if self.device is not None else "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return int(t.item())

    def all_gather_float(self, v: float):
        """[v of rank 0, v of rank 1, …] on every rank (one all_gather): what bench.py prints per rank so that the first N-GPU
        run validates itself (every rank present, no straggler hidden behind the MAX)."""
        if self.dist is None:
            return [float(v)]
        import torch
        dev = self.device if self.device is not None else "cpu"
        mine = torch.tensor([v], dtype=torch.float64, device=dev)
        outs = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(self.world)]
        self.dist.all_gather(outs, mine)
        return [float(t.item()) for t in outs]

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()
            self.dist = None
