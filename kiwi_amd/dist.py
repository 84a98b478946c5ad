"""Multi-GPU driver helpers: one process per GPU, texts sharded by rank, no collective inside the analysis.

The analyze path shards by independent texts: every rank holds a model replica and analyses its
shard; the only exchange step is the FINAL RESULT GATHER -- an all-gather of the packed sizes, then a gather of the packed
token records (kamd_res_pack) to rank 0, which merges them back into input order (kamd_res_merge_strided).  Besides that:
a barrier and a MAX over ranks of the measured time.  On the GPU box the backend is "nccl" (= RCCL over xGMI; the packed
buffers travel as device tensors), the CPU tests run the same code over "gloo".
"""
from __future__ import annotations

import os


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_indices(n_texts: int, rank: int, world: int):
    """Strong-scaling split of one corpus: text i goes to rank i % world (keeps length mixes even)."""
    return list(range(rank, n_texts, world))


def weak_shard(texts, rank: int):
    """Weak scaling (bench.py): every rank analyses a same-sized batch; rotate so batches differ."""
    n = len(texts)
    if not n:
        return []
    shift = (rank * 977) % n
    return texts[shift:] + texts[:shift]


def init(backend: str, device_index: int | None = None):
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return
    if backend == "nccl":
        torch.cuda.set_device(device_index or 0)
        dist.init_process_group("nccl", device_id=torch.device("cuda", device_index or 0))
    else:
        dist.init_process_group(backend)


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device: str = "cpu") -> float:
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_counts(values, device: str = "cpu"):
    """All-gather of a small per-rank integer vector (e.g. [texts, tokens]); returns a list per rank."""
    import torch
    import torch.distributed as dist
    t = torch.tensor(list(values), dtype=torch.int64, device=device)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [t.tolist()]
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [o.tolist() for o in out]


def gather_packed(buf, device: str = "cpu", dst: int = 0):
    """Gathers one byte buffer per rank (numpy uint8: a rank's packed results) on rank `dst`.

    Message sizes differ per rank: the sizes are all-gathered first (one int64 each), then every rank contributes a buffer
    padded to the largest size to ONE gather collective (RCCL: ncclGather-style send/recv to rank 0 over xGMI; payload is
    tens of MB at a million sentences, far below a link's bandwidth, so padding costs nothing measurable).
    Returns the list of buffers (rank order) on `dst`, None elsewhere; with one process, [buf]."""
    import numpy as np
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return [buf]
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [s[0] for s in gather_counts([int(buf.nbytes)], device=device)]
    cap = max(max(sizes), 1)
    mine = torch.zeros(cap, dtype=torch.uint8, device=device)
    if buf.nbytes:
        mine[:buf.nbytes] = torch.from_numpy(np.ascontiguousarray(buf)).to(device)
    out = [torch.empty(cap, dtype=torch.uint8, device=device) for _ in range(world)] if rank == dst else None
    dist.gather(mine, out, dst=dst)
    if rank != dst:
        return None
    return [o[:n].cpu().numpy() for o, n in zip(out, sizes)]
