"""Stream sharding across the GPUs of one node: independent video streams, one per device, no data-path collective
(SURVEY.md section 8e).  Only the timing barrier / reductions go through torch.distributed (RCCL on GPUs, gloo on CPU)."""
import os


def rank_info():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def streams_for_rank(n_streams, rank, world):
    """stream i -> rank i mod world (round robin); returns the stream ids this rank owns."""
    return [i for i in range(n_streams) if i % world == rank]


def reduce_timing(elapsed_s, units, device=None):
    """(max elapsed over ranks, total units over ranks). Falls back to the local values without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(elapsed_s), int(units)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    u = torch.tensor([units], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), int(u.item())
