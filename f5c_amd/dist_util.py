"""Multi-GPU plumbing: reads shard across ranks with no data-path collective (f5c_amd.synth.shard_batch);
the only communication is the final gather of per-rank statistics / per-read counts over RCCL
(`nccl` backend on ROCm; `gloo` in the CPU tests)."""
import numpy as np


def gather_stats(local, device="cuda"):
    """local: dict(elapsed, events, reads, pairs). Returns t_max and the sums over ranks (all ranks get them)."""
    import torch
    import torch.distributed as dist
    keys = ["elapsed", "events", "reads", "pairs"]
    v = torch.tensor([float(local[k]) for k in keys], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        allv = [torch.zeros_like(v) for _ in range(dist.get_world_size())]
        dist.all_gather(allv, v)
        allv = torch.stack(allv).cpu().numpy()
    else:
        allv = v.cpu().numpy()[None, :]
    return dict(t_max=float(allv[:, 0].max()), events=float(allv[:, 1].sum()), reads=float(allv[:, 2].sum()),
                pairs=float(allv[:, 3].sum()), per_rank=allv)


def gather_per_read(idx, n_pairs, n_total, device="cuda"):
    """Reassemble n_event_align_pairs[] of the whole batch from the shards (4 B/read)."""
    import torch
    import torch.distributed as dist
    full = torch.zeros(n_total, dtype=torch.int32, device=device)
    full[torch.as_tensor(np.asarray(idx), device=device)] = torch.as_tensor(np.asarray(n_pairs, dtype=np.int32),
                                                                             device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(full, op=dist.ReduceOp.SUM)       # shards are disjoint, so SUM == gather
    return full.cpu().numpy()
