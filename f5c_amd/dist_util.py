"""Multi-GPU plumbing: reads shard across ranks with no data-path collective (f5c_amd.synth.shard_batch);
the only communication is the final gather of per-rank statistics / per-read counts over RCCL
(`nccl` backend on ROCm; `gloo` in the CPU tests)."""
import numpy as np


def gather_rows(row, device="cuda"):
    """all_gather of one float64 row per rank -> array [world, len(row)] on every rank.  With an initialised process group the
    collective runs even at world size 1 (so that a 1-GPU box executes the RCCL path: `nccl` init + all_gather on `cuda`)."""
    import torch
    import torch.distributed as dist
    v = torch.tensor([float(x) for x in row], dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized():
        allv = [torch.zeros_like(v) for _ in range(dist.get_world_size())]
        dist.all_gather(allv, v)
        return torch.stack(allv).cpu().numpy()
    return v.cpu().numpy()[None, :]


def gather_stats(local, device="cuda"):
    """local: dict(elapsed, events, reads, pairs). Returns t_max and the sums over ranks (all ranks get them)."""
    allv = gather_rows([local[k] for k in ("elapsed", "events", "reads", "pairs")], device=device)
    return dict(t_max=float(allv[:, 0].max()), events=float(allv[:, 1].sum()), reads=float(allv[:, 2].sum()),
                pairs=float(allv[:, 3].sum()), per_rank=allv)


def gather_per_read(idx, n_pairs, n_total, device="cuda"):
    """Reassemble n_event_align_pairs[] of the whole batch from the shards (4 B/read)."""
    import torch
    import torch.distributed as dist
    full = torch.zeros(n_total, dtype=torch.int32, device=device)
    full[torch.as_tensor(np.asarray(idx), device=device)] = torch.as_tensor(np.asarray(n_pairs, dtype=np.int32),
                                                                             device=device)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(full, op=dist.ReduceOp.SUM)       # shards are disjoint, so SUM == gather
    return full.cpu().numpy()
