"""FOV sharding across GPUs and the final spot-table gather.

The reference fans (round-folder x FOV) images out over a ``multiprocessing.Pool`` with no exchange
between tasks (classes/field_of_view.py:1027-1138) and collects per-image spot tables into a
zero-padded ``(n_ids, max_seeds, 11) float32`` array (:1361-1382).  Here: one process per GPU,
static sharding of FOV indices, and ONE collective at the end — an all-gather of the per-FOV
counts and of the padded table (RCCL has no gatherv) — over RCCL/xGMI (``nccl`` backend) or gloo.
"""
import numpy as np


def shard_fovs(n_fovs, rank, world_size, keep_rounds_together=1):
    """Indices of the FOVs rank ``rank`` owns: blocks of ``keep_rounds_together`` consecutive images
    (e.g. all rounds of one FOV, so its reference bead crops stay resident) dealt round-robin."""
    n_fovs, k = int(n_fovs), max(1, int(keep_rounds_together))
    groups = np.arange((n_fovs + k - 1) // k)
    mine = groups[groups % world_size == rank]
    idx = (mine[:, None] * k + np.arange(k)[None, :]).reshape(-1)
    return idx[idx < n_fovs]


def pad_tables(tables, max_seeds):
    """list of (n_i, 11) float32 -> ((F, max_seeds, 11) zero-padded, (F,) int32 counts)."""
    F = len(tables)
    out = np.zeros((F, int(max_seeds), 11), dtype=np.float32)
    counts = np.zeros(F, dtype=np.int32)
    for i, t in enumerate(tables):
        t = np.asarray(t, dtype=np.float32).reshape(-1, 11)
        if len(t) > max_seeds:
            raise ValueError("table of FOV %d has %d rows > max_seeds=%d" % (i, len(t), max_seeds))
        out[i, :len(t)] = t
        counts[i] = len(t)
    return out, counts


def gather_spot_tables(rows, max_seeds, world_size=None):
    """All-gather one (n,11) table (or a list of per-FOV tables) per rank.

    Returns the concatenated (sum n, 11) float32 table in (rank, fov) order on every rank.  With a
    single process this is a no-op copy; otherwise two collectives: counts, then the padded block."""
    tables = rows if isinstance(rows, (list, tuple)) else [rows]
    if world_size is None:
        import torch.distributed as dist
        world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world_size == 1:
        return np.concatenate([np.asarray(t, dtype=np.float32).reshape(-1, 11) for t in tables], axis=0)
    import torch
    import torch.distributed as dist
    padded, counts = pad_tables(tables, max_seeds)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t_counts = torch.from_numpy(counts).to(dev)
    t_pad = torch.from_numpy(padded).to(dev)
    l_counts = [torch.empty_like(t_counts) for _ in range(world_size)]
    l_pad = [torch.empty_like(t_pad) for _ in range(world_size)]
    dist.all_gather(l_counts, t_counts)
    dist.all_gather(l_pad, t_pad)
    all_counts = torch.stack(l_counts).cpu().numpy()
    all_pad = torch.stack(l_pad).cpu().numpy()
    parts = []
    for r in range(world_size):
        for f in range(all_counts.shape[1]):
            parts.append(all_pad[r, f, :all_counts[r, f]])
    return np.concatenate(parts, axis=0) if parts else np.zeros((0, 11), dtype=np.float32)
