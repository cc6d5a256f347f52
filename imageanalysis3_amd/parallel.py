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


def gather_spot_tables(rows, max_seeds, world_size=None, return_counts=False):
    """All-gather the per-FOV tables of every rank: ``rows`` is one (n,11) table or a list of them (any number per rank,
    none included — the reference's pool takes any task count, classes/field_of_view.py:1129-1142).

    Returns the concatenated (sum n, 11) float32 table in (rank, fov) order on every rank (with ``return_counts`` also
    the list of per-rank int32 count vectors).  With a single process this is a no-op copy; otherwise one tiny
    all-reduce (the largest number of FOVs any rank holds, so that every rank contributes blocks of the same shape —
    RCCL has no gatherv) and two ``all_gather_into_tensor`` collectives: the counts ``int32 [F]`` and the zero-padded
    block ``float32 [F, max_seeds, 11]``, the reference's on-disk layout (:1361-1382).  With the ``nccl`` (= RCCL)
    backend both live on the device from one upload to one download; gloo stays on the host."""
    tables = rows if isinstance(rows, (list, tuple)) else [rows]
    tables = [np.asarray(t, dtype=np.float32).reshape(-1, 11) for t in tables]
    if world_size is None:
        import torch.distributed as dist
        world_size = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if world_size == 1:
        out = np.concatenate(tables, axis=0) if tables else np.zeros((0, 11), dtype=np.float32)
        return (out, [np.array([len(t) for t in tables], dtype=np.int32)]) if return_counts else out
    import torch
    import torch.distributed as dist
    nccl = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")
    # every rank contributes F blocks: F = the most FOVs any rank holds, missing ones are empty (count 0)
    t_f = torch.tensor([len(tables)], dtype=torch.int32, device=dev)
    dist.all_reduce(t_f, op=dist.ReduceOp.MAX)
    F = int(t_f.item())
    if F == 0:
        out = np.zeros((0, 11), dtype=np.float32)
        return (out, [np.zeros(0, np.int32) for _ in range(world_size)]) if return_counts else out
    padded, counts = pad_tables(tables + [np.zeros((0, 11), np.float32)] * (F - len(tables)), max_seeds)
    nmine = np.array([len(tables)], dtype=np.int32)
    head = np.concatenate([nmine, counts])                     # [number of real FOVs | counts of the F blocks]
    t_head = torch.from_numpy(head).to(dev)
    t_pad = torch.from_numpy(padded).to(dev)
    # (outputs are the rank blocks concatenated along the first axis: the form both backends accept)
    g_head = torch.empty((world_size * (F + 1),), dtype=torch.int32, device=dev)
    g_pad = torch.empty((world_size * F, int(max_seeds), 11), dtype=torch.float32, device=dev)
    dist.all_gather_into_tensor(g_head, t_head)
    dist.all_gather_into_tensor(g_pad, t_pad)
    all_head = g_head.cpu().numpy().reshape(world_size, F + 1)
    all_pad = g_pad.cpu().numpy().reshape(world_size, F, int(max_seeds), 11)
    parts, per_rank = [], []
    for r in range(world_size):
        nr = int(all_head[r, 0])
        per_rank.append(all_head[r, 1:1 + nr].astype(np.int32))
        for f in range(nr):
            parts.append(all_pad[r, f, :all_head[r, 1 + f]])
    out = np.concatenate(parts, axis=0) if parts else np.zeros((0, 11), dtype=np.float32)
    return (out, per_rank) if return_counts else out
