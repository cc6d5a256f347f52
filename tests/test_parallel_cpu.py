"""CPU (gloo, world_size 2): FOV sharding and the spot-table all-gather — the N>1 path of bench.py."""
import os
import sys
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_fovs_partition():
    from imageanalysis3_amd.parallel import shard_fovs
    for n, w, k in ((512, 8, 1), (6000, 8, 60), (7, 2, 1), (10, 4, 3), (0, 2, 1)):
        parts = [shard_fovs(n, r, w, k) for r in range(w)]
        allidx = np.sort(np.concatenate(parts)) if n else np.zeros(0, int)
        assert np.array_equal(allidx, np.arange(n))
        if k > 1:
            for p in parts:  # groups of k consecutive images stay on one rank
                assert len(set(p // k)) * k >= len(p)
                for g in set(p // k):
                    assert all(np.isin(np.arange(g * k, min((g + 1) * k, n)), p))


def test_pad_tables():
    from imageanalysis3_amd.parallel import pad_tables
    a = np.arange(22, dtype=np.float32).reshape(2, 11)
    pad, cnt = pad_tables([a, np.zeros((0, 11))], 4)
    assert pad.shape == (2, 4, 11) and list(cnt) == [2, 0]
    assert np.array_equal(pad[0, :2], a) and not pad[0, 2:].any()
    with pytest.raises(ValueError):
        pad_tables([np.zeros((5, 11))], 4)


def _worker(rank, world, port, q, mode):
    try:
        sys.path.insert(0, ROOT)
        import torch.distributed as dist
        from imageanalysis3_amd.parallel import gather_spot_tables
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        rng = np.random.default_rng(rank)
        if mode == "even":
            tables = [rng.random((3 + rank, 11)).astype(np.float32), rng.random((rank, 11)).astype(np.float32)]
        elif mode == "uneven":   # 7 FOVs over 2 ranks: shard_fovs gives 4 and 3
            from imageanalysis3_amd.parallel import shard_fovs
            tables = [rng.random((1 + int(i) % 5, 11)).astype(np.float32) for i in shard_fovs(7, rank, world)]
        else:                    # a rank without any FOV
            tables = [] if rank == 1 else [rng.random((4, 11)).astype(np.float32)]
        out, counts = gather_spot_tables(tables, max_seeds=8, return_counts=True)
        assert [len(c) for c in counts] == ([2, 2] if mode == "even" else [4, 3] if mode == "uneven" else [1, 0])
        q.put((rank, out, tables))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:  # surface the failure instead of letting the parent time out
        q.put((rank, repr(e), None))


@pytest.mark.parametrize("mode", ["even", "uneven", "empty_rank"])
def test_gather_spot_tables_gloo_world2(mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + {"even": 0, "uneven": 1, "empty_rank": 2}[mode]
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for r in res:
        assert r[2] is not None, r[1]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = np.concatenate([t for _, _, tabs in res for t in tabs] + [np.zeros((0, 11), np.float32)], axis=0)
    for _, out, _ in res:
        assert out.shape == expect.shape and np.array_equal(out, expect)
