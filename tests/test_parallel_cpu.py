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


def test_bench_gpus_flag_must_match_the_launcher():
    """`bench.py --gpus N` either starts N ranks itself or runs under a launcher that did: a WORLD_SIZE that disagrees
    with --gpus, or --gpus 0, ends the run before any work is done (non-zero exit, message on stderr)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr, r.stderr[-400:]
    env.pop("WORLD_SIZE")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "0"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "--gpus" in r.stderr


def test_bench_launches_its_own_ranks(tmp_path, monkeypatch):
    """Without a launcher `--gpus 2` goes through launch_ranks: child ranks are started with torch.distributed.run on
    127.0.0.1 before the parent touches a GPU, rank 0's line is relayed, and a line that does not report 2 ranks is an
    error.  (The children need GPUs; here the launcher command is replaced by a stub that plays rank 0.)"""
    import subprocess
    import sys
    import json
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class FakeProc(object):
        def __init__(self, line):
            self.stdout = iter(["rank chatter\n", line + "\n"])

        def wait(self):
            return 0

    def fake_popen(cmd, stdout=None, env=None, text=None):
        seen["cmd"], seen["env"] = cmd, env
        return FakeProc(seen["line"])

    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    a = bench.parse()
    seen["line"] = json.dumps({"metric": "m", "n_gpus": 2, "ms_per_step_by_rank": [1.0, 1.1]})
    bench.launch_ranks(a)
    cmd = seen["cmd"]
    assert "torch.distributed.run" in cmd and "--nproc-per-node=2" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "2", "--steps", "3"] and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    seen["line"] = json.dumps({"metric": "m", "n_gpus": 1, "ms_per_step_by_rank": [1.0]})
    with pytest.raises(SystemExit):
        bench.launch_ranks(a)
