#!/usr/bin/env python
"""bench.py — per-FOV spot calling (DoG seed + 3-D Gaussian LM fit) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched through
torch.distributed.run (one rank per GPU, RCCL).  Prints ONE JSON line on rank 0.

Workload = BASELINE.json configs[1]: one 2048x2048x50 float32 FOV with ~5k spots per GPU per step
(synthetic, repo generator G((50,2048,2048), 5000, seed 3+rank)), resident in HBM before the timed
region.  A step = `fit_fov_image` semantics on that stack through the C ABI (`ia3_fit_fov_dev`:
get_seeds -> firstfit -> repeatfit -> NaN/boundary row filters), spot table returned to the host.
FOVs shard embarrassingly across ranks (weak scaling: one FOV per GPU per step); the only collective
is the final all-gather of the padded spot table (+ counts), inside the timed region.

value = fitted spots/s over all ranks; `fovs_per_sec` is reported beside it.
roofline: the dominant kernel by HIP-event time on the library stream (see DESIGN.md §4 for the
algorithmic bytes per launch); cpu_baseline: the NumPy/SciPy oracle (restatement of the reference's
CPU path, pinned against it) timed on a crop of the same FOV on this box's host cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

SHAPE = (50, 2048, 2048)
N_SPOTS = 5000
TH_SEED = 600.0
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
F64_VALU_PEAK_TFLOPS = 78.6    # MI355X FP64 vector peak (FMA = 2 flop)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--shape", type=int, nargs=3, default=list(SHAPE))
    ap.add_argument("--spots", type=int, default=N_SPOTS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-crop", type=int, default=384, help="x/y edge of the calibration crop of the CPU baseline")
    return ap.parse_args()


def algorithmic_bytes(kernel, shape, esize=4):
    """ALGORITHMIC HBM bytes of one launch: every Gaussian axis pass reads and writes the
    stack once (8 B/voxel for f32), and so does the fused three-axis short-filter kernel; seed_detect reads the
    two filtered stacks (8 B/voxel)."""
    vox = float(shape[0]) * shape[1] * shape[2]
    if kernel.startswith("gauss_axis") or kernel.startswith("gauss_fused3"):
        return 2 * esize * vox
    if kernel == "seed_detect":
        return 2 * esize * vox
    return None


def measured_traffic(kernel, shape):
    """HBM bytes per launch of `kernel` from the committed PMC passes (profiles/traffic.json: separate rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE runs of this same command, gfx950 read correction applied), or None when there is
    no measurement for this kernel and stack shape.  Counters cannot be collected from inside the timed run."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            t = json.load(f)
        if list(t.get("_shape", [])) != list(shape) or kernel not in t:
            return None
        return int(t[kernel]["read"] + t[kernel]["write"])
    except Exception:
        return None


def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Host-side work first: the synthetic FOV and (rank 0 of a 1-GPU run) the CPU baselines, which fork worker
    # processes - that has to happen before this process touches the GPU.
    from imageanalysis3_amd import synth
    shape = tuple(a.shape)
    t0 = time.time()
    im, centers, heights = synth.make_fov(shape, a.spots, 3 + rank)
    gen_s = time.time() - t0
    cpu_base = None
    if not a.no_cpu_baseline and world == 1:
        cpu_base = cpu_baseline(im, a.cpu_crop)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    # rehearsal hooks (one-GPU box): IA3_BENCH_DEVICE pins every rank to one device, IA3_BENCH_BACKEND=gloo swaps
    # the collective backend; the driver's multi-GPU runs set neither
    if os.environ.get("IA3_BENCH_DEVICE"):
        local_rank = int(os.environ["IA3_BENCH_DEVICE"])
    backend = os.environ.get("IA3_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    from imageanalysis3_amd import _lib as L
    from imageanalysis3_amd.parallel import gather_spot_tables
    import ctypes as C
    lib = L.lib()
    L.check(lib.ia3_init(local_rank))
    stack = L.DeviceStack.upload(im)
    sp, keep = L.make_seed_params(TH_SEED, max_num_seeds=None)
    fp = L.make_fit_params()
    cap = max(4 * a.spots, 16384)
    rows = np.empty((cap, 11), dtype=np.float32)

    def step():
        n_rows, n_seeds, n_iter = C.c_int(0), C.c_int(0), C.c_int(0)
        L.check(lib.ia3_fit_fov_dev(stack._h, C.byref(sp), C.byref(fp), L.ptr(rows), cap,
                                    C.byref(n_rows), C.byref(n_seeds), C.byref(n_iter)))
        return n_rows.value, n_seeds.value, n_iter.value

    def barrier():
        L.check(lib.ia3_sync())
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    warm = (0, 0, 0)
    for _ in range(a.warmup):
        warm = step()
    if world > 1:   # untimed: the first collective builds the communicator's rings and staging buffers
        gather_spot_tables(rows[:warm[0]], cap, world)
    L.profile_enable(True)
    L.profile_collect()
    barrier()
    t0 = time.perf_counter()
    total_rows = 0
    last = (0, 0, 0)
    for _ in range(a.steps):
        last = step()
        total_rows += last[0]
    # final spot-table all-gather (counts + padded [fovs, max_seeds, 11] f32), the path's only exchange
    table = gather_spot_tables(rows[:last[0]], cap, world)
    barrier()
    dt = time.perf_counter() - t0
    prof = L.profile_collect()
    L.profile_enable(False)
    if world > 1:
        t = torch.tensor([dt, float(total_rows)], dtype=torch.float64,
                         device="cuda" if dist.get_backend() == "nccl" else "cpu")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        dt_max, rows_all = float(tmax[0]), float(tsum[1])
    else:
        dt_max, rows_all = dt, float(total_rows)

    out = None
    if rank == 0:
        ms_per_step = dt_max / a.steps * 1e3
        # ---- roofline of the dominant kernel (HIP events on the library stream) --------------------
        kern = sorted(prof.items(), key=lambda kv: -kv[1][1])
        stage_ms = {k: v[1] / a.steps for k, v in prof.items()}
        roof = None
        # the front filter (gauss_fused3_*) and the first pass of the background filter (gauss_axis0_*) run side by side on
        # two streams: their event times include each other, so neither is used for the per-kernel roofline
        overlapped = [k for k in prof if k.startswith("gauss_fused3")]
        if overlapped:
            overlapped += [k for k in prof if k.startswith("gauss_axis0")]
        hbm_kernels = [(k, v) for k, v in kern if algorithmic_bytes(k, shape) is not None and k not in overlapped]
        dom_name, (dom_n, dom_ms) = kern[0]
        if (algorithmic_bytes(dom_name, shape) is None or dom_name in overlapped) and hbm_kernels:
            # the LM fit has no meaningful HBM figure (2 KB gathered per fit); report it in `fit`
            # below and give the HBM roofline of the heaviest stack-streaming kernel
            dom_name, (dom_n, dom_ms) = hbm_kernels[0]
        if algorithmic_bytes(dom_name, shape) is not None:
            avg_s = dom_ms / dom_n * 1e-3
            ach = algorithmic_bytes(dom_name, shape) / avg_s / 1e9
            roof = {"bound": "hbm", "kernel": dom_name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": measured_traffic(dom_name, shape),
                    "avg_launch_ms": round(dom_ms / dom_n, 4), "launches": dom_n,
                    "note": "scipy.ndimage's float64 accumulation (bit-identical results) makes the Gaussian passes "
                            "f64-VALU-bound: valu_frac = f64 instructions issued / peak issue rate"}
            if dom_name.startswith("gauss_axis") or dom_name.startswith("gauss_fused3"):
                R = int(dom_name.split("_R")[1])
                # per output: one multiply + per tap pair (add, multiply, add); the long passes fuse the last two where
                # the rounded result is provably the same (gauss.hip), so they issue 2 instructions per pair
                per_out = (2 * R + 1) if (R >= 16 and dom_name.startswith("gauss_axis")) else (3 * R + 1)
                ops = per_out * float(shape[0]) * shape[1] * shape[2]
                if dom_name.startswith("gauss_fused3"):
                    ops *= 3                                                  # three axes in one launch
                roof["valu_frac"] = round(ops / avg_s / (F64_VALU_PEAK_TFLOPS / 2 * 1e12), 4)
        filt_seed_ms = sum(v for k, v in stage_ms.items()
                           if (k.startswith("gauss") or k == "seed_detect") and k not in overlapped)
        if overlapped:
            filt_seed_ms += max(stage_ms[k] for k in overlapped)   # the two overlapped launches share their wall time
        vox_bytes = 4.0 * shape[0] * shape[1] * shape[2]
        out = {
            "metric": "fitted spots/sec (+ FOVs/sec), 2048x2048x50 float32 stack, ~5k spots/FOV",
            "value": round(rows_all / dt_max, 1), "unit": "spots/s",
            "fovs_per_sec": round(world * a.steps / dt_max, 3),
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "configs[1]: single %dx%dx%d float32 FOV, %d injected spots, DoG seed + LM fit"
                                   % (shape[1], shape[2], shape[0], a.spots),
                       "fovs_per_gpu_per_step": 1, "th_seed": TH_SEED, "parallelism": "fov-shard x%d" % world,
                       "spots_per_fov": last[0], "seeds_per_fov": last[1], "repeat_sweeps": last[2],
                       "gathered_table_rows": int(table.shape[0]) if table is not None else None},
            "roofline": roof,
            "stage_ms_per_fov": {k: round(v, 4) for k, v in sorted(stage_ms.items())},
            "stages_overlapped": sorted(overlapped),
            "filter_seed": {"ms_per_fov": round(filt_seed_ms, 4),
                            "algorithmic_GBps": round(vox_bytes / (filt_seed_ms * 1e-3) / 1e9, 1) if filt_seed_ms else None,
                            "frac_of_hbm_peak": round(vox_bytes / (filt_seed_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if filt_seed_ms else None},
            "host_gen_s": round(gen_s, 1),
        }
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_base   # None for multi-GPU runs (measured at N=1 only)
    stack.free()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def _cpu_run(args):
    """One oracle pass over a crop (module level: runs in forked workers too)."""
    sub, th = args
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import np_oracle as O
    t0 = time.perf_counter()
    t = O.fit_fov_image(sub, "647", th_seed=th, max_num_seeds=None, voronoi="lowest_index")
    return len(t), time.perf_counter() - t0


def cpu_baseline(im, crop):
    """The oracle (NumPy/SciPy restatement of the reference's CPU path: ndimage-exact filters, MINPACK
    lmder through scipy.optimize.leastsq, Python loop over seeds — same structure and cost profile as the
    reference) on a bounded crop of the same FOV.  A small calibration crop is timed first and the reported
    sample is then sized for roughly 15 s of CPU work on one core.  `all_cores` repeats the sample in one worker
    process per host core at once — the way the reference parallelises (one image per process,
    classes/field_of_view.py:1129-1138)."""
    def run(c):
        return _cpu_run((np.ascontiguousarray(im[:, :c, :c]), TH_SEED))

    side = min(im.shape[1], im.shape[2])
    crop = min(crop, side)
    n, dt = run(crop)
    target = 15.0
    if dt < 0.6 * target and crop < side:
        big = int(min(side, crop * (target / max(dt, 1e-3)) ** 0.5)) // 32 * 32
        if big > crop:
            crop = big
            n, dt = run(crop)
    frac = (crop * crop) / float(im.shape[1] * im.shape[2])
    out = {"value": round(n / dt, 2), "unit": "spots/s", "cores": 1, "kind": "port",
           "fovs_per_sec": round(frac / dt, 5), "seconds": round(dt, 1), "spots": int(n),
           "sample": "oracle fit_fov_image on the [0:%d, 0:%d, 0:%d] crop of the same FOV (%.1f%% of the voxels)"
                     % (im.shape[0], crop, crop, 100.0 * frac)}
    try:
        import multiprocessing as mp
        ncore = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        ncore = max(1, min(ncore, 16))   # a 1-GPU box gives this job a 16-core share
        if ncore > 1:
            sub = np.ascontiguousarray(im[:, :crop, :crop])
            t0 = time.perf_counter()
            with mp.get_context("fork").Pool(ncore) as pool:
                res = pool.map(_cpu_run, [(sub, TH_SEED)] * ncore)
            wall = time.perf_counter() - t0
            out["all_cores"] = {"value": round(sum(r[0] for r in res) / wall, 2), "unit": "spots/s", "cores": ncore,
                                "fovs_per_sec": round(ncore * frac / wall, 5), "seconds": round(wall, 1),
                                "sample": "the same crop in %d worker processes at once" % ncore}
    except Exception as e:   # the single-core figure is the contract; the pool is extra
        out["all_cores"] = {"error": str(e)}
    return out


if __name__ == "__main__":
    main()
