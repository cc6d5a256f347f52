#!/usr/bin/env python
"""configs[4] on one GPU through ia3_process_movies (io_tools.load.MoviePlan): seconds per movie for a few pipeline
shapes, the per-stage host times and the stage time stamps.  python scripts/time_movies.py [n_movies] [out.json]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np  # noqa: E402
import bench  # noqa: E402


def main():
    n_mov = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    shape = bench.SHAPE
    Z, X, Y = shape
    jobs = {"u16_%d" % k: ("fov", (shape, 5000, 40 + k, np.uint16), [(shape, np.uint16)]) for k in range(3)}
    jobs["beads"] = ("beads", (shape, 300, 43, (0.6, -3.4, 5.2), np.uint16), [(shape, np.uint16), (shape, np.uint16)])
    t0 = time.time()
    arrs = bench.generate(jobs, 8)
    print("generated in %.0f s" % (time.time() - t0), flush=True)
    if os.environ.get("IA3_WITH_TORCH"):   # as bench.py: torch imported and its device context made first
        import torch
        torch.cuda.set_device(0)
        x = torch.zeros(1 << 20).sum().item()
    from imageanalysis3_amd import _lib as L
    from imageanalysis3_amd.io_tools.load import MoviePlan, DeviceBuffer
    lib = L.lib()
    L.check(lib.ia3_init(0))
    ref_b, src_b = arrs["beads"]
    chs = ['750', '647', '561', '488']
    raws = []
    for k in range(3):
        r_ = np.empty((4 * Z, X, Y), np.uint16)
        for i in range(3):
            r_[i::4] = arrs["u16_%d" % ((i + k) % 3)][0]
        r_[3::4] = np.roll(src_b, (k, -2 * k), axis=(1, 2))
        raws.append(r_)
    yy, xx = np.meshgrid(np.linspace(-1, 1, Y), np.linspace(-1, 1, X))
    bump = (0.55 + 0.45 * np.exp(-(xx ** 2 + yy ** 2))).astype(np.float32)
    illum = {c: DeviceBuffer(bump) for c in chs}
    bleed = np.zeros((3, 3, X, Y), np.float32)
    for p in range(3):
        for q in range(3):
            bleed[p, q] = 1.0 if p == q else 0.05
    bleed = DeviceBuffer(bleed)
    zz = np.linspace(-1, 1, Z, dtype=np.float32)[:, None, None]
    field = np.stack([0.2 * zz + 0 * xx[None].astype(np.float32), (0.8 * xx[None] + 0 * zz).astype(np.float32),
                      (0.8 * yy[None] + 0 * zz).astype(np.float32)]).astype(np.float32)
    chrom = {'750': DeviceBuffer(field), '647': None, '561': DeviceBuffer(-field)}
    res = {}
    with L.DeviceStack.upload(ref_b) as dref:
        bw = bench._pcie_GBps(L, raws[0])
        print("PCIe %.1f GB/s -> %.4f s per movie" % (bw, raws[0].nbytes / 1e9 / bw), flush=True)
        res["pcie_GBps"] = bw
        shapes = ((3, 12, 2),) if os.environ.get('IA3_MOVIE_ONLY') else ((2, 12, 2), (3, 12, 2), (3, 24, 3), (3, 18, 3), (2, 24, 2), (4, 24, 3))
        for (nc, grp, ahead) in shapes:
            plan = MoviePlan(chs[:3], ref_image=dref, single_im_size=[Z, X, Y], all_channels=chs, num_buffer_frames=0,
                             num_empty_frames=0, calculate_drift=True, corr_channels=chs[:3], illumination_profile=illum,
                             bleed_profile=bleed, chromatic_profile=chrom, warp_image=True, verbose=True,
                             seed_th={c: 600.0 for c in chs[:3]}, fitting_args=dict(max_num_seeds=None), frames=4 * Z,
                             correct_threads=nc, fit_group_images=grp, upload_ahead=ahead)
            plan.run([raws[k % 3] for k in range(max(4, nc + 2))])
            L.check(lib.ia3_sync())
            reps = []
            for _ in range(3):
                t0 = time.perf_counter()
                out = plan.run([raws[k % 3] for k in range(n_mov)])
                reps.append(time.perf_counter() - t0)
            print("   runs:", [round(r / n_mov, 4) for r in reps], flush=True)
            dt = sorted(reps)[1]
            ms = {k_: float(np.mean([o["ms"][k_] for o in out])) for k_ in ("upload", "correct", "fit")}
            key = "correct%d_group%d_ahead%d" % (nc, grp, ahead)
            res[key] = {"s_per_movie": dt / n_mov, "stage_ms": ms, "rows": [len(t) for t in out[0]["tables"]],
                        "drift": [float(x) for x in out[0]["drift"]], "timeline_ms": [o["stamps"] for o in out]}
            print(key, "%.4f s per movie" % (dt / n_mov), {k_: round(v, 1) for k_, v in ms.items()},
                  res[key]["rows"], np.round(out[0]["drift"], 3), flush=True)
        # the same without any fit (upload + corrections only) and without uploads' competition: where the time goes
        plan = MoviePlan(chs[:3], ref_image=dref, single_im_size=[Z, X, Y], all_channels=chs, num_buffer_frames=0,
                         num_empty_frames=0, calculate_drift=True, corr_channels=chs[:3], illumination_profile=illum,
                         bleed_profile=bleed, chromatic_profile=chrom, warp_image=True, verbose=True, fit_spots=False,
                         frames=4 * Z, correct_threads=3)
        plan.run([raws[k % 3] for k in range(4)])
        t0 = time.perf_counter()
        out = plan.run([raws[k % 3] for k in range(n_mov)])
        dt = time.perf_counter() - t0
        res["no_fit"] = {"s_per_movie": dt / n_mov, "stage_ms": {k_: float(np.mean([o["ms"][k_] for o in out])) for k_ in ("upload", "correct")}}
        print("no fit: %.4f s per movie" % (dt / n_mov), res["no_fit"]["stage_ms"], flush=True)
        L.profile_enable(True); L.profile_collect()
        out = plan.run([raws[0]])
        prof = L.profile_collect(); L.profile_enable(False)
        res["one_movie_kernels_ms"] = {k: [v[0], round(v[1], 3)] for k, v in sorted(prof.items())}
        print("one movie, kernels:", res["one_movie_kernels_ms"], flush=True)
    if out_path:
        os.makedirs(os.path.dirname(os.path.abspath(out_path)), exist_ok=True)
        with open(out_path, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
