#!/usr/bin/env python
"""profiles/traffic.json (bench.py's `roofline.traffic`) from the FETCH_SIZE / WRITE_SIZE passes of one profile folder:
python profiles/make_traffic.py profiles/r02e"""
import csv, collections, json, os, sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]


SCOPE = {"gauss_axis0_folded<float, 50, 30, 3>": "gauss_axis0_pair", "gauss_xy_short<float, 3>": "gauss_xy_R3",
         "blockmin_k<float, 32, 4>": "seed_blockmin", "stripbound_k": "seed_blockmin", "seed_cand3_tiled<float, 64, 32>": "seed_detect",
         "bg_sparse_k<float>": "seed_sparse_bg", "fit_stages_k": "fit_first"}


def main(folder):
    out = {"_source": "%s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 2 "
                      "--warmup 1 --no-cpu-baseline --no-secondary --pool 1` (2048x2048x50 float32, 1 GPU); per-launch averages. "
                      "fetch_x1024 / write_x1024 = counter x 1024 B.  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes "
                      "of 16-B-per-lane streaming reads and WRITE_SIZE is exact for 16-B-per-lane stores (guide_read = 2 x fetch_x1024, "
                      "guide_write = write_x1024), other patterns to be calibrated on a known byte count.  Calibration on this pool: "
                      "blockmin_k (profiles/r02d, the per-plane block minima that stripbound_k replaces on the bench shape) reads exactly one stack (419.43 MB) once with 16-B loads and reports fetch_x1024 = 419.5 MB; the "
                      "column kernel stores exactly two stacks (838.86 MB) and reports write_x1024 = 1677.7 MB, the plane-wise kernel "
                      "stores one stack with 16-B stores and reports 842 MB.  Calibrated: read = fetch_x1024, write = write_x1024 / 2" % folder,
           "_shape": [50, 2048, 2048], "_read_factor": 1.0, "_write_factor": 0.5}
    for tag, f in (("fetch_x1024", "pmc_fetch_size.csv"), ("write_x1024", "pmc_write_size.csv")):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(os.path.join(folder, f))):
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if k in SCOPE:
                out.setdefault(SCOPE[k], {})[tag] = round(sum(v) / len(v) * 1024, -5)
    for v in out.values():
        if isinstance(v, dict):
            v["read"] = out["_read_factor"] * v["fetch_x1024"]
            v["write"] = out["_write_factor"] * v["write_x1024"]
            v["guide_read"] = 2 * v["fetch_x1024"]
            v["guide_write"] = v["write_x1024"]
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1].rstrip("/"))
