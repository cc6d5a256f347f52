#!/usr/bin/env python
"""profiles/traffic.json (bench.py's `roofline.traffic`) from the FETCH_SIZE / WRITE_SIZE passes of one profile folder:
python profiles/make_traffic.py profiles/r02e"""
import csv, collections, json, os, sys


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("ia3colk::", "").replace("void ", "").split("(")[0]


SCOPE = {"gauss_axis0_folded<float, 50, 30, 3>": "gauss_axis0_pair", "gauss_xy_short<float, 3>": "gauss_xy_R3",
         "blockmin_k<float, 32, 4>": "seed_blockmin", "stripbound_k": "seed_blockmin", "seed_cand3_tiled<float, 64, 32>": "seed_detect",
         "seed_cand3_tiled<float, 32, 32>": "seed_detect",
         "bg_sparse_k<float>": "seed_sparse_bg", "bg_sparse3_k<float, 30>": "seed_sparse_bg", "fit_stages_k": "fit_first"}


def main(folder):
    out = {"_source": "%s: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python3 bench.py --steps 2 "
                      "--warmup 1 --no-cpu-baseline --no-secondary --pool 1` (2048x2048x50 float32 = 838.9 MB per stack, 1 GPU); "
                      "per-launch averages; read = 2 x FETCH_SIZE x 1024 B (gfx950 correction of MI355X_MICROARCH.md), write = "
                      "WRITE_SIZE x 1024 B.  Checked on kernels of this path whose byte counts are known exactly: blockmin_k "
                      "(profiles/r02d) reads one stack once with 16-B loads: 2 x 419.5 = 839.0 MB; the column kernel reads one stack "
                      "(2 x 420.5 = 841 MB) and stores two (1677.7 MB counted = 2 x 838.9)" % folder,
           "_shape": [50, 2048, 2048]}
    for tag, f, mul in (("read", "pmc_fetch_size.csv", 2.0), ("write", "pmc_write_size.csv", 1.0)):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(os.path.join(folder, f))):
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if k in SCOPE:
                out.setdefault(SCOPE[k], {})[tag] = round(mul * sum(v) / len(v) * 1024, -5)
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1].rstrip("/"))
