"""Round-2 probe (developer tool): H2D bandwidth of ia3_stack_upload with 0/2/4/8 staging helpers, per-kernel profile of
one float32 and one uint16 FOV, and ia3_fit_fovs (batch entry) throughput for resident and host inputs at several
in-flight depths."""
import ctypes as C, sys, time, os, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imageanalysis3_amd import synth, _lib as L
lib = L.lib(); L.check(lib.ia3_init(0))
SHAPE = (50, 2048, 2048)
res = {}
f32, c, h = synth.make_fov(SHAPE, 5000, 3, dtype=np.float32)
u16, c, h = synth.make_fov(SHAPE, 5000, 3, dtype=np.uint16)
print("generated", flush=True)
# ---- upload bandwidth --------------------------------------------------------------------------------------------
for name, im in (("f32", f32), ("u16", u16)):
    for T in (0, 4):
        L.check(lib.ia3_set_tuning(3, T))
        ts = []
        for rep in range(4):
            t0 = time.perf_counter(); st = L.DeviceStack.upload(im); dt = time.perf_counter() - t0; st.free()
            ts.append(dt)
        res["upload_%s_T%d_GBps" % (name, T)] = round(im.nbytes / min(ts[1:]) / 1e9, 2)
        print("upload", name, "helpers", T, ["%.1f ms" % (t * 1e3) for t in ts], res["upload_%s_T%d_GBps" % (name, T)], "GB/s", flush=True)
L.check(lib.ia3_set_tuning(3, 0))
sp, keep = L.make_seed_params(600.0, max_num_seeds=None); fp = L.make_fit_params()
# ---- one resident FOV, per-kernel --------------------------------------------------------------------------------
for name, im in (("f32", f32), ("u16", u16)):
    st = L.DeviceStack.upload(im)
    rows = np.empty((65536, 11), np.float32); nr, ns, ni = C.c_int(0), C.c_int(0), C.c_int(0)
    ts = []
    for rep in range(5):
        lib.ia3_sync(); t0 = time.perf_counter()
        L.check(lib.ia3_fit_fov_dev(st._h, C.byref(sp), C.byref(fp), L.ptr(rows), len(rows), C.byref(nr), C.byref(ns), C.byref(ni)))
        ts.append(time.perf_counter() - t0)
    a, b, d = C.c_int64(0), C.c_int64(0), C.c_int64(0); lib.ia3_fit_fov_stats(C.byref(a), C.byref(b), C.byref(d))
    print(name, "one FOV: %d seeds %d rows %d sweeps" % (ns.value, nr.value, ni.value), ["%.2f" % (t * 1e3) for t in ts],
          "fits %d nfev %d voxel_evals %d" % (a.value, b.value, d.value), flush=True)
    res["one_fov_%s_ms" % name] = round(min(ts) * 1e3, 3)
    L.profile_enable(True); L.profile_collect()
    L.check(lib.ia3_fit_fov_dev(st._h, C.byref(sp), C.byref(fp), L.ptr(rows), len(rows), C.byref(nr), C.byref(ns), C.byref(ni)))
    prof = L.profile_collect(); L.profile_enable(False)
    print("  ", {k: (v[0], round(v[1], 3)) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])}, flush=True)
    res["profile_%s" % name] = {k: [v[0], round(v[1], 3)] for k, v in prof.items()}
    # ---- batch entry, resident inputs ----
    for depth in (1, 4, 8, 12, 16):
        n = 16
        t0 = time.perf_counter()
        tabs, info = L.fit_fovs([st] * n, sp, fp, in_flight=depth)
        dt = time.perf_counter() - t0
        print("  batch resident %s depth %d: %.2f ms/FOV (%d rows each)" % (name, depth, dt / n * 1e3, len(tabs[0])), flush=True)
        res["batch_resident_%s_d%d_ms" % (name, depth)] = round(dt / n * 1e3, 3)
    st.free()
    # ---- batch entry, host inputs (upload inside) ----
    ims = [im, np.ascontiguousarray(im[:, ::-1]), np.ascontiguousarray(im[:, :, ::-1]), np.ascontiguousarray(im[::-1])]
    for depth in (1, 2, 3, 4, 8):
        n = 12
        t0 = time.perf_counter()
        tabs, info = L.fit_fovs([ims[k % 4] for k in range(n)], sp, fp, in_flight=depth)
        dt = time.perf_counter() - t0
        print("  batch host %s depth %d: %.2f ms/FOV -> %.1f GB/s through PCIe" % (name, depth, dt / n * 1e3, im.nbytes * n / dt / 1e9), flush=True)
        res["batch_host_%s_d%d_ms" % (name, depth)] = round(dt / n * 1e3, 3)
    del ims
print(json.dumps(res))
