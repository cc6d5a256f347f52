"""CPU: libia3.so loads and exports every symbol include/ia3.h declares; host-side logic without a GPU."""
import os
import re
import ctypes as C
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "ia3.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ia3_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from imageanalysis3_amd import _lib
    lib = _lib.lib()
    names = _declared()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(set(_lib.EXPORTS)) == names, (set(names) ^ set(_lib.EXPORTS))


def test_struct_layout_matches_header(tmp_path):
    """ctypes mirrors of the parameter structs have the layout gcc gives include/ia3.h (sizes + every offset)."""
    import subprocess
    from imageanalysis3_amd import _lib
    pairs = [("ia3_seed_params", _lib.SeedParams), ("ia3_fit_params", _lib.FitParams),
             ("ia3_legacy_seed_params", _lib.LegacySeedParams), ("ia3_fov_job", _lib.FovJob),
             ("ia3_movie_params", _lib.MovieParams), ("ia3_movie_job", _lib.MovieJob)]
    lines = []
    for cname, ct in pairs:
        lines.append('printf("%s %%zu", sizeof(%s));' % (cname, cname))
        for f in ct._fields_:
            lines.append('printf(" %%zu", offsetof(%s, %s));' % (cname, f[0]))
        lines.append('printf("\\n");')
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "ia3.h"\nint main(void){%s return 0;}\n'
                   % "".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split("\n")
    for (cname, ct), line in zip(pairs, out):
        tok = line.split()
        assert tok[0] == cname and int(tok[1]) == C.sizeof(ct), (cname, tok[1], C.sizeof(ct))
        for f, off in zip(ct._fields_, tok[2:]):
            assert getattr(ct, f[0]).offset == int(off), (cname, f[0])


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from imageanalysis3_amd import _lib
    from imageanalysis3_amd.spot_tools.fitting import get_seeds
    im = np.zeros((8, 16, 16), dtype=np.float32)
    with pytest.raises(_lib.IA3Error) as e:
        get_seeds(im)
    assert "no CPU fallback" in str(e.value)


def test_argument_errors_before_device():
    from imageanalysis3_amd.spot_tools.fitting import get_seeds, fit_fov_image
    from imageanalysis3_amd import _lib
    with pytest.raises(TypeError):
        get_seeds("not an array")
    with pytest.raises(TypeError):
        fit_fov_image([[0]], "647", verbose=False)
    with pytest.raises(TypeError):
        _lib.as_stack_array(np.zeros((4, 4, 4), dtype=np.float64))
    with pytest.raises(IndexError):
        get_seeds(np.zeros((4, 4, 4), dtype=np.float32), sel_center=[1, 1])


def test_gaussian_taps_match_scipy():
    from scipy.ndimage import _filters
    from imageanalysis3_amd import _lib
    for sigma, tr in ((0.75, 4.0), (7.5, 4.0), (3, 2), (5, 2), (2.2, 3.0)):
        w, r = _lib.gaussian_taps(sigma, tr)
        ref = _filters._gaussian_kernel1d(float(sigma), 0, r)[::-1]
        assert r == int(tr * sigma + 0.5) and np.array_equal(w, ref)


def test_seed_param_threshold_promotion_rule():
    from imageanalysis3_amd import _lib
    assert _lib.make_seed_params(600)[0].th_compare_f32 == 1
    assert _lib.make_seed_params(600.0)[0].th_compare_f32 == 1
    assert _lib.make_seed_params(np.float64(600.0))[0].th_compare_f32 == 0
    assert _lib.make_seed_params(np.float32(600.0))[0].th_compare_f32 == 1


def test_containers():
    from imageanalysis3_amd.classes.preprocess import Spots3D, ImageCrop_3d, _3d_spot_infos
    from imageanalysis3_amd.io_tools.crop import generate_neighboring_crop
    assert _3d_spot_infos[:5] == ['height', 'z', 'x', 'y', 'background'] and len(_3d_spot_infos) == 11
    t = np.arange(33, dtype=np.float32).reshape(3, 11)
    s = Spots3D(t, bits=5, pixel_sizes=[200, 108, 108], channels='647')
    assert np.array_equal(s.to_coords(), t[:, 1:4]) and np.array_equal(s.to_intensities(), t[:, 0])
    assert np.array_equal(s.to_positions(), t[:, 1:4] * np.array([200, 108, 108]))
    assert list(s.bits) == [5, 5, 5] and list(s[1:].bits) == [5, 5]
    c = generate_neighboring_crop([5.2, 10.7, 3.1], crop_size=10, single_im_size=np.array([30, 64, 64]))
    assert c.to_slices() == (slice(0, 16), slice(1, 22), slice(0, 14))
    b = ImageCrop_3d([[0, 10], [5, 20], [5, 20]], [30, 64, 64])
    assert list(b.inside([[5, 6, 7], [11, 6, 7]])) == [True, False]
    assert np.array_equal(b.translate_drift([0.4, -2.6, 3]).array, [[0, 10], [8, 23], [2, 17]])


def test_chromatic_function_matches_reference_golden():
    """correction_tools/chromatic.py:41-143 (host arithmetic on spot tables) against the reference's own output."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    from make_golden import chromfn_inputs
    from imageanalysis3_amd.correction_tools.chromatic import generate_chromatic_function, generate_polynomial_data
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "chromfn.npz")))
    info, coords, spots, drift = chromfn_inputs()
    f = generate_chromatic_function(info, drift)
    assert np.array_equal(f(coords), g["coords"])
    out = f(spots)
    assert out.dtype == g["spots"].dtype and np.array_equal(out, g["spots"])
    assert np.array_equal(generate_chromatic_function(info, None)(coords), g["coords_nodrift"])
    assert np.array_equal(generate_chromatic_function(None, drift)(spots), g["drift_only"])
    assert np.array_equal(generate_polynomial_data(coords[:7], 2), g["poly2"])
    assert generate_chromatic_function(None, None)(spots) is spots
    with pytest.raises(ValueError):
        f(np.zeros((3, 5)))
    with pytest.raises(TypeError):
        generate_chromatic_function(3.0)
