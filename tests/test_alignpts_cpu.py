"""Image-free bead alignment (host logic): ``alignment_tools.translation_align_pts`` and ``align_beads(use_fft=False)``
against the reference's own outputs (tests/golden/alignpts.npz, made by oracle/make_golden_pts.py from
alignment_tools.py:356-419 and correction_tools/alignment.py:139-216)."""
import contextlib
import io
import numpy as np
import pytest
from conftest import load_golden


@pytest.mark.parametrize("name", ["beads", "random"])
def test_translation_align_pts_golden(name):
    from imageanalysis3_amd.alignment_tools import translation_align_pts
    g = load_golden("alignpts.npz")
    fix, tar = g[name + "_fix"], g[name + "_tar"]
    t, pf, pt = translation_align_pts(fix, tar, cutoff=2., return_pts=True)
    assert np.array_equal(t, g[name + "_t"]) and np.array_equal(pf, g[name + "_pf"]) and np.array_equal(pt, g[name + "_pt"])
    assert np.array_equal(translation_align_pts(fix, tar, cutoff=1., xyz_res=2), g[name + "_t_res2"])
    with pytest.raises(ValueError):
        translation_align_pts(fix, tar + 1000.0 * np.arange(len(tar))[:, None], cutoff=1e-9)


@pytest.mark.parametrize("name", ["beads", "random"])
def test_align_beads_without_fft_golden(name):
    from imageanalysis3_amd.correction_tools.alignment import align_beads
    g = load_golden("alignpts.npz")
    with contextlib.redirect_stdout(io.StringIO()):
        drift, ptar, pref = align_beads(g[name + "_tar"], g[name + "_fix"], use_fft=False, match_distance_th=2.,
                                        check_paired_cts=True, outlier_sigma=1.5, return_paired_cts=True, verbose=False)
    assert np.array_equal(drift, g[name + "_ab_drift"])
    assert np.array_equal(ptar, g[name + "_ab_tar"]) and np.array_equal(pref, g[name + "_ab_ref"])
