"""TEST INFRASTRUCTURE ONLY — loader for the upstream reference's hot-path files.

Runs only where ``/root/reference`` exists (the development container).  It is
used by ``oracle/make_golden.py`` to produce the committed fixtures under
``tests/golden/`` and by the optional ``tests/test_oracle_vs_reference.py``
(skipped when the reference tree is absent, e.g. on the GPU box).

Nothing in the product package imports this module.  It contains no reference
code: it only wires ``importlib`` so the reference's own files can be executed
file-by-file (``import ImageAnalysis3`` as a package is impossible because its
``__init__`` eagerly imports cv2 / skimage / h5py / pyfftw, all absent here).
Recipe: SURVEY.md Appendix A.
"""
import os
import sys
import types
import importlib.util
import warnings

REF = os.environ.get("IA3_REFERENCE", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF, "External", "Fitting_v4.py"))


_loaded = None


def load_reference():
    """Return a namespace with the reference modules (cached)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    warnings.filterwarnings("ignore")
    import numpy as np
    import scipy.signal
    import matplotlib
    matplotlib.use("Agg")
    for n, t in (("int", int), ("bool", bool), ("float", float)):  # removed in numpy>=1.24
        if not hasattr(np, n):
            setattr(np, n, t)
    if not hasattr(scipy.signal, "gaussian"):
        scipy.signal.gaussian = scipy.signal.windows.gaussian

    def stub(name, **kw):
        m = types.ModuleType(name)
        m.__dict__.update(kw)
        sys.modules[name] = m
        return m

    def load(modname, path):
        spec = importlib.util.spec_from_file_location(modname, path)
        m = importlib.util.module_from_spec(spec)
        sys.modules[modname] = m
        spec.loader.exec_module(m)
        return m

    pfn = stub("pyfftw.interfaces.numpy_fft", rfftn=np.fft.rfftn, irfftn=np.fft.irfftn)
    stub("pyfftw", interfaces=stub("pyfftw.interfaces", numpy_fft=pfn))
    stub("cv2")
    try:   # the real scikit-image where an interpreter has it (/opt/conda's python3.9: 0.18.3)
        import skimage.registration  # noqa: F401
        import skimage.segmentation  # noqa: F401
    except Exception:
        stub("skimage", morphology=None, restoration=None, measure=None)
        stub("skimage.segmentation", random_walker=None)

        def _no_pcc(*a, **k):
            raise RuntimeError("scikit-image not installed")

        stub("skimage.registration", phase_cross_correlation=_no_pcc)
    G = dict(_correction_folder="", _temp_folder="", _distance_zxy=[200, 108, 108],
             _sigma_zxy=[1.35, 1.9, 1.9],
             _allowed_colors=["750", "647", "561", "488", "405"],
             _image_size=[30, 2048, 2048], _corr_channels=["750", "647", "561"],
             _num_buffer_frames=0, _num_empty_frames=0, _image_dtype=np.uint16)
    root = stub("IA3", **G)
    root.__path__ = []
    ext = stub("IA3.External")
    ext.__path__ = []
    F4 = load("IA3.External.Fitting_v4", REF + "/External/Fitting_v4.py")
    ext.Fitting_v4 = F4
    ext.Fitting_v3 = F4
    vt = stub("IA3.visual_tools", get_seed_points_base=F4.get_seed_points_base,
              translate_spot_coordinates=None)
    root.visual_tools = vt
    root.get_img_info = stub("IA3.get_img_info")
    root.corrections = stub("IA3.corrections")
    st = stub("IA3.spot_tools", _seed_th={"750": 600, "647": 600, "561": 600}, **G)
    st.__path__ = []
    fitting = load("IA3.spot_tools.fitting", REF + "/spot_tools/fitting.py")
    ct = stub("IA3.correction_tools")
    ct.__path__ = []
    filt = load("IA3.correction_tools.filter", REF + "/correction_tools/filter.py")
    trans = load("IA3.correction_tools.translate", REF + "/correction_tools/translate.py")
    at = load("IA3.alignment_tools", REF + "/alignment_tools.py")
    root.alignment_tools = at
    matching = load("IA3.spot_tools.matching", REF + "/spot_tools/matching.py")
    st.matching = matching
    stub("IA3.io_tools").__path__ = []
    stub("IA3.io_tools.load", correct_fov_image=None)
    align = load("IA3.correction_tools.alignment", REF + "/correction_tools/alignment.py")
    ns = types.SimpleNamespace(F4=F4, fitting=fitting, filter=filt, translate=trans,
                               alignment_tools=at, matching=matching, alignment=align)
    ns._load, ns._ext, ns._root = load, ext, root
    _loaded = ns
    return ns


def load_legacy():
    """Additionally execute the reference's External/Fitting_v3.py and visual_tools.py (the legacy per-cell
    path, classes/__init__.py:57-88).  Returns (Fitting_v3 module, visual_tools module)."""
    ns = load_reference()
    if getattr(ns, "F3", None) is not None:
        return ns.F3, ns.visual_tools
    ns._ext._sigma_zxy = [1.35, 1.9, 1.9]
    F3 = ns._load("IA3.External.Fitting_v3", REF + "/External/Fitting_v3.py")
    ns._ext.Fitting_v3 = F3
    vt = ns._load("IA3.visual_tools", REF + "/visual_tools.py")
    ns._root.visual_tools = vt
    ns.F3, ns.visual_tools = F3, vt
    return F3, vt


def load_io():
    """Execute the reference's io_tools/crop.py, io_tools/load.py and classes/preprocess.py (for the background
    normalisation of fit_fov_image, spot_tools/fitting.py:240-258).  Returns (load module, crop module)."""
    ns = load_reference()
    if getattr(ns, "io_load", None) is not None:
        return ns.io_load, ns.io_crop
    import sys
    load_legacy()   # classes/preprocess.py:330 imports DaxReader from the real visual_tools
    if "h5py" not in sys.modules:
        sys.modules["h5py"] = types.ModuleType("h5py")   # imported at preprocess.py:335, unused on this path
    G = {k: getattr(ns._root, k) for k in ("_distance_zxy", "_image_size", "_allowed_colors", "_corr_channels",
                                           "_correction_folder", "_num_buffer_frames", "_num_empty_frames",
                                           "_image_dtype")}
    io = sys.modules["IA3.io_tools"]
    io.__dict__.update(G)
    cl = types.ModuleType("IA3.classes")
    cl.__path__ = []
    sys.modules["IA3.classes"] = cl
    crop = ns._load("IA3.io_tools.crop", REF + "/io_tools/crop.py")
    io.crop = crop
    load = ns._load("IA3.io_tools.load", REF + "/io_tools/load.py")
    io.load = load
    pre = ns._load("IA3.classes.preprocess", REF + "/classes/preprocess.py")
    cl.preprocess = pre
    ns.io_load, ns.io_crop = load, crop
    return load, crop


def load_corrections():
    """Additionally execute the reference's corrections.py (Remove_Hot_Pixels / Z_Shift_Correction used by
    io_tools/load.py:323-345) and rebind it where io_tools.load looks it up.  Returns the module."""
    ns = load_reference()
    if getattr(ns, "corrections", None) is not None:
        return ns.corrections
    import sys
    load, crop = load_io()
    ns._root._temp_folder = ""
    ns._root.io_tools = sys.modules["IA3.io_tools"]
    cor = ns._load("IA3.corrections", REF + "/corrections.py")
    ns._root.corrections = cor
    load.corrections = cor
    ns.corrections = cor
    return cor


def load_chromatic():
    """Execute the reference's correction_tools/chromatic.py (generate_chromatic_function)."""
    ns = load_reference()
    if getattr(ns, "chromatic", None) is not None:
        return ns.chromatic
    import sys
    load_io()
    ct = sys.modules["IA3.correction_tools"]
    ct._drift_channel = '488'
    ch = ns._load("IA3.correction_tools.chromatic", REF + "/correction_tools/chromatic.py")
    ns.chromatic = ch
    return ch


def load_batch():
    """Execute the reference's classes/batch_functions.py (save-file helpers + batch_process_image_to_spots).
    Needs h5py: run under an interpreter that has it (``/opt/conda/bin/python3.9`` in this image; see
    oracle/make_golden_h5.py).  Returns the module."""
    ns = load_reference()
    if getattr(ns, "batch", None) is not None:
        return ns.batch
    import ast
    import sys
    import h5py  # noqa: F401  (the real one)
    load_corrections()
    load_chromatic()
    cl = sys.modules["IA3.classes"]
    src = open(REF + "/classes/__init__.py").read()
    for node in ast.parse(src).body:   # the literal dict of allowed data types, :22-32
        if isinstance(node, ast.Assign) and getattr(node.targets[0], "id", "") == "_allowed_kwds":
            cl._allowed_kwds = ast.literal_eval(node.value)
    cl._image_dtype = ns._root._image_dtype
    sys.modules["h5py"] = h5py   # load_io may have parked an empty stand-in there
    b = ns._load("IA3.classes.batch_functions", REF + "/classes/batch_functions.py")
    ns.batch = b
    return b


def load_partition():
    """Execute what DaxProcesser._fit_spots_by_segmentation imports (classes/preprocess.py:1102-1104): the reference's
    segmentation_tools/cell.py (for segmentation_mask_2_bounding_box) and classes/partition_spots.py (for
    Spots_Partition.spots_to_labels).  Plotting modules are stubbed.  Returns (cell module, partition module)."""
    ns = load_reference()
    if getattr(ns, "partition", None) is not None:
        return ns.seg_cell, ns.partition
    import sys
    load_corrections()
    cl = sys.modules["IA3.classes"]
    cl.default_pixel_sizes = [250, 108, 108]
    ft = types.ModuleType("IA3.figure_tools")
    ft.__path__ = []
    sys.modules["IA3.figure_tools"] = ft
    for name, attr in (("plot_segmentation", "plot_segmentation"), ("plot_partition", "plot_cell_spot_counts")):
        m = types.ModuleType("IA3.figure_tools." + name)
        setattr(m, attr, None)
        sys.modules["IA3.figure_tools." + name] = m
    io = sys.modules["IA3.io_tools"]
    io.parameters = ns._load("IA3.io_tools.parameters", REF + "/io_tools/parameters.py")
    io.spots = ns._load("IA3.io_tools.spots", REF + "/io_tools/spots.py")
    import numpy as np
    import numpy.lib.npyio as _npyio
    if not hasattr(_npyio, "save"):   # cell.py:3 imports it from its pre-2.0 location (unused)
        _npyio.save = np.save
    st = types.ModuleType("IA3.segmentation_tools")
    st.__path__ = []
    sys.modules["IA3.segmentation_tools"] = st
    cell = ns._load("IA3.segmentation_tools.cell", REF + "/segmentation_tools/cell.py")
    st.cell = cell
    part = ns._load("IA3.classes.partition_spots", REF + "/classes/partition_spots.py")
    cl.partition_spots = part
    ns.seg_cell, ns.partition = cell, part
    return cell, part
