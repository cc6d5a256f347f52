"""Coordinate-space chromatic/drift translation of spot tables (reference: correction_tools/chromatic.py:41-143),
the alternative to warping the image that ``correct_fov_image(warp_image=False)`` hands back.  Host arithmetic on
(N,3) / (N,11) spot tables — a few thousand rows; nothing here touches the stack."""
import itertools
import pickle
import numpy as np


def generate_polynomial_data(coords, max_order):
    """Columns = all monomials of the coordinate columns up to ``max_order`` (orders ascending, within an order
    ``itertools.combinations_with_replacement`` order), n_points x n_columns (chromatic.py:123-143)."""
    coords = np.asarray(coords)
    cols = coords.transpose()
    feats = []
    for order in range(int(max_order) + 1):
        for combo in itertools.combinations_with_replacement(range(len(cols)), order):
            x = np.ones(coords.shape[0])
            for k in combo:
                x *= cols[k]
            feats.append(x)
    return np.array(feats).transpose()


def generate_chromatic_function(chromatic_const_file, drift=None):
    """chromatic.py:41-115 — returns ``f(coords)`` with ``coords`` (N,ndim) or an (N,11) spot table:
    ``coords - polynomial_shift(coords - ref_center) + drift``."""
    if isinstance(chromatic_const_file, dict):
        info = dict(chromatic_const_file)
    elif isinstance(chromatic_const_file, str):
        with open(chromatic_const_file, 'rb') as f:
            info = pickle.load(f)
    elif chromatic_const_file is None:
        if drift is None:
            return lambda _coords, _drift=None: _coords
        info = {'constants': [np.array([0]) for _ in drift],
                'fitting_orders': np.zeros(len(drift), dtype=int),
                'ref_center': np.zeros(len(drift))}
    else:
        raise TypeError("Wrong input chromatic_const_file")
    consts, orders, ref_center = info['constants'], info['fitting_orders'], info['ref_center']
    nd = len(ref_center)
    shift0 = np.zeros(nd) if drift is None else drift[:nd]

    def _shift_function(_coords, _drift=shift0, _consts=consts, _fitting_orders=orders, _ref_center=ref_center):
        if len(_coords) == 0:
            return _coords
        table = np.array(_coords)
        width = table.shape[1]
        if width == nd:
            pts = table.copy()
        elif width == 11:
            pts = table[:, 1:1 + nd].copy()
        else:
            raise ValueError("Wrong input coords")
        rel = pts - np.asarray(_ref_center)[np.newaxis, :]
        shifts = np.array([np.dot(generate_polynomial_data(rel, o), c)
                           for c, o in zip(_consts, _fitting_orders)]).transpose()
        moved = pts - shifts + _drift
        if width == nd:
            return moved
        out = table.copy()
        out[:, 1:1 + nd] = moved
        return out

    return _shift_function
