"""One align_image(use_autocorr=True) on a 50x1024x1024 bead pair (developer tool for rocprofv3)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from imageanalysis3_amd import synth, _lib as L
from imageanalysis3_amd.correction_tools.alignment import align_image
shape = (50, 2048, 2048)
ref, src, c, h = synth.make_bead_pair(shape, 400, 21, (0.6, -3.4, 5.2), dtype=np.uint16)
for _ in range(2):
    d, flag = align_image(src, ref, use_autocorr=True, verbose=False, correction_args={'single_im_size': shape})
print(d, flag)
