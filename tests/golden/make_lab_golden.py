#!/opt/conda/bin/python3.9
"""Pins the rgb2lab restatement (oracle/boxinst_oracle.c: bxo_rgb2lab_u8; kernels: csrc/image_device.hpp) to the REAL
scikit-image, which the build container happens to carry in an Anaconda tree (/opt/conda, python 3.9, scikit-image 0.18.3;
the reference imports it unpinned at condinst_head.py:8 and calls color.rgb2lab on a uint8 image at :1413).

Run with /opt/conda/bin/python3.9 (the system python has no skimage):
  * compares the C oracle with skimage.color.rgb2lab over ALL 2^24 uint8 RGB triples, after the reference's cast to
    float32 (:1415-1416), and writes the tally to tests/golden/lab_skimage_exhaustive.json;
  * writes tests/golden/lab_skimage.npz: 65 536 seeded random triples + the grey axis + the six primaries with skimage's
    float32 result, which the test-suite (any python, no skimage) checks the oracle -- and, on the GPU, the kernels -- against.
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import skimage
from skimage import color

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
so = os.path.join(ROOT, 'oracle', '_build', 'libboxinst_oracle.so')
lib = C.CDLL(so)
lib.bxo_rgb2lab_u8.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]


def oracle_lab(rgb_u8):          # [n,3] uint8 -> [n,3] f32 (the C oracle works on planar [3,n])
    n = rgb_u8.shape[0]
    planar = np.ascontiguousarray(rgb_u8.T)
    out = np.empty((3, n), np.float32)
    lib.bxo_rgb2lab_u8(planar.ctypes.data, n, out.ctypes.data)
    return out.T


def skimage_lab(rgb_u8):         # exactly the reference's call: uint8 [h,w,3] -> float64 -> .float()
    return color.rgb2lab(rgb_u8.reshape(1, -1, 3)).astype(np.float32).reshape(-1, 3)


tally = dict(skimage=skimage.__version__, numpy=np.__version__, inputs=0, f32_mismatches=0, max_abs_diff=0.0, max_ulp=0)
g = np.arange(256, dtype=np.uint8)
for r in range(256):
    rgb = np.stack([np.full(65536, r, np.uint8), np.repeat(g, 256), np.tile(g, 256)], axis=1)
    a, b = oracle_lab(rgb), skimage_lab(rgb)
    bad = a != b
    tally['inputs'] += rgb.shape[0]
    tally['f32_mismatches'] += int(bad.any(axis=1).sum())
    if bad.any():
        tally['max_abs_diff'] = max(tally['max_abs_diff'], float(np.abs(a - b).max()))
        ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
        tally['max_ulp'] = max(tally['max_ulp'], int(ulp[bad].max()))
    if r % 32 == 0:
        print(r, tally, file=sys.stderr)
json.dump(tally, open(os.path.join(HERE, 'lab_skimage_exhaustive.json'), 'w'), indent=1)
print(tally)

rng = np.random.default_rng(2024)
sample = np.concatenate([rng.integers(0, 256, (65536, 3), dtype=np.uint8), np.stack([g, g, g], axis=1),
                         np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0], [0, 255, 255], [255, 0, 255],
                                   [10, 200, 77]], np.uint8)])
np.savez_compressed(os.path.join(HERE, 'lab_skimage.npz'), rgb=sample, lab=skimage_lab(sample),
                    skimage_version=np.array(skimage.__version__))
