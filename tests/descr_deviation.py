#!/usr/bin/env python3
"""Largest deviation of the GPU descriptors from the CPU oracle's on one image (the
parity tests bound it; this prints the actual figures)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift_amd  # noqa: E402
from tests.oracle_lib import Oracle  # noqa: E402

img = sift_amd.blob_image(1280, 720, seed=1)
ctx = sift_amd.SiftHip(0)
f, h = ctx.extract(img, sift_amd.SiftParams(octaves=4), want_hist=True)
fo, ho = Oracle().extract(img, octaves=4)
assert len(f) == len(fo)
l2 = np.sqrt(((h.astype(np.float64) - ho) ** 2).sum(axis=1))
diff = f["descr"].astype(np.int16) - fo["descr"].astype(np.int16)
print(f"features {len(f)}  max L2 of clamped histogram {l2.max():.3e}  mean {l2.mean():.3e}  "
      f"bytes differing {int((diff != 0).sum())} of {diff.size} (max |diff| {int(np.abs(diff).max())})")
