#!/usr/bin/env python3
"""The reference's own use (Sift_main.cpp:31-55: two images -> Sift::pyramid x 2 -> Stitching::Check) through the
C++ drop-in classes on a synthetic RGB pair, second image = first moved by a twelfth of its width:
    python tools/check_pipeline.py [--width 640 --height 480]
prints the self-test's wall times (two pyramids; Check = match + RANSAC + blen2img) and what it found."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sift_amd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--height", type=int, default=480)
a = ap.parse_args()
w, h = a.width, a.height
shift = w // 12
g1 = sift_amd.blob_image(w, h, seed=7)
g2 = sift_amd.blob_image(w, h, seed=7, shift_x=-shift)
rgb = lambda g: np.stack([g, g * np.float32(0.95), g * np.float32(0.9)], axis=-1)  # noqa: E731
with tempfile.TemporaryDirectory() as d:
    inp, outp = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
    with open(inp, "wb") as f:
        f.write(np.array([w, h, 3], np.int32).tobytes())
        f.write(np.ascontiguousarray(rgb(g1), np.float32).tobytes())
        f.write(np.ascontiguousarray(rgb(g2), np.float32).tobytes())
    res = subprocess.run([os.path.join(ROOT, "sift_amd", "host", "selftest"), inp, outp, "0.5"], capture_output=True,
                         text=True, timeout=600)
    if res.returncode:
        sys.exit(res.stderr)
    if os.environ.get("SIFT_HOST_TIMING") == "1":
        sys.stderr.write(res.stderr)  # per-step wall times of the host classes
    info = json.loads(res.stdout.strip().splitlines()[-1])
print("%dx%d RGB pair: two pyramids %.2f ms, Check %.2f ms; %d / %d features, %d matches, consensus %d, panorama %s" % (
    w, h, info["ms_two_pyramids"], info["ms_check"], info["n1"], info["n2"], info["matched"], info["ransac_num"], info["pano"]))
