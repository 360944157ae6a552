#!/usr/bin/env python3
"""Distribution of the wall time of single 4K extracts (200 frames each) for placements of octave 3 on the
side streams: mean, median, deciles and a histogram around 2 ms.  python tools/frame_hist.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import sift_amd
from sift_amd import SiftParams
ctx = sift_amd.SiftHip(0)
W,H=3840,2160
d = torch.from_numpy(sift_amd.blob_image(W, H, seed=1)).cuda()
out = ctx.pinned_array(1 << 17, sift_amd.FEAT_DTYPE)
prm = SiftParams(octaves=4)
for place in (0, 3, 1):
    ctx.set_tuning("o3_place", place)
    for _ in range(20): ctx.extract_dev(d, W, H, prm, out)
    ts=[]
    for _ in range(200):
        t0=time.perf_counter(); ctx.extract_dev(d, W, H, prm, out); ts.append((time.perf_counter()-t0)*1e3)
    ts=np.array(ts)
    print("place", place, "mean %.3f median %.3f p10 %.3f p90 %.3f min %.3f max %.3f" % (ts.mean(), np.median(ts), np.percentile(ts,10), np.percentile(ts,90), ts.min(), ts.max()))
    print("  hist:", np.histogram(ts, bins=[1.8,1.85,1.9,1.95,1.98,1.99,2.0,2.01,2.02,2.05,2.1,3])[0])
