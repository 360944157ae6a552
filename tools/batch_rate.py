#!/usr/bin/env python3
"""Frames/s of sift_hip_extract_batch_dev on a batch of 1080p frames resident in HBM:
    python tools/batch_rate.py [--frames 32 --width 1920 --height 1080]
Sharded over the GPUs of a node (frame k -> rank k mod N, no collective on the data path):
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/batch_rate.py --frames 1024
(--backend gloo with SIFT_BENCH_ONE_GPU=1 rehearses the multi-rank logic on one GPU.)"""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import sift_amd  # noqa: E402
from sift_amd import SiftParams  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=32)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--octaves", type=int, default=4)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--cap", type=int, default=1 << 14, help="feature capacity per frame")
ap.add_argument("--backend", default="nccl", choices=("nccl", "gloo"))
a = ap.parse_args()
import torch.distributed as dist  # noqa: E402

from sift_amd.sharded import frames_of_rank, gather_feature_counts  # noqa: E402

rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
local = 0 if os.environ.get("SIFT_BENCH_ONE_GPU") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if a.backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
ctx = sift_amd.SiftHip(local)
mine = list(frames_of_rank(a.frames, world, rank))
total_frames = a.frames
a.frames = len(mine)
base = [sift_amd.blob_image(a.width, a.height, seed=40 + k) for k in range(4)]
frames = np.stack([base[k % 4] for k in mine])
d = torch.from_numpy(frames).cuda()
cap = a.cap
out = ctx.pinned_array(a.frames * cap, sift_amd.FEAT_DTYPE).reshape(a.frames, cap)
n = np.zeros(a.frames, np.int32)
prm = SiftParams(octaves=a.octaves)._c()


def run():
    rc = ctx.lib.sift_hip_extract_batch_dev(ctx.h, C.c_void_p(d.data_ptr()), a.width, a.height, a.frames, C.byref(prm),
                                            C.c_void_p(out.ctypes.data), cap, n.ctypes.data_as(C.POINTER(C.c_int)))
    assert rc == 0, rc


def barrier():
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()


run()
run()
best = 1e9
for _ in range(a.reps):
    barrier()
    t = time.perf_counter()
    run()
    barrier()
    dt = time.perf_counter() - t
    if world > 1:  # the slowest rank's time
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda" if a.backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    best = min(best, dt)
cnt_dev = torch.from_numpy(n).cuda() if (world > 1 and a.backend == "nccl") else torch.from_numpy(n)
counts = gather_feature_counts(cnt_dev, total_frames)
if rank == 0:
    print(f"{total_frames} frames {a.width}x{a.height} on {world} rank(s): {best / total_frames * 1e3:.3f} ms/frame, "
          f"{total_frames / best:.1f} frames/s, features/frame {counts.float().mean():.0f}")
if world > 1:
    dist.destroy_process_group()
