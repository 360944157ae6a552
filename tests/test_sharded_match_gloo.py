"""world_size-2 CPU (gloo) test of the sharded-match host logic: the result of
query-sharding + reference all-gather must equal the single-process result."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _worker(rank, world, port, nq, nr, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sift_amd
    from sift_amd.sharded import shard_bounds, sharded_match
    from tests.oracle_lib import Oracle

    orc = Oracle()
    q, r = sift_amd.planted_descriptors(nq, nr, plant=0.2)
    qb, qe = shard_bounds(nq, world, rank)
    rb, re = shard_bounds(nr, world, rank)

    def cpu_match(qt, rt, ratio, form):  # the checker stands in for the GPU kernel on CPU
        b, d1, d2 = orc.match_u8(qt.numpy(), rt.numpy(), ratio, form)
        return torch.from_numpy(b), torch.from_numpy(d1), torch.from_numpy(d2)

    best, d1, d2 = sharded_match(cpu_match, torch.from_numpy(q[qb:qe]), torch.from_numpy(r[rb:re]), 0.5, 0)
    np.savez(os.path.join(outdir, f"r{rank}.npz"), best=best.numpy(), d1=d1.numpy(), d2=d2.numpy(), qb=qb, qe=qe)
    dist.destroy_process_group()


def _run(world, nq, nr, tmp_path, port):
    mp.spawn(_worker, args=(world, port, nq, nr, str(tmp_path)), nprocs=world, join=True)
    import sift_amd
    from tests.oracle_lib import Oracle

    q, r = sift_amd.planted_descriptors(nq, nr, plant=0.2)
    bo, d1o, d2o = Oracle().match_u8(q, r, 0.5, 0)
    got = np.full(nq, -2, np.int32)
    for rank in range(world):
        z = np.load(os.path.join(str(tmp_path), f"r{rank}.npz"))
        got[int(z["qb"]): int(z["qe"])] = z["best"]
        assert np.array_equal(z["d1"], d1o[int(z["qb"]): int(z["qe"])])
    assert np.array_equal(got, bo)
    assert (got >= 0).sum() > 0


def test_sharded_match_even(tmp_path):
    _run(2, 200, 300, tmp_path, 29611)


def test_sharded_match_ragged(tmp_path):
    _run(2, 201, 301, tmp_path, 29612)  # shards of different length: padded all-gather path


def test_shard_bounds_cover():
    from sift_amd.sharded import shard_bounds

    for n in (0, 1, 7, 8, 1000003):
        for w in (1, 2, 3, 8):
            segs = [shard_bounds(n, w, r) for r in range(w)]
            assert segs[0][0] == 0 and segs[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(segs, segs[1:]))
            assert max(e - b for b, e in segs) - min(e - b for b, e in segs) <= 1


def _frames_worker(rank, world, port, nframes, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sift_amd.sharded import frames_of_rank, gather_feature_counts

    mine = list(frames_of_rank(nframes, world, rank))
    counts = torch.tensor([100 + 3 * k for k in mine], dtype=torch.int32)  # stands for the extract of frame k
    allc = gather_feature_counts(counts, nframes)
    np.save(os.path.join(outdir, f"c{rank}.npy"), allc.numpy())
    dist.destroy_process_group()


def test_frames_shard_round_robin_and_counts_gather(tmp_path):
    from sift_amd.sharded import frames_of_rank

    for n, w in ((1024, 8), (10, 4), (3, 8), (7, 2)):
        seen = sorted(k for r in range(w) for k in frames_of_rank(n, w, r))
        assert seen == list(range(n))
        assert all(k % w == r for r in range(w) for k in frames_of_rank(n, w, r))
    mp.spawn(_frames_worker, args=(2, 29533, 7, str(tmp_path)), nprocs=2, join=True)
    want = np.array([100 + 3 * k for k in range(7)], np.int32)
    for rank in range(2):
        assert np.array_equal(np.load(os.path.join(str(tmp_path), f"c{rank}.npy")), want)


def test_ragged_offsets_of_the_c_abi_match_the_concatenation():
    """sift_hip_ragged_offsets (host arithmetic of sift_hip_match_u8_sharded_ragged): rank k's
    reference i sits at offsets[k] + i of the rank-ordered concatenation without padding, which is
    the order sift_amd.sharded.all_gather_refs builds."""
    import ctypes as C

    import numpy as np

    import sift_amd
    from sift_amd import capi

    lib = sift_amd.load_library()
    rng = np.random.default_rng(4)
    for nranks in (1, 2, 3, 8):
        counts = rng.integers(0, 50, nranks).astype(np.int32)
        off = np.zeros(nranks + 1, np.int64)
        largest = lib.sift_hip_ragged_offsets(counts.ctypes.data_as(C.c_void_p), nranks, off.ctypes.data_as(C.c_void_p))
        assert largest == int(counts.max())
        # the padded gather, squeezed: what the device copies do
        shards = [np.full((int(c), 1), k, np.int32) + 1000 * np.arange(int(c))[:, None] for k, c in enumerate(counts)]
        padded = np.zeros((nranks, max(largest, 1), 1), np.int32) - 1
        for k, sh in enumerate(shards):
            padded[k, : len(sh)] = sh
        squeezed = np.full((int(off[-1]), 1), -7, np.int32)
        for k in range(nranks):
            squeezed[off[k]: off[k] + counts[k]] = padded[k, : counts[k]]
        assert np.array_equal(squeezed, np.concatenate(shards) if off[-1] else squeezed[:0])
        assert off[-1] == counts.sum()
    bad = np.array([3, -1], np.int32)
    off = np.zeros(3, np.int64)
    assert lib.sift_hip_ragged_offsets(bad.ctypes.data_as(C.c_void_p), 2, off.ctypes.data_as(C.c_void_p)) == capi.ERR_ARG
    huge = np.array([2 ** 31 - 1, 5], np.int32)
    assert lib.sift_hip_ragged_offsets(huge.ctypes.data_as(C.c_void_p), 2, off.ctypes.data_as(C.c_void_p)) == capi.ERR_LIMIT
