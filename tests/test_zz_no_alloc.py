"""sift_hip_match_u8_noalloc (the matcher for callers that cannot allocate) against
sift_hip_match_u8_dev, through the C ABI, bit for bit, on two calls in a row -- the
second with smaller inputs, so that whatever the first call left in the library's
static work area would show if the second looked at it."""
import numpy as np
import pytest

import sift_amd
from sift_amd import capi

pytestmark = pytest.mark.gpu


def run_both(ctx, torch, q, r, ratio, form):
    dev = torch.device("cuda", 0)
    d_q = torch.from_numpy(q).to(dev)
    d_r = torch.from_numpy(r).to(dev)
    nq, nr = len(q), len(r)
    outs = []
    for fn in (ctx.match_u8_dev, ctx.match_u8_noalloc):
        best = torch.full((nq,), -7, dtype=torch.int32, device=dev)
        d1 = torch.full((nq,), -7, dtype=torch.int32, device=dev)
        d2 = torch.full((nq,), -7, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        fn(d_q, nq, d_r, nr, ratio, form, best, d1, d2)
        ctx.synchronize()
        outs.append((best.cpu().numpy(), d1.cpu().numpy(), d2.cpu().numpy()))
    return outs


def test_noalloc_match_equals_allocating_match_twice():
    import torch

    ctx = sift_amd.SiftHip(0)
    try:
        # first call: sizes that are not multiples of any tile; second: smaller on both sides
        for nq, nr, seed, form, ratio in ((3001, 5003, 1, 0, 0.5), (257, 300, 2, 1, 0.8), (1, 1, 3, 0, 0.5)):
            q, r = sift_amd.planted_descriptors(nq, nr, seed_q=seed, seed_r=seed + 40, plant=0.2)
            (b0, d10, d20), (b1, d11, d21) = run_both(ctx, torch, q, r, ratio, form)
            assert (b0 >= 0).sum() > 0 or nq == 1
            assert np.array_equal(b0, b1) and np.array_equal(d10, d11) and np.array_equal(d20, d21), (nq, nr)
            if nq == 257:  # and once against the CPU restatement of forceMatchFeat directly
                from tests.oracle_lib import Oracle

                bo, d1o, d2o = Oracle().match_u8(q, r, ratio, form)
                assert np.array_equal(b1, bo) and np.array_equal(d11, d1o) and np.array_equal(d21, d2o)
    finally:
        ctx.close()


def test_noalloc_match_rejects_sizes_beyond_its_work_area():
    import torch

    ctx = sift_amd.SiftHip(0)
    try:
        dev = torch.device("cuda", 0)
        big = 65536 + 1
        d = torch.zeros((big, 128), dtype=torch.uint8, device=dev)
        out = torch.full((big,), -7, dtype=torch.int32, device=dev)
        for nq, nr in ((big, 16), (16, big)):
            rc = ctx.lib.sift_hip_match_u8_noalloc(ctx.h, capi._ptr(d), nq, capi._ptr(d), nr, 0.5, 0, capi._ptr(out), None,
                                                   None)
            assert rc == capi.ERR_LIMIT
            assert b"65536" in ctx.lib.sift_hip_last_error(ctx.h)
        torch.cuda.synchronize()
        assert int((out != -7).sum().item()) == 0  # nothing was launched
        # the largest accepted size runs and agrees with the allocating entry point on a slice
        q, r = sift_amd.planted_descriptors(512, 65536, seed_q=9, seed_r=10, plant=0.5)
        (b0, d10, d20), (b1, d11, d21) = run_both(ctx, torch, q, r, 0.5, 0)
        assert np.array_equal(b0, b1) and np.array_equal(d10, d11) and np.array_equal(d20, d21)
    finally:
        ctx.close()
