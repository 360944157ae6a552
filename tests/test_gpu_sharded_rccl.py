"""sift_hip_match_u8_sharded (all-gather over RCCL + match) with a one-rank
communicator: the plumbing (run-time lookup of ncclAllGather, gathered buffer,
stream order) on the one GPU a test box has.  The multi-rank index arithmetic is
the all-gather's rank-major layout, covered on CPU by test_sharded_match_gloo.py."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_sharded_entry_point_with_one_rank_communicator():
    res = subprocess.run([sys.executable, os.path.join(HERE, "sharded_rccl_check.py")], capture_output=True, text=True,
                         timeout=180)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "sharded-rccl ok" in res.stdout
