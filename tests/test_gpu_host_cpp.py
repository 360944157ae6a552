"""GPU test of the C++ drop-in classes (sift_amd/host): Sift::pyramid() and
Stitching::Check() driven the way the reference's StitchImg drives them."""
import json
import os
import subprocess

import numpy as np
import pytest

import sift_amd

import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import ransac_oracle  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "sift_amd", "host", "selftest")


def run_selftest(tmp_path, a, b, th):
    h, w = a.shape[:2]
    ch = 1 if a.ndim == 2 else a.shape[2]
    inp, outp = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(inp, "wb") as f:
        f.write(np.array([w, h, ch], np.int32).tobytes())
        f.write(np.ascontiguousarray(a, np.float32).tobytes())
        f.write(np.ascontiguousarray(b, np.float32).tobytes())
    res = subprocess.run([EXE, str(inp), str(outp), str(th)], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    info = json.loads(res.stdout.strip().splitlines()[-1])
    raw = open(outp, "rb").read()
    n1, n2 = np.frombuffer(raw[:8], np.int32)
    f1 = np.frombuffer(raw[8: 8 + 156 * n1], sift_amd.FEAT_DTYPE)
    f2 = np.frombuffer(raw[8 + 156 * n1: 8 + 156 * (n1 + n2)], sift_amd.FEAT_DTYPE)
    m = np.frombuffer(raw[8 + 156 * (n1 + n2):], np.int32)
    return info, f1, f2, m


def test_cpp_classes_equal_c_abi_and_oracle(tmp_path, sift, oracle):
    assert os.path.exists(EXE), "build with __graft_entry__.build()"
    a = sift_amd.blob_image(256, 192, seed=1)
    b = sift_amd.blob_image(256, 192, seed=1, shift_x=-8)
    info, f1, f2, m = run_selftest(tmp_path, a, b, 0.5)
    assert info["file_same"] is True
    assert info["threads_equal"] is True  # two Sift objects in two threads share the context under its lock
    fo, _ = oracle.extract(a)
    assert len(f1) == len(fo) and np.array_equal(f1["x"], fo["x"]) and np.array_equal(f1["sita"], fo["sita"])
    fa, _ = sift.extract(a)
    fb, _ = sift.extract(b)
    assert np.array_equal(f1["descr"], fa["descr"]) and np.array_equal(f2["descr"], fb["descr"])
    # Check(): queries = image 2, references = image 1 (Sift.cpp:674), live acceptance form
    bo, _, _ = oracle.match_u8(fb["descr"], fa["descr"], 0.5, 0)
    assert np.array_equal(m, bo)
    assert info["matched"] == int((bo >= 0).sum()) > 20
    # ransac_xform over those matches (Sift.cpp:684): same libc random stream as the numpy
    # restatement -> same consensus; image 2 is image 1 moved 8 px, so H is that translation
    sel = np.nonzero(m >= 0)[0]
    pts = np.stack([f2["x"][sel] / f2["size"][sel], f2["y"][sel] / f2["size"][sel]], axis=1).astype(np.float32)
    mp = f1[m[sel]]
    mpts = np.stack([mp["x"] / mp["size"], mp["y"] / mp["size"]], axis=1).astype(np.float32)
    Ho, fin, in_min = ransac_oracle.ransac_xform(pts, mpts, 4, 0.005, 3.0, 1234)
    assert info["ransac_equal"] is True  # GPU (inside Check) == CPU ransac_xform
    # matching the descriptors kept on the device == packing Feature[] and uploading it; an edited Feature[] is honoured
    assert info["match_paths_equal"] is True
    assert Ho is not None and info["ransac_num"] == len(fin) >= in_min
    assert info["ransac_feat"] == [int(v) for v in sel[fin]]
    H = np.array(info["homog"])
    np.testing.assert_allclose(H, Ho, rtol=1e-3, atol=1e-5)
    cx, cy = 128.0, 96.0
    den = H[6] * cx + H[7] * cy + H[8]
    moved = np.array([(H[0] * cx + H[1] * cy + H[2]) / den - cx, (H[3] * cx + H[4] * cy + H[5]) / den - cy])
    assert abs(abs(moved[0]) - 8.0) < 1.0 and abs(moved[1]) < 1.0
    err = ransac_oracle.xfer_err(H.astype(np.float32), pts[fin], mpts[fin])
    assert err.max() <= 3.0
    # options().subpixel = 1: Feature::img_pt = (x + xc, y + xr) / size (matrix.hpp:188-189)
    fs, _ = sift.extract(a, sift_amd.SiftParams(subpixel=1))
    sub = sift.last_offsets(len(fs))
    assert info["n_sub"] == len(fs) > 5
    px = ((fs["x"] + sub[:, 0]) / fs["size"]).astype(np.float32).astype(np.float64)
    py = ((fs["y"] + sub[:, 1]) / fs["size"]).astype(np.float32).astype(np.float64)
    assert abs(info["sub_sum_x"] - px.sum()) < 1e-2 and abs(info["sub_sum_y"] - py.sum()) < 1e-2


def test_cpp_classes_rgb_input(tmp_path, oracle):
    g = sift_amd.blob_image(192, 128, seed=4)
    rgb = np.stack([g, g * np.float32(0.9), g * np.float32(0.8)], axis=-1)
    info, f1, f2, m = run_selftest(tmp_path, rgb, rgb, 0.5)
    fo, _ = oracle.extract(oracle.convert_gray(rgb))
    assert len(f1) == len(fo) and np.array_equal(f1["y"], fo["y"])
    # identical images: every feature's nearest neighbour is itself at distance 0
    assert info["n1"] == info["n2"]


def test_plumbing_config_rgb_pair_640x480(tmp_path, sift, oracle):
    """The reference's own use (Sift_main.cpp:31-55) on a synthetic RGB pair, second image =
    first moved 40 px: Sift -> Sift -> Stitching::Check(0.5) through the C++ classes; the match
    list equals the oracle's exact search and RANSAC returns that translation."""
    g1 = sift_amd.blob_image(640, 480, seed=7)
    g2 = sift_amd.blob_image(640, 480, seed=7, shift_x=-40)
    rgb = lambda g: np.stack([g, g * np.float32(0.95), g * np.float32(0.9)], axis=-1)  # noqa: E731
    info, f1, f2, m = run_selftest(tmp_path, rgb(g1), rgb(g2), 0.5)
    fo1, _ = oracle.extract(oracle.convert_gray(rgb(g1)))
    fo2, _ = oracle.extract(oracle.convert_gray(rgb(g2)))
    assert len(f1) == len(fo1) and len(f2) == len(fo2)
    for k in ("x", "y", "kai", "sita"):
        assert np.array_equal(f1[k], fo1[k]) and np.array_equal(f2[k], fo2[k])
    bo, _, _ = oracle.match_u8(f2["descr"], f1["descr"], 0.5, 0)
    assert np.array_equal(m, bo) and (bo >= 0).sum() > 100
    assert info["ransac_equal"] is True and info["ransac_num"] > 0.8 * (bo >= 0).sum()
    H = np.array(info["homog"])
    for cx, cy in ((100.0, 100.0), (320.0, 240.0), (600.0, 400.0)):
        den = H[6] * cx + H[7] * cy + H[8]
        mx, my = (H[0] * cx + H[1] * cy + H[2]) / den - cx, (H[3] * cx + H[4] * cy + H[5]) / den - cy
        assert abs(abs(mx) - 40.0) < 1.0 and abs(my) < 1.0
    # the stitcher's warps through the C++ mirrors (_WarpPerspective, warping) = oracle, byte for byte
    raw = open(str(tmp_path / "out.bin") + ".warp", "rb").read()
    pos = 0

    def take_img():
        nonlocal pos
        c, r = np.frombuffer(raw[pos:pos + 8], np.int32)
        img = np.frombuffer(raw[pos + 8:pos + 8 + c * r * 3], np.uint8).reshape(r, c, 3)
        pos += 8 + c * r * 3
        return img

    def take_pts():
        nonlocal pos
        n = int(np.frombuffer(raw[pos:pos + 4], np.int32)[0])
        pts = np.frombuffer(raw[pos + 4:pos + 4 + 8 * n], np.float32).reshape(n, 2)
        pos += 4 + 8 * n
        return pts

    to_u8 = lambda a: (a.astype(np.float32).astype(np.float64) * 255.0).astype(np.uint8)  # noqa: E731  Imgraw.hpp:59-70
    u1, u2 = to_u8(rgb(g1)), to_u8(rgb(g2))
    canvas = take_img()
    assert canvas.shape == (int(480 * 1.1), int(640 * 1.6), 3)
    want = oracle.warp_perspective(u2, np.array(info["homog"], np.float32), np.zeros_like(canvas))
    assert np.array_equal(canvas, want) and canvas.any()
    up_all, down_all = [], []
    for u in (u1, u2):
        w_img, w_up, w_down = oracle.warp_cylinder(u, 700.0)
        assert np.array_equal(take_img(), w_img)
        up_all.append(w_up)
        down_all.append(w_down)
    assert np.array_equal(take_pts(), np.concatenate(up_all)) and np.array_equal(take_pts(), np.concatenate(down_all))
    # ... and Check() went on to blen2img: the panorama equals the restated pipeline (focal length from
    # the homography, cylindrical projection, alignment of the consensus matches, multi-band blend,
    # canvas assembly), value for value
    Hf = np.array(info["homog"], np.float32)
    focal = oracle.get_focal(Hf, 640 * 480 * 3, 640 * 480 * 3)
    assert np.float32(info["focal"]) == np.float32(focal)
    cons = np.array(info["ransac_feat"], np.int64)
    a_xy = np.stack([f2["x"][cons] / f2["size"][cons], f2["y"][cons] / f2["size"][cons]], axis=1).astype(np.float32)
    p1 = f1[m[cons]]
    b_xy = np.stack([p1["x"] / p1["size"], p1["y"] / p1["size"]], axis=1).astype(np.float32)
    dx, dy = oracle.align_match((640, 480), (640, 480), a_xy, b_xy, focal)
    c1, _, _ = oracle.warp_cylinder(u1, focal)
    c2, _, _ = oracle.warp_cylinder(u2, focal)
    want_canvas, _, _ = oracle.blend_compose(c1, c2, dx, dy)
    praw = open(str(tmp_path / "out.bin") + ".pano", "rb").read()
    pc, pr = np.frombuffer(praw[:8], np.int32)
    pano = np.frombuffer(praw[8:], np.float32).reshape(pr, pc, 3)
    print("panorama %d x %d, offsets (%d, %d), focal %.1f" % (pc, pr, dx, dy, focal))
    assert info["pano"] == [int(pc), int(pr)] and (pr, pc) == want_canvas.shape[:2]
    assert np.array_equal(pano, (want_canvas.astype(np.float64) / 255.0).astype(np.float32))
    assert want_canvas.any() and abs(dx - (640 - 40)) <= 3  # image 2 = image 1 moved 40 px: 600 columns in common
