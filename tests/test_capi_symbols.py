"""CPU tests of the C ABI library: it loads, exports every symbol the header
declares, and its pure-host entry points behave (no kernel is launched here)."""
import ctypes as C
import os
import re

import numpy as np

import sift_amd
from sift_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sift_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sift_hip_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = sift_amd.load_library()
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(lib._protos), set(names) ^ set(lib._protos)


def test_feat_struct_layout():
    assert capi.FEAT_DTYPE.itemsize == 156
    assert capi.FEAT_DTYPE.fields["descr"][1] == 28


def test_default_params_are_the_reference_macros():
    lib = sift_amd.load_library()
    p = capi._Params()
    lib.sift_hip_default_params(C.byref(p))
    assert (p.octaves, p.intvls, p.subpixel, p.blur_mode) == (0, 3, 0, 0)
    assert np.float32(p.sigma0) == np.float32(1.6) and np.float32(p.dx_thr) == np.float32(0.03)


def test_host_entry_points():
    lib = sift_amd.load_library()
    assert sift_amd.pyramid_octaves(3840, 2160) == 7
    assert sift_amd.pyramid_octaves(0, 5) == capi.ERR_ARG
    g = sift_amd.gau_matrix(1.6)
    assert len(g) == 11 and abs(float(g.sum()) - 1.0) < 1e-6 and np.array_equal(g, g[::-1])
    assert lib.sift_hip_strerror(capi.ERR_NOSPC) == b"output capacity too small"
    assert lib.sift_hip_kernel_class_name(1) == b"blur_row"
    buf = np.zeros(4, np.float32)
    assert lib.sift_hip_gau_matrix(1.6, 0, buf.ctypes.data_as(C.POINTER(C.c_float)), 4) == -11


def test_library_taps_equal_golden_reference_taps():
    z = np.load(os.path.join(ROOT, "tests", "golden", "imglib_ref.npz"))
    for i, (p, ml) in enumerate(zip(z["tap_p"], z["tap_len"])):
        want = z["taps"][z["tap_off"][i]: z["tap_off"][i + 1]]
        got = sift_amd.gau_matrix(float(p), int(ml))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (p, ml)


def test_null_context_is_rejected():
    lib = sift_amd.load_library()
    n = C.c_int(0)
    assert lib.sift_hip_extract(None, None, 4, 4, 1, None, None, 0, C.byref(n), None) == capi.ERR_ARG
    assert lib.sift_hip_synchronize(None) == capi.ERR_ARG


def test_synthetic_inputs_are_deterministic():
    a = sift_amd.blob_image(96, 64, seed=7, as_u8=True)
    z = np.load(os.path.join(ROOT, "tests", "golden", "oracle_extract.npz"))
    assert np.array_equal(sift_amd.blob_image(256, 192, seed=1, as_u8=True), z["img_a"])
    assert a.shape == (64, 96) and a.min() == 0 and a.max() == 255
    q, r = sift_amd.planted_descriptors(64, 32)
    q2, r2 = sift_amd.planted_descriptors(64, 32)
    assert np.array_equal(q, q2) and np.array_equal(r, r2) and q.shape == (64, 128)
