"""MI355X-native SIFT detect / describe / match path.

The compute lives in ``sift_amd/lib/libsift_hip.so`` (hand-written HIP for
gfx950, C ABI in ``include/sift_hip.h``).  This package is the thin Python
binding used by the tests and the benchmark; the C++ drop-in classes are in
``sift_amd/host``.
"""
from .capi import (  # noqa: F401
    SiftHip,
    SiftHipError,
    SiftParams,
    FEAT_DTYPE,
    KERNEL_CLASSES,
    gau_matrix,
    library_path,
    load_library,
    pyramid_octaves,
)
from .synth import blob_image, planted_descriptors, textured_image  # noqa: F401
