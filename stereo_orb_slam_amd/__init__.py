"""MI355X-native bundle-adjustment / pose-graph backend for the stereo_orb_slam hot path.

The product is ``libsoslam_ba.so`` (hand-written HIP kernels for gfx950 + a C++ host solver behind the C ABI
of ``include/soslam_ba.h`` / ``include/soslam_pg.h``) and the C++ host shim in ``stereo_orb_slam_amd/host/``
that keeps the reference's ``BundleAdjuster`` / ``PoseGraphOptimizer`` class API.  The Python modules here
are plumbing for tests and ``bench.py``.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
