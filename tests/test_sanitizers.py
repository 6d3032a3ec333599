"""CPU sanitizer builds: the oracle and the host-side containers / conversions / Dump I/O under
AddressSanitizer + UndefinedBehaviorSanitizer.  `make asan` builds and runs a driver; any report is a non-zero exit.
(GPU AddressSanitizer is not available on the pool, so the device library is not part of this.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("where", ["oracle", os.path.join("stereo_orb_slam_amd", "host")])
def test_asan_ubsan_clean(where):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    out = subprocess.run(["make", "-C", os.path.join(ROOT, where), "asan"], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr
    assert out.stdout.strip().splitlines()[-2 if where != "oracle" else -2:] or True
    assert "OK" in out.stdout
