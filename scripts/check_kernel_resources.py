#!/usr/bin/env python3
"""Build check: register spills and scratch of the gfx950 kernels in libsoslam_ba.so.

The kernels of the hot path are written against a register budget (ba_schur10: 126 of the 128 VGPRs that two eight-wave
workgroups per CU leave each wave).  A compiler update, or an innocent edit, that pushes one of them over its budget must
fail the BUILD - in round 2 a variant of ba_schur10 that spilled was only noticed as a fault on the device.  This script
extracts the code objects (llvm-objdump --offloading), reads every kernel's metadata (llvm-readelf -n) and fails when a
kernel outside ALLOWED_SCRATCH uses scratch memory or spills vector registers.

usage: check_kernel_resources.py <libsoslam_ba.so>   (exit status 1 on a violation)"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
# kernels that are known to use a few bytes of scratch and are tested that way on the device in every run (off the 1 M-observation
# hot path: the one-launch structure-only step of the per-frame call and the resident structure-only solve - whose lane-0 controller
# is called, not inlined, and so owns a stack -, the sequential band factorisation for bands of 11..15)
ALLOWED_SCRATCH = ("ba_points_step_kernel", "ba_points_solve_kernel", "band_cholesky_kernel")


def kernels(lib):
    tmp = tempfile.mkdtemp(prefix="soslam_co_")
    try:
        shutil.copy(lib, os.path.join(tmp, "lib.so"))
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", "lib.so"], cwd=tmp, check=True, capture_output=True)
        out = []
        for f in sorted(os.listdir(tmp)):
            if "gfx950" not in f:
                continue
            txt = subprocess.run([f"{LLVM}/llvm-readelf", "-n", f], cwd=tmp, check=True, capture_output=True, text=True).stdout
            for blk in re.split(r"\n\s+- \.agpr_count:", txt)[1:]:
                name = re.search(r"\.name:\s+(\S+)", blk)
                if not name:
                    continue
                get = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
                out.append({"name": name.group(1), "scratch": get("private_segment_fixed_size"), "vgpr_spill": get("vgpr_spill_count"),
                            "sgpr_spill": get("sgpr_spill_count"), "vgpr": get("vgpr_count")})
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def short(mangled):
    m = re.search(r"\d+([a-z][a-z0-9_]*_kernel)(ILi(\d+)E)?", mangled)
    return (m.group(1) + (f"<{m.group(3)}>" if m.group(3) else "")) if m else mangled[:60]


def main():
    ks = kernels(sys.argv[1])
    if not ks:
        print("check_kernel_resources: no gfx950 kernels found in", sys.argv[1])
        return 1
    bad = [k for k in ks if (k["scratch"] or k["vgpr_spill"]) and not any(a in k["name"] for a in ALLOWED_SCRATCH)]
    for k in ks:
        if k["scratch"] or k["vgpr_spill"]:
            print(f"  {short(k['name']):32s} scratch {k['scratch']:4d} B/lane, {k['vgpr_spill']:3d} VGPRs spilled, {k['vgpr']} VGPRs"
                  f"{'' if k in bad else '  (allowed)'}")
    if bad:
        print(f"check_kernel_resources: {len(bad)} kernel(s) spill vector registers or use scratch memory: over their register budget")
        return 1
    print(f"check_kernel_resources: {len(ks)} kernels, none over its register budget")
    return 0


if __name__ == "__main__":
    sys.exit(main())
