"""CPU suite: every build macro of the search kernel that is kept as a measurement aid still COMPILES for gfx950 (device
pass only, no GPU needed).  The GPU suite runs the default build; a knob that nobody builds rots -- round 5's
-DFXJPS_KN=32 had a null-pointer read nobody had seen.  (DESIGN.md section 8b lists what each of them is for.)"""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fuxi-planner_amd", "csrc", "fxjps.hip")
HIPCC = "/opt/rocm/bin/hipcc"

KNOBS = [
    ["-DFXJPS_KN=32"], ["-DFXJPS_KN=8"], ["-DFXJPS_XCC=1"], ["-DFXJPS_PROF"], ["-DFXJPS_PROF", "-DFXJPS_PROF_LIGHT"],
    ["-DFXJPS_PHASE_S=7", "-DFXJPS_PHASE_E=2"], ["-DFXJPS_MARK"], ["-DFXJPS_HWID"],
    ["-DFXJPS_SORT_BITONIC=0"], ["-DFXJPS_R2_HASH=0", "-DFXJPS_ONE_INSERT=0"], ["-DFXJPS_R2_REC=1", "-DFXJPS_RFILL_F32=1"],
    ["-DFXJPS_OCC=5", "-DFXJPS_WPB=5", "-DFXJPS_PTR_SGPR=1", "-DFXJPS_HZ_LOG2=7"],
]


def _compile(flags):
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden",
                        "--cuda-device-only", "-c", "-o", "/dev/null", SRC] + flags, capture_output=True, text=True, timeout=900)
    return flags, r.returncode, r.stderr[-1500:]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_every_kept_build_knob_compiles():
    with ThreadPoolExecutor(max_workers=4) as ex:  # (~ 35 s and ~ 1.5 GB per compile: four at a time fit the build container)
        res = list(ex.map(_compile, KNOBS))
    bad = [(f, err) for f, rc, err in res if rc != 0]
    assert not bad, bad
