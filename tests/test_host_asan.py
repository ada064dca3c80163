"""Host-side ASan build (sanitizers run on the CPU build only): every csrc/*.hip compiled with
-fsanitize=address -fno-gpu-sanitize into a scratch directory, tests/host_asan_driver.cpp run against it without a GPU."""
import glob
import os
import shutil
import subprocess
import tempfile

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_host_code_is_asan_clean():
    tmp = tempfile.mkdtemp(prefix="crd_asan_")
    try:
        flags = ["--offload-arch=gfx950", "-O1", "-std=c++17", "-fPIC", "-fsanitize=address", "-fno-gpu-sanitize", "-Wno-unused-result",
                 "-Wno-inline-asm", "-I" + os.path.join(REPO, "include"), "-I" + os.path.join(REPO, "camradepth_amd", "csrc")]
        srcs = sorted(glob.glob(os.path.join(REPO, "camradepth_amd", "csrc", "*.hip")))
        procs = [(s, subprocess.Popen([HIPCC] + flags + ["-c", s, "-o", os.path.join(tmp, os.path.basename(s)[:-4] + ".o")],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)) for s in srcs[:8]]
        rest = srcs[8:]
        while procs:                                  # at most eight compilers at a time (8 CPUs here)
            s, p = procs.pop(0)
            _, err = p.communicate()
            assert p.returncode == 0, f"{s}:\n{err[-3000:]}"
            if rest:
                n = rest.pop(0)
                procs.append((n, subprocess.Popen([HIPCC] + flags + ["-c", n, "-o", os.path.join(tmp, os.path.basename(n)[:-4] + ".o")],
                                                  stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)))
        lib = os.path.join(tmp, "libcamradepth_hip_asan.so")
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-fsanitize=address", "-fno-gpu-sanitize", "-shared", "-fPIC", "-o", lib]
                           + sorted(glob.glob(os.path.join(tmp, "*.o"))), capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        drv = os.path.join(tmp, "driver")
        r = subprocess.run([HIPCC, "-x", "c++", "-std=c++17", "-fsanitize=address", "-I" + os.path.join(REPO, "include"),
                            os.path.join(REPO, "tests", "host_asan_driver.cpp"), "-o", drv, "-L" + tmp, "-l:libcamradepth_hip_asan.so",
                            "-Wl,-rpath," + tmp], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        r = subprocess.run([drv], capture_output=True, text=True, timeout=300, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        out = r.stdout + r.stderr
        assert r.returncode == 0 and "AddressSanitizer" not in out, out[-4000:]
        assert "build rc 0" in out and "short table rc -1" in out and "splits(cap 12) = 12" in out and "null conv: -1" in out, out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
