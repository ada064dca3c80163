"""ROCTX ranges (camradepth_amd.trace): a no-op unless CRD_ROCTX=1; with it, the ranges push / pop through the ROCTX library."""
import importlib
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ranges_are_noops_by_default():
    os.environ.pop("CRD_ROCTX", None)
    from camradepth_amd import trace
    importlib.reload(trace)
    assert not trace.enabled
    with trace.range("forward"):
        pass


def test_ranges_reach_roctx_when_enabled():
    code = ("from camradepth_amd import trace\n"
            "assert trace.enabled\n"
            "with trace.range('forward'):\n"
            "    with trace.range('backward:dec'):\n"
            "        pass\n"
            "print('nested ranges ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=dict(os.environ, CRD_ROCTX="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "nested ranges ok" in r.stdout, r.stderr[-2000:]
