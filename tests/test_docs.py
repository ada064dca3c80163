"""CPU: the 'Parity' numbers of README.md are generated from the committed log of the GPU test run (tools/readme_parity.py):
the golden-weights figure was typed by hand for three rounds and disagreed with its log each time.  This test fails when the block in
README.md is not what the newest committed profiles/rNN_gpu_tests.log generates."""
import glob
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_readme_parity_block_is_generated_from_the_committed_gpu_test_log():
    logs = sorted(glob.glob(os.path.join(REPO, "profiles", "r*_gpu_tests.log")))
    assert logs, "no committed GPU test log under profiles/"
    rel = os.path.relpath(logs[-1], REPO)
    r = subprocess.run([sys.executable, "tools/readme_parity.py", rel, "--check"], cwd=REPO, capture_output=True, text=True)
    assert r.returncode == 0, f"README.md's parity block is stale: run `python tools/readme_parity.py {rel}`\n{r.stdout}{r.stderr}"
