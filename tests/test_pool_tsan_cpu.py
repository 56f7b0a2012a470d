"""ThreadSanitizer on the host worker pool of sbe_step_batch (sbayes_amd/csrc/sbe_pool.h): built WITHOUT HIP and run
on the CPU (GPU sanitizers are not available on the pool).  tests/c/pool_tsan.cpp drives thousands
of generations of 1..64 items with randomised idle gaps and pool destruction; any data race ThreadSanitizer sees makes
the binary exit with code 66.  The committed log of a longer run: profiles/r3/tsan_pool.log."""
import os
import shutil
import subprocess
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_step_pool_is_race_free_under_thread_sanitizer(tmp_path):
    exe = tmp_path / "pool_tsan"
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread",
                            str(REPO / "tests" / "c" / "pool_tsan.cpp"), "-o", str(exe)], capture_output=True, text=True)
    if build.returncode != 0 and "tsan" in build.stderr.lower():
        pytest.skip("libtsan not installed")
    assert build.returncode == 0, build.stderr[-2000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    run = subprocess.run([str(exe), "4000"], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, (run.returncode, run.stderr[-3000:])
    assert "ok" in run.stdout and "WARNING: ThreadSanitizer" not in run.stderr
