"""CPU: the host-side parallel copier of the host-pointer path (csrc/pimemb_hostcopy.h) under
ThreadSanitizer -- contents, no overruns, no data races, clean start/stop of the worker threads."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sanitizer", ["thread", "address"])
def test_host_copier_under_sanitizers(tmp_path, sanitizer):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / ("hostcopy_" + sanitizer)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + sanitizer, "-pthread",
           "-I", os.path.join(ROOT, "pim-embedding-lookup_amd", "csrc"),
           os.path.join(ROOT, "tests", "cpp", "hostcopy_check.cpp"), "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtime not installed: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1")
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0 and "hostcopy ok" in run.stdout, run.stdout[-1000:] + run.stderr[-3000:]
