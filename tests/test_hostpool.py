"""CPU: kiwi_amd/csrc/hostpool.hpp -- the persistent host worker pool of the text preparation / result assembly stages: its reading of the container's CPU
quota (cgroup cpu.max; round 4: a worker per VISIBLE CPU used a 16-CPU quota up and the kernel stopped the process for the rest of the period), every item
handed out exactly once, exceptions rethrown, concurrent callers sharing the workers."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_pool(tmp_path):
    exe = str(tmp_path / "hostpool_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I" + os.path.join(ROOT, "kiwi_amd", "csrc"), os.path.join(ROOT, "tests", "cxx", "hostpool_check.cpp"), "-o", exe])
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout.split()
    assert out == ["0"]


import pytest


@pytest.mark.parametrize("sanitize", [False, True])
def test_concurrent_callers_share_the_workers(tmp_path, sanitize):
    """tests/hostpool_stress.cpp: eight threads submit jobs at once (with and without a thread limit, some throwing) -- one caller per GPU is what
    kiwi_analyze_m produces when one process drives every visible device, and the pipelined parts of a batch add more; built plain and with ThreadSanitizer."""
    exe = str(tmp_path / "hostpool_stress")
    cmd = ["g++", "-std=c++17", "-O2", "-pthread", "-I", os.path.join(ROOT, "kiwi_amd", "csrc"), os.path.join(ROOT, "tests", "hostpool_stress.cpp"), "-o", exe]
    if sanitize:
        cmd.insert(1, "-fsanitize=thread")
    subprocess.check_call(cmd)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    if sanitize and "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot run in this address-space layout")
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]
    assert "ThreadSanitizer" not in r.stderr, r.stderr[:3000]
