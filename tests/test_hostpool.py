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
