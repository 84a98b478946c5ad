"""CPU: the host worker pool (kiwi_amd/csrc/hostpool.hpp) under concurrent callers -- one caller per GPU is what kiwi_analyze_m produces when one
process drives every visible device.  tests/hostpool_stress.cpp: eight threads submit jobs at once (with and without a thread limit, some
throwing); built plain and with ThreadSanitizer."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "kiwi_amd", "csrc")


@pytest.mark.parametrize("sanitize", [False, True])
def test_concurrent_callers_share_the_workers(tmp_path, sanitize):
    exe = str(tmp_path / "hostpool_stress")
    cmd = ["g++", "-std=c++17", "-O2", "-pthread", "-I", CSRC, os.path.join(HERE, "hostpool_stress.cpp"), "-o", exe]
    if sanitize:
        cmd.insert(1, "-fsanitize=thread")
    subprocess.check_call(cmd)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    if sanitize and "unexpected memory mapping" in r.stderr:
        pytest.skip("ThreadSanitizer cannot run in this address-space layout")
    assert r.returncode == 0 and "bad 0" in r.stdout, r.stdout[-500:] + r.stderr[-2000:]
    assert "ThreadSanitizer" not in r.stderr, r.stderr[:3000]
