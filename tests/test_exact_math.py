"""csrc/exact_math.hpp: expf / logf / tanhf restated from glibc's algorithm so that device code can agree bit-for-bit with the reference's
std::exp / std::log on floats (SkipBigram / CoNgram mixtures).  CPU: against libm itself, dense sweeps + random arguments.
GPU (-m gpu): the same functions evaluated on the device through kamd_debug_exact_math, against libm."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHECKER = r'''
#include <cmath>
#include <cstdio>
#include <random>
#include "exact_math.hpp"
int main()
{
	using namespace kamd::exact;
	std::mt19937_64 rng(1);
	size_t bad = 0, n = 0;
	for (uint32_t u = f2u(-110.f); u > f2u(-1e-6f); u -= 4999) { float x = u2f(u); ++n; bad += f2u(expf_glibc(x)) != f2u(expf(x)); }
	for (uint32_t u = f2u(1e-30f); u < f2u(100.f); u += 5003) { float x = u2f(u); ++n; bad += f2u(expf_glibc(x)) != f2u(expf(x)); }
	for (uint32_t u = 1; u < 0x7f800000u; u += 5009) { float x = u2f(u); ++n; bad += f2u(logf_glibc(x)) != f2u(logf(x)); }
	for (int i = 0; i < 3000000; ++i)
	{
		float x = -(float)(rng() % 2000000) * 1e-5f; ++n; bad += f2u(expf_glibc(x)) != f2u(expf(x));
		float y = 1.0f + (float)(rng() % 16000000) * 1e-6f; ++n; bad += f2u(logf_glibc(y)) != f2u(logf(y));
	}
	// tanhf / expm1f (the frequency-based unknown-form scores: arguments are frequencies over weights, >= 0; negative ones for completeness)
	for (uint32_t u = 0; u < f2u(40.f); u += 97) { float x = u2f(u); ++n; bad += f2u(tanhf_glibc(x)) != f2u(tanhf(x)); bad += f2u(tanhf_glibc(-x)) != f2u(tanhf(-x)); }
	for (uint32_t u = 0; u < f2u(90.f); u += 101) { float x = u2f(u); ++n; bad += f2u(expm1f_glibc(x)) != f2u(expm1f(x)); bad += f2u(expm1f_glibc(-x)) != f2u(expm1f(-x)); }
	for (int q = 1; q < 4000; ++q) for (float w : { 3.f, 35.f, 60.f, 1.f, 0.37f }) { float x = (float)q / w; ++n; bad += f2u(tanhf_glibc(x)) != f2u(tanhf(x)); }
	printf("%zu %zu\n", n, bad);
	return bad != 0;
}
'''


def test_host_exact_math_equals_libm(tmp_path):
    src = tmp_path / "check.cpp"
    src.write_text(CHECKER)
    exe = str(tmp_path / "check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-I" + os.path.join(ROOT, "kiwi_amd", "csrc"), str(src), "-o", exe])
    n, bad = map(int, subprocess.run([exe], check=True, capture_output=True).stdout.split())
    assert n > 5_000_000 and bad == 0


@pytest.mark.gpu
def test_device_exact_math_equals_libm():
    lib = C.CDLL(os.path.join(ROOT, "kiwi_amd", "libkiwi_hip.so"))
    lib.kamd_debug_exact_math.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    libm = C.CDLL("libm.so.6")
    libm.expf.restype = C.c_float
    libm.expf.argtypes = [C.c_float]
    libm.logf.restype = C.c_float
    libm.logf.argtypes = [C.c_float]
    rng = np.random.default_rng(3)
    xs = np.concatenate([-rng.random(100000, np.float32) * 30, rng.random(100000, np.float32) * 16 + 1e-6,
                         np.array([0.0, -0.0, 1.0, -13.0, -103.0, -104.5, 88.0, 1e-40, 3e38], np.float32)]).astype(np.float32)
    e = np.zeros_like(xs)
    l = np.zeros_like(xs)
    assert lib.kamd_debug_exact_math(xs.ctypes.data, e.ctypes.data, l.ctypes.data, len(xs)) == 0
    pos = xs > 0
    we = np.array([libm.expf(float(v)) for v in xs], np.float32)
    wl = np.array([libm.logf(float(v)) for v in xs[pos]], np.float32)
    assert (e.view(np.uint32) == we.view(np.uint32)).all()
    assert (l[pos].view(np.uint32) == wl.view(np.uint32)).all()
    lib.kamd_debug_exact_tanh.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
    libm.tanhf.restype = C.c_float
    libm.tanhf.argtypes = [C.c_float]
    ts = np.concatenate([rng.random(100000, np.float32) * 3, rng.random(50000, np.float32) * 25, -rng.random(20000, np.float32) * 5,
                         (np.arange(1, 4000, dtype=np.float32) / np.float32(35.0)), (np.arange(1, 4000, dtype=np.float32) / np.float32(3.0)),
                         np.array([0.0, -0.0, 1e-30, 1e-9, 0.34657, 0.34658, 1.0397, 1.0398, 1.0, 21.99, 22.0, 23.0, 100.0], np.float32)]).astype(np.float32)
    t = np.zeros_like(ts)
    assert lib.kamd_debug_exact_tanh(ts.ctypes.data, t.ctypes.data, len(ts)) == 0
    wt = np.array([libm.tanhf(float(v)) for v in ts], np.float32)
    assert (t.view(np.uint32) == wt.view(np.uint32)).all()
