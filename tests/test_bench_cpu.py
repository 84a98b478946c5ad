"""CPU (not gpu): bench.py's main() end to end against the lane-emulated library (tests/hipemu) on a few sentences -- the whole
stage / warm-up / timed steps / fetch / cpu_baseline / JSON path of the driver contract, so that a Python error in the benchmark
is found here and not at the end of a round on the GPU box.  The numbers mean nothing (emulated kernels); the shape is checked."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

DRIVER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import torch
torch.cuda.synchronize = lambda *a, **k: None          # no GPU here; the emulated launches are synchronous
import kiwi_amd.workloads as W
orig = W.get_workload
W.get_workload = lambda name: (lambda p, t, d: (p, t[:24], d))(*orig(name))
import bench
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--workload", "small-c2"]
bench.main()
'''


def test_bench_main_emits_the_contract_line(tmp_path):
    subprocess.check_call(["make", "-C", os.path.join(HERE, "hipemu"), "-j8"], stdout=subprocess.DEVNULL)
    env = dict(os.environ, KAMD_LIB=os.path.join(HERE, "hipemu", "_build", "libkiwi_hipemu.so"))
    r = subprocess.run([sys.executable, "-c", DRIVER % {"root": ROOT}], env=env, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    assert out["n_gpus"] == 1 and out["steps"] == 2 and out["warmup"] == 1 and out["unit"] == "sentences/s" and out["higher_is_better"] is True
    assert out["scaling"] == "weak" and out["vs_baseline"] is None and out["data"] == "synthetic" and "workload" in out["config"]
    assert out["value"] > 0 and abs(out["value"] - 24 * 2 / (out["ms_per_step"] * 2 / 1000.0)) < 1e-6 * out["value"] + 1e-9
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in out["roofline"], k
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["unit"] == "GB/s" and abs(out["roofline"]["frac"] - out["roofline"]["achieved"] / out["roofline"]["peak"]) < 1e-12
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in out["cpu_baseline"], k
    assert out["cpu_baseline"]["kind"] in ("reference", "port") and out["cpu_baseline"]["value"] > 0
