"""CPU, world_size 2, gloo: the multi-process driver logic (sharding, timing reduction, result-summary gather)
that bench.py uses with the nccl/RCCL backend on GPUs.  The per-rank 'analysis' here is the CPU oracle."""
import os
import subprocess
import sys

from corpora import free_port

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "oracle"))
from kiwi_amd import dist
from kiwi_amd.synth import SynthModel, SMALL_SPEC
import oraclelib
rank, local, world = dist.env_rank_world()
dist.init("gloo")
sm = SynthModel(SMALL_SPEC)
path = os.path.join({root!r}, "_data", "small.raw")
texts = sm.make_corpus(40, 9, min_jamo=5, max_jamo=40)
mine = dist.shard_indices(len(texts), rank, world)
o = oraclelib.OracleKiwi(path)
tokens = sum(len(o.analyze(texts[i])[0][0]) for i in mine)
dist.barrier()
t = dist.max_over_ranks(1.0 + rank)
summary = dist.gather_counts([len(mine), tokens])
w = dist.weak_shard(texts, rank)
if rank == 0:
    print(json.dumps({{"world": world, "tmax": t, "summary": summary, "weak_first": w[0] == texts[0]}}))
else:
    assert w[0] != texts[0] or len(texts) == 1
'''


def test_two_rank_gloo_driver(small_model, tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(free_port()), str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    r = json.loads(line)
    assert r["world"] == 2 and r["tmax"] == 2.0 and r["weak_first"]
    assert sum(s[0] for s in r["summary"]) == 40 and all(s[1] > 0 for s in r["summary"])
    # the union of the shards equals a single-process run
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oraclelib
    sm, path = small_model
    o = oraclelib.OracleKiwi(path)
    texts = sm.make_corpus(40, 9, min_jamo=5, max_jamo=40)
    assert sum(s[1] for s in r["summary"]) == sum(len(o.analyze(t)[0][0]) for t in texts)


GATHER_WORKER = r'''
import os, sys, json, hashlib
sys.path.insert(0, {root!r})
import numpy as np
from kiwi_amd import dist
from kiwi_amd.api import KiwiAmd, Results
from kiwi_amd.synth import SynthModel, SMALL_SPEC
rank, local, world = dist.env_rank_world()
dist.init("gloo")
sm = SynthModel(SMALL_SPEC)
path = os.path.join({root!r}, "_data", "small.raw")
texts = sm.make_corpus(301, 19, min_jamo=5, max_jamo=90) + ["", " ", "가나다 \"라마\" 바사."]
eng = KiwiAmd(path, lib_path={lib!r})                       # the kernels run on the CPU lane emulator here (tests/hipemu), on cuda:LOCAL_RANK on a GPU box
mine = [texts[i] for i in dist.shard_indices(len(texts), rank, world)]
res = eng.analyze_batch(mine, top_n=2)
parts = dist.gather_packed(res.pack())                      # all-gather of sizes + gather of the packed token records to rank 0
if rank == 0:
    merged = Results.merge_strided(eng.lib, parts)
    single = eng.analyze_batch(texts, top_n=2)                # the same corpus in one process
    a, b = merged.pack(), single.pack()
    print(json.dumps({{"world": world, "texts": merged.n_texts(), "bytes": int(a.nbytes), "equal": bool(a.nbytes == b.nbytes and (a == b).all()),
                      "sha": hashlib.sha256(a.tobytes()).hexdigest()[:16], "part_bytes": [int(p.nbytes) for p in parts]}}))
else:
    assert parts is None
eng.close()
'''


def test_two_rank_gather_of_packed_token_records_equals_single_process(small_model, tmp_path):
    """north_star: 'RCCL over xGMI only for the final result gather'.  Two ranks (gloo, CPU; device kernels on the lane emulator)
    analyse an index-strided split of one corpus, pack their results (kamd_res_pack), gather the packed token records on rank 0
    (kiwi_amd.dist.gather_packed) and merge them in input order (kamd_res_merge_strided): the merged records equal a
    single-process run of the whole corpus BYTE FOR BYTE (tokens, positions, fp32 scores, forms, top-2 analyses)."""
    import json
    lib = os.path.join(ROOT, "tests", "hipemu", "_build", "libkiwi_hipemu.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "hipemu"), "-j8"])
    script = tmp_path / "gather_worker.py"
    script.write_text(GATHER_WORKER.format(root=ROOT, lib=lib))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                          "--master-port", str(free_port()), str(script)], capture_output=True, text=True, env=env, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert r["world"] == 2 and r["texts"] == 304 and r["equal"], r
    assert len(r["part_bytes"]) == 2 and all(b > 1000 for b in r["part_bytes"])
