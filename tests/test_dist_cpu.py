"""CPU, world_size 2, gloo: the multi-process driver logic (sharding, timing reduction, result-summary gather)
that bench.py uses with the nccl/RCCL backend on GPUs.  The per-rank 'analysis' here is the CPU oracle."""
import os
import subprocess
import sys

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
                          "--master-port", "29613", str(script)], capture_output=True, text=True, env=env, timeout=600)
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
