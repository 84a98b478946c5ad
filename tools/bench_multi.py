"""Developer measurement: several engine configurations (environment switches read at engine open) on the same workloads in ONE process --
device-resident step times and per-kernel HIP-event durations only (no CPU baseline, no end-to-end pass).  One JSON line per (workload, configuration).
  python tools/bench_multi.py c2,c2-64k "pos:;general:KAMD_POS_PATH=0" [steps]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from kiwi_amd.api import KiwiAmd
from kiwi_amd.workloads import fill_typo_rules, get_workload, workload_top_n, workload_typo

workloads = sys.argv[1].split(",")
configs = [(c.split(":", 1)[0], dict(kv.split("=", 1) for kv in c.split(":", 1)[1].split(",") if kv)) for c in sys.argv[2].split(";")]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
for w in workloads:
    model_path, texts, desc = get_workload(w)
    top_n = workload_top_n(w)
    for name, env in configs:
        saved = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            eng = KiwiAmd(model_path, 0)
            typo_cfg, typo = workload_typo(w), None
            if typo_cfg is not None:
                from kiwi_amd.api import Typo
                typo = Typo(eng.lib, typo_cfg[0], typo_cfg[1]); fill_typo_rules(typo); typo.prepare(True)
            batch = eng.stage(texts) if typo is None else eng.stage(texts, typo=typo, typo_threshold=typo_cfg[2])
            if top_n > 1:
                eng.fetch(batch, top_n).close()
            for _ in range(3):
                eng.run(batch)
            kt = {"scan_ms": 0.0, "lattice_ms": 0.0, "search_ms": 0.0, "finish_ms": 0.0}
            t0 = time.perf_counter()
            for _ in range(steps):
                r = eng.run(batch)
                for k in kt:
                    kt[k] += r[k]
            el = time.perf_counter() - t0
            res = eng.fetch(batch, top_n)
            ntok = sum(res.lib.kamd_res_token_num(res.h, i, 0) for i in range(min(256, len(texts))))
            res.close()
            print(json.dumps({"workload": w, "config": name, "env": env, "sentences_per_s": len(texts) * steps / el, "ms_per_step": 1000 * el / steps,
                              "kernel_ms": {k: v / steps for k, v in kt.items()}, "reruns": eng.reruns(batch), "tokens_first_256": ntok}), flush=True)
            batch.close()
            if typo is not None:
                typo.close()
            eng.close()
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
