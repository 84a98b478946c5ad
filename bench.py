#!/usr/bin/env python3
"""Benchmark of the batched analyze hot path (dictionary scan -> lattice -> Viterbi/Knlm) on MI355X.

One "step" = one pass of the three HIP kernels over one batch whose inputs are already resident in HBM
(BASELINE.json metric: sentences/sec on batched analyze(); workload = BASELINE configs[1]: 8k synthetic
40-jamo sentences, Knlm, top-1).  Prints ONE JSON line (rank 0).  Multi-GPU: one process per GPU, each rank
analyses its own shard of the same size (weak scaling; the path has no data-path collective).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E ~8 TB/s


def cpu_baseline(model_path, texts, budget_s=20.0, top_n=1, typo=None, time_reference=True, cong_global=False):
    """Times the CPU path on this box's host cores on a bounded sample of the same workload.
    Uses the real reference TUs (oracle/_ref) when the prebuilt library travelled with the repo, else this
    repo's CPU oracle ("port").  Also returns the oracle's ALG_BYTES event counts on its sample."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oraclelib
    import refbridge
    cores = os.cpu_count() or 1
    orc = oraclelib.OracleKiwi(model_path)
    if cong_global:
        orc.set_cong_global(True)      # (the global scoring: event counts here, and both timed runners below)
    orc_typo = ref_typo = None
    thr = 2.5
    if typo is not None:      # the same rules on both CPU sides
        from kiwi_amd.workloads import fill_typo_rules
        cont, leng, thr = typo
        orc_typo = oraclelib.OracleTypo(cont, leng); fill_typo_rules(orc_typo); orc_typo.prepare(True)
        if refbridge.available():
            ref_typo = refbridge.RefTypo(cont, leng); fill_typo_rules(ref_typo, cond_by_name=True); ref_typo.prepare(True)
    # single-thread sample: 2048 sentences, fewer when the model is slow on the CPU (SkipBigram lattices: tens of sentences/s)
    probe, _ = orc.analyze_batch(texts[:32], top_n=top_n, threads=1, typo=orc_typo, typo_threshold=thr)
    sample = texts[:int(min(2048, max(32, 32 / max(probe, 1e-6) * 4.0)))]
    orc.counters(reset=True)
    sec1, _ = orc.analyze_batch(sample, top_n=top_n, threads=1, typo=orc_typo, typo_threshold=thr)
    counts = orc.counters()
    alg = oraclelib.alg_bytes(counts)
    per_sentence = {k: v / len(sample) for k, v in alg.items()}
    out = {"alg_bytes_per_sentence": per_sentence, "alg_sample": len(sample)}
    if not time_reference:      # (N > 1: the CPU baseline is a rank-0, N = 1 measurement; the algorithmic bytes are still needed for `roofline`)
        return out
    arch_name = "none"
    if refbridge.available():
        # the reference picks its best SIMD architecture at run time; so does its baseline here (AVX-512 builds measured no faster than AVX2).
        # The quantised CoNgram path exists for the SIMD builds only (arch none falls back to fp32).
        arch = 0
        if refbridge.x86_available():
            try:
                flags = open("/proc/cpuinfo").read()
                arch, arch_name = (4, "avx2") if " avx2" in flags else (3, "sse4_1") if " sse4_1" in flags else (0, "none")
            except OSError:
                pass
        ref = refbridge.RefKiwi(model_path, arch=arch, x86=arch > 0, model_dir_sbg="cong_global") if cong_global else refbridge.RefKiwi(model_path, arch=arch, x86=arch > 0)
        kind, runner, rtypo = "reference", ref, ref_typo
    else:
        kind, runner, rtypo = "port", orc, orc_typo
    # Sound timing (oracle/timed_pool.hpp): the worker threads exist before the clock starts and have analysed the sample once (warm thread_local
    # containers and allocator arenas), the timed region is whole passes over the sample, repeated until >= 2 s of wall time.
    s1, p1, _ = runner.analyze_batch_timed(sample, top_n=top_n, threads=1, min_seconds=min(2.0, budget_s / 4), typo=rtypo, typo_threshold=thr)
    rate1 = len(sample) * p1 / s1
    phys = physical_cores()
    # multi-thread sample: about one second of work for all threads per pass (never fewer than 64 texts per thread, never more than the workload)
    n_mt = int(min(len(texts), max(64 * cores, rate1 * min(cores, phys) * 1.0)))
    smt, pmt, _ = runner.analyze_batch_timed(texts[:n_mt], top_n=top_n, threads=cores, min_seconds=max(2.0, budget_s / 4), typo=rtypo, typo_threshold=thr)
    rate = n_mt * pmt / smt
    # Under a CFS quota far below the visible CPUs a thread per logical CPU is not the reference's best showing (its pool is sized by the caller:
    # kiwi_init(num_threads)): the baseline is its BEST rate over {all logical CPUs, 4 x, 2 x, 1 x the quota}, every count reported.
    by_threads = {cores: rate}
    quota = cpu_quota_cores()
    best_threads = cores
    if quota and quota * 4 < cores:
        for k in (4, 2, 1):
            th = max(1, int(round(k * quota)))
            s_k, p_k, _ = runner.analyze_batch_timed(texts[:n_mt], top_n=top_n, threads=th, min_seconds=2.0, typo=rtypo, typo_threshold=thr)
            by_threads[th] = n_mt * p_k / s_k
            if by_threads[th] > rate:
                rate, best_threads, smt, pmt = by_threads[th], th, s_k, p_k
    # what the box lets this process use: the same thread count on register-only work (a container can see every logical CPU of the host and be
    # scheduled on a fraction of them); the baseline cannot scale beyond it
    usable = oraclelib.cpu_capacity(cores, 1.0)
    out["cpu_baseline"] = {"value": rate, "unit": "sentences/s", "cores": best_threads, "logical_cpus": cores, "kind": kind,
                           "threads": best_threads, "rate_by_threads": {str(k): round(v, 1) for k, v in sorted(by_threads.items())}, "physical_cores": phys, "cpu_quota_cores": cpu_quota_cores(), "usable_cores_measured": usable, "single_thread": rate1,
                           "scaling_efficiency_vs_physical_cores": rate / (rate1 * max(1, min(cores, phys))),
                           "scaling_efficiency_vs_usable_cores": rate / (rate1 * max(1.0, min(usable, float(phys)))),
                           "sample": f"{pmt} timed passes over {n_mt} sentences of the same workload ({smt:.2f} s) on {best_threads} persistent threads (the best of the thread counts in rate_by_threads) after one untimed warm-up pass "
                                     f"(reference arch {arch_name}; texts handed out through one atomic counter, results dropped); single thread: {rate1:.0f} sentences/s ({p1} passes over {len(sample)})"}
    return out


def physical_cores():
    """Physical cores of this box (unique (package, core id) pairs of /proc/cpuinfo); the logical count where that cannot be read."""
    try:
        seen, pkg = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                pkg = line.split(":")[1].strip()
            elif line.startswith("core id"):
                seen.add((pkg, line.split(":")[1].strip()))
        return len(seen) or (os.cpu_count() or 1)
    except OSError:
        return os.cpu_count() or 1


def cpu_quota_cores():
    """CPU time this container may use, in cores (cgroup CFS quota, v2 cpu.max or v1 cfs_quota_us / cfs_period_us); None when unlimited."""
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(p)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def model_facts(workload):
    """Facts of the synthetic model a workload runs on (SURVEY.md section 8(d) asks for order 4 / 2^20 n-gram nodes; the generator packs an n-gram
    into 63 bits, which caps the 'full' model at order 3 -- stated here rather than implied)."""
    import struct
    from dataclasses import asdict
    from kiwi_amd.container import read_container
    from kiwi_amd.workloads import DATA, WORKLOADS, _spec
    spec_name = WORKLOADS[workload][0]
    sp = asdict(_spec(spec_name))
    out = {"spec": spec_name, "knlm_order": sp.get("lm_order"), "dictionary_words": sp.get("n_words"), "skipbigram": bool(sp.get("use_sbg"))}
    try:
        _, sec = read_container(os.path.join(DATA, f"{spec_name}.raw"))
        for name, a in sec.items():
            if name.lower().startswith("knlm") and a.nbytes >= 96:
                out["knlm_nodes"] = struct.unpack_from("<Q", a.tobytes()[:8])[0]
                out["knlm_bytes"] = int(a.nbytes)
            if "form" in name.lower() and "ptr" in name.lower():
                out["forms"] = int(a.nbytes // 4 - 1)
    except Exception:
        pass
    return out


def measured_traffic(workload, kernels):
    """HBM bytes per launch of the named kernels together, from the PMC passes committed under profiles/ (FETCH_SIZE + WRITE_SIZE, separate
    rocprofv3 --pmc runs of `bench.py --workload W --kernels-only`, tools/measure_round.sh; see profiles/README.md).  profiles/traffic.json
    holds one entry per measured workload; None when this workload is not on file."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "traffic.json")
    try:
        t = json.load(open(path))
        entry = t.get(workload) if "kernels" not in t else (t if t.get("workload") == workload else None)
        if entry is None:
            return None
        vals = [entry["kernels"][k]["hbm_bytes_per_launch"] for k in kernels if k in entry["kernels"]]
        return float(sum(vals)) if vals else None
    except (OSError, KeyError, ValueError, TypeError):
        return None


def copy_bandwidth_gbs():
    """Measured device-to-device copy bandwidth (read + written bytes per second) of this GPU: what a purely streaming kernel achieves, beside the nominal peak."""
    import torch
    if not torch.cuda.is_available():
        return None
    n = 1 << 28
    a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
    for _ in range(2):
        b.copy_(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps):
        b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    del a, b
    return 2.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def side_measurement(eng, workload, steps=20, limit=0, min_seconds=0.0):
    """Device-resident rate, per-kernel times and roofline fraction of another workload (an entry of config.also).  eng: an engine on the workload's
    model, or None -- then one is opened (and closed) here.  limit: only the first N sentences (named in the entry).  The algorithmic bytes per
    sentence come from the oracle's event counters on a bounded sample of the same workload, as for the headline workload."""
    from kiwi_amd.api import KiwiAmd, Typo
    from kiwi_amd.workloads import fill_typo_rules, get_workload, workload_lm_mode, workload_top_n, workload_typo
    model_path, texts, desc = get_workload(workload)
    if limit and limit < len(texts):
        texts = texts[:limit]
        desc += f" [first {limit} sentences only]"
    own = eng is None
    if own:
        eng = KiwiAmd(model_path, 0, lm_mode=workload_lm_mode(workload))
    top_n = workload_top_n(workload)
    typo_cfg, typo = workload_typo(workload), None
    if typo_cfg is not None:
        typo = Typo(eng.lib, typo_cfg[0], typo_cfg[1]); fill_typo_rules(typo); typo.prepare(True)
    if len(texts) > 4096 and typo is None:
        eng.analyze_batch(texts[:4096], top_n=top_n).close()      # (untimed sample batch: the engine's capacities follow what its model needs)
    batch = eng.stage(texts) if typo is None else eng.stage(texts, typo=typo, typo_threshold=typo_cfg[2])
    if top_n > 1:
        eng.fetch(batch, top_n).close()
    t1 = time.perf_counter()
    eng.run(batch)
    one = time.perf_counter() - t1
    if one > 1.0:
        steps = 1      # (seconds per batch: one timed pass)
    else:
        for _ in range(2):
            eng.run(batch)
        steps = max(steps, int(min_seconds / max(one, 1e-6)) + 1) if min_seconds else steps
    kt = {"scan_ms": 0.0, "lattice_ms": 0.0, "search_ms": 0.0, "finish_ms": 0.0}
    t0 = time.perf_counter()
    for _ in range(steps):
        r = eng.run(batch)
        for k in kt:
            kt[k] += r[k]
    el = time.perf_counter() - t0
    rerun_chunks, rerun_ms = eng.reruns(batch)
    info = batch.info()
    batch.close()
    if typo is not None:
        typo.close()
    if own:
        eng.close()
    out = {"workload": desc, "value": len(texts) * steps / el, "unit": "sentences/s", "steps": steps, "ms_per_step": 1000.0 * el / steps, "kernel_ms": {k: v / steps for k, v in kt.items()},
           "sentences": len(texts), "top_n": top_n, "device_bytes": info["device_bytes"], "rerun_chunks": rerun_chunks,
           "m_jamo_per_s": info["units"] * steps / el / 1e6}
    try:
        per = cpu_baseline(model_path, texts, top_n=top_n, typo=typo_cfg, time_reference=False, cong_global=workload_lm_mode(workload) == 4)["alg_bytes_per_sentence"]
        out["alg_bytes_per_sentence"] = per
        out["roofline_frac"] = per["search"] * len(texts) / (out["kernel_ms"]["search_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
        out["all_kernels_achieved"] = per["total"] * len(texts) / (sum(out["kernel_ms"].values()) * 1e-3) / 1e9
    except Exception as e:      # (informative: never fail the bench line over it)
        out["roofline_error"] = repr(e)[:200]
    return out


def capi_rate(model_path, workload, top_n, passes=5):
    """kiwi_analyze_m -- the symbol this library replaces -- timed by a C client (tools/capi_bench.c) in its own process: reader -> receiver, results
    delivered in input order, every visible GPU of this process driven by the one handle."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "_build", "capi_bench")
    corpus = os.path.join(ROOT, "_data", f"{workload}.corpus.txt")
    if not (os.path.exists(exe) and os.path.exists(corpus)):
        return None
    try:
        r = subprocess.run([exe, model_path, corpus, str(passes), str(top_n)], capture_output=True, text=True, timeout=300)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:      # the block is informative: never fail the bench line over it
        return {"error": repr(e)[:200]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default: 100 for the headline workload c2 -- a 0.2 s timed region that a utilisation sampler can see --, 20 otherwise)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, help="default: c2-64k on one GPU (the >= 64k-sentence regime the north-star target is quoted on; c2 is measured beside it as config.also), "
                                                      "c4-cong on several (131 072 sentences per GPU = BASELINE config 4's 1M sentences at 8 GPUs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kernels-only", action="store_true", help="profiler passes: only the warm-up and the timed region of the named workload (no CPU baseline, no side measurement, "
                    "no C-API client, no end-to-end pass), so that per-kernel counters and statistics belong to this workload alone")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default): every rank analyses a batch of the workload's size; strong: ONE corpus split over the ranks by index (text i -> rank i %% N)")
    ap.add_argument("--no-side-models", action="store_true", help="config.also without the workloads that need another model loaded (c4-cong, c3-sbg)")
    ap.add_argument("--limit", type=int, default=0, help="diagnostics: only the first N sentences of the workload (named in config.workload)")
    args = ap.parse_args()
    if args.workload is None:
        args.workload = "c2-64k" if args.gpus <= 1 else "c4-cong"
    if args.steps is None:
        args.steps = 400 if args.workload == "c2" else 200 if args.workload == "c2-64k" else 40      # a timed region of >= 1 s (a utilisation sampler sees it)

    import torch
    from kiwi_amd import dist
    from kiwi_amd.api import KiwiAmd
    from kiwi_amd.workloads import fill_typo_rules, get_workload, workload_lm_mode, workload_top_n, workload_typo
    rank, local_rank, world = dist.env_rank_world()
    if world > 1:
        dist.init("nccl", local_rank)

    if world > 1 and rank != 0:
        dist.barrier()     # rank 0 generates / caches the model and corpus first
    model_path, texts, desc = get_workload(args.workload)
    if world > 1 and rank == 0:
        dist.barrier()
    if args.limit and args.limit < len(texts):
        texts = texts[:args.limit]
        desc += f" [first {args.limit} sentences only]"
    # weak scaling: every rank analyses a same-sized shard; rotate so shards differ
    strong = args.scaling == "strong" and world > 1
    shard = [texts[i] for i in dist.shard_indices(len(texts), rank, world)] if strong else dist.weak_shard(texts, rank)
    n = len(shard)
    n_job = len(texts) if strong else n * world      # sentences the whole job analyses per step

    top_n = workload_top_n(args.workload)
    if world > 1 and "KAMD_HOST_THREADS" not in os.environ:
        # one process per GPU on one host: the ranks share the container's CPUs -- each takes its share of the two-workers-per-quota-CPU pool
        # (hostpool.hpp) instead of a whole one (N pools of that size used the quota up: CFS throttling, profiles/r04_t_cfs_throttling.txt)
        os.environ["KAMD_HOST_THREADS"] = str(max(2, int(2 * (cpu_quota_cores() or os.cpu_count() or 2) / world)))
    eng = KiwiAmd(model_path, local_rank, lm_mode=workload_lm_mode(args.workload))
    typo_cfg, typo = workload_typo(args.workload), None
    if typo_cfg is not None:
        from kiwi_amd.api import Typo
        typo = Typo(eng.lib, typo_cfg[0], typo_cfg[1])
        fill_typo_rules(typo)
        typo.prepare(True)
    # one untimed sample batch first: the engine sizes its LDS arrays and state arenas by what the model's dictionary and search produced in the batches
    # before (matches per text unit, states per chunk) -- a fresh engine starts from worst-case capacities
    if len(shard) > 4096 and typo is None:
        eng.analyze_batch(shard[:4096], top_n=top_n).close()
    batch = eng.stage(shard) if typo is None else eng.stage(shard, typo=typo, typo_threshold=typo_cfg[2])
    info = batch.info()
    if top_n > 1:
        eng.fetch(batch, top_n).close()     # (runs the batch with this N) a batch keeps the top-N of its last fetch: every run below searches with it

    def sync():
        torch.cuda.synchronize()
        dist.barrier()

    tw0 = time.perf_counter()
    for _ in range(args.warmup):
        eng.run(batch)
    sync()
    # --steps is a MINIMUM: the timed region lasts at least a second (a 20-step region of a 5 ms step is 0.1 s -- too short for a utilisation sampler, and
    # for a stable mean); the step count actually timed is what the line reports.  Profiler passes (--kernels-only) time exactly what they were asked.
    steps_requested = args.steps
    if args.warmup and not args.kernels_only:
        est = dist.max_over_ranks((time.perf_counter() - tw0) / args.warmup, device="cuda" if world > 1 else "cpu")
        if est > 0 and args.steps * est < 1.0:
            args.steps = min(100000, int(1.0 / est) + 1)
    t0 = time.perf_counter()
    kt = {"scan_ms": 0.0, "lattice_ms": 0.0, "search_ms": 0.0, "finish_ms": 0.0}
    for _ in range(args.steps):
        r = eng.run(batch)          # launches + stream sync; per-kernel durations come from HIP events on the engine's stream
        for k in kt:
            kt[k] += r[k]
    sync()
    elapsed = time.perf_counter() - t0
    elapsed = dist.max_over_ranks(elapsed, device="cuda" if world > 1 else "cpu")
    for k in kt:
        kt[k] /= args.steps
    # kamd_run searches every chunk to the end (chunks that outgrow their scratch are searched again inside it, with larger capacities): the
    # wall time above contains those extra passes, the per-kernel HIP-event durations are the first pass's
    rerun_chunks, rerun_ms = eng.reruns(batch)

    # sanity: the staged batch really was analysed (token count > 0, no failed chunk)
    res = eng.fetch(batch, top_n)
    n_tok = sum(res.lib.kamd_res_token_num(res.h, i, 0) for i in range(min(256, n)))
    assert n_tok > 0
    summary = dist.gather_counts([n, n_tok], device="cuda" if world > 1 else "cpu")   # the only result "gather": per-rank counts
    assert len(summary) == world

    # The final result gather (the path's only exchange step): every rank packs its token records, the packed buffers are gathered on rank 0
    # over the process group (RCCL over xGMI; device tensors) and merged -- in input order when the corpus was split by index.  Outside the timed
    # regions above: `value` is device-resident by contract; the gather's own time is reported.
    gather = None
    if world > 1:
        from kiwi_amd.api import Results
        packed = res.pack()
        sync()
        tg = time.perf_counter()
        parts = dist.gather_packed(packed, device="cuda")
        merged_texts = 0
        if rank == 0:
            if strong:
                m = Results.merge_strided(eng.lib, parts); merged_texts = m.n_texts(); m.close()
            else:
                for p in parts:
                    m = Results.merge_strided(eng.lib, [p]); merged_texts += m.n_texts(); m.close()
        sync()
        gather = {"ms": 1000.0 * (time.perf_counter() - tg), "bytes_per_rank": int(packed.nbytes), "merged_texts": merged_texts,
                  "what": "all-gather of packed sizes + gather of packed token records to rank 0 (kiwi_amd.dist.gather_packed) + merge" + (" in input order" if strong else "")}
        if rank == 0:
            assert merged_texts == n_job, (merged_texts, n_job)

    res.close()
    batch.close()      # (the end-to-end pass below stages its own batch: a large workload does not fit the device twice)
    also = None
    if args.workload == "c2-64k" and world == 1 and not args.limit and not args.kernels_only:
        # the other BASELINE configurations beside the headline, each with its own roofline fraction: configs[1] at its own batch size (8192 sentences:
        # the latency-bound regime), configs[4] (typo correction), configs[3]'s model and length mix on one GPU, configs[2] (SkipBigram, top-3: the
        # synthetic tables do not prune like a real model -- seconds per batch, hence a bounded sample)
        also = [side_measurement(eng, "c2", min_seconds=0.5), side_measurement(eng, "c5", min_seconds=0.5)]
        if not args.no_side_models:
            also += [side_measurement(None, "c4-cong", steps=10), side_measurement(None, "c4-cong-global", steps=3), side_measurement(None, "c3-sbg", steps=3)]

    # End to end: UTF-16 strings resident on the host -> kamd_analyze_batch (host text preparation, H2D, kernels,
    # D2H, result assembly) -> packed token records resident on the host.  Timed through the C ABI on an already packed buffer.
    e2e = None
    if not args.kernels_only:
        tkw = {} if typo is None else {"typo": typo, "typo_threshold": typo_cfg[2]}
        from kiwi_amd.api import pack_texts
        flat, offs = pack_texts(shard)
        e2e_steps = 30      # (a spread needs a sample: >= 30 batches unless one takes seconds)
        tw = time.perf_counter()
        eng.analyze_packed(flat, offs, top_n, **tkw).close()      # warm-up (device blocks, pinned buffers, host pool)
        first = time.perf_counter() - tw
        if first > 2.0:
            e2e_steps = 1      # a slow workload (seconds per batch): one timed batch
        else:
            if first > 0.25:
                e2e_steps = 5
            eng.analyze_packed(flat, offs, top_n, **tkw).close()
        sync()
        te = time.perf_counter()
        d2h = 0
        per_batch = []
        for _ in range(e2e_steps):
            tb = time.perf_counter()
            r = eng.analyze_packed(flat, offs, top_n, **tkw)
            d2h = r.d2h_bytes()
            r.close()
            per_batch.append(1000.0 * (time.perf_counter() - tb))
        sync()
        e2e_elapsed = dist.max_over_ranks(time.perf_counter() - te, device="cuda" if world > 1 else "cpu")
        per_batch.sort()
        pct = lambda q: per_batch[min(len(per_batch) - 1, int(q * len(per_batch)))]
        host_cores = cpu_quota_cores() or os.cpu_count() or 1
        e2e = {"value": n_job * e2e_steps / e2e_elapsed, "unit": "sentences/s", "ms_per_batch": 1000.0 * e2e_elapsed / e2e_steps, "steps": e2e_steps,
               "ms_per_batch_median": pct(0.5), "ms_per_batch_p10": pct(0.1), "ms_per_batch_p90": pct(0.9),
               # what the host side of a batch costs in CPU time if the whole quota works on it for the whole batch (an upper bound: the kernels hide behind it)
               "host_cores": host_cores, "host_us_per_sentence_core": 1000.0 * pct(0.5) * host_cores / max(1, n),
               "region": "host UTF-16 strings -> kamd_analyze_batch -> host token records (text preparation, H2D, kernels, D2H, result assembly)",
               "h2d_bytes_per_batch": int(flat.nbytes), "d2h_bytes_per_batch": d2h}

    if rank == 0:
        total_sent = n_job * args.steps
        value = total_sent / elapsed
        out = {
            "metric": "sentences/sec on batched analyze(), device kernels with inputs resident in HBM (dictionary scan + lattice + Viterbi/%s, top-%d); host-to-host rate: see e2e" % ("Knlm+SkipBigram" if args.workload.endswith("-sbg") else "CoNgram (local, 8-bit)" if "cong" in args.workload else "Knlm, typo correction" if typo is not None else "Knlm", top_n),
            "value": value, "unit": "sentences/s", "n_gpus": world, "steps": args.steps, "steps_requested": steps_requested, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong" if strong else "weak",
            "vs_baseline": None, "dtype": "int32+f32", "data": "synthetic",
            "config": {"workload": desc, "sentences_per_gpu": n, "chunks_per_gpu": info["chunks"], "jamo_per_gpu": info["units"],
                       "parallelism": f"shard{world}", "m_jamo_per_s": info["units"] * world * args.steps / elapsed / 1e6,
                       "model": model_facts(args.workload),
                       "kernel_ms": kt, "device_bytes": info["device_bytes"], "rerun_chunks": rerun_chunks, "rerun_ms": rerun_ms},
        }
        if not args.no_cpu_baseline and not args.kernels_only:
            cb = cpu_baseline(model_path, texts, top_n=top_n, typo=typo_cfg, time_reference=world == 1, cong_global=workload_lm_mode(args.workload) == 4)
            per = cb["alg_bytes_per_sentence"]
            search_bytes = per["search"] * n
            achieved = search_bytes / (kt["search_ms"] * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "k_pos_path + k_best_path (the search: position steps, then what they hand over and the end stage)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(args.workload, ("k_pos_path", "k_best_path")),
                               "traffic_source": "profiles/traffic.json: FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc passes over this command with --kernels-only (tools/measure_round.sh), NOT counted during this run",
                               "measured_copy_GBs": copy_bandwidth_gbs() if world == 1 else None,
                               "alg_bytes_per_sentence": per, "all_kernels_achieved": per["total"] * n / ((kt["scan_ms"] + kt["lattice_ms"] + kt["search_ms"] + kt["finish_ms"]) * 1e-3) / 1e9}
            if "cpu_baseline" in cb:
                out["cpu_baseline"] = cb["cpu_baseline"]
                if e2e is not None:
                    e2e["vs_cpu_baseline"] = e2e["value"] / cb["cpu_baseline"]["value"]
        if also is not None:
            out["config"]["also"] = also
        if world == 1 and typo is None and not args.limit and not args.kernels_only:
            cr = capi_rate(model_path, args.workload, top_n)
            if cr is not None:
                out["capi"] = cr
        if e2e is not None:
            out["e2e"] = e2e
        if gather is not None:
            out["gather"] = gather
        print(json.dumps(out))
    if typo is not None:
        typo.close()
    eng.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
