#!/usr/bin/env python3
"""Throughput of the alignment-scoring hot path on MI355X: scored (offset, mutant)
candidates per second, whole job, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W [--config c2]

One process per GPU (for N > 1 launch through torch.distributed.run).  A step is one
pass of the full path -- fixed-point enumeration of every candidate, the float32-exact
resolve, class counts, result records -- over one batch of synthetic Seq2.  The batch
is uploaded once before the timed region.  Ranks hold independent batches (no data-path
collective); rank 0 prints one JSON line.  Before the W warm-up steps the same workload runs
untimed for --prewarm-s seconds (default 0.5): the GPU reaches its sustained clocks only after
tens of milliseconds of load, and the default workload's step lasts a fifth of a millisecond.

Workloads (BASELINE.json configs):
  c2  Seq1 = 3000, 4 x Seq2 = 2000              (default; the metric's configuration)
  c1  the reference's own input.txt             (Seq1 = 1489; Seq2 = 138, 1202, 466, 377)
  c3  Seq1 = 10000, --nseq2 Seq2 of 500..2000   (4096 in BASELINE; default 256 here per rank)
  c4  Seq1 = 200000, Seq2 = 8000, offsets sharded over the ranks
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "parallel-sequence-alignment_amd"))

ALPHA20 = "ACDEFGHIKLMNPQRSTVWY"
WEIGHTS = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)  # the reference's input.txt weights

# 32-bit VALU peak of one MI355X (int32 or float32 add): 256 CUs x 4 SIMDs x 32 lanes per clock
# x 2.4 GHz (MI355X_MICROARCH.md: SIMD-32, a wave64 VALU instruction issues in 2 cycles, max
# clock 2400 MHz)
PEAK_INT32_LANE_OPS = 256 * 4 * 32 * 2.4e9
OPS_PER_CANDIDATE = 2  # one add (f[p] + g[q]) and one max per (offset, mutant) candidate


def synth(seed, L1, L2s):
    """Seeded residues, uniform over the 20 amino-acid letters; every other Seq2 is a window of
    Seq1 with two inserted letters and ~10 % substitutions (a planted hit, like the reference's inputs)."""
    rng = np.random.default_rng(seed)
    pick = lambda n: "".join(ALPHA20[i] for i in rng.integers(0, 20, size=n))
    seq1 = pick(L1)
    out = []
    for i, L2 in enumerate(L2s):
        if i % 2 == 0 and L2 > 4:
            st = int(rng.integers(0, L1 - (L2 - 2) + 1))
            core = list(seq1[st:st + L2 - 2])
            for _ in range(2):
                core.insert(int(rng.integers(0, len(core) + 1)), ALPHA20[int(rng.integers(0, 20))])
            for _ in range(max(1, L2 // 10)):
                core[int(rng.integers(0, len(core)))] = ALPHA20[int(rng.integers(0, 20))]
            out.append("".join(core))
        else:
            out.append(pick(L2))
    return seq1, out


def make_workload(cfg, rank, world, nseq2):
    """Returns (name, seq1, seq2 list, offset range or None, description dict)."""
    if cfg == "c2":
        seq1, seq2s = synth(2 + 1000 * rank, 3000, [2000] * 4)
        return "C2: Seq1=3000, 4 x Seq2=2000 per GPU", seq1, seq2s, None
    if cfg == "c1":
        import psa_amd
        w, seq1, seq2s = psa_amd.readFromFile(os.path.join(ROOT, "tests", "golden", "inputs", "ref_input.txt"))
        return "C1: reference input.txt (Seq1=1489; Seq2=138,1202,466,377) per GPU", seq1, seq2s, None
    if cfg == "c3":
        rng = np.random.default_rng(3 + 1000 * rank)
        lens = [int(x) for x in rng.integers(500, 2001, size=nseq2)]
        seq1, seq2s = synth(3 + 1000 * rank, 10000, lens)
        return "C3: Seq1=10000, %d x Seq2 in [500,2000] per GPU" % nseq2, seq1, seq2s, None
    if cfg == "c4":
        seq1, seq2s = synth(4, 200000, [8000])
        S = 200000 - 8000 + 3
        lo, hi = S * rank // world, S * (rank + 1) // world
        return "C4: Seq1=200000, Seq2=8000, offsets sharded over %d GPU(s)" % world, seq1, seq2s, (lo, hi)
    raise SystemExit("unknown config " + cfg)


def cpu_baseline(seq1, seq2, budget_s=12.0):
    """Host-CPU baseline: the oracle's literal restatement of the reference's Sequential Solution
    (same triple loop, float32) on a bounded sample -- all offsets x a run of consecutive mutants of
    one Seq2 of the workload -- single-threaded like the reference, and on all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    table = oracle_py.comp_matrix(WEIGHTS)
    L1, L2 = len(seq1), len(seq2)
    S, M = L1 - L2 + 3, L2 * (L2 - 1) // 2
    # calibrate on a few mutants, then size the samples to the time budget
    t0 = time.perf_counter()
    probe = min(M, 40)
    oracle_py.brute(seq1, seq2, table, 1, 1 + probe)
    per_mutant = max((time.perf_counter() - t0) / probe, 1e-7)
    n1 = int(max(1, min(M, budget_s / per_mutant)))
    t0 = time.perf_counter()
    oracle_py.brute(seq1, seq2, table, 1, 1 + n1)
    t1 = time.perf_counter() - t0
    single = S * n1 / t1
    cores = max(1, min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))))
    nm = int(max(1, min(M, n1 * cores)))
    t0 = time.perf_counter()
    oracle_py.brute(seq1, seq2, table, 1, 1 + nm, threads=cores)
    tm = time.perf_counter() - t0
    multi = S * nm / tm
    return {"value": multi, "unit": "candidates/s", "cores": cores, "kind": "port",
            "single_core_value": single,
            "sample": "all %d offsets x mutants 1..%d (1 thread, %.1f s) / 1..%d (%d threads, %.1f s) of one "
                      "Seq2 of length %d vs Seq1 of length %d; O(L2) adds per candidate as in the reference"
                      % (S, n1, t1, nm, cores, tm, L2, L1)}


def measured_traffic(cfg, f_bytes):
    """Memory-side bytes per launch of the candidate-loop kernel as the TCC counters saw them in the
    committed rocprofv3 run of this workload (profiles/r01_c2_pmc.json); None for workloads not
    profiled.  Calibration (profiles/r01_fetch_size_calibration.txt): the kernel's scalar reads are
    counted in full, its dword-per-lane reads (f of an offset: the next offset's row of g, f_bytes) at half."""
    path = os.path.join(ROOT, "profiles", "r01_c2_pmc.json")
    if cfg != "c2" or not os.path.exists(path):
        return None
    with open(path) as f:
        kernels = json.load(f)["kernels"]
    for name, c in kernels.items():
        if "psa_enum_wave" in name and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            return (c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0 + 0.5 * f_bytes
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed steps (default: 200 for c1/c2, whose steps take a fraction of a millisecond; 5 for c3, 3 for c4)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prewarm-s", type=float, default=0.5,
                    help="seconds of the same workload run before the warm-up steps, untimed: the GPU reaches its "
                         "sustained clocks only after tens of milliseconds of load (a 5 ms timed region on an idle GPU "
                         "measures the ramp: 0.27 instead of 0.22 ms per C2 step)")
    ap.add_argument("--config", default="c2", choices=["c1", "c2", "c3", "c4"])
    ap.add_argument("--nseq2", type=int, default=256, help="Seq2 per GPU for --config c3")
    ap.add_argument("--stages", default="both", choices=["both", "enumerate", "resolve"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--sync-steps", action="store_true", help="wait for every step before launching the next")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses GPU 0")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {"c1": 200, "c2": 200, "c3": 5, "c4": 3}[args.config]

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:  # gloo: rehearsal of the N > 1 logic where the ranks cannot each have a GPU of their own
            dist.init_process_group(args.backend)
    cdev = "cuda" if args.backend == "nccl" else "cpu"  # where the few collective payloads live

    import psa_amd
    name, seq1, seq2s, rng = make_workload(args.config, rank, world, args.nseq2)
    ctx = psa_amd.Context(local_rank)
    ctx.set_seq1(seq1)
    ctx.set_weights(WEIGHTS)
    batch = ctx.batch(seq2s) if rng is None else ctx.batch_range(seq2s[0], rng[0], rng[1])
    stages = {"both": psa_amd.STAGE_BOTH, "enumerate": psa_amd.STAGE_ENUMERATE,
              "resolve": psa_amd.STAGE_RESOLVE}[args.stages]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # clock ramp: the same work, untimed, until the GPU has been busy for --prewarm-s seconds
    t_pre = time.perf_counter()
    batch.launch(stages)
    group = 1 if args.sync_steps else max(1, min(20, int(0.005 / max(time.perf_counter() - t_pre, 1e-6))))
    while time.perf_counter() - t_pre < args.prewarm_s:
        for _ in range(group):
            batch.enqueue(stages)
        batch.wait()
    for _ in range(args.warmup):
        batch.launch(stages)
    barrier()
    t0 = time.perf_counter()
    enum_ms = gpu_ms = 0.0
    if args.sync_steps:
        for _ in range(args.steps):
            batch.launch(stages)  # returns when the step's kernels and the record copy have finished
            em, tm = batch.timing()
            enum_ms += em
            gpu_ms += tm
    else:
        # the K steps are enqueued back to back (they execute in order on the context's streams) and
        # awaited once: the host's launch latency overlaps the previous step instead of idling the GPU
        for _ in range(args.steps):
            batch.enqueue(stages)
        batch.wait()
        em, tm = batch.timing()  # HIP events of the last step
        enum_ms, gpu_ms = em * args.steps, tm * args.steps
    barrier()
    elapsed = time.perf_counter() - t0

    # Duration of the candidate-loop kernel on its own.  In the timed region above two of its
    # launches (consecutive steps) share the GPU, so the dispatch events of one launch span more
    # than its own work; here every step is awaited before the next is launched (its own
    # pre-pass, resolve chain and records still run beside it).  Outside the timed region.
    iso_enum_ms = iso_gpu_ms = 0.0
    n_iso = 0 if args.sync_steps else max(1, min(args.steps, 10))
    for _ in range(n_iso):
        batch.launch(stages)
        em, tm = batch.timing()
        iso_enum_ms += em
        iso_gpu_ms += tm
    torch.cuda.synchronize()

    res = batch.results()
    bad = [r.status for r in res if r.status != 0]
    if bad:
        raise SystemExit("workload has invalid Seq2: %r" % bad)
    cand_rank = sum(r.candidates for r in res)
    if dist is not None:
        t = torch.tensor([elapsed, float(cand_rank)], dtype=torch.float64, device=cdev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, cand_total = float(tmax[0]), float(tsum[1])
        if args.config == "c4":  # one long pair: the job's answer is the max-loc combine of the shards
            rec = np.frombuffer(bytes(res[0]), dtype=np.uint8).copy()
            mine = torch.from_numpy(rec).to(cdev)
            allrec = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allrec, mine)
            parts = [psa_amd.Result.from_buffer_copy(bytes(x.cpu().numpy())) for x in allrec]
            res = [psa_amd.combine_results(parts)]
    else:
        cand_total = float(cand_rank)

    if rank == 0:
        # float32 throughout by default (the enumeration on integer-valued float32, the resolve on the reference's sums)
        enum_dtype = "i32+f32" if os.environ.get("PSA_ENUM_I32", "0") not in ("", "0") else "f32"
        value = cand_total * args.steps / elapsed
        out = {
            "metric": "scored (offset,mutant) candidates/sec",
            "value": value, "unit": "candidates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            # first kernel start to last kernel end of one step run on its own (HIP events)
            "gpu_ms_per_step_alone": (iso_gpu_ms / n_iso) if n_iso else gpu_ms / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if args.config == "c4" else "weak",
            "vs_baseline": None, "dtype": enum_dtype, "data": "synthetic"
            if args.config != "c1" else "reference sample input",
            "config": {"workload": name, "stages": args.stages, "weights": [float(x) for x in WEIGHTS],
                       "candidates_per_step": cand_total, "seed": "2 + 1000*rank", "prewarm_s": args.prewarm_s},
            "result_sample": {"n": res[0].n, "k": res[0].k, "offset": res[0].offset,
                              "score_bits": "0x%08x" % res[0].score_bits, "enum_n": res[0].enum_n,
                              "enum_k": res[0].enum_k, "enum_offset": res[0].enum_offset},
        }
        if stages & psa_amd.STAGE_ENUMERATE and enum_ms > 0:
            # this rank's average candidate-loop kernel time per step, launches not sharing the GPU
            kern_s = (iso_enum_ms / n_iso if n_iso else enum_ms / args.steps) / 1e3
            achieved = cand_rank * OPS_PER_CANDIDATE / kern_s
            # slab rows the candidate loop reads: one 4-byte g per (offset, position); f of an offset is the
            # next offset's row (a chunk of 4 to 64 offsets stores one row more than it has offsets: not counted)
            slab_bytes = sum(4.0 * len(s2) * (r.candidates / (len(s2) * (len(s2) - 1) // 2))
                             for s2, r in zip(seq2s, res) if r.candidates > 0) if rng is None else \
                4.0 * len(seq2s[0]) * (rng[1] - rng[0])
            out["roofline"] = {
                "bound": "valu-32bit", "kernel": "psa_enum_wave",
                "achieved": achieved / 1e12, "peak": PEAK_INT32_LANE_OPS / 1e12, "unit": "Tlane-op/s",
                "frac": achieved / PEAK_INT32_LANE_OPS, "traffic": measured_traffic(args.config, slab_bytes),
                "traffic_unit": "bytes per launch (FETCH_SIZE + WRITE_SIZE of separate rocprofv3 --pmc runs, plus the half of "
                                "the per-lane row reads that FETCH_SIZE does not count; Infinity-Cache hits included)",
                "kernel_ms": kern_s * 1e3,
                # the same launch inside the timed region, where it shares the GPU with the launch
                # of the step before or after it (null with --sync-steps: no sharing)
                "kernel_ms_pipelined": None if args.sync_steps else enum_ms / args.steps,
                # the same fraction with the whole step in the denominator (pre-pass, locate,
                # records and every gap included)
                "step_frac": cand_rank * OPS_PER_CANDIDATE * args.steps / elapsed / PEAK_INT32_LANE_OPS,
                "hbm_bytes_algorithmic": slab_bytes,
                "hbm_frac": slab_bytes / kern_s / 8.0e12,
                "note": "2 lane-ops (add + max) per candidate x candidates per step / HIP-event time of the "
                        "candidate-loop kernel(s) of a step, events attached to the kernel dispatches on "
                        "their stream, measured on steps launched one at a time after the timed region "
                        "(inside it two launches share the GPU: kernel_ms_pipelined); the kernel reads 4 bytes "
                        "of slab per (offset, position) against L2/2 candidates each (hbm_frac of the 8 TB/s "
                        "peak), so 32-bit VALU issue bounds it, not HBM or MFMA",
            }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(seq1, seq2s[0])
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
