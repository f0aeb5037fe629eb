#!/usr/bin/env python3
"""Throughput of the alignment-scoring hot path on MI355X: scored (offset, mutant)
candidates per second, whole job, inputs resident in HBM.

    python bench.py --gpus N --steps K --warmup W [--config c2]

One process per GPU (for N > 1 launch through torch.distributed.run).  A step is one
pass of the full path -- fixed-point enumeration of every candidate, the float32-exact
resolve, class counts, result records -- over one batch of synthetic Seq2.  The batch
is uploaded once before the timed region.  Rank 0 prints one JSON line.  Before the W
warm-up steps the same workload runs untimed for --prewarm-s seconds (default 0.5): the GPU
reaches its sustained clocks only after tens of milliseconds of load, and the default
workload's step lasts a fifth of a millisecond.

Workloads (BASELINE.json configs):
  c2  Seq1 = 3000, 4 x Seq2 = 2000 per GPU      (default; the metric's configuration; each rank
                                                 holds its own batch: weak scaling)
  c1  the reference's own input.txt             (Seq1 = 1489; Seq2 = 138, 1202, 466, 377) per GPU
  c3  Seq1 = 10000, ONE list of --nseq2 Seq2 of 500..2000 characters (4096 in BASELINE), divided
      among the ranks by psa_partition as the reference divides its list between its two ranks
      (P/main.c:62-76): strong scaling, no data-path collective, records gathered to rank 0
  c4  Seq1 = 200000, Seq2 = 8000, offsets sharded over the ranks; every step ends with the
      all-gather of one record per rank (RCCL) and the max-loc combine, inside the timed region
Process groups at N > 1: the default group is gloo and carries only control traffic outside the
timed regions and off the data path (barriers, the broadcast of the c3 list, gathers of timings
and records); the RCCL communicator is a second group, created when c4 first needs it.

The default run (c2) also carries a `scale` block: c3 at full size and c4, a few steps each, on
the same N ranks, so that runs at N = 1, 2, 4, 8 give strong-scaling curves for both
(--no-scale skips it).

Parity gate: after the timed region every record of c1/c2, a seeded sample of 32 records of c3
and the record of c4 are compared with the CPU oracle (float32 answer, fixed-point enumeration,
class counts, and the rigorous bound that ties the two), and at N > 1 the N-rank records of
c3/c4 with what rank 0 computes alone.  On a mismatch the JSON says "parity": false and the
exit status is 1.  The oracle is used here as a checker and, in cpu_baseline, as the timed CPU
port; never on the measured path.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

# The library's context runs four streams side by side and needs a hardware queue for each.  HIP gives a
# process four (GPU_MAX_HW_QUEUES); RCCL's own streams take some of them, two of the context's streams
# then share a queue and run one after the other (measured with one rank: 0.219 instead of 0.172 ms per
# step; with eight queues 0.181).  Read by the HIP runtime when it starts, so set before torch is imported.
if int(os.environ.get("GPU_MAX_HW_QUEUES") or 4) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "parallel-sequence-alignment_amd"))

ALPHA20 = "ACDEFGHIKLMNPQRSTVWY"
WEIGHTS = np.array([2, 1.5, 1.1, 1.3], dtype=np.float32)  # the reference's input.txt weights

# 32-bit VALU peak of one MI355X (int32 or float32 add): 256 CUs x 4 SIMDs x 32 lanes per clock
# x 2.4 GHz (MI355X_MICROARCH.md: SIMD-32, a wave64 VALU instruction issues in 2 cycles, max
# clock 2400 MHz)
PEAK_INT32_LANE_OPS = 256 * 4 * 32 * 2.4e9
OPS_PER_CANDIDATE = 2  # one add (f[p] + g[q]) and one max per (offset, mutant) candidate
# The candidate loop forms the sums of 32 x 32 candidates with one v_mfma_f32_32x32x8_bf16_1k and takes
# their maxima with 8 v_max3_f32.  Either half bounds a tile at 32 cycles of one SIMD: the matrix core
# is occupied for 32 cycles by the instruction (tools/valu_bench10.hip mode 7: 32.0 measured back to
# back), and the vector issue port for 8 x 4 cycles by the maxima (MI355X_MICROARCH.md, cycle
# constants: vector-instruction issue cost).  Measured with nothing else in the loop: 37.4 cycles per
# tile at 8 waves per SIMD, 39.5 at 4 (mode 8; profiles/r02_valu_mfma_ceilings.txt) -- the matrix
# instruction also takes issue slots of the port.
SIMDS = 256 * 4
PEAK_CLOCK_HZ = 2.4e9
CYCLES_PER_TILE_BOUND = 32.0
BARE_LOOP_CYCLES_PER_TILE = {"4 waves per SIMD": 39.5, "8 waves per SIMD": 37.4}
# What this instruction mix can reach at best: a matrix instruction holds the SIMD's vector issue for 8 of its
# 32 cycles, issue costs add, "and an MFMA gap runs about max(32, their sum)" (MI355X_MICROARCH.md,
# per-instruction cycle constants, row 'vector-instruction ISSUE cost'): 8 v_max3_f32 x 4 + 8 = 40.
ISSUE_FLOOR_CYCLES_PER_TILE = 40.0
PEAK_CANDIDATES = SIMDS * PEAK_CLOCK_HZ * 1024.0 / CYCLES_PER_TILE_BOUND  # 7.86e13 candidates/s


def tiles_per_offset(L2):
    """32 x 32 tiles the candidate loop works through for one offset of a Seq2 of L2 characters: per
    level of 256 positions p the level's own 8 blocks of q meet 1, 2, .. 8 blocks of p (the last along
    the diagonal), every later block of q all 8; rows are padded to 64 positions and a last block of
    32 that holds no position is skipped."""
    nruns = (L2 + 63) // 64
    qend = 64 * nruns
    tiles = 0
    for q00 in range(0, qend, 256):
        for q in range(q00, qend, 64):
            rb = (q - q00) // 32
            tiles += min(rb + 1, 8)
            if q + 32 < L2:
                tiles += min(rb + 2, 8)
    return tiles


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


def c3_list(nseq2, lo=500, hi=2000):
    """The one seeded list of the c3 workload (the same on every rank)."""
    rng = np.random.default_rng(3)
    lens = [int(x) for x in rng.integers(lo, hi + 1, size=nseq2)]
    return synth(3, 10000, lens)


# ---------------------------------------------------------------------------------------------
# the CPU oracle: checker and timed CPU port (nothing else in this file touches oracle/)
# ---------------------------------------------------------------------------------------------
def _oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    return oracle_py


def check_record(oracle, seq1, seq2, table, r, stages, s_begin=0, s_end=None):
    """One record against the oracle; returns a list of complaints (empty = bit-exact)."""
    bad = []
    L2 = len(seq2)
    m = L2 - 2
    if r.status != 0:
        return ["status %d" % r.status]
    if stages & 2:
        o = oracle.fast(seq1, seq2, table, s_begin, s_end)
        if r.key() != o.as_tuple():
            bad.append("float32 answer %r != oracle %r" % (r.key(), o.as_tuple()))
        elif tuple(r.counts) != oracle.counts(seq1, seq2, r.offset, r.n, r.k) or r.mutant_num != o.mutant_num:
            bad.append("class counts / mutant number")
    if stages & 1:
        fr, itab, exact = oracle.fixed_table(table, L2)
        if (r.enum_frac_bits, r.enum_exact) != (fr, exact):
            bad.append("fixed-point scale %r != oracle %r" % ((r.enum_frac_bits, r.enum_exact), (fr, exact)))
        elif fr >= 0:
            e = oracle.enum_best(seq1, seq2, itab, s_begin, s_end)
            if r.enum_key() != (e.n, e.k, e.offset, e.score):
                bad.append("enumeration %r != oracle %r" % (r.enum_key(), (e.n, e.k, e.offset, e.score)))
            elif tuple(r.enum_counts) != oracle.counts(seq1, seq2, e.offset, e.n, e.k):
                bad.append("enumeration class counts")
            if stages & 2 and not bad:
                # the enumerated maximum and the float32 answer are the same quantity up to rounding:
                # |fl(score) - E 2^-r| <= b for one candidate, b = gamma_{m-1} sum|x| + m 2^-(r+1)
                u = 2.0 ** -24
                tmax = float(np.abs(table).max())
                b = (m - 1) * u / (1 - (m - 1) * u) * m * tmax + m * 2.0 ** -(fr + 1)
                if abs(float(r.score) - r.enum_score * 2.0 ** -fr) > b:
                    bad.append("float32 maximum and fixed-point maximum differ by more than the rounding bound")
                if r.enum_check not in (1, 2):  # 2: the replayed shortlist gave this very record; 1: it overflowed
                    bad.append("the library's own cross-check of the two stages says %d" % r.enum_check)
                ef = oracle.enum_score(seq1, seq2, itab, r.offset, r.n, r.k)
                if (r.enum_score - ef) * 2.0 ** -fr > 2 * b or ef > r.enum_score:
                    bad.append("the float32 winner's fixed-point score is not within 2b of the enumerated maximum")
    return bad


def check_records(seq1, seq2s, results, stages, ranges=None, threads=16):
    """Records against the oracle, in a thread pool (the oracle calls release the GIL)."""
    oracle = _oracle()
    table = oracle.comp_matrix(WEIGHTS)
    ranges = ranges or [(0, None)] * len(seq2s)
    with ThreadPoolExecutor(max_workers=max(1, min(threads, len(seq2s)))) as ex:
        outs = list(ex.map(lambda a: check_record(oracle, seq1, a[0], table, a[1], stages, a[2][0], a[2][1]),
                           zip(seq2s, results, ranges)))
    return [(i, b) for i, b in enumerate(outs) if b]


def cpu_baseline(seq1, seq2, budget_s=12.0):
    """Host-CPU baseline: the oracle's literal restatement of the reference's Sequential Solution
    (same triple loop, float32) on a bounded sample -- all offsets x a run of consecutive mutants of
    one Seq2 of the workload -- single-threaded like the reference, and on all host cores."""
    oracle_py = _oracle()
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


def measured_traffic(cfg):
    """Memory-side bytes per LAUNCH of the candidate-loop kernel as the TCC counters saw them in the committed
    rocprofv3 --pmc passes of this workload (profiles/r03_<cfg>_pmc.json, else round 2's file); None for
    workloads not profiled.  The kernel reads its rows with one dword per lane, which FETCH_SIZE counts at
    half (calibration: profiles/r01_fetch_size_calibration.txt), so traffic = 2 x FETCH_SIZE + WRITE_SIZE."""
    for rnd in ("r03", "r02"):
        path = os.path.join(ROOT, "profiles", "%s_%s_pmc.json" % (rnd, cfg))
        if not os.path.exists(path):
            continue
        with open(path) as f:
            kernels = json.load(f)["kernels"]
        for kname, c in kernels.items():
            if "psa_enum_wave_mfma<false>" in kname and "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                return (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0, "profiles/%s_%s_pmc.json" % (rnd, cfg)
    return None, None


# ---------------------------------------------------------------------------------------------
# one workload on the ranks of this job
# ---------------------------------------------------------------------------------------------
class Job:
    """Process-group plumbing shared by the workloads."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = 0 if args.share_gpu else int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus and self.world > 1:
            raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, self.world))
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (no CPU fallback)")
        torch.cuda.set_device(self.local_rank)
        self.dist = None
        if self.world > 1 or args.force_dist:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            # The default group is gloo in either case: barriers, the broadcast of the list and the gathers
            # of timings and records are control plane, off the data path and outside the timed regions.
            # With --backend nccl the RCCL communicator is a second group, created when the first workload
            # with a data-path collective (the long pair's all-gather) needs it: a live RCCL communicator
            # costs the other workloads' steps 7 % (one rank, C2: 0.166 ms per step without a process
            # group, 0.170 with a gloo group, 0.182 with an RCCL one), and they never use it.
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")  # one node: the loopback interface
            dist.init_process_group("gloo")
            self.dist = dist
        self.backend = args.backend
        self._data_group = None

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def gather_objects(self, obj):
        """obj of every rank, in rank order, on every rank (small control data, never the data path)."""
        if self.dist is None:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    def data_group(self):
        """The group of the data-path collective: RCCL (created on first use, by all ranks together) with
        --backend nccl, the default gloo group in a rehearsal."""
        if self.dist is None or self.backend != "nccl":
            return None
        if self._data_group is None:
            try:
                g = self.dist.new_group(backend="nccl")
                t = self.torch.zeros(1, device="cuda")
                self.dist.all_reduce(t, group=g)  # builds the communicator here, not inside a timed step
                self.torch.cuda.synchronize()
                self._data_group = g
            except Exception as e:  # no RCCL on this box: the rehearsal path, and the JSON says so
                sys.stderr.write("bench.py: RCCL group not available (%s); the long pair's all-gather runs over gloo\n" % e)
                self.backend = "gloo (RCCL group failed)"
                return None
        return self._data_group

    def collective_label(self):
        """What carried the long pair's exchanges: "rccl", "gloo" (rehearsal, or RCCL could not be created) or none."""
        if self._data_group is not None:
            return "rccl"
        return "gloo" if self.dist is not None else "none (one rank, no process group)"

    def allgather_bytes(self):
        from psa_amd import dist as pdist
        return pdist.allgather_bytes(self.data_group()) if self.dist is not None else None


def run_workload(job, ctx, args, cfg, steps, warmup, prewarm_s, nseq2, sync_steps, repeats=0, host_pass=False):
    """Runs one workload; returns a dict with the timing, the records and the parity verdict."""
    import psa_amd
    from psa_amd import dist as pdist
    rank, world = job.rank, job.world
    stages = {"both": psa_amd.STAGE_BOTH, "enumerate": psa_amd.STAGE_ENUMERATE,
              "resolve": psa_amd.STAGE_RESOLVE}[args.stages]
    mine = None          # c3: indices of the list this rank scores
    shard = None         # c4: this rank's offsets
    if cfg == "c2":
        seq1, seq2s = synth(2 + 1000 * rank, 3000, [2000] * 4)
        name, scaling, my_seq2 = "C2: Seq1=3000, 4 x Seq2=2000 per GPU", "weak", seq2s
    elif cfg == "c1":
        _, seq1, seq2s = psa_amd.readFromFile(os.path.join(ROOT, "tests", "golden", "inputs", "ref_input.txt"))
        name, scaling, my_seq2 = "C1: reference input.txt (Seq1=1489; Seq2=138,1202,466,377) per GPU", "weak", seq2s
    elif cfg == "c3":
        # rank 0 makes the list and hands it out, as the reference's rank 0 reads input.txt and sends
        # the sequences (P/main.c:43-76); every rank then computes the same split
        payload = [c3_list(nseq2, args.len_lo, args.len_hi) if rank == 0 else None]
        if job.dist is not None:
            job.dist.broadcast_object_list(payload, src=0)
        seq1, seq2s = payload[0]
        owner = psa_amd.partition([len(s) for s in seq2s], len(seq1), world, stages)
        mine = [i for i in range(len(seq2s)) if owner[i] == rank]
        my_seq2 = [seq2s[i] for i in mine]
        name = "C3: Seq1=10000, one list of %d Seq2 in [%d,%d] split over %d GPU(s) by psa_partition" % (
            nseq2, args.len_lo, args.len_hi, world)
        scaling = "strong"
    elif cfg == "c4":
        seq1, seq2s = synth(4, 200000, [8000])
        S = len(seq1) - len(seq2s[0]) + 3
        shard = pdist.offset_shard(S, rank, world)
        my_seq2 = seq2s
        name = "C4: Seq1=200000, Seq2=8000, offsets sharded over %d GPU(s), all-gather + max-loc combine per step" % world
        scaling = "strong"
    else:
        raise SystemExit("unknown config " + cfg)

    ctx.set_seq1(seq1)
    ctx.set_weights(WEIGHTS)
    batch = ctx.batch(my_seq2) if shard is None else ctx.batch_range(my_seq2[0], shard[0], shard[1])
    gather_cb = job.allgather_bytes() if shard is not None else None  # only the long pair exchanges anything
    combined = [None]

    def step_sync():
        if shard is None:
            batch.launch(stages)
        else:  # the collective and the combine belong to the step
            combined[0] = batch.run_allgather(world, gather_cb, stages)

    pipelined = shard is None and not sync_steps
    # clock ramp: the same work, untimed, until the GPU has been busy for prewarm_s seconds
    t_pre = time.perf_counter()
    batch.launch(stages)
    group = max(1, min(20, int(0.005 / max(time.perf_counter() - t_pre, 1e-6)))) if pipelined else 1
    while time.perf_counter() - t_pre < prewarm_s:
        for _ in range(group):
            batch.enqueue(stages)
        batch.wait()
    for _ in range(warmup):
        step_sync()
    job.barrier()
    t0 = time.perf_counter()
    enum_ms = gpu_ms = 0.0
    if not pipelined:
        for _ in range(steps):
            step_sync()  # returns when the step's kernels (and for c4 the exchange) have finished
            em, tm = batch.timing()
            enum_ms += em
            gpu_ms += tm
    else:
        # the K steps are enqueued back to back (they execute in order on the context's streams) and
        # awaited once: the host's launch latency overlaps the previous step instead of idling the GPU
        for _ in range(steps):
            batch.enqueue(stages)
        batch.wait()
        em, tm = batch.timing()  # HIP events of the last step
        enum_ms, gpu_ms = em * steps, tm * steps
    my_elapsed = time.perf_counter() - t0
    job.barrier()
    elapsed = time.perf_counter() - t0

    # The same timed region again, `repeats` times (reported beside the first, never instead of it): a region
    # of a few milliseconds says little about the box's run-to-run spread by itself.
    region_ms = []
    for _ in range(repeats if pipelined else 0):
        job.barrier()
        t1 = time.perf_counter()
        for _ in range(steps):
            batch.enqueue(stages)
        batch.wait()
        job.barrier()
        region_ms.append(1e3 * (time.perf_counter() - t1) / steps)

    # Duration of the candidate-loop kernel on its own.  In a pipelined timed region two of its
    # launches (consecutive steps) share the GPU, so the dispatch events of one launch span more
    # than its own work; here every step is awaited before the next is launched (its own
    # pre-pass, resolve chain and records still run beside it).  Outside the timed region.
    iso_enum_ms = iso_gpu_ms = 0.0
    n_iso = max(1, min(steps, 10)) if pipelined else 0
    for _ in range(n_iso):
        batch.launch(stages)
        em, tm = batch.timing()
        iso_enum_ms += em
        iso_gpu_ms += tm
    job.torch.cuda.synchronize()
    # the clock the chip holds under this kernel: a few more steps with the candidate loop's stamped
    # variant (its waves sum their shader cycles and 100 MHz ticks); no timing is taken from these
    clock_mhz = 0.0
    if stages & psa_amd.STAGE_ENUMERATE:
        ctx.clock_probe(True)
        t_probe = time.perf_counter()
        while True:
            batch.launch(stages)
            if time.perf_counter() - t_probe > 0.05:
                break
        clock_mhz, _ = ctx.clock_probe(False)

    res = batch.results()
    npasses = batch.passes()
    # c3 through the calls a host program makes: the rank's share from host buffers, in sub-batches that are
    # created (work lists built and uploaded), enqueued and waited for three deep -- against the resident step
    host_pass_ms = None
    if host_pass and cfg == "c3":
        job.barrier()
        t1 = time.perf_counter()
        sub, inflight, got = 256, [], []
        for i0 in range(0, len(my_seq2), sub):
            bb = ctx.batch(my_seq2[i0:i0 + sub])
            bb.enqueue(stages)
            inflight.append(bb)
            if len(inflight) == 3:
                b0 = inflight.pop(0)
                b0.wait()
                got += b0.results()
                b0.close()
        for b0 in inflight:
            b0.wait()
            got += b0.results()
            b0.close()
        host_pass_ms = 1e3 * (time.perf_counter() - t1)
        if [bytes(a) for a in got] != [bytes(a) for a in res]:
            raise SystemExit("c3 from host buffers in sub-batches: records differ from the resident batch's")
    bad_status = [r.status for r in res if r.status != 0]
    if bad_status:
        raise SystemExit("workload has invalid Seq2: %r" % bad_status)
    cand_rank = sum(r.candidates for r in res)
    per_rank = job.gather_objects((my_elapsed, elapsed, cand_rank))
    elapsed = max(p[1] for p in per_rank)
    cand_total = float(sum(p[2] for p in per_rank))

    # ---- parity gate -------------------------------------------------------------------------
    complaints = []
    checked = 0
    if cfg in ("c1", "c2"):
        mine_bad = check_records(seq1, my_seq2, res, stages)
        checked = len(res)
        for rk, b in enumerate(job.gather_objects(mine_bad)):
            complaints += ["rank %d Seq2 #%d: %s" % (rk, i, "; ".join(x)) for i, x in b]
        answer = res
    elif cfg == "c3":
        allrec = pdist.gather_results(res, mine, len(seq2s)) if job.dist is not None else res
        answer = allrec
        if rank == 0:
            rng = np.random.default_rng(33)
            sample = sorted(int(i) for i in rng.choice(len(seq2s), size=min(32, len(seq2s)), replace=False))
            bad = check_records(seq1, [seq2s[i] for i in sample], [allrec[i] for i in sample], stages)
            complaints += ["Seq2 #%d: %s" % (sample[i], "; ".join(x)) for i, x in bad]
            checked = len(sample)
            if sum(r.candidates for r in allrec) != sum((len(seq1) - len(s) + 3) * (len(s) * (len(s) - 1) // 2)
                                                        for s in seq2s):
                complaints.append("candidate count of the gathered records")
        if world > 1:
            # the N-rank records against what one rank computes for the whole list (rank 0, alone, untimed)
            batch.close()
            job.barrier()
            if rank == 0:
                whole = ctx.score_batch(seq2s, stages)
                diff = [i for i, (a, b) in enumerate(zip(allrec, whole))
                        if bytes(a) != bytes(b)]
                if diff:
                    complaints.append("%d of %d records differ between %d ranks and one rank (first: #%d)"
                                      % (len(diff), len(whole), world, diff[0]))
            job.barrier()
    else:  # c4
        answer = [combined[0] if combined[0] is not None else res[0]]
        if rank == 0:
            bad = check_records(seq1, my_seq2, answer, stages)
            complaints += ["combined record: %s" % "; ".join(x) for _, x in bad]
            checked = 1
        if world > 1:
            keys = job.gather_objects(bytes(answer[0]))
            if any(k != keys[0] for k in keys):
                complaints.append("the ranks returned different combined records")
            batch.close()
            job.barrier()
            if rank == 0:
                whole = ctx.score_batch(my_seq2, stages)[0]
                if (whole.key(), whole.enum_key(), tuple(whole.counts), whole.candidates) != \
                        (answer[0].key(), answer[0].enum_key(), tuple(answer[0].counts), answer[0].candidates):
                    complaints.append("combined record of %d shards differs from the one-rank record" % world)
            job.barrier()
    batch.close()
    parity_flags = job.gather_objects(not complaints)
    parity = all(parity_flags)

    out = {
        "name": name, "scaling": scaling, "elapsed": elapsed, "steps": steps, "cand_total": cand_total,
        "cand_rank": cand_rank, "per_rank": per_rank, "answer": answer, "parity": parity, "complaints": complaints,
        "parity_checked": checked, "enum_ms": enum_ms, "gpu_ms": gpu_ms, "iso_enum_ms": iso_enum_ms,
        "iso_gpu_ms": iso_gpu_ms, "n_iso": n_iso, "pipelined": pipelined, "stages": stages, "clock_mhz": clock_mhz,
        "seq1": seq1, "seq2s": seq2s, "my_seq2": my_seq2, "res": res, "shard": shard,
        "npasses": npasses, "region_ms": region_ms, "host_pass_ms": job.gather_objects(host_pass_ms),
    }
    return out


def mfma_loop():
    """The candidate loop runs on the matrix cores (float32 arithmetic, not switched off)."""
    return os.environ.get("PSA_ENUM_I32", "0") in ("", "0") and os.environ.get("PSA_ENUM_MFMA", "1") != "0"


def roofline_block(cfg, w):
    """Roofline of the candidate-loop kernel for this rank's share of the workload."""
    steps, cand_rank = w["steps"], w["cand_rank"]
    # this rank's average candidate-loop kernel time per step, launches not sharing the GPU
    kern_s = (w["iso_enum_ms"] / w["n_iso"] if w["n_iso"] else w["enum_ms"] / steps) / 1e3
    # slab rows the candidate loop reads: one 4-byte g per (offset, position); f of an offset is the
    # next offset's row (a chunk of 4 to 64 offsets stores one row more than it has offsets: not counted)
    if w["shard"] is None:
        offsets = [r.candidates / (len(s2) * (len(s2) - 1) // 2) for s2, r in zip(w["my_seq2"], w["res"])]
    else:
        offsets = [float(w["shard"][1] - w["shard"][0])]
    slab_bytes = sum(4.0 * len(s2) * n for s2, n in zip(w["my_seq2"], offsets))
    traffic, traffic_src = measured_traffic(cfg) if mfma_loop() else (None, None)
    launches = max(1, w.get("npasses") or 1)  # candidate-loop launches per step (slab passes)
    my_elapsed = w["per_rank"][0][0]
    lane_op_frac = cand_rank * OPS_PER_CANDIDATE / kern_s / PEAK_INT32_LANE_OPS
    mfma = mfma_loop()
    common = {
        "traffic": traffic,
        "traffic_unit": "bytes per launch from the committed rocprofv3 --pmc passes of this command (%s: 2 x FETCH_SIZE, "
                        "which counts the kernel's dword-per-lane reads at half, + WRITE_SIZE; reads served by the "
                        "Infinity Cache included), mean per launch of the kernel; a constant read from that file, not "
                        "measured by this run"
                        % traffic_src if traffic is not None else None,
        # measured bytes against the rows one launch covers (a step's rows / its launches): above 1 = re-reads
        "traffic_ratio": traffic / (slab_bytes / launches) if traffic is not None else None,
        "launches_per_step": launches,
        "kernel_ms": kern_s * 1e3,
        # the same launch inside the timed region, where it shares the GPU with the launch
        # of the step before or after it (null when steps are awaited one by one: no sharing)
        "kernel_ms_pipelined": w["enum_ms"] / steps if w["pipelined"] else None,
        "hbm_bytes_algorithmic": slab_bytes,
        "hbm_frac": slab_bytes / kern_s / 8.0e12,
        # the fraction this line carried before the sums moved to the matrix cores: 2 lane-ops (add +
        # max) per candidate against 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz; above 1 now, because
        # the adds no longer run on the vector ALU
        "lane_op_frac": lane_op_frac,
    }
    if not mfma:
        achieved = cand_rank * OPS_PER_CANDIDATE / kern_s
        return dict(common, **{
            "bound": "valu-32bit", "kernel": "psa_enum_wave", "achieved": achieved / 1e12,
            "peak": PEAK_INT32_LANE_OPS / 1e12, "unit": "Tlane-op/s", "frac": lane_op_frac,
            "step_frac": cand_rank * OPS_PER_CANDIDATE * steps / my_elapsed / PEAK_INT32_LANE_OPS,
            "note": "vector-ALU form of the candidate loop (PSA_ENUM_MFMA=0 or int32 arithmetic): 2 lane-ops (add + "
                    "max) per candidate against the 32-bit VALU peak"})
    tiles = sum(tiles_per_offset(len(s2)) * n for s2, n in zip(w["my_seq2"], offsets))
    achieved = cand_rank / kern_s
    clk = w["clock_mhz"] * 1e6
    cyc_per_tile = kern_s * clk * SIMDS / tiles if clk > 0 else None
    return dict(common, **{
        "bound": "mfma", "kernel": "psa_enum_wave_mfma",
        "achieved": achieved / 1e12, "peak": PEAK_CANDIDATES / 1e12, "unit": "Tcandidates/s",
        "frac": achieved / PEAK_CANDIDATES,
        # the same in the matrix instruction's own flops (32 x 32 x 8 x 2 per 1024 candidates): half the bf16
        # dense peak of the chip, because the K = 8 form does half the K = 16 form's products in the same 32 cycles
        "achieved_tflops": achieved * 16.0 / 1e12, "peak_tflops": PEAK_CANDIDATES * 16.0 / 1e12,
        # for reference only: against the chip's bf16 dense matrix peak (2.5 PFLOP/s, K = 16 forms); of the 16 flops
        # per candidate 6 multiply a piece by one and carry the sum, the rest multiply zeros
        "frac_of_bf16_dense_peak": achieved * 16.0 / 2.5e15,
        # the same with the whole step in the denominator (pre-pass, locate, records and every gap included)
        "step_frac": cand_rank * steps / my_elapsed / PEAK_CANDIDATES,
        # clock the chip held under this kernel (its waves' shader cycles / 100 MHz ticks, measured by this run)
        "clock_mhz_measured": w["clock_mhz"],
        "tiles_per_launch": tiles,
        "candidates_per_tile": cand_rank / tiles,
        "bound_cycles_per_tile": CYCLES_PER_TILE_BOUND,
        "bare_loop_cycles_per_tile": BARE_LOOP_CYCLES_PER_TILE,
        "cycles_per_tile_measured": cyc_per_tile,
        # the floor of this instruction mix by the guide's issue-cost rule (8 maxima x 4 cycles + the matrix
        # instruction's 8): the 32-cycle bound above is the ceiling to quote, this is what the loop can reach
        "issue_floor_cycles_per_tile": ISSUE_FLOOR_CYCLES_PER_TILE,
        "frac_of_issue_floor": ISSUE_FLOOR_CYCLES_PER_TILE / cyc_per_tile if cyc_per_tile else None,
        # tiles against the 32-cycle bound at the clock the kernel really ran at (<= 1 by construction):
        # the share of the time the matrix cores are busy, and equally the share the maxima alone need of the port
        "matrix_core_busy_frac": CYCLES_PER_TILE_BOUND / cyc_per_tile if cyc_per_tile else None,
        "frac_at_measured_clock": achieved / (PEAK_CANDIDATES * clk / PEAK_CLOCK_HZ) if clk > 0 else None,
        "note": "peak = 256 CUs x 4 SIMDs x 2.4 GHz x 1024 candidates per 32 cycles: one "
                "v_mfma_f32_32x32x8_bf16_1k occupies the SIMD's matrix core for 32 cycles, and the 8 v_max3_f32 that "
                "reduce its tile take 8 x 4 cycles of the vector issue port -- either bounds a tile at 32 (the bare "
                "loop measures 37.4-39.5: bare_loop_cycles_per_tile).  achieved = candidates per launch / "
                "HIP-event time of the candidate-loop kernel(s) of a step, events attached to the kernel dispatches "
                "on their stream, measured on steps launched one at a time after the timed region (inside it two "
                "launches share the GPU: kernel_ms_pipelined).  frac counts candidates, so tiles along the diagonal "
                "(half empty) and row padding weigh against it; matrix_core_busy_frac counts tiles at the measured "
                "clock.  The kernel reads 4 bytes of slab per (offset, position) against L2/2 candidates each "
                "(hbm_frac of the 8 TB/s peak): HBM does not bound it"})


def host_inclusive(ctx, seq1, seq2s, calls=20):
    """The same work through the calls a host program makes, host buffers in and records out
    (work lists rebuilt and uploaded per call): never the headline value."""
    import psa_amd
    ctx.set_seq1(seq1)
    ctx.set_weights(WEIGHTS)
    out = {}
    cand = sum((len(seq1) - len(s) + 3) * (len(s) * (len(s) - 1) // 2) for s in seq2s)
    for label, st in (("psa_score_batch_both_stages", psa_amd.STAGE_BOTH),
                      ("psa_score_batch_resolve_only", psa_amd.STAGE_RESOLVE)):
        ctx.score_batch(seq2s, st)
        t0 = time.perf_counter()
        for _ in range(calls):
            ctx.score_batch(seq2s, st)
        ms = 1e3 * (time.perf_counter() - t0) / calls
        out[label] = {"ms_per_call": ms, "candidates_per_s": cand / ms * 1e3}
    # a caller that streams: a FRESH batch per call (created from host buffers, uploaded, scored with both stages,
    # records read, released), the next one enqueued before the last is awaited -- three in flight
    for rep in range(2):
        queue, n = [], 10 * calls
        t0 = time.perf_counter()
        for _ in range(n):
            b = ctx.batch(seq2s)
            b.enqueue(psa_amd.STAGE_BOTH)
            queue.append(b)
            if len(queue) >= 3:
                o = queue.pop(0)
                o.wait()
                o.results()
                o.close()
        for o in queue:
            o.wait()
            o.close()
        ms = 1e3 * (time.perf_counter() - t0) / n
    out["fresh_batches_streamed_three_deep_both_stages"] = {"ms_per_batch": ms, "candidates_per_s": cand / ms * 1e3}
    os.environ["PSA_SHIM_QUIET"] = "1"
    table = psa_amd.createCompMatrix(WEIGHTS)
    psa_amd.calc_best_score(seq1, seq2s[0], table)
    t0 = time.perf_counter()
    for _ in range(calls):
        for s in seq2s:
            psa_amd.calc_best_score(seq1, s, table)
    out["psa_calc_best_score"] = {"ms_per_seq2": 1e3 * (time.perf_counter() - t0) / (calls * len(seq2s)),
                                  "note": "the drop-in for calc_best_score_CUDA, one Seq2 per call, resolve stage"}
    return out


def boundary_block(seq1, seq2s, records, calls=10):
    """The workload through the drop-in boundary with the candidate loop ON (P/cudaFunctions.cu:290-296: every
    call of the reference through the boundary runs its per-candidate kernel): psa_calc_best_score -- what the
    reference's main.c calls once per Seq2, P/main.c:81, :130 -- under PSA_STAGES=enumerate and =both, and the
    host program psa_host on a generated input.txt of this workload (process start, HIP start-up, file I/O and
    all).  The returned lines are compared with the records the parity gate has checked.  Never the headline."""
    import subprocess
    import tempfile
    import psa_amd
    os.environ["PSA_SHIM_QUIET"] = "1"
    table = psa_amd.createCompMatrix(WEIGHTS)
    want = [psa_amd.format_result(r) for r in records]
    cand = sum((len(seq1) - len(s) + 3) * (len(s) * (len(s) - 1) // 2) for s in seq2s)
    out = {"workload_candidates": cand, "lines_equal_parity_checked_records": True}
    for mode in ("resolve", "enumerate", "both"):
        os.environ["PSA_STAGES"] = mode
        lines = [psa_amd.calc_best_score(seq1, s, table) for s in seq2s]  # (first calls: buffers)
        t0 = time.perf_counter()
        for _ in range(calls):
            for s in seq2s:
                psa_amd.calc_best_score(seq1, s, table)
        ms = 1e3 * (time.perf_counter() - t0) / (calls * len(seq2s))
        out["psa_calc_best_score_PSA_STAGES=" + mode] = {
            "ms_per_seq2": ms, "candidates_per_s": (cand / len(seq2s)) / ms * 1e3 if mode != "resolve" else None}
        if lines != want:
            out["lines_equal_parity_checked_records"] = False
    os.environ.pop("PSA_STAGES", None)
    exe = os.path.join(ROOT, "parallel-sequence-alignment_amd", "psa_host")
    if os.path.exists(exe):
        with tempfile.TemporaryDirectory() as tmp:
            inp, outp = os.path.join(tmp, "input.txt"), os.path.join(tmp, "output.txt")
            with open(inp, "w") as f:
                f.write("%s\n%s\n%d\n%s\n" % (" ".join("%g" % x for x in WEIGHTS), seq1, len(seq2s), "\n".join(seq2s)))
            for mode in ("resolve", "enumerate"):
                t0 = time.perf_counter()
                r = subprocess.run([exe, "--stages=" + mode, inp, outp, "1"], capture_output=True, text=True, timeout=300)
                wall = time.perf_counter() - t0
                own = [l for l in r.stdout.splitlines() if l.startswith("Total Time: ")]
                ok = r.returncode == 0 and open(outp).read().splitlines(keepends=True) == want
                out["psa_host --stages=" + mode] = {
                    "wall_s": wall, "program_total_time_s": float(own[0].split()[-1]) if own else None,
                    "candidates_per_s_of_wall": cand / wall if mode == "enumerate" else None, "output_txt_equal": ok}
                if not ok:
                    out["psa_host --stages=" + mode]["failure"] = (r.returncode, r.stdout[-300:], r.stderr[-300:])
                    # a child killed by a signal before it wrote anything says nothing about the records;
                    # lines that differ do
                    if r.returncode >= 0:
                        out["lines_equal_parity_checked_records"] = False
    out["note"] = ("psa_calc_best_score = one Seq2 per call, host strings in, formatted line out (batch built, uploaded, "
                   "scored and released per call); psa_host = a fresh process per run: its wall time is dominated by "
                   "process and HIP start-up, program_total_time_s is what the program itself prints (P/main.c:97)")
    return out


def shortlist_stats(records):
    """How the records' answers are tied to the enumerated candidates: enum_check 2 = the float32 replay of the
    shortlisted candidates (those within the rounding window of the enumerated maximum) returned this very record,
    1 = the shortlist was given up (too many near-maximal candidates) and only the rounding bound was checked."""
    recs = [r for r in records if r is not None and r.status == 0 and r.candidates > 0]
    return {"records": len(recs), "replay_equals_answer": sum(1 for r in recs if r.enum_check == 2),
            "bound_only": sum(1 for r in recs if r.enum_check == 1),
            "other": sum(1 for r in recs if r.enum_check not in (1, 2))}


def scale_entry(w, world):
    """The strong-scaling figures of one workload of the scale block."""
    return {"workload": w["name"], "scaling": w["scaling"], "value": w["cand_total"] * w["steps"] / w["elapsed"],
            "unit": "candidates/s", "ms_per_step": 1e3 * w["elapsed"] / w["steps"], "steps": w["steps"],
            "n_ranks_seen": len(w["per_rank"]), "candidates_per_step": w["cand_total"],
            "per_rank_candidates": [p[2] for p in w["per_rank"]],
            # each rank's own time for its steps, before the closing barrier: the imbalance of the split
            "per_rank_ms_per_step": [1e3 * p[0] / w["steps"] for p in w["per_rank"]],
            "imbalance_max_over_mean": max(p[0] for p in w["per_rank"]) / (sum(p[0] for p in w["per_rank"]) / world),
            "parity": w["parity"], "parity_checked_records": w["parity_checked"],
            # the same list from host buffers: sub-batches of 256 Seq2 created, enqueued and awaited three deep
            # (work lists built and uploaded per sub-batch), each rank its share; ms for the whole share, once
            "host_inclusive_ms_per_rank": w["host_pass_ms"] if any(x is not None for x in w["host_pass_ms"]) else None,
            "shortlist": shortlist_stats(w["answer"]) if w["stages"] == 3 else None,
            "result_sample": {"n": w["answer"][0].n, "k": w["answer"][0].k, "offset": w["answer"][0].offset,
                              "score_bits": "0x%08x" % w["answer"][0].score_bits} if w["answer"] else None}


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
    ap.add_argument("--nseq2", type=int, default=4096, help="length of the Seq2 list of --config c3 (whole job)")
    ap.add_argument("--len-lo", type=int, default=500, help="shortest Seq2 of --config c3 (BASELINE: 500)")
    ap.add_argument("--len-hi", type=int, default=2000, help="longest Seq2 of --config c3 (BASELINE: 2000)")
    ap.add_argument("--stages", default="both", choices=["both", "enumerate", "resolve"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scale", action="store_true", help="skip the scale block (c3 and c4) of the default run")
    ap.add_argument("--no-host-inclusive", action="store_true")
    ap.add_argument("--sync-steps", action="store_true", help="wait for every step before launching the next")
    ap.add_argument("--repeats", type=int, default=10,
                    help="how often the timed region is run again after the reported one (c1/c2; median, min and max "
                         "of ms per step go into timing_repeats)")
    ap.add_argument("--no-boundary", action="store_true", help="skip the boundary block (drop-in shim and psa_host with the candidate loop on)")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the process group even for one rank (under torch.distributed.run): runs the "
                         "broadcast, gathers and the long pair's all-gather through the backend on a one-GPU box")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses GPU 0")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = {"c1": 200, "c2": 200, "c3": 3, "c4": 3}[args.config]

    job = Job(args)
    rank, world = job.rank, job.world
    import psa_amd
    ctx = psa_amd.Context(job.local_rank)
    w = run_workload(job, ctx, args, args.config, args.steps, args.warmup, args.prewarm_s, args.nseq2, args.sync_steps,
                     repeats=args.repeats, host_pass=args.config == "c3" and not args.no_host_inclusive)

    scale = None
    if args.config == "c2" and not args.no_scale:
        # strong scaling of the two sharded workloads on the same ranks: few steps, full sizes
        scale = {}
        for cfg, steps in (("c3", 2), ("c4", 3)):
            ws = run_workload(job, ctx, args, cfg, steps, 1, 0.0, args.nseq2, False,
                              host_pass=cfg == "c3" and not args.no_host_inclusive)
            scale[cfg] = scale_entry(ws, world)
            scale[cfg]["complaints"] = ws["complaints"][:5]
        scale["c4"]["collective"] = job.collective_label()
        scale["note"] = ("c3 = BASELINE's list of 4096 Seq2 divided among the ranks by psa_partition (no data-path "
                         "collective, records gathered to rank 0 and compared with a one-rank run); c4 = one pair, "
                         "offsets sharded, one all-gather of a 96-byte record per rank and the max-loc combine "
                         "inside every timed step; strong scaling: compare `value` across runs with different --gpus")

    parity = w["parity"] and (scale is None or all(scale[c]["parity"] for c in ("c3", "c4")))
    if rank == 0:
        # float32 throughout by default (the enumeration on integer-valued float32, the resolve on the reference's sums)
        enum_dtype = "i32+f32" if os.environ.get("PSA_ENUM_I32", "0") not in ("", "0") else "f32"
        res0 = w["answer"][0]
        out = {
            "metric": "scored (offset,mutant) candidates/sec",
            "value": w["cand_total"] * w["steps"] / w["elapsed"], "unit": "candidates/s", "n_gpus": world,
            "steps": w["steps"], "warmup": args.warmup,
            "ms_per_step": 1e3 * w["elapsed"] / w["steps"],
            # first kernel start to last kernel end of one step run on its own (HIP events)
            "gpu_ms_per_step_alone": (w["iso_gpu_ms"] / w["n_iso"]) if w["n_iso"] else w["gpu_ms"] / w["steps"],
            "higher_is_better": True,
            "scaling": w["scaling"],
            "vs_baseline": None, "dtype": enum_dtype, "data": "synthetic"
            if args.config != "c1" else "reference sample input",
            "parity": parity,
            "parity_detail": {"records_checked_against_oracle": w["parity_checked"] * (world if args.config in ("c1", "c2") else 1),
                              "complaints": w["complaints"][:5],
                              "shortlist": shortlist_stats(w["answer"]) if w["stages"] == 3 else None},
            "config": {"workload": w["name"], "stages": args.stages, "weights": [float(x) for x in WEIGHTS],
                       "collective": job.collective_label() if args.config == "c4" else None,
                       "exchanges_per_step": ((2 if w["stages"] & 1 else 1) if world > 1 or job.dist is not None else 0)
                       if args.config == "c4" else None,
                       "candidates_per_step": w["cand_total"], "prewarm_s": args.prewarm_s,
                       "per_rank_ms_per_step": [1e3 * p[0] / w["steps"] for p in w["per_rank"]]},
            "result_sample": {"n": res0.n, "k": res0.k, "offset": res0.offset,
                              "score_bits": "0x%08x" % res0.score_bits, "enum_n": res0.enum_n,
                              "enum_k": res0.enum_k, "enum_offset": res0.enum_offset},
        }
        if w["stages"] & psa_amd.STAGE_ENUMERATE and w["enum_ms"] > 0:
            out["roofline"] = roofline_block(args.config, w)
        if scale is not None:
            out["scale"] = scale
        if w["region_ms"]:
            rm = sorted(w["region_ms"])
            out["timing_repeats"] = {"regions": len(rm), "steps_per_region": w["steps"],
                                     "ms_per_step_median": rm[len(rm) // 2], "ms_per_step_min": rm[0], "ms_per_step_max": rm[-1],
                                     "note": "the timed region run again after the reported one; value and ms_per_step "
                                             "above are the first region's"}
        if not args.no_host_inclusive and args.config in ("c1", "c2"):
            out["host_inclusive"] = host_inclusive(ctx, w["seq1"], w["my_seq2"])
        if not args.no_boundary and args.config in ("c1", "c2") and w["parity"]:
            out["boundary"] = boundary_block(w["seq1"], w["my_seq2"], w["res"])
            parity = parity and out["boundary"]["lines_equal_parity_checked_records"]
            out["parity"] = parity
        if args.config == "c3" and any(x is not None for x in w["host_pass_ms"]):
            out["host_inclusive"] = {"ms_per_rank_whole_share": w["host_pass_ms"],
                                     "note": "the rank's share from host buffers in sub-batches of 256 Seq2, created, "
                                             "enqueued and awaited three deep; compare with ms_per_step (resident)"}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(w["seq1"], w["my_seq2"][0])
        print(json.dumps(out))
        sys.stdout.flush()
    if job.dist is not None:
        job.dist.barrier()
        job.dist.destroy_process_group()
    ctx.close()
    if not parity:
        sys.exit(1)


if __name__ == "__main__":
    main()
