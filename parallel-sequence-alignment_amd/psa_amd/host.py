"""The reference's host program (P/main.c) with one rank per GPU.

    python -m torch.distributed.run --nproc-per-node N -m psa_amd.host [--stages=...] [input.txt [output.txt]]
    python -m psa_amd.host [--stages=resolve|enumerate|both] [--sequential-format] [input.txt [output.txt]]   (one GPU)

Rank 0 reads input.txt (P/cFunctions.c:17-37) and sends Seq1, the weights and the Seq2
list to the other ranks (P/main.c:54-76 does this for exactly one other rank); every
rank scores the Seq2 assigned to it (cost-balanced instead of the reference's first
half / second half, P/cFunctions.c:49-52); rank 0 collects the results and writes one
line per Seq2 in input order to output.txt (P/main.c:79-94) and echoes them.

--stages (default: PSA_STAGES in the environment, else resolve) chooses where the answer comes from: the
float32 recurrence alone, the candidate loop over every (offset, mutant) candidate -- what every call of the
reference runs, P/cudaFunctions.cu:290-296 -- with the float32 replay of the candidates near its maximum, or
both with the two required to agree.  --sequential-format prints the line of the reference's Sequential
Solution (S/main.c:82) instead.
"""
import os
import sys
import time

import numpy as np

from . import Context, STAGE_RESOLVE, createCompMatrix, format_result, parse_stages, readFromFile, writeToFile
from . import dist as pdist


def gpu_scorer(seq1, weights, seq2s, device, stages=STAGE_RESOLVE):
    """Scores seq2s on one GPU; raises if none is usable (there is no CPU path)."""
    if not seq2s:
        return []
    ctx = Context(device)
    try:
        ctx.set_seq1(seq1)
        ctx.set_table(createCompMatrix(weights))
        return ctx.score_batch_answer(seq2s, stages)[0]
    finally:
        ctx.close()


def run(input_file_name="input.txt", output_file_name="output.txt", scorer=None, echo=True, stages=None,
        sequential_format=False):
    """Returns the result lines (on rank 0; None elsewhere).  `scorer(seq1, weights, seq2s, device)`
    replaces the GPU call in tests of the multi-rank protocol.  stages: STAGE_* bits (None: PSA_STAGES in
    the environment, else the resolve stage)."""
    if stages is None:
        stages = parse_stages(os.environ.get("PSA_STAGES"))
    t_start = time.time()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if scorer is None:
        scorer = gpu_scorer if stages == STAGE_RESOLVE else (lambda a, w, b, d: gpu_scorer(a, w, b, d, stages))
        scorer.uses_gpu = True
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            use_gpu = scorer is gpu_scorer or getattr(scorer, "uses_gpu", False)
            if use_gpu:
                torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl" if use_gpu else "gloo")
    payload = [None]
    if rank == 0:
        weights, seq1, seq2s = readFromFile(input_file_name)
        if echo and not sequential_format:
            print("num of seq2 = %d" % len(seq2s))  # P/main.c:51
        payload = [(weights.tolist(), seq1, seq2s)]
    if dist is not None:
        dist.broadcast_object_list(payload, src=0)
    weights, seq1, seq2s = payload[0]
    weights = np.array(weights, dtype=np.float32)
    len2 = [len(s) for s in seq2s]
    mine, _ = pdist.my_share(len2, len(seq1), rank, world, stages)
    local = scorer(seq1, weights, [seq2s[i] for i in mine], local_rank)
    results = pdist.gather_results(local, mine, len(seq2s)) if dist is not None else local
    lines = None
    if rank == 0:
        bad = [(i, r.status) for i, r in enumerate(results) if r.status != 0]
        if bad:
            raise SystemExit("invalid Seq2 (index, status): %r" % bad)
        if sequential_format:
            # S/main.c:82 (and :87); that program opens output.txt and leaves it empty (S/main.c:28-32)
            lines = ["MS(%d,%d)\tbest score: %1.4f\tbest offset: %d\n" % (r.n, r.k, r.score, r.offset) for r in results]
            open(output_file_name, "w").close()
            if echo:
                sys.stdout.write("".join(lines) + "Hello World!\n")
        else:
            lines = [format_result(r) for r in results]
            with open(output_file_name, "w") as f:
                for line in lines:
                    writeToFile(f, line)  # P/cFunctions.c:45
                    if echo:
                        sys.stdout.write(line)  # P/cudaFunctions.cu:321
            if echo:
                print("Total Time: %1.4f" % (time.time() - t_start))  # P/main.c:97
    if dist is not None:
        dist.barrier()
    return lines


if __name__ == "__main__":
    _stages, _seq, _pos = None, False, []
    _args = sys.argv[1:]
    while _args:
        a = _args.pop(0)
        if a == "--sequential-format":
            _seq = True
        elif a.startswith("--stages="):
            _stages = parse_stages(a[9:])
        elif a == "--stages" and _args:
            _stages = parse_stages(_args.pop(0))
        else:
            _pos.append(a)
    run(*_pos[:2], stages=_stages, sequential_format=_seq)
