"""The reference's host program (P/main.c) with one rank per GPU.

    python -m torch.distributed.run --nproc-per-node N -m psa_amd.host [input.txt [output.txt]]
    python -m psa_amd.host [input.txt [output.txt]]          (one GPU)

Rank 0 reads input.txt (P/cFunctions.c:17-37) and sends Seq1, the weights and the Seq2
list to the other ranks (P/main.c:54-76 does this for exactly one other rank); every
rank scores the Seq2 assigned to it (cost-balanced instead of the reference's first
half / second half, P/cFunctions.c:49-52); rank 0 collects the results and writes one
line per Seq2 in input order to output.txt (P/main.c:79-94) and echoes them.
"""
import os
import sys
import time

import numpy as np

from . import Context, STAGE_RESOLVE, createCompMatrix, format_result, readFromFile, writeToFile
from . import dist as pdist


def gpu_scorer(seq1, weights, seq2s, device):
    """Scores seq2s on one GPU; raises if none is usable (there is no CPU path)."""
    if not seq2s:
        return []
    ctx = Context(device)
    try:
        ctx.set_seq1(seq1)
        ctx.set_table(createCompMatrix(weights))
        return ctx.score_batch(seq2s, STAGE_RESOLVE)
    finally:
        ctx.close()


def run(input_file_name="input.txt", output_file_name="output.txt", scorer=None, echo=True):
    """Returns the result lines (on rank 0; None elsewhere).  `scorer(seq1, weights, seq2s, device)`
    replaces the GPU call in tests of the multi-rank protocol."""
    t_start = time.time()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    scorer = scorer or gpu_scorer
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            use_gpu = scorer is gpu_scorer
            if use_gpu:
                torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl" if use_gpu else "gloo")
    payload = [None]
    if rank == 0:
        weights, seq1, seq2s = readFromFile(input_file_name)
        if echo:
            print("num of seq2 = %d" % len(seq2s))  # P/main.c:51
        payload = [(weights.tolist(), seq1, seq2s)]
    if dist is not None:
        dist.broadcast_object_list(payload, src=0)
    weights, seq1, seq2s = payload[0]
    weights = np.array(weights, dtype=np.float32)
    len2 = [len(s) for s in seq2s]
    mine, _ = pdist.my_share(len2, len(seq1), rank, world, STAGE_RESOLVE)
    local = scorer(seq1, weights, [seq2s[i] for i in mine], local_rank)
    results = pdist.gather_results(local, mine, len(seq2s)) if dist is not None else local
    lines = None
    if rank == 0:
        bad = [(i, r.status) for i, r in enumerate(results) if r.status != 0]
        if bad:
            raise SystemExit("invalid Seq2 (index, status): %r" % bad)
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
    run(*(sys.argv[1:3]))
