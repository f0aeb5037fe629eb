"""Worker of tests/test_dist_cpu.py: runs under torch.distributed.run with the gloo backend
(no GPU).  The GPU call is replaced by an oracle-based scorer so that the multi-rank
protocol -- broadcast of the input, cost-balanced split, gather in input order, max-loc
combine of offset shards -- is what gets tested."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "parallel-sequence-alignment_amd"))
import oracle_py  # noqa: E402
import psa_amd  # noqa: E402
from psa_amd import dist as pdist  # noqa: E402
from psa_amd import host  # noqa: E402


def record_from_oracle(seq1, seq2, table, s_begin=0, s_end=None):
    o = oracle_py.fast(seq1, seq2, table, s_begin, s_end)
    r = psa_amd.Result()
    r.n, r.k, r.offset, r.mutant_num, r.score, r.score_bits = o.n, o.k, o.offset, o.mutant_num, o.score, o.score_bits
    L2 = len(seq2)
    S = len(seq1) - L2 + 3
    lo, hi = s_begin, S if s_end is None else s_end
    r.candidates = (hi - lo) * (L2 * (L2 - 1) // 2)
    fr, itab, exact = oracle_py.fixed_table(table, L2)
    e = oracle_py.enum_best(seq1, seq2, itab, lo, hi)
    r.enum_n, r.enum_k, r.enum_offset, r.enum_score = e.n, e.k, e.offset, e.score
    r.enum_frac_bits, r.enum_exact = fr, exact
    return r


def oracle_scorer(seq1, weights, seq2s, device):
    table = oracle_py.comp_matrix(weights)
    return [record_from_oracle(seq1, s2, table) for s2 in seq2s]


def main():
    import torch.distributed as dist
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # 1. the host program on the reference's sample (Seq2 of 60..54 characters; the 7th line is ignored)
    inp = os.path.join(ROOT, "tests", "golden", "inputs", "ref_s_input3.txt")
    lines = host.run(inp, os.path.join(out_dir, "output.txt"), scorer=oracle_scorer, echo=False)
    if rank == 0:
        with open(os.path.join(out_dir, "lines.txt"), "w") as f:
            f.writelines(lines)
    # 2. long pair: offset shards combine to the whole-range answer, ties to the smallest (n, k, offset)
    rng = np.random.default_rng(5)
    alpha = "ACDEFGHIKLMNPQRSTVWY"
    checks = []
    for seq1, seq2 in (("".join(alpha[i] for i in rng.integers(0, 20, 700)), "".join(alpha[i] for i in rng.integers(0, 20, 60))),
                       ("A" * 300, "A" * 40), ("AB" * 150, "BA" * 21)):
        table = oracle_py.comp_matrix(np.array([2, 1.5, 1.1, 1.3], dtype=np.float32))
        S = len(seq1) - len(seq2) + 3
        lo, hi = pdist.offset_shard(S, rank, world)
        mine = record_from_oracle(seq1, seq2, table, lo, hi)
        comb = pdist.combine_shards(mine)
        whole = record_from_oracle(seq1, seq2, table)
        checks.append(comb.key() == whole.key() and comb.enum_key() == whole.enum_key()
                      and comb.candidates == whole.candidates)
        # the all-gather callback that psa_score_range_allgather is given: every rank's bytes, in rank order
        got = pdist.allgather_bytes()(bytes(mine))
        parts = pdist.array_to_records(np.frombuffer(got, dtype=np.uint8).reshape(world, pdist.RECORD_BYTES))
        assert bytes(parts[rank]) == bytes(mine)
        assert psa_amd.combine_results(parts).key() == whole.key()
    with open(os.path.join(out_dir, "combine_rank%d.txt" % rank), "w") as f:
        f.write(" ".join("ok" if c else "BAD" for c in checks))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
