"""The one-process-per-GPU path on CPU: world_size 2, gloo backend, the GPU call replaced by
an oracle-based scorer (tests/dist_worker.py)."""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import parse_weights

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_ranks_gloo(tmp_path, golden, oracle):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    case = [c for c in golden if c["name"] == "ref_s_input3.txt"][0]
    lines = open(tmp_path / "lines.txt").read().splitlines(keepends=True)
    assert lines == open(tmp_path / "output.txt").read().splitlines(keepends=True)
    assert len(lines) == 6  # the input announces 6 Seq2 and carries a 7th line, which is ignored
    for line, seq2, e in zip(lines, case["seq2"], case["expected"]):
        score = np.array([e["bits"]], dtype=np.uint32).view(np.float32)[0]
        mnum = oracle.mutant_number(e["n"], e["k"], len(seq2)) - 1
        assert line == "mutant num: %d, MS(%d,%d), score: %1.2f, offset: %d\n" % (mnum, e["n"], e["k"], score,
                                                                                   e["offset"])
    for rank in (0, 1):
        assert open(tmp_path / ("combine_rank%d.txt" % rank)).read() == "ok ok ok"


def test_share_is_a_partition():
    sys.path.insert(0, os.path.join(ROOT, "parallel-sequence-alignment_amd"))
    from psa_amd import dist as pdist
    len2 = [2000, 500, 1999, 700, 1200, 1500, 900, 64]
    seen = []
    for rank in range(4):
        mine, owner = pdist.my_share(len2, 10000, rank, 4)
        seen += mine
    assert sorted(seen) == list(range(len(len2)))
    assert pdist.offset_shard(1003, 0, 4)[0] == 0 and pdist.offset_shard(1003, 3, 4)[1] == 1003
    spans = [pdist.offset_shard(1003, r, 4) for r in range(4)]
    assert all(spans[i][1] == spans[i + 1][0] for i in range(3))
