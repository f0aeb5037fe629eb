import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "parallel-sequence-alignment_amd"))
sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def parse_weights(tokens):
    """Decimal text -> float32 the way fscanf("%f") does it (P/cFunctions.c:25-26)."""
    return np.array([np.float32(t) for t in tokens], dtype=np.float32)


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "golden.json")) as f:
        return json.load(f)["cases"]


@pytest.fixture(scope="session")
def oracle():
    import oracle_py
    oracle_py.build()
    return oracle_py


ALPHA20 = "ACDEFGHIKLMNPQRSTVWY"


def synth_seq(rng, n, alphabet=ALPHA20):
    idx = rng.integers(0, len(alphabet), size=n)
    return "".join(alphabet[i] for i in idx)


def synth_problem(seed, L1, L2s, alphabet=ALPHA20, plant=True):
    """Seeded synthetic problem: Seq1 plus Seq2s; every other Seq2 is a window of
    Seq1 with two inserted letters and a few substitutions."""
    rng = np.random.default_rng(seed)
    seq1 = synth_seq(rng, L1, alphabet)
    seq2s = []
    for i, L2 in enumerate(L2s):
        if plant and i % 2 == 0 and L2 - 2 <= L1 and L2 > 4:
            st = int(rng.integers(0, L1 - (L2 - 2) + 1))
            core = list(seq1[st:st + L2 - 2])
            for _ in range(2):
                core.insert(int(rng.integers(0, len(core) + 1)), alphabet[int(rng.integers(0, len(alphabet)))])
            for _ in range(max(1, L2 // 10)):
                core[int(rng.integers(0, len(core)))] = alphabet[int(rng.integers(0, len(alphabet)))]
            seq2s.append("".join(core))
        else:
            seq2s.append(synth_seq(rng, L2, alphabet))
    return seq1, seq2s
