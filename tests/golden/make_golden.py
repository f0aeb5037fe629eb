#!/usr/bin/env python3
"""Regenerates the golden vectors under tests/golden/.

Runs ONLY in the build container, where /root/reference exists: it compiles the
reference's Sequential Solution (oracle/Makefile target `ref`, sources used where
they lie) and records what that binary prints -- n, k, offset and the raw float32
bits of the score -- for
  * the reference's own sample inputs (copied here as data files), and
  * seeded random inputs written in the reference's input grammar
    (4 weights, Seq1, count, Seq2 lines).
The output, golden.json, holds inputs and expected outputs only.

usage: python tests/golden/make_golden.py [--full]
  --full also runs the reference on its large input.txt (about 15 minutes); without it the
  case is taken from ref_input_reference_stdout.txt, the saved stdout of such a run.
"""
import json
import os
import random
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py  # noqa: E402

REF_S = "/root/reference/Sequential Solution"
REF_P = "/root/reference/Parallel Solution"

# reference sample inputs kept as data files: (source, name here)
SAMPLES = [
    (os.path.join(REF_S, "input1.txt"), "ref_s_input1.txt"),
    (os.path.join(REF_S, "input2.txt"), "ref_s_input2.txt"),
    (os.path.join(REF_S, "input3.txt"), "ref_s_input3.txt"),
    (os.path.join(REF_P, "input1.txt"), "ref_p_input1.txt"),
    (os.path.join(REF_S, "input.txt"), "ref_input.txt"),
]


def parse_stdout(text):
    """result lines of the compiled reference (see oracle/ref_hook.h) -> expected records"""
    res, last = [], None
    for line in text.splitlines():
        if line.startswith("MS("):
            last = line
        elif line.startswith("PSA_REF_BITS"):
            kv = dict(x.split("=") for x in line.split()[1:])
            res.append({"n": int(kv["n"]), "k": int(kv["k"]), "offset": int(kv["offset"]),
                        "bits": int(kv["bits"], 16), "text": last})
    return res


def parse_input(path):
    tok = open(path).read().split()
    weights = tok[0:4]
    seq1 = tok[4]
    cnt = int(tok[5])
    return weights, seq1, tok[6:6 + cnt]


ALPHABETS = ["AB", "NDEQ", "STAG", "HY", "ACDEFGHIKLMNPQRSTVWY", "ACDEFGHIKLMNPQRSTVWYBJOUXZ", "A"]
WEIGHTS = [
    "2 1.5 1.1 1.3", "1 1 1 1", "3 0 0 0", "1.1 1.1 1.3 1.3", "0.1 0.2 0.3 0.7",
    "1000.1 0.3 0.7 0.001", "2 1.5 1 1.25", "2.5 1.5 1.1 1.3", "0 0 0 0", "0.7 0.1 0.2 0.3",
]


def random_cases(seed=20251111):
    rng = random.Random(seed)
    files = []
    # tiny, tie-heavy
    for i in range(60):
        alpha = ALPHABETS[i % len(ALPHABETS)]
        w = WEIGHTS[i % len(WEIGHTS)]
        L1 = rng.randint(1, 22)
        seq1 = "".join(rng.choice(alpha) for _ in range(L1))
        seq2s = []
        for _ in range(6):
            L2 = rng.randint(2, min(L1 + 2, 12))
            seq2s.append("".join(rng.choice(alpha) for _ in range(L2)))
        files.append((w, seq1, seq2s))
    # medium: planted hits with two inserted letters, like the reference's samples
    for i in range(12):
        alpha = ALPHABETS[4 + (i % 2)]
        w = WEIGHTS[i % len(WEIGHTS)]
        L1 = rng.randint(120, 400)
        seq1 = "".join(rng.choice(alpha) for _ in range(L1))
        seq2s = []
        for j in range(4):
            L2 = rng.randint(20, 70)
            if j % 2 == 0:
                st = rng.randint(0, L1 - L2)
                core = list(seq1[st:st + L2 - 2])
                for _ in range(2):
                    core.insert(rng.randint(0, len(core)), rng.choice(alpha))
                # a few point changes
                for _ in range(rng.randint(0, 6)):
                    core[rng.randrange(len(core))] = rng.choice(alpha)
                seq2s.append("".join(core))
            else:
                seq2s.append("".join(rng.choice(alpha) for _ in range(L2)))
        files.append((w, seq1, seq2s))
    # period-2/3 repeats, L2 = L1-1, L2 = L1+2
    for i, (unit, L1) in enumerate([("AB", 30), ("ABC", 31), ("NDE", 24), ("HYH", 17), ("ST", 12)]):
        seq1 = (unit * 40)[:L1]
        seq2s = [(unit * 40)[1:L1], (unit * 40)[:L1 + 2], (unit[::-1] * 40)[:L1 - 3], seq1]
        files.append((WEIGHTS[(3 * i) % len(WEIGHTS)], seq1, seq2s))
    return files


def main():
    full = "--full" in sys.argv
    oracle_py.build()
    os.makedirs(os.path.join(HERE, "inputs"), exist_ok=True)
    cases = []
    for src, name in SAMPLES:
        dst = os.path.join(HERE, "inputs", name)
        shutil.copyfile(src, dst)
        weights, seq1, seq2s = parse_input(dst)
        if name == "ref_input.txt" and not full:
            res = parse_stdout(open(os.path.join(HERE, "ref_input_reference_stdout.txt")).read())
        else:
            res = oracle_py.run_reference(dst)
            if name == "ref_input.txt":
                env = dict(os.environ, PSA_REF_INPUT=dst)
                import subprocess
                open(os.path.join(HERE, "ref_input_reference_stdout.txt"), "w").write(
                    subprocess.run([oracle_py.REF_BIN], env=env, cwd="/tmp", capture_output=True, text=True).stdout)
        assert len(res) == len(seq2s)
        cases.append({"name": name, "source": "reference sample input", "weights": weights, "seq1": seq1,
                      "seq2": seq2s, "expected": res})
    tmp = tempfile.mkdtemp()
    for i, (w, seq1, seq2s) in enumerate(random_cases()):
        path = os.path.join(tmp, "rand%03d.txt" % i)
        with open(path, "w") as f:
            f.write(w + "\n" + seq1 + "\n%d\n" % len(seq2s) + "\n".join(seq2s) + "\n")
        res = oracle_py.run_reference(path)
        assert len(res) == len(seq2s)
        cases.append({"name": "rand%03d" % i, "source": "seeded random, run through the compiled reference",
                      "weights": w.split(), "seq1": seq1, "seq2": seq2s, "expected": res})
    shutil.rmtree(tmp)
    out = os.path.join(HERE, "golden.json")
    json.dump({"generator": "tests/golden/make_golden.py", "cases": cases}, open(out, "w"), indent=0)
    print("wrote", out, len(cases), "files,", sum(len(c["seq2"]) for c in cases), "Seq2 cases")


if __name__ == "__main__":
    main()
