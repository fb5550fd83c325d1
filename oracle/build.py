"""Build recipe for the CPU oracle (TEST INFRASTRUCTURE — see oracle/oracle.c).

`python oracle/build.py` compiles oracle/oracle.c with gcc into
oracle/_build/liboracle.so.  The reference is pure Python (no compilable C/C++
hot path), so there is no oracle/_ref build; the oracle is pinned instead by
golden vectors generated from the imported reference
(tests/golden/gen_golden.py).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "liboracle.so")


def build(force: bool = False) -> str:
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    cmd = ["gcc", "-O2", "-std=c11", "-ffp-contract=off", "-fno-fast-math", "-shared", "-fPIC",
           "-Wall", "-o", OUT, SRC, "-lm"]
    subprocess.run(cmd, check=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
