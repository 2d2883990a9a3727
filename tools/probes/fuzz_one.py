"""Replays ONE fuzz case (family, seed) of tools/fuzz_parity.py, optionally under a list of environment variants in child processes:
    python tools/probes/fuzz_one.py case_many_tiles 20260988784363 "" ISX_ROLL=0 ISX_FEED_FUSE=0"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools"))

if __name__ == "__main__":
    fam, seed = sys.argv[1], int(sys.argv[2])
    variants = sys.argv[3:]
    if not variants:
        import numpy as np
        import fuzz_parity
        fuzz_parity.G.load()
        try:
            r = getattr(fuzz_parity, fam)(np.random.default_rng(seed))
            print("PASS", r)
        except AssertionError as e:
            print("MISMATCH", str(e)[:600].replace("\n", " "))
        sys.exit(0)
    for v in variants:
        env = dict(os.environ)
        for kv in v.split(","):
            if kv:
                k, val = kv.split("=")
                env[k] = val
        out = subprocess.run([sys.executable, os.path.abspath(__file__), fam, str(seed)], env=env, capture_output=True, text=True)
        print("[%s] %s" % (v, (out.stdout.strip().splitlines() or [out.stderr.strip()[-300:]])[-1]))
