#!/usr/bin/env python3
"""TEST INFRASTRUCTURE.  Regenerates tests/golden/*.npz from the compiled reference.

Runs only in the build container (needs /root/reference): builds oracle/_ref/ref_dump with
oracle/Makefile, runs it, and packs the .npy arrays it writes into one compressed .npz per scene plus
kats.npz (known-answer vectors for the leaf functions).  The GPU box only ever reads the committed
.npz files.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "..", "tests", "golden")


def main():
    # python oracle/gen_golden.py [group ...]: only the named groups (npz files) are rewritten; none = all
    only = set(sys.argv[1:])
    subprocess.check_call(["make", "-s", "-C", HERE, "ref"])
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.check_call([os.path.join(HERE, "_ref", "ref_dump"), tmp], stdout=subprocess.DEVNULL)
        groups = {}
        for fn in sorted(os.listdir(tmp)):
            if not fn.endswith(".npy"):
                continue
            stem = fn[:-4]
            if stem.startswith("kat_"):
                group, key = "kats", stem[4:]
            else:
                group, key = stem.split(".", 1)
            groups.setdefault(group, {})[key] = np.load(os.path.join(tmp, fn))
        os.makedirs(GOLDEN, exist_ok=True)
        total = 0
        for group, arrays in groups.items():
            if only and group not in only:
                continue
            path = os.path.join(GOLDEN, group + ".npz")
            np.savez_compressed(path, **arrays)
            total += os.path.getsize(path)
            print(f"{group}: {len(arrays)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")
        print(f"total {total / 1e6:.2f} MB")


if __name__ == "__main__":
    sys.exit(main())
