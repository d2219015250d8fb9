"""Generates tests/golden/obj_own.npz: what the reference's loadMesh (compiled, oracle/_ref/ref_dump --obj) makes of
the OBJ/MTL files under tests/golden/obj_own/, which were written for this repository (they are not the
reference's data files).  Run here, where /root/reference exists; the result is committed.

    python oracle/gen_obj_golden.py
"""
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
OWN = os.path.join(ROOT, "tests", "golden", "obj_own")


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name in sorted(f for f in os.listdir(OWN) if f.endswith(".obj")):
            for flag in (0, 1):
                prefix = os.path.join(tmp, f"{name}.{flag}.")
                subprocess.run([REF, "--obj", os.path.join(OWN, name), str(flag), prefix], check=True)
                for key in ("vertices", "triangles", "mesh_table", "materials"):
                    out[f"{name}.{flag}.{key}"] = np.load(prefix + key + ".npy")
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "obj_own.npz"), **out)
    print({k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
