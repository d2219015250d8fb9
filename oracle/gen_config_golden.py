"""Generates tests/golden/config/<name>.out / .err: what the reference's readConfigFile + printer (compiled into
oracle/_ref/ref_dump, mode --config) make of the TOML files under tests/golden/config/, which were written for
this repository.  The reference's compile-time data directory is replaced by the token <DATA_DIR>.  Run where
/root/reference exists; the results are committed.

    python oracle/gen_config_golden.py
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
DIR = os.path.join(ROOT, "tests", "golden", "config")


def main():
    for name in sorted(f for f in os.listdir(DIR) if f.endswith(".toml")):
        res = subprocess.run([REF, "--config", os.path.join("tests", "golden", "config", name)], cwd=ROOT, capture_output=True, text=True)
        assert res.returncode == 0, (name, res.returncode, res.stderr)
        base = os.path.join(DIR, name[:-5])
        with open(base + ".out", "w") as f:
            f.write(res.stdout.replace("/root/reference/data/", "<DATA_DIR>"))
        with open(base + ".err", "w") as f:
            f.write(res.stderr)
        print(name, len(res.stdout), "bytes,", len(res.stderr), "bytes on stderr")


if __name__ == "__main__":
    main()
