"""profiles/summarize_db.py <results.db> <log> "<command line>": the per-kernel table of a rocprofv3 --kernel-trace run
(rocpd database) of tools/build_on_device.py, as committed under profiles/r3_build*/summary.txt."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("command: " + sys.argv[3])
for line in open(sys.argv[2]):
    if "triangles" in line or "rt_bvh_build_device call" in line or "same tree" in line:
        print(line.rstrip())
rows = list(db.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name order by 3 desc"))
total = sum(r[2] for r in rows)
launches = sum(r[1] for r in rows)
rounds = [r[1] for r in rows if "k_round_cuts" in r[0]][0] // 3
print(f"== kernel stats: {total:.2f} ms of kernel time over {launches} launches (three builds; {rounds} partition rounds per build) ==")
for r in rows:
    print(f"{r[0][:100]:100s} calls={r[1]:6d} avg_us={r[3]:8.1f} total_ms={r[2]:8.2f} {100 * r[2] / total:5.1f}%")
