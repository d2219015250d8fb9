"""profiles/make_traffic.py <tag> [output.json]: writes profiles/traffic.json (or the named file under profiles/) from
profiles/<tag>/summary.txt (the FETCH_SIZE and WRITE_SIZE passes of profiles/run_profile.sh).  Check on a known byte count: k_resolve reads 16 sample planes of
1920 x 1080 x 12 B = 398 MB, and 2 x FETCH_SIZE x 1024 gives 398 MB."""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
tag = sys.argv[1]
out_name = sys.argv[2] if len(sys.argv) > 2 else "traffic.json"
txt = open(os.path.join(HERE, tag, "summary.txt")).read()
command = txt.splitlines()[0].replace("command: ", "") if txt.startswith("command: ") else "python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline"


def grab(section, counter):
    sec = txt.split("== PMC pass %s" % section)[1].split("== PMC")[0]
    out = {}
    for line in sec.splitlines():
        m = re.search(r"(k_\w+(?:<[^>]*>)?).*?dispatches=\s*(\d+).*?%s=([\d.e+]+)" % counter, line)
        if m:
            out[m.group(1)] = float(m.group(3))
    return out


fetch, write = grab("fetch", "FETCH_SIZE"), grab("write", "WRITE_SIZE")
per_kernel = {k: {"fetch_KiB": fetch[k], "write_KiB": write.get(k, 0.0),
                  "hbm_bytes_per_launch": int(2 * fetch[k] * 1024 + write.get(k, 0.0) * 1024)} for k in fetch}
# L2 read requests (per-CU cache -> L2), and the share of lanes that are active in the vector instructions a kernel
# issues: SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)
l2req = grab("tcc", "TCP_TCC_READ_REQ_sum")
l1acc = grab("tcc", "TCP_TOTAL_CACHE_ACCESSES_sum")
thread_cycles, active_valu = grab("sq2", "SQ_THREAD_CYCLES_VALU"), grab("sq1", "SQ_ACTIVE_INST_VALU")
insts_valu, insts_salu, insts_lds = grab("sq1", "SQ_INSTS_VALU"), grab("sq2", "SQ_INSTS_SALU"), grab("sq2", "SQ_INSTS_LDS")
lds_conflict, lds_active = grab("sq2", "SQ_LDS_BANK_CONFLICT"), grab("sq2", "SQ_ACTIVE_INST_LDS")
busy_cycles = grab("sq1", "SQ_BUSY_CYCLES")
SHADER_ENGINES, SIMDS = 32, 1024  # SQ_BUSY_CYCLES is summed over the chip's 32 shader engines; 256 CUs x 4 SIMDs
for k in per_kernel:
    if k in l2req:
        per_kernel[k]["l2_read_requests_per_launch"] = l2req[k]
        per_kernel[k]["l1_accesses_per_launch"] = l1acc.get(k)
    if k in thread_cycles and active_valu.get(k):
        per_kernel[k]["active_lane_frac"] = thread_cycles[k] / (64.0 * active_valu[k])
    if k in insts_valu and busy_cycles.get(k):
        # share of the chip's vector issue slots the launch uses: a wave64 vector instruction occupies its SIMD for 4 cycles
        per_kernel[k]["valu_issue_util"] = insts_valu[k] * 4.0 / (SIMDS * busy_cycles[k] / SHADER_ENGINES)
    for name, table in (("valu_insts_per_launch", insts_valu), ("salu_insts_per_launch", insts_salu), ("lds_insts_per_launch", insts_lds),
                        ("lds_bank_conflict_cycles", lds_conflict), ("lds_active_cycles", lds_active)):
        if k in table:
            per_kernel[k][name] = table[k]
frame = ["k_primary<true, false>", "k_stream<false, true, 1>", "k_closest<true, false>", "k_cull", "k_stream<false, false, 2>", "k_shade", "k_resolve"]
doc = {
    "source": "profiles/%s/summary.txt: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, one run each, of `%s` on one MI355X; "
              "per-dispatch averages" % (tag, command),
    "counters": "l2_read_requests = TCP_TCC_READ_REQ_sum (requests of 64 or 128 bytes); active_lane_frac = "
                "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); instruction counts are wave instructions; valu_issue_util = "
                "SQ_INSTS_VALU x 4 / (1024 SIMDs x SQ_BUSY_CYCLES / 32 shader engines)",
    "units": "FETCH_SIZE and WRITE_SIZE are reported in KiB; bytes = 2 x FETCH_SIZE x 1024 (gfx950 counts 128-byte read "
             "requests as 64 bytes) + WRITE_SIZE x 1024",
    "per_kernel": per_kernel,
    "dominant_kernel": "k_stream<false, false, 2>",
    "dominant_kernel_hbm_bytes_per_launch": per_kernel["k_stream<false, false, 2>"]["hbm_bytes_per_launch"],
    "hbm_bytes_per_frame": sum(per_kernel[k]["hbm_bytes_per_launch"] for k in frame if k in per_kernel),
}
json.dump(doc, open(os.path.join(HERE, out_name), "w"), indent=1)
print(doc["dominant_kernel_hbm_bytes_per_launch"], doc["hbm_bytes_per_frame"])
