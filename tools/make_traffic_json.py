#!/usr/bin/env python3
"""profiles/<tag>_dp_traffic.json from the per-dispatch PMC averages of tools/pmc_dp.sh.
usage: python3 tools/make_traffic_json.py <tag> <pmc_target.txt> <pmc_config4.txt> [kernel name]"""
import json
import re
import sys

NOTE = ("bytes between L2 and the fabric per sweep launch = 2*FETCH_SIZE + WRITE_SIZE (FETCH_SIZE in KB; "
        "x2 = the gfx950 correction of MI355X_MICROARCH.md, calibrated there for 16 B/lane reads, ours are "
        "12 B/lane: true value in [FETCH+WRITE, 2*FETCH+WRITE]). Infinity-Cache hits are included, so this "
        "bounds HBM bytes from above.")


def parse(path):
    vals = {}
    for line in open(path):
        m = re.match(r"^(\w+)\s+([0-9.]+)\s+\(n=", line)
        if m:
            vals[m.group(1)] = float(m.group(2))
    return vals


def entry(v, kernel):
    fetch = v["FETCH_SIZE"] * 1024.0
    write = v["WRITE_SIZE"] * 1024.0
    return {
        "kernel": kernel,
        "fetch_size_bytes_raw": fetch,
        "write_size_bytes": write,
        "traffic_bytes_per_launch": 2.0 * fetch + write,
        "l2_hit_rate": round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4),
        "valu_insts_per_wave": round(v["SQ_INSTS_VALU"] / v["SQ_WAVES"], 1),
        "vmem_reads_per_wave": round(v["SQ_INSTS_VMEM_RD"] / v["SQ_WAVES"], 1),
        "note": NOTE,
    }


tag, tgt, c4 = sys.argv[1:4]
kernel = sys.argv[4] if len(sys.argv) > 4 else "k_backward3v"
out = {"256x128x64": entry(parse(tgt), kernel), "512x256x128": entry(parse(c4), kernel)}
with open(f"profiles/{tag}_dp_traffic.json", "w") as f:
    json.dump(out, f, indent=1)
print(json.dumps(out, indent=1))
