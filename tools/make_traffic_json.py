#!/usr/bin/env python3
"""profiles/<tag>_dp_traffic.json from the per-dispatch PMC averages of tools/pmc_dp.sh.
usage: python3 tools/make_traffic_json.py <tag> <key>=<pmc summary.txt> [<key>=<file> ...] [--kernel name]
keys: a grid "SxVxA" (one plan per launch) or "SxVxA@N" (N scenarios per launch, the Monte-Carlo workload)."""
import json
import re
import sys

NOTE = ("bytes between L2 and the fabric per sweep launch = 2*FETCH_SIZE + WRITE_SIZE (FETCH_SIZE / WRITE_SIZE in KB). "
        "The factor 2 is MI355X_MICROARCH.md's gfx950 correction, CALIBRATED for this kernel's access shape in round 3: "
        "tools/micro/fetch_cal.hip streams 1 GiB from a 1.5 GiB buffer with buffer_load_dwordx3 (12 B/lane, 3 KiB rows) and "
        "FETCH_SIZE reports 524297.5 KB = exactly half (4- and 16-byte lanes: the same 2.0000); 12- and 16-byte stores: "
        "WRITE_SIZE exact (profiles/r03_fetch_size_calibration.txt). Infinity-Cache hits are included in FETCH_SIZE, so this "
        "bounds HBM bytes from above.")
MODEL = ("issue model from the PMC instruction counts: valu_issue = VALU instructions per wave x 2.5 cycles (plain fp32 issue "
         "rate at 8 waves per SIMD, profiles/r02_micro_valu_issue_8waves.txt) x waves per SIMD / kernel cycles; texture_addr = "
         "vector-memory instructions per wave x 16.1 cycles (a 12-byte-per-lane wave load, profiles/r02_micro_vmem_issue.txt) x "
         "waves per CU / kernel cycles; kernel cycles = SQ_BUSY_CYCLES / 32 shader engines (clock_ghz = that / the launch time)")


def parse(path):
    vals = {}
    for line in open(path):
        m = re.match(r"^(\w+)\s+([0-9.]+)\s+\(n=", line)
        if m:
            vals[m.group(1)] = float(m.group(2))
    return vals


def entry(v, kernel, scenarios):
    fetch = v["FETCH_SIZE"] * 1024.0
    write = v["WRITE_SIZE"] * 1024.0
    waves = v["SQ_WAVES"]
    cycles = v["SQ_BUSY_CYCLES"] / 32.0
    valu = v["SQ_INSTS_VALU"] / waves
    vmem = (v["SQ_INSTS_VMEM_RD"] + v["SQ_INSTS_VMEM_WR"]) / waves
    out = {
        "kernel": kernel,
        "scenarios_per_launch": scenarios,
        "fetch_size_bytes_raw": fetch,
        "write_size_bytes": write,
        "traffic_bytes_per_launch": 2.0 * fetch + write,
        "l2_hit_rate": round(v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"]), 4),
        "valu_insts_per_wave": round(valu, 1),
        "vmem_insts_per_wave": round(vmem, 1),
        "kernel_cycles": round(cycles, 0),
        "valu_issue": round(valu * 2.5 * (waves / 1024.0) / cycles, 3),
        "texture_addr": round(vmem * 16.1 * (waves / 256.0) / cycles, 3),
        "wave_parked_frac": round(v["SQ_WAIT_ANY"] / v["SQ_WAVE_CYCLES"], 3),
        "wave_issue_stalled_frac": round(v["SQ_WAIT_INST_ANY"] / v["SQ_WAVE_CYCLES"], 3),
        "note": NOTE,
        "model": MODEL,
    }
    return out


def main():
    args = sys.argv[1:]
    kernel = "k_backward3v"
    if "--kernel" in args:
        i = args.index("--kernel")
        kernel = args[i + 1]
        del args[i:i + 2]
    tag = args[0]
    out = {}
    for kv in args[1:]:
        key, path = kv.split("=", 1)
        grid, _, n = key.partition("@")
        out[key] = entry(parse(path), kernel, int(n) if n else 1)
    # which sweep source these counters were taken on: bench.py copies them into its line as constants of the repository,
    # and says which revision of the kernel they describe
    import subprocess
    try:
        rev = subprocess.check_output(["git", "log", "-1", "--format=%h", "--", "tpl_amd/csrc/dp_sweep.hip"], text=True).strip()
    except Exception:
        rev = "unknown"
    for v in out.values():
        v["profiled_kernel_rev"] = rev
    with open(f"profiles/{tag}_dp_traffic.json", "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk not in ("note", "model")} for k, v in out.items()}, indent=1))


if __name__ == "__main__":
    main()
