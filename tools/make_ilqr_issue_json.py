#!/usr/bin/env python3
"""profiles/<tag>_ilqr_issue.json from the PMC summaries of tools/profile_round.sh: the share of fp64 arithmetic among ALL issued
instructions of the batched-iLQR kernels (benchkit/ilqr.py copies it into roofline.fp64_issue_share, keyed by problem / form).

    python tools/make_ilqr_issue_json.py r05 <kernel revision>"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, rev = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "?")
out = {}
for form, fname, batch in (("single", "pmc_ilqr.txt", 1024), ("fused", "pmc_ilqr_fused.txt", 4096)):
    path = os.path.join(ROOT, "gpurun_out", tag, fname)
    if not os.path.exists(path):
        continue
    c = {m.group(1): float(m.group(2)) for m in re.finditer(r"^(\w+)\s+([0-9.]+)\s+\(n=", open(path).read(), re.M)}
    fp64 = c["SQ_INSTS_VALU_ADD_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_TRANS_F64"]
    issued = c["SQ_INSTS_VALU"] + c["SQ_INSTS_SALU"] + c["SQ_INSTS_SMEM"] + c["SQ_INSTS_LDS"] + c["SQ_INSTS_VMEM_RD"] + c["SQ_INSTS_VMEM_WR"]
    out[f"trajectory_tracking_mpc/{form}"] = {
        "fp64_issue_share": fp64 / issued, "fp64_share_of_valu": fp64 / c["SQ_INSTS_VALU"],
        "valu_active_share_of_wave_cycles": c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"],
        "waitcnt_share_of_wave_cycles": c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"],
        "instructions_per_wavefront": {k: c[k] / c["SQ_WAVES"] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR")},
        "what": f"rocprofv3 --pmc passes of tools/pmc_ilqr.sh on tools/ilqr_only.py mpc {batch}: (ADD + MUL + FMA + TRANS)_F64 / (VALU + SALU + SMEM + LDS + VMEM)",
        "batch": batch, "profiled_kernel_rev": rev}
dst = os.path.join(ROOT, "profiles", f"{tag}_ilqr_issue.json")
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
