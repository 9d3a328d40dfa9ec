#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -S listing, priced with the issue model of tools/micro/valu_tp.hip:
every VALU instruction takes an issue slot (~2.5 nominal cycles); 'slow' ones (everything but plain fp32
add/sub/mul/fma/fmac, v_mov, integer add/sub/and with vector or literal operands) also hold a second unit ~4.5.
    python tools/isa_mix.py tpl_amd/csrc/dp_sweep.s k_backward3vILi4ELi3 [top]"""
import collections
import re
import sys

FAST = ("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mov_b32", "v_add_u32", "v_sub_u32",
        "v_subrev_u32", "v_and_b32", "v_ashrrev_i32", "v_xor_b32", "v_or_b32", "v_mad_f32")


def main():
    text = open(sys.argv[1]).read()
    name = sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    m = re.search(r"^(\S*%s\S*):" % re.escape(name), text, re.M)
    if not m:
        raise SystemExit("kernel not found")
    body = text[m.end():text.index("s_endpgm", m.end())]
    cnt = collections.Counter()
    nv = nslow = nsg = 0
    for line in body.split("\n"):
        mm = re.match(r"\s+([a-z_0-9]+)\s*(.*)", line)
        if not mm:
            continue
        op, args = mm.group(1), mm.group(2).split(";")[0]
        cnt[op] += 1
        if op.startswith("v_"):
            nv += 1
            base = re.sub(r"_e(32|64)$", "", op)
            has_s = bool(re.search(r"\bs\d+\b|\bs\[|\bvcc\b|\bexec\b", args))
            if base not in FAST or has_s:
                nslow += 1
            if base in FAST and has_s:
                nsg += 1
    print(f"{m.group(1)[:80]}")
    print(f"VALU {nv}  slow-class {nslow} (of which plain ops with a scalar operand: {nsg})  "
          f"model: max(2.5*{nv}, 4.5*{nslow}) = {max(2.5 * nv, 4.5 * nslow):.0f} nominal cycles per wave")
    for k, v in cnt.most_common(top):
        print(f"{v:6d} {k}")


if __name__ == "__main__":
    main()
