#!/usr/bin/env python3
"""Static check of a hipcc listing (-S / -save-temps, gfx9 wave64) for vector instructions that execute with EXEC = 0.

Why: round 4 shipped the solver libraries with `-mllvm -amdgpu-sched-strategy=max-ilp` for a few revisions until one
refactoring came out of it with mu = 1 where mu_step = 2 says 10 (profiles/r04_ilqr_split_ab.txt).  Root cause (round 5,
profiles/r05_ilqr_max_ilp_root_cause.txt): a divergent loop leaves through

        s_andn2_b64 exec, exec, s[4:5]
        s_cbranch_execnz .LBB14_599          ; back edge: taken while any lane is still in the loop
    ; %bb.600:                                ; loop exit: EXEC IS ZERO here ...
        s_mov_b32 s74, 0                      ; (a scalar constant the max-ilp scheduler moved up to the block's top)
        v_accvgpr_write_b32 a226, v5          ;  Reload Reuse   <- the register allocator's copy of the loop's RESULT,
        v_accvgpr_write_b32 a227, v4          ;  Reload Reuse      executed by no lane
        s_or_b64 exec, exec, s[4:5]           ; ... until here

LLVM's AMDGPU backend keeps spill code behind the exec restore of a block by treating the restore as the block's
PROLOGUE -- which only works while the restore is the block's first instruction; a scalar move scheduled above it ends
the prologue early and the allocator's copies land in front of the restore.  A compiler bug (ROCm 7.2.0, AMD clang 22.0.0git roc-7.2.0),
not undefined behaviour in the source: the default scheduler did not move scalar code above the restore on those sources, so
its listings were clean -- and whether a given source comes out of max-ilp with the misplaced copies turned out to depend on
unrelated code (a change of builtins.h at the end of round 5 moved all nine models out of it).  So this script CHECKS instead
of assuming: tpl_amd/optim/genopt.py runs it on every solver library it builds -- under max-ilp first (2 % faster on the heavy
models), and on a hit the library is thrown away and built under the default strategy; only a second hit fails the build.

LIMITS.  The scan recognises ONE code shape, the one the miscompile was observed in: `s_andn2_b64 exec, exec, ...` directly
followed by `s_cbranch_execnz` with a fall-through exit, and vector instructions between that branch and the next write of
EXEC.  It stops at the first `s_* exec, ...` write even if that write leaves EXEC at 0 (a second andn2 / and), at any branch,
and after 400 lines; other ways of reaching EXEC = 0 (a `s_and_saveexec` whose mask is empty, structured `if` regions) are not
modelled.  A clean scan therefore says "the known pattern is absent", NOT "no instruction executes at EXEC = 0": the bit-level
tests (13 goldens x 4 launch forms, the other-scheduler test, the fuzzes) remain the proof of the build.

    python -m tpl_amd.optim.isa_check listing.s [...]      (or tools/find_exec0_spills.py) exit code 1 when anything is found
"""
import re
import sys

EXEC_USERS = re.compile(r"^\s+(v_|ds_|buffer_|global_|flat_|scratch_|image_|tbuffer_)")
# vector instructions that do not depend on EXEC (they address lanes explicitly)
EXEC_FREE = re.compile(r"^\s+(v_readlane_b32|v_writelane_b32|v_readfirstlane_b32)\b")
WRITES_EXEC = re.compile(r"^\s+s_\w+\s+exec(_lo|_hi)?\s*,|^\s+s_(or|and|xor|andn2|orn2|nand|nor|xnor)_saveexec_b64|^\s+s_and_saveexec|^\s+s_or_saveexec")


def scan(path):
    hits = []
    func = None
    lines = open(path, errors="replace").read().split("\n")
    for i, line in enumerate(lines):
        m = re.match(r"^([A-Za-z_][\w$.]*):\s*(;.*)?$", line)
        if m and not m.group(1).startswith(".L"):
            func = m.group(1)
        if not re.match(r"^\s+s_cbranch_execnz\s", line):
            continue
        # the back edge of a divergent loop is `s_andn2_b64 exec, exec, s[..]` + `s_cbranch_execnz`: falling through means EXEC == 0
        prev = next((lines[j] for j in range(i - 1, max(i - 6, -1), -1) if lines[j].strip() and not lines[j].lstrip().startswith(";")), "")
        if not re.match(r"^\s+s_andn2_b64\s+exec\s*,\s*exec\s*,", prev):
            continue
        for j in range(i + 1, min(i + 400, len(lines))):
            ln = lines[j]
            if WRITES_EXEC.match(ln):
                break
            if re.match(r"^\s+s_(branch|cbranch_\w+|endpgm|setpc_b64|swappc_b64)\b", ln):
                break                                   # left the block without restoring EXEC: not this pattern
            if EXEC_USERS.match(ln) and not EXEC_FREE.match(ln):
                hits.append((path, func, j + 1, ln.strip()))
    return hits


def main(paths):
    bad = []
    for p in paths:
        bad += scan(p)
    for path, func, n, text in bad:
        print(f"{path}:{n}: in {func}: executes with EXEC = 0 (between a divergent loop's exit and its exec restore): {text}")
    print(f"{len(bad)} vector instruction(s) at EXEC = 0 in {len(paths)} listing(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
