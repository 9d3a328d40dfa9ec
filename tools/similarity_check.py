#!/usr/bin/env python3
"""Self-check against transcription: every source file of this repository against every reference file of the same
language (build container only).  Prints the pairs whose normalised-line similarity or character similarity is high.

    python tools/similarity_check.py [threshold]"""
import difflib
import os
import sys

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXT = {".py": ".py", ".hip": (".cu", ".cuh", ".cpp", ".hpp", ".c", ".h"), ".h": (".cu", ".cuh", ".cpp", ".hpp", ".c", ".h"),
       ".c": (".cu", ".cuh", ".cpp", ".hpp", ".c", ".h"), ".cpp": (".cu", ".cuh", ".cpp", ".hpp", ".c", ".h")}


def norm(text):
    out = []
    for l in text.splitlines():
        l = l.strip()
        if l and not l.startswith(("#", "//", "*", "/*")):
            out.append(l)
    return out


def files(top, skip=()):
    for d, dirs, names in os.walk(top):
        dirs[:] = [x for x in dirs if x not in (".git", "gpurun_out", "__pycache__", "_ref", "_native", "_sens", "ab_tmp")]
        for n in names:
            p = os.path.join(d, n)
            if os.path.splitext(n)[1] in EXT and not any(s in p for s in skip):
                yield p


def main():
    thr = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
    ref = {}
    for p in files(REF):
        try:
            ref[p] = open(p, errors="replace").read()
        except OSError:
            pass
    ref_lines = {p: norm(t) for p, t in ref.items()}
    worst = []
    for p in files(ROOT, skip=("/models/",)):
        text = open(p, errors="replace").read()
        lines = norm(text)
        if len(lines) < 8:
            continue
        want = EXT[os.path.splitext(p)[1]]
        best = (0.0, None)
        for q, ql in ref_lines.items():
            if not q.endswith(want) or len(ql) < 8:
                continue
            if not (0.33 < len(lines) / max(1, len(ql)) < 3.0) and os.path.basename(p) != os.path.basename(q):
                continue
            sm = difflib.SequenceMatcher(None, lines, ql, autojunk=False)
            if sm.real_quick_ratio() < best[0] or sm.quick_ratio() < best[0]:
                continue
            r = sm.ratio()
            if r > best[0]:
                best = (r, q)
        if best[1] is not None:
            worst.append((best[0], os.path.relpath(p, ROOT), os.path.relpath(best[1], REF)))
    worst.sort(reverse=True)
    for r, p, q in worst[:25]:
        flag = "  <-- above threshold" if r >= thr else ""
        print(f"{r:.2f}  {p:62s} {q}{flag}")


if __name__ == "__main__":
    main()
