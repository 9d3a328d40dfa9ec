#!/usr/bin/env python3
"""Writes tests/golden/tplcpp_surface.json: the NAMES tplcpp's pybind11 module exposes for the DP family -- classes,
their methods / read-write attributes / properties, and module functions -- collected from the binding sources of the
reference (library/src/python/bindings_dp_{common,lon,env,poly_lat,lat_lon}.cu).  Names only, no source text: the
fixture is the interface tests/test_surface.py holds tpl_amd to.  Run here (needs /root/reference)."""
import json
import os
import re

REF = "/root/reference/library/src/python"
FILES = ["bindings_dp_common.cu", "bindings_dp_lon.cu", "bindings_dp_env.cu", "bindings_dp_poly_lat.cu",
         "bindings_dp_lat_lon.cu"]
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    surface = {"classes": {}, "functions": []}
    for f in FILES:
        text = open(os.path.join(REF, f)).read()
        # split at class openings; the text before the next class / module function belongs to the class
        marks = [(m.start(), "class", m.group(1)) for m in re.finditer(r'py::class_<[^>]+>\(m,\s*"(\w+)"\)', text)]
        marks += [(m.start(), "func", m.group(1)) for m in re.finditer(r'\bm\.def\("(\w+)"', text)]
        marks.sort()
        for i, (pos, kind, name) in enumerate(marks):
            end = marks[i + 1][0] if i + 1 < len(marks) else len(text)
            if kind == "func":
                surface["functions"].append(name)
                continue
            body = text[pos:end]
            cls = surface["classes"].setdefault(name, {"methods": [], "attributes": [], "properties": [], "file": f})
            for m in re.finditer(r'\.\s*def(?:_static)?\("(\w+)"', body):
                if m.group(1) not in cls["methods"]:
                    cls["methods"].append(m.group(1))
            cls["attributes"] += [m.group(1) for m in re.finditer(r'\.def_readwrite\("(\w+)"', body)]
            cls["properties"] += [m.group(1) for m in re.finditer(r'\.def_property(?:_readonly)?\("(\w+)"', body)]
            cls["init"] = bool(re.search(r'\.def\(py::init', body))
            m = re.search(r'"__slots__".*?std::vector<std::string>\(\{(.*?)\}\)', body, re.S)
            if m:
                cls["slots"] = re.findall(r'"(\w+)"', m.group(1))
    path = os.path.join(HERE, "tplcpp_surface.json")
    with open(path, "w") as fh:
        json.dump(surface, fh, indent=1, sort_keys=True)
    print(path, len(surface["classes"]), "classes", len(surface["functions"]), "functions")


if __name__ == "__main__":
    main()
