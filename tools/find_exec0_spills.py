#!/usr/bin/env python3
"""Command-line front of tpl_amd/optim/isa_check.py: scan hipcc listings for vector instructions that execute with EXEC = 0
(the round-4 max-ilp miscompile, profiles/r05_ilqr_max_ilp_root_cause.txt).   python tools/find_exec0_spills.py listing.s [...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tpl_amd.optim import isa_check  # noqa: E402

if __name__ == "__main__":
    sys.exit(isa_check.main(sys.argv[1:]))
