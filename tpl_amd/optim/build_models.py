"""Generate and compile every shipped solver library (called by
__graft_entry__.build()).  mtime-based like make: a library is rebuilt when its
model header or the solver template is newer."""
import concurrent.futures as cf
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main(force=False, jobs=4):
    from tpl_amd.optim import genopt, optimizers as o
    fns = o.CONFIG_FUNCTIONS + o.EXTRA_CONFIG_FUNCTIONS
    cfgs = [f() for f in fns]
    for c in cfgs:
        genopt.generate(c, c.name, force=force)
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        for so in ex.map(lambda c: genopt.compile_model(c.name, force=force), cfgs):
            print("built", os.path.relpath(so, ROOT))


if __name__ == "__main__":
    main(force="--force" in sys.argv)
