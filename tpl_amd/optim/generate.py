"""Generate the model headers (tpl_amd/models/<problem>.inc) of every OCP tpl
ships.  Run by __graft_entry__.build() before the HIP and oracle builds."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def main(force=False):
    # import the generator modules without importing tpl_amd/__init__ (which needs libtplhip.so)
    import importlib.util
    import types
    pkg = types.ModuleType("tpl_amd_gen")
    pkg.__path__ = [os.path.join(ROOT, "tpl_amd", "optim")]
    sys.modules.setdefault("tpl_amd_gen", pkg)
    mods = {}
    for name in ("symext", "genopt", "optimizers"):
        spec = importlib.util.spec_from_file_location(
            f"tpl_amd_gen.{name}", os.path.join(ROOT, "tpl_amd", "optim", f"{name}.py"))
        m = importlib.util.module_from_spec(spec)
        sys.modules[f"tpl_amd_gen.{name}"] = m
        spec.loader.exec_module(m)
        mods[name] = m
    o, g = mods["optimizers"], mods["genopt"]
    for fn in o.CONFIG_FUNCTIONS + o.EXTRA_CONFIG_FUNCTIONS:
        cfg = fn()
        path = g.generate(cfg, cfg.name, force=force)
        print("model", cfg.name, "->", os.path.relpath(path, ROOT))


if __name__ == "__main__":
    main(force="--force" in sys.argv)
