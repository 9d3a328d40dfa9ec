"""Same-box A/B of the batched iLQR update(): the single launch (k_ilqr_update) against the split sequence
(k_split_begin / _refresh / _chain / _end), over the batch size -- optionally for several builds of the solver
library.  TPLHIP_DEBUG is read when a solver handle is created, so every leg runs in one process on one box; variant
builds are loaded from their own path and never installed over the shipped library.

    here:       python tools/ab_ilqr_split.py build NAME [--ilp] [--model=M] [-DMACRO=V ...]  -> tools/ab_tmp/ilqr_NAME.so
    on the box: python tools/ab_ilqr_split.py [mpc|lateral] [--libs NAME,NAME] [--forms 0,1] [batches...]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
TMP = os.path.join(ROOT, "tools", "ab_tmp")


def build(name, defines, model="trajectory_tracking_mpc"):
    """(the shipped flags of tpl_amd/optim/genopt.py:compile_model; `--ilp` among the defines adds the max-ilp scheduler
    strategy that was shipped for a while in round 4)"""
    os.makedirs(TMP, exist_ok=True)
    sched = ["-mllvm", "-amdgpu-sched-strategy=max-ilp"] if "--ilp" in defines else []
    defines = [d for d in defines if d not in ("--plain", "--ilp")]
    inc = os.path.join(ROOT, "tpl_amd", "models", f"{model}.inc")
    so = os.path.join(TMP, f"ilqr_{name}.so")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-fno-fast-math", "-ffp-contract=off", "-Wno-unused-function", f'-DTPL_MODEL_INC="{inc}"',
           "-I", os.path.join(ROOT, "include"), *sched, *defines, "-o", so,
           os.path.join(ROOT, "tpl_amd", "csrc", "ilqr", "ilqr_model.hip")]
    subprocess.check_call(cmd)
    print("built", os.path.relpath(so, ROOT), " ".join(defines))


def main():
    args = sys.argv[1:]
    if args and args[0] == "build":
        model = "trajectory_tracking_mpc"
        rest = []
        for a in args[2:]:
            if a.startswith("--model="):
                model = a.split("=", 1)[1]
            else:
                rest.append(a)
        build(args[1], rest, model)
        return
    from benchkit import ilqr
    workload = args.pop(0) if args and args[0] in ilqr.WORKLOADS else "mpc"
    libs, forms = [None], [0, 1]
    while args and args[0].startswith("--"):
        k = args.pop(0)
        v = args.pop(0)
        if k == "--libs":
            libs = [None if n == "shipped" else os.path.join(TMP, f"ilqr_{n}.so") for n in v.split(",")]
        elif k == "--forms":
            forms = [int(f) for f in v.split(",")]
    batches = [int(a) for a in args] or [1024, 2048, 4096, 8192]
    extra = os.environ.get("TPLHIP_DEBUG", "")
    for b in batches:
        for lib in libs:
            for form in forms:
                os.environ["TPLHIP_DEBUG"] = f"ilqr_split={form}" + ("," + extra if extra else "")
                r = ilqr.run(batch=b, updates=3, warmup=1, workload=workload, lib=lib)
                assert r["all_finite"]
                print(json.dumps({"batch": b, "lib": os.path.basename(lib)[5:-3] if lib else "shipped", "form": r["form"],
                                  "problem_iterations_per_s": round(r["problem_iterations_per_s"]),
                                  "batched_iters_per_s": round(r["value"], 1),
                                  "kernel_ms_per_update": round(r["kernel_ms_per_update"], 3)}), flush=True)


if __name__ == "__main__":
    main()
