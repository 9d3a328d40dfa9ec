"""bench.py's contract on a GPU box: ONE JSON line with the fields the driver reads, and the N > 1 code path run
through RCCL itself.  A one-GPU box cannot hold two RCCL ranks, so the second test forces a process group of one
rank (TPL_BENCH_FORCE_DIST=1): init_process_group("nccl"), the barriers, the all-gather of the device-resident record
(`DynProgLonPlanner.best_record_device` -> __cuda_array_interface__ -> torch, no host copy) and the reductions are the
calls ranks 0..N-1 make at N > 1."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _bench(extra_env, *flags):
    env = dict(os.environ, **extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", *flags], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines                       # exactly one line on stdout
    return json.loads(lines[0])


def test_one_line_with_the_contract_fields():
    d = _bench({}, "--no-secondary")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and "workload" in d["config"]
    rf = d["roofline"]
    # `frac` is the HBM fraction SURVEY 8d prescribes; `bound` names what the committed PMC passes say binds the kernel
    assert rf["bound"] == "valu_issue" and rf["prescribed_bound"] == "hbm" and 0.0 < rf["valu_issue"] < 1.0
    assert rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert 0.05 < rf["frac"] < 1.0
    assert rf["kernel"].startswith("k_backward3v")       # the large-grid sweep is the one that ran
    # `achieved` / `frac` are ALWAYS the per-launch figure (algorithmic bytes of one launch / the kernel's HIP-event duration
    # alone on the chip); what two plans in flight add sits under `in_flight` and is never mixed into `frac`
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["launch_us"] * 1e-6) / 1e9) / rf["achieved"] < 1e-6
    fl = rf["in_flight"]
    assert fl["launches_in_flight"] == 2 == d["config"]["plans_in_flight_per_gpu"]
    assert abs(fl["achieved"] - d["value"] * rf["algorithmic_bytes_per_launch"] / 1e9) / fl["achieved"] < 1e-6
    assert abs(fl["frac"] - fl["achieved"] / rf["peak"]) < 1e-9
    # (orders of magnitude only: on a 2-step run the margins between these three are ~15 % and noisy)
    assert fl["stream_us_per_launch"] > 20.0 and rf["launch_us"] > 20.0 and fl["effective_us_per_launch"] > 15.0
    assert fl["effective_us_per_launch"] < 1.25 * rf["launch_us"]
    by_rule = rf["launch_us_by_rule"]                   # the kernel alone under both settings of R1, same box
    assert set(by_rule) == {"R1=1", "R1=0"} and all(20.0 < v < 80.0 for v in by_rule.values()), by_rule
    # plans in flight are swept stage by stage; the same six plans one at a time under BOTH launch forms sit beside it
    assert d["config"]["chained_sweep"] is False
    forms = rf["stage_us_by_launch_form"]
    assert set(forms) == {"stage_by_stage", "chained"} and all(20.0 < v < 80.0 for v in forms.values()), forms
    assert forms["chained"] < 1.02 * forms["stage_by_stage"], forms
    assert rf["stages_per_chained_launch"] == d["config"]["t_steps"] - 2
    # the arithmetic the sweeps ran under is named: the contracted default (nvcc's -fmad=true), rint weights, a -> v -> s
    assert d["config"]["arithmetic"].startswith("R1=1") and "R2=rint" in d["config"]["arithmetic"] and d["config"]["arithmetic"].endswith("avs")
    # every plan of the timed region had its own (obstacle set, start state) pair, and the last one per planner was checked
    assert d["config"]["inputs"] == {"obstacle_sets": 4, "start_states": 7, "plans_spot_checked": 2}


def test_one_plan_in_flight_keeps_the_per_launch_roofline():
    d = _bench({}, "--no-secondary", "--in-flight", "1")
    rf = d["roofline"]
    assert rf["in_flight"]["launches_in_flight"] == 1
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["launch_us"] * 1e-6) / 1e9) / rf["achieved"] < 1e-6
    assert abs(rf["launch_us"] - rf["in_flight"]["stream_us_per_launch"]) / rf["launch_us"] < 1e-9
    # one plan at a time takes the chained launch (all stages of a plan in one launch); `launch_us` is then per STAGE SWEEP
    assert d["config"]["chained_sweep"] is True and set(rf["stage_us_by_launch_form"]) == {"stage_by_stage", "chained"}
    d0 = _bench({}, "--no-secondary", "--in-flight", "1", "--chained", "0")
    assert d0["config"]["chained_sweep"] is False


def test_collectives_run_through_rccl_with_one_rank():
    d = _bench({"TPL_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29547", "RANK": "0",
                "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert d["n_gpus"] == 1 and "best_plan" in d and d["best_plan"]["rank"] == 0
    mc = d["secondary"]["mc_rollouts"]
    assert mc["best_plan"]["identical_on_all_ranks"] and mc["value"] > 0
    for name in ("batched_ilqr", "batched_ilqr_lateral_profile"):
        assert d["secondary"][name]["all_finite"] and "best_plan" in d["secondary"][name]
    # the other two thirds of BASELINE's metric are mirrored into fields the driver keeps (roofline / config)
    il = d["roofline"]["ilqr"]
    assert il["value"] == d["secondary"]["batched_ilqr"]["value"] > 0 and il["unit"].startswith("batched iLQR") and 0.0 < il["frac"] < 1.0
    assert il["kernel"].startswith("k_ilqr") and il["ms_per_update"] > 0
    lat = d["config"]["p50_plan_latency_ms"]
    assert lat["dp_default_grid"] > 0 and lat["mpc_tick"] > 0
    # update() on the headline grid takes the chained sweep by default: not slower than the same call stage by stage
    assert 0.5 < lat["dp_target_grid"] < 1.05 * lat["dp_target_grid_stage_by_stage"] < 10.0, lat


def test_two_ranks_on_one_device_rehearsal():
    """`bench.py --gpus 2` the way the driver's scaling run starts it (no torchrun: bench.py launches its ranks as child
    processes).  A one-GPU box cannot give each rank its own device, so both share device 0 and the process group is
    gloo (RCCL refuses two ranks on one device): what this exercises is everything around the collective -- the
    launcher, the rendezvous, the rank-dependent scenario, the all-gather of the fixed-size records, the identical
    final argmin on every rank and rank 0's single line."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-secondary"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    col = d["collective"]
    assert col["ranks_seen"] == 2 and len(col["per_rank_value"]) == 2 and len(col["per_rank_plan"]) == 2
    a, b = col["per_rank_plan"]
    assert (a["cost"], a["constr"]) != (b["cost"], b["constr"]), "the two ranks planned the same scenario"
    want = min(range(2), key=lambda i: (col["per_rank_plan"][i]["constr"], col["per_rank_plan"][i]["cost"], i))
    assert d["best_plan"]["rank"] == want and d["best_plan"]["cost"] == col["per_rank_plan"][want]["cost"]
    assert abs(d["value"] - sum(col["per_rank_value"])) / d["value"] < 0.5      # whole-job value is of the order of the ranks' sum
