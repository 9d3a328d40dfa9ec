"""The build-time scan of the solver libraries' device listings for vector instructions that execute with EXEC = 0
(tpl_amd/optim/isa_check.py) -- the root cause of round 4's max-ilp divergence (profiles/r05_ilqr_max_ilp_root_cause.txt): a
register-allocator copy of the value a divergent loop computes, placed in the loop's exit block in front of the exec restore."""
import os
import time
import shutil

import pytest

from tpl_amd.optim import genopt, isa_check

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BAD = """
_ZN1kE:                                 ; @_ZN1kE
.LBB14_599:
	s_add_i32 s6, s6, 1
	v_cmp_eq_u32_e32 vcc, s6, v2
	v_mul_f64 v[4:5], v[4:5], s[8:9]
	s_or_b64 s[4:5], vcc, s[4:5]
	s_andn2_b64 exec, exec, s[4:5]
	s_cbranch_execnz .LBB14_599
; %bb.600:
	s_mov_b32 s74, 0
	v_accvgpr_write_b32 a226, v5            ;  Reload Reuse
	v_accvgpr_write_b32 a227, v4            ;  Reload Reuse
	s_mov_b32 s75, 0x40240000
	s_or_b64 exec, exec, s[4:5]
.LBB14_601:
	s_or_b64 exec, exec, s[2:3]
	v_accvgpr_write_b32 a31, v2
"""
GOOD = BAD.replace("""	s_mov_b32 s74, 0
	v_accvgpr_write_b32 a226, v5            ;  Reload Reuse
	v_accvgpr_write_b32 a227, v4            ;  Reload Reuse
	s_mov_b32 s75, 0x40240000
	s_or_b64 exec, exec, s[4:5]
""", """	s_or_b64 exec, exec, s[4:5]
	s_mov_b32 s74, 0
	v_accvgpr_write_b32 a226, v5            ;  Reload Reuse
	v_accvgpr_write_b32 a227, v4            ;  Reload Reuse
""")


def test_scan_flags_copies_in_front_of_the_exec_restore(tmp_path):
    bad, good = tmp_path / "bad.s", tmp_path / "good.s"
    bad.write_text(BAD)
    good.write_text(GOOD)
    hits = isa_check.scan(str(bad))
    assert [h[3].split()[0] for h in hits] == ["v_accvgpr_write_b32", "v_accvgpr_write_b32"] and hits[0][1] == "_ZN1kE"
    assert isa_check.scan(str(good)) == []
    # lane-addressed moves do not depend on EXEC
    lanes = tmp_path / "lanes.s"
    lanes.write_text(BAD.replace("v_accvgpr_write_b32 a226, v5", "v_readlane_b32 s3, v255, 4").replace("v_accvgpr_write_b32 a227, v4", "v_writelane_b32 v255, s3, 5"))
    assert isa_check.scan(str(lanes)) == []


def _fake_toolchain(monkeypatch, bad_under):
    """compile_model with hipcc replaced by a stub that writes the output file and a device listing, and a scan that reports a
    hit for the strategies in `bad_under` -- the POLICY is under test here, the scan itself above and on the real listings below."""
    import subprocess
    seen = []

    def run(cmd, stdout=None, stderr=None, cwd=None):
        sched = "max-ilp" if any(str(c).endswith("=max-ilp") for c in cmd) else "default"
        seen.append(sched)
        open(cmd[cmd.index("-o") + 1], "wb").write(sched.encode())
        open(os.path.join(cwd, "ilqr_model-hip-amdgcn-amd-amdhsa-gfx950.s"), "w").write(sched)
        return subprocess.CompletedProcess(cmd, 0, b"", b"")

    monkeypatch.setattr(genopt.subprocess, "run", run)
    monkeypatch.setattr(genopt, "exec0_hits", lambda lst: [("k", "_ZN1kE", 7, "v_accvgpr_write_b32 a226, v5")] if open(lst).read() in bad_under else [])
    return seen


def test_scheduler_policy_max_ilp_first_then_default(tmp_path, monkeypatch, capsys):
    """genopt.compile_model: max-ilp where its listing passes the scan; a listing with vector instructions at EXEC = 0 is thrown
    away and the library built again under the default strategy (said on stderr); refused only when both fail; a caller that
    asks for ONE strategy gets that one or an error."""
    model = "trajectory_tracking_mpc_time"
    if not os.path.exists(os.path.join(ROOT, "tpl_amd", "models", f"{model}.inc")):
        pytest.skip("generated model header not in this tree")
    so = str(tmp_path / "lib.so")
    seen = _fake_toolchain(monkeypatch, bad_under=())
    assert genopt.compile_model(model, force=True, so_path=so) == so and seen == ["max-ilp"] and open(so).read() == "max-ilp"
    seen = _fake_toolchain(monkeypatch, bad_under=("max-ilp",))
    assert genopt.compile_model(model, force=True, so_path=so) == so and seen == ["max-ilp", "default"] and open(so).read() == "default"
    assert "rebuilt under the default scheduler" in capsys.readouterr().err
    os.remove(so)
    seen = _fake_toolchain(monkeypatch, bad_under=("max-ilp", "default"))
    with pytest.raises(RuntimeError, match="EXEC = 0") as e:
        genopt.compile_model(model, force=True, so_path=so)
    assert seen == ["max-ilp", "default"] and "v_accvgpr_write_b32" in str(e.value) and not os.path.exists(so)
    seen = _fake_toolchain(monkeypatch, bad_under=("max-ilp",))
    with pytest.raises(RuntimeError, match="EXEC = 0"):
        genopt.compile_model(model, force=True, so_path=so, scheduler="max-ilp")
    assert seen == ["max-ilp"] and not os.path.exists(so)
    # check_isa=False keeps whatever the first strategy gave (A/B tools, the symptom test of round 4)
    assert genopt.compile_model(model, force=True, so_path=so, check_isa=False) == so and open(so).read() == "max-ilp"
    # a CACHED library is reused only under the scheduler the caller names: the label is a marker in the file (built_scheduler)
    seen = _fake_toolchain(monkeypatch, bad_under=())
    open(so, "wb").write(b"...tplhip_ilqr_sched=default\0...")
    os.utime(so, (time.time() + 10, time.time() + 10))
    assert genopt.built_scheduler(so) == "default"
    assert genopt.compile_model(model, so_path=so) == so and seen == []                          # fresh, no scheduler named: reused
    assert genopt.compile_model(model, so_path=so, scheduler="default") == so and seen == []     # ... or named and matching
    assert genopt.compile_model(model, so_path=so, scheduler="max-ilp") == so and seen == ["max-ilp"]   # another one: rebuilt


def test_shipped_libraries_say_which_scheduler(tmp_path):
    """The real thing, cross-compiled here: one model through compile_model -- its listing passes the scan under max-ilp (round
    5's sources do for all nine models) or the library comes back built under the default strategy; either way the strategy is
    compiled into the library (`Optim.info()["scheduler"]`)."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not shutil.which(hipcc) and not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    model = "lateral_profile"
    if not os.path.exists(os.path.join(ROOT, "tpl_amd", "models", f"{model}.inc")):
        pytest.skip("generated model header not in this tree")
    so = str(tmp_path / "libilqr.so")
    assert genopt.compile_model(model, force=True, so_path=so) == so
    blob = open(so, "rb").read()
    assert b"\"scheduler\": \"%s\"" in blob and (b"max-ilp\0" in blob or b"default\0" in blob)
    assert genopt.built_scheduler(so) in genopt.SCHEDULERS


def test_libtplhip_listings_pass_the_scan():
    """The same scan over every translation unit of libtplhip.so (`make -C tpl_amd/csrc check-isa`: device listings into a
    scratch directory, scanned, removed)."""
    import subprocess
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not shutil.which(hipcc) and not os.path.exists(hipcc):
        pytest.skip("no hipcc here")
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "tpl_amd", "csrc"), "-j8", "check-isa"], capture_output=True, text=True,
                       env=dict(os.environ, PYTHONPATH=ROOT), timeout=1500)
    assert r.returncode == 0 and "0 vector instruction(s) at EXEC = 0" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
