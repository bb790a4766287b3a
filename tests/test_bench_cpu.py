"""bench.py's host-side pieces that need no GPU: the in-run traffic measurement (its rocprofv3 invocation and CSV handling, against a
stand-in profiler on PATH) and its refusal to nest profilers."""
import os
import stat
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FAKE = r'''#!/bin/bash
# stand-in for rocprofv3: records its command line, writes a counter-collection CSV the way rocprofv3 --output-format csv does
out=""; ctr=""; args=("$@")
for ((i = 0; i < ${#args[@]}; i++)); do
  case "${args[$i]}" in -d) out="${args[$((i+1))]}";; --pmc) ctr="${args[$((i+1))]}";; esac
done
echo "$@" >> "$FAKE_LOG"
[ -n "$FAKE_FAIL" ] && exit 3
mkdir -p "$out/host/123"
f="$out/host/123/bench_counter_collection.csv"
echo '"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name","Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name","Counter_Value","Start_Timestamp","End_Timestamp"' > "$f"
v=24000; [ "$ctr" = "WRITE_SIZE" ] && v=800
for n in 1 2 3 4; do echo "$n,$n,1,1,1,1,131072,7,\"rh_grad_fused_kernel\",64,0,0,192,0,96,\"$ctr\",$v.0,100,200" >> "$f"; done
echo "9,9,1,1,1,1,1024,8,\"rh_tick_kernel\",64,0,0,64,0,96,\"$ctr\",5.0,100,200" >> "$f"
exit 0
'''


@pytest.fixture
def fake_profiler(tmp_path, monkeypatch):
    exe = tmp_path / "rocprofv3"
    exe.write_text(FAKE)
    exe.chmod(exe.stat().st_mode | stat.S_IEXEC)
    log = tmp_path / "calls.log"
    monkeypatch.setenv("FAKE_LOG", str(log))
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ["PATH"])
    return log


def _args(**kw):
    a = dict(chains_per_gpu=1024, rows=1_000_000, leapfrog=32, rows_unroll=0, engine="auto", grad_chains=8, grad_unroll=0, grad_splits=0,
             strict=False, no_factor=False)
    a.update(kw)
    return types.SimpleNamespace(**a)


def test_live_traffic_runs_two_separate_pmc_passes_and_averages_the_dominant_kernel(fake_profiler):
    import bench
    traffic, how = bench.live_traffic(_args(), "rh_grad_fused_kernel")
    assert traffic == (24000 + 800) * 1024.0 and how.startswith("measured in this run")
    calls = fake_profiler.read_text().splitlines()
    assert len(calls) == 2 and "--pmc FETCH_SIZE" in calls[0] and "--pmc WRITE_SIZE" in calls[1]
    for c in calls:   # counters in their own passes, kernel trace only, the child does not measure again and is the same workload
        assert "--kernel-trace" in c and "--sys-trace" not in c and "--hip-trace" not in c and "--no-live-traffic" in c
        assert "--chains-per-gpu 1024" in c and "--rows 1000000" in c and "--grad-chains 8" in c and "--strict" not in c
    assert "--strict" in (bench.live_traffic(_args(strict=True), "rh_grad_fused_kernel"), fake_profiler.read_text())[1]


def test_live_traffic_reports_failure_instead_of_a_number(fake_profiler, monkeypatch):
    import bench
    assert bench.live_traffic(_args(), "rh_no_such_kernel")[0] is None          # the kernel never ran under the profiler
    monkeypatch.setenv("FAKE_FAIL", "1")
    traffic, how = bench.live_traffic(_args(), "rh_grad_fused_kernel")
    assert traffic is None and "failed" in how


def test_configs_block_runs_each_leg_as_a_watchdogged_child_and_survives_a_leg_that_dies(monkeypatch):
    """The `configs` block: every BASELINE configuration is a child `bench.py --workload ...` under a watchdog (a leg that hangs or dies
    must not take the judged line with it); the default plan keeps cfg 5's NUTS leg short and cfg 4's at DefaultConfig's own mass
    windows (warm-up 150), `--long-configs` swaps in cfg 5's converging 150 + 400."""
    import json
    import subprocess
    import bench
    seen = []

    def fake_run(cmd, capture_output, text, timeout):
        w = cmd[cmd.index("--workload") + 1]
        seen.append(dict(workload=w, sampler=cmd[cmd.index("--sampler") + 1], steps=int(cmd[cmd.index("--steps") + 1]),
                         warmup=int(cmd[cmd.index("--warmup") + 1]), chains=int(cmd[cmd.index("--chains-per-gpu") + 1]), timeout=timeout))
        if w == "cfg3":
            raise subprocess.TimeoutExpired(cmd, timeout)
        if w == "cfg1":
            return types.SimpleNamespace(returncode=1, stdout="", stderr="boom")
        line = json.dumps({"value": 1.0, "unit": "leapfrog steps/s", "steps": 1, "warmup": 1, "ms_per_step": 1.0, "config": {"workload": w}, "roofline": {"frac": 0.5}})
        return types.SimpleNamespace(returncode=0, stdout="noise\n" + line + "\n", stderr="")
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    out = bench.all_configs()
    assert len(out) == 8 and len(seen) == 8
    assert "error" in out["cfg1_funnel_hmc5_1024"] and "watchdog" in out["cfg3_eight_schools_ehmc_1024"]["error"]
    assert out["cfg2_default_config_ehmc_diag_mass_1024"]["roofline"]["frac"] == 0.5
    by = {(s["workload"], s["sampler"]): s for s in seen}
    assert by[("cfg4", "default")]["warmup"] == 150 and by[("cfg4", "default")]["chains"] == 256      # DefaultConfig's own mass windows
    assert by[("cfg5c", "default")]["warmup"] < 150 and by[("cfg5c", "default")]["chains"] == 1024     # the converging leg is --long-configs'
    assert all(0 < s["timeout"] <= 640 for s in seen)
    seen.clear()
    bench.all_configs(budget_s=1100.0, long_legs=True)
    by = {(s["workload"], s["sampler"]): s for s in seen}
    assert by[("cfg5c", "default")]["warmup"] == 150 and by[("cfg5c", "default")]["steps"] >= 100
    # a spent budget skips the remaining legs instead of starting them
    seen.clear()
    out = bench.all_configs(budget_s=0.0)
    assert not seen and all("skipped" in v for v in out.values())
