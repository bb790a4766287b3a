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
