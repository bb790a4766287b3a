"""The CPU test tier with the host runtime under AddressSanitizer + UBSan (parser, the data-dependent passes of rh_model_create,
the emitter: everything `pytest -m "not gpu"` drives through the C ABI).  Test infrastructure only.

    make -C rainier_amd/csrc sanitize                 # g++ build -> /tmp/librainier_hip_asan.so (58 MB, stays out of the tree)
    python tools/run_sanitized.py -m "not gpu" -q     # arguments go to pytest

The sanitizer runtime has to be in the process before python starts (LD_PRELOAD, together with libstdc++ so that its
__cxa_throw interceptor resolves), so the script re-executes itself once with the environment set; pytest runs with -s because a
sanitizer report written to a captured stderr is lost when the process halts."""
import os
import subprocess
import sys

LIB = os.environ.get("RH_ASAN_LIB", "/tmp/librainier_hip_asan.so")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if os.environ.get("RH_SANITIZED") != "1":
    if not os.path.exists(LIB):
        sys.exit("missing %s: run `make -C rainier_amd/csrc sanitize` first" % LIB)
    rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"]).decode().strip()
    cxx = subprocess.check_output(["gcc", "-print-file-name=libstdc++.so.6"]).decode().strip()
    env = dict(os.environ, RH_SANITIZED="1", LD_PRELOAD="%s %s" % (rt, cxx),
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    os.execve(sys.executable, [sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env)

sys.path.insert(0, ROOT)
os.chdir(ROOT)
import rainier_amd._capi as capi  # noqa: E402

capi.LIB_PATH = LIB
import pytest  # noqa: E402

where = [] if any(os.path.exists(a.split("::")[0]) for a in sys.argv[1:]) else ["tests"]   # explicit test paths replace the default
sys.exit(pytest.main(where + ["-s", "-p", "no:cacheprovider"] + sys.argv[1:]))
