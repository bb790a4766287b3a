"""Device math that is plain C arithmetic (FMAs, ldexp, rint, a table) is checked on the CPU too: the very text of
rh_logit_link in rainier_amd/csrc/device/rh_prelude.hip.h is compiled with the host g++ (contraction off; __builtin_fma is a real
fused multiply-add) and compared with binary128 references.  The GPU tier repeats the check on the device
(tests/test_gpu_parity.py::test_logit_link_softplus_and_sigmoid_ulp)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_MAIN = r'''
#include <quadmath.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
static double ulp_err(double got, __float128 ref) {
  if (ref == 0) return got == 0 ? 0 : 1e9;
  int e; frexp((double)ref, &e);
  __float128 ulp = ldexpq(1, e - 53);
  if (fabs((double)ref) < 2.3e-308) ulp = 4.9e-324Q;
  return (double)(fabsq((__float128)got - ref) / ulp);
}
int main() {
  double worst_sp = 0, worst_sg = 0;
  uint64_t s = 12345;
  for (long i = 0; i < 1500000; i++) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const double a = (double)(s >> 11) / 9007199254740992.0;
    double t;
    switch (i % 9) { case 0: t = (a - 0.5) * 80; break; case 1: t = (a - 0.5) * 2; break; case 2: t = (a - 0.5) * 1500; break;
      case 3: t = (a - 0.5) * 1e-3; break; case 4: t = -a * 12; break; case 5: t = a * 12; break; case 6: t = (a - 0.5) * 16; break;
      case 7: t = (a - 0.5) * 1e-300; break; default: t = (a - 0.5) * 40; }
    double sp, sg; rh_logit_link(t, sp, sg);
    const __float128 T = t, E = expq(-fabsq(T));
    const double e1 = ulp_err(sp, (T > 0 ? T : 0) + log1pq(E)), e2 = ulp_err(sg, T >= 0 ? 1 / (1 + E) : E / (1 + E));
    if (e1 > worst_sp) worst_sp = e1;
    if (e2 > worst_sg) worst_sg = e2;
  }
  // limits: exact at +-inf and beyond the underflow of e^-|t|; NaN reaches the value
  double sp, sg;
  rh_logit_link(800.0, sp, sg); int ok = (sp == 800.0 && sg == 1.0);
  rh_logit_link(-800.0, sp, sg); ok = ok && (sp == 0.0 && sg == 0.0);
  rh_logit_link(__builtin_inf(), sp, sg); ok = ok && (sp == __builtin_inf() && sg == 1.0);
  rh_logit_link(-__builtin_inf(), sp, sg); ok = ok && (sp == 0.0 && sg == 0.0);
  rh_logit_link(0.0, sp, sg); ok = ok && (sg == 0.5 && fabs(sp - 0.69314718055994530942) < 2.3e-16);
  rh_logit_link(__builtin_nan(""), sp, sg); ok = ok && (sp != sp);
  printf("%.4f %.4f %d\n", worst_sp, worst_sg, ok);
  return 0;
}
'''


def test_logit_link_text_on_the_host_is_within_three_ulp(tmp_path):
    pre = open(os.path.join(ROOT, "rainier_amd", "csrc", "device", "rh_prelude.hip.h")).read()
    i = pre.index("__device__ __attribute__((aligned(16))) const double rh_lk_tab")
    j = pre.index("RH_DEV double rh_strict_sqrt")
    src = tmp_path / "lk.cpp"
    src.write_text("#define __device__\n#define RH_DEV static inline\n" + pre[i:j] + _MAIN)
    exe = str(tmp_path / "lk")
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-mfma", str(src), "-o", exe, "-lquadmath"])
    sp, sg, ok = subprocess.check_output([exe]).decode().split()
    assert float(sp) <= 3.0 and float(sg) <= 3.0 and ok == "1", (sp, sg, ok)
    # the table in the header is the generator's output
    tab = subprocess.check_output(["python", os.path.join(ROOT, "tools", "gen_lk_table.py")]).decode()
    assert tab in pre
