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
    assert subprocess.check_output(["python", os.path.join(ROOT, "tools", "gen_lk_table.py"), "--exp"]).decode() in pre     # rh_ex_tab: 2^(j/128) as (hi, lo)


def test_split_wave_reduction_has_the_butterfly_s_bits():
    """rh_wave_sum_split (csrc/device/rh_engine.hip.h, the gather walk's per-group reduction of K scatter sums): at the first log2(K)
    levels of the xor butterfly a lane keeps half of its values and hands the other half to its partner.  Lane-for-lane emulation in
    numpy (fp64 adds are the hardware's; `a + b` commutes): the sum of value kk it leaves in the lanes whose top bits spell kk is the
    very double rh_wave_sum's butterfly (masks 32, 16, ..., 1; own + partner) leaves in every lane."""
    import numpy as np
    rng = np.random.default_rng(7)
    lanes = np.arange(64)

    def butterfly(v):                       # v[64] -> every lane's result
        v = v.copy()
        off = 32
        while off >= 1:
            v = v + v[lanes ^ off]
            off >>= 1
        return v

    def split_level(lo, hi, up, off):       # rh_split_level: the upper lanes keep `hi`, the lower ones `lo`
        keep = np.where(up, hi, lo)
        send = np.where(up, lo, hi)
        return keep + send[lanes ^ off]

    for K in (1, 2, 4, 8):
        for trial in range(50):
            scale = 10.0 ** rng.integers(-8, 8, size=(K, 64))
            v = rng.standard_normal((K, 64)) * scale
            want = np.array([butterfly(v[kk]) for kk in range(K)])
            assert all(np.all(want[kk] == want[kk][0]) for kk in range(K))          # the butterfly leaves the same bits in every lane
            u5, u4, u3 = (lanes & 32) != 0, (lanes & 16) != 0, (lanes & 8) != 0
            if K == 8:
                a = [split_level(v[i], v[i + 4], u5, 32) for i in range(4)]
                b = [split_level(a[0], a[2], u4, 16), split_level(a[1], a[3], u4, 16)]
                x, mine, off = split_level(b[0], b[1], u3, 8), 4 * u5 + 2 * u4 + 1 * u3, 4
            elif K == 4:
                a = [split_level(v[0], v[2], u5, 32), split_level(v[1], v[3], u5, 32)]
                x, mine, off = split_level(a[0], a[1], u4, 16), 2 * u5 + 1 * u4, 8
            elif K == 2:
                x, mine, off = split_level(v[0], v[1], u5, 32), 1 * u5, 16
            else:
                x, mine, off = v[0].copy(), np.zeros(64, dtype=int), 32
            while off >= 1:
                x = x + x[lanes ^ off]
                off >>= 1
            for lane in range(64):
                assert x[lane] == want[int(mine[lane])][0], (K, trial, lane)
            # the lanes the kernel stores from: (lane & (64 / K - 1)) == 0, chain slot = mine
            assert sorted(int(mine[l]) for l in range(64) if (l & (64 // K - 1)) == 0) == list(range(K))
