/* oracle/closed_form.c -- TEST / BENCH INFRASTRUCTURE ONLY (cpu_baseline's second figure).
 *
 * What an optimising CPU compiler makes of cfg 2's compiled log-density + gradient when it is written out by hand:
 * the un-inlined streaming form of the README linear regression (theta = (s, a, b0, b1, b2),
 * row term -(y - a - b.x)^2 e^{-2s}/2 - s - log(2 pi)/2), one pass over the four columns, five running sums --
 * exactly the work rh_grad_kernel does per chain.  Built -O3 with AVX2 + FMA allowed ("speed build");
 * it is an upper bound for any JVM on the same cores, reported beside the faithful interpreter figure, never a target.
 */
#include <math.h>

void orc_linreg_streamed(const double *y, const double *x0, const double *x1, const double *x2, long n,
                         const double *th, double *out /* [6]: logp, d/ds, d/da, d/db0, d/db1, d/db2 */) {
  const double s = th[0], a = th[1], b0 = th[2], b1 = th[3], b2 = th[4];
  double sr2 = 0, sr = 0, s0 = 0, s1 = 0, s2 = 0;
  for (long i = 0; i < n; i++) {
    const double r = y[i] - (a + b0 * x0[i] + b1 * x1[i] + b2 * x2[i]);
    sr2 += r * r; sr += r; s0 += r * x0[i]; s1 += r * x1[i]; s2 += r * x2[i];
  }
  const double iv = exp(-2.0 * s), c = 0.91893853320467274178;
  const double prior = (s - exp(s)) + (-0.5 * a * a - c) + (-0.5 * b0 * b0 - c) + (-0.5 * b1 * b1 - c) + (-0.5 * b2 * b2 - c);
  out[0] = prior + (-0.5 * iv * sr2 - (double)n * (s + c));
  out[1] = (1.0 - exp(s)) + (iv * sr2 - (double)n);
  out[2] = -a + iv * sr;
  out[3] = -b0 + iv * s0;
  out[4] = -b1 + iv * s1;
  out[5] = -b2 + iv * s2;
}

/* `reps` gradient evaluations at slightly different points (so the loop cannot be hoisted); returns a checksum */
double orc_linreg_streamed_reps(const double *y, const double *x0, const double *x1, const double *x2, long n, int reps) {
  double th[5] = {-0.3, 0.5, 1.0, -2.0, 0.5}, out[6], acc = 0;
  for (int r = 0; r < reps; r++) {
    th[1] += 1e-9;
    orc_linreg_streamed(y, x0, x1, x2, n, th, out);
    acc += out[0] + out[3];
  }
  return acc;
}
