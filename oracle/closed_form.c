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

/* ---- the INLINED form: what real Rainier runs for this model -------------------------------------------------------
 * compute/Target.scala:20-24 + TargetGroup.inlinable (Target.scala:136-207) + PartialEvaluator.inline
 * (compute/PartialEvaluator.scala:90-97): the squared residual Line (y - a - b.x)^2 has 5 terms, so
 * LogLineOps.distribute (compute/LogLineOps.scala:47-91, limit DistributeToMaxTerms = 20 > 15) expands it into 15
 * products whose coefficients are columns; every term is then a function of either the data or the parameters, the
 * target is "inlinable", and the row sum is folded into scalar coefficients at compile time.  The compiled density is
 * O(1) in the row count (rainier-benchmark/benchmarks.txt: Normal, 1 us per gradient at every N).  This is that
 * function, written by hand over the 15 sufficient statistics: S = [n, Sy, Sx_k, Syy, Syx_k, Sx_jx_k (j <= k)]. */
void orc_linreg_suffstats(const double *y, const double *x0, const double *x1, const double *x2, long n, double *S /* [15] */) {
  const double *x[3] = {x0, x1, x2};
  for (int i = 0; i < 15; i++) S[i] = 0;
  S[0] = (double)n;
  for (long i = 0; i < n; i++) {
    S[1] += y[i]; S[5] += y[i] * y[i];
    int p = 9;
    for (int j = 0; j < 3; j++) {
      S[2 + j] += x[j][i]; S[6 + j] += y[i] * x[j][i];
      for (int k = j; k < 3; k++) S[p++] += x[j][i] * x[k][i];
    }
  }
}
void orc_linreg_inlined(const double *S, const double *th, double *out /* [6] */) {
  const double s = th[0], a = th[1], b[3] = {th[2], th[3], th[4]};
  const double n = S[0], Sy = S[1], *Sx = S + 2, Syy = S[5], *Syx = S + 6;
  const double Sxx[3][3] = {{S[9], S[10], S[11]}, {S[10], S[12], S[13]}, {S[11], S[13], S[14]}};
  /* sum r, sum r x_k, sum r^2 with r = y - a - b.x, from the statistics */
  double bSx = 0, bSyx = 0, bSxxb = 0, Sxxb[3];
  for (int j = 0; j < 3; j++) {
    Sxxb[j] = Sxx[j][0] * b[0] + Sxx[j][1] * b[1] + Sxx[j][2] * b[2];
    bSx += b[j] * Sx[j]; bSyx += b[j] * Syx[j]; bSxxb += b[j] * Sxxb[j];
  }
  const double sr = Sy - n * a - bSx;
  const double sr2 = Syy - 2 * a * Sy - 2 * bSyx + n * a * a + 2 * a * bSx + bSxxb;
  const double iv = exp(-2.0 * s), c = 0.91893853320467274178, es = exp(s);
  const double prior = (s - es) + (-0.5 * a * a - c) + (-0.5 * b[0] * b[0] - c) + (-0.5 * b[1] * b[1] - c) + (-0.5 * b[2] * b[2] - c);
  out[0] = prior + (-0.5 * iv * sr2 - n * (s + c));
  out[1] = (1.0 - es) + (iv * sr2 - n);
  out[2] = -a + iv * sr;
  for (int j = 0; j < 3; j++) out[3 + j] = -b[j] + iv * (Syx[j] - a * Sx[j] - Sxxb[j]);
}
/* `steps` leapfrog steps of one chain on the inlined density (identity mass; one gradient per step, the count this
 * repo uses -- the reference's LeapFrog spends two): the whole inner loop of an HMC trajectory.  Returns a checksum. */
double orc_linreg_inlined_leapfrog(const double *S, long steps, double eps) {
  double q[5] = {-0.3, 0.5, 1.0, -2.0, 0.5}, p[5] = {0.1, -0.2, 0.3, -0.1, 0.2}, g[6];
  orc_linreg_inlined(S, q, g);
  for (long t = 0; t < steps; t++) {
    for (int i = 0; i < 5; i++) { p[i] += eps * g[1 + i]; q[i] += eps * p[i]; }
    orc_linreg_inlined(S, q, g);
    if ((t & 31) == 31) for (int i = 0; i < 5; i++) p[i] = -p[i] * 0.5;  /* keep the trajectory bounded */
  }
  return g[0] + q[0] + p[1];
}
