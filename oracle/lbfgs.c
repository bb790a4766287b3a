/* oracle/lbfgs.c -- TEST INFRASTRUCTURE ONLY.  See lbfgs.h for what is restated and how it is pinned. */
#include "lbfgs.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* java.lang.Math.min/max: NaN wins, -0.0 < +0.0 */
static double jmin(double a, double b) {
  if (a != a) return a;
  if (a == 0.0 && b == 0.0) return signbit(a) ? a : b;
  return a <= b ? a : b;
}
static double jmax(double a, double b) {
  if (a != a) return a;
  if (a == 0.0 && b == 0.0) return signbit(a) ? b : a;
  return a >= b ? a : b;
}
static double sq(double v) { return v * v; }
/* O/LBFGS.java:753 */
static double max3(double a, double b, double c) { return a < b ? (b < c ? c : b) : (a < c ? c : a); }

/* ddot (O/LBFGS.java:695-731): the 5-way unrolled body is `t = t + a0*b0 + a1*b1 + ...`, which Java evaluates left to
 * right, so it is the plain ascending sum. */
static double dot(int n, const double *a, const double *b) {
  double t = 0;
  for (int i = 0; i < n; ++i) t = t + a[i] * b[i];
  return t;
}
/* daxpy (O/LBFGS.java:640-688): y += da * x, skipped entirely when da == 0 */
static void axpy(int n, double da, const double *x, double *y) {
  if (da == 0) return;
  for (int i = 0; i < n; ++i) y[i] = y[i] + da * x[i];
}

struct orc_lbfgs {
  int n, m;
  double eps;
  double *x;
  int iter, point, last; /* last = slot the previous iteration wrote (reference: npt / n) */
  double *S, *Y;         /* [m][n] steps and gradient differences (reference: w[ispt..], w[iypt..]) */
  double *rho, *alpha;   /* [m]   (reference: w[n..n+m), w[n+m..n+2m)) */
  double *w;             /* [n]   two-loop work vector, then the gradient at the line-search start (reference: w[0..n)) */
  double *xsave;         /* [n]   line-search start point (reference: diag, reused) */
  double stp, stp1;
  /* More'-Thuente state (O/LBFGS.java:183-220) */
  int nfev, infoc, brackt, stage1;
  double dginit, dgtest, finit, stmin, stmax, width, width1;
  double stx, fx, dgx, sty, fy, dgy;
};

static const double GTOL = 0.9, STPMIN = 1e-20, STPMAX = 1e20, XTOL = 1e-16, FTOL = 0.0001, P5 = 0.5, P66 = 0.66,
                    XTRAPF = 4;
static const int MAXFEV = 20;

orc_lbfgs *orc_lbfgs_new(double *x, int n, int m, double eps) {
  orc_lbfgs *s = calloc(1, sizeof *s);
  s->n = n; s->m = m; s->eps = eps; s->x = x;
  s->S = calloc((size_t)m * n + 1, sizeof(double));
  s->Y = calloc((size_t)m * n + 1, sizeof(double));
  s->rho = calloc(m, sizeof(double));
  s->alpha = calloc(m, sizeof(double));
  s->w = calloc(n + 1, sizeof(double));
  s->xsave = calloc(n + 1, sizeof(double));
  return s;
}
void orc_lbfgs_free(orc_lbfgs *s) {
  if (!s) return;
  free(s->S); free(s->Y); free(s->rho); free(s->alpha); free(s->w); free(s->xsave); free(s);
}

/* O/LBFGS.java:82-157: new search direction into S[point] */
static void begin_iteration(orc_lbfgs *s, const double *g) {
  const int n = s->n, m = s->m;
  s->iter += 1;
  int bound = s->iter - 1;
  if (s->iter != 1) {
    if (s->iter > m) bound = m;
    const double *yl = s->Y + (size_t)s->last * n, *sl = s->S + (size_t)s->last * n;
    double ys = dot(n, yl, sl), yy = dot(n, yl, yl);
    double h0 = ys / yy; /* diag[i], the same for every i (:93-94) */
    int cp = s->point == 0 ? m : s->point;
    s->rho[cp - 1] = 1 / ys;
    for (int i = 0; i < n; ++i) s->w[i] = -g[i];
    cp = s->point;
    for (int k = 0; k < bound; ++k) {
      cp -= 1;
      if (cp == -1) cp = m - 1;
      double sqv = dot(n, s->S + (size_t)cp * n, s->w);
      s->alpha[cp] = s->rho[cp] * sqv;
      axpy(n, -s->alpha[cp], s->Y + (size_t)cp * n, s->w);
    }
    for (int i = 0; i < n; ++i) s->w[i] = h0 * s->w[i];
    for (int k = 0; k < bound; ++k) {
      double yr = dot(n, s->Y + (size_t)cp * n, s->w);
      double beta = s->rho[cp] * yr;
      beta = s->alpha[cp] - beta;
      axpy(n, beta, s->S + (size_t)cp * n, s->w);
      cp += 1;
      if (cp == m) cp = 0;
    }
    memcpy(s->S + (size_t)s->point * n, s->w, sizeof(double) * n);
  }
  s->nfev = 0;
  s->stp = 1;
  if (s->iter == 1) s->stp = s->stp1;
  for (int i = 0; i < n; ++i) s->w[i] = g[i];
}

/* mcsrch entry with info != -1 (O/LBFGS.java:225-268) */
static int ls_begin(orc_lbfgs *s, double f, const double *g) {
  const double *d = s->S + (size_t)s->point * s->n;
  s->infoc = 1;
  s->dginit = 0;
  for (int j = 0; j < s->n; ++j) s->dginit = s->dginit + g[j] * d[j];
  if (s->dginit >= 0) return -1; /* throw new RuntimeException("dginit") */
  s->brackt = 0; s->stage1 = 1; s->nfev = 0;
  s->finit = f;
  s->dgtest = FTOL * s->dginit;
  s->width = STPMAX - STPMIN;
  s->width1 = s->width / P5;
  memcpy(s->xsave, s->x, sizeof(double) * s->n);
  s->stx = 0; s->fx = s->finit; s->dgx = s->dginit;
  s->sty = 0; s->fy = s->finit; s->dgy = s->dginit;
  return 0;
}

/* O/LBFGS.java:272-307: clamp the trial step and move x there */
static void ls_propose(orc_lbfgs *s) {
  if (s->brackt) {
    s->stmin = jmin(s->stx, s->sty);
    s->stmax = jmax(s->stx, s->sty);
  } else {
    s->stmin = s->stx;
    s->stmax = s->stp + XTRAPF * (s->stp - s->stx);
  }
  s->stp = jmax(s->stp, STPMIN);
  s->stp = jmin(s->stp, STPMAX);
  if ((s->brackt && (s->stp <= s->stmin || s->stp >= s->stmax)) || s->nfev >= MAXFEV - 1 || s->infoc == 0 ||
      (s->brackt && s->stmax - s->stmin <= XTOL * s->stmax))
    s->stp = s->stx;
  const double *d = s->S + (size_t)s->point * s->n;
  for (int j = 0; j < s->n; ++j) s->x[j] = s->xsave[j] + s->stp * d[j];
}

/* O/LBFGS.java:446-632.  fx/dx/fy/dy point at either the plain or the modified-function values. */
static void mcstep(orc_lbfgs *s, double *fx, double *dx, double *fy, double *dy, double fp, double dp) {
  int bound;
  double gamma, p, q, r, t, sgnd, stpc, stpf, stpq, theta;
  const double stp = s->stp, stx = s->stx, sty = s->sty;
  s->infoc = 0;
  if ((s->brackt && (stp <= jmin(stx, sty) || stp >= jmax(stx, sty))) || *dx * (stp - stx) >= 0.0 || s->stmax < s->stmin)
    return;
  sgnd = dp * (*dx / fabs(*dx));
  if (fp > *fx) { /* case 1: higher value, minimum bracketed */
    s->infoc = 1; bound = 1;
    theta = 3 * (*fx - fp) / (stp - stx) + *dx + dp;
    t = max3(fabs(theta), fabs(*dx), fabs(dp));
    gamma = t * sqrt(sq(theta / t) - (*dx / t) * (dp / t));
    if (stp < stx) gamma = -gamma;
    p = (gamma - *dx) + theta;
    q = ((gamma - *dx) + gamma) + dp;
    r = p / q;
    stpc = stx + r * (stp - stx);
    stpq = stx + ((*dx / ((*fx - fp) / (stp - stx) + *dx)) / 2) * (stp - stx);
    if (fabs(stpc - stx) < fabs(stpq - stx)) stpf = stpc;
    else stpf = stpc + (stpq - stpc) / 2;
    s->brackt = 1;
  } else if (sgnd < 0.0) { /* case 2: lower value, derivatives of opposite sign */
    s->infoc = 2; bound = 0;
    theta = 3 * (*fx - fp) / (stp - stx) + *dx + dp;
    t = max3(fabs(theta), fabs(*dx), fabs(dp));
    gamma = t * sqrt(sq(theta / t) - (*dx / t) * (dp / t));
    if (stp > stx) gamma = -gamma;
    p = (gamma - dp) + theta;
    q = ((gamma - dp) + gamma) + *dx;
    r = p / q;
    stpc = stp + r * (stx - stp);
    stpq = stp + (dp / (dp - *dx)) * (stx - stp);
    if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
    else stpf = stpq;
    s->brackt = 1;
  } else if (fabs(dp) < fabs(*dx)) { /* case 3: lower value, same sign, derivative shrinks */
    s->infoc = 3; bound = 1;
    theta = 3 * (*fx - fp) / (stp - stx) + *dx + dp;
    t = max3(fabs(theta), fabs(*dx), fabs(dp));
    gamma = t * sqrt(jmax(0, sq(theta / t) - (*dx / t) * (dp / t)));
    if (stp > stx) gamma = -gamma;
    p = (gamma - dp) + theta;
    q = (gamma + (*dx - dp)) + gamma;
    r = p / q;
    if (r < 0.0 && gamma != 0.0) stpc = stp + r * (stx - stp);
    else if (stp > stx) stpc = s->stmax;
    else stpc = s->stmin;
    stpq = stp + (dp / (dp - *dx)) * (stx - stp);
    if (s->brackt) {
      if (fabs(stp - stpc) < fabs(stp - stpq)) stpf = stpc;
      else stpf = stpq;
    } else {
      if (fabs(stp - stpc) > fabs(stp - stpq)) stpf = stpc;
      else stpf = stpq;
    }
  } else { /* case 4: lower value, same sign, derivative does not shrink */
    s->infoc = 4; bound = 0;
    if (s->brackt) {
      theta = 3 * (fp - *fy) / (sty - stp) + *dy + dp;
      t = max3(fabs(theta), fabs(*dy), fabs(dp));
      gamma = t * sqrt(sq(theta / t) - (*dy / t) * (dp / t));
      if (stp > sty) gamma = -gamma;
      p = (gamma - dp) + theta;
      q = ((gamma - dp) + gamma) + *dy;
      r = p / q;
      stpc = stp + r * (sty - stp);
      stpf = stpc;
    } else if (stp > stx) stpf = s->stmax;
    else stpf = s->stmin;
  }
  /* update the interval of uncertainty (:594-611) */
  if (fp > *fx) {
    s->sty = stp; *fy = fp; *dy = dp;
  } else {
    if (sgnd < 0.0) { s->sty = stx; *fy = *fx; *dy = *dx; }
    s->stx = stp; *fx = fp; *dx = dp;
  }
  /* new step, safeguarded (:615-629) */
  stpf = jmin(s->stmax, stpf);
  stpf = jmax(s->stmin, stpf);
  s->stp = stpf;
  if (s->brackt && bound) {
    if (s->sty > s->stx) s->stp = jmin(s->stx + 0.66 * (s->sty - s->stx), s->stp);
    else s->stp = jmax(s->stx + 0.66 * (s->sty - s->stx), s->stp);
  }
}

/* mcsrch re-entered with info == -1 (O/LBFGS.java:310-396): 1 = line search over, 0 = x moved to the next trial */
static int ls_absorb(orc_lbfgs *s, double f, const double *g) {
  const double *d = s->S + (size_t)s->point * s->n;
  int info = 0;
  s->nfev += 1;
  double dg = 0;
  for (int j = 0; j < s->n; ++j) dg = dg + g[j] * d[j];
  double ftest1 = s->finit + s->stp * s->dgtest;
  if ((s->brackt && (s->stp <= s->stmin || s->stp >= s->stmax)) || s->infoc == 0) info = 6;
  if (s->stp == STPMAX && f <= ftest1 && dg <= s->dgtest) info = 5;
  if (s->stp == STPMIN && (f > ftest1 || dg >= s->dgtest)) info = 4;
  if (s->nfev >= MAXFEV) info = 3;
  if (s->brackt && s->stmax - s->stmin <= XTOL * s->stmax) info = 2;
  if (f <= ftest1 && fabs(dg) <= GTOL * (-s->dginit)) info = 1;
  if (info != 0) return 1;
  if (s->stage1 && f <= ftest1 && dg >= jmin(FTOL, GTOL) * s->dginit) s->stage1 = 0;
  if (s->stage1 && f <= s->fx && f > ftest1) {
    double fm = f - s->stp * s->dgtest;
    double fxm = s->fx - s->stx * s->dgtest, fym = s->fy - s->sty * s->dgtest;
    double dgm = dg - s->dgtest;
    double dgxm = s->dgx - s->dgtest, dgym = s->dgy - s->dgtest;
    mcstep(s, &fxm, &dgxm, &fym, &dgym, fm, dgm);
    s->fx = fxm + s->stx * s->dgtest;
    s->fy = fym + s->sty * s->dgtest;
    s->dgx = dgxm + s->dgtest;
    s->dgy = dgym + s->dgtest;
  } else {
    mcstep(s, &s->fx, &s->dgx, &s->fy, &s->dgy, f, dg);
  }
  if (s->brackt) {
    if (fabs(s->sty - s->stx) >= P66 * s->width1) s->stp = s->stx + P5 * (s->sty - s->stx);
    s->width1 = s->width;
    s->width = fabs(s->sty - s->stx);
  }
  ls_propose(s);
  return 0;
}

/* LBFGS.apply (O/LBFGS.java:64-180) */
int orc_lbfgs_apply(orc_lbfgs *s, double f, const double *g) {
  const int n = s->n;
  if (s->iter == 0) {
    for (int i = 0; i < n; ++i) s->S[i] = -g[i] * 1.0; /* -g[i] * diag[i], diag = 1 (:71-74) */
    double gnorm = sqrt(dot(n, g, g));
    s->stp1 = 1 / gnorm;
  } else {
    if (!ls_absorb(s, f, g)) return 0;
    double *sp = s->S + (size_t)s->point * n, *yp = s->Y + (size_t)s->point * n;
    for (int i = 0; i < n; ++i) {
      sp[i] = s->stp * sp[i];
      yp[i] = g[i] - s->w[i];
    }
    s->last = s->point;
    s->point += 1;
    if (s->point == s->m) s->point = 0;
    double gnorm = sqrt(dot(n, g, g));
    double xnorm = sqrt(dot(n, s->x, s->x));
    xnorm = jmax(1.0, xnorm);
    if (gnorm / xnorm <= s->eps) return 1;
  }
  begin_iteration(s, g);
  if (ls_begin(s, f, g) != 0) return -1;
  ls_propose(s);
  return 0;
}

/* Optimizer.lbfgs (O/Optimizer.scala:6-24) */
int orc_optimize(orc_density_fn fn, void *ctx, int nvars, const double *x0, int max_evals, double *x_out) {
  double *out = calloc(nvars + 1, sizeof(double)), *g = calloc(nvars + 1, sizeof(double));
  for (int i = 0; i < nvars; ++i) x_out[i] = x0 ? x0[i] : 0.0;
  orc_lbfgs *lb = orc_lbfgs_new(x_out, nvars, 5, 0.1);
  int evals = 0, rc = 0;
  for (;;) {
    if (evals >= max_evals) { rc = -2; break; }
    if (fn(ctx, x_out, out) != 0) { rc = -3; break; }
    evals += 1;
    for (int i = 0; i < nvars; ++i) g[i] = out[1 + i] * -1;
    int r = orc_lbfgs_apply(lb, out[0] * -1, g);
    if (r == 1) { rc = evals; break; }
    if (r < 0) { rc = -1; break; }
  }
  orc_lbfgs_free(lb); free(out); free(g);
  return rc;
}
