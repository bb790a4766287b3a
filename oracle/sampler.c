/* oracle/sampler.c -- TEST INFRASTRUCTURE ONLY.  See sampler.h. */
#include "sampler.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- S/RNG.scala:6-26 ------------------------------------------------------------------ */
static double rng_uniform(jrandom *r) { return jrandom_next_double(r); }
static double rng_normal(jrandom *r) { return jrandom_next_gaussian(r); }
static int rng_int(jrandom *r, int until) { /* RNG.int: min((u*until).toInt, until-1) */
  int v = (int)(rng_uniform(r) * until);
  return v < until - 1 ? v : until - 1;
}

/* ---- S/Stats.scala:19-58 RingBuffer ------------------------------------------------------ */
typedef struct { int full, i, size; double *buf; } ringbuf;
static void ring_init(ringbuf *rb, int size) { rb->full = 0; rb->i = 0; rb->size = size; rb->buf = calloc(size, sizeof(double)); }
static void ring_add(ringbuf *rb, double v) { /* :24-30 pre-increment, slot 0 written last */
  rb->i += 1;
  if (rb->i == rb->size) rb->full = 1;
  rb->i = rb->i % rb->size;
  rb->buf[rb->i] = v;
}
static double ring_sample(ringbuf *rb, jrandom *r) { /* :40-45 */
  if (rb->full) return rb->buf[rng_int(r, rb->size)];
  return rb->buf[rng_int(r, rb->i + 1)];
}

/* ---- S/LeapFrog.scala ------------------------------------------------------------------ */
struct orc_leapfrog {
  orc_density_fn f; void *ctx; jrandom *rng; int math_mode;
  int n;          /* nVars */
  double *pqBuf;  /* [2n+1]  :118-129 layout p | q | potential */
  double *buf;    /* [n] */
  double *out;    /* [n+1] density outputs */
  double prevH;
  int64_t gradientEvaluations; /* Stats.gradientEvaluations :199 */
  int64_t leapfrogSteps;       /* not in the reference: number of newQs calls */
  int64_t iterations; int64_t accepted; double sumAccept;
  /* Stats.energyVariance (a VarianceEstimator(1), S/MassMatrixEstimator.scala:69-91) and energyTransitions2 (S/Stats.scala:14-16) */
  int64_t eSamples; double eMean, eRaw, eTrans2;
  int density_error;
  const double *dense; /* DenseMassMatrix.elements [n*n] when the current mass is dense (mass == dense) */
  double *chol_u;      /* its choleskyUpperTriangular, packed */
};

orc_leapfrog *orc_lf_new(orc_density_fn f, void *ctx, int nvars, jrandom *rng, int math_mode) {
  orc_leapfrog *lf = calloc(1, sizeof(*lf));
  lf->f = f; lf->ctx = ctx; lf->rng = rng; lf->n = nvars; lf->math_mode = math_mode;
  lf->pqBuf = calloc(2 * nvars + 1, sizeof(double));
  lf->buf = calloc(nvars, sizeof(double));
  lf->out = calloc(nvars + 1, sizeof(double));
  return lf;
}
void orc_lf_free(orc_leapfrog *lf) { if (lf) { free(lf->pqBuf); free(lf->buf); free(lf->out); free(lf); } }

static void lf_copy(const orc_leapfrog *lf, const double *src, double *dst) { memcpy(dst, src, sizeof(double) * (2 * lf->n + 1)); }

/* :194-200 */
static void copyQsAndUpdateDensity(orc_leapfrog *lf) {
  memcpy(lf->buf, lf->pqBuf + lf->n, sizeof(double) * lf->n);
  if (lf->f(lf->ctx, lf->buf, lf->out)) lf->density_error = 1;
  lf->gradientEvaluations += 1;
}
/* ---- S/MassMatrix.scala:33-117 DenseMassMatrix ---------------------------------------------- */
static int tri(int k) { return (k * (k + 1)) / 2; }
void orc_square_multiply(const double *matrix, const double *vector, int n, double *out) { /* :34-48 */
  for (int i = 0; i < n; i++) {
    double y = 0.0;
    for (int j = 0; j < n; j++) y += vector[j] * matrix[(i * n) + j];
    out[i] = y;
  }
}
void orc_upper_triangular_solve(const double *packed, const double *vector, int n, double *out) { /* :52-69 */
  int i = n - 1;
  int m = tri(i + 1) - 1;
  while (i >= 0) {
    int j = n - 1;
    double dot = 0.0;
    while (j > i) { dot += out[j] * packed[m]; j -= 1; m -= 1; }
    out[i] = (vector[i] - dot) / packed[m];
    i -= 1; m -= 1;
  }
}
void orc_cholesky_upper(const double *matrix, int n, double *upper) { /* :74-116 */
  double *lower = calloc(tri(n) ? tri(n) : 1, sizeof(double));
  int l = 0;
  for (int i = 0; i < n; i++)
    for (int k = 0; k <= i; k++) {
      double sum = 0.0;
      for (int j = 0; j < k; j++) sum += lower[tri(i) + j] * lower[tri(k) + j];
      double x = matrix[(i * n) + k] - sum;
      if (i == k) lower[l] = sqrt(x);
      else { double diag = lower[tri(k + 1) - 1]; lower[l] = (1.0 / diag * x); }
      l += 1;
    }
  l = 0;
  for (int i = 0; i < n; i++)
    for (int k = 0; k < (n - i); k++) { upper[l] = lower[tri(k + i) + i]; l += 1; }
  free(lower);
}

/* :202-216 (Identity, Diagonal, Dense) */
static void velocity(const orc_leapfrog *lf, const double *in, double *out, const double *mass) {
  if (!mass) memcpy(out, in, sizeof(double) * lf->n);
  else if (mass == lf->dense) orc_square_multiply(mass, in, lf->n, out);
  else for (int i = 0; i < lf->n; i++) out[i] = in[i] * mass[i];
}
/* :218-227 */
static double dot(const double *x, const double *y, int n) {
  double k = 0.0;
  for (int i = 0; i < n; i++) k += (x[i] * y[i]);
  return k;
}
/* :131-136 */
static double energy(orc_leapfrog *lf, const double *params, const double *mass) {
  double potential = params[2 * lf->n];
  velocity(lf, params, lf->buf, mass);
  double kinetic = dot(lf->buf, params, lf->n) / 2.0;
  return potential + kinetic;
}
/* :138-142 */
static double logAcceptanceProb(const orc_leapfrog *lf, double deltaH) {
  if (isnan(deltaH)) return jm_log(lf->math_mode, 0.0);
  return (-deltaH) < 0.0 ? (-deltaH) : 0.0; /* (-deltaH).min(0.0) */
}
/* :144-151 */
static void newQs(orc_leapfrog *lf, double stepSize, const double *mass) {
  velocity(lf, lf->pqBuf, lf->buf, mass);
  for (int i = 0; i < lf->n; i++) lf->pqBuf[i + lf->n] += (stepSize * lf->buf[i]);
  lf->leapfrogSteps += 1;
}
/* :165-173 */
static void fullPs(orc_leapfrog *lf, double stepSize) {
  copyQsAndUpdateDensity(lf);
  for (int i = 0; i < lf->n; i++) lf->pqBuf[i] += stepSize * lf->out[i + 1];
}
/* :153-163 */
static void initialHalfThenFullStep(orc_leapfrog *lf, double stepSize, const double *mass) {
  fullPs(lf, stepSize / 2.0);
  newQs(lf, stepSize, mass);
  copyQsAndUpdateDensity(lf);
  lf->pqBuf[2 * lf->n] = lf->out[0] * -1;
}
/* :175-184 */
static void twoFullSteps(orc_leapfrog *lf, double stepSize, const double *mass) {
  fullPs(lf, stepSize);
  newQs(lf, stepSize, mass);
  copyQsAndUpdateDensity(lf);
  lf->pqBuf[2 * lf->n] = lf->out[0] * -1;
}
/* :186-188 */
static void finalHalfStep(orc_leapfrog *lf, double stepSize) { fullPs(lf, stepSize / 2.0); }

/* :229-251 */
static void initializePs(orc_leapfrog *lf, double *params, const double *mass) {
  for (int i = 0; i < lf->n; i++) lf->buf[i] = rng_normal(lf->rng);
  if (!mass) memcpy(params, lf->buf, sizeof(double) * lf->n);
  else if (mass == lf->dense) orc_upper_triangular_solve(lf->chol_u, lf->buf, lf->n, params);
  else for (int i = 0; i < lf->n; i++) params[i] = lf->buf[i] / sqrt(mass[i]); /* stdDevs S/MassMatrix.scala:10-12 */
}

/* :14-22 */
double orc_lf_try_stepping(orc_leapfrog *lf, const double *params, double stepSize, const double *mass) {
  lf_copy(lf, params, lf->pqBuf);
  initialHalfThenFullStep(lf, stepSize, mass);
  finalHalfStep(lf, stepSize);
  double deltaH = energy(lf, lf->pqBuf, mass) - energy(lf, params, mass);
  return logAcceptanceProb(lf, deltaH);
}
/* :24-33 */
void orc_lf_take_steps(orc_leapfrog *lf, int l, double stepSize, const double *mass) {
  initialHalfThenFullStep(lf, stepSize, mass);
  int i = 1;
  while (i < l) { twoFullSteps(lf, stepSize, mass); i += 1; }
  finalHalfStep(lf, stepSize);
}
/* :35-47 */
int orc_lf_is_uturn(orc_leapfrog *lf, const double *params) {
  double out = 0.0;
  for (int i = 0; i < lf->n; i++) out += (lf->pqBuf[i + lf->n] - params[i + lf->n]) * lf->pqBuf[i];
  if (isnan(out)) return 1;
  return out < 0;
}
/* :52-59 */
void orc_lf_start_iteration(orc_leapfrog *lf, double *params, const double *mass) {
  lf->prevH = energy(lf, params, mass);
  initializePs(lf, params, mass);
  lf_copy(lf, params, lf->pqBuf);
}
/* :61-82 */
double orc_lf_finish_iteration(orc_leapfrog *lf, double *params, const double *mass) {
  double startH = energy(lf, params, mass);
  double endH = energy(lf, lf->pqBuf, mass);
  double deltaH = endH - startH;
  double a = logAcceptanceProb(lf, deltaH);
  double h; /* the energy the chain ends the iteration with (:68-74) */
  if (a > jm_log(lf->math_mode, rng_uniform(lf->rng))) { lf_copy(lf, lf->pqBuf, params); lf->accepted += 1; h = endH; }
  else h = startH;
  { /* stats.energyVariance.update(h); stats.energyTransitions2 += Math.pow(h - prevH, 2) */
    lf->eSamples += 1;
    const double oldDiff = h - lf->eMean;
    lf->eMean += oldDiff / (double)lf->eSamples;
    const double newDiff = h - lf->eMean;
    lf->eRaw += oldDiff * newDiff;
    lf->eTrans2 += (h - lf->prevH) * (h - lf->prevH);
  }
  lf->iterations += 1;
  lf->sumAccept += jm_exp(lf->math_mode, a);
  return a;
}
/* :102-116 */
void orc_lf_initialize(orc_leapfrog *lf, const double *mass, double *params) {
  int n = lf->n;
  memset(lf->pqBuf, 0, sizeof(double) * (2 * n + 1));
  for (int i = n; i < 2 * n; i++) lf->pqBuf[i] = rng_normal(lf->rng);
  copyQsAndUpdateDensity(lf);
  lf->pqBuf[2 * n] = lf->out[0] * -1;
  lf_copy(lf, lf->pqBuf, params);
  initializePs(lf, params, mass);
}
static void lf_snapshot(const orc_leapfrog *lf, double *out) { lf_copy(lf, lf->pqBuf, out); }  /* :84-85 */
static void lf_restore(orc_leapfrog *lf, const double *in) { lf_copy(lf, in, lf->pqBuf); }     /* :87-88 */

/* ---- S/DualAvg.scala --------------------------------------------------------------------- */
typedef struct { /* final class DualAvg :43-78 */
  double delta, logStepSize, logStepSizeBar, avgError, shrinkageTarget;
  int iteration; int math_mode;
} dualavg;
static dualavg dualavg_new(double delta, double stepSize, int mode) { /* :80-89 */
  dualavg d; d.delta = delta; d.logStepSize = jm_log(mode, stepSize); d.logStepSizeBar = 0.0; d.avgError = 0.0;
  d.iteration = 0; d.shrinkageTarget = jm_log(mode, 10 * stepSize); d.math_mode = mode; return d;
}
static void dualavg_update(dualavg *d, double logAcceptanceProb) { /* :58-77 */
  const double stepSizeUpdateDenom = 0.05; const int acceptanceProbUpdateDenom = 10;
  double newAcceptanceProb = jm_exp(d->math_mode, logAcceptanceProb);
  d->iteration = d->iteration + 1;
  double avgErrorMultiplier = 1.0 / ((double)d->iteration + acceptanceProbUpdateDenom);
  double stepSizeMultiplier = jm_pow_neg075(d->math_mode, (double)d->iteration); /* Math.pow(it, -decayRate) */
  d->avgError = ((1.0 - avgErrorMultiplier) * d->avgError + (avgErrorMultiplier * (d->delta - newAcceptanceProb)));
  d->logStepSize = (d->shrinkageTarget - (d->avgError * jm_sqrt((double)d->iteration) / stepSizeUpdateDenom));
  d->logStepSizeBar = (stepSizeMultiplier * d->logStepSize + (1.0 - stepSizeMultiplier) * d->logStepSizeBar);
}
/* :27-41 */
static double findReasonableStepSize(orc_leapfrog *lf, const double *params, const double *mass) {
  int mode = lf->math_mode;
  double stepSize = 1.0;
  double logAcceptanceProb = orc_lf_try_stepping(lf, params, stepSize, mass);
  double exponent = (logAcceptanceProb > jm_log(mode, 0.5)) ? 1.0 : -1.0;
  double doubleOrHalf = exponent > 0 ? 2.0 : 0.5; /* Math.pow(2, +-1) is exact */
  while (stepSize != 0.0 && (exponent * logAcceptanceProb > -exponent * jm_log(mode, 2))) {
    stepSize *= doubleOrHalf;
    logAcceptanceProb = orc_lf_try_stepping(lf, params, stepSize, mass);
  }
  return stepSize;
}

/* ---- S/MassMatrixEstimator.scala:52-113 VarianceEstimator -------------------------------- */
typedef struct { int samples, size; double *mean, *raw, *oldDiff, *newDiff; } varest;
static void varest_init(varest *v, int size) {
  v->samples = 0; v->size = size;
  v->mean = calloc(size, sizeof(double)); v->raw = calloc(size, sizeof(double));
  v->oldDiff = calloc(size, sizeof(double)); v->newDiff = calloc(size, sizeof(double));
}
static void varest_free(varest *v) { free(v->mean); free(v->raw); free(v->oldDiff); free(v->newDiff); }
static void varest_reset(varest *v) { /* :60-67 -- does NOT reset samples */
  for (int i = 0; i < v->size; i++) { v->mean[i] = 0.0; v->raw[i] = 0.0; }
}
static void varest_update(varest *v, const double *sample) { /* :69-84 */
  v->samples += 1;
  for (int i = 0; i < v->size; i++) v->oldDiff[i] = sample[i] - v->mean[i];
  for (int i = 0; i < v->size; i++) v->mean[i] += (v->oldDiff[i] / (double)v->samples);
  for (int i = 0; i < v->size; i++) v->newDiff[i] = sample[i] - v->mean[i];
  for (int j = 0; j < v->size; j++) v->raw[j] += v->oldDiff[j] * v->newDiff[j];
}
static void varest_variance(const varest *v, double *elements) { /* :93-101 */
  for (int i = 0; i < v->size; i++) elements[i] = v->raw[i] / (double)v->samples;
}

/* ---- S/MassMatrix.scala:126-173 WindowedMassMatrixTuner (Diagonal) ------------------------ */
typedef struct {
  int kind; int windowSize; double windowExpansion; int skipFirst, skipLast;
  int i, j, totalIterations; varest est;
  double *cov; /* CovarianceEstimator.cov [n*n] (S/MassMatrixEstimator.scala:9-50) */
} masstuner;
/* returns 1 and writes `mass` when a new matrix is produced (Some(m)) */
static int masstuner_update(masstuner *t, const double *sample, double *mass) { /* :147-164 */
  if (t->kind != ORC_MASS_DIAG_WINDOWED && t->kind != ORC_MASS_DENSE_WINDOWED) return 0;
  t->j += 1;
  if (t->j < t->skipFirst || (t->totalIterations - t->j) < t->skipLast) return 0;
  t->i += 1;
  varest_update(&t->est, sample);
  const int n = t->est.size;
  if (t->kind == ORC_MASS_DENSE_WINDOWED) /* CovarianceEstimator.update :23-36 */
    for (int j = 0; j < n; j++)
      for (int k = 0; k < n; k++) t->cov[j * n + k] += t->est.newDiff[j] * t->est.oldDiff[k];
  if (t->i == t->windowSize) {
    t->i = 0;
    t->windowSize = (int)(t->windowSize * t->windowExpansion);
    if (t->kind == ORC_MASS_DENSE_WINDOWED) { /* covariance :38-47, reset :13-21 */
      const double z = (double)(t->est.samples - 1);
      for (int e = 0; e < n * n; e++) { mass[e] = t->cov[e] / z; t->cov[e] = 0.0; }
    } else varest_variance(&t->est, mass);
    varest_reset(&t->est);
    return 1;
  }
  return 0;
}


/* ---- NUTS (EXTENSION: the reference has no NUTS; parity against the reference is unpinned) ----
 * Specification = iterative multinomial NUTS: Phan, Pradhan, Jankowiak, "Composable Effects for Flexible and Accelerated
 * Probabilistic Programming in NumPyro" (2019), Appendix A (the algorithm NumPyro/BlackJAX ship): trajectory doubling
 * with a uniformly random direction, leaves generated one leapfrog step at a time, uniform progressive sampling inside a
 * subtree, biased progressive sampling between the old tree and the new subtree, U-turn checks of every aligned
 * power-of-two sub-trajectory through O(depth) momentum checkpoints, divergence when H - H0 > 1000.
 * Draws per iteration from the chain's java.util.Random stream: n gaussians (momenta), then per doubling 1 uniform
 * (direction), 1 uniform per leaf (subtree sampling) and 1 uniform for the merge.  It plugs in as a `Sampler`
 * (S/Sampler.scala:52-62): warmup/run return log(mean leaf acceptance) to the step-size tuner.
 * This C text and rh_engine.hip.h's NUTS states are two statements of the same algorithm, compared bit for bit. */
#define ORC_NUTS_MAXD 12
static double logaddexp_det(int mode, double a, double b) {
  double m = a > b ? a : b;
  if (m == -INFINITY) return -INFINITY;
  return m + jm_log(mode, jm_exp(mode, a - m) + jm_exp(mode, b - m));
}
/* _is_turning: v = M^-1 r; r_sum' = r_sum - (r_left + r_right)/2; turning iff v_left.r_sum' <= 0 or v_right.r_sum' <= 0 */
static int nuts_is_turning(const orc_leapfrog *lf, int n, const double *mass, const double *rl, const double *rr, const double *rsum) {
  double dl = 0.0, dr = 0.0;
  double *vl = malloc(sizeof(double) * n), *vr = malloc(sizeof(double) * n);
  velocity(lf, rl, vl, mass); velocity(lf, rr, vr, mass);
  for (int i = 0; i < n; i++) {
    double adj = rsum[i] - (rl[i] + rr[i]) / 2.0;
    dl += vl[i] * adj; dr += vr[i] * adj;
  }
  free(vl); free(vr);
  return !(dl > 0.0) || !(dr > 0.0); /* NaN counts as turning */
}
static double nuts_iteration(const orc_config *cfg, double *params, orc_leapfrog *lf, double stepSize, const double *mass) {
  const int n = lf->n, mode = cfg->math_mode, sz = 2 * n + 1;
  orc_lf_start_iteration(lf, params, mass); /* fresh momenta; pqBuf := params */
  const double H0 = energy(lf, params, mass);
  double *L = malloc(sizeof(double) * sz), *R = malloc(sizeof(double) * sz), *prop = malloc(sizeof(double) * sz);
  double *subprop = malloc(sizeof(double) * sz), *rsum = calloc(n, sizeof(double)), *srsum = calloc(n, sizeof(double));
  double *ckr = calloc((size_t)ORC_NUTS_MAXD * n, sizeof(double)), *ckrs = calloc((size_t)ORC_NUTS_MAXD * n, sizeof(double));
  memcpy(L, params, sizeof(double) * sz); memcpy(R, params, sizeof(double) * sz); memcpy(prop, params, sizeof(double) * sz);
  memcpy(rsum, params, sizeof(double) * n);
  double tree_logw = 0.0, sum_accept = 0.0; long nleaf_total = 0;
  int depth = 0, stop = 0;
  while (depth < cfg->nuts_max_depth && !stop) {
    const int going_right = rng_uniform(lf->rng) > 0.5;
    const double eps = going_right ? stepSize : -stepSize;
    lf_restore(lf, going_right ? R : L);
    double sub_logw = -INFINITY; int sub_turning = 0, sub_div = 0;
    for (int i = 0; i < n; i++) srsum[i] = 0.0;
    const long nleaf = 1L << depth;
    for (long leaf = 0; leaf < nleaf && !sub_turning && !sub_div; leaf++) {
      orc_lf_take_steps(lf, 1, eps, mass); /* one leapfrog step in the chosen direction */
      double delta = energy(lf, lf->pqBuf, mass) - H0;
      if (isnan(delta)) delta = INFINITY;
      const double leaf_logw = -delta;
      if (delta > 1000.0) sub_div = 1;
      sum_accept += delta <= 0.0 ? 1.0 : jm_exp(mode, -delta); nleaf_total++;
      const double new_logw = logaddexp_det(mode, sub_logw, leaf_logw);
      const double u = rng_uniform(lf->rng);
      if (leaf == 0 || u < jm_exp(mode, leaf_logw - new_logw)) lf_snapshot(lf, subprop);
      sub_logw = new_logw;
      for (int i = 0; i < n; i++) srsum[i] += lf->pqBuf[i];
      /* checkpoints (NumPyro _leaf_idx_to_ckpt_idxs) */
      int idx_max = 0; for (long x = leaf >> 1; x > 0; x >>= 1) idx_max += (int)(x & 1);
      int nsub = 0; for (long x = leaf; (x & 1) != 0; x >>= 1) nsub++;
      const int idx_min = idx_max - nsub + 1;
      if ((leaf & 1) == 0) {
        memcpy(ckr + (size_t)idx_max * n, lf->pqBuf, sizeof(double) * n);
        memcpy(ckrs + (size_t)idx_max * n, srsum, sizeof(double) * n);
      } else {
        for (int k = idx_max; k >= idx_min && !sub_turning; k--) {
          double *sub = lf->buf; /* subtree r_sum = r_sum - r_sum_ckpt + r_ckpt */
          for (int i = 0; i < n; i++) sub[i] = srsum[i] - ckrs[(size_t)k * n + i] + ckr[(size_t)k * n + i];
          double *tmp = malloc(sizeof(double) * n); memcpy(tmp, sub, sizeof(double) * n);
          sub_turning = nuts_is_turning(lf, n, mass, ckr + (size_t)k * n, lf->pqBuf, tmp);
          free(tmp);
        }
      }
    }
    if (sub_turning || sub_div) { stop = 1; break; }
    /* merge: biased progressive sampling, then the U-turn check of the doubled tree */
    const double u = rng_uniform(lf->rng);
    if (u < jm_exp(mode, sub_logw - tree_logw)) memcpy(prop, subprop, sizeof(double) * sz);
    tree_logw = logaddexp_det(mode, tree_logw, sub_logw);
    for (int i = 0; i < n; i++) rsum[i] += srsum[i];
    lf_snapshot(lf, going_right ? R : L);
    depth++;
    if (nuts_is_turning(lf, n, mass, L, R, rsum)) stop = 1;
  }
  /* the draw: position, potential (and its gradient) of the selected point; momentum is irrelevant from here on */
  memcpy(params + n, prop + n, sizeof(double) * (n + 1));
  lf->iterations += 1; lf->accepted += 1;
  const double mean_acc = nleaf_total ? sum_accept / (double)nleaf_total : 0.0;
  lf->sumAccept += mean_acc;
  free(L); free(R); free(prop); free(subprop); free(rsum); free(srsum); free(ckr); free(ckrs);
  return jm_log(mode, mean_acc);
}

/* ---- NUTS, second statement: RECURSIVE tree building (cross-validation of the checkpoint bookkeeping above) ----------
 * The same transition written the way Hoffman & Gelman (2014, Alg. 6) and Stan's multinomial variant are usually stated:
 * build_tree(depth) = build_tree(depth-1) ++ build_tree(depth-1), every node keeps its first and last momentum and the
 * sum of its momenta, the U-turn criterion of a node is evaluated from those three directly, a stopped child stops the
 * parent.  There are NO checkpoints and no leaf-index arithmetic here -- that is the point: nuts_iteration's
 * _leaf_idx_to_ckpt_idxs indexing (idx_max / idx_min / "sub = srsum - ckrs[k] + ckr[k]") must enumerate exactly the nodes
 * this recursion visits, in the same (post-)order, and stop at the same leaf.  Leaves are visited left to right by both,
 * so the per-leaf uniforms of the progressive sampling are consumed in the same order and the same leaf is selected.
 * The node sums are formed as left + right here and as differences of running sums there: a U-turn decision can differ only
 * when a dot product is within rounding of zero.  tests/test_oracle.py runs both over >= 1e4 trajectories (divergences,
 * max-depth saturation, diagonal mass) and requires identical chains and identical leapfrog counts. */
typedef struct {
  double *r_first, *r_last, *rsum; /* [n] each */
  int stopped;                     /* a sub-trajectory made a U-turn, or a leaf diverged */
} nuts_node;
typedef struct {
  const orc_config *cfg; orc_leapfrog *lf; const double *mass; double eps, H0;
  double sub_logw; long leaf_count; double *subprop; double sum_accept; long nleaf_total;
} nuts_rec;
static nuts_node nuts_node_new(int n) { nuts_node t; t.r_first = malloc(sizeof(double) * n); t.r_last = malloc(sizeof(double) * n); t.rsum = malloc(sizeof(double) * n); t.stopped = 0; return t; }
static void nuts_node_free(nuts_node *t) { free(t->r_first); free(t->r_last); free(t->rsum); }
static nuts_node nuts_build(nuts_rec *c, int depth) {
  const int n = c->lf->n, mode = c->cfg->math_mode;
  if (depth == 0) { /* one leaf = one leapfrog step */
    nuts_node t = nuts_node_new(n);
    orc_lf_take_steps(c->lf, 1, c->eps, c->mass);
    double delta = energy(c->lf, c->lf->pqBuf, c->mass) - c->H0;
    if (isnan(delta)) delta = INFINITY;
    const double leaf_logw = -delta;
    c->sum_accept += delta <= 0.0 ? 1.0 : jm_exp(mode, -delta); c->nleaf_total++;
    const double new_logw = logaddexp_det(mode, c->sub_logw, leaf_logw);
    const double u = rng_uniform(c->lf->rng);
    if (c->leaf_count == 0 || u < jm_exp(mode, leaf_logw - new_logw)) lf_snapshot(c->lf, c->subprop);
    c->sub_logw = new_logw; c->leaf_count++;
    memcpy(t.r_first, c->lf->pqBuf, sizeof(double) * n); memcpy(t.r_last, c->lf->pqBuf, sizeof(double) * n);
    memcpy(t.rsum, c->lf->pqBuf, sizeof(double) * n);
    t.stopped = delta > 1000.0;
    return t;
  }
  nuts_node a = nuts_build(c, depth - 1);
  if (a.stopped) return a;
  nuts_node b = nuts_build(c, depth - 1);
  nuts_node t = nuts_node_new(n);
  memcpy(t.r_first, a.r_first, sizeof(double) * n); memcpy(t.r_last, b.r_last, sizeof(double) * n);
  for (int i = 0; i < n; i++) t.rsum[i] = a.rsum[i] + b.rsum[i];
  t.stopped = b.stopped || nuts_is_turning(c->lf, n, c->mass, t.r_first, t.r_last, t.rsum);
  nuts_node_free(&a); nuts_node_free(&b);
  return t;
}
static double nuts_iteration_recursive(const orc_config *cfg, double *params, orc_leapfrog *lf, double stepSize, const double *mass) {
  const int n = lf->n, mode = cfg->math_mode, sz = 2 * n + 1;
  orc_lf_start_iteration(lf, params, mass);
  nuts_rec c; c.cfg = cfg; c.lf = lf; c.mass = mass; c.H0 = energy(lf, params, mass); c.sum_accept = 0.0; c.nleaf_total = 0;
  c.subprop = malloc(sizeof(double) * sz);
  double *L = malloc(sizeof(double) * sz), *R = malloc(sizeof(double) * sz), *prop = malloc(sizeof(double) * sz), *rsum = malloc(sizeof(double) * n);
  memcpy(L, params, sizeof(double) * sz); memcpy(R, params, sizeof(double) * sz); memcpy(prop, params, sizeof(double) * sz);
  memcpy(rsum, params, sizeof(double) * n);
  double tree_logw = 0.0;
  for (int depth = 0; depth < cfg->nuts_max_depth; depth++) {
    const int going_right = rng_uniform(lf->rng) > 0.5;
    c.eps = going_right ? stepSize : -stepSize;
    lf_restore(lf, going_right ? R : L);
    c.sub_logw = -INFINITY; c.leaf_count = 0;
    nuts_node t = nuts_build(&c, depth);
    if (t.stopped) { nuts_node_free(&t); break; }
    const double u = rng_uniform(lf->rng);
    if (u < jm_exp(mode, c.sub_logw - tree_logw)) memcpy(prop, c.subprop, sizeof(double) * sz);
    tree_logw = logaddexp_det(mode, tree_logw, c.sub_logw);
    for (int i = 0; i < n; i++) rsum[i] += t.rsum[i];
    lf_snapshot(lf, going_right ? R : L);
    nuts_node_free(&t);
    if (nuts_is_turning(lf, n, mass, L, R, rsum)) break;
  }
  memcpy(params + n, prop + n, sizeof(double) * (n + 1));
  lf->iterations += 1; lf->accepted += 1;
  const double mean_acc = c.nleaf_total ? c.sum_accept / (double)c.nleaf_total : 0.0;
  lf->sumAccept += mean_acc;
  free(L); free(R); free(prop); free(rsum); free(c.subprop);
  return jm_log(mode, mean_acc);
}

/* ---- S/HMC.scala, S/EHMC.scala ----------------------------------------------------------- */
typedef struct { const orc_config *cfg; ringbuf steps; double *snap; } samplerst;

static void ehmc_countSteps(samplerst *s, double *params, orc_leapfrog *lf, double stepSize, const double *mass) { /* S/EHMC.scala:32-50 */
  int l = 0;
  while (!orc_lf_is_uturn(lf, params) && l < s->cfg->max_steps) {
    l += 1;
    orc_lf_take_steps(lf, 1, stepSize, mass);
    if (l == s->cfg->min_steps) lf_snapshot(lf, s->snap);
  }
  if (l < s->cfg->min_steps) orc_lf_take_steps(lf, s->cfg->min_steps - l, stepSize, mass);
  else lf_restore(lf, s->snap);
  ring_add(&s->steps, (double)l);
}
static double sampler_warmup(samplerst *s, double *params, orc_leapfrog *lf, double stepSize, const double *mass) {
  if (s->cfg->sampler == ORC_NUTS) return nuts_iteration(s->cfg, params, lf, stepSize, mass);
  if (s->cfg->sampler == ORC_NUTS_RECURSIVE) return nuts_iteration_recursive(s->cfg, params, lf, stepSize, mass);
  orc_lf_start_iteration(lf, params, mass);
  if (s->cfg->sampler == ORC_HMC) { /* S/HMC.scala:6-13 */
    orc_lf_take_steps(lf, s->cfg->n_steps, stepSize, mass);
  } else { /* S/EHMC.scala:15-30 */
    int shouldCount = !s->steps.full || rng_uniform(lf->rng) < s->cfg->p_count;
    if (shouldCount) ehmc_countSteps(s, params, lf, stepSize, mass);
    else { int n = (int)ring_sample(&s->steps, lf->rng); orc_lf_take_steps(lf, n, stepSize, mass); }
  }
  return orc_lf_finish_iteration(lf, params, mass);
}
static void sampler_run(samplerst *s, double *params, orc_leapfrog *lf, double stepSize, const double *mass) {
  if (s->cfg->sampler == ORC_NUTS) { (void)nuts_iteration(s->cfg, params, lf, stepSize, mass); return; }
  if (s->cfg->sampler == ORC_NUTS_RECURSIVE) { (void)nuts_iteration_recursive(s->cfg, params, lf, stepSize, mass); return; }
  orc_lf_start_iteration(lf, params, mass);
  if (s->cfg->sampler == ORC_HMC) orc_lf_take_steps(lf, s->cfg->n_steps, stepSize, mass); /* S/HMC.scala:15-23 */
  else { int n = (int)ring_sample(&s->steps, lf->rng); orc_lf_take_steps(lf, n, stepSize, mass); } /* S/EHMC.scala:52-61 */
  (void)orc_lf_finish_iteration(lf, params, mass);
}

/* ---- S/Driver.scala:7-119 ------------------------------------------------------------------ */
int orc_sample_chain(const orc_config *cfg, orc_density_fn f, void *ctx, int n, int64_t seed,
                     double *draws, double *mass_out, orc_stats *stats) {
  jrandom rng; jrandom_init(&rng, seed); /* ScalaRNG(seed) */
  return orc_sample_chain_state(cfg, f, ctx, n, &rng, draws, mass_out, stats);
}
/* the same, continuing an existing stream (e.g. the one SBC.synthesize already consumed from, incl. a cached gaussian) */
int orc_sample_chain_state(const orc_config *cfg, orc_density_fn f, void *ctx, int n, const jrandom *init,
                           double *draws, double *mass_out, orc_stats *stats) {
  jrandom rng = *init;
  orc_leapfrog *lf = orc_lf_new(f, ctx, n, &rng, cfg->math_mode);
  double *params = calloc(2 * n + 1, sizeof(double));
  double *massbuf = calloc((size_t)n * n + n, sizeof(double)); /* diag [n] or dense [n*n] */
  lf->chol_u = calloc((size_t)n * (n + 1) / 2 + 1, sizeof(double));
  const double *mass = NULL; /* IdentityMassMatrix */
  samplerst s; s.cfg = cfg; s.snap = calloc(2 * n + 1, sizeof(double));
  ring_init(&s.steps, cfg->sampler == ORC_EHMC ? cfg->buf_size : 1);

  orc_lf_initialize(lf, NULL, params); /* :22 lf.initialize(IdentityMassMatrix) */

  /* warmup :49-90 */
  /* sampler.initialize: HMC no-op, EHMC allocates snapshot buffer */
  dualavg da; double stepSize;
  if (cfg->step_tuner == ORC_STEP_DUALAVG) { /* DualAvgTuner.initialize S/DualAvg.scala:6-10 (always Identity) */
    stepSize = findReasonableStepSize(lf, params, NULL);
    da = dualavg_new(cfg->delta, stepSize, cfg->math_mode);
  } else { stepSize = cfg->static_step; memset(&da, 0, sizeof(da)); }
  masstuner mt; memset(&mt, 0, sizeof(mt));
  mt.kind = cfg->mass_tuner;
  if (cfg->mass_tuner == ORC_MASS_DENSE_WINDOWED) mt.cov = calloc((size_t)n * n, sizeof(double));
  if (cfg->mass_tuner == ORC_MASS_DIAG_WINDOWED || cfg->mass_tuner == ORC_MASS_DENSE_WINDOWED) { /* S/MassMatrix.scala:139-143 */
    mt.windowSize = cfg->init_window; mt.windowExpansion = cfg->expansion; mt.skipFirst = cfg->skip_first; mt.skipLast = cfg->skip_last;
    mt.totalIterations = cfg->warmup; varest_init(&mt.est, n);
  } else if (cfg->mass_tuner == ORC_MASS_STATIC_DIAG) { memcpy(massbuf, cfg->static_mass, sizeof(double) * n); mass = massbuf; }

  double *sample = calloc(n, sizeof(double));
  for (int i = 0; i < cfg->warmup; i++) {
    double logAcceptProb = sampler_warmup(&s, params, lf, stepSize, mass);
    if (cfg->step_tuner == ORC_STEP_DUALAVG) { dualavg_update(&da, logAcceptProb); stepSize = jm_exp(cfg->math_mode, da.logStepSize); }
    memcpy(sample, params + n, sizeof(double) * n); /* lf.variables */
    if (masstuner_update(&mt, sample, massbuf)) {
      mass = massbuf; /* DiagonalMassMatrix(variance) / DenseMassMatrix(covariance): require(!elements.contains(0.0)) S/MassMatrix.scala:8,16 */
      for (int e = 0; e < (cfg->mass_tuner == ORC_MASS_DENSE_WINDOWED ? n * n : n); e++) if (massbuf[e] == 0.0) lf->density_error |= 2;
      if (cfg->mass_tuner == ORC_MASS_DENSE_WINDOWED) { lf->dense = massbuf; orc_cholesky_upper(massbuf, n, lf->chol_u); }
      if (cfg->step_tuner == ORC_STEP_DUALAVG) { /* DualAvgTuner.reset S/DualAvg.scala:17-21 */
        double ss = jm_exp(cfg->math_mode, da.logStepSizeBar);
        da = dualavg_new(cfg->delta, ss, cfg->math_mode);
        stepSize = ss;
      }
    }
  }
  if (stats) { stats->warmup_leapfrog_steps = lf->leapfrogSteps; stats->warmup_gradient_evaluations = lf->gradientEvaluations; }
  /* lf.resetStats() :31 */
  lf->gradientEvaluations = 0; lf->leapfrogSteps = 0; lf->iterations = 0; lf->accepted = 0; lf->sumAccept = 0;
  lf->eSamples = 0; lf->eMean = 0; lf->eRaw = 0; lf->eTrans2 = 0;
  /* stepSizeTuner.stepSize :37 */
  double finalStep = cfg->step_tuner == ORC_STEP_DUALAVG ? jm_exp(cfg->math_mode, da.logStepSizeBar) : cfg->static_step;
  for (int i = 0; i < cfg->iterations; i++) { /* collectSamples :92-119 */
    sampler_run(&s, params, lf, finalStep, mass);
    memcpy(draws + (size_t)i * n, params + n, sizeof(double) * n);
  }
  for (int i = 0; i < n; i++) mass_out[i] = !mass ? 1.0 : (mass == lf->dense ? mass[i * n + i] : mass[i]);
  if (cfg->dense_out) for (int e = 0; e < n * n; e++) cfg->dense_out[e] = (mass && mass == lf->dense) ? mass[e] : ((e / n == e % n) ? mass_out[e / n] : 0.0);
  if (stats) {
    stats->gradient_evaluations = lf->gradientEvaluations; stats->leapfrog_steps = lf->leapfrogSteps;
    stats->accepted = lf->accepted; stats->mean_accept_prob = lf->iterations ? lf->sumAccept / lf->iterations : 0.0;
    stats->step_size = finalStep; stats->density_error = lf->density_error;
    stats->bfmi = lf->eTrans2 / lf->eRaw; /* Stats.bfmi; meaningful for HMC / EHMC (finishIteration), NaN when no iteration ran */
  }
  int rc = lf->density_error;
  if (cfg->mass_tuner == ORC_MASS_DIAG_WINDOWED || cfg->mass_tuner == ORC_MASS_DENSE_WINDOWED) varest_free(&mt.est);
  free(mt.cov); free(lf->chol_u); lf->chol_u = NULL;
  free(sample); free(s.snap); free(s.steps.buf); free(params); free(massbuf); orc_lf_free(lf);
  return rc;
}

/* ---- rainier-core/.../core/Trace.scala:52-120 -------------------------------------------- */
static double variogram(const double *trace, int n, int lag) { /* :111-119 */
  double sum = 0.0;
  for (int i = lag; i < n; i++) { double d = trace[i] - trace[i - lag]; sum += d * d; /* Math.pow(x,2) == x*x exactly */ }
  return sum / (double)(n - lag);
}
void orc_diagnostics(const double *traces, int mi, int ni, double *rhat, double *ess) {
  double m = mi, n = ni;
  double *means = malloc(sizeof(double) * mi);
  for (int c = 0; c < mi; c++) { double s = 0.0; for (int i = 0; i < ni; i++) s += traces[(size_t)c * ni + i]; means[c] = s / n; }
  double meanMean = 0.0; for (int c = 0; c < mi; c++) meanMean += means[c]; meanMean /= m;
  double bs = 0.0; for (int c = 0; c < mi; c++) { double d = means[c] - meanMean; bs += d * d; }
  double b = (n / (m - 1)) * bs;
  double ws = 0.0;
  for (int c = 0; c < mi; c++) { double s = 0.0; for (int i = 0; i < ni; i++) { double d = traces[(size_t)c * ni + i] - means[c]; s += d * d; } ws += s / (n - 1); }
  double w = ws / m;
  double v = (n - 1) / n * w + b / n;
  *rhat = sqrt(v / w);
  double acc = 0.0; int lag = 1; /* autocorrelation :91-109 */
  for (;;) {
    double vt = 0.0; for (int c = 0; c < mi; c++) vt += variogram(traces + (size_t)c * ni, ni, lag); vt /= m;
    double pt = 1.0 - (vt / (2.0 * v));
    if (pt > 0.0 && lag < 100) { acc += pt; lag += 1; } else break;
  }
  *ess = n * m / (1 + (2 * acc));
  free(means);
}
