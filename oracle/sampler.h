/* oracle/sampler.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of rainier-sampler for one chain.  Every function cites the
 * reference lines it follows; S/ = rainier-sampler/src/main/scala/com/stripe/rainier/sampler/.
 * The reference cannot run here (no JVM); pins are listed in oracle/README.md.
 */
#ifndef ORACLE_SAMPLER_H
#define ORACLE_SAMPLER_H
#include "jmath.h"
#include <stdint.h>

/* S/DensityFunction.scala:3-8 folded into one call: out[0] = density, out[1+i] = gradient(i) */
typedef int (*orc_density_fn)(void *ctx, const double *q, double *out);

enum { ORC_HMC = 0, ORC_EHMC = 1, ORC_NUTS = 2,
       ORC_NUTS_RECURSIVE = 3 /* the same transition built recursively, without checkpoints: cross-validation only */ };
enum { ORC_STEP_DUALAVG = 0, ORC_STEP_STATIC = 1 };
enum { ORC_MASS_IDENTITY = 0, ORC_MASS_DIAG_WINDOWED = 1, ORC_MASS_STATIC_DIAG = 2, ORC_MASS_DENSE_WINDOWED = 3 };

typedef struct {
  int sampler;         /* ORC_HMC | ORC_EHMC */
  int n_steps;         /* HMCSampler(nSteps)                       S/HMC.scala:3 */
  int max_steps, min_steps, buf_size; /* EHMCSampler(maxSteps,minSteps=1,bufSize=100,pCount=0.1) S/EHMC.scala:3-6 */
  double p_count;
  int step_tuner;      /* DualAvgTuner(delta) | StaticStepSize(eps) S/DualAvg.scala:3, S/Sampler.scala:36-40 */
  double delta, static_step;
  int mass_tuner;      /* IdentityMassMatrixTuner | DiagonalMassMatrixTuner(50,1.5,50,50) | StaticMassMatrix(Diagonal) */
  int init_window; double expansion; int skip_first, skip_last;
  const double *static_mass; /* [nvars] DiagonalMassMatrix.elements */
  int nuts_max_depth;        /* ORC_NUTS (extension; NOT in the reference -- see sampler.c) */
  int iterations, warmup;    /* SamplerConfig   S/Sampler.scala:3-11 */
  int math_mode;             /* JM_LIBM | JM_DET (oracle/jmath.h) */
  double *dense_out;         /* optional [nvars*nvars]: final DenseMassMatrix.elements (DenseMassMatrixTuner) */
} orc_config;

typedef struct {
  int64_t gradient_evaluations; /* Stats.gradientEvaluations, sampling phase (reference counts 2L+1 per trajectory) */
  int64_t leapfrog_steps;       /* distinct (p,q) updates, sampling phase */
  int64_t warmup_leapfrog_steps;
  int64_t warmup_gradient_evaluations;
  int64_t accepted;             /* sampling phase */
  double mean_accept_prob;      /* sampling phase mean of exp(a) */
  double step_size;             /* step size used for sampling */
  int density_error;
  double bfmi;                  /* Stats.bfmi = energyTransitions2 / energyVariance.raw(0), sampling phase (S/Stats.scala:14-16) */
} orc_stats;

/* Driver.sample for one chain (S/Driver.scala:7-46) with rng = ScalaRNG(seed).
 * draws: [iterations][nvars]; mass_out: [nvars] final DiagonalMassMatrix.elements (1.0 for identity). */
int orc_sample_chain(const orc_config *cfg, orc_density_fn f, void *ctx, int nvars, int64_t seed,
                     double *draws, double *mass_out, orc_stats *stats);

int orc_sample_chain_state(const orc_config *cfg, orc_density_fn f, void *ctx, int nvars, const jrandom *init,
                           double *draws, double *mass_out, orc_stats *stats);

/* ---- exposed pieces (unit tests) ------------------------------------------------------- */
typedef struct orc_leapfrog orc_leapfrog;
orc_leapfrog *orc_lf_new(orc_density_fn f, void *ctx, int nvars, jrandom *rng, int math_mode);
void orc_lf_free(orc_leapfrog *lf);
/* mass == NULL: IdentityMassMatrix; else DiagonalMassMatrix(elements=mass) */
void orc_lf_initialize(orc_leapfrog *lf, const double *mass, double *params);
void orc_lf_start_iteration(orc_leapfrog *lf, double *params, const double *mass);
void orc_lf_take_steps(orc_leapfrog *lf, int l, double step, const double *mass);
double orc_lf_finish_iteration(orc_leapfrog *lf, double *params, const double *mass);
double orc_lf_try_stepping(orc_leapfrog *lf, const double *params, double step, const double *mass);
int orc_lf_is_uturn(orc_leapfrog *lf, const double *params);

/* DenseMassMatrix.choleskyUpperTriangular / upperTriangularSolve / squareMultiply (S/MassMatrix.scala:33-117), exposed for tests */
void orc_cholesky_upper(const double *matrix, int n, double *upper_packed);
void orc_upper_triangular_solve(const double *packed, const double *vector, int n, double *out);
void orc_square_multiply(const double *matrix, const double *vector, int n, double *out);

/* Trace.diagnostics (rainier-core/.../core/Trace.scala:52-120): traces [m chains][n draws] of one parameter */
void orc_diagnostics(const double *traces, int m, int n, double *rhat, double *ess);
#endif
