/* oracle/lbfgs.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's L-BFGS (`Optimizer.lbfgs`, `LBFGS.apply/mcsrch/mcstep`):
 *   O/ = rainier-sampler/src/main/scala/com/stripe/rainier/optimizer/
 *   O/Optimizer.scala:6-24   driver: x = 0, m = 5, eps = 0.1, minimise f = -density, g = -gradient
 *   O/LBFGS.java:44-180      two-loop recursion, reverse communication
 *   O/LBFGS.java:222-397     More'-Thuente line search (ftol 1e-4, gtol 0.9, xtol 1e-16, maxfev 20)
 *   O/LBFGS.java:446-632     safeguarded cubic/quadratic step
 * The reference keeps the line-search start point in `diag` and re-enters through `info == -1`; here those are an explicit
 * `xsave` vector and a phase flag, and the scalar ys/yy replaces the (uniform) `diag` vector -- every floating-point
 * expression is kept operation for operation.  Parity pin: the reference's own test (OptimizerTest.scala) compares its
 * LBFGS with RISO's original on "fit normal"; there is no numeric golden vector, so the pins here are the closed-form
 * optimum of that model and the termination criterion (tests/test_oracle.py) -- bit-level parity is UNPINNED.
 */
#ifndef ORACLE_LBFGS_H
#define ORACLE_LBFGS_H
#include "sampler.h"

typedef struct orc_lbfgs orc_lbfgs;

/* new LBFGS(x, m, eps)  (O/LBFGS.java:44-62); x is the caller's array, updated in place */
orc_lbfgs *orc_lbfgs_new(double *x, int n, int m, double eps);
void orc_lbfgs_free(orc_lbfgs *s);
/* LBFGS.apply(f, g): 1 = converged, 0 = evaluate f,g at the updated x and call again, -1 = "dginit" exception */
int orc_lbfgs_apply(orc_lbfgs *s, double f, const double *g);

/* Optimizer.lbfgs(df) (O/Optimizer.scala:6-24) from x = x0 (NULL: zeros, the reference's start).
 * Returns the number of density evaluations, or -1 on the "dginit" exception, -2 when max_evals ran out, -3 density error. */
int orc_optimize(orc_density_fn f, void *ctx, int nvars, const double *x0, int max_evals, double *x_out);
#endif
