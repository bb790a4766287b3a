// optimize.hpp -- host side of rh_optimize: the reference's L-BFGS driven over the batched device density.
#pragma once
#include <cstdint>
#include <functional>
#include <string>

namespace rh {
// q [starts][n] -> logp [starts], grad [starts][n]; returns an rh_status
using BatchDensity = std::function<int(const double *q, int starts, double *logp, double *grad)>;

// Optimizer.lbfgs (optimizer/Optimizer.scala:6-24) for `starts` independent starting points in lockstep: every round is
// ONE batched density launch for all starts still searching.  status: 0 converged, 1 "dginit" (the reference throws),
// 2 max_evals reached.  Returns an rh_status (density failures only).
int lbfgs_multistart(int n, int starts, const double *x0, int max_evals, const BatchDensity &density, double *x_out,
                     int32_t *evals_out, int32_t *status_out, std::string &err);
} // namespace rh
