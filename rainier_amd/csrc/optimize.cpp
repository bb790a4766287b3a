// optimize.cpp -- Optimizer.lbfgs / LBFGS.apply / mcsrch / mcstep of the reference
// (rainier-sampler/src/main/scala/com/stripe/rainier/optimizer/Optimizer.scala:6-24, LBFGS.java:44-632) as two small
// host classes.  The vector work is O(m n) per evaluation and stays on the host; the evaluations themselves -- the only
// part that touches the observations -- are batched device launches (rh_density_eval's kernels), one per round for all
// starting points.  Arithmetic follows the reference expression by expression (no contraction: see the Makefile flags),
// so that with a bit-exact density the iterates are those of the JVM.
#include "optimize.hpp"
#include "../../include/rainier_hip.h"
#include <cmath>
#include <cstring>
#include <vector>

namespace rh {
namespace {

// java.lang.Math.min / max (NaN propagates, -0.0 < +0.0)
inline double jmin(double a, double b) { return a != a ? a : (a == 0.0 && b == 0.0) ? (std::signbit(a) ? a : b) : (a <= b ? a : b); }
inline double jmax(double a, double b) { return a != a ? a : (a == 0.0 && b == 0.0) ? (std::signbit(a) ? b : a) : (a >= b ? a : b); }
inline double largest(double a, double b, double c) { return a < b ? (b < c ? c : b) : (a < c ? c : a); } // LBFGS.java:753

// LBFGS.ddot / daxpy (LBFGS.java:640-731): ascending sums, no fusion; daxpy is a no-op for a zero factor
inline double dot(const double *a, const double *b, int n) {
  double acc = 0;
  for (int i = 0; i < n; ++i) acc = acc + a[i] * b[i];
  return acc;
}
inline void axpy(double f, const double *x, double *y, int n) {
  if (f == 0) return;
  for (int i = 0; i < n; ++i) y[i] = y[i] + f * x[i];
}

// More'-Thuente line search along `dir` from `origin` (LBFGS.java:183-632).  begin() -> trial(); then absorb(f, dphi)
// after every evaluation until it reports done.
struct LineSearch {
  static constexpr double kFtol = 0.0001, kGtol = 0.9, kXtol = 1e-16, kStepMin = 1e-20, kStepMax = 1e20;
  static constexpr int kMaxEvals = 20;
  double step = 0;                       // stp
  int evals = 0;                         // nfev
  int last_case = 1;                     // infoc
  bool bracketed = false, first_stage = true;
  double phi0 = 0, dphi0 = 0, dec = 0;   // finit, dginit, dgtest = ftol * dginit
  double lo = 0, hi = 0, width = 0, width_prev = 0; // stmin, stmax, width, width1
  double best = 0, f_best = 0, d_best = 0;          // stx, fx, dgx
  double other = 0, f_other = 0, d_other = 0;       // sty, fy, dgy

  bool begin(double f, double slope) { // LBFGS.java:225-268; false = not a descent direction ("dginit")
    last_case = 1;
    dphi0 = slope;
    if (dphi0 >= 0) return false;
    bracketed = false; first_stage = true; evals = 0;
    phi0 = f;
    dec = kFtol * dphi0;
    width = kStepMax - kStepMin;
    width_prev = width / 0.5;
    best = 0; f_best = phi0; d_best = dphi0;
    other = 0; f_other = phi0; d_other = dphi0;
    return true;
  }

  double trial() { // LBFGS.java:272-300: the step the next evaluation is taken at
    if (bracketed) { lo = jmin(best, other); hi = jmax(best, other); }
    else { lo = best; hi = step + 4 * (step - best); }
    step = jmax(step, kStepMin);
    step = jmin(step, kStepMax);
    if ((bracketed && (step <= lo || step >= hi)) || evals >= kMaxEvals - 1 || last_case == 0 ||
        (bracketed && hi - lo <= kXtol * hi))
      step = best;
    return step;
  }

  bool absorb(double f, double slope) { // LBFGS.java:310-396: true = line search finished at `step`
    evals += 1;
    const double bar = phi0 + step * dec; // ftest1
    int verdict = 0;
    if ((bracketed && (step <= lo || step >= hi)) || last_case == 0) verdict = 6;
    if (step == kStepMax && f <= bar && slope <= dec) verdict = 5;
    if (step == kStepMin && (f > bar || slope >= dec)) verdict = 4;
    if (evals >= kMaxEvals) verdict = 3;
    if (bracketed && hi - lo <= kXtol * hi) verdict = 2;
    if (f <= bar && std::fabs(slope) <= kGtol * (-dphi0)) verdict = 1;
    if (verdict != 0) return true;
    if (first_stage && f <= bar && slope >= jmin(kFtol, kGtol) * dphi0) first_stage = false;
    if (first_stage && f <= f_best && f > bar) { // interpolate on the modified function
      double fm = f - step * dec, fbm = f_best - best * dec, fom = f_other - other * dec;
      double dm = slope - dec, dbm = d_best - dec, dom = d_other - dec;
      interpolate(fbm, dbm, fom, dom, fm, dm);
      f_best = fbm + best * dec;
      f_other = fom + other * dec;
      d_best = dbm + dec;
      d_other = dom + dec;
    } else {
      interpolate(f_best, d_best, f_other, d_other, f, slope);
    }
    if (bracketed) {
      if (std::fabs(other - best) >= 0.66 * width_prev) step = best + 0.5 * (other - best);
      width_prev = width;
      width = std::fabs(other - best);
    }
    return false;
  }

  // mcstep (LBFGS.java:446-632): safeguarded cubic / quadratic step and interval update
  void interpolate(double &fb, double &db, double &fo, double &dob, double fp, double dp) {
    const double t = step, b = best, o = other;
    last_case = 0;
    if ((bracketed && (t <= jmin(b, o) || t >= jmax(b, o))) || db * (t - b) >= 0.0 || hi < lo) return;
    const double sgnd = dp * (db / std::fabs(db));
    bool clamp;
    double next, theta, s, gamma, p, q, r, cubic, quad;
    if (fp > fb) {
      last_case = 1; clamp = true;
      theta = 3 * (fb - fp) / (t - b) + db + dp;
      s = largest(std::fabs(theta), std::fabs(db), std::fabs(dp));
      gamma = s * std::sqrt((theta / s) * (theta / s) - (db / s) * (dp / s));
      if (t < b) gamma = -gamma;
      p = (gamma - db) + theta;
      q = ((gamma - db) + gamma) + dp;
      r = p / q;
      cubic = b + r * (t - b);
      quad = b + ((db / ((fb - fp) / (t - b) + db)) / 2) * (t - b);
      next = std::fabs(cubic - b) < std::fabs(quad - b) ? cubic : cubic + (quad - cubic) / 2;
      bracketed = true;
    } else if (sgnd < 0.0) {
      last_case = 2; clamp = false;
      theta = 3 * (fb - fp) / (t - b) + db + dp;
      s = largest(std::fabs(theta), std::fabs(db), std::fabs(dp));
      gamma = s * std::sqrt((theta / s) * (theta / s) - (db / s) * (dp / s));
      if (t > b) gamma = -gamma;
      p = (gamma - dp) + theta;
      q = ((gamma - dp) + gamma) + db;
      r = p / q;
      cubic = t + r * (b - t);
      quad = t + (dp / (dp - db)) * (b - t);
      next = std::fabs(cubic - t) > std::fabs(quad - t) ? cubic : quad;
      bracketed = true;
    } else if (std::fabs(dp) < std::fabs(db)) {
      last_case = 3; clamp = true;
      theta = 3 * (fb - fp) / (t - b) + db + dp;
      s = largest(std::fabs(theta), std::fabs(db), std::fabs(dp));
      gamma = s * std::sqrt(jmax(0, (theta / s) * (theta / s) - (db / s) * (dp / s)));
      if (t > b) gamma = -gamma;
      p = (gamma - dp) + theta;
      q = (gamma + (db - dp)) + gamma;
      r = p / q;
      if (r < 0.0 && gamma != 0.0) cubic = t + r * (b - t);
      else cubic = t > b ? hi : lo;
      quad = t + (dp / (dp - db)) * (b - t);
      if (bracketed) next = std::fabs(t - cubic) < std::fabs(t - quad) ? cubic : quad;
      else next = std::fabs(t - cubic) > std::fabs(t - quad) ? cubic : quad;
    } else {
      last_case = 4; clamp = false;
      if (bracketed) {
        theta = 3 * (fp - fo) / (o - t) + dob + dp;
        s = largest(std::fabs(theta), std::fabs(dob), std::fabs(dp));
        gamma = s * std::sqrt((theta / s) * (theta / s) - (dob / s) * (dp / s));
        if (t > o) gamma = -gamma;
        p = (gamma - dp) + theta;
        q = ((gamma - dp) + gamma) + dob;
        r = p / q;
        next = t + r * (o - t);
      } else next = t > b ? hi : lo;
    }
    if (fp > fb) { other = t; fo = fp; dob = dp; }
    else {
      if (sgnd < 0.0) { other = b; fo = fb; dob = db; }
      best = t; fb = fp; db = dp;
    }
    next = jmin(hi, next);
    next = jmax(lo, next);
    step = next;
    if (bracketed && clamp) {
      if (other > best) step = jmin(best + 0.66 * (other - best), step);
      else step = jmax(best + 0.66 * (other - best), step);
    }
  }
};

// LBFGS (LBFGS.java:44-180) with m corrections; x is owned by the caller (a row of the batched q array)
class Lbfgs {
 public:
  Lbfgs(double *x, int n, int m, double eps)
      : x_(x), n_(n), m_(m), eps_(eps), s_((size_t)m * n), y_((size_t)m * n), rho_(m), alpha_(m), work_(n), origin_(n) {}

  enum Verdict { kEvaluate = 0, kConverged = 1, kNotDescent = -1 };

  Verdict apply(double f, const double *g) {
    if (iter_ == 0) {
      for (int i = 0; i < n_; ++i) s_[i] = -g[i] * 1.0; // -g * diag, diag = 1 (LBFGS.java:71-74)
      first_step_ = 1 / std::sqrt(dot(g, g, n_));
    } else {
      double *dir = &s_[(size_t)slot_ * n_];
      if (!ls_.absorb(f, directional(g, dir))) { move_to(ls_.trial()); return kEvaluate; }
      double *yd = &y_[(size_t)slot_ * n_];
      for (int i = 0; i < n_; ++i) { dir[i] = ls_.step * dir[i]; yd[i] = g[i] - work_[i]; }
      newest_ = slot_;
      slot_ = slot_ + 1 == m_ ? 0 : slot_ + 1;
      double gnorm = std::sqrt(dot(g, g, n_)), xnorm = jmax(1.0, std::sqrt(dot(x_, x_, n_)));
      if (gnorm / xnorm <= eps_) return kConverged;
    }
    direction(g);
    const double *dir = &s_[(size_t)slot_ * n_];
    if (!ls_.begin(f, directional(g, dir))) return kNotDescent;
    std::memcpy(origin_.data(), x_, sizeof(double) * n_);
    move_to(ls_.trial());
    return kEvaluate;
  }

 private:
  static double directional(const double *g, const double *dir, int n) {
    double d = 0;
    for (int j = 0; j < n; ++j) d = d + g[j] * dir[j];
    return d;
  }
  double directional(const double *g, const double *dir) const { return directional(g, dir, n_); }
  void move_to(double step) {
    const double *dir = &s_[(size_t)slot_ * n_];
    for (int j = 0; j < n_; ++j) x_[j] = origin_[j] + step * dir[j];
  }
  // two-loop recursion (LBFGS.java:82-157); leaves the new direction in s_[slot_] and g in work_
  void direction(const double *g) {
    iter_ += 1;
    if (iter_ != 1) {
      const int bound = iter_ > m_ ? m_ : iter_ - 1;
      const double *yn = &y_[(size_t)newest_ * n_], *sn = &s_[(size_t)newest_ * n_];
      const double ys = dot(yn, sn, n_), yy = dot(yn, yn, n_);
      const double scale = ys / yy;
      rho_[(slot_ == 0 ? m_ : slot_) - 1] = 1 / ys;
      for (int i = 0; i < n_; ++i) work_[i] = -g[i];
      int cp = slot_;
      for (int k = 0; k < bound; ++k) {
        cp = cp == 0 ? m_ - 1 : cp - 1;
        alpha_[cp] = rho_[cp] * dot(&s_[(size_t)cp * n_], work_.data(), n_);
        axpy(-alpha_[cp], &y_[(size_t)cp * n_], work_.data(), n_);
      }
      for (int i = 0; i < n_; ++i) work_[i] = scale * work_[i];
      for (int k = 0; k < bound; ++k) {
        double beta = rho_[cp] * dot(&y_[(size_t)cp * n_], work_.data(), n_);
        beta = alpha_[cp] - beta;
        axpy(beta, &s_[(size_t)cp * n_], work_.data(), n_);
        cp = cp + 1 == m_ ? 0 : cp + 1;
      }
      std::memcpy(&s_[(size_t)slot_ * n_], work_.data(), sizeof(double) * n_);
    }
    ls_.evals = 0;
    ls_.step = iter_ == 1 ? first_step_ : 1;
    std::memcpy(work_.data(), g, sizeof(double) * n_);
  }

  double *x_;
  int n_, m_;
  double eps_;
  std::vector<double> s_, y_, rho_, alpha_, work_, origin_;
  int iter_ = 0, slot_ = 0, newest_ = 0;
  double first_step_ = 0;
  LineSearch ls_;
};

} // namespace

int lbfgs_multistart(int n, int starts, const double *x0, int max_evals, const BatchDensity &density, double *x_out,
                     int32_t *evals_out, int32_t *status_out, std::string &err) {
  // active starts are kept compacted in q so that each round is one dense batched launch
  std::vector<double> q((size_t)starts * n), logp(starts), grad((size_t)starts * n), g(n);
  for (size_t i = 0; i < q.size(); ++i) q[i] = x0 ? x0[i] : 0.0;
  std::vector<Lbfgs> opt;
  opt.reserve(starts);
  std::vector<int> evals(starts, 0);
  for (int s = 0; s < starts; ++s) opt.emplace_back(&q[(size_t)s * n], n, 5, 0.1); // m = 5, eps = 0.1 (Optimizer.scala:11-13)
  // each optimiser writes its iterate into its own row of q; `live` lists the rows still searching
  std::vector<int> live(starts);
  for (int s = 0; s < starts; ++s) live[s] = s;
  std::vector<double> qa, la, ga;
  auto retire = [&](int s, int status) {
    std::memcpy(x_out + (size_t)s * n, &q[(size_t)s * n], sizeof(double) * n);
    if (evals_out) evals_out[s] = evals[s];
    if (status_out) status_out[s] = status;
  };
  while (!live.empty()) {
    const int k = (int)live.size();
    qa.resize((size_t)k * n); la.resize(k); ga.resize((size_t)k * n);
    for (int a = 0; a < k; ++a) std::memcpy(&qa[(size_t)a * n], &q[(size_t)live[a] * n], sizeof(double) * n);
    int rc = density(qa.data(), k, la.data(), ga.data());
    if (rc != RH_OK) { err = "rh_optimize: density evaluation failed"; return rc; }
    std::vector<int> next;
    next.reserve(k);
    for (int a = 0; a < k; ++a) {
      const int s = live[a];
      evals[s] += 1;
      for (int i = 0; i < n; ++i) g[i] = ga[(size_t)a * n + i] * -1; // minimise -density (Optimizer.scala:16-21)
      Lbfgs::Verdict v = opt[s].apply(la[a] * -1, g.data());
      if (v == Lbfgs::kConverged) retire(s, RH_OPT_CONVERGED);
      else if (v == Lbfgs::kNotDescent) retire(s, RH_OPT_NOT_DESCENT);
      else if (max_evals > 0 && evals[s] >= max_evals) retire(s, RH_OPT_MAX_EVALS);
      else next.push_back(s);
    }
    live.swap(next);
  }
  return RH_OK;
}
} // namespace rh
