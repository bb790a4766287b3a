// tests/stubs/host_target_shim.hpp -- TEST INFRASTRUCTURE: lets the per-model code the emitter generates (the rh_target<t> structs:
// invariants / row / finish, plain scalar C++) be compiled with the HOST compiler and evaluated row by row the way
// ir/DataFunction.scala:32-84 does, so that the CPU suite can check the EMITTER's algebra (output factoring, linear-combination
// outputs, scatter families, invariant tables, closed-form links) against the oracle without a GPU.  The product path never
// includes this file; the functions below stand in for the few device-library entry points generated row code may call
// (rainier_amd/csrc/device/rh_prelude.hip.h), with libm in place of the device math.
#pragma once
#include <cmath>
#include <cstring>
// the generated functions' accumulator / invariant element type
#ifndef RH_HEAVY
#define RH_HEAVY 0
#endif
typedef double rh_acc_t;
static inline int rh_oz() { return 0; }   // device: a volatile asm that yields 0 (an index the compiler cannot see through)
#define RH_DEV inline
#define RH_DEV_NOINLINE
#define RH_NAN (__builtin_nan(""))
#define RH_INF (__builtin_inf())
static inline double rh_one() { return 1.0; }
static inline int rh_d2i(double x) { if (x != x) return 0; if (x >= 2147483647.0) return 2147483647; if (x <= -2147483648.0) return (-2147483647 - 1); return (int)x; }
static inline double rh_compare(double a, double b) { return a > b ? 1.0 : (a == b ? 0.0 : -1.0); }
static inline double rh_pow_half(double x) { return x == -RH_INF ? RH_INF : std::sqrt(x + 0.0); }
static inline double rh_java_pow(double x, double y) { if (y == 0.0) return 1.0; if (y != y) return RH_NAN; if (std::isinf(y) && std::fabs(x) == 1.0) return RH_NAN; return std::pow(x, y); }
static inline double rh_fast_log(double x) { return std::log(x); }
static inline double rh_strict_log(double x) { return std::log(x); }
static inline double rh_strict_exp(double x) { return std::exp(x); }
static inline void rh_logit_link(double t, double &softplus, double &sigmoid) {
  const double u = std::exp(-std::fabs(t));
  sigmoid = t >= 0 ? 1.0 / (1.0 + u) : u / (1.0 + u);
  softplus = (t <= 0.0 ? 0.0 : t) + std::log1p(u);
}
using std::exp; using std::log; using std::sin; using std::cos; using std::tan; using std::asin; using std::acos; using std::atan; using std::pow;
