/* oracle/jmath.c -- TEST INFRASTRUCTURE ONLY.  See jmath.h for provenance. */
#include "jmath.h"
#include <math.h>
#include <string.h>

static inline int32_t hi_word(double x) { uint64_t u; memcpy(&u, &x, 8); return (int32_t)(u >> 32); }
static inline uint32_t lo_word(double x) { uint64_t u; memcpy(&u, &x, 8); return (uint32_t)u; }
static inline double with_hi(double x, int32_t h) {
  uint64_t u; memcpy(&u, &x, 8);
  u = (u & 0xffffffffull) | ((uint64_t)(uint32_t)h << 32);
  memcpy(&x, &u, 8); return x;
}

/* ---- fdlibm 5.3 e_log.c ---------------------------------------------------------------- */
double jm_strict_log(double x) {
  static const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
                      two54 = 1.80143985094819840000e+16, Lg1 = 6.666666666666735130e-01,
                      Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
                      Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01,
                      Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
  const double zero = 0.0;
  double hfsq, f, s, z, R, w, t1, t2, dk;
  int32_t k, hx, i, j;
  uint32_t lx;
  hx = hi_word(x); lx = lo_word(x);
  k = 0;
  if (hx < 0x00100000) {
    if (((hx & 0x7fffffff) | lx) == 0) return -two54 / zero;
    if (hx < 0) return (x - x) / zero;
    k -= 54; x *= two54; hx = hi_word(x);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  i = (hx + 0x95f64) & 0x100000;
  x = with_hi(x, hx | (i ^ 0x3ff00000));
  k += (i >> 20);
  f = x - 1.0;
  if ((0x000fffff & (2 + hx)) < 3) {
    if (f == zero) {
      if (k == 0) return zero;
      dk = (double)k; return dk * ln2_hi + dk * ln2_lo;
    }
    R = f * f * (0.5 - 0.33333333333333333 * f);
    if (k == 0) return f - R;
    dk = (double)k; return dk * ln2_hi - ((R - dk * ln2_lo) - f);
  }
  s = f / (2.0 + f);
  dk = (double)k;
  z = s * s;
  i = hx - 0x6147a;
  w = z * z;
  j = 0x6b851 - hx;
  t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  R = t2 + t1;
  if (i > 0) {
    hfsq = 0.5 * f * f;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  }
  if (k == 0) return f - s * (f - R);
  return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

/* ---- fdlibm 5.3 e_exp.c ---------------------------------------------------------------- */
double jm_strict_exp(double x) {
  static const double one = 1.0, halF[2] = {0.5, -0.5}, huge = 1.0e+300,
                      twom1000 = 9.33263618503218878990e-302,
                      o_threshold = 7.09782712893383973096e+02,
                      u_threshold = -7.45133219101941108420e+02,
                      ln2HI[2] = {6.93147180369123816490e-01, -6.93147180369123816490e-01},
                      ln2LO[2] = {1.90821492927058770002e-10, -1.90821492927058770002e-10},
                      invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
                      P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
                      P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  double y, hi = 0.0, lo = 0.0, c, t;
  int32_t k = 0, xsb;
  uint32_t hx;
  hx = (uint32_t)hi_word(x);
  xsb = (int32_t)((hx >> 31) & 1);
  hx &= 0x7fffffff;
  if (hx >= 0x40862E42) {
    if (hx >= 0x7ff00000) {
      if (((hx & 0xfffff) | lo_word(x)) != 0) return x + x;
      return (xsb == 0) ? x : 0.0;
    }
    if (x > o_threshold) return huge * huge;
    if (x < u_threshold) return twom1000 * twom1000;
  }
  if (hx > 0x3fd62e42) {
    if (hx < 0x3FF0A2B2) {
      hi = x - ln2HI[xsb]; lo = ln2LO[xsb]; k = 1 - xsb - xsb;
    } else {
      k = (int32_t)(invln2 * x + halF[xsb]);
      t = k;
      hi = x - t * ln2HI[0];
      lo = t * ln2LO[0];
    }
    x = hi - lo;
  } else if (hx < 0x3e300000) {
    if (huge + x > one) return one + x;
  } else
    k = 0;
  t = x * x;
  c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return one - ((x * c) / (c - 2.0) - x);
  y = one - ((lo - (x * c) / (2.0 - c)) - hi);
  if (k >= -1021) return with_hi(y, hi_word(y) + (k << 20));
  y = with_hi(y, hi_word(y) + ((k + 1000) << 20));
  return y * twom1000;
}

double jm_exp(int mode, double x) { return mode == JM_DET ? jm_strict_exp(x) : exp(x); }
double jm_log(int mode, double x) { return mode == JM_DET ? jm_strict_log(x) : log(x); }
double jm_sqrt(double x) { return sqrt(x); }
double jm_pow_neg075(int mode, double t) {
  if (mode == JM_DET) { double r = sqrt(t); return 1.0 / (r * sqrt(r)); }
  return pow(t, -0.75);
}

/* ---- java.util.Random ------------------------------------------------------------------ */
#define JR_MULT 0x5DEECE66DULL
#define JR_MASK ((1ULL << 48) - 1)
void jrandom_init(jrandom *r, int64_t seed) {
  r->seed = ((uint64_t)seed ^ JR_MULT) & JR_MASK;
  r->have_next = 0;
  r->next_next = 0.0;
}
int32_t jrandom_next(jrandom *r, int bits) {
  r->seed = (r->seed * JR_MULT + 0xBULL) & JR_MASK;
  return (int32_t)(int64_t)(r->seed >> (48 - bits));
}
int32_t jrandom_next_int(jrandom *r) { return jrandom_next(r, 32); }
double jrandom_next_double(jrandom *r) {
  int64_t a = (int64_t)jrandom_next(r, 26);
  int64_t b = (int64_t)jrandom_next(r, 27);
  return (double)((a << 27) + b) * 0x1.0p-53;
}
double jrandom_next_gaussian(jrandom *r) {
  if (r->have_next) { r->have_next = 0; return r->next_next; }
  double v1, v2, s;
  do {
    v1 = 2 * jrandom_next_double(r) - 1;
    v2 = 2 * jrandom_next_double(r) - 1;
    s = v1 * v1 + v2 * v2;
  } while (s >= 1 || s == 0);
  double multiplier = sqrt(-2 * jm_strict_log(s) / s);
  r->next_next = v2 * multiplier;
  r->have_next = 1;
  return v1 * multiplier;
}
