// rh_prelude.hip.h -- hand-written gfx950 device library, part 1 (wave64, one chain per wavefront).
//
//   * strict fp64 helpers: fdlibm log/exp (java.lang.StrictMath), Java pow/compare/d2i semantics
//   * device java.util.Random (bit-exact LCG + polar nextGaussian)  -- replaces sampler/RNG.scala:20-26
//   * wvec: a length-RH_NVARS vector distributed over the 64 lanes of the chain's wavefront
//
// Everything here is compiled with FP contraction OFF: the JVM never fuses a*b+c, and the sampler
// arithmetic (LeapFrog.scala:144-173 `x(i) += s * y(i)`) must round the product before the add.
#pragma clang fp contract(off)

typedef unsigned long long rh_u64;
typedef long long rh_i64;
#define RH_DEV __device__ __forceinline__
#define RH_INF (__builtin_huge_val())
#define RH_NAN (__builtin_nan(""))

RH_DEV int rh_hi(double x) { return __double2hiint(x); }
RH_DEV unsigned rh_lo(double x) { return (unsigned)__double2loint(x); }
RH_DEV double rh_with_hi(double x, int h) { return __hiloint2double(h, __double2loint(x)); }

// ---- fdlibm 5.3 e_log.c (== java.lang.StrictMath.log) --------------------------------------------
RH_DEV double rh_strict_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
               two54 = 1.80143985094819840000e+16, Lg1 = 6.666666666666735130e-01,
               Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01,
               Lg6 = 1.531383769920937332e-01, Lg7 = 1.479819860511658591e-01;
  int hx = rh_hi(x);
  unsigned lx = rh_lo(x);
  int k = 0;
  if (hx < 0x00100000) {
    if (((hx & 0x7fffffff) | lx) == 0) return -RH_INF;
    if (hx < 0) return RH_NAN;
    k -= 54; x *= two54; hx = rh_hi(x);
  }
  if (hx >= 0x7ff00000) return x + x;
  k += (hx >> 20) - 1023;
  hx &= 0x000fffff;
  int i = (hx + 0x95f64) & 0x100000;
  x = rh_with_hi(x, hx | (i ^ 0x3ff00000));
  k += (i >> 20);
  double f = x - 1.0;
  double dk = (double)k;
  if ((0x000fffff & (2 + hx)) < 3) {
    if (f == 0.0) { if (k == 0) return 0.0; return dk * ln2_hi + dk * ln2_lo; }
    double R = f * f * (0.5 - 0.33333333333333333 * f);
    if (k == 0) return f - R;
    return dk * ln2_hi - ((R - dk * ln2_lo) - f);
  }
  double s = f / (2.0 + f);
  double z = s * s;
  i = hx - 0x6147a;
  double w = z * z;
  int j = 0x6b851 - hx;
  double t1 = w * (Lg2 + w * (Lg4 + w * Lg6));
  double t2 = z * (Lg1 + w * (Lg3 + w * (Lg5 + w * Lg7)));
  i |= j;
  double R = t2 + t1;
  if (i > 0) {
    double hfsq = 0.5 * f * f;
    if (k == 0) return f - (hfsq - s * (hfsq + R));
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
  }
  if (k == 0) return f - s * (f - R);
  return dk * ln2_hi - ((s * (f - R) - dk * ln2_lo) - f);
}

// 1.0 the optimiser cannot see through: fma(x, rh_one(), y) stays a v_fma_f64 (it is x + y exactly) instead of folding back
// into v_add_f64.  Used by the emitter's opt-in fma_adds spelling (RH_FMA_ADDS=1); cycle-accurate measurements
// (tools/ubench/fma64_cycles.hip, profiles/r1_d_fp64_ceiling) show fma/add mixes issue as fast as pure FMA streams.
RH_DEV double rh_one() {
  double o = 1.0;
  asm("" : "+v"(o));
  return o;
}

// ---- fast-mode log: fdlibm's e_log.c algorithm on the hardware frexp / rcp instructions --------------------------------
// java.lang.Math.log is only specified to 1 ulp, which this keeps (tests: <= 1 ulp against a correctly rounded log over
// 2^20 arguments incl. subnormals); ~40 VALU instructions instead of the 97 of the device library's log(), which dominated
// the per-row cost of the Bernoulli / NegativeBinomial likelihoods (cfg 4, cfg 5).  Strict mode keeps rh_strict_log.
RH_DEV double rh_fast_log(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01, Lg3 = 2.857142874366239149e-01,
               Lg4 = 2.222219843214978396e-01, Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
               Lg7 = 1.479819860511658591e-01;
  int e = __builtin_amdgcn_frexp_exp(x);            // x = m 2^e, m in [0.5, 1), subnormals included
  double m = __builtin_amdgcn_frexp_mant(x);
  const int lo = m < 0.70710678118654752440 ? 1 : 0;  // -> m in [sqrt(1/2), sqrt(2))
  m = __builtin_amdgcn_ldexp(m, lo);
  e -= lo;
  const double f = m - 1.0;
  const double t = 2.0 + f;
  double r = __builtin_amdgcn_rcp(t);
  r = __builtin_fma(__builtin_fma(-t, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-t, r, 1.0), r, r);
  double s = f * r;
  s = __builtin_fma(__builtin_fma(-t, s, f), r, s);  // s = f / (2 + f) to working precision
  const double z = s * s, w = z * z;
  const double t1 = w * __builtin_fma(w, __builtin_fma(w, Lg6, Lg4), Lg2);
  const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, Lg7, Lg5), Lg3), Lg1);
  const double R = t2 + t1;
  const double hfsq = 0.5 * f * f;
  const double dk = (double)e;
  double res = dk * ln2_hi - ((hfsq - __builtin_fma(s, hfsq + R, dk * ln2_lo)) - f);
  res = (x == __builtin_inf()) ? x : res;
  return (x > 0.0) ? res : ((x == 0.0) ? -__builtin_inf() : RH_NAN);
}

// ---- fdlibm 5.3 e_exp.c (== java.lang.StrictMath.exp) --------------------------------------------
RH_DEV double rh_strict_exp(double x) {
  const double huge = 1.0e+300, twom1000 = 9.33263618503218878990e-302,
               o_threshold = 7.09782712893383973096e+02, u_threshold = -7.45133219101941108420e+02,
               ln2HI = 6.93147180369123816490e-01, ln2LO = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00, P1 = 1.66666666666666019037e-01,
               P2 = -2.77777777770155933842e-03, P3 = 6.61375632143793436117e-05,
               P4 = -1.65339022054652515390e-06, P5 = 4.13813679705723846039e-08;
  double hi = 0.0, lo = 0.0;
  int k = 0;
  unsigned hx = (unsigned)rh_hi(x);
  const int xsb = (int)((hx >> 31) & 1);
  hx &= 0x7fffffff;
  if (hx >= 0x40862E42) {
    if (hx >= 0x7ff00000) {
      if (((hx & 0xfffff) | rh_lo(x)) != 0) return x + x;
      return (xsb == 0) ? x : 0.0;
    }
    if (x > o_threshold) return huge * huge;
    if (x < u_threshold) return twom1000 * twom1000;
  }
  if (hx > 0x3fd62e42) {
    if (hx < 0x3FF0A2B2) {
      hi = x - (xsb ? -ln2HI : ln2HI); lo = xsb ? -ln2LO : ln2LO; k = 1 - xsb - xsb;
    } else {
      k = (int)(invln2 * x + (xsb ? -0.5 : 0.5));
      const double t = (double)k;
      hi = x - t * ln2HI;
      lo = t * ln2LO;
    }
    x = hi - lo;
  } else if (hx < 0x3e300000) {
    return 1.0 + x;
  }
  const double t = x * x;
  const double c = x - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  if (k == 0) return 1.0 - ((x * c) / (c - 2.0) - x);
  double y = 1.0 - ((lo - (x * c) / (2.0 - c)) - hi);
  if (k >= -1021) return rh_with_hi(y, rh_hi(y) + (k << 20));
  y = rh_with_hi(y, rh_hi(y) + ((k + 1000) << 20));
  return y * twom1000;
}

// ---- fast mode: softplus(t) = log(1 + e^t) and sigmoid(t) = 1 / (1 + e^-t) from ONE exponential ----------------------------
// (the closed form of a Bernoulli-logit likelihood term and of its adjoint, see emit.cpp detect_logit).
//   u = e^{-|t|} in (0, 1]   (never overflows; underflows to 0 for |t| > 745, where softplus = max(t, 0) and sigmoid = 0 / 1 exactly)
//   d = 1 + u in (1, 2],  r = 1 / d
//   sigmoid(t)  = t >= 0 ? r : u r
//   softplus(t) = max(t, 0) + log1p(u),  log1p by fdlibm's log kernel on f = u (or (1 + u)/2 - 1 and + ln 2 when 1 + u > sqrt 2):
//                 z = f / (2 + f), log(1 + f) = f - f^2/2 + z (f^2/2 + R(z^2))   (e_log.c, the Lg1..Lg7 minimax polynomial)
// Both results are within 3 ulp of the exact values over the whole range (host model of this code: 2.1 / 2.7 ulp;
// tests/test_gpu_parity.py asserts it on the device); the two divisions are v_rcp_f64 + two Newton steps (operands in [1, 3]:
// no scaling, no special cases).
RH_DEV double rh_rcp_1to4(double d) {          // 1/d for d in [1, 4]: reciprocal estimate + 2 Newton-Raphson steps
  double r = __builtin_amdgcn_rcp(d);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
  return r;
}
RH_DEV void rh_logit_link(const double t, double &softplus, double &sigmoid) {
  const double at = __builtin_fabs(t);
  // e^{-|t|}: k = round(x / ln 2), r = x - k ln 2 in [-0.347, 0.347], degree-13 Taylor polynomial, scale by 2^k.
  // x = max(-|t|, -800): e^{-800} is exactly 0 in fp64 (v_ldexp_f64 underflows gradually and correctly down to it)
  const double x = __builtin_fmax(-at, -800.0);
  const double kf = __builtin_rint(x * 0x1.71547652b82fep+0);
  double r = __builtin_fma(kf, -0x1.62e42fee00000p-1, x);
  r = __builtin_fma(kf, -0x1.a39ef35793c76p-33, r);
  double p = 0x1.6124613a86d09p-33;                       // 1/13!
  p = __builtin_fma(p, r, 0x1.1eed8eff8d898p-29);         // 1/12!
  p = __builtin_fma(p, r, 0x1.ae64567f544e4p-26);         // 1/11!
  p = __builtin_fma(p, r, 0x1.27e4fb7789f5cp-22);         // 1/10!
  p = __builtin_fma(p, r, 0x1.71de3a556c734p-19);         // 1/9!
  p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-16);         // 1/8!
  p = __builtin_fma(p, r, 0x1.a01a01a01a01ap-13);         // 1/7!
  p = __builtin_fma(p, r, 0x1.6c16c16c16c17p-10);         // 1/6!
  p = __builtin_fma(p, r, 0x1.1111111111111p-7);          // 1/5!
  p = __builtin_fma(p, r, 0x1.5555555555555p-5);          // 1/4!
  p = __builtin_fma(p, r, 0x1.5555555555555p-3);          // 1/3!
  p = __builtin_fma(p, r, 0x1p-1);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  const double u = __builtin_ldexp(p, (int)kf);            // v_ldexp_f64: correct gradual underflow
  const double d = 1.0 + u;
  const double rd = rh_rcp_1to4(d);
  const double sg = t >= 0.0 ? rd : u * rd;
  // log(1 + u) = log1p(u): f is taken from u itself, not from d - 1, so nothing is lost when u is tiny
  const bool hi = u > 0x1.a827999fcef34p-2;                // 1 + u > sqrt 2: halve the argument, f = (1 + u) / 2 - 1
  const double f = hi ? __builtin_fma(u, 0.5, -0.5) : u;
  const double z = f * rh_rcp_1to4(2.0 + f);
  const double z2 = z * z, z4 = z2 * z2;
  const double t1 = z4 * __builtin_fma(z4, __builtin_fma(z4, 0x1.39a09d078c69fp-3, 0x1.c71c51d8e78afp-3), 0x1.999999997fa04p-2);
  const double t2 = z2 * __builtin_fma(z4, __builtin_fma(z4, __builtin_fma(z4, 0x1.2f112df3e5244p-3, 0x1.7466496cb03dep-3), 0x1.2492494229359p-2),
                                      0x1.5555555555593p-1);
  const double R = t1 + t2, hfsq = 0.5 * f * f;
  const double lg = (f - (hfsq - z * (hfsq + R))) + (hi ? 0x1.62e42fefa39efp-1 : 0.0);   // + ln 2 in the halved branch
  // max(t, 0) spelled as a select on (t <= 0): a NaN t (a diverged chain) fails the comparison and passes through, so the VALUE
  // -- hence the energy -- is NaN and the proposal is rejected (LeapFrog.scala:138-142); a max instruction would drop it, and
  // (t + |t|) / 2 would turn t = -inf into NaN.  sigmoid of a NaN t is finite garbage that is never used.
  softplus = (t <= 0.0 ? 0.0 : t) + lg;
  sigmoid = sg;
}

RH_DEV double rh_strict_sqrt(double x) { return __builtin_sqrt(x); } // IEEE correctly rounded (checked by tests)
// Math.pow(t, -0.75) for the dual-averaging decay, t a positive integer: sqrt-composed, bit-reproducible
RH_DEV double rh_pow_neg075(double t) { const double r = rh_strict_sqrt(t); return 1.0 / (r * rh_strict_sqrt(r)); }

// ---- JVM op semantics used by generated model code (ir/MethodGenerator.scala:56-94,134-167) -------
RH_DEV double rh_compare(double l, double r) { return l > r ? 1.0 : (l == r ? 0.0 : -1.0); } // DCMPL; I2D
RH_DEV int rh_d2i(double x) { // D2I: NaN -> 0, saturating
  if (x != x) return 0;
  if (x >= 2147483647.0) return 2147483647;
  if (x <= -2147483648.0) return (-2147483647 - 1);
  return (int)x;
}
// fast mode: Math.pow(x, 0.5).  Differs from sqrt only at -0.0 (pow: +0.0) and -inf (pow: +inf); x + 0.0 turns -0.0 into +0.0.
RH_DEV double rh_pow_half(double x) { return x == -RH_INF ? RH_INF : __builtin_sqrt(x + 0.0); }
RH_DEV double rh_java_pow(double x, double y) { // java.lang.Math.pow = C99 pow + two Java-specific NaN cases
  if (y == 0.0) return 1.0;
  if (y != y) return RH_NAN;
  if (__builtin_isinf(y) && __builtin_fabs(x) == 1.0) return RH_NAN;
  return pow(x, y);
}

// ---- java.util.Random, one stream per chain, replicated in every lane (wave-uniform) -------------
struct rh_rng {
  rh_u64 seed;
  rh_i64 have; // haveNextNextGaussian
  double nn;   // nextNextGaussian
};
RH_DEV void rh_rng_init(rh_rng &r, rh_i64 seed) {
  r.seed = ((rh_u64)seed ^ 0x5DEECE66DULL) & ((1ULL << 48) - 1);
  r.have = 0; r.nn = 0.0;
}
RH_DEV int rh_rng_next(rh_rng &r, int bits) {
  r.seed = (r.seed * 0x5DEECE66DULL + 0xBULL) & ((1ULL << 48) - 1);
  return (int)(r.seed >> (48 - bits));
}
RH_DEV double rh_rng_uniform(rh_rng &r) { // nextDouble  == RNG.standardUniform
  const rh_i64 a = (rh_i64)rh_rng_next(r, 26);
  const rh_i64 b = (rh_i64)rh_rng_next(r, 27);
  return (double)((a << 27) + b) * 0x1.0p-53;
}
RH_DEV double rh_rng_normal(rh_rng &r) { // nextGaussian == RNG.standardNormal
  if (r.have) { r.have = 0; return r.nn; }
  double v1, v2, s;
  do {
    v1 = 2 * rh_rng_uniform(r) - 1;
    v2 = 2 * rh_rng_uniform(r) - 1;
    s = v1 * v1 + v2 * v2;
  } while (s >= 1 || s == 0);
  const double multiplier = rh_strict_sqrt(-2 * rh_strict_log(s) / s);
  r.nn = v2 * multiplier;
  r.have = 1;
  return v1 * multiplier;
}
RH_DEV int rh_rng_int(rh_rng &r, int until) { // RNG.int (sampler/RNG.scala:9-10)
  const int v = (int)(rh_rng_uniform(r) * until);
  return v < until - 1 ? v : until - 1;
}

// ---- wave64 primitives ---------------------------------------------------------------------------
// Chain packing (data-free models with few parameters: cfg 1, cfg 3): RH_PACK_L lanes per chain, 64 / RH_PACK_L chains per
// wavefront.  Everything a chain does is written against these few primitives; "wave-uniform" then means "identical in the
// RH_PACK_L lanes of the chain" (such values simply live in VGPRs) and the automaton's control flow diverges between the
// chains of a wavefront like any SIMT code.  The arithmetic of a chain does not change, so packed chains stay bit-exact.
#ifndef RH_PACK_L
#define RH_PACK_L 64
#endif
#define RH_LANES RH_PACK_L
#if RH_PACK_L == 64
RH_DEV double rh_readlane(double v, int lane) { // wave-uniform broadcast of one lane's value (SGPR pair)
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
RH_DEV double rh_uniform(double v) { return rh_readlane(v, 0); }
RH_DEV int rh_uniform_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
#else
RH_DEV double rh_readlane(double v, int lane) { // broadcast of lane `lane` of the chain's own lane group (ds_bpermute)
  return __shfl(v, (int)(threadIdx.x & ~(unsigned)(RH_PACK_L - 1)) + lane, 64);
}
RH_DEV double rh_uniform(double v) { return v; }
RH_DEV int rh_uniform_i(int v) { return v; }
#endif
// value of lane `src` (a per-lane index within the chain's lane group)
RH_DEV double rh_gather(double v, int src) {
#if RH_PACK_L == 64
  return __shfl(v, src, 64);
#else
  return __shfl(v, (int)(threadIdx.x & ~(unsigned)(RH_PACK_L - 1)) + src, 64);
#endif
}
// fixed-order butterfly over the chain's lanes: every lane ends with the bitwise-identical sum (run-to-run reproducible)
RH_DEV double rh_wave_sum(double v) {
#pragma unroll
  for (int off = RH_LANES / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// length-RH_NVARS vector, element i in lane (i % 64), slot (i / 64); unused lanes hold 0.
// Two storage classes behind one interface (v.s[k], wv_* helpers, assignment copies the data):
//   register mode (RH_BIGN == 0, nVars <= 512): RH_SLOTS doubles per lane in VGPRs, slot loops fully unrolled;
//   big mode      (RH_BIGN == 1, e.g. cfg 5's 10 004 parameters): the vector lives in the chain's state block in HBM,
//                 element i at base[i]; v.s[k] is a reference to base[k*64 + lane]; slot loops stay rolled;
//                 temporaries (RH_TMP) come from a small per-chain pool in the same block (stack discipline).
#ifndef RH_BIGN
#define RH_BIGN 0
#endif
#if !RH_BIGN
#define RH_UNROLL_SLOTS _Pragma("unroll")
struct wvec { double s[RH_SLOTS]; };
#define RH_TMP(name) wvec name
RH_DEV double wv_elem(const wvec &v, int i) { // wave-uniform i -> wave-uniform value (SGPR pair)
  return rh_readlane(v.s[i >> 6], i & 63);
}
#else
#define RH_UNROLL_SLOTS _Pragma("unroll 1")
struct rh_slotref {
  double *p;
  RH_DEV double &operator[](int k) const { return p[(size_t)k * 64 + threadIdx.x]; }
};
#define RH_POOL_VECS 12
__shared__ double *rh_pool_base; // per chain (one chain per workgroup): RH_POOL_VECS scratch vectors
__shared__ int rh_pool_depth[64];
struct wvec {
  rh_slotref s;
  bool pooled;
  RH_DEV wvec() { s.p = nullptr; pooled = false; }
  struct from_pool {};
  RH_DEV explicit wvec(from_pool) {
    const int d = rh_pool_depth[threadIdx.x];
    rh_pool_depth[threadIdx.x] = d + 1;
    s.p = rh_pool_base + (size_t)d * RH_SLOTS * 64;
    pooled = true;
  }
  RH_DEV ~wvec() { if (pooled) rh_pool_depth[threadIdx.x] -= 1; }
  RH_DEV wvec(const wvec &) = delete;
  RH_DEV wvec &operator=(const wvec &o) { // copies the DATA (each lane its own elements); 8 loads in flight per lane
    int k = 0;
    _Pragma("unroll 1") for (; k + 8 <= RH_SLOTS; k += 8) {
      double t[8];
      _Pragma("unroll") for (int j = 0; j < 8; j++) t[j] = o.s[k + j];
      _Pragma("unroll") for (int j = 0; j < 8; j++) s[k + j] = t[j];
    }
    _Pragma("unroll 1") for (; k < RH_SLOTS; k++) s[k] = o.s[k];
    return *this;
  }
};
#define RH_TMP(name) wvec name{wvec::from_pool{}}
RH_DEV double wv_elem(const wvec &v, int i) { return v.s.p[i]; } // wave-uniform address
#endif
RH_DEV void wv_zero(wvec &v) {
  RH_UNROLL_SLOTS
  for (int k = 0; k < RH_SLOTS; k++) v.s[k] = 0.0;
}
RH_DEV void wv_fill(wvec &v, double x, int lane) {
  RH_UNROLL_SLOTS
  for (int k = 0; k < RH_SLOTS; k++) v.s[k] = (k * 64 + lane < RH_NVARS) ? x : 0.0;
}
#if RH_BIGN
// big mode: the vectors live in HBM and one wavefront walks RH_SLOTS x 512 B of them; the loads of 8 slots are issued
// before the first dependent store so that the walk is bandwidth- rather than latency-bound (the pointers may alias as far
// as the compiler knows, so it cannot do this itself)
#define RH_BIG2(expr)                                                                                  \
  int k = 0;                                                                                             \
  _Pragma("unroll 1") for (; k + 8 <= RH_SLOTS; k += 8) {                                              \
    double xa[8], ya[8];                                                                                 \
    _Pragma("unroll") for (int j = 0; j < 8; j++) { xa[j] = x.s[k + j]; ya[j] = y.s[k + j]; }          \
    _Pragma("unroll") for (int j = 0; j < 8; j++) { const double xv = xa[j], yv = ya[j]; (void)xv; (void)yv; o.s[k + j] = (expr); } \
  }                                                                                                      \
  _Pragma("unroll 1") for (; k < RH_SLOTS; k++) { const double xv = x.s[k], yv = y.s[k]; (void)xv; (void)yv; o.s[k] = (expr); }
// y(i) += a * x(i)   -- multiply, round, add, round (LeapFrog.scala:148,170)
RH_DEV void wv_axpy(wvec &y, double a, const wvec &x) { wvec &o = y; RH_BIG2(yv + a * xv) }
RH_DEV void wv_mul(wvec &o, const wvec &x, const wvec &y) { RH_BIG2(xv * yv) }
RH_DEV void wv_sub(wvec &o, const wvec &x, const wvec &y) { RH_BIG2(xv - yv) }
#else
// y(i) += a * x(i)   -- multiply, round, add, round (LeapFrog.scala:148,170)
RH_DEV void wv_axpy(wvec &y, double a, const wvec &x) {
  RH_UNROLL_SLOTS
  for (int k = 0; k < RH_SLOTS; k++) y.s[k] += a * x.s[k];
}
RH_DEV void wv_mul(wvec &out, const wvec &x, const wvec &y) {
  RH_UNROLL_SLOTS
  for (int k = 0; k < RH_SLOTS; k++) out.s[k] = x.s[k] * y.s[k];
}
RH_DEV void wv_sub(wvec &out, const wvec &x, const wvec &y) {
  RH_UNROLL_SLOTS
  for (int k = 0; k < RH_SLOTS; k++) out.s[k] = x.s[k] - y.s[k];
}
#endif
RH_DEV void wv_set(wvec &v, int i, double x, int lane) { // i wave-uniform
#if RH_BIGN
  if (lane == (i & 63)) v.s.p[i] = x;
#else
  RH_UNROLL_SLOTS
  for (int k = 0; k < RH_SLOTS; k++) v.s[k] = (k * 64 + lane == i) ? x : v.s[k];
#endif
}
RH_DEV double wv_get(const wvec &v, int i) { return wv_elem(v, i); } // i wave-uniform; returns wave-uniform value
// sum_{i<n} x(i), strictly left to right like the reference's while loops (LeapFrog.scala:218-227)
RH_DEV double wv_sum_seq(const wvec &x) {
#if RH_BIGN
  // big mode: per-lane partial sums over ascending slots + the fixed-order butterfly (deterministic; not the
  // reference's strictly sequential order -- tolerance parity, like the row sums)
  double part = 0.0;
  _Pragma("unroll 1") for (int k = 0; k < RH_SLOTS; k++) part += (k * 64 + (int)threadIdx.x < RH_NVARS) ? x.s[k] : 0.0;
  return rh_wave_sum(part);
#else
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < RH_SLOTS; k++) {
    const int m = (RH_NVARS - k * 64) < 64 ? (RH_NVARS - k * 64) : 64;
    if (RH_NVARS <= 64) {
#pragma unroll
      for (int l = 0; l < m; l++) acc += rh_readlane(x.s[k], l);
    } else {
      for (int l = 0; l < m; l++) acc += rh_readlane(x.s[k], l);
    }
  }
  return acc;
#endif
}
