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
// (the closed form of a Bernoulli-logit likelihood term and of its adjoint, see emit.cpp detect_logit / detect_link; the scalar
// part of cfg 4 and cfg 5 -- it was 71 of cfg 4's 95 and of cfg 5's ~100 VALU instructions per row-chain evaluation, so it is
// written for instruction count: no division, no reciprocal instruction (v_rcp_f64 issues at a quarter of the FMA rate).)
//   u = e^{-|t|} in (0, 1]   (never overflows; underflows to 0 for |t| > 745, where softplus = max(t, 0) and sigmoid = 0 / 1 exactly)
//   table bucket j = round(256 u):  c_j = 1 + j/256,  rc_j = fl(1 / c_j),  L_j = fl(-log rc_j)      (rh_lk_tab, tools/gen_lk_table.py)
//   r = (1 + u) rc_j - 1 = fma(u, rc_j, rc_j - 1):  ONE rounding (rc_j - 1 is exact), |r| < 2^-9, and r = u exactly in bucket 0 --
//       a tiny u keeps all its digits
//   log1p(u)    = L_j + log1p(r),   log1p(r) = r - r^2 (1/2 - r/3 + r^2/4 - r^3/5 + r^4/6)               (|r|^7 / 7 < 2^-65)
//   1 / (1 + u) = rc_j / (1 + r):   y0 = rc_j (1 - r + r^2)  (relative error r^3 < 2^-27), one Newton step against 1 + u
//   sigmoid(t)  = t >= 0 ? y : u y;   softplus(t) = max(t, 0) + log1p(u)
// Both results are within 3 ulp of the exact values over the whole range (host build of this very text against binary128:
// 2.0 / 2.7 ulp, tests/test_oracle.py; tests/test_gpu_parity.py asserts 3 ulp on the device).
__device__ __attribute__((aligned(16))) const double rh_lk_tab[2 * 257] = {
  0x1.0000000000000p+0, 0x0.0p+0, 0x1.fe01fe01fe020p-1, 0x1.ff00aa2b10ba0p-9, 0x1.fc07f01fc07f0p-1, 0x1.fe02a6b106799p-8,
  0x1.fa11caa01fa12p-1, 0x1.7dc475f810a69p-7, 0x1.f81f81f81f820p-1, 0x1.fc0a8b0fc03c4p-7, 0x1.f6310aca0dbb5p-1, 0x1.3cea44346a584p-6,
  0x1.f44659e4a4271p-1, 0x1.7b91b07d5b126p-6, 0x1.f25f644230ab5p-1, 0x1.b9fc027af919ap-6, 0x1.f07c1f07c1f08p-1, 0x1.f829b0e7832f8p-6,
  0x1.ee9c7f8458e02p-1, 0x1.1b0d98923d97fp-5, 0x1.ecc07b301ecc0p-1, 0x1.39e87b9febd68p-5, 0x1.eae807aba01ebp-1, 0x1.58a5bafc8e4d3p-5,
  0x1.e9131abf0b767p-1, 0x1.77458f632dcffp-5, 0x1.e741aa59750e4p-1, 0x1.95c830ec8e3f2p-5, 0x1.e573ac901e574p-1, 0x1.b42dd711971b9p-5,
  0x1.e3a9179dc1a73p-1, 0x1.d276b8adb0b56p-5, 0x1.e1e1e1e1e1e1ep-1, 0x1.f0a30c01162a8p-5, 0x1.e01e01e01e01ep-1, 0x1.075983598e471p-4,
  0x1.de5d6e3f8868ap-1, 0x1.16536eea37ae3p-4, 0x1.dca01dca01dcap-1, 0x1.253f62f0a1417p-4, 0x1.dae6076b981dbp-1, 0x1.341d7961bd1d0p-4,
  0x1.d92f2231e7f8ap-1, 0x1.42edcbea646eep-4, 0x1.d77b654b82c34p-1, 0x1.51b073f06183cp-4, 0x1.d5cac807572b2p-1, 0x1.60658a93750c4p-4,
  0x1.d41d41d41d41dp-1, 0x1.6f0d28ae56b4ep-4, 0x1.d272ca3fc5b1ap-1, 0x1.7da766d7b12d0p-4, 0x1.d0cb58f6ec074p-1, 0x1.8c345d6319b23p-4,
  0x1.cf26e5c44bfc6p-1, 0x1.9ab42462033aep-4, 0x1.cd85689039b0bp-1, 0x1.a926d3a4ad562p-4, 0x1.cbe6d9601cbe7p-1, 0x1.b78c82bb0eda0p-4,
  0x1.ca4b3055ee191p-1, 0x1.c5e548f5bc743p-4, 0x1.c8b265afb8a42p-1, 0x1.d4313d66cb35dp-4, 0x1.c71c71c71c71cp-1, 0x1.e27076e2af2eap-4,
  0x1.c5894d10d4986p-1, 0x1.f0a30c01162a4p-4, 0x1.c3f8f01c3f8f0p-1, 0x1.fec9131dbeabcp-4, 0x1.c26b5392ea01cp-1, 0x1.0671512ca596fp-3,
  0x1.c0e070381c0e0p-1, 0x1.0d77e7cd08e5bp-3, 0x1.bf583ee868d8bp-1, 0x1.14785846742acp-3, 0x1.bdd2b899406f7p-1, 0x1.1b72ad52f67a2p-3,
  0x1.bc4fd65883e7bp-1, 0x1.2266f190a5acdp-3, 0x1.bacf914c1bad0p-1, 0x1.29552f81ff521p-3, 0x1.b951e2b18ff23p-1, 0x1.303d718e47fd5p-3,
  0x1.b7d6c3dda338bp-1, 0x1.371fc201e8f75p-3, 0x1.b65e2e3beee05p-1, 0x1.3dfc2b0ecc62ap-3, 0x1.b4e81b4e81b4fp-1, 0x1.44d2b6ccb7d1cp-3,
  0x1.b37484ad806cep-1, 0x1.4ba36f39a55e5p-3, 0x1.b2036406c80d9p-1, 0x1.526e5e3a1b438p-3, 0x1.b094b31d922a4p-1, 0x1.59338d9982085p-3,
  0x1.af286bca1af28p-1, 0x1.5ff3070a793d6p-3, 0x1.adbe87f94905ep-1, 0x1.66acd4272ad51p-3, 0x1.ac5701ac5701bp-1, 0x1.6d60fe719d21bp-3,
  0x1.aaf1d2f87ebfdp-1, 0x1.740f8f54037a3p-3, 0x1.a98ef606a63bep-1, 0x1.7ab890210d907p-3, 0x1.a82e65130e159p-1, 0x1.815c0a14357e9p-3,
  0x1.a6d01a6d01a6dp-1, 0x1.87fa06520c911p-3, 0x1.a574107688a4ap-1, 0x1.8e928de886d41p-3, 0x1.a41a41a41a41ap-1, 0x1.9525a9cf456b6p-3,
  0x1.a2c2a87c51ca0p-1, 0x1.9bb362e7dfb85p-3, 0x1.a16d3f97a4b02p-1, 0x1.a23bc1fe2b561p-3, 0x1.a01a01a01a01ap-1, 0x1.a8becfc882f19p-3,
  0x1.9ec8e951033d9p-1, 0x1.af3c94e80bff3p-3, 0x1.9d79f176b682dp-1, 0x1.b5b519e8fb5a6p-3, 0x1.9c2d14ee4a102p-1, 0x1.bc286742d8cd4p-3,
  0x1.9ae24ea5510dap-1, 0x1.c2968558c18c2p-3, 0x1.999999999999ap-1, 0x1.c8ff7c79a9a20p-3, 0x1.9852f0d8ec0ffp-1, 0x1.cf6354e09c5ddp-3,
  0x1.970e4f80cb872p-1, 0x1.d5c216b4fbb94p-3, 0x1.95cbb0be377aep-1, 0x1.dc1bca0abec7bp-3, 0x1.948b0fcd6e9e0p-1, 0x1.e27076e2af2e8p-3,
  0x1.934c67f9b2ce6p-1, 0x1.e8c0252aa5a60p-3, 0x1.920fb49d0e229p-1, 0x1.ef0adcbdc5935p-3, 0x1.90d4f120190d5p-1, 0x1.f550a564b7b37p-3,
  0x1.8f9c18f9c18fap-1, 0x1.fb9186d5e3e29p-3, 0x1.8e6527af1373fp-1, 0x1.00e6c45ad501dp-2, 0x1.8d3018d3018d3p-1, 0x1.0402594b4d041p-2,
  0x1.8bfce8062ff3ap-1, 0x1.071b85fcd590dp-2, 0x1.8acb90f6bf3aap-1, 0x1.0a324e27390e2p-2, 0x1.899c0f601899cp-1, 0x1.0d46b579ab74bp-2,
  0x1.886e5f0abb04ap-1, 0x1.1058bf9ae4ad4p-2, 0x1.87427bcc092b9p-1, 0x1.136870293a8b0p-2, 0x1.8618618618618p-1, 0x1.1675cababa60fp-2,
  0x1.84f00c2780614p-1, 0x1.1980d2dd4236fp-2, 0x1.83c977ab2beddp-1, 0x1.1c898c16999fbp-2, 0x1.82a4a0182a4a0p-1, 0x1.1f8ff9e48a2f3p-2,
  0x1.8181818181818p-1, 0x1.22941fbcf7966p-2, 0x1.8060180601806p-1, 0x1.2596010df763ap-2, 0x1.7f405fd017f40p-1, 0x1.2895a13de86a4p-2,
  0x1.7e225515a4f1dp-1, 0x1.2b9303ab89d25p-2, 0x1.7d05f417d05f4p-1, 0x1.2e8e2bae11d31p-2, 0x1.7beb3922e017cp-1, 0x1.31871c9544185p-2,
  0x1.7ad2208e0ecc3p-1, 0x1.347dd9a987d56p-2, 0x1.79baa6bb6398bp-1, 0x1.3772662bfd85cp-2, 0x1.78a4c8178a4c8p-1, 0x1.3a64c556945eap-2,
  0x1.77908119ac60dp-1, 0x1.3d54fa5c1f710p-2, 0x1.767dce434a9b1p-1, 0x1.404308686a7e4p-2, 0x1.756cac201756dp-1, 0x1.432ef2a04e813p-2,
  0x1.745d1745d1746p-1, 0x1.4618bc21c5ec2p-2, 0x1.734f0c541fe8dp-1, 0x1.49006804009d0p-2, 0x1.724287f46debcp-1, 0x1.4be5f957778a1p-2,
  0x1.713786d9c7c09p-1, 0x1.4ec9732600269p-2, 0x1.702e05c0b8170p-1, 0x1.51aad872df82ep-2, 0x1.6f26016f26017p-1, 0x1.548a2c3add263p-2,
  0x1.6e1f76b4337c7p-1, 0x1.5767717455a6cp-2, 0x1.6d1a62681c861p-1, 0x1.5a42ab0f4cfe2p-2, 0x1.6c16c16c16c17p-1, 0x1.5d1bdbf5809cap-2,
  0x1.6b1490aa31a3dp-1, 0x1.5ff3070a793d4p-2, 0x1.6a13cd1537290p-1, 0x1.62c82f2b9c796p-2, 0x1.691473a88d0c0p-1, 0x1.659b57303e1f2p-2,
  0x1.6816816816817p-1, 0x1.686c81e9b14adp-2, 0x1.6719f3601671ap-1, 0x1.6b3bb2235943dp-2, 0x1.661ec6a5122f9p-1, 0x1.6e08eaa2ba1e4p-2,
  0x1.6524f853b4aa3p-1, 0x1.70d42e2789236p-2, 0x1.642c8590b2164p-1, 0x1.739d7f6bbd007p-2, 0x1.63356b88ac0dep-1, 0x1.7664e1239dbcfp-2,
  0x1.623fa77016240p-1, 0x1.792a55fdd47a1p-2, 0x1.614b36831ae94p-1, 0x1.7bede0a37afbfp-2, 0x1.6058160581606p-1, 0x1.7eaf83b82afc2p-2,
  0x1.5f66434292dfcp-1, 0x1.816f41da0d495p-2, 0x1.5e75bb8d015e7p-1, 0x1.842d1da1e8b18p-2, 0x1.5d867c3ece2a5p-1, 0x1.86e919a330ba1p-2,
  0x1.5c9882b931057p-1, 0x1.89a3386c1425bp-2, 0x1.5babcc647fa91p-1, 0x1.8c5b7c858b48bp-2, 0x1.5ac056b015ac0p-1, 0x1.8f11e873662c8p-2,
  0x1.59d61f123ccaap-1, 0x1.91c67eb45a83ep-2, 0x1.58ed2308158edp-1, 0x1.947941c2116fbp-2, 0x1.5805601580560p-1, 0x1.972a341135159p-2,
  0x1.571ed3c506b3ap-1, 0x1.99d958117e08ap-2, 0x1.56397ba7c52e2p-1, 0x1.9c86b02dc0862p-2, 0x1.5555555555555p-1, 0x1.9f323ecbf984dp-2,
  0x1.54725e6bb82fep-1, 0x1.a1dc064d5b995p-2, 0x1.5390948f40febp-1, 0x1.a484090e5bb09p-2, 0x1.52aff56a8054bp-1, 0x1.a72a4966bd9e9p-2,
  0x1.51d07eae2f815p-1, 0x1.a9cec9a9a084ap-2, 0x1.50f22e111c4c5p-1, 0x1.ac718c258b0e5p-2, 0x1.5015015015015p-1, 0x1.af1293247786bp-2,
  0x1.4f38f62dd4c9bp-1, 0x1.b1b1e0ebdfc5ap-2, 0x1.4e5e0a72f0539p-1, 0x1.b44f77bcc8f64p-2, 0x1.4d843bedc2c4cp-1, 0x1.b6eb59d3cf35cp-2,
  0x1.4cab88725af6ep-1, 0x1.b9858969310fdp-2, 0x1.4bd3edda68fe1p-1, 0x1.bc1e08b0dad0ap-2, 0x1.4afd6a052bf5bp-1, 0x1.beb4d9da71b7ap-2,
  0x1.4a27fad76014ap-1, 0x1.c149ff115f027p-2, 0x1.49539e3b2d067p-1, 0x1.c3dd7a7cdad4dp-2, 0x1.4880522014880p-1, 0x1.c66f4e3ff6ff9p-2,
  0x1.47ae147ae147bp-1, 0x1.c8ff7c79a9a21p-2, 0x1.46dce34596066p-1, 0x1.cb8e0744d7acap-2, 0x1.460cbc7f5cf9ap-1, 0x1.ce1af0b85f3ecp-2,
  0x1.453d9e2c776cap-1, 0x1.d0a63ae721e64p-2, 0x1.446f86562d9fbp-1, 0x1.d32fe7e00ebd5p-2, 0x1.43a2730abee4dp-1, 0x1.d5b7f9ae2c684p-2,
  0x1.42d6625d51f87p-1, 0x1.d83e7258a2f3ep-2, 0x1.420b5265e5951p-1, 0x1.dac353e2c5955p-2, 0x1.4141414141414p-1, 0x1.dd46a04c1c4a1p-2,
  0x1.40782d10e6566p-1, 0x1.dfc859906d5b5p-2, 0x1.3fb013fb013fbp-1, 0x1.e24881a7c6c26p-2, 0x1.3ee8f42a5af07p-1, 0x1.e4c71a8687704p-2,
  0x1.3e22cbce4a902p-1, 0x1.e744261d68789p-2, 0x1.3d5d991aa75c6p-1, 0x1.e9bfa659861f5p-2, 0x1.3c995a47babe7p-1, 0x1.ec399d2468cc1p-2,
  0x1.3bd60d9232955p-1, 0x1.eeb20c640ddf3p-2, 0x1.3b13b13b13b14p-1, 0x1.f128f5faf06ecp-2, 0x1.3a524387ac822p-1, 0x1.f39e5bc811e5dp-2,
  0x1.3991c2c187f63p-1, 0x1.f6123fa7028adp-2, 0x1.38d22d366088ep-1, 0x1.f884a36fe9ec1p-2, 0x1.3813813813814p-1, 0x1.faf588f78f31dp-2,
  0x1.3755bd1c945eep-1, 0x1.fd64f20f61571p-2, 0x1.3698df3de0748p-1, 0x1.ffd2e0857f497p-2, 0x1.35dce5f9f2af8p-1, 0x1.011fab125ff8ap-1,
  0x1.3521cfb2b78c1p-1, 0x1.02552a5a5d0ffp-1, 0x1.34679ace01346p-1, 0x1.0389eefce633cp-1, 0x1.33ae45b57bcb2p-1, 0x1.04bdf9da926d2p-1,
  0x1.32f5ced6a1dfap-1, 0x1.05f14bd26459cp-1, 0x1.323e34a2b10bfp-1, 0x1.0723e5c1cdf41p-1, 0x1.3187758e9ebb6p-1, 0x1.0855c884b450ep-1,
  0x1.30d190130d190p-1, 0x1.0986f4f573521p-1, 0x1.301c82ac40260p-1, 0x1.0ab76bece14d2p-1, 0x1.2f684bda12f68p-1, 0x1.0be72e4252a83p-1,
  0x1.2eb4ea1fed14bp-1, 0x1.0d163ccb9d6b8p-1, 0x1.2e025c04b8097p-1, 0x1.0e44985d1cc8cp-1, 0x1.2d50a012d50a0p-1, 0x1.0f7241c9b497dp-1,
  0x1.2c9fb4d812ca0p-1, 0x1.109f39e2d4c96p-1, 0x1.2bef98e5a3711p-1, 0x1.11cb81787ccf8p-1, 0x1.2b404ad012b40p-1, 0x1.12f719593efbdp-1,
  0x1.2a91c92f3c105p-1, 0x1.1422025243d45p-1, 0x1.29e4129e4129ep-1, 0x1.154c3d2f4d5eap-1, 0x1.293725bb804a5p-1, 0x1.1675cababa60ep-1,
  0x1.288b01288b013p-1, 0x1.179eabbd899a0p-1, 0x1.27dfa38a1ce4dp-1, 0x1.18c6e0ff5cf07p-1, 0x1.27350b8812735p-1, 0x1.19ee6b467c96fp-1,
  0x1.268b37cd60127p-1, 0x1.1b154b57da29ep-1, 0x1.25e22708092f1p-1, 0x1.1c3b81f713c25p-1, 0x1.2539d7e9177b2p-1, 0x1.1d610fe677003p-1,
  0x1.2492492492492p-1, 0x1.1e85f5e7040d1p-1, 0x1.23eb79717605bp-1, 0x1.1faa34b87094cp-1, 0x1.23456789abcdfp-1, 0x1.20cdcd192ab6ep-1,
  0x1.22a0122a0122ap-1, 0x1.21f0bfc65beecp-1, 0x1.21fb78121fb78p-1, 0x1.23130d7bebf43p-1, 0x1.21579804855e6p-1, 0x1.2434b6f483934p-1,
  0x1.20b470c67c0d9p-1, 0x1.2555bce98f7cap-1, 0x1.2012012012012p-1, 0x1.26762013430e0p-1, 0x1.1f7047dc11f70p-1, 0x1.2795e1289b11bp-1,
  0x1.1ecf43c7fb84cp-1, 0x1.28b500df60783p-1, 0x1.1e2ef3b3fb874p-1, 0x1.29d37fec2b08bp-1, 0x1.1d8f5672e4abdp-1, 0x1.2af15f02640acp-1,
  0x1.1cf06ada2811dp-1, 0x1.2c0e9ed448e8cp-1, 0x1.1c522fc1ce059p-1, 0x1.2d2b4012edc9dp-1, 0x1.1bb4a4046ed29p-1, 0x1.2e47436e40268p-1,
  0x1.1b17c67f2bae3p-1, 0x1.2f62a99509546p-1, 0x1.1a7b9611a7b96p-1, 0x1.307d7334f10bep-1, 0x1.19e0119e0119ep-1, 0x1.3197a0fa7fe6ap-1,
  0x1.19453808ca29cp-1, 0x1.32b1339121d71p-1, 0x1.18ab083902bdbp-1, 0x1.33ca2ba328994p-1, 0x1.1811811811812p-1, 0x1.34e289d9ce1d2p-1,
  0x1.1778a191bd684p-1, 0x1.35fa4edd36ea0p-1, 0x1.16e0689427379p-1, 0x1.37117b54747b6p-1, 0x1.1648d50fc3201p-1, 0x1.38280fe58797fp-1,
  0x1.15b1e5f75270dp-1, 0x1.393e0d3562a1ap-1, 0x1.151b9a3fdd5c9p-1, 0x1.3a5373e7ebdf9p-1, 0x1.1485f0e0acd3bp-1, 0x1.3b68449fffc23p-1,
  0x1.13f0e8d344724p-1, 0x1.3c7c7fff73206p-1, 0x1.135c81135c811p-1, 0x1.3d9026a7156fbp-1, 0x1.12c8b89edc0acp-1, 0x1.3ea33936b2f5bp-1,
  0x1.12358e75d3033p-1, 0x1.3fb5b84d16f43p-1, 0x1.11a3019a74826p-1, 0x1.40c7a4880dceap-1, 0x1.1111111111111p-1, 0x1.41d8fe84672afp-1,
  0x1.107fbbe011080p-1, 0x1.42e9c6ddf80bfp-1, 0x1.0fef010fef011p-1, 0x1.43f9fe2f9ce67p-1, 0x1.0f5edfab325a2p-1, 0x1.4509a5133bb0ap-1,
  0x1.0ecf56be69c90p-1, 0x1.4618bc21c5ec2p-1, 0x1.0e40655826011p-1, 0x1.472743f33aaadp-1, 0x1.0db20a88f4696p-1, 0x1.48353d1ea88dfp-1,
  0x1.0d24456359e3ap-1, 0x1.4942a83a2fc07p-1, 0x1.0c9714fbcda3bp-1, 0x1.4a4f85db03ebbp-1, 0x1.0c0a7868b4171p-1, 0x1.4b5bd6956e273p-1,
  0x1.0b7e6ec259dc8p-1, 0x1.4c679afccee39p-1, 0x1.0af2f722eecb5p-1, 0x1.4d72d3a39fd01p-1, 0x1.0a6810a6810a7p-1, 0x1.4e7d811b75bb0p-1,
  0x1.09ddba6af8360p-1, 0x1.4f87a3f5026e9p-1, 0x1.0953f39010954p-1, 0x1.50913cc01686bp-1, 0x1.08cabb37565e2p-1, 0x1.519a4c0ba3446p-1,
  0x1.0842108421084p-1, 0x1.52a2d265bc5abp-1, 0x1.07b9f29b8eae2p-1, 0x1.53aad05b99b7cp-1, 0x1.073260a47f7c6p-1, 0x1.54b2467999498p-1,
  0x1.06ab59c7912fbp-1, 0x1.55b9354b40bcep-1, 0x1.0624dd2f1a9fcp-1, 0x1.56bf9d5b3f399p-1, 0x1.059eea0727586p-1, 0x1.57c57f336f191p-1,
  0x1.05197f7d73404p-1, 0x1.58cadb5cd7989p-1, 0x1.04949cc1664c5p-1, 0x1.59cfb25fae87fp-1, 0x1.0410410410410p-1, 0x1.5ad404c359f2dp-1,
  0x1.038c6b78247fcp-1, 0x1.5bd7d30e71c73p-1, 0x1.03091b51f5e1ap-1, 0x1.5cdb1dc6c1765p-1, 0x1.02864fc7729e9p-1, 0x1.5ddde57149923p-1,
  0x1.0204081020408p-1, 0x1.5ee02a9241676p-1, 0x1.0182436517a37p-1, 0x1.5fe1edad18919p-1, 0x1.0101010101010p-1, 0x1.60e32f44788d9p-1,
  0x1.0080402010080p-1, 0x1.61e3efda46467p-1, 0x1.0000000000000p-1, 0x1.62e42fefa39efp-1
};
// 2^(j/128), j = 0..127, as pairs (hi, lo): hi the correctly rounded value, lo = 2^(j/128) - hi (tools/gen_lk_table.py --exp).
// rh_logit_link's e^{-|t|} = 2^k 2^(j/128) e^r with |r| <= ln2 / 256; the lo word keeps the table's own rounding out of the result.
__device__ __attribute__((aligned(16))) const double rh_ex_tab[2 * 128] = {
  0x1.0000000000000p+0, 0x0.0p+0, 0x1.0163da9fb3335p+0, 0x1.b61299ab8cdb7p-54, 0x1.02c9a3e778061p+0, -0x1.19083535b085dp-56,
  0x1.04315e86e7f85p+0, -0x1.0a31c1977c96ep-54, 0x1.059b0d3158574p+0, 0x1.d73e2a475b465p-55, 0x1.0706b29ddf6dep+0, -0x1.c91dfe2b13c27p-55,
  0x1.0874518759bc8p+0, 0x1.186be4bb284ffp-57, 0x1.09e3ecac6f383p+0, 0x1.1487818316136p-54, 0x1.0b5586cf9890fp+0, 0x1.8a62e4adc610bp-54,
  0x1.0cc922b7247f7p+0, 0x1.01edc16e24f71p-54, 0x1.0e3ec32d3d1a2p+0, 0x1.03a1727c57b53p-59, 0x1.0fb66affed31bp+0, -0x1.b9bedc44ebd7bp-57,
  0x1.11301d0125b51p+0, -0x1.6c51039449b3ap-54, 0x1.12abdc06c31ccp+0, -0x1.1b514b36ca5c7p-58, 0x1.1429aaea92de0p+0, -0x1.32fbf9af1369ep-54,
  0x1.15a98c8a58e51p+0, 0x1.2406ab9eeab0ap-55, 0x1.172b83c7d517bp+0, -0x1.19041b9d78a76p-55, 0x1.18af9388c8deap+0, -0x1.11023d1970f6cp-54,
  0x1.1a35beb6fcb75p+0, 0x1.e5b4c7b4968e4p-55, 0x1.1bbe084045cd4p+0, -0x1.95386352ef607p-54, 0x1.1d4873168b9aap+0, 0x1.e016e00a2643cp-54,
  0x1.1ed5022fcd91dp+0, -0x1.1df98027bb78cp-54, 0x1.2063b88628cd6p+0, 0x1.dc775814a8495p-55, 0x1.21f49917ddc96p+0, 0x1.2a97e9494a5eep-55,
  0x1.2387a6e756238p+0, 0x1.9b07eb6c70573p-54, 0x1.251ce4fb2a63fp+0, 0x1.ac155bef4f4a4p-55, 0x1.26b4565e27cddp+0, 0x1.2bd339940e9d9p-55,
  0x1.284dfe1f56381p+0, -0x1.a4c3a8c3f0d7ep-54, 0x1.29e9df51fdee1p+0, 0x1.612e8afad1255p-55, 0x1.2b87fd0dad990p+0, -0x1.10adcd6381aa4p-59,
  0x1.2d285a6e4030bp+0, 0x1.0024754db41d5p-54, 0x1.2ecafa93e2f56p+0, 0x1.1ca0f45d52383p-56, 0x1.306fe0a31b715p+0, 0x1.6f46ad23182e4p-55,
  0x1.32170fc4cd831p+0, 0x1.a9ce78e18047cp-55, 0x1.33c08b26416ffp+0, 0x1.32721843659a6p-54, 0x1.356c55f929ff1p+0, -0x1.b5cee5c4e4628p-55,
  0x1.371a7373aa9cbp+0, -0x1.63aeabf42eae2p-54, 0x1.38cae6d05d866p+0, -0x1.e958d3c9904bdp-54, 0x1.3a7db34e59ff7p+0, -0x1.5e436d661f5e3p-56,
  0x1.3c32dc313a8e5p+0, -0x1.efff8375d29c3p-54, 0x1.3dea64c123422p+0, 0x1.ada0911f09ebcp-55, 0x1.3fa4504ac801cp+0, -0x1.7d023f956f9f3p-54,
  0x1.4160a21f72e2ap+0, -0x1.ef3691c309278p-58, 0x1.431f5d950a897p+0, -0x1.1c7dde35f7999p-55, 0x1.44e086061892dp+0, 0x1.89b7a04ef80d0p-59,
  0x1.46a41ed1d0057p+0, 0x1.c944bd1648a76p-54, 0x1.486a2b5c13cd0p+0, 0x1.3c1a3b69062f0p-56, 0x1.4a32af0d7d3dep+0, 0x1.9cb62f3d1be56p-54,
  0x1.4bfdad5362a27p+0, 0x1.d4397afec42e2p-56, 0x1.4dcb299fddd0dp+0, 0x1.8ecdbbc6a7833p-54, 0x1.4f9b2769d2ca7p+0, -0x1.4b309d25957e3p-54,
  0x1.516daa2cf6642p+0, -0x1.f768569bd93efp-55, 0x1.5342b569d4f82p+0, -0x1.07abe1db13cadp-55, 0x1.551a4ca5d920fp+0, -0x1.d689cefede59bp-55,
  0x1.56f4736b527dap+0, 0x1.9bb2c011d93adp-54, 0x1.58d12d497c7fdp+0, 0x1.295e15b9a1de8p-55, 0x1.5ab07dd485429p+0, 0x1.6324c054647adp-54,
  0x1.5c9268a5946b7p+0, 0x1.c4b1b816986a2p-60, 0x1.5e76f15ad2148p+0, 0x1.ba6f93080e65ep-54, 0x1.605e1b976dc09p+0, -0x1.3e2429b56de47p-54,
  0x1.6247eb03a5585p+0, -0x1.383c17e40b497p-54, 0x1.6434634ccc320p+0, -0x1.c483c759d8933p-55, 0x1.6623882552225p+0, -0x1.bb60987591c34p-54,
  0x1.68155d44ca973p+0, 0x1.038ae44f73e65p-57, 0x1.6a09e667f3bcdp+0, -0x1.bdd3413b26456p-54, 0x1.6c012750bdabfp+0, -0x1.2895667ff0b0dp-56,
  0x1.6dfb23c651a2fp+0, -0x1.bbe3a683c88abp-57, 0x1.6ff7df9519484p+0, -0x1.83c0f25860ef6p-55, 0x1.71f75e8ec5f74p+0, -0x1.16e4786887a99p-55,
  0x1.73f9a48a58174p+0, -0x1.0a8d96c65d53cp-54, 0x1.75feb564267c9p+0, -0x1.0245957316dd3p-54, 0x1.780694fde5d3fp+0, 0x1.866b80a02162dp-54,
  0x1.7a11473eb0187p+0, -0x1.41577ee04992fp-55, 0x1.7c1ed0130c132p+0, 0x1.f124cd1164dd6p-54, 0x1.7e2f336cf4e62p+0, 0x1.05d02ba15797ep-56,
  0x1.80427543e1a12p+0, -0x1.27c86626d972bp-54, 0x1.82589994cce13p+0, -0x1.d4c1dd41532d8p-54, 0x1.8471a4623c7adp+0, -0x1.8d684a341cdfbp-55,
  0x1.868d99b4492edp+0, -0x1.fc6f89bd4f6bap-54, 0x1.88ac7d98a6699p+0, 0x1.994c2f37cb53ap-54, 0x1.8ace5422aa0dbp+0, 0x1.6e9f156864b27p-54,
  0x1.8cf3216b5448cp+0, -0x1.0d55e32e9e3aap-56, 0x1.8f1ae99157736p+0, 0x1.5cc13a2e3976cp-55, 0x1.9145b0b91ffc6p+0, -0x1.dd6792e582524p-54,
  0x1.93737b0cdc5e5p+0, -0x1.75fc781b57ebcp-57, 0x1.95a44cbc8520fp+0, -0x1.64b7c96a5f039p-56, 0x1.97d829fde4e50p+0, -0x1.d185b7c1b85d1p-54,
  0x1.9a0f170ca07bap+0, -0x1.173bd91cee632p-54, 0x1.9c49182a3f090p+0, 0x1.c7c46b071f2bep-56, 0x1.9e86319e32323p+0, 0x1.824ca78e64c6ep-56,
  0x1.a0c667b5de565p+0, -0x1.359495d1cd533p-54, 0x1.a309bec4a2d33p+0, 0x1.6305c7ddc36abp-54, 0x1.a5503b23e255dp+0, -0x1.d2f6edb8d41e1p-54,
  0x1.a799e1330b358p+0, 0x1.bcb7ecac563c7p-54, 0x1.a9e6b5579fdbfp+0, 0x1.0fac90ef7fd31p-54, 0x1.ac36bbfd3f37ap+0, -0x1.f9234cae76cd0p-55,
  0x1.ae89f995ad3adp+0, 0x1.7a1cd345dcc81p-54, 0x1.b0e07298db666p+0, -0x1.bdef54c80e425p-54, 0x1.b33a2b84f15fbp+0, -0x1.2805e3084d708p-57,
  0x1.b59728de5593ap+0, -0x1.c71dfbbba6de3p-54, 0x1.b7f76f2fb5e47p+0, -0x1.5584f7e54ac3bp-56, 0x1.ba5b030a1064ap+0, -0x1.efcd30e54292ep-54,
  0x1.bcc1e904bc1d2p+0, 0x1.23dd07a2d9e84p-55, 0x1.bf2c25bd71e09p+0, -0x1.efdca3f6b9c73p-54, 0x1.c199bdd85529cp+0, 0x1.11065895048ddp-55,
  0x1.c40ab5fffd07ap+0, 0x1.b4537e083c60ap-54, 0x1.c67f12e57d14bp+0, 0x1.2884dff483cadp-54, 0x1.c8f6d9406e7b5p+0, 0x1.1acbc48805c44p-56,
  0x1.cb720dcef9069p+0, 0x1.503cbd1e949dbp-56, 0x1.cdf0b555dc3fap+0, -0x1.dd83b53829d72p-55, 0x1.d072d4a07897cp+0, -0x1.cbc3743797a9cp-54,
  0x1.d2f87080d89f2p+0, -0x1.d487b719d8578p-54, 0x1.d5818dcfba487p+0, 0x1.2ed02d75b3707p-55, 0x1.d80e316c98398p+0, -0x1.11ec18beddfe8p-54,
  0x1.da9e603db3285p+0, 0x1.c2300696db532p-54, 0x1.dd321f301b460p+0, 0x1.2da5778f018c3p-54, 0x1.dfc97337b9b5fp+0, -0x1.1a5cd4f184b5cp-54,
  0x1.e264614f5a129p+0, -0x1.7b627817a1496p-54, 0x1.e502ee78b3ff6p+0, 0x1.39e8980a9cc8fp-55, 0x1.e7a51fbc74c83p+0, 0x1.2d522ca0c8de2p-54,
  0x1.ea4afa2a490dap+0, -0x1.e9c23179c2893p-54, 0x1.ecf482d8e67f1p+0, -0x1.c93f3b411ad8cp-54, 0x1.efa1bee615a27p+0, 0x1.dc7f486a4b6b0p-54,
  0x1.f252b376bba97p+0, 0x1.3a1a5bf0d8e43p-54, 0x1.f50765b6e4540p+0, 0x1.9d3e12dd8a18bp-54, 0x1.f7bfdad9cbe14p+0, -0x1.dbb12d006350ap-54,
  0x1.fa7c1819e90d8p+0, 0x1.74853f3a5931ep-55, 0x1.fd3c22b8f71f1p+0, 0x1.2eb74966579e7p-57
};
// Where the table is read from.  With RH_LK_LDS (set by the host for models whose row code calls rh_logit_link) every kernel copies
// it into LDS first (rh_lk_init) and the per-lane 16-byte read is a ds_read_b128: reading it from global memory shares the
// in-order vmcnt counter with the row-tile prefetches of the gradient kernels, so the first table read of a tile waits for the
// NEXT tile's rows to arrive (measured on cfg 4: 17.4 -> 21.9 ms per gradient).
#ifndef RH_LK_LDS
#define RH_LK_LDS 0
#endif
#define RH_LK_LDS_BYTES ((2 * 257 + 2 * 128 + 2) * 8)   /* 6176: both tables in LDS */
#if RH_LK_LDS
__shared__ __attribute__((aligned(16))) double rh_lk_lds[2 * 257 + 2 * 128 + 2];   // (RH_LK_LDS_BYTES: the LDS budget of the GLM kernel's tiles knows)
#define RH_LK_TAB rh_lk_lds
#define RH_EX_TAB (rh_lk_lds + 2 * 257 + 2)   /* (16-byte aligned: one ds_read_b128 per pair) */
RH_DEV void rh_lk_init() {   // at the top of a kernel, before any thread returns
  for (int i = threadIdx.x; i < 2 * 257; i += blockDim.x) rh_lk_lds[i] = rh_lk_tab[i];
  for (int i = threadIdx.x; i < 2 * 128; i += blockDim.x) rh_lk_lds[2 * 257 + 2 + i] = rh_ex_tab[i];
  __syncthreads();
}
#else
#define RH_LK_TAB rh_lk_tab
#define RH_EX_TAB rh_ex_tab
RH_DEV void rh_lk_init() {}
#endif
// low word of a double (the integer a 2^52-scale shift leaves in the mantissa)
RH_DEV int rh_lo32(const double v) { long long b; __builtin_memcpy(&b, &v, 8); return (int)b; }
RH_DEV void rh_logit_link(const double t, double &softplus, double &sigmoid) {
  const double at = __builtin_fabs(t);
  // e^{-|t|} = 2^k 2^(j/128) e^r  (round 6: table-driven; until then k = round(x / ln 2) and a degree-13 Taylor polynomial, 20 vector
  // instructions for what is 15 now).  n = round(128 x log2 e) = 128 k + j with 0 <= j < 128, r = x - n ln2 / 128 in
  // [-ln2 / 256, ln2 / 256] = +-2.7e-3: a degree-5 polynomial leaves r^6 / 720 < 6e-19; 2^(j/128) from a 128-entry (hi, lo) table
  // that sits in LDS behind the log table.  x = max(-|t|, -800): e^{-800} is exactly 0 in fp64 (v_ldexp_f64 underflows gradually
  // and correctly down to it).  Both roundings to an integer go through a shift: x 128 log2(e) + 1.5 * 2^52 has unit spacing, so the
  // fused sum IS the nearest integer -- as a double after subtracting the shift, as a two's-complement int in its low word
  // (|n| < 2^18; n ln2_hi / 128 is exact: 18 + 33 significant bits); u + 2^44 has spacing 2^-8, so its low word is round(256 u).
  const double x = __builtin_fmax(-at, -800.0);
  const double ks = __builtin_fma(x, 0x1.71547652b82fep+7, 0x1.8p+52);
  const double kf = ks - 0x1.8p+52;
  double r = __builtin_fma(kf, -0x1.62e42fee00000p-8, x);
  r = __builtin_fma(kf, -0x1.a39ef35793c76p-40, r);
  double p = 0x1.1111111111111p-7;                        // 1/5!
  p = __builtin_fma(p, r, 0x1.5555555555555p-5);          // 1/4!
  p = __builtin_fma(p, r, 0x1.5555555555555p-3);          // 1/3!
  p = __builtin_fma(p, r, 0x1p-1);
  p = __builtin_fma(p, r, 1.0);                            // (e^r - 1) / r
  const int n = rh_lo32(ks);
  const double tj = RH_EX_TAB[2 * (n & 127)], tl = RH_EX_TAB[2 * (n & 127) + 1];   // one 16-byte load
  // 2^(j/128) e^r as tj + (tj (e^r - 1) + tl): ONE rounding at full size (|e^r - 1| < 2.8e-3 carries the polynomial's own error)
  const double u = __builtin_ldexp(tj + __builtin_fma(tj, p * r, tl), n >> 7);   // (arithmetic shift: n = 128 (n >> 7) + (n & 127) for negative n too)
  const int j = rh_lo32(u + 0x1p+44);                      // round(256 u): 0 .. 256
  const double rc = RH_LK_TAB[2 * j], L = RH_LK_TAB[2 * j + 1];   // one 16-byte load
  const double r2 = __builtin_fma(u, rc, rc - 1.0);
  double q = __builtin_fma(r2, 0x1.5555555555555p-3, -0x1.999999999999ap-3);   // 1/6, -1/5
  q = __builtin_fma(q, r2, 0.25);
  q = __builtin_fma(q, r2, -0x1.5555555555555p-2);                             // -1/3
  q = __builtin_fma(q, r2, 0.5);
  const double lg = L + __builtin_fma(-(r2 * r2), q, r2);
  const double w = 1.0 + u;
  double y = __builtin_fma(rc, __builtin_fma(r2, r2, -r2), rc);
  y = __builtin_fma(y, __builtin_fma(-w, y, 1.0), y);
  // max(t, 0) spelled as a select on (t <= 0): a NaN t (a diverged chain) fails the comparison and passes through, so the VALUE
  // -- hence the energy -- is NaN and the proposal is rejected (LeapFrog.scala:138-142); a max instruction would drop it, and
  // (t + |t|) / 2 would turn t = -inf into NaN.  sigmoid of a NaN t is finite garbage that is never used.
  softplus = (t <= 0.0 ? 0.0 : t) + lg;
  sigmoid = t >= 0.0 ? y : u * y;
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
// The same butterfly for N values at once, level by level: every value goes through exactly the operations rh_wave_sum applies to
// it (bit-identical), but the N cross-lane exchanges of a level are in flight together.  One after the other they cost a
// ds_bpermute round trip per level and value -- 40 sums at the end of rh_grad_kernel's row walk were 240 serial round trips
// (~11 us of pure latency per workgroup, at the tail of the launch where nothing else hides it).
// The memory-resident lowering (RH_HEAVY: emit.cpp chunk_body, the engine's last resort for models that do not fit the register
// file): the generated functions reach their accumulators and invariants through an opaque index (rh_oz), so those arrays stay in
// scratch memory by design, and the loops over them here stay rolled -- the live state is a few elements, not the array.
#ifndef RH_HEAVY
#define RH_HEAVY 0
#endif
typedef double rh_acc_t;
#if RH_HEAVY
#define RH_UNROLL_ACC _Pragma("unroll 2")   /* loops over a target's accumulators: a few elements in flight, not all of them */
// an index the compiler cannot see through: `a[j + rh_oz()]` keeps `a` in memory (no splitting into registers), and no load of it is
// forwarded from a store, merged with another or hoisted out of a loop -- the asm is volatile and yields 0
RH_DEV int rh_oz() { int z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); return z; }
#else
#define RH_UNROLL_ACC _Pragma("unroll")
#endif
template <int N>
RH_DEV void rh_wave_sum_all(double (&v)[N]) {
#if RH_HEAVY
  // one element at a time: the same fixed-order butterfly per element, hence the same bits
  _Pragma("unroll 1") for (int i = 0; i < N; i++) {
    double x = v[i];
#pragma unroll
    for (int off = RH_LANES / 2; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
    v[i] = x;
  }
  return;
#endif
#pragma unroll
  for (int off = RH_LANES / 2; off >= 1; off >>= 1) {
    double t[N];
#pragma unroll
    for (int i = 0; i < N; i++) t[i] = __shfl_xor(v[i], off, 64);
#pragma unroll
    for (int i = 0; i < N; i++) v[i] += t[i];
  }
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
// slots whose loads are issued together before the first dependent store / add (one wavefront walks RH_SLOTS x 512 B per vector:
// with a single load in flight the walk costs a memory round trip per slot -- cfg 5: 157 of them).  The last block is masked, not
// walked slot by slot.
#ifndef RH_BIGU
#define RH_BIGU 16
#endif
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
  RH_DEV wvec &operator=(const wvec &o) { // copies the DATA (each lane its own elements); RH_BIGU loads in flight per lane
    _Pragma("unroll 1") for (int k = 0; k < RH_SLOTS; k += RH_BIGU) {
      double t[RH_BIGU];
      _Pragma("unroll") for (int j = 0; j < RH_BIGU; j++) t[j] = o.s[k + j < RH_SLOTS ? k + j : RH_SLOTS - 1];
      _Pragma("unroll") for (int j = 0; j < RH_BIGU; j++) if (k + j < RH_SLOTS) s[k + j] = t[j];
    }
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
  _Pragma("unroll 1") for (int k = 0; k < RH_SLOTS; k += RH_BIGU) {                                     \
    double xa[RH_BIGU], ya[RH_BIGU];                                                                     \
    _Pragma("unroll") for (int j = 0; j < RH_BIGU; j++) {                                                \
      const int kc = k + j < RH_SLOTS ? k + j : RH_SLOTS - 1;                                            \
      xa[j] = x.s[kc]; ya[j] = y.s[kc];                                                                  \
    }                                                                                                    \
    _Pragma("unroll") for (int j = 0; j < RH_BIGU; j++) {                                                \
      const double xv = xa[j], yv = ya[j]; (void)xv; (void)yv;                                           \
      if (k + j < RH_SLOTS) o.s[k + j] = (expr);                                                         \
    }                                                                                                    \
  }
// y(i) += a * x(i)   -- multiply, round, add, round (LeapFrog.scala:148,170)
RH_DEV void wv_axpy(wvec &y, double a, const wvec &x) { wvec &o = y; RH_BIG2(yv + a * xv) }
RH_DEV void wv_mul(wvec &o, const wvec &x, const wvec &y) { RH_BIG2(xv * yv) }
RH_DEV void wv_sub(wvec &o, const wvec &x, const wvec &y) { RH_BIG2(xv - yv) }
RH_DEV void wv_acc(wvec &y, const wvec &x) { wvec &o = y; RH_BIG2(yv + xv) }   // y(i) += x(i)
#define RH_BIG3(expr)                                                                                  \
  _Pragma("unroll 1") for (int k = 0; k < RH_SLOTS; k += RH_BIGU) {                                     \
    double xa[RH_BIGU], ya[RH_BIGU], za[RH_BIGU];                                                        \
    _Pragma("unroll") for (int j = 0; j < RH_BIGU; j++) {                                                \
      const int kc = k + j < RH_SLOTS ? k + j : RH_SLOTS - 1;                                            \
      xa[j] = x.s[kc]; ya[j] = y.s[kc]; za[j] = z.s[kc];                                                 \
    }                                                                                                    \
    _Pragma("unroll") for (int j = 0; j < RH_BIGU; j++) {                                                \
      const double xv = xa[j], yv = ya[j], zv = za[j];                                                   \
      if (k + j < RH_SLOTS) o.s[k + j] = (expr);                                                         \
    }                                                                                                    \
  }
RH_DEV void wv_sub_add(wvec &o, const wvec &x, const wvec &y, const wvec &z) { RH_BIG3(xv - yv + zv) }          // o = x - y + z
RH_DEV void wv_nuts_adj(wvec &o, const wvec &x, const wvec &y, const wvec &z) { RH_BIG3(xv - (yv + zv) / 2.0) }  // o = x - (y + z) / 2
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
RH_DEV void wv_acc(wvec &y, const wvec &x) {   // y(i) += x(i)
  RH_UNROLL_SLOTS
  for (int k = 0; k < RH_SLOTS; k++) y.s[k] += x.s[k];
}
RH_DEV void wv_sub_add(wvec &o, const wvec &x, const wvec &y, const wvec &z) {   // o = x - y + z
  RH_UNROLL_SLOTS
  for (int k = 0; k < RH_SLOTS; k++) o.s[k] = x.s[k] - y.s[k] + z.s[k];
}
RH_DEV void wv_nuts_adj(wvec &o, const wvec &x, const wvec &y, const wvec &z) {  // o = x - (y + z) / 2
  RH_UNROLL_SLOTS
  for (int k = 0; k < RH_SLOTS; k++) o.s[k] = x.s[k] - (y.s[k] + z.s[k]) / 2.0;
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
  _Pragma("unroll 1") for (int k = 0; k < RH_SLOTS; k += RH_BIGU) {
    double t[RH_BIGU];
    _Pragma("unroll") for (int j = 0; j < RH_BIGU; j++) t[j] = x.s[k + j < RH_SLOTS ? k + j : RH_SLOTS - 1];
    _Pragma("unroll") for (int j = 0; j < RH_BIGU; j++) part += ((k + j) * 64 + (int)threadIdx.x < RH_NVARS) ? t[j] : 0.0;   // (k + j >= RH_SLOTS: past RH_NVARS)
  }
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
