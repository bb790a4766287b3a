/* oracle/jmath.h -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
 *
 * CPU restatement of the JDK arithmetic the reference reaches through
 * scala.util.Random / java.lang.StrictMath (rainier-sampler/.../sampler/RNG.scala:20-26).
 * The JDK is a third-party dependency that is NOT under /root/reference: CI pins JDK 11
 * (.github/workflows/ci.yml:12-16).  java.util.Random is specified bit-for-bit by its Javadoc;
 * StrictMath.log / StrictMath.exp are specified to be fdlibm 5.3 (e_log.c / e_exp.c), whose
 * published algorithm is restated here.  StrictMath.sqrt is the correctly rounded IEEE sqrt.
 */
#ifndef ORACLE_JMATH_H
#define ORACLE_JMATH_H
#include <stdint.h>

double jm_strict_log(double x);  /* fdlibm __ieee754_log  == java.lang.StrictMath.log  */
double jm_strict_exp(double x);  /* fdlibm __ieee754_exp  == java.lang.StrictMath.exp  */

/* "Math backend" used for the calls the reference makes to java.lang.Math (HotSpot
 * intrinsics, <=1 ulp, not bit-specified):  S/DualAvg.scala:33-34,55-56,59,63,72 and
 * S/LeapFrog.scala:67,70,78,140.
 *   JM_LIBM : glibc exp/log/pow/sqrt   (the faithful stand-in for java.lang.Math)
 *   JM_DET  : fdlibm exp/log + sqrt-composed t^-0.75 (bit-reproducible on any IEEE machine;
 *             the mode the HIP engine implements, so chains can be compared bit-for-bit) */
enum { JM_LIBM = 0, JM_DET = 1 };
double jm_exp(int mode, double x);
double jm_log(int mode, double x);
double jm_sqrt(double x);
double jm_pow_neg075(int mode, double t); /* Math.pow(t, -0.75) with t a positive integer value */

/* java.util.Random (JDK Javadoc algorithm) */
typedef struct {
  uint64_t seed;
  int have_next;
  double next_next;
} jrandom;
void jrandom_init(jrandom *r, int64_t seed);
int32_t jrandom_next(jrandom *r, int bits);
int32_t jrandom_next_int(jrandom *r);
double jrandom_next_double(jrandom *r);
double jrandom_next_gaussian(jrandom *r);
#endif
