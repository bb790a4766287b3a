/* rainier_hip_jni.c -- thin JNI shim over include/rainier_hip.h.
 *
 * A Rainier maintainer builds it next to librainier_hip.so with
 *     cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *        rainier_hip_jni.c -L.. -lrainier_hip -o librainier_hip_jni.so
 * The build image of this repository has no JDK, so here the shim is compiled against tests/stubs/jni.h (the JNI
 * specification's signatures for the calls it uses) and EXECUTED against tests/stubs/fake_jni.c, a function table that
 * implements those calls over malloc'ed arrays with the JVM's copy-in / copy-back / JNI_ABORT semantics
 * (tests/test_jni_shim.py: CPU argument marshalling, `-m gpu` end-to-end draws bit-identical to the ctypes path).
 *
 * The Scala side that binds these functions is integration/scala/Native.scala (object com.stripe.rainier.hip.Native).
 * Every function pins/copies the Java arrays, calls the C ABI and maps a non-zero status to an exception -- no logic
 * lives here.  The flat int/double arrays that carry rh_compile_opts and rh_config are described by the X-macro lists
 * below; integration/scala/HipModel.scala builds them from locals with THE SAME NAMES IN THE SAME ORDER, which
 * tests/test_jni_shim.py checks textually.
 */
#include <jni.h>
#include <stdlib.h>
#include <string.h>

#include "rainier_hip.h"

/* int[] copts -> rh_compile_opts (every field after struct_size, in declaration order) */
#define RH_JNI_COPTS(X) X(device) X(math_mode) X(fp_contract) X(rows_unroll) X(grad_chains) X(grad_unroll) X(factor_outputs) X(with_nuts)
/* int[] icfg / double[] dcfg -> rh_config (every scalar field; the two pointer fields travel as their own arrays) */
#define RH_JNI_ICFG(X) \
  X(iterations) X(warmup) X(sampler) X(hmc_steps) X(ehmc_max_steps) X(ehmc_min_steps) X(ehmc_buf_size) X(step_tuner) \
  X(mass_tuner) X(mass_init_window) X(mass_skip_first) X(mass_skip_last) X(nuts_max_depth) X(engine) X(grad_splits)
#define RH_JNI_DCFG(X) X(ehmc_p_count) X(dualavg_delta) X(static_step) X(mass_expansion)
#define RH_JNI_COUNT(n) +1
enum { RH_JNI_NCOPTS = 0 RH_JNI_COPTS(RH_JNI_COUNT), RH_JNI_NICFG = 0 RH_JNI_ICFG(RH_JNI_COUNT), RH_JNI_NDCFG = 0 RH_JNI_DCFG(RH_JNI_COUNT) };
/* double[] stats: RH_JNI_NSTATS doubles per chain */
#define RH_JNI_STATS(X) X(leapfrog_steps) X(warmup_leapfrog_steps) X(gradient_evaluations) X(accepted) X(mean_accept_prob) X(step_size) X(bfmi)
enum { RH_JNI_NSTATS = 0 RH_JNI_STATS(RH_JNI_COUNT) };

static void throw_msg(JNIEnv *env, int rc, const char *msg) {
  jclass cls = (*env)->FindClass(env, rc == RH_E_INVALID ? "java/lang/IllegalArgumentException"
                                      : rc == RH_E_UNSUPPORTED ? "java/lang/UnsupportedOperationException" : "java/lang/RuntimeException");
  (*env)->ThrowNew(env, cls, msg && *msg ? msg : "rainier-hip error");
}
static void throw_rh(JNIEnv *env, rh_model *m, int rc) { throw_msg(env, rc, rh_last_error(m)); }

/* int abiVersion() / int deviceCount() */
JNIEXPORT jint JNICALL Java_com_stripe_rainier_hip_Native_00024_abiVersion(JNIEnv *env, jobject self) { (void)env; (void)self; return rh_abi_version(); }
JNIEXPORT jint JNICALL Java_com_stripe_rainier_hip_Native_00024_deviceCount(JNIEnv *env, jobject self) { (void)env; (void)self; return rh_device_count(); }

/* long modelCreate(byte[] rir, double[][] columns, long[] nrows, int[] copts) */
JNIEXPORT jlong JNICALL Java_com_stripe_rainier_hip_Native_00024_modelCreate(
    JNIEnv *env, jobject self, jbyteArray rir, jobjectArray columns, jlongArray nrows, jintArray copts) {
  (void)self;
  if ((*env)->GetArrayLength(env, copts) != RH_JNI_NCOPTS) { throw_msg(env, RH_E_INVALID, "copts has the wrong length"); return 0; }
  const jsize ncols = (*env)->GetArrayLength(env, columns);
  const double **cols = (const double **)calloc(ncols ? ncols : 1, sizeof(double *));
  jdoubleArray *arrs = (jdoubleArray *)calloc(ncols ? ncols : 1, sizeof(jdoubleArray));
  for (jsize i = 0; i < ncols; i++) {
    arrs[i] = (jdoubleArray)(*env)->GetObjectArrayElement(env, columns, i);
    cols[i] = (*env)->GetDoubleArrayElements(env, arrs[i], NULL);
  }
  jbyte *blob = (*env)->GetByteArrayElements(env, rir, NULL);
  jlong *rows = (*env)->GetLongArrayElements(env, nrows, NULL);
  jint *co = (*env)->GetIntArrayElements(env, copts, NULL);
  rh_compile_opts opts;
  memset(&opts, 0, sizeof opts);
  opts.struct_size = (int32_t)sizeof opts;
  { int i = 0;
#define X(f) opts.f = co[i++];
    RH_JNI_COPTS(X)
#undef X
  }
  rh_model *m = NULL;
  const int rc = rh_model_create(blob, (size_t)(*env)->GetArrayLength(env, rir), cols, (const int64_t *)rows, &opts, &m);
  (*env)->ReleaseIntArrayElements(env, copts, co, JNI_ABORT);
  (*env)->ReleaseLongArrayElements(env, nrows, rows, JNI_ABORT);
  (*env)->ReleaseByteArrayElements(env, rir, blob, JNI_ABORT);
  for (jsize i = 0; i < ncols; i++) (*env)->ReleaseDoubleArrayElements(env, arrs[i], (jdouble *)cols[i], JNI_ABORT);
  free(arrs); free(cols);
  if (rc != RH_OK) { throw_rh(env, NULL, rc); return 0; }
  return (jlong)(intptr_t)m;
}

/* long modelClone(long model, int device): the same compiled model on another device (rh_model_clone) */
JNIEXPORT jlong JNICALL Java_com_stripe_rainier_hip_Native_00024_modelClone(JNIEnv *env, jobject self, jlong h, jint device) {
  (void)self;
  rh_model *m = NULL;
  const int rc = rh_model_clone((const rh_model *)(intptr_t)h, device, &m);
  if (rc != RH_OK) { throw_rh(env, NULL, rc); return 0; }
  return (jlong)(intptr_t)m;
}

/* int modelEngines(long model): which engines the model can use on this toolchain (rh_model_engines, ABI 5):
 * bit 0 chain engine, bit 1 tick engine, bit 2 density seam; bits 8.. = code objects built or fetched while lowering it */
JNIEXPORT jint JNICALL Java_com_stripe_rainier_hip_Native_00024_modelEngines(JNIEnv *env, jobject self, jlong h) {
  (void)self;
  int32_t chain = 0, tick = 0, density = 0, attempts = 0;
  const int rc = rh_model_engines((const rh_model *)(intptr_t)h, &chain, &tick, &density, &attempts, NULL, 0);
  if (rc != RH_OK) { throw_rh(env, NULL, rc); return 0; }
  return (jint)((chain ? 1 : 0) | (tick ? 2 : 0) | (density ? 4 : 0) | (attempts << 8));
}

JNIEXPORT void JNICALL Java_com_stripe_rainier_hip_Native_00024_modelDestroy(JNIEnv *env, jobject self, jlong h) {
  (void)env; (void)self;
  rh_model_destroy((rh_model *)(intptr_t)h);
}

JNIEXPORT jint JNICALL Java_com_stripe_rainier_hip_Native_00024_modelNVars(JNIEnv *env, jobject self, jlong h) {
  (void)env; (void)self;
  return rh_model_nvars((const rh_model *)(intptr_t)h);
}

/* void densityEval(long model, double[] q, int chains, int engine, int gradSplits, double[] logp, double[] grad) */
JNIEXPORT void JNICALL Java_com_stripe_rainier_hip_Native_00024_densityEval(
    JNIEnv *env, jobject self, jlong h, jdoubleArray q, jint chains, jint engine, jint gradSplits, jdoubleArray logp,
    jdoubleArray grad) {
  (void)self;
  rh_model *m = (rh_model *)(intptr_t)h;
  jdouble *qp = (*env)->GetDoubleArrayElements(env, q, NULL);
  jdouble *lp = (*env)->GetDoubleArrayElements(env, logp, NULL);
  jdouble *gp = (*env)->GetDoubleArrayElements(env, grad, NULL);
  const int rc = rh_density_eval_ex(m, qp, chains, engine, gradSplits, lp, gp);
  (*env)->ReleaseDoubleArrayElements(env, q, qp, JNI_ABORT);
  (*env)->ReleaseDoubleArrayElements(env, logp, lp, 0);
  (*env)->ReleaseDoubleArrayElements(env, grad, gp, 0);
  if (rc != RH_OK) throw_rh(env, m, rc);
}

/* void optimize(long model, double[] x0 (null = zeros), int starts, int maxEvals, double[] x, int[] evals, int[] status) */
JNIEXPORT void JNICALL Java_com_stripe_rainier_hip_Native_00024_optimize(
    JNIEnv *env, jobject self, jlong h, jdoubleArray x0, jint starts, jint maxEvals, jdoubleArray x, jintArray evals,
    jintArray status) {
  (void)self;
  rh_model *m = (rh_model *)(intptr_t)h;
  jdouble *x0p = x0 ? (*env)->GetDoubleArrayElements(env, x0, NULL) : NULL;
  jdouble *xp = (*env)->GetDoubleArrayElements(env, x, NULL);
  jint *ep = (*env)->GetIntArrayElements(env, evals, NULL);
  jint *sp = (*env)->GetIntArrayElements(env, status, NULL);
  const int rc = rh_optimize(m, x0p, starts, maxEvals, xp, (int32_t *)ep, (int32_t *)sp);
  if (x0p) (*env)->ReleaseDoubleArrayElements(env, x0, x0p, JNI_ABORT);
  (*env)->ReleaseDoubleArrayElements(env, x, xp, 0);
  (*env)->ReleaseIntArrayElements(env, evals, ep, 0);
  (*env)->ReleaseIntArrayElements(env, status, sp, 0);
  if (rc != RH_OK) throw_rh(env, m, rc);
}

/* void sample(long[] models, int[] icfg, double[] dcfg, double[] staticMass (null), double[] rngNextGaussian (null),
 *             long[] seeds, double[] draws, double[] mass, double[] stats (null))
 * models: one handle per device (the same program compiled with copts.device = g); one handle = rh_sample, several =
 * rh_sample_multi (chains cut into contiguous shards by global chain id; the result does not depend on the count). */
JNIEXPORT void JNICALL Java_com_stripe_rainier_hip_Native_00024_sample(
    JNIEnv *env, jobject self, jlongArray models, jintArray icfg, jdoubleArray dcfg, jdoubleArray staticMass,
    jdoubleArray rngNextGaussian, jlongArray seeds, jdoubleArray draws, jdoubleArray mass, jdoubleArray stats) {
  (void)self;
  const jsize nmodels = (*env)->GetArrayLength(env, models);
  if (nmodels < 1 || (*env)->GetArrayLength(env, icfg) != RH_JNI_NICFG || (*env)->GetArrayLength(env, dcfg) != RH_JNI_NDCFG) {
    throw_msg(env, RH_E_INVALID, "models / icfg / dcfg have the wrong length");
    return;
  }
  const jsize chains = (*env)->GetArrayLength(env, seeds);
  if (stats && (*env)->GetArrayLength(env, stats) != chains * RH_JNI_NSTATS) { throw_msg(env, RH_E_INVALID, "stats has the wrong length"); return; }
  if (rngNextGaussian && (*env)->GetArrayLength(env, rngNextGaussian) != chains) { throw_msg(env, RH_E_INVALID, "rngNextGaussian has the wrong length"); return; }
  jlong *mh = (*env)->GetLongArrayElements(env, models, NULL);
  rh_model **ms = (rh_model **)calloc((size_t)nmodels, sizeof(rh_model *));
  for (jsize g = 0; g < nmodels; g++) ms[g] = (rh_model *)(intptr_t)mh[g];
  (*env)->ReleaseLongArrayElements(env, models, mh, JNI_ABORT);
  jint *ic = (*env)->GetIntArrayElements(env, icfg, NULL);
  jdouble *dc = (*env)->GetDoubleArrayElements(env, dcfg, NULL);
  rh_config cfg;
  rh_config_default(&cfg);
  { int i = 0;
#define X(f) cfg.f = ic[i++];
    RH_JNI_ICFG(X)
#undef X
    i = 0;
#define X(f) cfg.f = dc[i++];
    RH_JNI_DCFG(X)
#undef X
  }
  { /* a wrongly sized Java array must become an exception, not a native heap overflow: draws [chains][iterations][nvars],
     * mass [chains][nvars] (rh_sample's contract), staticMass [nvars] */
    const int nv = ms[0] ? rh_model_nvars(ms[0]) : -1;
    const long long want_draws = (long long)chains * (long long)cfg.iterations * (long long)nv;
    const char *bad = NULL;
    if (nv <= 0) bad = "models[0] is not a model handle";
    else if (cfg.iterations < 0 || (long long)(*env)->GetArrayLength(env, draws) != want_draws) bad = "draws must hold chains * iterations * nvars doubles";
    else if ((long long)(*env)->GetArrayLength(env, mass) != (long long)chains * nv) bad = "mass must hold chains * nvars doubles";
    else if (staticMass && (*env)->GetArrayLength(env, staticMass) != nv) bad = "staticMass must hold nvars doubles";
    if (bad) {
      (*env)->ReleaseDoubleArrayElements(env, dcfg, dc, JNI_ABORT);
      (*env)->ReleaseIntArrayElements(env, icfg, ic, JNI_ABORT);
      free(ms);
      throw_msg(env, RH_E_INVALID, bad);
      return;
    }
  }
  jdouble *sm = staticMass ? (*env)->GetDoubleArrayElements(env, staticMass, NULL) : NULL;
  jdouble *nn = rngNextGaussian ? (*env)->GetDoubleArrayElements(env, rngNextGaussian, NULL) : NULL;
  cfg.static_mass = sm;
  cfg.rng_next_gaussian = nn;
  jlong *sd = (*env)->GetLongArrayElements(env, seeds, NULL);
  jdouble *dr = (*env)->GetDoubleArrayElements(env, draws, NULL);
  jdouble *mp = (*env)->GetDoubleArrayElements(env, mass, NULL);
  rh_chain_stats *st = (rh_chain_stats *)calloc(chains ? (size_t)chains : 1, sizeof(rh_chain_stats));
  const int rc = nmodels == 1 ? rh_sample(ms[0], &cfg, (const int64_t *)sd, chains, dr, mp, st)
                              : rh_sample_multi(ms, nmodels, &cfg, (const int64_t *)sd, chains, dr, mp, st);
  if (rc == RH_OK && stats) {
    jdouble *sp = (*env)->GetDoubleArrayElements(env, stats, NULL);
    for (jsize c = 0; c < chains; c++) {
      int i = 0;
#define X(f) sp[RH_JNI_NSTATS * c + (i++)] = (double)st[c].f;
      RH_JNI_STATS(X)
#undef X
    }
    (*env)->ReleaseDoubleArrayElements(env, stats, sp, 0);
  }
  free(st);
  (*env)->ReleaseDoubleArrayElements(env, mass, mp, rc == RH_OK ? 0 : JNI_ABORT);
  (*env)->ReleaseDoubleArrayElements(env, draws, dr, rc == RH_OK ? 0 : JNI_ABORT);
  (*env)->ReleaseLongArrayElements(env, seeds, sd, JNI_ABORT);
  if (nn) (*env)->ReleaseDoubleArrayElements(env, rngNextGaussian, nn, JNI_ABORT);
  if (sm) (*env)->ReleaseDoubleArrayElements(env, staticMass, sm, JNI_ABORT);
  (*env)->ReleaseDoubleArrayElements(env, dcfg, dc, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, icfg, ic, JNI_ABORT);
  /* rh_sample_multi reports a shard's failure through the calling thread's error slot (rh_last_error(NULL)), not through models[0] */
  if (rc != RH_OK) throw_rh(env, nmodels == 1 ? ms[0] : NULL, rc);
  free(ms);
}

/* void requirementsEval(byte[] rir, int[] copts, double[] draws, long ndraws, double[] out)   (Trace.predict's compiled part) */
JNIEXPORT void JNICALL Java_com_stripe_rainier_hip_Native_00024_requirementsEval(
    JNIEnv *env, jobject self, jbyteArray rir, jintArray copts, jdoubleArray draws, jlong ndraws, jdoubleArray out) {
  (void)self;
  if ((*env)->GetArrayLength(env, copts) != RH_JNI_NCOPTS) { throw_msg(env, RH_E_INVALID, "copts has the wrong length"); return; }
  jbyte *blob = (*env)->GetByteArrayElements(env, rir, NULL);
  jint *co = (*env)->GetIntArrayElements(env, copts, NULL);
  rh_compile_opts opts;
  memset(&opts, 0, sizeof opts);
  opts.struct_size = (int32_t)sizeof opts;
  { int i = 0;
#define X(f) opts.f = co[i++];
    RH_JNI_COPTS(X)
#undef X
  }
  jdouble *dp = (*env)->GetDoubleArrayElements(env, draws, NULL);
  jdouble *op = (*env)->GetDoubleArrayElements(env, out, NULL);
  const int rc = rh_requirements_eval(blob, (size_t)(*env)->GetArrayLength(env, rir), &opts, dp, ndraws, op);
  (*env)->ReleaseDoubleArrayElements(env, out, op, rc == RH_OK ? 0 : JNI_ABORT);
  (*env)->ReleaseDoubleArrayElements(env, draws, dp, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, copts, co, JNI_ABORT);
  (*env)->ReleaseByteArrayElements(env, rir, blob, JNI_ABORT);
  if (rc != RH_OK) throw_rh(env, NULL, rc);
}

/* void diagnostics(double[] draws, int chains, int iterations, int nvars, double[] rhat, double[] ess)   (Trace.diagnostics) */
JNIEXPORT void JNICALL Java_com_stripe_rainier_hip_Native_00024_diagnostics(
    JNIEnv *env, jobject self, jdoubleArray draws, jint chains, jint iterations, jint nvars, jdoubleArray rhat, jdoubleArray ess) {
  (void)self;
  jdouble *dp = (*env)->GetDoubleArrayElements(env, draws, NULL);
  jdouble *rp = (*env)->GetDoubleArrayElements(env, rhat, NULL);
  jdouble *ep = (*env)->GetDoubleArrayElements(env, ess, NULL);
  const int rc = rh_diagnostics(dp, chains, iterations, nvars, rp, ep);
  (*env)->ReleaseDoubleArrayElements(env, ess, ep, 0);
  (*env)->ReleaseDoubleArrayElements(env, rhat, rp, 0);
  (*env)->ReleaseDoubleArrayElements(env, draws, dp, JNI_ABORT);
  if (rc != RH_OK) throw_rh(env, NULL, rc);
}
