/* rainier_hip_jni.c -- thin JNI shim over include/rainier_hip.h.
 *
 * NOT COMPILED IN THIS REPOSITORY'S BUILD: the build image has no JDK (no jni.h).  A Rainier maintainer builds it
 * next to librainier_hip.so with
 *     cc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -I../../include \
 *        rainier_hip_jni.c -L.. -lrainier_hip -o librainier_hip_jni.so
 * The Scala side that binds these functions is shown in INTEGRATION.md
 * (object com.stripe.rainier.hip.Native).  Every function pins/copies the Java arrays, calls the C ABI and maps a
 * non-zero status to RuntimeException -- no logic lives here.
 */
#include <jni.h>
#include <stdlib.h>
#include <string.h>

#include "rainier_hip.h"

static void throw_rh(JNIEnv *env, rh_model *m, int rc) {
  const char *msg = rh_last_error(m);
  jclass cls = (*env)->FindClass(env, rc == RH_E_INVALID ? "java/lang/IllegalArgumentException" : "java/lang/RuntimeException");
  (*env)->ThrowNew(env, cls, msg && *msg ? msg : "rainier-hip error");
}

/* long modelCreate(byte[] rir, double[][] columns, long[] nrows, int device, int mathMode, boolean fpContract) */
JNIEXPORT jlong JNICALL Java_com_stripe_rainier_hip_Native_00024_modelCreate(
    JNIEnv *env, jobject self, jbyteArray rir, jobjectArray columns, jlongArray nrows, jint device, jint mathMode,
    jboolean fpContract) {
  (void)self;
  const jsize ncols = (*env)->GetArrayLength(env, columns);
  const double **cols = (const double **)calloc(ncols ? ncols : 1, sizeof(double *));
  jdoubleArray *arrs = (jdoubleArray *)calloc(ncols ? ncols : 1, sizeof(jdoubleArray));
  for (jsize i = 0; i < ncols; i++) {
    arrs[i] = (jdoubleArray)(*env)->GetObjectArrayElement(env, columns, i);
    cols[i] = (*env)->GetDoubleArrayElements(env, arrs[i], NULL);
  }
  jbyte *blob = (*env)->GetByteArrayElements(env, rir, NULL);
  jlong *rows = (*env)->GetLongArrayElements(env, nrows, NULL);
  rh_compile_opts opts;
  memset(&opts, 0, sizeof opts);
  opts.struct_size = (int32_t)sizeof opts; opts.device = device; opts.math_mode = mathMode; opts.fp_contract = fpContract ? 1 : 0;
  rh_model *m = NULL;
  const int rc = rh_model_create(blob, (size_t)(*env)->GetArrayLength(env, rir), cols, (const int64_t *)rows, &opts, &m);
  (*env)->ReleaseLongArrayElements(env, nrows, rows, JNI_ABORT);
  (*env)->ReleaseByteArrayElements(env, rir, blob, JNI_ABORT);
  for (jsize i = 0; i < ncols; i++) (*env)->ReleaseDoubleArrayElements(env, arrs[i], (jdouble *)cols[i], JNI_ABORT);
  free(arrs); free(cols);
  if (rc != RH_OK) { throw_rh(env, NULL, rc); return 0; }
  return (jlong)(intptr_t)m;
}

JNIEXPORT void JNICALL Java_com_stripe_rainier_hip_Native_00024_modelDestroy(JNIEnv *env, jobject self, jlong h) {
  (void)env; (void)self;
  rh_model_destroy((rh_model *)(intptr_t)h);
}

/* void densityEval(long model, double[] q, int chains, double[] logp, double[] grad) */
JNIEXPORT void JNICALL Java_com_stripe_rainier_hip_Native_00024_densityEval(
    JNIEnv *env, jobject self, jlong h, jdoubleArray q, jint chains, jdoubleArray logp, jdoubleArray grad) {
  (void)self;
  rh_model *m = (rh_model *)(intptr_t)h;
  jdouble *qp = (*env)->GetDoubleArrayElements(env, q, NULL);
  jdouble *lp = (*env)->GetDoubleArrayElements(env, logp, NULL);
  jdouble *gp = (*env)->GetDoubleArrayElements(env, grad, NULL);
  const int rc = rh_density_eval(m, qp, chains, lp, gp);
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

/* void sample(long model, int[] icfg, double[] dcfg, double[] staticMass, long[] seeds, double[] draws, double[] mass,
 *             double[] stats)   -- icfg/dcfg carry rh_config field by field (see Native.scala in INTEGRATION.md) */
JNIEXPORT void JNICALL Java_com_stripe_rainier_hip_Native_00024_sample(
    JNIEnv *env, jobject self, jlong h, jintArray icfg, jdoubleArray dcfg, jdoubleArray staticMass, jlongArray seeds,
    jdoubleArray draws, jdoubleArray mass, jdoubleArray stats) {
  (void)self;
  rh_model *m = (rh_model *)(intptr_t)h;
  jint *ic = (*env)->GetIntArrayElements(env, icfg, NULL);
  jdouble *dc = (*env)->GetDoubleArrayElements(env, dcfg, NULL);
  rh_config cfg;
  rh_config_default(&cfg);
  cfg.iterations = ic[0]; cfg.warmup = ic[1]; cfg.sampler = ic[2]; cfg.hmc_steps = ic[3];
  cfg.ehmc_max_steps = ic[4]; cfg.ehmc_min_steps = ic[5]; cfg.ehmc_buf_size = ic[6]; cfg.step_tuner = ic[7];
  cfg.mass_tuner = ic[8]; cfg.mass_init_window = ic[9]; cfg.mass_skip_first = ic[10]; cfg.mass_skip_last = ic[11];
  cfg.ehmc_p_count = dc[0]; cfg.dualavg_delta = dc[1]; cfg.static_step = dc[2]; cfg.mass_expansion = dc[3];
  jdouble *sm = staticMass ? (*env)->GetDoubleArrayElements(env, staticMass, NULL) : NULL;
  cfg.static_mass = sm;
  const jsize chains = (*env)->GetArrayLength(env, seeds);
  jlong *sd = (*env)->GetLongArrayElements(env, seeds, NULL);
  jdouble *dr = (*env)->GetDoubleArrayElements(env, draws, NULL);
  jdouble *ms = (*env)->GetDoubleArrayElements(env, mass, NULL);
  rh_chain_stats *st = (rh_chain_stats *)calloc(chains, sizeof(rh_chain_stats));
  const int rc = rh_sample(m, &cfg, (const int64_t *)sd, chains, dr, ms, st);
  if (rc == RH_OK && stats) { /* 7 doubles per chain: steps, warmupSteps, gradEvals, accepted, meanAccept, stepSize, bfmi */
    jdouble *sp = (*env)->GetDoubleArrayElements(env, stats, NULL);
    for (jsize c = 0; c < chains; c++) {
      sp[7 * c + 0] = (double)st[c].leapfrog_steps; sp[7 * c + 1] = (double)st[c].warmup_leapfrog_steps;
      sp[7 * c + 2] = (double)st[c].gradient_evaluations; sp[7 * c + 3] = (double)st[c].accepted;
      sp[7 * c + 4] = st[c].mean_accept_prob; sp[7 * c + 5] = st[c].step_size; sp[7 * c + 6] = st[c].bfmi;
    }
    (*env)->ReleaseDoubleArrayElements(env, stats, sp, 0);
  }
  free(st);
  (*env)->ReleaseDoubleArrayElements(env, mass, ms, 0);
  (*env)->ReleaseDoubleArrayElements(env, draws, dr, 0);
  (*env)->ReleaseLongArrayElements(env, seeds, sd, JNI_ABORT);
  if (sm) (*env)->ReleaseDoubleArrayElements(env, staticMass, sm, JNI_ABORT);
  (*env)->ReleaseDoubleArrayElements(env, dcfg, dc, JNI_ABORT);
  (*env)->ReleaseIntArrayElements(env, icfg, ic, JNI_ABORT);
  if (rc != RH_OK) throw_rh(env, m, rc);
}
