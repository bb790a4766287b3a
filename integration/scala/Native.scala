// integration/scala/Native.scala -- NOT BUILT HERE (no JDK in the image).  JNI object bound by
// rainier_amd/jni/rainier_hip_jni.c (Java_com_stripe_rainier_hip_Native_00024_*); every method is one C-ABI call of
// include/rainier_hip.h.  The shim itself IS compiled and executed in this repository, against a JVM-less JNIEnv
// (tests/stubs/fake_jni.c, tests/test_jni_shim.py); the signatures below are checked against it textually.
package com.stripe.rainier.hip

object Native {
  System.loadLibrary("rainier_hip_jni")
  @native def abiVersion(): Int                                                                   // rh_abi_version
  @native def deviceCount(): Int                                                                  // rh_device_count
  /** copts = rh_compile_opts field by field: see HipOptions.copts */
  @native def modelCreate(rir: Array[Byte], columns: Array[Array[Double]], nrows: Array[Long],
                          copts: Array[Int]): Long                                                // rh_model_create
  /** the same compiled model on another device: no second lowering, columns copied device to device */
  @native def modelClone(model: Long, device: Int): Long                                          // rh_model_clone
  @native def modelEngines(model: Long): Int                                                      // rh_model_engines: bit 0 chain, 1 tick, 2 density; >> 8 = compile attempts
  @native def modelDestroy(model: Long): Unit                                                     // rh_model_destroy
  @native def modelNVars(model: Long): Int                                                        // rh_model_nvars
  /** engine / gradSplits: 0, 0 = the engine's choice (rh_density_eval) */
  @native def densityEval(model: Long, q: Array[Double], chains: Int, engine: Int, gradSplits: Int,
                          logp: Array[Double], grad: Array[Double]): Unit                         // rh_density_eval_ex
  @native def optimize(model: Long, x0: Array[Double], starts: Int, maxEvals: Int,
                       x: Array[Double], evals: Array[Int], status: Array[Int]): Unit             // rh_optimize
  /** models: one handle per device (one = rh_sample, several = rh_sample_multi); icfg/dcfg = rh_config field by field
    * (HipConfig.flatten); staticMass, rngNextGaussian and stats may be null */
  @native def sample(models: Array[Long], icfg: Array[Int], dcfg: Array[Double], staticMass: Array[Double],
                     rngNextGaussian: Array[Double], seeds: Array[Long], draws: Array[Double], mass: Array[Double],
                     stats: Array[Double]): Unit                                                  // rh_sample / rh_sample_multi
  @native def requirementsEval(rir: Array[Byte], copts: Array[Int], draws: Array[Double], ndraws: Long,
                               out: Array[Double]): Unit                                          // rh_requirements_eval
  @native def diagnostics(draws: Array[Double], chains: Int, iterations: Int, nvars: Int,
                          rhat: Array[Double], ess: Array[Double]): Unit                          // rh_diagnostics
}
