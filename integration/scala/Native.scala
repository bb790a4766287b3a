// integration/scala/Native.scala -- NOT BUILT HERE.  JNI object bound by rainier_amd/jni/rainier_hip_jni.c
// (Java_com_stripe_rainier_hip_Native_00024_*); every method is one C-ABI call of include/rainier_hip.h.
package com.stripe.rainier.hip

object Native {
  System.loadLibrary("rainier_hip_jni")
  @native def modelCreate(rir: Array[Byte], columns: Array[Array[Double]], nrows: Array[Long],
                          device: Int, mathMode: Int, fpContract: Boolean): Long                 // rh_model_create
  @native def modelDestroy(model: Long): Unit                                                     // rh_model_destroy
  @native def densityEval(model: Long, q: Array[Double], chains: Int,
                          logp: Array[Double], grad: Array[Double]): Unit                         // rh_density_eval
  @native def optimize(model: Long, x0: Array[Double], starts: Int, maxEvals: Int,
                       x: Array[Double], evals: Array[Int], status: Array[Int]): Unit             // rh_optimize
  @native def sample(model: Long, icfg: Array[Int], dcfg: Array[Double], staticMass: Array[Double],
                     seeds: Array[Long], draws: Array[Double], mass: Array[Double],
                     stats: Array[Double]): Unit                                                  // rh_sample
}
