// integration/scala/HipModel.scala -- NOT BUILT HERE.  Drop into rainier-core/src/main/scala/com/stripe/rainier/core/.
// The three seams of INTEGRATION.md as an enrichment of core.Model: density() / sample() / optimize() on the device.
package com.stripe.rainier.core

import com.stripe.rainier.compute._
import com.stripe.rainier.hip.Native
import com.stripe.rainier.sampler._

/** Drop-in for Model.density() (core/Model.scala:38-50): the DensityFunction trait over rh_density_eval(chains = 1). */
final class NativeDensityFunction(model: Long, val nVars: Int) extends DensityFunction {
  private val lp = new Array[Double](1)
  private val g = new Array[Double](nVars)
  def update(vars: Array[Double]): Unit = Native.densityEval(model, vars, 1, lp, g)
  def density: Double = lp(0)
  def gradient(index: Int): Double = g(index)
}

/** SamplerConfig -> the flat arrays rainier_hip_jni.c unpacks into rh_config.  The reference's plugin classes keep their
  * constructor parameters private (class HMCSampler(nSteps: Int) ...), so they are read by reflection; a maintainer may
  * prefer to turn them into `val`s. */
object HipConfig {
  private def field[T](o: AnyRef, name: String): T = {
    val f = o.getClass.getDeclaredField(name); f.setAccessible(true); f.get(o).asInstanceOf[T]
  }
  /** (icfg, dcfg, staticMass or null); order documented in INTEGRATION.md section 3. */
  def flatten(c: SamplerConfig): (Array[Int], Array[Double], Array[Double]) = {
    var staticMass: Array[Double] = null
    val (sampler, nSteps, maxSteps, minSteps, bufSize, pCount) = c.sampler() match {
      case s: HMCSampler  => (0, field[Int](s, "nSteps"), 1024, 1, 100, 0.1)
      case s: EHMCSampler => (1, 1, field[Int](s, "maxSteps"), field[Int](s, "minSteps"), field[Int](s, "bufSize"), field[Double](s, "pCount"))
      case other          => throw new IllegalArgumentException(s"no device implementation of sampler $other")
    }
    val (stepTuner, delta, staticStep) = c.stepSizeTuner() match {
      case t: DualAvgTuner   => (0, field[Double](t, "delta"), 0.0)
      case StaticStepSize(e) => (1, 0.8, e)
      case other             => throw new IllegalArgumentException(s"no device implementation of step-size tuner $other")
    }
    val (massTuner, window, expansion, skipFirst, skipLast) = c.massMatrixTuner() match {
      case _: IdentityMassMatrixTuner => (0, 50, 1.5, 50, 50)
      case t: DiagonalMassMatrixTuner => (1, t.initialWindowSize, t.windowExpansion, t.skipFirst, t.skipLast)
      case t: DenseMassMatrixTuner    => (3, t.initialWindowSize, t.windowExpansion, t.skipFirst, t.skipLast)
      case StaticMassMatrix(DiagonalMassMatrix(elements)) => staticMass = elements; (2, 50, 1.5, 50, 50)
      case StaticMassMatrix(IdentityMassMatrix)           => (0, 50, 1.5, 50, 50)
      case other => throw new IllegalArgumentException(s"no device implementation of mass-matrix tuner $other")
    }
    (Array(c.iterations, c.warmupIterations, sampler, nSteps, maxSteps, minSteps, bufSize, stepTuner, massTuner, window, skipFirst, skipLast),
     Array(pCount, delta, staticStep, expansion), staticMass)
  }
}

object HipModel {
  implicit final class Ops(private val model: Model) extends AnyVal {
    private def create(strictMath: Boolean): Long = {
      val rir = HipCompiler.compileTargets(model.targetGroup)
      Native.modelCreate(rir.bytes, rir.columns, rir.rows, -1, if (strictMath) 1 else 0, false)
    }

    /** Model.density() on the device; the caller owns the handle through `close`. */
    def hipDensity(): (DensityFunction, () => Unit) = {
      val h = create(strictMath = false)
      (new NativeDensityFunction(h, model.parameters.size), () => Native.modelDestroy(h))
    }

    /** Model.sample (core/Model.scala:13-24): all chains in ONE native call.  Chain c is the reference run with
      * nChains = 1 and ScalaRNG(seeds(c)); the seeds themselves are drawn from the caller's rng. */
    def hipSample(config: SamplerConfig = SamplerConfig.default, nChains: Int = 4)(implicit rng: RNG = RNG.default): Trace = {
      val n = model.parameters.size
      val h = create(strictMath = false)
      try {
        val seeds = Array.fill(nChains)(java.lang.Double.doubleToRawLongBits(rng.standardUniform))
        val (icfg, dcfg, staticMass) = HipConfig.flatten(config)
        val draws = new Array[Double](nChains * config.iterations * n)
        val mass = new Array[Double](nChains * n)
        val stats = new Array[Double](nChains * 7)
        Native.sample(h, icfg, dcfg, staticMass, seeds, draws, mass, stats)
        val chains = draws.grouped(config.iterations * n).map(_.grouped(n).toList).toList
        val masses: List[MassMatrix] = mass.grouped(n).map { m =>
          if (m.forall(_ == 1.0)) IdentityMassMatrix else DiagonalMassMatrix(m)
        }.toList
        val sts = stats.grouped(7).map { s =>
          val st = new Stats(config.statsWindow)
          st.gradientEvaluations = s(2).toLong
          st.iterations = config.iterations
          st.stepSizes.add(s(5)); st.acceptanceRates.add(s(4))
          st.gradsPerIteration.add(s(2) / math.max(1, config.iterations))
          st                                                     // s(6) = Stats.bfmi as computed on the device
        }.toList
        Trace(chains, masses, sts, model)
      } finally Native.modelDestroy(h)
    }

    /** Model.optimize's numeric part (Optimizer.lbfgs, optimizer/Optimizer.scala:6-24) from the reference's start x = 0. */
    def hipOptimum(): Array[Double] = {
      val n = model.parameters.size
      val h = create(strictMath = false)
      try {
        val x = new Array[Double](n); val evals = new Array[Int](1); val status = new Array[Int](1)
        Native.optimize(h, null, 1, 0, x, evals, status)
        if (status(0) == 1) throw new RuntimeException("dginit")   // optimizer/LBFGS.java:236-237
        x
      } finally Native.modelDestroy(h)
    }
  }
}
