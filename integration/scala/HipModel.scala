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
  def update(vars: Array[Double]): Unit = Native.densityEval(model, vars, 1, 0, 0, lp, g)
  def density: Double = lp(0)
  def gradient(index: Int): Double = g(index)
}

/** NUTS (extension: the reference has no NUTS sampler, core/Trace.scala:104 is a comment).  A marker in the reference's
  * Sampler plugin slot (sampler/Sampler.scala:52-62); it only exists on the device, the JVM Driver cannot run it. */
final case class NUTSSampler(maxDepth: Int = 10) extends Sampler {
  private def jvm = throw new UnsupportedOperationException("NUTSSampler runs on the HIP engine only (Model.hipSample)")
  def initialize(params: Array[Double], lf: LeapFrog)(implicit rng: RNG): Unit = jvm
  def warmup(params: Array[Double], lf: LeapFrog, stepSize: Double, mass: MassMatrix)(implicit rng: RNG): Double = jvm
  def run(params: Array[Double], lf: LeapFrog, stepSize: Double, mass: MassMatrix)(implicit rng: RNG): Unit = jvm
}

/** rh_compile_opts as the int array rainier_hip_jni.c unpacks (RH_JNI_COPTS: same names, same order).
  * Defaults = the build bench.py measures: FMA contraction + output factoring (results within the stated fp64 tolerance of
  * the JVM's, DESIGN.md section 4); `jvmFaithful` = JVM arithmetic (no FMA, un-factored, StrictMath exp/log). */
final case class HipOptions(device: Int = -1, math_mode: Int = 0, fp_contract: Int = 1, rows_unroll: Int = 0,
                            grad_chains: Int = 0, grad_unroll: Int = 0, factor_outputs: Int = 1, with_nuts: Int = 0) {
  def copts: Array[Int] =
    Array(device, math_mode, fp_contract, rows_unroll, grad_chains, grad_unroll, factor_outputs, with_nuts)
}
object HipOptions {
  val fast = HipOptions()
  val jvmFaithful = HipOptions(math_mode = 1, fp_contract = 0, factor_outputs = 0)
}

/** SamplerConfig -> the flat arrays rainier_hip_jni.c unpacks into rh_config (RH_JNI_ICFG / RH_JNI_DCFG: same names, same
  * order).  The reference's plugin classes keep their constructor parameters private (class HMCSampler(nSteps: Int) ...),
  * so they are read by reflection; a maintainer may prefer to turn them into `val`s. */
object HipConfig {
  private def field[T](o: AnyRef, name: String): T = {
    val f = o.getClass.getDeclaredField(name); f.setAccessible(true); f.get(o).asInstanceOf[T]
  }
  /** (icfg, dcfg, staticMass or null).  engine: 0 auto, 1 chain-per-wavefront, 2 tick; grad_splits 0 = engine default. */
  def flatten(c: SamplerConfig, engine: Int = 0, grad_splits: Int = 0): (Array[Int], Array[Double], Array[Double]) = {
    var staticMass: Array[Double] = null
    val iterations = c.iterations
    val warmup = c.warmupIterations
    val (sampler, hmc_steps, ehmc_max_steps, ehmc_min_steps, ehmc_buf_size, ehmc_p_count, nuts_max_depth) = c.sampler() match {
      case s: HMCSampler  => (0, field[Int](s, "nSteps"), 1024, 1, 100, 0.1, 10)
      case s: EHMCSampler => (1, 1, field[Int](s, "maxSteps"), field[Int](s, "minSteps"), field[Int](s, "bufSize"), field[Double](s, "pCount"), 10)
      case NUTSSampler(d) => (2, 1, 1024, 1, 100, 0.1, d)
      case other          => throw new IllegalArgumentException(s"no device implementation of sampler $other")
    }
    val (step_tuner, dualavg_delta, static_step) = c.stepSizeTuner() match {
      case t: DualAvgTuner   => (0, field[Double](t, "delta"), 0.0)
      case StaticStepSize(e) => (1, 0.8, e)
      case other             => throw new IllegalArgumentException(s"no device implementation of step-size tuner $other")
    }
    val (mass_tuner, mass_init_window, mass_expansion, mass_skip_first, mass_skip_last) = c.massMatrixTuner() match {
      case _: IdentityMassMatrixTuner => (0, 50, 1.5, 50, 50)
      case t: DiagonalMassMatrixTuner => (1, t.initialWindowSize, t.windowExpansion, t.skipFirst, t.skipLast)
      case t: DenseMassMatrixTuner    => (3, t.initialWindowSize, t.windowExpansion, t.skipFirst, t.skipLast)
      case StaticMassMatrix(DiagonalMassMatrix(elements)) => staticMass = elements; (2, 50, 1.5, 50, 50)
      case StaticMassMatrix(IdentityMassMatrix)           => (0, 50, 1.5, 50, 50)
      case other => throw new IllegalArgumentException(s"no device implementation of mass-matrix tuner $other")
    }
    val icfg: Array[Int] =
      Array(iterations, warmup, sampler, hmc_steps, ehmc_max_steps, ehmc_min_steps, ehmc_buf_size, step_tuner,
            mass_tuner, mass_init_window, mass_skip_first, mass_skip_last, nuts_max_depth, engine, grad_splits)
    val dcfg: Array[Double] =
      Array(ehmc_p_count, dualavg_delta, static_step, mass_expansion)
    (icfg, dcfg, staticMass)
  }
}

object HipModel {
  /** sampler.Stats from the 7 doubles per chain the shim writes (RH_JNI_STATS order):
    * leapfrog_steps, warmup_leapfrog_steps, gradient_evaluations, accepted, mean_accept_prob, step_size, bfmi */
  private def toStats(s: Array[Double], config: SamplerConfig): Stats = {
    val st = new Stats(config.statsWindow)
    st.gradientEvaluations = s(2).toLong
    st.iterations = config.iterations
    st.stepSizes.add(s(5)); st.acceptanceRates.add(s(4))
    st.gradsPerIteration.add(s(2) / math.max(1, config.iterations))
    st                                                     // s(6) = Stats.bfmi as computed on the device
  }

  implicit final class Ops(private val model: Model) extends AnyVal {
    private def create(opts: HipOptions): Long = {
      val rir = HipCompiler.compileTargets(model.targetGroup)
      Native.modelCreate(rir.bytes, rir.columns, rir.rows, opts.copts)
    }

    /** Which engines the model can use on this toolchain (rh_model_engines, ABI 5): (chain engine, tick engine, density seam,
      * code objects built or fetched while lowering).  The engine inspects every kernel before it launches it and replaces what is
      * not fit to run (INTEGRATION.md section 5); RH_ENGINE_AUTO never needs this, an explicit engine request should look first. */
    def hipEngines(opts: HipOptions = HipOptions.fast): (Boolean, Boolean, Boolean, Int) = {
      val h = create(opts)
      try { val e = Native.modelEngines(h); ((e & 1) != 0, (e & 2) != 0, (e & 4) != 0, e >> 8) } finally Native.modelDestroy(h)
    }

    /** Model.density() on the device; the caller owns the handle through `close`. */
    def hipDensity(opts: HipOptions = HipOptions.fast): (DensityFunction, () => Unit) = {
      val h = create(opts)
      (new NativeDensityFunction(h, model.parameters.size), () => Native.modelDestroy(h))
    }

    /** Model.sample (core/Model.scala:13-24): all chains in ONE native call, on `devices` GPUs (chains sharded by global
      * chain id inside the engine; the result does not depend on the device count).  Chain c is the reference run with
      * nChains = 1 and ScalaRNG(seeds(c)); the seeds themselves are drawn from the caller's rng. */
    def hipSample(config: SamplerConfig = SamplerConfig.default, nChains: Int = 4, opts: HipOptions = HipOptions.fast,
                  devices: Seq[Int] = Seq(-1))(implicit rng: RNG = RNG.default): Trace = {
      val n = model.parameters.size
      val nuts = config.sampler().isInstanceOf[NUTSSampler]
      // lowered and compiled ONCE (for the first device); the other devices get clones: code object reused, columns copied over xGMI
      val first = create(opts.copy(device = devices.head, with_nuts = if (nuts) 1 else opts.with_nuts))
      val handles = (first +: devices.tail.map(d => Native.modelClone(first, d))).toArray
      try {
        val seeds = Array.fill(nChains)(java.lang.Double.doubleToRawLongBits(rng.standardUniform))
        val (icfg, dcfg, staticMass) = HipConfig.flatten(config)
        val draws = new Array[Double](nChains * config.iterations * n)
        val mass = new Array[Double](nChains * n)
        val stats = new Array[Double](nChains * 7)
        Native.sample(handles, icfg, dcfg, staticMass, null, seeds, draws, mass, stats)
        val chains = draws.grouped(config.iterations * n).map(_.grouped(n).toList).toList
        val masses: List[MassMatrix] = mass.grouped(n).map { m =>
          if (m.forall(_ == 1.0)) IdentityMassMatrix else DiagonalMassMatrix(m)
        }.toList
        Trace(chains, masses, stats.grouped(7).map(toStats(_, config)).toList, model)
      } finally handles.foreach(Native.modelDestroy)
    }

    /** Driver.sample(chain, config, density, progress)(rng) for ONE chain that CONTINUES the caller's java.util.Random
      * stream exactly as the reference does (sampler/Driver.scala:7-11; what the SBC goldsets rely on): the stream's
      * internal state and its pending nextNextGaussian are read by reflection and handed to the engine. */
    def hipSampleSharingRng(config: SamplerConfig, random: java.util.Random, opts: HipOptions = HipOptions.jvmFaithful): Trace = {
      def get[T](name: String): T = { val f = classOf[java.util.Random].getDeclaredField(name); f.setAccessible(true); f.get(random).asInstanceOf[T] }
      val state = get[java.util.concurrent.atomic.AtomicLong]("seed").get
      val pending = if (get[Boolean]("haveNextNextGaussian")) get[Double]("nextNextGaussian") else Double.NaN
      val n = model.parameters.size
      val h = create(opts)
      try {
        val (icfg, dcfg, staticMass) = HipConfig.flatten(config)
        val draws = new Array[Double](config.iterations * n); val mass = new Array[Double](n); val stats = new Array[Double](7)
        Native.sample(Array(h), icfg, dcfg, staticMass, Array(pending), Array(state ^ 0x5DEECE66DL), draws, mass, stats)
        Trace(List(draws.grouped(n).toList), List(if (mass.forall(_ == 1.0)) IdentityMassMatrix else DiagonalMassMatrix(mass)),
              List(toStats(stats, config)), model)
      } finally Native.modelDestroy(h)
    }

    /** Model.optimize's numeric part (Optimizer.lbfgs, optimizer/Optimizer.scala:6-24) from the reference's start x = 0. */
    def hipOptimum(opts: HipOptions = HipOptions.fast): Array[Double] = {
      val n = model.parameters.size
      val h = create(opts)
      try {
        val x = new Array[Double](n); val evals = new Array[Int](1); val status = new Array[Int](1)
        Native.optimize(h, null, 1, 0, x, evals, status)
        if (status(0) == 1) throw new RuntimeException("dginit")   // optimizer/LBFGS.java:236-237
        x
      } finally Native.modelDestroy(h)
    }
  }
}
