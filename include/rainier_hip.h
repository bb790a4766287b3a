/* rainier_hip.h -- C ABI of the MI355X-native HMC engine that drops in behind Rainier's
 * model.sample().  Plain pointers and sizes only; no C++/torch types.  A JVM reaches it through
 * the thin JNI shim in rainier_amd/jni/ (see INTEGRATION.md for the Scala-side binding).
 *
 * Each entry point replaces one reference seam (paths under /root/reference):
 *
 *   rh_model_create     <- Compiler.compileTargets(group): ir.DataFunction
 *                          rainier-compute/src/main/scala/com/stripe/rainier/compute/Compiler.scala:14-30
 *                          (the ASM bytecode back end ir/CompiledFunction.scala:42-120 is what is replaced:
 *                           RIR is lowered once per model to a fused fp64 HIP gradient kernel)
 *   rh_density_eval     <- trait DensityFunction { nVars; update(vars); density; gradient(i) }
 *                          rainier-sampler/src/main/scala/com/stripe/rainier/sampler/DensityFunction.scala:3-8
 *                          as built by Model.density()  rainier-core/.../core/Model.scala:38-50  (batched over chains)
 *   rh_sample           <- Driver.sample(chain, config, density, progress)(rng)
 *                          rainier-sampler/.../sampler/Driver.scala:7-46, called once per chain by
 *                          Model.sample  rainier-core/.../core/Model.scala:13-24   (here: one call, all chains)
 *   rh_config           <- SamplerConfig + the concrete plugin classes
 *                          sampler/Sampler.scala:3-62, HMC.scala:3-33, EHMC.scala:3-74, DualAvg.scala:3-25,
 *                          MassMatrix.scala:120-173
 *   rh_diagnostics      <- Trace.diagnostics  rainier-core/.../core/Trace.scala:11-21,52-120
 *
 * Errors: every function returns an int status (RH_OK == 0); nothing throws or aborts across the
 * ABI.  NaN log-density is data, not an error (LeapFrog.scala:138-142: the proposal is rejected).
 * An out-of-range Lookup index (the reference's generated NullPointerException,
 * ir/MethodGenerator.scala:164-167) is reported as RH_E_LOOKUP after the launch.
 */
#ifndef RAINIER_HIP_H
#define RAINIER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RH_ABI_VERSION 6  /* 2: rh_chain_stats.bfmi, rh_config.rng_next_gaussian, rh_optimize, rh_sampler_progress / mass_dense;
                             3: rh_density_eval_ex, rh_sample_multi, rh_comm_* (RCCL all-gather of the draws);
                             4: rh_model_clone;
                             5: rh_model_engines, rh_compile_count;
                             6: rh_timing.chain_slots / steady_* (the tick engine's gradient launches serve the live chains only) */

enum rh_status {
  RH_OK = 0,
  RH_E_INVALID = 1,   /* bad argument / malformed RIR (IllegalArgumentException) */
  RH_E_COMPILE = 2,   /* lowering or hiprtc failed */
  RH_E_DEVICE = 3,    /* HIP runtime error / no usable device */
  RH_E_LOOKUP = 4,    /* Lookup index outside [low, low+size) during evaluation */
  RH_E_UNSUPPORTED = 5
};

typedef struct rh_model rh_model;     /* compiled model + device-resident observation columns */
typedef struct rh_sampler rh_sampler; /* device-resident state of `chains` independent chains */

/* ---- seam 1: compile ------------------------------------------------------------------------ */

enum rh_math_mode {
  RH_MATH_FAST = 0,   /* exp/log nodes -> ROCm device libm / the prelude's fast log, <= 1 ulp measured, like java.lang.Math's bound; the other
                         device math functions measure 0.69-0.85 ulp except tan 1.07, atan 1.32, pow 1.29: inside the
                         stated fp64 tolerance, above java.lang.Math's 1-ulp specification */
  RH_MATH_STRICT = 1  /* exp/log nodes -> fdlibm (java.lang.StrictMath): bit-reproducible across machines */
};

typedef struct rh_compile_opts {
  int32_t struct_size;   /* = sizeof(rh_compile_opts) */
  int32_t device;        /* HIP device ordinal; -1 = current device */
  int32_t math_mode;     /* rh_math_mode */
  int32_t fp_contract;   /* 0 = never fuse a*b+c in model code (JVM semantics, default); 1 = allow FMA */
  /* The three shape fields are UPPER BOUNDS: the engine lowers a model again with a smaller value while a kernel of the requested
     shape is not fit to run on this toolchain (spilled registers / vector code ahead of a join's exec restore: rh_model_engines),
     so a caller never gets a kernel the engine would not launch; the generated source (rh_model_hip_source) says what was built. */
  int32_t rows_unroll;   /* row-loop unroll factor of the chain-per-wavefront kernel; 0 = engine default */
  int32_t grad_chains;   /* tick engine: chains that share one pass over the rows per wavefront; 0 = default (4) */
  int32_t grad_unroll;   /* tick engine: row-loop unroll; 0 = default (2) */
  int32_t factor_outputs; /* 1 = accumulate basis terms t and apply alpha*sum(t) + nrows*beta once per gradient, for outputs of
                            the form alpha*t + beta with alpha, beta parameter-only (changes rounding only; default 0) */
  int32_t with_nuts;      /* sampler-kernel variants to compile now instead of lazily on first use -- bit 0: NUTS, bit 1: dense
                             mass matrix, bit 2: one chain per wavefront for a model whose chains are packed (data-free
                             models; used for EHMC / NUTS below 4096 chains) */
  int32_t reserved;
} rh_compile_opts;

/* rir/rir_len      : RIR blob (include/rainier_hip_rir.h).
 * columns          : host pointers to the observation columns, flattened target by target then column
 *                    by column (DataFunction.data(target)(col), ir/DataFunction.scala:13-16), each nrows[target] long.
 * nrows            : one entry per target (ignored for targets without columns).
 * The engine copies the columns to HBM; the caller keeps ownership of its arrays. */
int rh_model_create(const void *rir, size_t rir_len, const double *const *columns, const int64_t *nrows,
                    const rh_compile_opts *opts, rh_model **out);
/* The same model on another device of this process -- what Model.sample's loop over chains (core/Model.scala:16-22) needs when
 * the chains are spread over several GPUs (rh_sample_multi): the lowered program and its code object are reused as they are
 * (no parsing, no data-dependent passes, no compilation unless the architectures differ) and the observation columns are
 * copied device to device over xGMI; the host arrays rh_model_create was given are not needed again.  device = -1: the
 * current device (a second, independent handle on the same device). */
int rh_model_clone(const rh_model *src, int32_t device, rh_model **out);
void rh_model_destroy(rh_model *m);
int rh_model_nvars(const rh_model *m);
/* the generated HIP source (the analogue of rainier-decompile's view of the generated bytecode) */
const char *rh_model_hip_source(const rh_model *m);
/* Which of the model's engines can run, and why not.  Before a kernel is launched its code object is inspected (no spilled vector
 * registers; no vector instruction ahead of a control-flow join's exec restore -- a fault of the ROCm 7.2 register allocator that
 * produced silent wrong sums) and, at creation, checked against another kernel of the same model on the device.  A
 * kernel that fails is replaced by a lighter build or by the other engine (RH_ENGINE_AUTO); an explicit request for an engine
 * that is out of use returns RH_E_UNSUPPORTED.  The reference has no counterpart: its back end splits any method that is too
 * heavy (ir/Packer.scala:10-71) instead of meeting a register file.
 * *chain_engine / *tick_engine / *density: 1 = usable (rh_sample with that engine / rh_density_eval); why: the reasons of the
 * unusable ones, text (may be NULL); compile_attempts: code objects built or fetched while lowering this model. */
int rh_model_engines(const rh_model *m, int32_t *chain_engine, int32_t *tick_engine, int32_t *density, int32_t *compile_attempts,
                     char *why, size_t why_cap);
/* hiprtc compilations this process has performed so far (kernel-cache misses; model creation costs seconds each) */
int64_t rh_compile_count(void);
/* last error message: of this model, or of the calling thread when m == NULL */
const char *rh_last_error(const rh_model *m);

/* ---- seam 2: DensityFunction, batched --------------------------------------------------------
 * q [chains][nvars] -> logp [chains], grad [chains][nvars] (host pointers).  Stateless, one call at a
 * time per model.  chains = 1 restores the reference trait exactly. */
int rh_density_eval(rh_model *m, const double *q, int32_t chains, double *logp, double *grad);
/* The same seam with the evaluation path chosen by the caller (rh_engine_kind, declared below):
 *   RH_ENGINE_CHAIN : one chain per wavefront streams every row itself (rh_density_kernel; what rh_density_eval does for
 *                     models without a parameter table)
 *   RH_ENGINE_TICK  : the sampler's batched gradient path -- the row-streaming gradient kernel the model was lowered to
 *                     (rh_grad_kernel | rh_grad_glm_kernel | rh_grad_gather_kernel) over `grad_splits` row splits (0 = the
 *                     sampler's default) followed by the fixed-order combine of rh_tick_kernel
 *   RH_ENGINE_AUTO  : rh_density_eval's choice.
 * Results of the two paths agree to rounding (different summation order); both are DensityFunction.update batched. */
int rh_density_eval_ex(rh_model *m, const double *q, int32_t chains, int32_t engine, int32_t grad_splits, double *logp,
                       double *grad);

/* ---- seam 3: SamplerConfig / Driver.sample ---------------------------------------------------- */

/* RH_ENGINE_CHAIN: one chain per wavefront, the whole Driver loop in one persistent kernel (best for models with
 *                  little or no data: cfg 1, cfg 3).
 * RH_ENGINE_TICK : per leapfrog step one batched gradient launch shared by all chains (rows read once per group of
 *                  chains) + one small per-chain automaton launch (best for data-heavy models: cfg 2, 4).
 * RH_ENGINE_AUTO : TICK when the model streams >= 65536 rows, else CHAIN. */
enum rh_engine_kind { RH_ENGINE_AUTO = 0, RH_ENGINE_CHAIN = 1, RH_ENGINE_TICK = 2 };
/* RH_SAMPLER_NUTS is an EXTENSION (the reference has no NUTS): iterative multinomial NUTS
 * (Phan, Pradhan, Jankowiak 2019, App. A) behind the same Sampler plugin point (sampler/Sampler.scala:52-62). */
enum rh_sampler_kind { RH_SAMPLER_HMC = 0, RH_SAMPLER_EHMC = 1, RH_SAMPLER_NUTS = 2 };
enum rh_step_kind { RH_STEP_DUALAVG = 0, RH_STEP_STATIC = 1 };
/* RH_MASS_DENSE_WINDOWED = DenseMassMatrixTuner(initialWindowSize, windowExpansion, skipFirst, skipLast)
 * (sampler/MassMatrix.scala:175-181, CovarianceEstimator + packed Cholesky :15-117); at most 64 parameters. */
enum rh_mass_kind { RH_MASS_IDENTITY = 0, RH_MASS_DIAG_WINDOWED = 1, RH_MASS_STATIC_DIAG = 2, RH_MASS_DENSE_WINDOWED = 3 };

typedef struct rh_config {
  int32_t struct_size;     /* = sizeof(rh_config) */
  int32_t iterations;      /* SamplerConfig.iterations        (DefaultConfig: 1000) */
  int32_t warmup;          /* SamplerConfig.warmupIterations  (DefaultConfig: 1000) */
  int32_t sampler;         /* rh_sampler_kind */
  int32_t hmc_steps;       /* HMCSampler(nSteps) */
  int32_t ehmc_max_steps;  /* EHMCSampler(maxSteps, minSteps = 1, bufSize = 100, pCount = 0.1) */
  int32_t ehmc_min_steps;
  int32_t ehmc_buf_size;   /* <= 256 */
  double ehmc_p_count;
  int32_t step_tuner;      /* rh_step_kind */
  int32_t mass_tuner;      /* rh_mass_kind */
  double dualavg_delta;    /* DualAvgTuner(delta) */
  double static_step;      /* StaticStepSize(stepSize) */
  int32_t mass_init_window; /* DiagonalMassMatrixTuner(initialWindowSize, windowExpansion, skipFirst, skipLast) */
  int32_t mass_skip_first;
  int32_t mass_skip_last;
  int32_t nuts_max_depth;  /* RH_SAMPLER_NUTS: maximum tree depth, 1..12 (10 in BASELINE configs 3-5) */
  double mass_expansion;
  const double *static_mass; /* [nvars] DiagonalMassMatrix.elements for RH_MASS_STATIC_DIAG (host pointer) */
  int32_t engine;          /* rh_engine_kind: how the chains are mapped onto the device (results agree to rounding) */
  int32_t grad_splits;     /* tick engine: row splits per chain group; 0 = engine default */
  /* Continuing an existing ScalaRNG / java.util.Random stream (Driver.sample takes the caller's rng, sampler/Driver.scala:7-11;
   * the SBC goldsets sample from the stream that generated the synthetic data): pass seeds[c] = internalState ^ 0x5DEECE66D
   * and here the stream's pending nextNextGaussian per chain ([chains] doubles, NaN = none pending).  NULL: fresh streams. */
  const double *rng_next_gaussian;
  int64_t reserved[2];
} rh_config;

/* Fills DefaultConfig (sampler/Sampler.scala:17-27): 1000/1000, DualAvgTuner(0.8),
 * DiagonalMassMatrixTuner(50, 1.5, 50, 50), EHMCSampler(1024). */
void rh_config_default(rh_config *cfg);

typedef struct rh_chain_stats {
  int64_t leapfrog_steps;        /* distinct (p,q) updates in the sampling phase */
  int64_t warmup_leapfrog_steps;
  int64_t gradient_evaluations;  /* density evaluations the engine performed (sampling phase) */
  int64_t accepted;              /* sampling phase */
  double mean_accept_prob;       /* sampling-phase mean of exp(logAcceptanceProb) */
  double step_size;              /* stepSizeTuner.stepSize used for sampling */
  int32_t error;                 /* rh_status of this chain (RH_E_LOOKUP ...) */
  int32_t reserved;
  double bfmi;                   /* Stats.bfmi = energyTransitions2 / energyVariance.raw(0) over the sampling phase
                                    (sampler/Stats.scala:14-16, LeapFrog.scala:68-74; HMC and EHMC; NaN for NUTS) */
} rh_chain_stats;

/* One call = Model.sample's loop over chains, run concurrently on the device.
 * seeds [chains]     : chain c behaves as the reference run with ScalaRNG(seeds[c]) and nChains = 1.
 * draws [chains][iterations][nvars], mass_diag [chains][nvars] (1.0 = identity), stats [chains]: host
 * pointers, caller-allocated; mass_diag and stats may be NULL. */
int rh_sample(rh_model *m, const rh_config *cfg, const int64_t *seeds, int32_t chains, double *draws,
              double *mass_diag, rh_chain_stats *stats);

/* Multi-GPU Model.sample: the chains are cut into n_models contiguous shards, shard g = global chains
 * [g*chains/n_models, ...) runs on models[g] -- the SAME program and data compiled once per device (rh_model_create with
 * rh_compile_opts.device = g), driven by one host thread per shard.  seeds / draws / mass_diag / stats are indexed by GLOBAL
 * chain id exactly as in rh_sample, so the result is identical for every n_models (chains never interact:
 * sampler/Driver.scala:13-17); there is no device-to-device traffic, the caller's host buffer is the gather.
 * (bench.py's one-process-per-GPU launch uses rh_sampler_* per rank and one RCCL all-gather of the device-resident draws
 * instead; both forms shard by global chain id.)  Replaces the loop  core/Model.scala:16-22. */
int rh_sample_multi(rh_model *const *models, int32_t n_models, const rh_config *cfg, const int64_t *seeds, int32_t chains,
                    double *draws, double *mass_diag, rh_chain_stats *stats);

/* Multi-PROCESS multi-GPU (one process per device -- what `torch.distributed.run`, MPI or one JVM per GPU give): every
 * rank samples its own shard of the chains (seeds by GLOBAL chain id; no collective on the data path) with rh_sampler_* and
 * the device-resident draws are gathered ONCE with an RCCL all-gather over xGMI.  RCCL is loaded with dlopen on first use;
 * without it these calls return RH_E_UNSUPPORTED.  Bootstrap: rank 0 calls rh_comm_unique_id and hands the 128 bytes to the
 * other ranks by whatever channel the host program has; all ranks then call rh_comm_create (collective). */
#define RH_COMM_ID_BYTES 128
typedef struct rh_comm rh_comm;
int rh_comm_unique_id(unsigned char id[RH_COMM_ID_BYTES]);
int rh_comm_create(const unsigned char id[RH_COMM_ID_BYTES], int32_t world, int32_t rank, int32_t device, rh_comm **out);
void rh_comm_destroy(rh_comm *c);
/* collective: gathers every rank's draws ([chains][iterations][nvars], equal on all ranks) into [world * chains][...],
 * rank-major == global chain id order.  host_out (may be NULL): caller-allocated host buffer; *dev_out (may be NULL): device
 * pointer of the gathered buffer, owned by the communicator (valid until the next gather or rh_comm_destroy). */
int rh_comm_allgather_draws(rh_comm *c, rh_sampler *s, double *host_out, void **dev_out);
/* collective: *value = max over ranks (the timing reduction of a benchmark; doubles as a device-side barrier) */
int rh_comm_allreduce_max(rh_comm *c, double *value);
/* hipDeviceSynchronize on `device` (for hosts that never touch the HIP runtime themselves) */
int rh_device_synchronize(int32_t device);

/* The same, split so that a caller can keep draws on the device, poll progress (sampler/Progress.scala)
 * and time phases.  Typical use: create -> warmup -> run(iterations) -> read stats -> destroy. */
int rh_sampler_create(rh_model *m, const rh_config *cfg, const int64_t *seeds, int32_t chains, rh_sampler **out);
void rh_sampler_destroy(rh_sampler *s);
/* advance every chain through LeapFrog.initialize, the step-size search and all warmup iterations */
int rh_sampler_warmup(rh_sampler *s);
/* advance every chain by n sampling iterations (n <= iterations remaining) */
int rh_sampler_run(rh_sampler *s, int32_t n);
/* copy draws of iterations [first, first+count) to the host: out [chains][count][nvars] */
int rh_sampler_draws(rh_sampler *s, int32_t first, int32_t count, double *out);
/* device pointer to all draws, layout [chains][iterations][nvars] (valid until destroy) */
int rh_sampler_draws_device(rh_sampler *s, void **dev_ptr);
int rh_sampler_stats(rh_sampler *s, rh_chain_stats *stats /*[chains]*/, double *mass_diag /*[chains][nvars] or NULL*/);
/* RH_MASS_DENSE_WINDOWED only: the final DenseMassMatrix.elements of every chain, out [chains][nvars][nvars] */
int rh_sampler_mass_dense(rh_sampler *s, double *out);

typedef struct rh_timing {
  double kernel_ms;          /* device time of the DOMINANT kernel's launches since the last reset (HIP events
                                recorded on the engine's own stream around each launch) */
  double total_ms;           /* device time of all of the engine's launches in the same interval */
  int64_t launches;          /* launches of the dominant kernel */
  int64_t density_evals;     /* chain-level density evaluations in those launches */
  int64_t row_chain_evals;   /* rows streamed x chains: the unit of SURVEY.md 8(d) */
  char dominant_kernel[64];  /* name of the kernel that accounts for kernel_ms */
  /* The tick engine's gradient launches serve the chains that wait for a gradient (under EHMCSampler -- DefaultConfig's,
   * sampler/Sampler.scala:17-27 -- and NUTS the chains' trajectories end at different launches).  chain_slots: the chains listed for
   * those launches, summed (= density_evals when no launch served a chain that did not need it; launches x chains without
   * compaction).  steady_*: the same three figures restricted to the launches that served >= 90 % of the sampler's chains -- the
   * dominant kernel at full occupancy, the tail of a run (few chains left) excluded.  Chain engine: chain_slots = density_evals, steady_* = 0. */
  int64_t chain_slots;
  double steady_kernel_ms;
  int64_t steady_launches;
  int64_t steady_density_evals;
} rh_timing;
int rh_sampler_timing(rh_sampler *s, rh_timing *out, int reset);
/* Progress polling instead of a callback into the caller (Driver.sample's `progress: Progress` argument,
 * sampler/Driver.scala:7-11, sampler/Progress.scala): how far the chains of this handle have been driven.
 * warmed: 1 once the warm-up phase is complete; iterations_done: sampling iterations completed (of cfg.iterations). */
int rh_sampler_progress(const rh_sampler *s, int32_t *warmed, int32_t *iterations_done);

/* ---- after the path: Generator.prepare / Trace.predict (next-row f3) -----------------------------------------------
 * The reference compiles a generator's `requirements` with Compiler.default.compile(parameters, namedReqs) and evaluates
 * them once per draw on the JVM (rainier-core/.../core/Generator.scala:59-94, core/Trace.scala:34-41).  Here the same
 * program is evaluated for every draw on the device, one thread per draw.
 * rir: a "requirements program" = an RIR blob whose header `reserved` word is 1 and whose targets are data-free, one per
 * requirement, with outputs[0] = the requirement (the other outputs are ignored).
 * draws [ndraws][nvars] (host) -> out [ndraws][n_requirements] (host). */
int rh_requirements_eval(const void *rir, size_t rir_len, const rh_compile_opts *opts, const double *draws,
                         int64_t ndraws, double *out);

/* ---- Model.optimize / Optimizer.lbfgs ---------------------------------------------------------------
 * Optimizer.lbfgs(df: DensityFunction): Array[Double]   rainier-sampler/.../optimizer/Optimizer.scala:6-24
 *   (L-BFGS m = 5, eps = 0.1 on f = -density; LBFGS.java:44-632), called by Model.optimize (core/Model.scala:26-30).
 * Batched over independent starting points: x0 [starts][nvars] or NULL (all zeros = the reference's start); every round
 * is one batched device density launch for the starts still searching; the O(m n) vector updates stay on the host.
 * x_out [starts][nvars]; evals_out / status_out [starts] may be NULL.  max_evals <= 0: unlimited (the reference's loop).
 * status_out: RH_OPT_CONVERGED, RH_OPT_NOT_DESCENT (where the reference throws RuntimeException("dginit")),
 * RH_OPT_MAX_EVALS. */
enum rh_opt_status { RH_OPT_CONVERGED = 0, RH_OPT_NOT_DESCENT = 1, RH_OPT_MAX_EVALS = 2 };
int rh_optimize(rh_model *m, const double *x0, int32_t starts, int32_t max_evals, double *x_out, int32_t *evals_out,
                int32_t *status_out);

/* ---- Trace.diagnostics --------------------------------------------------------------------------
 * draws [chains][iterations][nvars] (host) -> rhat [nvars], ess [nvars]   (core/Trace.scala:52-120) */
int rh_diagnostics(const double *draws, int32_t chains, int32_t iterations, int32_t nvars, double *rhat, double *ess);

int rh_abi_version(void);
/* number of visible HIP devices, or a negative rh_status */
int rh_device_count(void);

#ifdef __cplusplus
}
#endif
#endif
