// rh_shared.h -- POD types shared verbatim by the host engine (C++) and the device code (hiprtc).
// Plain C; no includes (it is also prepended to the runtime-compiled source).
#ifndef RH_SHARED_H
#define RH_SHARED_H

#define RH_MAX_TARGETS 64
#define RH_RING_SLOTS 4  /* EHMC step-count ring buffer: <= 256 entries, lane-distributed */
#define RH_NUTS_MAXD 12  /* NUTS: deepest tree (2^12 leaves); momentum checkpoints per chain */

/* return codes of the per-chain automaton */
#define RH_ADV_NEED_GRAD 0
#define RH_ADV_PAUSED 1
#define RH_ADV_DONE 2

/* device copy of rh_config (include/rainier_hip.h); static mass is uploaded separately */
typedef struct rh_cfg_dev {
  int iterations, warmup;
  int sampler, hmc_steps;
  int ehmc_max_steps, ehmc_min_steps, ehmc_buf_size, step_tuner;
  double ehmc_p_count;
  double dualavg_delta, static_step;
  int mass_tuner, mass_init_window, mass_skip_first, mass_skip_last;
  double mass_expansion;
  int nuts_max_depth, reserved;
} rh_cfg_dev;

/* observation columns resident in HBM: `cols` is a device-resident table of column pointers (flattened target-major, then
 * column), so the number of columns is not bounded by the kernel-argument segment; a kernel reads the few pointers it
 * needs with wave-uniform scalar loads */
typedef struct rh_model_data {
  const double *const *cols;
  long long nrows[RH_MAX_TARGETS];
  const double *kpool;   /* the constants of the data-free targets that are not spelled in the generated source (csrc/rir.hpp EmitInfo::kpool) */
} rh_model_data;

/* gather mode (a parameter table indexed by a data column): rows are sorted by index, group g = rows whose index is
 * low + g; per row target (by ROWT) the group offsets, the group range of every row split and the scatter sums */
typedef struct rh_gather_data {
  const int *goff[RH_MAX_TARGETS];   /* [ngroups + 1] first row of each group */
  const int *gsplit[RH_MAX_TARGETS]; /* [nsplit + 1] first group of each split */
  double *sbuf[RH_MAX_TARGETS];      /* [chains][G_COUNT] per-group sums of the scatter value (gather targets only) */
  int gmin[RH_MAX_TARGETS];          /* rows of the smallest non-empty group: >= 64 means a 64-row tile touches at most two groups */
} rh_gather_data;

/* per-chain result record written by the kernels (mirrors rh_chain_stats) */
typedef struct rh_chain_stats_dev {
  long long leapfrog_steps, warmup_leapfrog_steps, gradient_evaluations, accepted;
  double sum_accept_prob, step_size;
  double e_trans2, e_raw; /* Stats.energyTransitions2, Stats.energyVariance.raw(0) of the sampling phase */
  long long sampling_iterations;
  int error, status; /* status: last RH_ADV_* */
} rh_chain_stats_dev;

#endif
