// rh_engine.hip.h -- hand-written gfx950 device library, part 2.
//
// Comes after the per-model generated code, which defines for every target t a
//   template<> struct rh_target<t> { NCOLS, COL0, NINV, HAS_ROWS, invariants(th, inv, err), row(th, inv, c, acc, err) }
// (the lowering of the reference's per-output bytecode methods, ir/CompiledFunction.scala:42-120).
//
//   * rh_density        : DataFunction.apply (ir/DataFunction.scala:32-84) for ONE chain on ONE wavefront:
//                         lanes stride the observation rows (coalesced 512 B per column per load), per-lane
//                         fp64 accumulators for all n+1 outputs, fixed-order wave butterfly at the end.
//   * rh_chain          : the device-resident state of one chain: LeapFrog's params/pqBuf (LeapFrog.scala:118-129),
//                         DualAvg (DualAvg.scala:43-78), VarianceEstimator + windowed tuner
//                         (MassMatrixEstimator.scala:52-113, MassMatrix.scala:126-173), EHMC ring buffer
//                         (Stats.scala:19-58) and the java.util.Random stream.
//   * rh_advance        : Driver.sample (Driver.scala:7-119) + HMCSampler/EHMCSampler as a resumable automaton
//                         that yields whenever it needs the gradient at pqBuf.q.  Each leapfrog step costs ONE
//                         gradient evaluation (the reference spends two, LeapFrog.scala:158-188, at identical
//                         inputs); the (p, q) sequence is the same.
//   * kernels           : rh_chain_kernel (whole Driver loop, one chain per wavefront), rh_density_kernel
//                         (seam 2, batched DensityFunction), rh_selftest_kernel.
#pragma clang fp contract(off)

template <int T> struct rh_target;

// Generic (non-gather) models with more than 512 parameters: the chain vectors live in HBM (big mode), and so do the things that
// used to be per-wavefront register arrays of RH_NVARS elements -- theta is read in place from the chain's q (wave-uniform
// addresses: scalar loads) and the n + 1 outputs are accumulated in a scratch area of the chain (`tot`: every lane executes the
// same stores of the same values; the accumulation is sequential in program order).  The reference's back end has no parameter
// limit (ir/Packer.scala:10-40); this is how a 2 000-parameter state-space model runs here: correct, not fast.
#define RH_BIGTH (RH_BIGN && !RH_HAS_GATHER && (RH_NVARS > 512 || RH_HEAVY))
#if RH_BIGTH
__shared__ double *rh_tot_base;   // the chain's (or the density call's) scratch for the outputs: RH_NOUT doubles
#endif

#if !RH_HAS_GATHER
// ---- DataFunction.apply for one chain ------------------------------------------------------------
#ifndef RH_ROWS_UNROLL
#define RH_ROWS_UNROLL 4
#endif

#if RH_FP_CONTRACT
#pragma clang fp contract(fast)
#endif
// Rows [kb, n) of target T, kb wave-uniform, n - kb arbitrary: every lane takes every step of the (scalar) loop.  A lane whose row
// lies past the end re-reads the last row -- the addresses stay in bounds, the row function runs with all lanes active -- and
// its contributions are dropped by a select.  t starts at -0.0, so that `t += x` is x in every bit (x + -0.0 == x also for the
// zeros), and the live lanes' `acc + t` is the very addition the plain loop performs.
template <int T, class TH, class INV, class CP, int NA>
RH_DEV void rh_rows_ragged(const TH &th, const INV &inv, const CP &cp, long long kb, const long long n, const int lane,
                           double (&acc)[NA], int &err) {
  typedef rh_target<T> TG;
  constexpr int NC = TG::NCOLS;
  for (; kb < n; kb += 64) {
    const long long k = kb + lane;
    const bool live = k < n;
    const long long kc = live ? k : n - 1;
    double c[NC];
RH_UNROLL_ACC
    for (int j = 0; j < NC; j++) c[j] = cp[j][kc];
    double t[NA];
RH_UNROLL_ACC
    for (int o = 0; o < NA; o++) t[o] = -0.0;
    TG::row(th, inv, c, t, err);   // (a dead lane can only repeat the error of the lane that owns the last row)
RH_UNROLL_ACC
    for (int o = 0; o < NA; o++) acc[o] = live ? acc[o] + t[o] : acc[o];
  }
}

template <int T>
RH_DEV void rh_accumulate_target(const double (&th)[RH_NTH], const rh_model_data &d, const int lane,
                                 double (&tot)[RH_NOUT], int &err) {
  typedef rh_target<T> TG;
  double inv[TG::NINV > 0 ? TG::NINV : 1];
  TG::invariants(th, inv, err);
  if constexpr (!TG::HAS_ROWS) {
    // data-free target: evaluated once, outputs(o) += f_o(theta)   (DataFunction.scala:73-83).  Its `c` is the constant pool: what of
    // the folded observations is not spelled in the generated source (csrc/rir.hpp EmitInfo::kpool)
    TG::row(th, inv, d.kpool, tot, err);
  } else {
    constexpr int NC = TG::NCOLS;
    constexpr int U = RH_ROWS_UNROLL;
    const long long n = d.nrows[T];
    const double *cp[NC];
#pragma unroll
    for (int j = 0; j < NC; j++) cp[j] = d.cols[TG::COL0 + j];
    constexpr int NA = TG::NACC > 0 ? TG::NACC : 1;
    double acc[NA];
RH_UNROLL_ACC
    for (int o = 0; o < NA; o++) acc[o] = 0.0;
    // The walk runs on a WAVE-UNIFORM base row (scalar loop branches), so the row function always executes with all 64 lanes
    // active; the ragged end is handled by rh_rows_ragged.  (A lane-strided loop `for (k = lane; k < n; k += 64)` is a divergent
    // region around the heaviest code of the kernel -- exactly where this toolchain's register allocator puts spill code and
    // copies ahead of the exec restore: DESIGN 8.5.)  A lane still sums its rows lane, lane + 64, ... in ascending order.
    long long kb = 0;
    for (; kb + (long long)RH_LANES * U <= n; kb += (long long)RH_LANES * U) {
      double c[U][NC];
#pragma unroll
      for (int u = 0; u < U; u++)
RH_UNROLL_ACC
        for (int j = 0; j < NC; j++) c[u][j] = cp[j][kb + (long long)RH_LANES * u + lane];
#pragma unroll
      for (int u = 0; u < U; u++) TG::row(th, inv, c[u], acc, err);
    }
    rh_rows_ragged<T>(th, inv, cp, kb, n, lane, acc, err);
    rh_wave_sum_all(acc);
    TG::finish(th, inv, acc, (double)n, tot);
  }
}
template <int T>
RH_DEV void rh_accumulate_all(const double (&th)[RH_NTH], const rh_model_data &d, const int lane,
                              double (&tot)[RH_NOUT], int &err) {
  if constexpr (T < RH_NTARGETS) {
    rh_accumulate_target<T>(th, d, lane, tot, err);
    rh_accumulate_all<T + 1>(th, d, lane, tot, err);
  }
}
#pragma clang fp contract(off)

// q (lane-distributed) -> logp (wave-uniform), grad (lane-distributed)
#if RH_BIGTH
RH_DEV void rh_density(const wvec &q, const rh_model_data &d, const int lane, double &logp, wvec &grad, int &err) {
  const double (&th)[RH_NTH] = *reinterpret_cast<const double (*)[RH_NTH]>(q.s.p);
  double (&tot)[RH_NOUT] = *reinterpret_cast<double (*)[RH_NOUT]>(rh_tot_base);
  for (int o = 0; o < RH_NOUT; o++) tot[o] = 0.0;   // (wave-uniform stores; every lane then reads back what the wavefront wrote)
  rh_accumulate_all<0>(th, d, lane, tot, err);
  logp = tot[0];
  for (int i = lane; i < RH_NVARS; i += 64) grad.s.p[i] = tot[1 + i];
}
#else
RH_DEV void rh_density(const wvec &q, const rh_model_data &d, const int lane, double &logp, wvec &grad, int &err) {
  double th[RH_NTH];
#pragma unroll
  for (int i = 0; i < RH_NTH; i++) th[i] = rh_readlane(q.s[i >> 6], i & 63); // theta in SGPR pairs
  double tot[RH_NOUT];
#pragma unroll
  for (int o = 0; o < RH_NOUT; o++) tot[o] = 0.0;
  rh_accumulate_all<0>(th, d, lane, tot, err);
  logp = tot[0];
  wv_zero(grad);
#pragma unroll
  for (int i = 0; i < RH_NVARS; i++) grad.s[i >> 6] = ((i & 63) == lane) ? tot[1 + i] : grad.s[i >> 6];
}
#endif  // RH_BIGTH

#endif  // !RH_HAS_GATHER

// ---- chain state ----------------------------------------------------------------------------------
#ifndef RH_WITH_NUTS
#define RH_WITH_NUTS 0
#endif
#if RH_WITH_NUTS
#define RH_STATE_VECS_NUTS(X) X(NLq) X(NLp) X(NLg) X(NRq) X(NRp) X(NRg) X(Nrsum)
#define RH_STATE_F64_NUTS(X) X(nH0) X(n_tree_logw) X(n_sub_logw) X(n_sum_acc)
#define RH_STATE_INT_NUTS(X) X(n_depth) X(n_leaf) X(n_right) X(n_leaf_total)
#else
#define RH_STATE_VECS_NUTS(X)
#define RH_STATE_F64_NUTS(X)
#define RH_STATE_INT_NUTS(X)
#endif
#ifndef RH_WITH_DENSE
#define RH_WITH_DENSE 0
#endif
#if RH_WITH_DENSE
#define RH_STATE_INT_DENSE(X) X(mass_dense)
#define RH_STATE_NDENSE (3 * RH_NVARS)
#else
#define RH_STATE_INT_DENSE(X)
#define RH_STATE_NDENSE 0
#endif
#define RH_STATE_VECS(X) \
  X(Pp) X(Pq) X(Pg) X(Bp) X(Bq) X(Bg) X(Sp) X(Sq) X(Sg) X(M) X(SD) X(ve_mean) X(ve_raw) X(pend_g) RH_STATE_VECS_NUTS(X)
#define RH_STATE_F64(X) \
  X(PU) X(BU) X(SU) X(eps) X(da_logEps) X(da_logEpsBar) X(da_avgErr) X(da_mu) X(exponent) X(pend_logp) \
  X(rng_nn) X(sum_accept) X(prevH) X(e_mean) X(e_raw) X(e_trans2) RH_STATE_F64_NUTS(X)
#define RH_STATE_INT(X) \
  X(rng_have) X(pc) X(ret) X(it) X(ts_l) X(ts_i) X(cnt_l) X(find_first) X(sampling_started) X(need_eval) \
  X(mass_identity) X(ve_samples) X(win_size) X(win_i) X(win_j) X(da_iter) X(ring_i) X(ring_full) X(n_accept) \
  X(n_samp_iters) X(err) X(e_samples) RH_STATE_INT_NUTS(X) RH_STATE_INT_DENSE(X)
#define RH_STATE_I64(X) X(rng_seed) X(n_leapfrog) X(n_warm_leapfrog) X(n_grad)

struct rh_chain {
#define X(n) wvec n;
  RH_STATE_VECS(X)
#undef X
#define X(n) double n;
  RH_STATE_F64(X)
#undef X
#define X(n) int n;
  RH_STATE_INT(X)
#undef X
#define X(n) rh_i64 n;
  RH_STATE_I64(X)
#undef X
#if RH_PACK_L == 64
  double ring[RH_RING_SLOTS]; // EHMC step counts, entry i in lane i%64 of slot i/64
#else
  double *ring;               // packed chains keep the ring buffer in their state image: entry i at ring[i]
#endif
#if RH_WITH_NUTS
#if RH_BIGN
  wvec ckr[RH_NUTS_MAXD], ckrs[RH_NUTS_MAXD]; // NUTS momentum / momentum-sum checkpoints (views on the state image)
#else
  rh_u64 *ck; // NUTS checkpoints stay in the chain's state image (word ((j * RH_SLOTS + k) * 2 + {0: r, 1: rsum}) * 64 + lane):
              // they are touched once per leaf, and 24 register vectors behind a dynamic index cost ~50 VGPRs + select chains
#endif
#endif
#if RH_WITH_DENSE
  // DenseMassMatrix (MassMatrix.scala:15-117), row i in lane i: Drow[j] = elements(i*n + j), Lrow[j] = Cholesky lower
  // L[i][j] (choleskyUpperTriangular's U[j][i]), Crow[k] = CovarianceEstimator.cov(i*n + k)
  double Drow[RH_NVARS], Lrow[RH_NVARS], Crow[RH_NVARS];
#endif
};
#if RH_WITH_NUTS
#define RH_STATE_NCK (2 * RH_NUTS_MAXD)
#else
#define RH_STATE_NCK 0
#endif

// positions inside the state image (small-vector layout): vector n starts at slot RH_VI_n * RH_SLOTS, scalar n is word RH_SI_n of
// the scalar block -- the order rh_chain_store / rh_chain_load walk
enum {
#define X(n) RH_VI_##n,
  RH_STATE_VECS(X)
#undef X
  RH_VI_COUNT
};
enum {
#define X(n) RH_SI_##n,
  RH_STATE_F64(X) RH_STATE_INT(X) RH_STATE_I64(X)
#undef X
  RH_SI_COUNT
};
#define RH_CNT(n) +1
#define RH_STATE_NVEC (0 RH_STATE_VECS(RH_CNT))
#define RH_STATE_NSCALAR ((0 RH_STATE_F64(RH_CNT)) + (0 RH_STATE_INT(RH_CNT)) + (0 RH_STATE_I64(RH_CNT)))
// u64 words of one chain's state image: lane-distributed vectors and the ring buffer take 64 words per slot,
// wave-uniform scalars are stored once (lane 0 writes, every lane reads the same address -> scalar loads).
#if RH_BIGN
#define RH_STATE_NPOOL (RH_POOL_VECS + (RH_BIGTH ? 2 : 0))   /* + the outputs' scratch (RH_NOUT = RH_NVARS + 1 doubles: two vectors) */
#else
#define RH_STATE_NPOOL 0
#endif
#define RH_STATE_DENSE_OFF ((RH_STATE_NVEC + RH_STATE_NCK + RH_STATE_NPOOL) * RH_SLOTS + RH_RING_SLOTS)
#define RH_STATE_U64 ((RH_STATE_DENSE_OFF + RH_STATE_NDENSE) * 64 + RH_STATE_NSCALAR)

#if RH_BIGN
// big mode: the vectors are used IN PLACE in the state block (vector v at st + v*RH_SLOTS*64, element i at [i]); only the
// ring buffer and the scalars travel through registers.  Layout: vectors | ring | NUTS checkpoints | pool | scalars.
RH_DEV void rh_chain_store(const rh_chain &c, rh_u64 *st, const int lane) {
  size_t w = (size_t)RH_STATE_NVEC * RH_SLOTS;
  for (int k = 0; k < RH_RING_SLOTS; k++) st[(w++) * 64 + lane] = (rh_u64)__double_as_longlong(c.ring[k]);
  w += (size_t)(RH_STATE_NCK + RH_STATE_NPOOL) * RH_SLOTS;
  if (lane == 0) {
    rh_u64 *sc = st + w * 64;
    int j = 0;
#define X(n) sc[j++] = (rh_u64)__double_as_longlong(c.n);
    RH_STATE_F64(X)
#undef X
#define X(n) sc[j++] = (rh_u64)(rh_i64)c.n;
    RH_STATE_INT(X)
#undef X
#define X(n) sc[j++] = (rh_u64)c.n;
    RH_STATE_I64(X)
#undef X
  }
}
RH_DEV void rh_chain_load(rh_chain &c, rh_u64 *st, const int lane) {
  size_t w = 0;
#define X(n) c.n.s.p = (double *)(st + w * 64); w += RH_SLOTS;
  RH_STATE_VECS(X)
#undef X
  for (int k = 0; k < RH_RING_SLOTS; k++) c.ring[k] = __longlong_as_double((rh_i64)st[(w++) * 64 + lane]);
#if RH_WITH_NUTS
  for (int j = 0; j < RH_NUTS_MAXD; j++) {
    c.ckr[j].s.p = (double *)(st + w * 64); w += RH_SLOTS;
    c.ckrs[j].s.p = (double *)(st + w * 64); w += RH_SLOTS;
  }
#endif
  rh_pool_base = (double *)(st + w * 64); w += (size_t)RH_STATE_NPOOL * RH_SLOTS;
#if RH_BIGTH
  rh_tot_base = rh_pool_base + (size_t)RH_POOL_VECS * RH_SLOTS * 64;
#endif
  rh_pool_depth[lane] = 0;
  const rh_u64 *sc = st + w * 64;
  int j = 0;
#define X(n) c.n = __longlong_as_double((rh_i64)sc[j++]);
  RH_STATE_F64(X)
#undef X
#define X(n) c.n = rh_uniform_i((int)(rh_i64)sc[j++]);
  RH_STATE_INT(X)
#undef X
#define X(n) c.n = (rh_i64)sc[j++];
  RH_STATE_I64(X)
#undef X
}
#else
// image of chain c starts at st = state + c * RH_STATE_U64: [vector slot w][lane] ..., [ring slot][lane] ..., scalars
RH_DEV void rh_chain_store(const rh_chain &c, rh_u64 *st, const int lane) {
  int w = 0;
#define X(n) \
  for (int k = 0; k < RH_SLOTS; k++) st[(w++) * 64 + lane] = (rh_u64)__double_as_longlong(c.n.s[k]);
  RH_STATE_VECS(X)
#undef X
#if RH_PACK_L == 64
  for (int k = 0; k < RH_RING_SLOTS; k++) st[(w++) * 64 + lane] = (rh_u64)__double_as_longlong(c.ring[k]);
#else
  w += RH_RING_SLOTS;
#endif
#if RH_WITH_NUTS
  w += 2 * RH_NUTS_MAXD * RH_SLOTS;  // the checkpoints live in the image (c.ck)
#endif
#if RH_WITH_DENSE
  for (int j = 0; j < RH_NVARS; j++) st[(w++) * 64 + lane] = (rh_u64)__double_as_longlong(c.Drow[j]);
  for (int j = 0; j < RH_NVARS; j++) st[(w++) * 64 + lane] = (rh_u64)__double_as_longlong(c.Lrow[j]);
  for (int j = 0; j < RH_NVARS; j++) st[(w++) * 64 + lane] = (rh_u64)__double_as_longlong(c.Crow[j]);
#endif
  if (lane == 0) {
    rh_u64 *sc = st + (size_t)w * 64;
    int j = 0;
#define X(n) sc[j++] = (rh_u64)__double_as_longlong(c.n);
    RH_STATE_F64(X)
#undef X
#define X(n) sc[j++] = (rh_u64)(rh_i64)c.n;
    RH_STATE_INT(X)
#undef X
#define X(n) sc[j++] = (rh_u64)c.n;
    RH_STATE_I64(X)
#undef X
  }
}
RH_DEV void rh_chain_load(rh_chain &c, rh_u64 *st, const int lane) {
  int w = 0;
#define X(n) \
  for (int k = 0; k < RH_SLOTS; k++) c.n.s[k] = __longlong_as_double((rh_i64)st[(w++) * 64 + lane]);
  RH_STATE_VECS(X)
#undef X
#if RH_PACK_L == 64
  for (int k = 0; k < RH_RING_SLOTS; k++) c.ring[k] = __longlong_as_double((rh_i64)st[(w++) * 64 + lane]);
#else
  c.ring = (double *)(st + (size_t)w * 64); w += RH_RING_SLOTS;
#endif
#if RH_WITH_NUTS
  c.ck = st + (size_t)w * 64; w += 2 * RH_NUTS_MAXD * RH_SLOTS;
#endif
#if RH_WITH_DENSE
  for (int j = 0; j < RH_NVARS; j++) c.Drow[j] = __longlong_as_double((rh_i64)st[(w++) * 64 + lane]);
  for (int j = 0; j < RH_NVARS; j++) c.Lrow[j] = __longlong_as_double((rh_i64)st[(w++) * 64 + lane]);
  for (int j = 0; j < RH_NVARS; j++) c.Crow[j] = __longlong_as_double((rh_i64)st[(w++) * 64 + lane]);
#endif
  const rh_u64 *sc = st + (size_t)w * 64;
  int j = 0;
#define X(n) c.n = __longlong_as_double((rh_i64)sc[j++]);
  RH_STATE_F64(X)
#undef X
#define X(n) c.n = rh_uniform_i((int)(rh_i64)sc[j++]);
  RH_STATE_INT(X)
#undef X
#define X(n) c.n = (rh_i64)sc[j++];
  RH_STATE_I64(X)
#undef X
}
#endif  // RH_BIGN
RH_DEV void rh_chain_zero(rh_u64 *st, const int lane) {
  for (int w = lane; w < RH_STATE_U64; w += RH_LANES) st[w] = 0;
}

// ---- automaton states ------------------------------------------------------------------------------
enum {
  RH_S_INIT = 0, RH_S_INIT2, RH_S_TRY_BEGIN, RH_S_TRY_END, RH_S_WARM_SETUP, RH_S_ITER_TOP, RH_S_COUNT_LOOP,
  RH_S_COUNT_AFTER, RH_S_COUNT_DONE, RH_S_TS_BEGIN, RH_S_TS_MID, RH_S_FINISH, RH_S_DONE,
  RH_S_NUTS_TOP, RH_S_NUTS_DOUBLE, RH_S_NUTS_LEAF, RH_S_NUTS_LEAF2, RH_S_NUTS_DONE
};

RH_DEV rh_rng rh_rng_of(const rh_chain &c) { rh_rng r; r.seed = (rh_u64)c.rng_seed; r.have = c.rng_have; r.nn = c.rng_nn; return r; }
RH_DEV void rh_rng_put(rh_chain &c, const rh_rng &r) { c.rng_seed = (rh_i64)r.seed; c.rng_have = (int)r.have; c.rng_nn = r.nn; }

// n gaussians in ascending coordinate order (LeapFrog.scala:105-110, 231-236)
RH_DEV void rh_fill_normal(rh_chain &c, wvec &v, const int lane) {
  rh_rng r = rh_rng_of(c);
  wv_zero(v);
#if !RH_BIGN && RH_SLOTS == 1 && RH_NVARS <= RH_LANES
  // The stream is sequential, the expensive part of nextGaussian is not: walk the LCG and the polar method's rejection
  // loop for all pairs first (cheap, wave-uniform), park accepted pair j in lane j, evaluate sqrt(-2 log(s) / s) for all
  // pairs at once, then hand element e its half of pair (e - first) / 2.  Same operations per value: bit-identical.
  int first = 0;
  double out = 0.0;
  if (r.have) { out = (lane == 0) ? r.nn : out; r.have = 0; first = 1; }   // the pending nextNextGaussian comes first
  const int npairs = (RH_NVARS - first + 1) / 2;
  double pv1 = 0.0, pv2 = 0.0, ps = 0.5;
  for (int j = 0; j < npairs; j++) {
    double v1, v2, sq;
    do {
      v1 = 2 * rh_rng_uniform(r) - 1;
      v2 = 2 * rh_rng_uniform(r) - 1;
      sq = v1 * v1 + v2 * v2;
    } while (sq >= 1 || sq == 0);
    pv1 = (lane == j) ? v1 : pv1; pv2 = (lane == j) ? v2 : pv2; ps = (lane == j) ? sq : ps;
  }
  const double multiplier = rh_strict_sqrt(-2 * rh_strict_log(ps) / ps);
  const double g1 = pv1 * multiplier, g2 = pv2 * multiplier;
  const int rel = lane - first, src = rel >= 0 ? (rel >> 1) : 0;
  const double a = rh_gather(g1, src), b = rh_gather(g2, src);
  if (rel >= 0 && lane < RH_NVARS) out = (rel & 1) ? b : a;
  if ((RH_NVARS - first) & 1) { r.nn = rh_readlane(g2, npairs - 1); r.have = 1; }  // an odd count leaves one value pending
  v.s[0] = (lane < RH_NVARS) ? out : 0.0;
#else
  for (int i = 0; i < RH_NVARS; i++) {
    const double g = rh_rng_normal(r);
    wv_set(v, i, g, lane);
  }
#endif
  rh_rng_put(c, r);
}
// velocity (LeapFrog.scala:202-216): Identity -> p, Diagonal -> p * elements
RH_DEV void rh_velocity(const rh_chain &c, const wvec &p, wvec &out, const bool identity) {
#if RH_WITH_DENSE
  if (!identity && c.mass_dense) { // DenseMassMatrix.squareMultiply (MassMatrix.scala:34-48): out(i) = sum_j vector(j) * matrix(i*n+j)
    double y = 0.0;
    for (int j = 0; j < RH_NVARS; j++) y += rh_readlane(p.s[0], j) * c.Drow[j];
    out.s[0] = ((int)(threadIdx.x & (RH_LANES - 1)) < RH_NVARS) ? y : 0.0;
    return;
  }
#endif
  if (identity) out = p; else wv_mul(out, p, c.M);
}
// energy (LeapFrog.scala:131-136)
RH_DEV double rh_energy(const rh_chain &c, const wvec &p, const double U, const bool identity) {
#if RH_BIGN
  // big mode: velocity, product and sum in ONE pass over p (and M) instead of two temporaries written to and read back from HBM;
  // the same per-element arithmetic and the same summation order as the composition below (lane partials over ascending slots)
  double part = 0.0;
  _Pragma("unroll 1") for (int k = 0; k < RH_SLOTS; k += RH_BIGU) {
    double pa[RH_BIGU], ma[RH_BIGU];
    _Pragma("unroll") for (int j = 0; j < RH_BIGU; j++) {
      const int kc = k + j < RH_SLOTS ? k + j : RH_SLOTS - 1;
      pa[j] = p.s[kc]; ma[j] = identity ? 1.0 : c.M.s[kc];
    }
    _Pragma("unroll") for (int j = 0; j < RH_BIGU; j++) {
      const double v = identity ? pa[j] : pa[j] * ma[j];
      part += ((k + j) * 64 + (int)threadIdx.x < RH_NVARS) ? v * pa[j] : 0.0;
    }
  }
  return U + rh_wave_sum(part) / 2.0;
#else
  RH_TMP(v); RH_TMP(pr);
  rh_velocity(c, p, v, identity);
  wv_mul(pr, v, p);
  const double kinetic = wv_sum_seq(pr) / 2.0;
  return U + kinetic;
#endif
}
RH_DEV double rh_log_accept(const double deltaH) { // LeapFrog.scala:138-142
  if (deltaH != deltaH) return -RH_INF;
  return (-deltaH) < 0.0 ? (-deltaH) : 0.0;
}
RH_DEV void rh_new_qs_e(rh_chain &c, const double e, const bool identity) { // LeapFrog.scala:144-151, signed step
  RH_TMP(v);
  rh_velocity(c, c.Bp, v, identity);
  wv_axpy(c.Bq, e, v);
  if (c.sampling_started) c.n_leapfrog += 1; else c.n_warm_leapfrog += 1;
}
RH_DEV void rh_new_qs(rh_chain &c, const bool identity) { // LeapFrog.scala:144-151
  RH_TMP(v);
  rh_velocity(c, c.Bp, v, identity);
  wv_axpy(c.Bq, c.eps, v);
  if (c.sampling_started) c.n_leapfrog += 1; else c.n_warm_leapfrog += 1;
}
RH_DEV void rh_copy_P_to_B(rh_chain &c) { c.Bp = c.Pp; c.Bq = c.Pq; c.Bg = c.Pg; c.BU = c.PU; }
RH_DEV void rh_copy_B_to_P(rh_chain &c) { c.Pp = c.Bp; c.Pq = c.Bq; c.Pg = c.Bg; c.PU = c.BU; }
// initializePs (LeapFrog.scala:229-251)
RH_DEV void rh_initialize_ps(rh_chain &c, const bool identity, const int lane) {
  RH_TMP(buf);
  rh_fill_normal(c, buf, lane);
#if RH_WITH_DENSE
  if (!identity && c.mass_dense) { // DenseMassMatrix.upperTriangularSolve(choleskyUpperTriangular, buf, params) (MassMatrix.scala:52-69)
    double out = 0.0;              // U[i][j] = L[j][i] = lane j's Lrow[i]; back substitution, dot over j descending
    for (int i = RH_NVARS - 1; i >= 0; i--) {
      double dot = 0.0;
      for (int j = RH_NVARS - 1; j > i; j--) dot += rh_readlane(out, j) * rh_readlane(c.Lrow[i], j);
      const double o = (rh_readlane(buf.s[0], i) - dot) / rh_readlane(c.Lrow[i], i);
      out = (lane == i) ? o : out;
    }
    c.Pp.s[0] = out;
    return;
  }
#endif
  if (identity) c.Pp = buf;
  else {
RH_UNROLL_SLOTS
    for (int k = 0; k < RH_SLOTS; k++) c.Pp.s[k] = (k * 64 + lane < RH_NVARS) ? buf.s[k] / c.SD.s[k] : 0.0;
  }
}
RH_DEV void rh_dualavg_new(rh_chain &c, const double stepSize) { // DualAvg.scala:80-89
  c.da_logEps = rh_strict_log(stepSize);
  c.da_logEpsBar = 0.0;
  c.da_avgErr = 0.0;
  c.da_iter = 0;
  c.da_mu = rh_strict_log(10 * stepSize);
}
RH_DEV void rh_dualavg_update(rh_chain &c, const double delta, const double logAcceptanceProb) { // DualAvg.scala:58-77
  const double newAcceptanceProb = rh_strict_exp(logAcceptanceProb);
  c.da_iter = c.da_iter + 1;
  const double it = (double)c.da_iter;
  const double avgErrorMultiplier = 1.0 / (it + 10);
  const double stepSizeMultiplier = rh_pow_neg075(it);
  c.da_avgErr = ((1.0 - avgErrorMultiplier) * c.da_avgErr + (avgErrorMultiplier * (delta - newAcceptanceProb)));
  c.da_logEps = (c.da_mu - (c.da_avgErr * rh_strict_sqrt(it) / 0.05));
  c.da_logEpsBar = (stepSizeMultiplier * c.da_logEps + (1.0 - stepSizeMultiplier) * c.da_logEpsBar);
}
// WindowedMassMatrixTuner.update + VarianceEstimator (MassMatrix.scala:147-164, MassMatrixEstimator.scala:60-101)
RH_DEV bool rh_mass_update(rh_chain &c, const rh_cfg_dev &cfg, const int lane) {
#if RH_WITH_DENSE
  if (cfg.mass_tuner == 3 /*RH_MASS_DENSE_WINDOWED: DenseMassMatrixTuner + CovarianceEstimator*/) {
    c.win_j += 1;
    if (c.win_j < cfg.mass_skip_first || (cfg.warmup - c.win_j) < cfg.mass_skip_last) return false;
    c.win_i += 1;
    c.ve_samples += 1;
    const double ns = (double)c.ve_samples;
    const double oldDiff = c.Pq.s[0] - c.ve_mean.s[0];   // VarianceEstimator.update (MassMatrixEstimator.scala:69-84)
    c.ve_mean.s[0] += (oldDiff / ns);
    const double newDiff = c.Pq.s[0] - c.ve_mean.s[0];
    c.ve_raw.s[0] += oldDiff * newDiff;
    for (int k = 0; k < RH_NVARS; k++) c.Crow[k] += newDiff * rh_readlane(oldDiff, k); // cov(j*n+k) += newDiff(j) * oldDiff(k)
    if (c.win_i != c.win_size) return false;
    c.win_i = 0;
    c.win_size = (int)(c.win_size * cfg.mass_expansion);
    const double z = (double)(c.ve_samples - 1);          // covariance = cov / (samples - 1); reset() keeps `samples`
    double zeros = 0.0;                                   // DenseMassMatrix: require(!elements.contains(0.0)) (MassMatrix.scala:16)
    for (int k = 0; k < RH_NVARS; k++) {
      c.Drow[k] = c.Crow[k] / z; c.Crow[k] = 0.0; c.Lrow[k] = 0.0;
      zeros += (lane < RH_NVARS && c.Drow[k] == 0.0) ? 1.0 : 0.0;
    }
    if (rh_wave_sum(zeros) > 0.0) c.err |= 2;
    c.ve_mean.s[0] = 0.0; c.ve_raw.s[0] = 0.0;
    // choleskyUpperTriangular (MassMatrix.scala:74-116): packed lower triangle, row by row
    for (int i = 0; i < RH_NVARS; i++)
      for (int k = 0; k <= i; k++) {
        double sum = 0.0;
        for (int j = 0; j < k; j++) sum += rh_readlane(c.Lrow[j], i) * rh_readlane(c.Lrow[j], k);
        const double x = rh_readlane(c.Drow[k], i) - sum;
        const double val = (i == k) ? rh_strict_sqrt(x) : (1.0 / rh_readlane(c.Lrow[k], k) * x);
        c.Lrow[k] = (lane == i) ? val : c.Lrow[k];
      }
    double dg = 0.0; // diagonal, for reporting through M / mass_diag
    for (int k = 0; k < RH_NVARS; k++) dg = (lane == k) ? c.Drow[k] : dg;
    c.M.s[0] = (lane < RH_NVARS) ? dg : 1.0;
    c.SD.s[0] = (lane < RH_NVARS) ? rh_strict_sqrt(dg) : 1.0;
    c.mass_identity = 0; c.mass_dense = 1;
    return true;
  }
#endif
  if (cfg.mass_tuner != 1 /*RH_MASS_DIAG_WINDOWED*/) return false;
  c.win_j += 1;
  if (c.win_j < cfg.mass_skip_first || (cfg.warmup - c.win_j) < cfg.mass_skip_last) return false;
  c.win_i += 1;
  c.ve_samples += 1;
  const double ns = (double)c.ve_samples;
RH_UNROLL_SLOTS
  for (int k = 0; k < RH_SLOTS; k++) {
    const double oldDiff = c.Pq.s[k] - c.ve_mean.s[k];
    c.ve_mean.s[k] += (oldDiff / ns);
    const double newDiff = c.Pq.s[k] - c.ve_mean.s[k];
    c.ve_raw.s[k] += oldDiff * newDiff;
  }
  if (c.win_i == c.win_size) {
    c.win_i = 0;
    c.win_size = (int)(c.win_size * cfg.mass_expansion);
RH_UNROLL_SLOTS
    for (int k = 0; k < RH_SLOTS; k++) {
      const bool live = (k * 64 + lane < RH_NVARS);
      c.M.s[k] = live ? c.ve_raw.s[k] / ns : 1.0;         // variance = raw / samples (population form)
      c.SD.s[k] = live ? rh_strict_sqrt(c.M.s[k]) : 1.0;  // DiagonalMassMatrix.stdDevs
      c.ve_mean.s[k] = 0.0;                               // reset() zeroes mean and raw, NOT samples
      c.ve_raw.s[k] = 0.0;
    }
    { // DiagonalMassMatrix: require(!elements.contains(0.0)) (MassMatrix.scala:8) -- the reference throws; here the chain is
      // flagged (rh_chain_stats.error = RH_E_INVALID) and the call returns that status
      double zeros = 0.0;
RH_UNROLL_SLOTS
      for (int k = 0; k < RH_SLOTS; k++) zeros += ((k * 64 + lane < RH_NVARS) && c.M.s[k] == 0.0) ? 1.0 : 0.0;
      if (rh_wave_sum(zeros) > 0.0) c.err |= 2;
    }
    c.mass_identity = 0;
    return true;
  }
  return false;
}
RH_DEV void rh_ring_add(rh_chain &c, const int size, const double v, const int lane) { // Stats.scala:24-30
  c.ring_i += 1;
  if (c.ring_i == size) c.ring_full = 1;
  c.ring_i = c.ring_i % size;
#if RH_PACK_L == 64
#pragma unroll
  for (int k = 0; k < RH_RING_SLOTS; k++) c.ring[k] = (k * 64 + lane == c.ring_i) ? v : c.ring[k];
#else
  c.ring[c.ring_i] = v;  // every lane of the chain writes the same value: each later reads back its own store
#endif
}
RH_DEV double rh_ring_sample(rh_chain &c, const int size) { // Stats.scala:40-45
  rh_rng r = rh_rng_of(c);
  const int idx = rh_uniform_i(c.ring_full ? rh_rng_int(r, size) : rh_rng_int(r, c.ring_i + 1));
  rh_rng_put(c, r);
#if RH_PACK_L == 64
  double out = 0.0;
#pragma unroll
  for (int k = 0; k < RH_RING_SLOTS; k++)
    if ((idx >> 6) == k) out = rh_readlane(c.ring[k], idx & 63);
  return out;
#else
  return c.ring[idx];
#endif
}
RH_DEV bool rh_is_uturn(const rh_chain &c) { // LeapFrog.scala:35-47
#if RH_BIGN
  double part = 0.0;  // one pass, no temporaries (see rh_energy)
  _Pragma("unroll 1") for (int k = 0; k < RH_SLOTS; k += RH_BIGU) {
    double qa[RH_BIGU], q0[RH_BIGU], pa[RH_BIGU];
    _Pragma("unroll") for (int j = 0; j < RH_BIGU; j++) {
      const int kc = k + j < RH_SLOTS ? k + j : RH_SLOTS - 1;
      qa[j] = c.Bq.s[kc]; q0[j] = c.Pq.s[kc]; pa[j] = c.Bp.s[kc];
    }
    _Pragma("unroll") for (int j = 0; j < RH_BIGU; j++)
      part += ((k + j) * 64 + (int)threadIdx.x < RH_NVARS) ? (qa[j] - q0[j]) * pa[j] : 0.0;
  }
  const double out = rh_wave_sum(part);
#else
  RH_TMP(dq); RH_TMP(pr);
  wv_sub(dq, c.Bq, c.Pq);
  wv_mul(pr, dq, c.Bp);
  const double out = wv_sum_seq(pr);
#endif
  if (out != out) return true;
  return out < 0;
}

// Driver.warmup / collectSamples bookkeeping after one sampler iteration (Driver.scala:68-80, 104-108)
RH_DEV void rh_iteration_done(rh_chain &c, const rh_cfg_dev &cfg, const double a, const bool accept, const double acc_prob,
                              double *draws, const int lane) {
  if (c.it < cfg.warmup) {
    if (cfg.step_tuner == 0) { rh_dualavg_update(c, cfg.dualavg_delta, a); c.eps = rh_strict_exp(c.da_logEps); }
    if (rh_mass_update(c, cfg, lane)) {
      if (cfg.step_tuner == 0) { // stepSizeTuner.reset() (DualAvg.scala:17-21)
        const double ss = rh_strict_exp(c.da_logEpsBar);
        rh_dualavg_new(c, ss);
        c.eps = ss;
      }
    }
  } else {
    c.n_accept += accept ? 1 : 0;
    c.sum_accept += acc_prob;
    c.n_samp_iters += 1;
    double *out = draws + (size_t)(c.it - cfg.warmup) * RH_NVARS;
RH_UNROLL_SLOTS
    for (int k = 0; k < RH_SLOTS; k++)
      if (k * 64 + lane < RH_NVARS) out[k * 64 + lane] = c.Pq.s[k];
  }
  c.it += 1;
  c.pc = RH_S_ITER_TOP;
}
#if RH_WITH_NUTS
RH_DEV double rh_logaddexp(const double a, const double b) {
  const double m = a > b ? a : b;
  if (m == -RH_INF) return -RH_INF;
  return m + rh_strict_log(rh_strict_exp(a - m) + rh_strict_exp(b - m));
}
// v = M^-1 r; r_sum' = r_sum - (r_left + r_right)/2; turning iff v_left . r_sum' <= 0 or v_right . r_sum' <= 0 (NaN: turning)
RH_DEV bool rh_nuts_is_turning(const rh_chain &c, const wvec &rl, const wvec &rr, const wvec &rsum, const bool identity) {
#if RH_BIGN
  double partl = 0.0, partr = 0.0;  // one pass over (rl, rr, rsum, M), no temporaries: 13 reads + 5 writes of 80 KB become 4 reads
  _Pragma("unroll 1") for (int k = 0; k < RH_SLOTS; k += RH_BIGU / 2) {
    double la[RH_BIGU / 2], ra[RH_BIGU / 2], sa[RH_BIGU / 2], ma[RH_BIGU / 2];
    _Pragma("unroll") for (int j = 0; j < RH_BIGU / 2; j++) {
      const int kc = k + j < RH_SLOTS ? k + j : RH_SLOTS - 1;
      la[j] = rl.s[kc]; ra[j] = rr.s[kc]; sa[j] = rsum.s[kc]; ma[j] = identity ? 1.0 : c.M.s[kc];
    }
    _Pragma("unroll") for (int j = 0; j < RH_BIGU / 2; j++) {
      const bool live = (k + j) * 64 + (int)threadIdx.x < RH_NVARS;
      const double adj = sa[j] - (la[j] + ra[j]) / 2.0;
      const double vl = identity ? la[j] : la[j] * ma[j], vr = identity ? ra[j] : ra[j] * ma[j];
      partl += live ? vl * adj : 0.0;
      partr += live ? vr * adj : 0.0;
    }
  }
  const double dl = rh_wave_sum(partl), dr = rh_wave_sum(partr);
  return !(dl > 0.0) || !(dr > 0.0);
#else
  RH_TMP(adj); RH_TMP(vl); RH_TMP(vr); RH_TMP(pl); RH_TMP(pr);
  wv_nuts_adj(adj, rsum, rl, rr);
  rh_velocity(c, rl, vl, identity);
  rh_velocity(c, rr, vr, identity);
  wv_mul(pl, vl, adj);
  wv_mul(pr, vr, adj);
  const double dl = wv_sum_seq(pl), dr = wv_sum_seq(pr);
  return !(dl > 0.0) || !(dr > 0.0);
#endif
}
#endif

// Consumes the pending gradient (c.pend_logp / c.pend_g, evaluated at c.Bq) if the automaton was waiting for
// one, and runs until it needs the next gradient (RH_ADV_NEED_GRAD), reaches iteration `it_stop`
// (RH_ADV_PAUSED) or finishes (RH_ADV_DONE).  All control flow is wave-uniform.
RH_DEV int rh_advance(rh_chain &c, const rh_cfg_dev &cfg, const int it_stop, const rh_i64 *seedrec,
                      const double *static_mass, double *draws, const int lane) {
  for (;;) {
    switch (c.pc) {
    case RH_S_INIT: { // LeapFrog.initialize (LeapFrog.scala:102-116), first half
      rh_rng r; rh_rng_init(r, seedrec[0]);
      { // a stream handed over mid-pair: its pending nextNextGaussian (NaN = none)
        const double nn = __longlong_as_double(seedrec[1]);
        if (nn == nn) { r.have = 1; r.nn = nn; }
      }
      rh_rng_put(c, r);
      wv_zero(c.Bp); wv_zero(c.Bg); c.BU = 0.0;
      rh_fill_normal(c, c.Bq, lane);
      wv_fill(c.M, 1.0, lane); wv_fill(c.SD, 1.0, lane); c.mass_identity = 1;
      c.pc = RH_S_INIT2; c.n_grad += 1;
      return RH_ADV_NEED_GRAD;
    }
    case RH_S_INIT2: {
      c.BU = c.pend_logp * -1; c.Bg = c.pend_g;
      rh_copy_B_to_P(c);
      rh_initialize_ps(c, true, lane);
      // sampler.initialize: nothing to do (EHMC's snapshot buffer is part of the state)
      if (cfg.step_tuner == 0 /*DualAvgTuner.initialize: findReasonableStepSize with IdentityMassMatrix*/) {
        c.eps = 1.0; c.find_first = 1; c.pc = RH_S_TRY_BEGIN;
      } else { c.eps = cfg.static_step; c.pc = RH_S_WARM_SETUP; }
      break;
    }
    case RH_S_TRY_BEGIN: { // LeapFrog.tryStepping (LeapFrog.scala:14-22) up to the gradient at the new q
      rh_copy_P_to_B(c);
      wv_axpy(c.Bp, c.eps / 2.0, c.Bg);
      rh_new_qs(c, true);
      c.pc = RH_S_TRY_END; c.n_grad += 1;
      return RH_ADV_NEED_GRAD;
    }
    case RH_S_TRY_END: { // DualAvg.scala:27-41
      c.BU = c.pend_logp * -1; c.Bg = c.pend_g;
      wv_axpy(c.Bp, c.eps / 2.0, c.Bg);
      const double a = rh_log_accept(rh_energy(c, c.Bp, c.BU, true) - rh_energy(c, c.Pp, c.PU, true));
      if (c.find_first) { c.exponent = (a > rh_strict_log(0.5)) ? 1.0 : -1.0; c.find_first = 0; }
      if (c.eps != 0.0 && (c.exponent * a > -c.exponent * rh_strict_log(2.0))) {
        c.eps *= (c.exponent > 0.0 ? 2.0 : 0.5);
        c.pc = RH_S_TRY_BEGIN;
      } else {
        rh_dualavg_new(c, c.eps);
        c.pc = RH_S_WARM_SETUP;
      }
      break;
    }
    case RH_S_WARM_SETUP: { // massMatrixTuner.initialize (MassMatrix.scala:139-143, Sampler.scala:47-50)
      if (cfg.mass_tuner == 2 /*StaticMassMatrix(DiagonalMassMatrix)*/) {
RH_UNROLL_SLOTS
        for (int k = 0; k < RH_SLOTS; k++) {
          const int i = k * 64 + lane;
          c.M.s[k] = i < RH_NVARS ? static_mass[i] : 1.0;
          c.SD.s[k] = rh_strict_sqrt(c.M.s[k]);
        }
        c.mass_identity = 0;
      }
      c.win_size = cfg.mass_init_window; c.win_i = 0; c.win_j = 0; c.ve_samples = 0;
      wv_zero(c.ve_mean); wv_zero(c.ve_raw);
      c.it = 0; c.pc = RH_S_ITER_TOP;
      break;
    }
    case RH_S_ITER_TOP: { // Driver.warmup / collectSamples loop heads (Driver.scala:67, 104)
      if (c.it >= cfg.warmup && !c.sampling_started) { // lf.resetStats(); stepSize = stepSizeTuner.stepSize
        if (cfg.step_tuner == 0) c.eps = rh_strict_exp(c.da_logEpsBar);
        c.sampling_started = 1; c.n_grad = 0;
        c.e_samples = 0; c.e_mean = 0.0; c.e_raw = 0.0; c.e_trans2 = 0.0;  // a fresh Stats (Stats.scala:3-17)
      }
      if (c.it >= cfg.warmup + cfg.iterations) { c.pc = RH_S_DONE; return RH_ADV_DONE; }
      if (c.it >= it_stop) return RH_ADV_PAUSED;
      // startIteration (LeapFrog.scala:52-59): prevH with the old momenta, fresh momenta, pqBuf := params
      c.prevH = rh_energy(c, c.Pp, c.PU, c.mass_identity != 0);
      rh_initialize_ps(c, c.mass_identity != 0, lane);
      rh_copy_P_to_B(c);
      if (cfg.sampler == 0) { // HMCSampler (HMC.scala:6-23)
        c.ts_l = cfg.hmc_steps; c.ret = RH_S_FINISH; c.pc = RH_S_TS_BEGIN;
      } else if (cfg.sampler == 2) { // NUTS (extension)
        c.pc = RH_S_NUTS_TOP;
      } else {                // EHMCSampler (EHMC.scala:15-30, 52-61)
        bool count = false;
        if (c.it < cfg.warmup) {
          if (!c.ring_full) count = true;
          else { rh_rng r = rh_rng_of(c); const double u = rh_rng_uniform(r); rh_rng_put(c, r); count = u < cfg.ehmc_p_count; }
        }
        if (count) { c.cnt_l = 0; c.pc = RH_S_COUNT_LOOP; }
        else { c.ts_l = rh_uniform_i((int)rh_ring_sample(c, cfg.ehmc_buf_size)); c.ret = RH_S_FINISH; c.pc = RH_S_TS_BEGIN; }
      }
      break;
    }
    case RH_S_COUNT_LOOP: { // EHMC.countSteps (EHMC.scala:32-50)
      if (!rh_is_uturn(c) && c.cnt_l < cfg.ehmc_max_steps) {
        c.cnt_l += 1; c.ts_l = 1; c.ret = RH_S_COUNT_AFTER; c.pc = RH_S_TS_BEGIN;
      } else if (c.cnt_l < cfg.ehmc_min_steps) {
        c.ts_l = cfg.ehmc_min_steps - c.cnt_l; c.ret = RH_S_COUNT_DONE; c.pc = RH_S_TS_BEGIN;
      } else {
        c.Bp = c.Sp; c.Bq = c.Sq; c.Bg = c.Sg; c.BU = c.SU; // lf.restore(buf)
        c.pc = RH_S_COUNT_DONE;
      }
      break;
    }
    case RH_S_COUNT_AFTER: {
      if (c.cnt_l == cfg.ehmc_min_steps) { c.Sp = c.Bp; c.Sq = c.Bq; c.Sg = c.Bg; c.SU = c.BU; } // lf.snapshot(buf)
      c.pc = RH_S_COUNT_LOOP;
      break;
    }
    case RH_S_COUNT_DONE: {
      rh_ring_add(c, cfg.ehmc_buf_size, (double)c.cnt_l, lane);
      c.pc = RH_S_FINISH;
      break;
    }
    case RH_S_TS_BEGIN: { // takeSteps: initialHalfThenFullStep (LeapFrog.scala:24-33, 153-163)
      wv_axpy(c.Bp, c.eps / 2.0, c.Bg);
      rh_new_qs(c, c.mass_identity != 0);
      c.ts_i = 1; c.pc = RH_S_TS_MID; c.n_grad += 1;
      return RH_ADV_NEED_GRAD;
    }
    case RH_S_TS_MID: {
      // INVARIANT (two of rh_tick_kernel's shortcuts rest on it): this state begins by overwriting BU and Bg from the pending
      // evaluation, and nothing reads either between two consecutive visits with ts_i < ts_l.  So (1) the big-mode gather fast path
      // need not store pend_g / Bg for a mid-trajectory step (tests/test_gpu_live_chains.py compares it with this general path), and
      // (2) the log-density of a mid-trajectory evaluation is never read: such requests are marked gradient-only (active == 2).
      c.BU = c.pend_logp * -1; c.Bg = c.pend_g;
      if (c.ts_i < c.ts_l) { // twoFullSteps (LeapFrog.scala:175-184): ONE p += eps * grad, then q
        wv_axpy(c.Bp, c.eps, c.Bg);
        rh_new_qs(c, c.mass_identity != 0);
        c.ts_i += 1; c.n_grad += 1;
        return RH_ADV_NEED_GRAD;
      }
      wv_axpy(c.Bp, c.eps / 2.0, c.Bg); // finalHalfStep
      c.pc = c.ret;
      break;
    }
    case RH_S_FINISH: { // finishIteration (LeapFrog.scala:61-82) + the Driver's per-iteration bookkeeping
      const bool ident = c.mass_identity != 0;
      const double startH = rh_energy(c, c.Pp, c.PU, ident);
      const double endH = rh_energy(c, c.Bp, c.BU, ident);
      const double a = rh_log_accept(endH - startH);
      rh_rng r = rh_rng_of(c);
      const double u = rh_rng_uniform(r);
      rh_rng_put(c, r);
      const bool accept = a > rh_strict_log(u);
      if (accept) rh_copy_B_to_P(c);
      { // stats.energyVariance.update(h); stats.energyTransitions2 += Math.pow(h - prevH, 2)   (LeapFrog.scala:68-74)
        const double h = accept ? endH : startH;
        c.e_samples += 1;
        const double oldDiff = h - c.e_mean;
        c.e_mean += oldDiff / (double)c.e_samples;
        const double newDiff = h - c.e_mean;
        c.e_raw += oldDiff * newDiff;
        c.e_trans2 += (h - c.prevH) * (h - c.prevH);
      }
      rh_iteration_done(c, cfg, a, accept, rh_strict_exp(a), draws, lane);
      break;
    }
#if RH_WITH_NUTS
    // ---- NUTS (extension; not in the reference).  Same algorithm statement as oracle/sampler.c nuts_iteration:
    // iterative multinomial NUTS (Phan, Pradhan, Jankowiak 2019, App. A).
    case RH_S_NUTS_TOP: {
      c.nH0 = rh_energy(c, c.Pp, c.PU, c.mass_identity != 0);
      c.NLq = c.Pq; c.NLp = c.Pp; c.NLg = c.Pg; c.NRq = c.Pq; c.NRp = c.Pp; c.NRg = c.Pg;
      c.Nrsum = c.Pp;
      c.n_tree_logw = 0.0; c.n_sum_acc = 0.0; c.n_leaf_total = 0; c.n_depth = 0;
      c.pc = RH_S_NUTS_DOUBLE;
      break;
    }
    case RH_S_NUTS_DOUBLE: {
      if (c.n_depth >= cfg.nuts_max_depth) { c.pc = RH_S_NUTS_DONE; break; }
      rh_rng r = rh_rng_of(c); const double u = rh_rng_uniform(r); rh_rng_put(c, r);
      c.n_right = rh_uniform_i(u > 0.5 ? 1 : 0);
      if (c.n_right) { c.Bq = c.NRq; c.Bp = c.NRp; c.Bg = c.NRg; } else { c.Bq = c.NLq; c.Bp = c.NLp; c.Bg = c.NLg; }
      c.n_sub_logw = -RH_INF; wv_zero(c.Sp); c.n_leaf = 0;
      c.pc = RH_S_NUTS_LEAF;
      break;
    }
    case RH_S_NUTS_LEAF: { // one leapfrog step in the chosen direction: takeSteps(1, +-eps)
      const double e = c.n_right ? c.eps : -c.eps;
      wv_axpy(c.Bp, e / 2.0, c.Bg);
      rh_new_qs_e(c, e, c.mass_identity != 0);
      c.pc = RH_S_NUTS_LEAF2; c.n_grad += 1;
      return RH_ADV_NEED_GRAD;
    }
    case RH_S_NUTS_LEAF2: {
      const bool ident = c.mass_identity != 0;
      const double e = c.n_right ? c.eps : -c.eps;
      c.BU = c.pend_logp * -1; c.Bg = c.pend_g;
      wv_axpy(c.Bp, e / 2.0, c.Bg);
      double delta = rh_energy(c, c.Bp, c.BU, ident) - c.nH0;
      if (delta != delta) delta = RH_INF;
      const double leaf_logw = -delta;
      const bool sub_div = delta > 1000.0;
      c.n_sum_acc += (delta <= 0.0) ? 1.0 : rh_strict_exp(-delta);
      c.n_leaf_total += 1;
      const double new_logw = rh_logaddexp(c.n_sub_logw, leaf_logw);
      rh_rng r = rh_rng_of(c); const double u = rh_rng_uniform(r); rh_rng_put(c, r);
      if (c.n_leaf == 0 || u < rh_strict_exp(leaf_logw - new_logw)) { c.Sq = c.Bq; c.Sg = c.Bg; c.SU = c.BU; }
      c.n_sub_logw = new_logw;
      wv_acc(c.Sp, c.Bp); // subtree momentum sum lives in Sp
      // checkpoints (NumPyro _leaf_idx_to_ckpt_idxs)
      int idx_max = 0;
      for (int x = c.n_leaf >> 1; x > 0; x >>= 1) idx_max += (x & 1);
      int nsub = 0;
      for (int x = c.n_leaf; (x & 1) != 0; x >>= 1) nsub++;
      const int idx_min = idx_max - nsub + 1;
      bool sub_turning = false;
      if ((c.n_leaf & 1) == 0) {
#if RH_BIGN
#pragma unroll
        for (int j = 0; j < RH_NUTS_MAXD; j++)
          if (j == idx_max) { c.ckr[j] = c.Bp; c.ckrs[j] = c.Sp; }
#else
RH_UNROLL_SLOTS
        for (int s2 = 0; s2 < RH_SLOTS; s2++) {
          c.ck[(size_t)((idx_max * RH_SLOTS + s2) * 2) * 64 + lane] = (rh_u64)__double_as_longlong(c.Bp.s[s2]);
          c.ck[(size_t)((idx_max * RH_SLOTS + s2) * 2 + 1) * 64 + lane] = (rh_u64)__double_as_longlong(c.Sp.s[s2]);
        }
#endif
      } else {
        for (int k = idx_max; k >= idx_min && !sub_turning; k--) {
#if RH_BIGN
          // checkpoint k is read IN PLACE (round 5): rk / rsk are views on the chain's state block -- checkpoint j sits 2 j vectors
          // behind checkpoint 0 (rh_chain_load) -- where round 4 zeroed two pool vectors and copied the checkpoint into them: four
          // 80 KB writes and two reads per U-turn check at cfg 5's size for values that are only read
          wvec rk, rsk;
          rk.s.p = c.ckr[0].s.p + (size_t)k * 2 * RH_SLOTS * 64;
          rsk.s.p = c.ckrs[0].s.p + (size_t)k * 2 * RH_SLOTS * 64;
#else
          RH_TMP(rk); RH_TMP(rsk);
RH_UNROLL_SLOTS
          for (int s2 = 0; s2 < RH_SLOTS; s2++) {
            rk.s[s2] = __longlong_as_double((rh_i64)c.ck[(size_t)((k * RH_SLOTS + s2) * 2) * 64 + lane]);
            rsk.s[s2] = __longlong_as_double((rh_i64)c.ck[(size_t)((k * RH_SLOTS + s2) * 2 + 1) * 64 + lane]);
          }
#endif
          RH_TMP(sub);
          wv_sub_add(sub, c.Sp, rsk, rk);
          sub_turning = rh_nuts_is_turning(c, rk, c.Bp, sub, ident);
        }
      }
      c.n_leaf += 1;
      if (sub_turning || sub_div) { c.pc = RH_S_NUTS_DONE; break; }
      if (c.n_leaf < (1 << c.n_depth)) { c.pc = RH_S_NUTS_LEAF; break; }
      { // merge the finished subtree: biased progressive sampling, then the U-turn check of the doubled tree
        rh_rng r2 = rh_rng_of(c); const double u2 = rh_rng_uniform(r2); rh_rng_put(c, r2);
        if (u2 < rh_strict_exp(c.n_sub_logw - c.n_tree_logw)) { c.Pq = c.Sq; c.Pg = c.Sg; c.PU = c.SU; }
        c.n_tree_logw = rh_logaddexp(c.n_tree_logw, c.n_sub_logw);
        wv_acc(c.Nrsum, c.Sp);
        if (c.n_right) { c.NRq = c.Bq; c.NRp = c.Bp; c.NRg = c.Bg; } else { c.NLq = c.Bq; c.NLp = c.Bp; c.NLg = c.Bg; }
        c.n_depth += 1;
        c.pc = rh_nuts_is_turning(c, c.NLp, c.NRp, c.Nrsum, ident) ? RH_S_NUTS_DONE : RH_S_NUTS_DOUBLE;
      }
      break;
    }
    case RH_S_NUTS_DONE: {
      const double mean_acc = c.n_leaf_total ? c.n_sum_acc / (double)c.n_leaf_total : 0.0;
      rh_iteration_done(c, cfg, rh_strict_log(mean_acc), true, mean_acc, draws, lane);
      break;
    }
#endif
    default:
      return RH_ADV_DONE;
    }
  }
}

RH_DEV void rh_stats_write(const rh_chain &c, rh_chain_stats_dev *out, const int status, const int lane) {
  if (lane == 0) {
    out->leapfrog_steps = c.n_leapfrog; out->warmup_leapfrog_steps = c.n_warm_leapfrog;
    out->gradient_evaluations = c.n_grad; out->accepted = c.n_accept;
    out->sum_accept_prob = c.sum_accept; out->step_size = c.eps;
    out->e_trans2 = c.e_trans2; out->e_raw = c.e_raw;
    out->sampling_iterations = c.n_samp_iters; out->error = c.err; out->status = status;
  }
}

#if !RH_HAS_GATHER
// ---- kernels ----------------------------------------------------------------------------------------
// One chain per wavefront (64-thread workgroup), the whole Driver loop on the device.  `fresh` starts the
// chains from their seeds; otherwise the state image is resumed.  The launch ends for a chain when it reaches
// iteration it_stop, finishes, or has spent max_ticks gradient evaluations (host relaunches until all paused).
#ifndef RH_CHAIN_WAVES
#define RH_CHAIN_WAVES 2  /* >= 2 wavefronts per SIMD: the automaton is latency-bound, a second wavefront hides it (cfg 3, 32768 chains: +47 %) */
#endif
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RH_CHAIN_WAVES)))
rh_chain_kernel(const rh_model_data d, const rh_cfg_dev cfg, rh_u64 *__restrict__ state,
                const rh_i64 *__restrict__ seeds, const double *__restrict__ static_mass,
                double *__restrict__ draws, rh_chain_stats_dev *__restrict__ stats, int *__restrict__ n_running,
                const int chains, const int it_stop, const int max_ticks, const int fresh) {
  rh_lk_init();
  const int chain = blockIdx.x * (64 / RH_LANES) + (int)(threadIdx.x / RH_LANES);  // RH_LANES lanes per chain (64 unless packed)
  const int lane = threadIdx.x & (RH_LANES - 1);
  if (chain >= chains) return;
  rh_u64 *st = state + (size_t)chain * RH_STATE_U64;
  rh_chain c;
  if (fresh) rh_chain_zero(st, lane);
  rh_chain_load(c, st, lane);
  double *my_draws = draws + (size_t)chain * cfg.iterations * RH_NVARS;
  const rh_i64 *seed = seeds + 2 * (size_t)chain;
  int ticks = 0, status;
  if (c.need_eval) {
    int err = 0;
    rh_density(c.Bq, d, lane, c.pend_logp, c.pend_g, err);
    c.err |= err; c.need_eval = 0; ticks = 1;
  }
  for (;;) {
    status = rh_advance(c, cfg, it_stop, seed, static_mass, my_draws, lane);
    if (status != RH_ADV_NEED_GRAD) break;
    if (ticks >= max_ticks) { c.need_eval = 1; break; }
    int err = 0;
    rh_density(c.Bq, d, lane, c.pend_logp, c.pend_g, err);
    c.err |= err;
    ticks += 1;
  }
  rh_chain_store(c, st, lane);
  rh_stats_write(c, stats + chain, status, lane);
  if (status == RH_ADV_NEED_GRAD && lane == 0) atomicAdd(n_running, 1);
}

// ---- engine 2: tick engine for data-heavy models ---------------------------------------------------------
// One "tick" = every chain that is waiting for a gradient gets one.  rh_grad_kernel streams the observation rows
// ONCE per group of RH_GRAD_K chains (row values in VGPRs, the K parameter vectors in SGPRs, K x (n+1) per-lane
// accumulators) for one contiguous row split, and writes per-split partial sums; rh_tick_kernel combines the
// partials in fixed split order (plus the data-free targets, in target order), feeds the result to the chain's
// automaton and publishes the next q.  The sampler automaton is the same rh_advance as in rh_chain_kernel.
#endif  // !RH_HAS_GATHER

#if RH_NROWTARGETS > 0
// ---- live chains (round 6) ------------------------------------------------------------------------------------------------------
// Under the reference's dynamic samplers (EHMCSampler is DefaultConfig's, sampler/Sampler.scala:17-27; NUTS is BASELINE's) the chains
// of a run finish their iterations at different launches: round 5's driver-timed legs served 46 % (cfg 2, EHMC), 32 % (cfg 4, NUTS)
// and 7 % (cfg 5, NUTS) of the chain slots they computed, because a chain group was fixed by chain id and a launch cost the same
// whatever the number of chains still waiting for a gradient.  Now rh_compact_kernel, after every tick, lists the chains that wait
// for a gradient in ascending chain id (`list`, `*nlive`); the gradient kernels take their chains from that list -- slot s of the
// launch is chain list[s], so chain groups / MFMA tiles are full -- and the tick runs one wavefront per listed chain.
//   Row splits.  The partial sums are indexed by CHAIN id: what a chain's sums are depends on the row split count only -- not on
// the other chains of its group, on the live count or on the slot it was served in; the combine adds the same `nsplit` values in
// the same order.  For the dynamic samplers the host chooses MORE, shorter splits than a lock-step launch needs (rh_sampler_create):
// a launch with few live chains then still spreads over the machine, and the workgroups of the slots past the live count -- the
// last ones of the grid -- return at once.  (A per-launch choice of splits per workgroup, with a loop over "fine" splits inside
// the kernels, was tried first: what the loop keeps alive across its iterations cost rh_grad_gather_kernel its third wavefront
// per SIMD, 166 -> 192..220 registers, for the same number of pipeline fills and wave reductions.)
RH_DEV int rh_live_count(const int *__restrict__ nlive, const int chains) {
  return nlive ? __builtin_amdgcn_readfirstlane(*nlive) : chains;
}
// the chain in slot `slot` of the launch (a slot past the live count reads the last live chain; its results are dropped)
RH_DEV int rh_live_chain(const int *__restrict__ list, const int slot, const int nl) {
  return list[slot < nl ? slot : nl - 1];   // (the host always passes a list to the gradient kernels: the identity when every chain is served)
}
// Gradient-only requests.  active[chain] is 1 when the chain's tick asked for (logp, gradient) and 2 when only the gradient will be
// read: a mid-trajectory step of takeSteps, whose potential the next step overwrites (sampler/LeapFrog.scala:175-184; rh_advance,
// RH_S_TS_MID with ts_i < ts_l).  A wavefront whose chains all sent 2 walks its rows through the generated row_g() / elem_g() --
// row() / elem() without what only the log-density needs (csrc/emit.cpp: value_only) -- and leaves zeros in those sums.
template <int T> struct rh_any_value_only { static constexpr bool v = rh_target<T>::HAS_VALUE_ONLY || rh_any_value_only<T + 1>::v; };
template <> struct rh_any_value_only<RH_NTARGETS> { static constexpr bool v = false; };
// active[chain] != 0 (the chain's last tick asked for a gradient) -> list of those chains, ascending, and their number.  One
// workgroup: 1024 flags per pass, wave ballots + a 16-entry carry through LDS; a launch of a few microseconds behind every tick.
extern "C" __global__ void __launch_bounds__(1024)
rh_compact_kernel(const int *__restrict__ active, int *__restrict__ list, int *__restrict__ nlive, int *__restrict__ log, const int chains) {
  __shared__ int wcount[16];
  __shared__ int base;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) base = 0;
  __syncthreads();
  for (int c0 = 0; c0 < chains; c0 += 1024) {
    const int c = c0 + tid;
    const bool a = c < chains && active[c] != 0;
    const unsigned long long bal = __ballot(a);
    if (lane == 0) wcount[wave] = __popcll(bal);
    __syncthreads();
    int off = base;
    for (int w = 0; w < wave; w++) off += wcount[w];
    if (a) list[off + __popcll(bal & ((1ull << lane) - 1ull))] = c;
    __syncthreads();
    if (tid == 0) { int t = 0; for (int w = 0; w < 16; w++) t += wcount[w]; base += t; }
    __syncthreads();
  }
  if (tid == 0) { *nlive = base; *log = base; }   // log: the host's per-launch record of the live counts (rh_timing.chain_slots / steady_*)
}

#if !RH_HAS_GATHER
#if RH_FP_CONTRACT
#pragma clang fp contract(fast)
#endif
// the K parameter vectors of a wavefront's chain group: register arrays (SGPR pairs), or -- RH_BIGTH -- K pointers into q
#if RH_BIGTH
typedef const double *rh_thk_t[RH_GRAD_K];
#define RH_THK(th, kk) (*reinterpret_cast<const double (*)[RH_NTH]>((th)[kk]))
#else
typedef double rh_thk_t[RH_GRAD_K][RH_NTH];
#define RH_THK(th, kk) (th)[kk]
#endif
// COHERENT: the partial sums are stored with agent-scope (write-through) stores -- rh_grad_fused_kernel, whose epilogue reads them
// from another XCD within the same launch
// row() or, for a gradient-only request of a target that has a value-only part, row_g()
template <class TG, bool NV, class TH, class INV, class CP, class ACC>
RH_DEV void rh_row(const TH &th, const INV &inv, const CP &c, ACC &acc, int &err) {
  if constexpr (NV || !TG::HAS_VALUE_ONLY) TG::row(th, inv, c, acc, err);
  else TG::row_g(th, inv, c, acc, err);
}
template <int T, bool COHERENT, bool NV = true>   // NV: the log-density is needed (row()); otherwise row_g()
RH_DEV void rh_grad_targets(const rh_thk_t &th, const rh_model_data &d, const int lane,
                            const int split, const int nsplit, const int *__restrict__ lp, const int nvalid, const int chains,
                            double *__restrict__ partial, int &err) {
  if constexpr (T < RH_NTARGETS) {
    typedef rh_target<T> TG;
    if constexpr (TG::HAS_ROWS) {
      constexpr int NC = TG::NCOLS, U = RH_GRAD_U, K = RH_GRAD_K;
      double inv[K][TG::NINV > 0 ? TG::NINV : 1];
#pragma unroll
      for (int kk = 0; kk < K; kk++) TG::invariants(RH_THK(th, kk), inv[kk], err);
      const long long n = d.nrows[T];
      const long long chunk = 64LL * U;
      const long long per = (((n + chunk - 1) / chunk) + nsplit - 1) / nsplit; // chunks per split
      long long r0 = (long long)split * per * chunk, r1 = r0 + per * chunk;
      if (r0 > n) r0 = n;
      if (r1 > n) r1 = n;
      const double *cp[NC];
#pragma unroll
      for (int j = 0; j < NC; j++) cp[j] = d.cols[TG::COL0 + j];
      constexpr int NA = TG::NACC > 0 ? TG::NACC : 1;
      double acc[K][NA];
#pragma unroll
      for (int kk = 0; kk < K; kk++)
RH_UNROLL_ACC
        for (int o = 0; o < NA; o++) acc[kk][o] = 0.0;
      long long kb = r0;   // wave-uniform base row: every loop below branches on scalars, the row code runs with all lanes active
#if RH_GRAD_PIPELINE == 2
      // rolling pipeline: tile u's registers are reloaded for the next chunk as soon as tile u has been consumed, so every load
      // has the other U-1 tiles' arithmetic to land behind and no second buffer is needed.  The chunk loop runs on the
      // wave-uniform base row (scalar branch, scalar base + lane offset addressing into the global address space).
      {
        typedef const double __attribute__((address_space(1))) *gcol_t;
        gcol_t gp[NC];
#pragma unroll
        for (int j = 0; j < NC; j++) gp[j] = (gcol_t)cp[j];
        if (kb + chunk <= r1) {
          double c[U][NC];
#pragma unroll
          for (int u = 0; u < U; u++)
#pragma unroll
            for (int j = 0; j < NC; j++) c[u][j] = gp[j][kb + 64LL * u + lane];
          for (;;) {
            const long long kn = kb + chunk;
            const bool more = kn + chunk <= r1;
            const long long kl = (more ? kn : kb) + lane;   // the last chunk reloads itself: no branch around the loads
#pragma unroll
            for (int u = 0; u < U; u++) {
#pragma unroll
              for (int kk = 0; kk < K; kk++) rh_row<TG, NV>(RH_THK(th, kk), inv[kk], c[u], acc[kk], err);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int j = 0; j < NC; j++) c[u][j] = gp[j][kl + 64LL * u];
              __builtin_amdgcn_sched_barrier(0);
            }
            kb = kn;
            if (!more) break;
          }
        }
      }
#elif RH_GRAD_PIPELINE
      // software pipeline: the loads of tile i+1 are in flight while tile i is consumed
      if (kb + chunk <= r1) {
        double cn[U][NC];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
          for (int j = 0; j < NC; j++) cn[u][j] = cp[j][kb + 64LL * u + lane];
        for (;;) {
          double c[U][NC];
#pragma unroll
          for (int u = 0; u < U; u++)
#pragma unroll
            for (int j = 0; j < NC; j++) c[u][j] = cn[u][j];
          const long long kn = kb + chunk;
          const bool more = kn + chunk <= r1;
          if (more) {
#pragma unroll
            for (int u = 0; u < U; u++)
#pragma unroll
              for (int j = 0; j < NC; j++) cn[u][j] = cp[j][kn + 64LL * u + lane];
          }
#pragma unroll
          for (int u = 0; u < U; u++)
#pragma unroll
            for (int kk = 0; kk < K; kk++) rh_row<TG, NV>(RH_THK(th, kk), inv[kk], c[u], acc[kk], err);
          kb = kn;
          if (!more) break;
        }
      }
#else
      for (; kb + chunk <= r1; kb += chunk) {
        double c[U][NC];
#pragma unroll
        for (int u = 0; u < U; u++)
RH_UNROLL_ACC
          for (int j = 0; j < NC; j++) c[u][j] = cp[j][kb + 64LL * u + lane];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
          for (int kk = 0; kk < K; kk++) rh_row<TG, NV>(RH_THK(th, kk), inv[kk], c[u], acc[kk], err);
      }
#endif
      // the ragged end of the last split: still a scalar loop; a lane past the end re-reads the last row and a select drops its
      // contributions (see rh_rows_ragged) -- no divergent region around the row code
      for (; kb < r1; kb += 64) {
        const long long k = kb + lane;
        const bool live = k < r1;
        const long long kc = live ? k : r1 - 1;
        double c[NC];
RH_UNROLL_ACC
        for (int j = 0; j < NC; j++) c[j] = cp[j][kc];
#pragma unroll
        for (int kk = 0; kk < K; kk++) {
          double t[NA];
RH_UNROLL_ACC
          for (int o = 0; o < NA; o++) t[o] = -0.0;
          rh_row<TG, NV>(RH_THK(th, kk), inv[kk], c, t, err);
RH_UNROLL_ACC
          for (int o = 0; o < NA; o++) acc[kk][o] = live ? acc[kk][o] + t[o] : acc[kk][o];
        }
      }
#pragma unroll
      for (int kk = 0; kk < K; kk++) {
        rh_wave_sum_all(acc[kk]);   // a chain's NA sums level by level: their exchanges overlap
        // (the chain id of slot kk is looked up again here, a scalar load, rather than held in a scalar register through the row
        //  walk: the K parameter vectors already take 2 K nVars of them.  lp = the group's part of the list, nvalid of its K slots hold a chain)
        double *out = partial + (((size_t)TG::ROWT * nsplit + split) * chains + lp[kk < nvalid ? kk : 0]) * RH_NACC_MAX;
        if (lane == 0 && kk < nvalid) {
RH_UNROLL_ACC
          for (int o = 0; o < NA; o++) {
            if constexpr (COHERENT) __hip_atomic_store(out + o, acc[kk][o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else out[o] = acc[kk][o];
          }
        }
      }
    }
    rh_grad_targets<T + 1, COHERENT, NV>(th, d, lane, split, nsplit, lp, nvalid, chains, partial, err);
  }
}
#pragma clang fp contract(off)

// grid: 1-D, ngroups * nsplit single-wave workgroups.  XCD-aware mapping: workgroup b lands on XCD b % 8, so XCD x
// takes the row splits {x, x+8, ...} of EVERY chain group: each XCD's private L2 then serves only 1/8 of the rows.
#ifndef RH_GRAD_WAVES
#define RH_GRAD_WAVES 1
#endif
RH_DEV void rh_grad_map(const int b, const int nsplit, const int xcd_aware, int &split, int &group) {
  if (xcd_aware && (nsplit % 8) == 0) {
    const int xcd = b & 7, idx = b >> 3, spx = nsplit >> 3;
    split = xcd + 8 * (idx % spx);
    group = idx / spx;
  } else { split = b % nsplit; group = b / nsplit; }
}
// one workgroup of the plain gradient kernel: its chain group's parameters, then its row split
template <bool COHERENT>
RH_DEV void rh_grad_body(const rh_model_data &d, const double *q, const int *__restrict__ list, const int *__restrict__ nlive,
                         const int *__restrict__ vflag, double *partial, int *__restrict__ err_out, int *__restrict__ n_running,
                         const int chains, const int nsplit, const int xcd_aware) {
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  if (b == 0 && lane == 0) *n_running = 0; // re-armed for the tick kernel that follows in stream order
  const int nl = rh_live_count(nlive, chains);
  int split, group;
  rh_grad_map(b, nsplit, xcd_aware, split, group);
  if (group * RH_GRAD_K >= nl) return;   // (no live chain for this workgroup)
  rh_thk_t th;
  bool vfree = rh_any_value_only<0>::v && vflag != nullptr;   // every chain of the group asked for the gradient only
#pragma unroll
  for (int kk = 0; kk < RH_GRAD_K; kk++) {
    const int c = rh_live_chain(list, group * RH_GRAD_K + kk, nl);
    if constexpr (rh_any_value_only<0>::v) vfree = vfree && vflag[c] == 2;
#if RH_BIGTH
    th[kk] = q + (size_t)c * RH_NVARS;
#else
#pragma unroll
    for (int i = 0; i < RH_NVARS; i++) th[kk][i] = q[(size_t)c * RH_NVARS + i]; // wave-uniform address -> s_load
#endif
  }
  int err = 0;
  if constexpr (rh_any_value_only<0>::v) {
    if (vfree) rh_grad_targets<0, COHERENT, false>(th, d, lane, split, nsplit, list + group * RH_GRAD_K, nl - group * RH_GRAD_K, chains, partial, err);
    else rh_grad_targets<0, COHERENT, true>(th, d, lane, split, nsplit, list + group * RH_GRAD_K, nl - group * RH_GRAD_K, chains, partial, err);
  } else
  rh_grad_targets<0, COHERENT>(th, d, lane, split, nsplit, list + group * RH_GRAD_K, nl - group * RH_GRAD_K, chains, partial, err);
  if (err && lane == 0) atomicOr(err_out, 1);
}
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RH_GRAD_WAVES)))
rh_grad_kernel(const rh_model_data d, const double *__restrict__ q, const int *__restrict__ list, const int *__restrict__ nlive,
               const int *__restrict__ vflag, double *__restrict__ partial, int *__restrict__ err_out, int *__restrict__ n_running,
               const int chains, const int nsplit, const int xcd_aware) {
  rh_lk_init();
  rh_grad_body<false>(d, q, list, nlive, vflag, partial, err_out, n_running, chains, nsplit, xcd_aware);
}

// ---- dense GLM targets on the fp64 matrix cores ---------------------------------------------------------------
// rh_glm<T> (generated) describes a target whose row term is f(eta, other columns) with eta = X.theta a dense linear
// predictor over P >= 8 columns.  Both contractions run on v_mfma_f64_16x16x4_f64 with 16 chains per wavefront:
//     forward   eta[16 rows x 16 chains] = sum_ks  X[16 rows x 4 preds] . B[4 preds x 16 chains]
//     backward  G[16 preds x 16 chains] += X^T[16 preds x 4 rows] . w[4 rows x 16 chains]
// The f64 C/D layout (col = lane & 15, row = (lane >> 4) + 4 * reg) makes D register s of the forward product exactly
// the B operand of backward k-step s, so w never moves between lanes.  Only the scalar part elem(eta) -> (w, others)
// runs on the VALU.  Row tiles (64 rows x all columns) are staged through LDS once per workgroup of RH_GLM_W waves
// (= 16 * RH_GLM_W chains) with a padded column stride of 66 doubles (backward reads conflict-free, forward 2-way).
// The G accumulators cost 4 VGPR pairs per 16 predictors for 16 chains -- the VALU path needs P+1 pairs per chain.
#if defined(RH_GLM_TARGET) && !RH_GLM_SMALL   // (<= 8 predictors: the plain VALU kernel wins -- profiles/r1_c -- and nothing else is built)
#ifndef RH_GLM_W
#define RH_GLM_W 4
#endif
#define RH_GLM_TRP 66
typedef double rh_v4d __attribute__((ext_vector_type(4)));
template <bool B> struct rh_bool { static constexpr bool v = B; };

#if RH_FP_CONTRACT
#pragma clang fp contract(fast)
#endif
#ifndef RH_GLM_WAVES_PER_SIMD
#define RH_GLM_WAVES_PER_SIMD 2
#endif
extern "C" __global__ void __launch_bounds__(64 * RH_GLM_W, RH_GLM_WAVES_PER_SIMD)
rh_grad_glm_kernel(const rh_model_data d, const double *__restrict__ q, const int *__restrict__ list, const int *__restrict__ nlive,
                   const int *__restrict__ vflag, double *__restrict__ partial, int *__restrict__ err_out, int *__restrict__ n_running,
                   const int chains, const int nsplit, const int xcd_aware) {
  rh_lk_init();
  typedef rh_glm<RH_GLM_TARGET> GL;
  typedef rh_target<RH_GLM_TARGET> TG;
  constexpr int P = GL::P, NC = GL::NCOLS, W = RH_GLM_W;
  // A last, nearly empty predictor tile would cost 4 backward MFMAs (and a last forward k-step 1) per 16 rows for <= 4
  // predictors; the fp64 matrix pipe delivers only 0.66x the VALU FMA rate (profiles/r1_d), so those RV predictors are
  // contracted on the VALU instead: RV FMAs per eval forward + RV backward, from the same LDS reads (cfg 4: 29 -> 24 MFMAs).
  constexpr int RV = (P % 16 != 0 && P % 16 <= 4 && P > 16) ? P % 16 : 0;
  constexpr int PM = P - RV;          // predictors on the matrix pipe (a multiple of 16 when RV > 0)
  constexpr int PT = (PM + 3) / 4;    // forward k-steps (4 predictors each)
  constexpr int CT = (PM + 15) / 16;  // backward predictor tiles
  constexpr int MYC = (NC + W - 1) / W;
  // row tiles are double-buffered in LDS while two of them fit the 160 KB of a CU (<= 155 columns); wider models (<= 310
  // columns) keep one tile and pay a second barrier per tile.  The host sizes the dynamic LDS with the same rule.
  constexpr int NBUF = (2 * NC * RH_GLM_TRP * 8 + RH_LK_LDS * RH_LK_LDS_BYTES <= 160 * 1024) ? 2 : 1;   // (the link table, when the model has one)
  extern __shared__ __attribute__((aligned(16))) double rh_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 15, lg = lane >> 4;
  const int b = blockIdx.x;
  if (b == 0 && threadIdx.x == 0) *n_running = 0;
  // live chains: slot s of the launch is chain list[s]; a workgroup column serves 16 W consecutive slots
  const int nl = rh_live_count(nlive, chains);
  int bgroup, split;
  rh_grad_map(b, nsplit, xcd_aware, split, bgroup);
  if (bgroup * W * 16 >= nl) return;               // (uniform over the workgroup: none of its slots holds a live chain)
  const int slot0 = (bgroup * W + wave) * 16;
  const bool mine = slot0 + li < nl;                // this lane's slot holds a live chain
  const int cl = rh_live_chain(list, slot0 + li, nl);
  const bool compute = slot0 < nl;                  // (a wavefront without one still stages its columns of the tile)
  // gradient-only: every live chain of this wavefront's tile sent active == 2 (see rh_any_value_only)
  const bool vfree = TG::HAS_VALUE_ONLY && vflag != nullptr && !__any(mine && vflag[cl] != 2);
  // forward B operands: lane (li, lg) holds theta[chain li][pred 4 ks + lg]
  double Bf[PT];
  int acol[PT];
#pragma unroll
  for (int ks = 0; ks < PT; ks++) {
    const int pred = 4 * ks + lg;
    Bf[ks] = (pred < PM) ? GL::pred_scale[pred < PM ? pred : 0] * q[(size_t)cl * RH_NVARS + GL::pred_param[pred < PM ? pred : 0]] : 0.0;
    acol[ks] = (pred < PM) ? GL::pred_col[pred < PM ? pred : 0] : -2;
  }
  double thv[RV > 0 ? RV : 1], Gv[RV > 0 ? RV : 1];  // VALU remainder: this lane's chain's coefficients and gradient sums
  int vcol[RV > 0 ? RV : 1];
#pragma unroll
  for (int k = 0; k < RV; k++) { thv[k] = GL::pred_scale[PM + k] * q[(size_t)cl * RH_NVARS + GL::pred_param[PM + k]]; vcol[k] = GL::pred_col[PM + k]; Gv[k] = 0.0; }
  int bcol[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ct++) {
    const int pred = 16 * ct + li;
    bcol[ct] = (pred < PM) ? GL::pred_col[pred < PM ? pred : 0] : -2;
  }
  double thu[GL::NTHU > 0 ? GL::NTHU : 1];
#pragma unroll
  for (int k = 0; k < GL::NTHU; k++) thu[k] = q[(size_t)cl * RH_NVARS + GL::thu_param[k]];
  rh_v4d G[CT];
  double oth[GL::NOTHER > 0 ? GL::NOTHER : 1];
  int err = 0;
  const long long n = d.nrows[RH_GLM_TARGET];
  const long long per = (((n + 63) / 64) + nsplit - 1) / nsplit;
#pragma unroll
  for (int ct = 0; ct < CT; ct++) G[ct] = (rh_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int k = 0; k < GL::NOTHER; k++) oth[k] = 0.0;
  long long r0 = (long long)split * per * 64, r1 = r0 + per * 64;
  if (r0 > n) r0 = n;
  if (r1 > n) r1 = n;
  const long long ntiles = (r1 - r0 + 63) / 64;
  double stage[MYC];
  auto fetch = [&](long long tile) {
    long long row = r0 + tile * 64 + lane;
    if (row >= n) row = n - 1;
#pragma unroll
    for (int m = 0; m < MYC; m++) {
      const int j = wave + m * W;
      stage[m] = (j < NC && n > 0) ? d.cols[TG::COL0 + j][row] : 0.0;
    }
  };
  auto park = [&](int buf) {
#pragma unroll
    for (int m = 0; m < MYC; m++) {
      const int j = wave + m * W;
      if (j < NC) rh_lds[(buf * NC + j) * RH_GLM_TRP + lane] = stage[m];
    }
  };
  if (ntiles > 0) { fetch(0); park(0); }
  __syncthreads();
  for (long long t = 0; t < ntiles; t++) {
    const int buf = NBUF == 2 ? (int)(t & 1) : 0;
    if (t + 1 < ntiles) fetch(t + 1);
    if (compute) {
      const double *tile = rh_lds + (size_t)buf * NC * RH_GLM_TRP;
#pragma unroll 1
      for (int sub = 0; sub < 4; sub++) {
        const int row0s = sub * 16;
        // two independent accumulators halve the dependent-MFMA chain of the forward product
        rh_v4d D = (rh_v4d){0.0, 0.0, 0.0, 0.0}, D2 = (rh_v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < PT; ks++) {
          const double a = acol[ks] >= 0 ? tile[acol[ks] * RH_GLM_TRP + row0s + li] : (acol[ks] == -1 ? 1.0 : 0.0);
          if (ks & 1) D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bf[ks], D2, 0, 0, 0);
          else D = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Bf[ks], D, 0, 0, 0);
        }
        D += D2;
        double xv[RV > 0 ? RV : 1][4];  // remainder predictors at this lane's 4 rows
#pragma unroll
        for (int k = 0; k < RV; k++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            xv[k][r] = vcol[k] >= 0 ? tile[vcol[k] * RH_GLM_TRP + row0s + lg + 4 * r] : 1.0;
            D[r] += thv[k] * xv[k][r];
          }
        rh_v4d Wv;
#ifndef RH_GLM_ELEM_UNROLL
#define RH_GLM_ELEM_UNROLL 4
#endif
#ifndef RH_GLM_ELEM_INLOOP
#define RH_GLM_ELEM_INLOOP 0
#endif
        // full tiles of a wavefront whose 16 chains all exist need no validity masks (the common case by far)
        const bool full = (r0 + t * 64 + 64 <= r1) && (slot0 + 16 <= nl);
        // The four evaluations of a lane are ONE straight-line block per (gradient-only?, full?) combination, chosen by wave-uniform
        // branches AROUND the unrolled loop: with the choice inside it (round 6, first form) every evaluation became a basic block of
        // its own -- 8 blocks of ~65 instructions, each waiting for its own two table reads -- and the scheduler could no longer
        // interleave the four dependent fp64 chains (cfg 4: 17.25 -> 18.9 ms per value launch; profiles/r6_cfg4).
#if RH_GLM_ELEM_INLOOP   /* (RH_DIAG: RH_HIPRTC_EXTRA=-DRH_GLM_ELEM_INLOOP=1 -- the first form, kept for the A/B measurement) */
        if (full) {
#pragma unroll RH_GLM_ELEM_UNROLL
          for (int r = 0; r < 4; r++) {
            const int rrow = row0s + lg + 4 * r;
            double w = 0.0, o[GL::NOTHER > 0 ? GL::NOTHER : 1];
            if (vfree) GL::elem_g(thu, D[r], [&](int j) { return tile[j * RH_GLM_TRP + rrow]; }, w, o, err);
            else GL::elem(thu, D[r], [&](int j) { return tile[j * RH_GLM_TRP + rrow]; }, w, o, err);
            Wv[r] = w;
#pragma unroll
            for (int k = 0; k < GL::NOTHER; k++) oth[k] += o[k];
          }
        } else {
#pragma unroll RH_GLM_ELEM_UNROLL
          for (int r = 0; r < 4; r++) {
            const int rrow = row0s + lg + 4 * r;
            const bool valid = (r0 + t * 64 + rrow < r1) && mine;
            double w = 0.0, o[GL::NOTHER > 0 ? GL::NOTHER : 1];
            if (vfree) GL::elem_g(thu, D[r], [&](int j) { return tile[j * RH_GLM_TRP + rrow]; }, w, o, err);
            else GL::elem(thu, D[r], [&](int j) { return tile[j * RH_GLM_TRP + rrow]; }, w, o, err);
            Wv[r] = valid ? w : 0.0;
#pragma unroll
            for (int k = 0; k < GL::NOTHER; k++) oth[k] += valid ? o[k] : 0.0;
          }
        }
#else
        auto elems = [&](auto nv, auto fl) {
#pragma unroll RH_GLM_ELEM_UNROLL
          for (int r = 0; r < 4; r++) {
            const int rrow = row0s + lg + 4 * r;
            double w = 0.0, o[GL::NOTHER > 0 ? GL::NOTHER : 1];
            if constexpr (decltype(nv)::v) GL::elem(thu, D[r], [&](int j) { return tile[j * RH_GLM_TRP + rrow]; }, w, o, err);
            else GL::elem_g(thu, D[r], [&](int j) { return tile[j * RH_GLM_TRP + rrow]; }, w, o, err);
            if constexpr (decltype(fl)::v) {
              Wv[r] = w;
#pragma unroll
              for (int k = 0; k < GL::NOTHER; k++) oth[k] += o[k];
            } else {
              const bool valid = (r0 + t * 64 + rrow < r1) && mine;
              Wv[r] = valid ? w : 0.0;
#pragma unroll
              for (int k = 0; k < GL::NOTHER; k++) oth[k] += valid ? o[k] : 0.0;
            }
          }
        };
        if (full) {
          if (vfree) elems(rh_bool<false>{}, rh_bool<true>{});
          else elems(rh_bool<true>{}, rh_bool<true>{});
        } else {
          if (vfree) elems(rh_bool<false>{}, rh_bool<false>{});
          else elems(rh_bool<true>{}, rh_bool<false>{});
        }
#endif
#pragma unroll
        for (int k = 0; k < RV; k++)
#pragma unroll
          for (int r = 0; r < 4; r++) Gv[k] += xv[k][r] * Wv[r];
        // backward: k-step outermost, so that consecutive MFMAs write DIFFERENT accumulator tiles (a dependent MFMA would wait
        // for the previous one to drain)
#pragma unroll
        for (int sstep = 0; sstep < 4; sstep++)
#pragma unroll
          for (int ct = 0; ct < CT; ct++) {
            const int rr = row0s + 4 * sstep + lg;
            const double a = bcol[ct] >= 0 ? tile[bcol[ct] * RH_GLM_TRP + rr] : (bcol[ct] == -1 ? 1.0 : 0.0);
            G[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, Wv[sstep], G[ct], 0, 0, 0);
          }
      }
    }
    if (NBUF == 1) __syncthreads();  // every wave is done with the only tile before it is overwritten
    if (t + 1 < ntiles) park(NBUF == 2 ? (buf ^ 1) : 0);
    __syncthreads();
  }
  if (compute) {
    double *out = partial + (((size_t)TG::ROWT * nsplit + split) * chains + cl) * RH_NACC_MAX;
    // G[ct][r] at lane (li, lg) = sum_rows x[pred 16 ct + lg + 4 r] * w  for chain li
#pragma unroll
    for (int ct = 0; ct < CT; ct++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int pred = 16 * ct + lg + 4 * r;
        if (pred < PM && mine) out[GL::pred_acc[pred < PM ? pred : 0]] = G[ct][r];
      }
#pragma unroll
    for (int k = 0; k < RV; k++) {  // fold the 4 lane groups (rows lg + 4 r) of the VALU remainder
      double v = Gv[k];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lg == 0 && mine) out[GL::pred_acc[PM + k]] = v;
    }
#pragma unroll
    for (int k = 0; k < GL::NOTHER; k++) {
      double v = oth[k];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lg == 0 && mine) out[GL::other_acc[k]] = v;
    }
  }
  if (__any(err != 0) && lane == 0) atomicOr(err_out, 1);
}
#pragma clang fp contract(off)

#endif  // RH_GLM_TARGET

#else  // RH_HAS_GATHER
// ---- gather mode: a parameter table indexed by a data column (cfg 5: group effects alphas(site)) ------------------
// Reference semantics: Lookup(index, table of G parameter expressions) evaluates ALL table entries per row and the
// gradient adds one eq-lookup per entry per row (compute/Translator.scala:51-61, compute/Gradient.scala:148-152):
// O(rows x G).  Here rows are sorted by index, so a group's rows are contiguous: the table parameter reaches a row through one
// per-lane load, the row term is evaluated with it, and the scatter value (the common adjoint g of all the eq-lookups) is summed
// over the group's rows by a fixed-order segmented scan and stored once per (chain, group): no atomics, deterministic.  Shared
// outputs are accumulated per lane across the whole split exactly as in rh_grad_kernel.
#if RH_FP_CONTRACT
#pragma clang fp contract(fast)
#endif
// fixed-order SEGMENTED inclusive scan over the wavefront: lane l ends with the sum of v over the lanes [max(start_l, 0), l] of its
// own segment (start = first lane of the segment, <= 0 when the segment began in an earlier tile).  Hillis-Steele with ds_bpermute.
RH_DEV double rh_segmented_scan(double v, const int start, const int lane) {
  const int s0 = start > 0 ? start : 0;
#pragma unroll
  for (int dlt = 1; dlt < 64; dlt <<= 1) {
    const double t = __shfl_up(v, dlt, 64);
    v += (lane - dlt >= s0) ? t : 0.0;
  }
  return v;
}
template <class TG, bool NV, class TH, class INV, class CP, class ACC>
RH_DEV void rh_row_gz(const TH &th, const INV &inv, const CP &c, const double gz, ACC &acc, double &sv, int &err) {   // (rh_row in gather mode)
  if constexpr (NV || !TG::HAS_VALUE_ONLY) TG::row(th, inv, c, gz, acc, sv, err);
  else TG::row_g(th, inv, c, gz, acc, sv, err);
}
// One 64-row tile of a gather-mode target for the K chains of the wavefront (the walks that rh_gather_walk_a does not serve: groups
// of fewer than 64 rows -- a lifted prior has ONE row per group -- and RH_GATHER_V2=0).  Full tiles run the row code with every
// lane active.  The ragged last tile of a split runs it on every lane too -- a lane past the end on the split's last row -- into
// temporaries that start at -0.0 (x + -0.0 is x in every bit), one chain after the other, and a select keeps the live lanes'
// `acc + t`: the very addition the full tiles perform.  Until round 6 the ragged tile ran inside `if (live)`: generated code in a
// divergent region, the one construct whose join block this toolchain has been seen to get wrong -- and cfg 5's
// centred form (second row target: the lifted prior, one row per group, every split ragged) gave chains that depended on which
// other chains shared the wavefront (profiles/r6_live: compacted vs uncompacted launches, 21 of 1.14e6 leapfrog steps apart).
template <int T, bool NV, class INV, int NC, int NA>
RH_DEV void rh_gather_rows(const double (&th)[RH_GRAD_K][RH_NTH], const INV &inv, const double (&cc)[NC], const double (&gz)[RH_GRAD_K],
                           double (&acc)[RH_GRAD_K][NA], double (&sv)[RH_GRAD_K], int &err, const bool full, const bool live) {
  typedef rh_target<T> TG;
  if (full) {   // wave-uniform
#pragma unroll
    for (int kk = 0; kk < RH_GRAD_K; kk++) rh_row_gz<TG, NV>(th[kk], inv[kk], cc, gz[kk], acc[kk], sv[kk], err);
  } else {
#pragma unroll
    for (int kk = 0; kk < RH_GRAD_K; kk++) {
      double t[NA];
#pragma unroll
      for (int o = 0; o < NA; o++) t[o] = -0.0;
      double svt = -0.0;
      rh_row_gz<TG, NV>(th[kk], inv[kk], cc, gz[kk], t, svt, err);
#pragma unroll
      for (int o = 0; o < NA; o++) acc[kk][o] = live ? acc[kk][o] + t[o] : acc[kk][o];
      sv[kk] = live ? sv[kk] + svt : sv[kk];
    }
  }
}
// K per-lane values -> K wave sums with 2(K-1) + (6 - log2 K) exchanges instead of 6K: at the first log2(K) levels of the butterfly a
// lane keeps only half of its values and hands the other half to its partner (who keeps exactly those), afterwards one value per
// lane goes through the remaining levels.  Every addition is the butterfly's own `own + partner` at that level, in that order, so the
// sum of value kk is bit-identical to rh_wave_sum's -- it just ends up in the lanes whose top log2(K) lane bits spell kk (for K = 4:
// lanes 0-15 hold sum 0, 16-31 sum 1, ...), not in all of them.  Returns this lane's sum; `mine` = which value it is.
RH_DEV double rh_split_level(const double lo, const double hi, const bool up, const int off) {   // the upper lanes keep `hi`, the lower ones `lo`
  const double keep = up ? hi : lo, send = up ? lo : hi;
  return keep + __shfl_xor(send, off, 64);
}
template <int K>
RH_DEV double rh_wave_sum_split(const double (&v)[K], const int lane, int &mine) {
  static_assert(K == 1 || K == 2 || K == 4 || K == 8, "K is a power of two");
  double x;
  int off = 32;
  if constexpr (K == 8) {
    const bool u5 = (lane & 32) != 0, u4 = (lane & 16) != 0, u3 = (lane & 8) != 0;
    const double a0 = rh_split_level(v[0], v[4], u5, 32), a1 = rh_split_level(v[1], v[5], u5, 32);
    const double a2 = rh_split_level(v[2], v[6], u5, 32), a3 = rh_split_level(v[3], v[7], u5, 32);
    const double b0 = rh_split_level(a0, a2, u4, 16), b1 = rh_split_level(a1, a3, u4, 16);
    x = rh_split_level(b0, b1, u3, 8);
    mine = (u5 ? 4 : 0) + (u4 ? 2 : 0) + (u3 ? 1 : 0); off = 4;
  } else if constexpr (K == 4) {
    const bool u5 = (lane & 32) != 0, u4 = (lane & 16) != 0;
    const double a0 = rh_split_level(v[0], v[2], u5, 32), a1 = rh_split_level(v[1], v[3], u5, 32);
    x = rh_split_level(a0, a1, u4, 16);
    mine = (u5 ? 2 : 0) + (u4 ? 1 : 0); off = 8;
  } else if constexpr (K == 2) {
    const bool u5 = (lane & 32) != 0;
    x = rh_split_level(v[0], v[1], u5, 32);
    mine = u5 ? 1 : 0; off = 16;
  } else { x = v[0]; mine = 0; }
#pragma unroll
  for (; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
  return x;
}

#ifndef RH_GATHER_V2
#define RH_GATHER_V2 1
#endif
#if RH_GATHER_V2 && !(RH_GRAD_K == 1 || RH_GRAD_K == 2 || RH_GRAD_K == 4 || RH_GRAD_K == 8)
#undef RH_GATHER_V2
#define RH_GATHER_V2 0   /* (a caller-chosen odd K keeps round 4's walk) */
#endif
#if RH_GATHER_V2
// Round 5: the group-major walk of a gather-mode target whose non-empty groups all have >= 64 rows (cfg 5), rebuilt around what the
// machine code and the counters of round 4's loop showed (profiles/r5_cfg5): per tile it made TWO dependent trips to memory with
// nothing in flight (index column -> wait -> per-lane table reads + the other columns -> wait), through flat (generic address space)
// loads that also tick the LDS counter; it reduced K scatter sums with 6K cross-lane exchanges per group; ran every tile's row
// code inside an exec-masked region; and ~1 in 8 of its vector instructions was a v_readlane re-loading a wave-uniform operand --
// the K parameter vectors arrive as 8-register scalar tuples and the allocator spills and reloads whole tuples.  Here:
//   * the group structure is walked on the SCALAR unit from the group offsets (rows are sorted by group: a lane's row is in the
//     open group A iff row < end(A); A is complete iff end(A) <= end of the tile), so the index column is not read at all;
//   * the wave-uniform operands of the row code -- the K chains' shared parameters and invariants, and the table entries of A and
//     of the next non-empty group B (scalar loads issued a whole group ahead; constant address space: q is not written while this
//     kernel runs) -- live in LDS: a use is a broadcast ds_read on the LDS pipe instead of a scalar register (or its reload) on the
//     vector pipe, and a lane picks its table entry by ADDRESS (A's slot or B's), not by a select;
//   * the columns are loaded through the global address space in a two-tile rolling pipeline (a tile's registers are reloaded for
//     the tile after next as soon as it has been consumed), full tiles run with all lanes active, and the split's last, ragged
//     tile runs its row code on all lanes too, behind a select (no divergent region is left around generated code in this walk);
//   * a completed group's K sums are reduced by rh_wave_sum_split (same bits as the butterfly, 7 exchanges instead of 24 at K = 4).
// Same per-row arithmetic, same order of every sum: results are bit-identical to the old walk's.
template <int T> struct rh_ninv_max { static constexpr int v = rh_target<T>::NINV > rh_ninv_max<T + 1>::v ? rh_target<T>::NINV : rh_ninv_max<T + 1>::v; };
template <> struct rh_ninv_max<RH_NTARGETS> { static constexpr int v = 1; };
#define RH_GU_TH 0                                            /* [K][RH_NTH]   shared parameters of the K chains   */
#define RH_GU_INV (RH_GRAD_K * RH_NTH)                        /* [K][NINV max] invariants of the target being walked */
#define RH_GU_Z (RH_GU_INV + RH_GRAD_K * rh_ninv_max<0>::v)   /* [2][K]        table entries of group A, of group B  */
#define RH_GU_SIZE (RH_GU_Z + 2 * RH_GRAD_K)
// (one wavefront per workgroup: no barriers; every lane stores the same value to the same address and reads back through an offset
//  the compiler cannot see through -- rh_gu_oz() -- so that nothing is forwarded from the stores or hoisted back into registers)
RH_DEV int rh_gu_oz() { int z; asm volatile("s_mov_b32 %0, 0" : "=s"(z)); return z; }
template <int T, bool NV, class TH, class INV, int NC, int NA>
RH_DEV void rh_gather_walk_a(double *gu, const TH &th, const INV &inv, const double *const (&cp)[NC], double (&acc)[RH_GRAD_K][NA],
                             const rh_gather_data &gd, const double *__restrict__ q, const int lane, const int g0, const int g1,
                             const int r0, const int r1, const int (&cid)[RH_GRAD_K], const bool (&cok)[RH_GRAD_K], int &err) {
  typedef rh_target<T> TG;
  constexpr int K = RH_GRAD_K, NI = TG::NINV > 0 ? TG::NINV : 1;
  typedef const double __attribute__((address_space(1))) *gcol_t;
  typedef const int __attribute__((address_space(4))) *cint_t;
  typedef const double __attribute__((address_space(4))) *cdbl_t;
  if (r0 >= r1) return;
  gcol_t gp[NC];
#pragma unroll
  for (int j = 0; j < NC; j++) gp[j] = (gcol_t)cp[j];
  const cint_t go = (cint_t)gd.goff[TG::ROWT];
  const cdbl_t qc = (cdbl_t)q;
  size_t qoff[K];
#pragma unroll
  for (int kk = 0; kk < K; kk++) {
    qoff[kk] = (size_t)cid[kk] * RH_NVARS + TG::G_FIRST;
#pragma unroll
    for (int i = 0; i < RH_NTH; i++) gu[RH_GU_TH + kk * RH_NTH + i] = th[kk][i];
#pragma unroll
    for (int i = 0; i < (TG::NINV > 0 ? TG::NINV : 0); i++) gu[RH_GU_INV + kk * NI + i] = inv[kk][i];
  }
  // the chain whose group sums end up in this lane (rh_wave_sum_split: the lanes whose top log2 K lane bits spell kk hold sum kk)
  int cmine = cid[0];
  bool okmine = cok[0];
#pragma unroll
  for (int kk = 1; kk < K; kk++) { const bool me = lane / (64 / K) == kk; cmine = me ? cid[kk] : cmine; okmine = me ? cok[kk] : okmine; }
  // A = the open group, B = the next non-empty group of this split (g1 = none); all wave-uniform
  int gA = g0, gB = g1, endA = r1;
  double accA[K], accB[K];
#pragma unroll
  for (int kk = 0; kk < K; kk++) { accA[kk] = 0.0; accB[kk] = 0.0; gu[RH_GU_Z + kk] = 0.0; gu[RH_GU_Z + K + kk] = 0.0; }
  if constexpr (TG::HAS_GATHER) {
    while (gA < g1 - 1 && go[gA + 1] == go[gA]) gA++;
    endA = go[gA + 1];
    gB = gA + 1;
    while (gB < g1 && go[gB + 1] == go[gB]) gB++;
#pragma unroll
    for (int kk = 0; kk < K; kk++) {
      const double za = qc[qoff[kk] + gA];
      gu[RH_GU_Z + kk] = za;
      gu[RH_GU_Z + K + kk] = gB < g1 ? qc[qoff[kk] + gB] : za;
    }
  }
  const int rlast = r1 - 1;
  auto load_tile = [&](double (&c)[NC], const int tb) {   // rows tb + lane, clamped into the split (a tile past the end re-reads its last row)
    const int r = tb + lane;
    const int rr = r < rlast ? r : rlast;
#pragma unroll
    for (int j = 0; j < NC; j++) c[j] = gp[j][rr];
  };
  auto run_tile = [&](const double (&c)[NC], const bool inA, const bool ragged, const bool live) {   // one chain after the other: gz and sv are transient
    const double *u = gu + rh_gu_oz();
    const double *uz = u + RH_GU_Z + (inA ? 0 : K);
#pragma unroll
    for (int kk = 0; kk < K; kk++) {
      const double (&thk)[RH_NTH] = *(const double (*)[RH_NTH])(u + RH_GU_TH + kk * RH_NTH);
      const double *invk = u + RH_GU_INV + kk * NI;
      const double gz = uz[kk];
      double sv = 0.0;
      if (!ragged) rh_row_gz<TG, NV>(thk, invk, c, gz, acc[kk], sv, err);
      else {
        // the ragged tile without a divergent region (as rh_rows_ragged): every lane runs the row code -- a lane past the end on the
        // split's last row, which load_tile gave it -- into temporaries that start at -0.0 (x + -0.0 is x in every bit), and a
        // select keeps the live lanes' `acc + t`: the very addition the full tiles perform
        double t[NA];
#pragma unroll
        for (int o = 0; o < NA; o++) t[o] = -0.0;
        double svt = -0.0;
        rh_row_gz<TG, NV>(thk, invk, c, gz, t, svt, err);
#pragma unroll
        for (int o = 0; o < NA; o++) acc[kk][o] = live ? acc[kk][o] + t[o] : acc[kk][o];
        sv = live ? 0.0 + svt : 0.0;
      }
      if constexpr (TG::HAS_GATHER) { accA[kk] += inA ? sv : 0.0; accB[kk] += inA ? 0.0 : sv; }
    }
  };
  auto close_tile = [&](const int tend) {   // group A complete? (wave-uniform)
    if constexpr (TG::HAS_GATHER) {
      if (endA <= tend) {
        int mine;
        const double sum = rh_wave_sum_split(accA, lane, mine);
        (void)mine;
        if ((lane & (64 / K - 1)) == 0 && okmine) gd.sbuf[TG::ROWT][(size_t)cmine * TG::G_COUNT + gA] = sum;
        gA = gB;
        endA = gA < g1 ? go[gA + 1] : 0x7fffffff;
        if (gB < g1) gB++;
        while (gB < g1 && go[gB + 1] == go[gB]) gB++;
        const double *u = gu + rh_gu_oz();
#pragma unroll
        for (int kk = 0; kk < K; kk++) {
          accA[kk] = accB[kk]; accB[kk] = 0.0;
          const double zb = u[RH_GU_Z + K + kk];
          gu[RH_GU_Z + kk] = zb;                                           // B becomes A ...
          gu[RH_GU_Z + K + kk] = gB < g1 ? qc[qoff[kk] + gB] : zb;          // ... and the group after it B
        }
      }
    }
  };
  double c[2][NC];
  load_tile(c[0], r0);
  load_tile(c[1], r0 + 64);
  int tb = r0;
  for (;;) {
    bool done = false;
#pragma unroll
    for (int u = 0; u < 2; u++) {
      if (tb + 64 > r1) { done = true; break; }   // wave-uniform: what is left (if anything) is the ragged tile
      run_tile(c[u], tb + lane < endA, false, true);
      __builtin_amdgcn_sched_barrier(0);
      load_tile(c[u], tb + 128);
      __builtin_amdgcn_sched_barrier(0);
      close_tile(tb + 64);
      tb += 64;
    }
    if (done) break;
  }
  if (tb < r1) {   // the ragged last tile of the split (its rows are in the slot whose turn it is)
    const int u = ((tb - r0) >> 6) & 1;
    if (u == 0) run_tile(c[0], tb + lane < endA, true, tb + lane < r1);
    else run_tile(c[1], tb + lane < endA, true, tb + lane < r1);
    close_tile(r1);
  }
}
#endif  // RH_GATHER_V2

// SCAN: which row targets this instantiation walks.  false (rh_grad_gather_kernel): the targets without a gather and those whose non-empty
// groups all have >= 64 rows -- the group-major walk; true (rh_grad_gather_scan_kernel): the targets with smaller groups (a lifted prior
// has ONE row per group) -- the segmented scan over every tile.  Which of the two a target is, is a property of the DATA (gd.gmin),
// so the test is made at run time, wave-uniformly; splitting the two kinds of walk into two kernels keeps the generated source
// independent of the data AND keeps the scan walk's registers out of the kernel that cfg 5's 1e6 likelihood rows run through
// (together: 205 vector registers, two wavefronts per SIMD; the group-major walk alone: 166, three).
template <int T, bool NV, bool SCAN>
RH_DEV void rh_gather_targets(const double (&th)[RH_GRAD_K][RH_NTH], const rh_model_data &d, const rh_gather_data &gd,
                              const double *__restrict__ q, const int lane, const int split, const int nsplit,
                              const int (&cid)[RH_GRAD_K], const bool (&cok)[RH_GRAD_K], const int chains, double *gu_lds,
                              double *__restrict__ partial, int &err) {
  if constexpr (T < RH_NTARGETS) {
    typedef rh_target<T> TG;
    if constexpr (TG::HAS_ROWS) {
     if ((!TG::HAS_GATHER || gd.gmin[TG::ROWT] >= 64) != SCAN) {   // this kernel's kind of target
      constexpr int NC = TG::NCOLS, K = RH_GRAD_K;
      double inv[K][TG::NINV > 0 ? TG::NINV : 1];
#pragma unroll
      for (int kk = 0; kk < K; kk++) TG::invariants(th[kk], inv[kk], err);
      const double *cp[NC];
#pragma unroll
      for (int j = 0; j < NC; j++) cp[j] = d.cols[TG::COL0 + j];
      constexpr int NA = TG::NACC > 0 ? TG::NACC : 1;
      double acc[K][NA];
#pragma unroll
      for (int kk = 0; kk < K; kk++)
#pragma unroll
        for (int o = 0; o < NA; o++) acc[kk][o] = 0.0;
      // The split's rows are one contiguous range (splits are cut at group boundaries, rows are sorted by group): the wavefront
      // walks it in full 64-row tiles, ACROSS group boundaries -- every lane evaluates a row whatever the group sizes are
      // (cfg 5: 100 rows per group used to run as 64 + 36 lanes).  A lane takes its row's table entry with a per-lane load (a
      // tile touches one to three neighbouring entries per chain).  The scatter value is summed per group in one of two ways:
      // groups of >= 64 rows through two per-lane running sums and ONE wave reduction per group (below); smaller groups through a
      // segmented scan over every tile, a group that continues in the next tile handing its running sum over in `carry`.
      const int *goff = gd.goff[TG::ROWT];
      const int g0 = gd.gsplit[TG::ROWT][split], g1 = gd.gsplit[TG::ROWT][split + 1];
      const int r0 = goff[g0], r1 = goff[g1];
      size_t qoff[K];
#pragma unroll
      for (int kk = 0; kk < K; kk++) qoff[kk] = (size_t)cid[kk] * RH_NVARS + TG::G_FIRST;
      if constexpr (!SCAN) {
#if RH_GATHER_V2
        rh_gather_walk_a<T, NV>(gu_lds, th, inv, cp, acc, gd, q, lane, g0, g1, r0, r1, cid, cok, err);
#else
        // Every non-empty group has at least 64 rows (cfg 5: 100): a tile touches at most two groups, A (the one that is open) and
        // the one after it.  Each lane keeps a running scatter sum for either; when the tile's last row is no longer in A, A is
        // complete: one wave reduction per GROUP (not per tile), as in the group-major walk, but with every lane busy.
        int gA = 0;
        if constexpr (TG::HAS_GATHER) { if (r0 < r1) gA = (int)cp[TG::G_COL][r0] - TG::G_LOW; }
        double accA[K], accB[K];
#pragma unroll
        for (int kk = 0; kk < K; kk++) { accA[kk] = 0.0; accB[kk] = 0.0; }
        auto flush = [&](const int gdone) {   // wave-uniform
          rh_wave_sum_all(accA);
#pragma unroll
          for (int kk = 0; kk < K; kk++)
            if (lane == 0 && cok[kk]) gd.sbuf[TG::ROWT][(size_t)cid[kk] * TG::G_COUNT + gdone] = accA[kk];
        };
        // (A software-pipelined walk -- index column two tiles ahead, the other columns and the table entries one ahead -- was
        //  measured in round 3: 162 VGPRs = three wavefronts per SIMD instead of four, 3.37 ms against 3.16-3.21 ms for cfg 5.  The
        //  walk is bound by its arithmetic, not by the two trips to memory per tile; the plain loop stays.)
        for (int base = r0; base < r1; base += 64) {
          const int r = base + lane;
          const bool live = r < r1;
          const int rr = live ? r : r1 - 1;
          double cc[NC];
RH_UNROLL_ACC
          for (int j = 0; j < NC; j++) cc[j] = cp[j][rr];
          int g = 0;
          if constexpr (TG::HAS_GATHER) g = (int)cc[TG::G_COL] - TG::G_LOW;
          double gz[K], sv[K];
#pragma unroll
          for (int kk = 0; kk < K; kk++) {
            gz[kk] = TG::HAS_GATHER ? q[qoff[kk] + g] : 0.0;
            sv[kk] = 0.0;
          }
          rh_gather_rows<T, NV>(th, inv, cc, gz, acc, sv, err, base + 64 <= r1, live);
          if constexpr (TG::HAS_GATHER) {
            if (g == gA) {
#pragma unroll
              for (int kk = 0; kk < K; kk++) accA[kk] += sv[kk];
            } else {
#pragma unroll
              for (int kk = 0; kk < K; kk++) accB[kk] += sv[kk];
            }
            const int last = (r1 - base < 64 ? r1 - base : 64) - 1;
            const int gLast = __builtin_amdgcn_readlane(g, last);
            if (gLast != gA) {
              flush(gA);
#pragma unroll
              for (int kk = 0; kk < K; kk++) { accA[kk] = accB[kk]; accB[kk] = 0.0; }
              gA = gLast;
            }
          }
        }
        if constexpr (TG::HAS_GATHER) { if (r0 < r1) flush(gA); }
#endif
      } else {
      double carry[K];
#pragma unroll
      for (int kk = 0; kk < K; kk++) carry[kk] = 0.0;
      for (int base = r0; base < r1; base += 64) {
        const int r = base + lane;
        const bool live = r < r1;
        const int rr = live ? r : r1 - 1;
        double cc[NC];
RH_UNROLL_ACC
        for (int j = 0; j < NC; j++) cc[j] = cp[j][rr];
        int g = g0, gbeg = base, gend = r1;   // a target without a gather: one segment, nothing is stored per group
        if constexpr (TG::HAS_GATHER) {
          g = (int)cc[TG::G_COL] - TG::G_LOW;  // in range: rh_model_create walked the index column (RH_E_LOOKUP)
          gbeg = goff[g]; gend = goff[g + 1];
        }
        double gz[K], sv[K];
#pragma unroll
        for (int kk = 0; kk < K; kk++) {
          gz[kk] = TG::HAS_GATHER ? q[qoff[kk] + g] : 0.0;
          sv[kk] = 0.0;
        }
        rh_gather_rows<T, NV>(th, inv, cc, gz, acc, sv, err, base + 64 <= r1, live);
        if constexpr (TG::HAS_GATHER) {
          const int start = gbeg - base;                      // first lane of this lane's group (<= 0: it began earlier)
          const bool tail = live && (r == gend - 1);          // last row of its group
          const int last = (r1 - base < 64 ? r1 - base : 64) - 1;   // last live lane of the tile (wave-uniform)
          const int last_open = __builtin_amdgcn_readlane(tail ? 0 : 1, last);
#pragma unroll
          for (int kk = 0; kk < K; kk++) {
            double v = rh_segmented_scan(sv[kk], start, lane);
            v += (start < 0) ? carry[kk] : 0.0;               // the group that was open at the end of the previous tile
            if (tail && cok[kk]) gd.sbuf[TG::ROWT][(size_t)cid[kk] * TG::G_COUNT + g] = v;
            carry[kk] = last_open ? rh_readlane(v, last) : 0.0;
          }
        }
      }
      }
#pragma unroll
      for (int kk = 0; kk < K; kk++) {
        rh_wave_sum_all(acc[kk]);
        double *out = partial + (((size_t)TG::ROWT * nsplit + split) * chains + cid[kk]) * RH_NACC_MAX;
        if (lane == 0 && cok[kk]) {
#pragma unroll
          for (int o = 0; o < NA; o++) out[o] = acc[kk][o];
        }
      }
     }
    }
    rh_gather_targets<T + 1, NV, SCAN>(th, d, gd, q, lane, split, nsplit, cid, cok, chains, gu_lds, partial, err);
  }
}
#pragma clang fp contract(off)

#ifndef RH_GATHER_WAVES
#define RH_GATHER_WAVES 1
#endif
template <bool SCAN>
RH_DEV void rh_gather_body(const rh_model_data &d, const rh_gather_data &gd, const double *__restrict__ q,
                           const int *__restrict__ list, const int *__restrict__ nlive, const int *__restrict__ vflag, double *__restrict__ partial,
                           int *__restrict__ err_out, int *__restrict__ n_running, const int chains, const int nsplit, double *gu) {
  const int lane = threadIdx.x;
  const int b = blockIdx.x;
  if (b == 0 && lane == 0) *n_running = 0;
  const int nl = rh_live_count(nlive, chains);
  const int split = b % nsplit, group = b / nsplit;
  if (group * RH_GRAD_K >= nl) return;
  int cid[RH_GRAD_K];
  bool cok[RH_GRAD_K];
  double th[RH_GRAD_K][RH_NTH];
  bool vfree = rh_any_value_only<0>::v && vflag != nullptr;   // every chain of the group asked for the gradient only
#pragma unroll
  for (int kk = 0; kk < RH_GRAD_K; kk++) {
    cok[kk] = group * RH_GRAD_K + kk < nl;
    cid[kk] = rh_live_chain(list, group * RH_GRAD_K + kk, nl);
    if constexpr (rh_any_value_only<0>::v) vfree = vfree && vflag[cid[kk]] == 2;
#pragma unroll
    for (int i = 0; i < RH_NTH; i++) th[kk][i] = q[(size_t)cid[kk] * RH_NVARS + i];
  }
  int err = 0;
  if constexpr (rh_any_value_only<0>::v) {
    if (vfree) rh_gather_targets<0, false, SCAN>(th, d, gd, q, lane, split, nsplit, cid, cok, chains, gu, partial, err);
    else rh_gather_targets<0, true, SCAN>(th, d, gd, q, lane, split, nsplit, cid, cok, chains, gu, partial, err);
  } else
  rh_gather_targets<0, true, SCAN>(th, d, gd, q, lane, split, nsplit, cid, cok, chains, gu, partial, err);
  if (err && lane == 0) atomicOr(err_out, 1);
}
extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RH_GATHER_WAVES)))
rh_grad_gather_kernel(const rh_model_data d, const rh_gather_data gd, const double *__restrict__ q,
                      const int *__restrict__ list, const int *__restrict__ nlive, const int *__restrict__ vflag, double *__restrict__ partial, int *__restrict__ err_out,
                      int *__restrict__ n_running, const int chains, const int nsplit) {
  rh_lk_init();
  // the walk's wave-uniform operands (ONE buffer for every target's walk: they run one after the other)
#if RH_GATHER_V2
  __shared__ double rh_gu[RH_GU_SIZE];
  double *const gu = rh_gu;
#else
  double *const gu = nullptr;
#endif
  rh_gather_body<false>(d, gd, q, list, nlive, vflag, partial, err_out, n_running, chains, nsplit, gu);
}
// the row targets whose groups are small (launched behind rh_grad_gather_kernel when the data have such a target)
extern "C" __global__ void __launch_bounds__(64)
rh_grad_gather_scan_kernel(const rh_model_data d, const rh_gather_data gd, const double *__restrict__ q,
                           const int *__restrict__ list, const int *__restrict__ nlive, const int *__restrict__ vflag, double *__restrict__ partial, int *__restrict__ err_out,
                           int *__restrict__ n_running, const int chains, const int nsplit) {
  rh_lk_init();
  rh_gather_body<true>(d, gd, q, list, nlive, vflag, partial, err_out, n_running, chains, nsplit, nullptr);
}
#endif  // RH_HAS_GATHER

// partial sums -> (logp, grad) of one chain, data-free targets evaluated here, everything in target order
template <int T>
RH_DEV void rh_combine_targets(const double (&th)[RH_NTH], const double *__restrict__ partial, const rh_model_data &d,
                               const int nsplit, const int chain, const int chains, const int lane,
                               double (&tot)[RH_NOUT], int &err) {
  const long long *const nrows = d.nrows;
  if constexpr (T < RH_NTARGETS) {
    typedef rh_target<T> TG;
    if constexpr (!TG::HAS_ROWS) {
      double inv[1];
      TG::row(th, inv, d.kpool, tot, err);
    } else {
      constexpr int NA = TG::NACC > 0 ? TG::NACC : 1;
      // lane l sums the splits l, l+64, ... (ascending), then the fixed-order butterfly: deterministic, and the
      // nsplit loads are issued in parallel instead of as one dependent chain
      double S[NA];
RH_UNROLL_ACC
      for (int o = 0; o < NA; o++) S[o] = 0.0;
      for (int s = lane; s < nsplit; s += 64) {
        const double *p = partial + (((size_t)TG::ROWT * nsplit + s) * chains + chain) * RH_NACC_MAX;
RH_UNROLL_ACC
        for (int o = 0; o < NA; o++) S[o] += p[o];
      }
      rh_wave_sum_all(S);
      double inv[TG::NINV > 0 ? TG::NINV : 1];
      TG::invariants(th, inv, err);
      TG::finish(th, inv, S, (double)nrows[T], tot);
    }
    rh_combine_targets<T + 1>(th, partial, d, nsplit, chain, chains, lane, tot, err);
  }
}

// (logp, grad) of one chain from the tick engine's buffers: shared outputs via rh_combine_targets, and in gather mode
// the table parameters' gradients = the per-group scatter sums of every gather target, in target order
template <int T>
RH_DEV void rh_scatter_sum(
#if RH_HAS_GATHER
    const rh_gather_data &gd,
#endif
    const int chain, const int idx, double &g) {
#if RH_HAS_GATHER
  if constexpr (T < RH_NTARGETS) {
    typedef rh_target<T> TG;
    if constexpr (TG::HAS_ROWS && TG::HAS_GATHER) g += gd.sbuf[TG::ROWT][(size_t)chain * TG::G_COUNT + idx];
    rh_scatter_sum<T + 1>(gd, chain, idx, g);
  }
#else
  (void)chain; (void)idx; (void)g;
#endif
}
RH_DEV void rh_combine_chain(const wvec &q, const rh_model_data &d,
#if RH_HAS_GATHER
                             const rh_gather_data &gd,
#endif
                             const double *__restrict__ partial, const int nsplit, const int chain, const int chains,
                             const int lane, double &logp, wvec &grad, int &err) {
#if RH_BIGTH
  const double (&th)[RH_NTH] = *reinterpret_cast<const double (*)[RH_NTH]>(q.s.p);
  double (&tot)[RH_NOUT] = *reinterpret_cast<double (*)[RH_NOUT]>(rh_tot_base);
  for (int o = 0; o < RH_NOUT; o++) tot[o] = 0.0;
  rh_combine_targets<0>(th, partial, d, nsplit, chain, chains, lane, tot, err);
  logp = tot[0];
  for (int i = lane; i < RH_NVARS; i += 64) grad.s.p[i] = tot[1 + i];
  return;
#else
  double th[RH_NTH];
#pragma unroll
  for (int i = 0; i < RH_NTH; i++) th[i] = wv_elem(q, i);
  double tot[RH_NOUT];
#pragma unroll
  for (int o = 0; o < RH_NOUT; o++) tot[o] = 0.0;
  rh_combine_targets<0>(th, partial, d, nsplit, chain, chains, lane, tot, err);
  logp = tot[0];
#if RH_BIGN
  RH_UNROLL_SLOTS
  for (int k = 0; k < RH_SLOTS; k++)
    if (k * 64 + lane < RH_NVARS) grad.s[k] = 0.0; // `grad` may be a view on an exactly-sized array
  // ... so only the lane that owns element i touches it.  (The select form below reads and writes back EVERY lane's element of the
  // slot: through a view on the caller's [chains][nVars] array the lanes past nVars did that to the first elements of the NEXT
  // chain's gradient, racing with the workgroup that computes them -- found by the round-4 create-time self-check.)
#pragma unroll
  for (int i = 0; i < RH_NTH; i++)
    if ((i & 63) == lane) grad.s[i >> 6] = tot[1 + i];
#else
  wv_zero(grad);
#pragma unroll
  for (int i = 0; i < RH_NTH; i++) grad.s[i >> 6] = ((i & 63) == lane) ? tot[1 + i] : grad.s[i >> 6];
#endif
#if RH_HAS_GATHER
  // the table parameters' gradients: 8 slots' scatter sums are loaded before the first store (see RH_BIG2)
  for (int k0 = 0; k0 < RH_SLOTS; k0 += 8) {
    double g[8];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int i = (k0 + j) * 64 + lane;
      g[j] = 0.0;
      if (k0 + j < RH_SLOTS && i >= RH_NSHARED && i < RH_NVARS) rh_scatter_sum<0>(gd, chain, i - RH_NSHARED, g[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int i = (k0 + j) * 64 + lane;
      if (k0 + j < RH_SLOTS && i >= RH_NSHARED && i < RH_NVARS) grad.s[k0 + j] = g[j];
    }
  }
#endif
#endif  // RH_BIGTH
}

#ifndef RH_TICK_FAST
#define RH_TICK_FAST 1
#endif

// ---- the mid-trajectory leapfrog update fused into the gradient launch ---------------------------------------------------------
// All but one of the L gradient requests of a static-HMC trajectory are followed by `twoFullSteps` (LeapFrog.scala:175-184):
//   p += eps * grad;  q += eps * velocity(p)
// -- a dozen flops per chain for which the tick engine paid a kernel launch (rh_tick_kernel's fast path: ~15 us + the gap, 6 % of a
// cfg-2 leapfrog step).  rh_grad_fused_kernel is rh_grad_kernel with that update as its PROLOGUE: every workgroup of a chain group
// combines the partial sums the PREVIOUS launch left (visible at the kernel boundary: nothing crosses XCDs inside a launch, no
// counter, no fence), advances the group's RH_GRAD_K chains itself -- all of them at once, 64 / RH_GRAD_K lanes per chain, each
// workgroup of the group redundantly and identically -- and walks its row split at the NEW q, which never goes through memory.
// The chains' advanced (q, p, g, logp, counters) go into a small per-chain RECORD (written by the group's split-0 workgroup); the
// main state image is not touched, so no workgroup can read what another one of the same launch has written: launch n reads
// partial sums and records of launch n-1 and writes its own (two buffers each, alternating).  Where the trajectory ends the host
// issues rh_absorb_kernel (record -> state image) and the ordinary tick.  (Round 3 first tried the update as an EPILOGUE run by
// the last workgroup of a group to arrive: the hand-off of the partial sums between XCDs cost what the launch cost)
// Bit-identical chains by construction: the splits are summed in the very order rh_combine_targets uses (per-slot ascending sums,
// then the 64-slot xor butterfly -- its upper levels happen between a lane's own registers here, the lower ones between the lanes
// of the chain's group; IEEE addition commutes), the same generated finish() / data-free row() code runs on the sums, and the
// update is spelled as wv_axpy / rh_velocity spell it (multiply, round, add, round; contraction off).  A chain that is not in
// that state (or a Lookup error) is left alone: its q does not move, its gradient is recomputed identically and the next tick
// consumes it.
#if RH_NROWTARGETS > 0 && !RH_HAS_GATHER && !RH_BIGN && !RH_WITH_DENSE && !RH_WITH_NUTS && RH_PACK_L == 64 && RH_SLOTS == 1
#if (RH_GRAD_K == 2 || RH_GRAD_K == 4 || RH_GRAD_K == 8 || RH_GRAD_K == 16) && (RH_NVARS * RH_GRAD_K <= 64) && !RH_BIGTH
#define RH_HAVE_FUSED 1
template <int T>
RH_DEV void rh_combine_targets_packed(const double (&th)[RH_NTH], const double *partial, const rh_model_data &d, const int nsplit,
                                      const int chain, const int chains, const int j, double (&tot)[RH_NOUT], int &err) {
  const long long *const nrows = d.nrows;
  if constexpr (T < RH_NTARGETS) {
    typedef rh_target<T> TG;
    if constexpr (!TG::HAS_ROWS) {
      double inv[1];
      TG::row(th, inv, d.kpool, tot, err);
    } else {
      constexpr int NA = TG::NACC > 0 ? TG::NACC : 1, M = RH_GRAD_K, LPC = 64 / RH_GRAD_K;
      double S[NA];
      double r[NA][M];  // r[o][m]: slot j + LPC * m of the 64-slot butterfly rh_combine_targets runs over the splits, output o
      // splits 0..63: one address per slot (clamped: every lane of a slot row that has a split at all loads, the surplus is masked
      // after the load), the outputs at fixed offsets from it; all the loads are in flight together
#pragma unroll
      for (int m = 0; m < M; m++) {
        const int sp = j + LPC * m, spc = sp < nsplit ? sp : nsplit - 1;
        const double *pm = partial + (((size_t)TG::ROWT * nsplit + spc) * chains + chain) * RH_NACC_MAX;
#pragma unroll
        for (int o = 0; o < NA; o++) r[o][m] = 0.0;
        if (LPC * m < nsplit) {   // wave-uniform: with 16 splits and 8 lanes per chain only m = 0, 1 touch memory
#pragma unroll
          for (int o = 0; o < NA; o++) r[o][m] = pm[o];
        }
      }
#pragma unroll
      for (int m = 0; m < M; m++) {
        const bool have = j + LPC * m < nsplit;
#pragma unroll
        for (int o = 0; o < NA; o++) r[o][m] = 0.0 + (have ? r[o][m] : 0.0);   // rh_combine_targets starts every slot at 0.0
      }
      for (int base = 64; base < nsplit; base += 64) {   // more than 64 splits (few chains): the slot's further splits, ascending
#pragma unroll
        for (int m = 0; m < M; m++) {
          const int sp = base + j + LPC * m;
          if (sp < nsplit) {
            const double *pm = partial + (((size_t)TG::ROWT * nsplit + sp) * chains + chain) * RH_NACC_MAX;
#pragma unroll
            for (int o = 0; o < NA; o++) r[o][m] += pm[o];
          }
        }
      }
#pragma unroll
      for (int o = 0; o < NA; o++) {
#pragma unroll
        for (int off = 32; off >= LPC; off >>= 1) {  // butterfly levels whose partner slot lives in this lane
          double t[M];
#pragma unroll
          for (int m = 0; m < M; m++) t[m] = r[o][m] + r[o][m ^ (off / LPC)];
#pragma unroll
          for (int m = 0; m < M; m++) r[o][m] = t[m];
        }
        double v = r[o][0];
#pragma unroll
        for (int off = LPC / 2; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);  // ... and in the chain's other lanes
        S[o] = v;
      }
      double inv[TG::NINV > 0 ? TG::NINV : 1];
      TG::invariants(th, inv, err);
      TG::finish(th, inv, S, (double)nrows[T], tot);
    }
    rh_combine_targets_packed<T + 1>(th, partial, d, nsplit, chain, chains, j, tot, err);
  }
}


// the per-chain record of the fused launches, in 8-byte words: q, p, g (RH_NVARS each), then
#define RH_REC_LOGP (3 * RH_NVARS)
#define RH_REC_TSI (3 * RH_NVARS + 1)
#define RH_REC_ERR (3 * RH_NVARS + 2)
#define RH_REC_NGRAD (3 * RH_NVARS + 3)
#define RH_REC_NLEAP (3 * RH_NVARS + 4)
#define RH_REC_NWARM (3 * RH_NVARS + 5)
#define RH_REC_VALID (3 * RH_NVARS + 6)
#define RH_REC_U64 (3 * RH_NVARS + 8)

// Prologue: the K chains' parameter vectors for this launch.  rec_in = the records of the previous launch (nullptr: the previous
// launch was a plain one and the state image is current), partial_in = its partial sums.
RH_DEV void rh_fused_prologue(double (&thk)[RH_GRAD_K][RH_NTH], const rh_model_data &d, const double *__restrict__ q,
                              const rh_u64 *__restrict__ state, const rh_u64 *__restrict__ rec_in, rh_u64 *__restrict__ rec_out, const double *__restrict__ partial_in,
                              const int chains, const int nsplit, const int *__restrict__ list, const int slot0, const int nl,
                              const int lane, const bool writer) {
  constexpr int LPC = 64 / RH_GRAD_K;
  const int kk = lane / LPC, j = lane & (LPC - 1), base = lane & ~(LPC - 1);
  const bool exists = slot0 + kk < nl;
  const int chain = rh_live_chain(list, slot0 + kk, nl);
  const rh_u64 *st = state + (size_t)chain * RH_STATE_U64;
  const rh_u64 *sc = st + (size_t)(RH_STATE_DENSE_OFF + RH_STATE_NDENSE) * 64;
  const bool live = j < RH_NVARS;
  const int jc = live ? j : 0;
  // what never changes between two ticks
  const int pc = (int)(rh_i64)sc[RH_SI_pc], need = (int)(rh_i64)sc[RH_SI_need_eval], ts_l = (int)(rh_i64)sc[RH_SI_ts_l];
  const bool identity = (rh_i64)sc[RH_SI_mass_identity] != 0, sampling = (rh_i64)sc[RH_SI_sampling_started] != 0;
  const double eps = __longlong_as_double((rh_i64)sc[RH_SI_eps]);
  const double mm = __longlong_as_double((rh_i64)st[(size_t)(RH_VI_M * RH_SLOTS) * 64 + jc]);
  // what the fused launches advance: from the previous launch's record if it left a valid one, else from the state image
  const rh_u64 *ri = rec_in ? rec_in + (size_t)chain * RH_REC_U64 : nullptr;
  const bool from_rec = ri != nullptr && ri[RH_REC_VALID] != 0;
  double bq, bp, g_old, logp_old;
  int ts_i, cerr;
  rh_i64 n_grad, n_leap, n_warm;
  if (from_rec) {
    bq = __longlong_as_double((rh_i64)ri[jc]); bp = __longlong_as_double((rh_i64)ri[RH_NVARS + jc]);
    g_old = __longlong_as_double((rh_i64)ri[2 * RH_NVARS + jc]); logp_old = __longlong_as_double((rh_i64)ri[RH_REC_LOGP]);
    ts_i = (int)(rh_i64)ri[RH_REC_TSI]; cerr = (int)(rh_i64)ri[RH_REC_ERR];
    n_grad = (rh_i64)ri[RH_REC_NGRAD]; n_leap = (rh_i64)ri[RH_REC_NLEAP]; n_warm = (rh_i64)ri[RH_REC_NWARM];
  } else {
    bq = __longlong_as_double((rh_i64)st[(size_t)(RH_VI_Bq * RH_SLOTS) * 64 + jc]);
    bp = __longlong_as_double((rh_i64)st[(size_t)(RH_VI_Bp * RH_SLOTS) * 64 + jc]);
    g_old = 0.0; logp_old = 0.0;
    ts_i = (int)(rh_i64)sc[RH_SI_ts_i]; cerr = (int)(rh_i64)sc[RH_SI_err];
    n_grad = (rh_i64)sc[RH_SI_n_grad]; n_leap = (rh_i64)sc[RH_SI_n_leapfrog]; n_warm = (rh_i64)sc[RH_SI_n_warm_leapfrog];
  }
  if (!live) { bq = 0.0; bp = 0.0; }
  const bool fast = exists && pc == RH_S_TS_MID && need != 0 && ts_i < ts_l;
  bool valid = from_rec;
  if (__any(fast)) {
    double th[RH_NTH];
#pragma unroll
    for (int i = 0; i < RH_NTH; i++) th[i] = __shfl(bq, base + i, 64);
    double tot[RH_NOUT];
#pragma unroll
    for (int o = 0; o < RH_NOUT; o++) tot[o] = 0.0;
    int err = 0;
    rh_combine_targets_packed<0>(th, partial_in, d, nsplit, chain, chains, j, tot, err);
    double g = 0.0;
#pragma unroll
    for (int i = 0; i < RH_NVARS; i++) g = (j == i) ? tot[1 + i] : g;
#pragma unroll
    for (int off = LPC / 2; off >= 1; off >>= 1) err |= __shfl_xor(err, off, 64);
    if (fast) {
      // RH_S_TS_MID with ts_i < ts_l:  BU = -logp; Bg = grad;  wv_axpy(Bp, eps, Bg);  rh_new_qs: wv_axpy(Bq, eps, velocity(Bp))
      if (live) {
        bp += eps * g;
        const double vel = identity ? bp : bp * mm;
        bq += eps * vel;
      }
      g_old = g; logp_old = tot[0];
      ts_i += 1; cerr |= err; n_grad += 1;
      if (sampling) n_leap += 1; else n_warm += 1;
      valid = true;
    }
  }
  if (writer && exists) {
    rh_u64 *ro = rec_out + (size_t)chain * RH_REC_U64;
    if (live && valid) {
      ro[j] = (rh_u64)__double_as_longlong(bq); ro[RH_NVARS + j] = (rh_u64)__double_as_longlong(bp);
      ro[2 * RH_NVARS + j] = (rh_u64)__double_as_longlong(g_old);
    }
    if (j == 0) {
      ro[RH_REC_LOGP] = (rh_u64)__double_as_longlong(logp_old);
      ro[RH_REC_TSI] = (rh_u64)(rh_i64)ts_i; ro[RH_REC_ERR] = (rh_u64)(rh_i64)cerr;
      ro[RH_REC_NGRAD] = (rh_u64)n_grad; ro[RH_REC_NLEAP] = (rh_u64)n_leap; ro[RH_REC_NWARM] = (rh_u64)n_warm;
      ro[RH_REC_VALID] = valid ? 1 : 0;
    }
  }
  // the row walk reads its parameters from scalar registers: lane `kk * LPC + i` holds chain kk's q[i].  A chain the fused launches
  // have not advanced is evaluated where its last tick asked for it (qbuf), as rh_grad_kernel would.
  if (!valid) bq = q[(size_t)chain * RH_NVARS + jc];
#pragma unroll
  for (int k2 = 0; k2 < RH_GRAD_K; k2++)
#pragma unroll
    for (int i = 0; i < RH_NTH; i++) thk[k2][i] = rh_readlane(bq, k2 * LPC + i);
}

extern "C" __global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(RH_GRAD_WAVES)))
rh_grad_fused_kernel(const rh_model_data d, const double *__restrict__ q, const int *__restrict__ list, const int *__restrict__ nlive,
                     const double *__restrict__ partial_in, double *__restrict__ partial_out, int *__restrict__ err_out, int *__restrict__ n_running,
                     const rh_u64 *__restrict__ state, const rh_u64 *__restrict__ rec_in, rh_u64 *__restrict__ rec_out,
                     const int chains, const int nsplit, const int xcd_aware) {
  rh_lk_init();
  const int lane = threadIdx.x, b = blockIdx.x;
  if (b == 0 && lane == 0) *n_running = 0;
  // (every listed chain waits for a gradient, and the list does not change between the launches of a trajectory: each launch writes
  //  the record of every chain the next one reads)
  const int nl = rh_live_count(nlive, chains);
  int split, group;
  rh_grad_map(b, nsplit, xcd_aware, split, group);
  if (group * RH_GRAD_K >= nl) return;
  double th[RH_GRAD_K][RH_NTH];
  rh_fused_prologue(th, d, q, state, rec_in, rec_out, partial_in, chains, nsplit, list, group * RH_GRAD_K, nl, lane, split == 0);
  int err = 0;
  rh_grad_targets<0, false>(th, d, lane, split, nsplit, list + group * RH_GRAD_K, nl - group * RH_GRAD_K, chains, partial_out, err);
  if (err && lane == 0) atomicOr(err_out, 1);
}

// records -> state image, before the tick that ends a trajectory (one wavefront per chain; a chain without a valid record is
// left as it is).  What the ordinary tick would have stored after each of the fused updates (rh_tick_kernel, RH_S_TS_MID).
extern "C" __global__ void __launch_bounds__(64)
rh_absorb_kernel(rh_u64 *__restrict__ state, rh_u64 *__restrict__ rec, double *__restrict__ qbuf, const int *__restrict__ active,
                 const int chains) {
  const int chain = blockIdx.x, j = threadIdx.x;
  if (chain >= chains) return;
  rh_u64 *ri = rec + (size_t)chain * RH_REC_U64;
  // only a chain that is waiting for a gradient can have been advanced by the launch that wrote this record; a record is consumed
  // once (a paused chain must never be overwritten with what an earlier trajectory left in the buffer)
  if (ri[RH_REC_VALID] == 0 || active[chain] == 0) return;
  rh_u64 *st = state + (size_t)chain * RH_STATE_U64;
  rh_u64 *sc = st + (size_t)(RH_STATE_DENSE_OFF + RH_STATE_NDENSE) * 64;
  if (j < RH_NVARS) {
    const rh_u64 q = ri[j], p = ri[RH_NVARS + j], g = ri[2 * RH_NVARS + j];
    qbuf[(size_t)chain * RH_NVARS + j] = __longlong_as_double((rh_i64)q);
    st[(size_t)(RH_VI_Bq * RH_SLOTS) * 64 + j] = q;
    st[(size_t)(RH_VI_Bp * RH_SLOTS) * 64 + j] = p;
    st[(size_t)(RH_VI_Bg * RH_SLOTS) * 64 + j] = g;
    st[(size_t)(RH_VI_pend_g * RH_SLOTS) * 64 + j] = g;
  }
  if (j == 0) {
    const double logp = __longlong_as_double((rh_i64)ri[RH_REC_LOGP]);
    sc[RH_SI_BU] = (rh_u64)__double_as_longlong(logp * -1); sc[RH_SI_pend_logp] = ri[RH_REC_LOGP];
    sc[RH_SI_ts_i] = ri[RH_REC_TSI]; sc[RH_SI_err] = ri[RH_REC_ERR];
    sc[RH_SI_n_grad] = ri[RH_REC_NGRAD]; sc[RH_SI_n_leapfrog] = ri[RH_REC_NLEAP]; sc[RH_SI_n_warm_leapfrog] = ri[RH_REC_NWARM];
    ri[RH_REC_VALID] = 0;   // (one wavefront per chain: every lane has read its part of the record by now)
  }
}
#endif
#endif
extern "C" __global__ void __launch_bounds__(64)
rh_tick_kernel(const rh_model_data d,
#if RH_HAS_GATHER
               const rh_gather_data gd,
#endif
               const rh_cfg_dev cfg, rh_u64 *__restrict__ state, const rh_i64 *__restrict__ seeds,
               const double *__restrict__ static_mass, double *__restrict__ draws,
               rh_chain_stats_dev *__restrict__ stats, int *__restrict__ n_running, double *__restrict__ qbuf,
               int *__restrict__ active, const double *__restrict__ partial, const int *__restrict__ grad_err,
               const int *__restrict__ list, const int *__restrict__ nlive,
               const int chains, const int nsplit, const int it_stop, const int fresh) {
  // one wavefront per chain of the gradient launch this tick follows (list: rh_compact_kernel's, see "live chains" above); the
  // first tick of a call (list == nullptr) visits every chain -- a paused one resumes there, and nothing else ever re-lists it
  int chain = blockIdx.x;
  const int lane = threadIdx.x;
  if (chain >= chains) return;
  if (list) {
    if (chain >= __builtin_amdgcn_readfirstlane(*nlive)) return;
    chain = __builtin_amdgcn_readfirstlane(list[chain]);
  }
  rh_u64 *st = state + (size_t)chain * RH_STATE_U64;
#if RH_TICK_FAST && !RH_BIGN && !RH_WITH_DENSE && !RH_HAS_GATHER && RH_PACK_L == 64
  // Fast path for the tick that follows a mid-trajectory gradient (all but one of the L ticks of an HMC / EHMC trajectory):
  // `twoFullSteps` (LeapFrog.scala:175-184) is  p += eps * grad;  q += eps * velocity(p)  and touches three vectors and a
  // handful of scalars of the ~20-vector state image.  Only those are loaded and stored here, with the same functions the
  // automaton's RH_S_TS_MID case applies to them, so the chains are bit-identical to the general path below -- which costs
  // ~13 us per wavefront in state traffic and instruction fetch for the same arithmetic.
  if (!fresh) {
    rh_u64 *sc = st + (size_t)(RH_STATE_DENSE_OFF + RH_STATE_NDENSE) * 64;
    const int pc = rh_uniform_i((int)(rh_i64)sc[RH_SI_pc]), need = rh_uniform_i((int)(rh_i64)sc[RH_SI_need_eval]);
    const int ts_i = rh_uniform_i((int)(rh_i64)sc[RH_SI_ts_i]), ts_l = rh_uniform_i((int)(rh_i64)sc[RH_SI_ts_l]);
    if (pc == RH_S_TS_MID && need != 0 && ts_i < ts_l) {
      rh_chain c;  // only the fields read below are filled
      c.mass_identity = rh_uniform_i((int)(rh_i64)sc[RH_SI_mass_identity]);
      c.sampling_started = rh_uniform_i((int)(rh_i64)sc[RH_SI_sampling_started]);
      c.err = rh_uniform_i((int)(rh_i64)sc[RH_SI_err]);
      c.eps = __longlong_as_double((rh_i64)sc[RH_SI_eps]);
      c.n_grad = (rh_i64)sc[RH_SI_n_grad]; c.n_leapfrog = (rh_i64)sc[RH_SI_n_leapfrog]; c.n_warm_leapfrog = (rh_i64)sc[RH_SI_n_warm_leapfrog];
RH_UNROLL_SLOTS
      for (int k = 0; k < RH_SLOTS; k++) {
        c.Bq.s[k] = __longlong_as_double((rh_i64)st[(size_t)(RH_VI_Bq * RH_SLOTS + k) * 64 + lane]);
        c.Bp.s[k] = __longlong_as_double((rh_i64)st[(size_t)(RH_VI_Bp * RH_SLOTS + k) * 64 + lane]);
        c.M.s[k] = __longlong_as_double((rh_i64)st[(size_t)(RH_VI_M * RH_SLOTS + k) * 64 + lane]);
      }
      int err = grad_err[0];
      rh_combine_chain(c.Bq, d, partial, nsplit, chain, chains, lane, c.pend_logp, c.pend_g, err);
      c.err |= err;
      c.BU = c.pend_logp * -1; c.Bg = c.pend_g;          // RH_S_TS_MID, ts_i < ts_l
      wv_axpy(c.Bp, c.eps, c.Bg);
      rh_new_qs(c, c.mass_identity != 0);
RH_UNROLL_SLOTS
      for (int k = 0; k < RH_SLOTS; k++) {
        if (k * 64 + lane < RH_NVARS) qbuf[(size_t)chain * RH_NVARS + k * 64 + lane] = c.Bq.s[k];
        st[(size_t)(RH_VI_Bq * RH_SLOTS + k) * 64 + lane] = (rh_u64)__double_as_longlong(c.Bq.s[k]);
        st[(size_t)(RH_VI_Bp * RH_SLOTS + k) * 64 + lane] = (rh_u64)__double_as_longlong(c.Bp.s[k]);
        st[(size_t)(RH_VI_Bg * RH_SLOTS + k) * 64 + lane] = (rh_u64)__double_as_longlong(c.Bg.s[k]);
        st[(size_t)(RH_VI_pend_g * RH_SLOTS + k) * 64 + lane] = (rh_u64)__double_as_longlong(c.pend_g.s[k]);
      }
      if (lane == 0) {
        sc[RH_SI_BU] = (rh_u64)__double_as_longlong(c.BU); sc[RH_SI_pend_logp] = (rh_u64)__double_as_longlong(c.pend_logp);
        sc[RH_SI_ts_i] = (rh_u64)(rh_i64)(ts_i + 1); sc[RH_SI_err] = (rh_u64)(rh_i64)c.err;
        sc[RH_SI_n_grad] = (rh_u64)(c.n_grad + 1); sc[RH_SI_n_leapfrog] = (rh_u64)c.n_leapfrog; sc[RH_SI_n_warm_leapfrog] = (rh_u64)c.n_warm_leapfrog;
        rh_chain_stats_dev *out = stats + chain;
        out->leapfrog_steps = c.n_leapfrog; out->warmup_leapfrog_steps = c.n_warm_leapfrog; out->gradient_evaluations = c.n_grad + 1;
        out->error = c.err; out->status = RH_ADV_NEED_GRAD;
        active[chain] = (ts_i + 1 < ts_l) ? 2 : 1;   // 2: the request just made is a mid-trajectory one too -- gradient only
        atomicAdd(n_running, 1);
      }
      return;
    }
  }
#endif
#if RH_TICK_FAST && RH_BIGN && RH_HAS_GATHER && !RH_WITH_DENSE && RH_PACK_L == 64 && RH_NTH == RH_NSHARED && RH_NTH <= 64
  // The same fast path for gather-mode models in big mode (cfg 5: 10 004 parameters, the chain's vectors in HBM), round 5.  The
  // general path walks the 80 KB vectors once per helper -- scatter sums -> pend_g, Bg = pend_g, p += eps g, v = velocity(p),
  // q += eps v, q -> qbuf: six loops, ~16 vector passes of memory traffic and ~75 000 vector instructions per chain and step
  // (profiles/r5_cfg5: 0.56 ms per launch at 1024 chains, one wavefront per chain).  Here ONE loop does the element's whole update
  // from five loads (the two scatter sums, p, q, the mass element), eight slots' loads in flight, and stores p, q and qbuf: the same
  // per-element arithmetic spelled the same way (multiply, round, add, round; contraction is off here), so the chains are
  // bit-identical to the general path's.  pend_g and Bg are NOT written: whatever tick follows -- this path again or the general
  // one -- begins by overwriting both (rh_combine_chain -> pend_g, RH_S_TS_MID: Bg = pend_g) before anything reads them.
  // Measured at 1024 chains (profiles/r5_cfg5): the launch is bound by its 0.6-0.8 GB of HBM traffic moved by one wavefront per
  // chain in lock-step bursts, not by instructions -- 0.73 -> 0.65 ms per step with the five-store form.
  if (!fresh) {
    rh_u64 *sc = st + (size_t)(RH_STATE_DENSE_OFF + RH_STATE_NDENSE) * 64;
    const int pc = rh_uniform_i((int)(rh_i64)sc[RH_SI_pc]), need = rh_uniform_i((int)(rh_i64)sc[RH_SI_need_eval]);
    const int ts_i = rh_uniform_i((int)(rh_i64)sc[RH_SI_ts_i]), ts_l = rh_uniform_i((int)(rh_i64)sc[RH_SI_ts_l]);
    if (pc == RH_S_TS_MID && need != 0 && ts_i < ts_l) {
      const bool ident = rh_uniform_i((int)(rh_i64)sc[RH_SI_mass_identity]) != 0;
      const int sampling_started = rh_uniform_i((int)(rh_i64)sc[RH_SI_sampling_started]);
      int cerr = rh_uniform_i((int)(rh_i64)sc[RH_SI_err]);
      const double eps = __longlong_as_double((rh_i64)sc[RH_SI_eps]);
      const rh_i64 n_grad = (rh_i64)sc[RH_SI_n_grad];
      rh_i64 n_leapfrog = (rh_i64)sc[RH_SI_n_leapfrog], n_warm_leapfrog = (rh_i64)sc[RH_SI_n_warm_leapfrog];
      double *const vBp = (double *)(st + (size_t)RH_VI_Bp * RH_SLOTS * 64), *const vBq = (double *)(st + (size_t)RH_VI_Bq * RH_SLOTS * 64);
      const double *const vM = (const double *)(st + (size_t)RH_VI_M * RH_SLOTS * 64);
      // shared outputs exactly as rh_combine_chain computes them (theta = the first RH_NTH elements of q, before the update)
      double th[RH_NTH];
#pragma unroll
      for (int i = 0; i < RH_NTH; i++) th[i] = vBq[i];
      double tot[RH_NOUT];
#pragma unroll
      for (int o = 0; o < RH_NOUT; o++) tot[o] = 0.0;
      int err = grad_err[0];
      rh_combine_targets<0>(th, partial, d, nsplit, chain, chains, lane, tot, err);
      cerr |= err;
      const double pend_logp = tot[0];
      double gsh = 0.0;   // this lane's shared-parameter gradient (element `lane` of slot 0, lane < RH_NTH)
#pragma unroll
      for (int i = 0; i < RH_NTH && i < 64; i++) gsh = (i == lane) ? tot[1 + i] : gsh;
      constexpr int U = 8;
      _Pragma("unroll 1") for (int k0 = 0; k0 < RH_SLOTS; k0 += U) {
        double g[U], pv[U], qv[U], mv[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
          const int kc = k0 + j < RH_SLOTS ? k0 + j : RH_SLOTS - 1;
          const int i = kc * 64 + lane;
          g[j] = 0.0;
          if (i >= RH_NSHARED && i < RH_NVARS) rh_scatter_sum<0>(gd, chain, i - RH_NSHARED, g[j]);
          pv[j] = vBp[i]; qv[j] = vBq[i]; mv[j] = ident ? 1.0 : vM[i];
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
          const int i = (k0 + j) * 64 + lane;
          if (k0 + j < RH_SLOTS && i < RH_NVARS) {   // (the lanes past RH_NVARS of the last slot hold zeros and keep them)
            const double gi = i < RH_NTH ? gsh : g[j];
            const double pn = pv[j] + eps * gi;                   // wv_axpy(Bp, eps, Bg)
            const double vel = ident ? pn : pn * mv[j];           // rh_velocity
            const double qn = qv[j] + eps * vel;                  // wv_axpy(Bq, eps, v)
            vBp[i] = pn; vBq[i] = qn;
            qbuf[(size_t)chain * RH_NVARS + i] = qn;
          }
        }
      }
      if (sampling_started) n_leapfrog += 1; else n_warm_leapfrog += 1;
      if (lane == 0) {
        sc[RH_SI_BU] = (rh_u64)__double_as_longlong(pend_logp * -1); sc[RH_SI_pend_logp] = (rh_u64)__double_as_longlong(pend_logp);
        sc[RH_SI_ts_i] = (rh_u64)(rh_i64)(ts_i + 1); sc[RH_SI_err] = (rh_u64)(rh_i64)cerr;
        sc[RH_SI_n_grad] = (rh_u64)(n_grad + 1); sc[RH_SI_n_leapfrog] = (rh_u64)n_leapfrog; sc[RH_SI_n_warm_leapfrog] = (rh_u64)n_warm_leapfrog;
        rh_chain_stats_dev *out = stats + chain;
        out->leapfrog_steps = n_leapfrog; out->warmup_leapfrog_steps = n_warm_leapfrog; out->gradient_evaluations = n_grad + 1;
        out->error = cerr; out->status = RH_ADV_NEED_GRAD;
        active[chain] = (ts_i + 1 < ts_l) ? 2 : 1;
        atomicAdd(n_running, 1);
      }
      return;
    }
  }
#endif
  rh_chain c;
  if (fresh) rh_chain_zero(st, lane);
  rh_chain_load(c, st, lane);
  if (c.need_eval) { // the gradient requested at the previous tick is in `partial`
    int err = grad_err[0];
    rh_combine_chain(c.Bq, d,
#if RH_HAS_GATHER
                     gd,
#endif
                     partial, nsplit, chain, chains, lane, c.pend_logp, c.pend_g, err);
    c.err |= err; c.need_eval = 0;
  }
  double *my_draws = draws + (size_t)chain * cfg.iterations * RH_NVARS;
  const int status = rh_advance(c, cfg, it_stop, seeds + 2 * (size_t)chain, static_mass, my_draws, lane);
  if (status == RH_ADV_NEED_GRAD) {
    c.need_eval = 1;
#if RH_BIGN
    for (int k0 = 0; k0 < RH_SLOTS; k0 += 8) {  // 8 loads in flight before the stores (see RH_BIG2)
      double t[8];
#pragma unroll
      for (int j = 0; j < 8; j++) t[j] = (k0 + j < RH_SLOTS) ? c.Bq.s[k0 + j] : 0.0;
#pragma unroll
      for (int j = 0; j < 8; j++)
        if (k0 + j < RH_SLOTS && (k0 + j) * 64 + lane < RH_NVARS) qbuf[(size_t)chain * RH_NVARS + (k0 + j) * 64 + lane] = t[j];
    }
#else
RH_UNROLL_SLOTS
    for (int k = 0; k < RH_SLOTS; k++)
      if (k * 64 + lane < RH_NVARS) qbuf[(size_t)chain * RH_NVARS + k * 64 + lane] = c.Bq.s[k];
#endif
  }
  // 1: (logp, gradient) wanted; 2: the gradient only -- the request will be consumed by RH_S_TS_MID as a mid-trajectory step, which
  // overwrites the potential it stores before anything reads it (LeapFrog.scala:175-184)
  if (lane == 0) active[chain] = (status == RH_ADV_NEED_GRAD) ? ((c.pc == RH_S_TS_MID && c.ts_i < c.ts_l) ? 2 : 1) : 0;
  rh_chain_store(c, st, lane);
  rh_stats_write(c, stats + chain, status, lane);
  if (status == RH_ADV_NEED_GRAD && lane == 0) atomicAdd(n_running, 1);
}
#endif  // RH_NROWTARGETS > 0

#if !RH_HAS_GATHER
// Seam 2: batched DensityFunction.  q [chains][nvars] -> logp [chains], grad [chains][nvars]; one wavefront per chain.
extern "C" __global__ void __launch_bounds__(64)
rh_density_kernel(const rh_model_data d, const double *__restrict__ q, double *__restrict__ logp,
                  double *__restrict__ grad, int *__restrict__ err_out, const int chains, double *tot_scratch) {
  rh_lk_init();
  const int chain = blockIdx.x * (64 / RH_LANES) + (int)(threadIdx.x / RH_LANES);  // RH_LANES lanes per chain (64 unless packed)
  const int lane = threadIdx.x & (RH_LANES - 1);
  if (chain >= chains) return;
#if RH_BIGTH
  rh_tot_base = tot_scratch + (size_t)chain * RH_NOUT;   // (one chain per workgroup in big mode; every lane stores the same value)
  wvec qv, gv;                                            // views on the caller's arrays (element i at [i])
  qv.s.p = const_cast<double *>(q) + (size_t)chain * RH_NVARS;
  gv.s.p = grad + (size_t)chain * RH_NVARS;
  double lp; int err = 0;
  rh_density(qv, d, lane, lp, gv, err);
  if (lane == 0) { logp[chain] = lp; if (err) atomicOr(err_out, 1); }
#else
  (void)tot_scratch;
  wvec qv, gv;
RH_UNROLL_SLOTS
  for (int k = 0; k < RH_SLOTS; k++) qv.s[k] = (k * 64 + lane < RH_NVARS) ? q[(size_t)chain * RH_NVARS + k * 64 + lane] : 0.0;
  double lp; int err = 0;
  rh_density(qv, d, lane, lp, gv, err);
  if (lane == 0) { logp[chain] = lp; if (err) atomicOr(err_out, 1); }
RH_UNROLL_SLOTS
  for (int k = 0; k < RH_SLOTS; k++)
    if (k * 64 + lane < RH_NVARS) grad[(size_t)chain * RH_NVARS + k * 64 + lane] = gv.s[k];
#endif
}

#endif  // !RH_HAS_GATHER
#if RH_NROWTARGETS > 0
// Seam 2 through the tick engine's gradient path (rh_density_eval_ex, RH_ENGINE_TICK; always in gather mode): the
// gradient kernel has filled `partial` (and the scatter sums) for q, this kernel finishes them exactly like rh_tick_kernel
extern "C" __global__ void __launch_bounds__(64)
rh_density_fin_kernel(const rh_model_data d,
#if RH_HAS_GATHER
                      const rh_gather_data gd,
#endif
                      const double *__restrict__ q,
                      const double *__restrict__ partial, double *__restrict__ logp, double *__restrict__ grad,
                      int *__restrict__ err_out, const int chains, const int nsplit, double *tot_scratch) {
  const int chain = blockIdx.x;
  const int lane = threadIdx.x;
  if (chain >= chains) return;
#if RH_BIGTH
  rh_tot_base = tot_scratch + (size_t)chain * RH_NOUT;
#else
  (void)tot_scratch;
#endif
  double lp; int err = 0;
#if RH_BIGN
  wvec qv, gv; // views on the caller's arrays (element i at [i])
  qv.s.p = const_cast<double *>(q) + (size_t)chain * RH_NVARS;
  gv.s.p = grad + (size_t)chain * RH_NVARS;
  rh_combine_chain(qv, d,
#if RH_HAS_GATHER
                   gd,
#endif
                   partial, nsplit, chain, chains, lane, lp, gv, err);
  if (lane == 0) { logp[chain] = lp; if (err) atomicOr(err_out, 1); }
#else
  wvec qv, gv;
RH_UNROLL_SLOTS
  for (int k = 0; k < RH_SLOTS; k++) qv.s[k] = (k * 64 + lane < RH_NVARS) ? q[(size_t)chain * RH_NVARS + k * 64 + lane] : 0.0;
  rh_combine_chain(qv, d,
#if RH_HAS_GATHER
                   gd,
#endif
                   partial, nsplit, chain, chains, lane, lp, gv, err);
  if (lane == 0) { logp[chain] = lp; if (err) atomicOr(err_out, 1); }
RH_UNROLL_SLOTS
  for (int k = 0; k < RH_SLOTS; k++)
    if (k * 64 + lane < RH_NVARS) grad[(size_t)chain * RH_NVARS + k * 64 + lane] = gv.s[k];
#endif
}
#endif

// Device self-test of the bit-exact pieces: mode 0 = n gaussians of ScalaRNG(seed), 1 = n uniforms,
// 2 = strict log, 3 = strict exp, 4 = sqrt, 5 = a/b (in[2i], in[2i+1]), 6 = t^-0.75, 7 = fast-mode log,
// and of the functions generated model code calls for the remaining IR ops (ir/MethodGenerator.scala:66-93):
// 8 = sin, 9 = cos, 10 = tan, 11 = asin, 12 = acos, 13 = atan, 14 = Math.pow (in[2i], in[2i+1]), 15 = abs, 16 = fast-mode exp, 17 / 18 = rh_logit_link's softplus / sigmoid
extern "C" __global__ void __launch_bounds__(64)
rh_selftest_kernel(const int mode, const rh_i64 seed, const double *__restrict__ in, double *__restrict__ out, const int n) {
  rh_lk_init();
  const int lane = threadIdx.x;
  if (mode <= 1) {
    rh_rng r; rh_rng_init(r, seed);
    for (int i = 0; i < n; i++) {
      const double v = mode == 0 ? rh_rng_normal(r) : rh_rng_uniform(r);
      if (lane == (i & 63)) out[i] = v;
    }
  } else {
    for (int i = lane + blockIdx.x * 64; i < n; i += 64 * gridDim.x) {
      double v;
      if (mode == 2) v = rh_strict_log(in[i]);
      else if (mode == 3) v = rh_strict_exp(in[i]);
      else if (mode == 4) v = rh_strict_sqrt(in[i]);
      else if (mode == 5) v = in[2 * i] / in[2 * i + 1];
      else if (mode == 7) v = rh_fast_log(in[i]);
      else if (mode == 8) v = sin(in[i]);
      else if (mode == 9) v = cos(in[i]);
      else if (mode == 10) v = tan(in[i]);
      else if (mode == 11) v = asin(in[i]);
      else if (mode == 12) v = acos(in[i]);
      else if (mode == 13) v = atan(in[i]);
      else if (mode == 14) v = rh_java_pow(in[2 * i], in[2 * i + 1]);
      else if (mode == 15) v = __builtin_fabs(in[i]);
      else if (mode == 16) v = exp(in[i]);
      else if (mode == 17 || mode == 18) { double sp, sg; rh_logit_link(in[i], sp, sg); v = mode == 17 ? sp : sg; }
      else v = rh_pow_neg075(in[i]);
      out[i] = v;
    }
  }
}

extern "C" __device__ __attribute__((used)) const int rh_state_words = RH_STATE_U64; // u64 words per chain image
extern "C" __device__ __attribute__((used)) const int rh_state_dense_off = RH_STATE_DENSE_OFF;
// position of the mass vector M among the state image's vectors (the host reads DiagonalMassMatrix.elements from there)
enum {
#define X(n) RH_VIDX_##n,
  RH_STATE_VECS(X)
#undef X
  RH_VIDX_COUNT
};
extern "C" __device__ __attribute__((used)) const int rh_state_mass_vec = RH_VIDX_M;
// u64 word offsets, inside a chain's image, of what the engine's create-time self-check reads back after LeapFrog.initialize: the
// point (Pq), the gradient the sampler kernel computed there (Pg) and PU = -logp.  Element i of a vector sits at offset + i in
// every layout (lane-distributed slots of 64, packed chains, big mode's in-place vectors).
extern "C" __device__ __attribute__((used)) const int rh_state_off_Pq = RH_VI_Pq * RH_SLOTS * 64;
extern "C" __device__ __attribute__((used)) const int rh_state_off_Pg = RH_VI_Pg * RH_SLOTS * 64;
extern "C" __device__ __attribute__((used)) const int rh_state_off_PU = RH_STATE_U64 - RH_STATE_NSCALAR + RH_SI_PU;
