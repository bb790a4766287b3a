// rh_glm4r.hip.h -- hand-written gfx950 device code, appended to the translation unit of models with a dense GLM target only.
//
// rh_grad_glm4r_kernel (round 5): both contractions of a dense GLM target on v_mfma_f64_4x4x4_4b_f64 with the FOUR BLOCKS OF THE
// INSTRUCTION AS FOUR ROW GROUPS -- the mapping that gives the four-block shape the operand reuse of the 16x16x4 one.
//
// What was measured before (profiles/r3_d_fp64_mfma, r3_cfg4): v_mfma_f64_16x16x4_f64 issues once per 105 cycles and SIMD (2048
// flop: 0.61 of the vector FMA rate) and is 60 % of rh_grad_glm_kernel's time on cfg 4; the four-block shape issues once per 16.7
// cycles (512 flop: 0.96 of the vector rate, 1.58x the big shape per flop).  rh_grad_glm4_kernel mapped the blocks to CHAIN groups:
// every instruction then needs its own operand read from LDS (96 reads per 16 rows x 16 chains instead of 24) and the kernel lost
// (24.9 vs 17.6 ms), LDS-latency-bound.  Here a block is a ROW group:
//
//   lane l:  k = l >> 4,  block b = (l >> 2) & 3,  e = l & 3.     A[b][i = e][k],  B[b][k][j = e],  D[b][i][j] in lane 16 i + 4 b + j.
//
//   forward   eta[row 4 b + i][chain 4 jg + j] += sum_k X[row 4 b + i][pred(k, s)] . theta[chain 4 jg + j][pred(k, s)]
//             A: lane (k, b, e) reads X[row 4 b + e][pred(k, s)] -- ONE LDS read per step s, shared by the four chain groups jg;
//             B: scale * theta of the lane's chain 4 jg + e and predictor pred(k, s), in registers (the same in all four blocks);
//             D: lane (i, b, j) ends with eta of row 4 b + i and chain 4 jg + j: a lane evaluates ONE row for FOUR chains.
//   backward  G[pred(i, s)][chain 4 jg + j] += sum_k X[row 4 b + k][pred(i, s)] . w[row 4 b + k][chain 4 jg + j]   (per block b)
//             A: lane (k, b, e) reads X[row 4 b + k][pred(e, s)] -- one LDS read per step, shared by the four chain groups;
//             B: w of row 4 b + k and chain 4 jg + e -- exactly the forward D register of that lane after the scalar part: w never
//             moves between lanes; D: the four blocks keep partial sums over their own rows, folded once at the end of the kernel.
//   with pred(g, s) = S4 g + s, S4 = ceil(P / 4) steps.  24-26 LDS operand reads and 96-104 MFMAs per 16 rows x 16 chains: the issue
//   time of the contractions falls from 24 x 105 = 2520 to ~104 x 16.7 = 1740 cycles per 256 evaluations.
//
// The price is registers: a lane holds theta for FOUR chains (S4 x 4 doubles) and four chains' gradient sums (S4 x 4 accumulators) --
// 208 registers at 51 predictors -- so the kernel runs ONE wavefront per SIMD (512 registers; the accumulators go to the AGPR half),
// hides its global loads in the fetch-ahead / park staging all GLM kernels share, and brings its own instruction-level parallelism
// (52 independent accumulators; four evaluations of the scalar part per lane, interleaved by the compiler).
// All P predictors go to the matrix pipe (no VALU remainder: a padded step costs what three remainder predictors cost on the VALU).
// The row tile sits in LDS row-major in PREDICTOR order -- position p < P holds predictor p's column (or the constant 1 of an
// intercept), the columns no predictor reads follow -- so that a lane's operand address is one base register plus the step as an
// immediate offset; pad positions up to 4 S4 hold zeros (their theta is zero too, but 0 x garbage must not be NaN).
#ifdef RH_GLM_TARGET
#ifndef RH_GLM4R_W
#define RH_GLM4R_W 4
#endif
// chain groups (of four chains) per wavefront: 4 = sixteen chains, one wavefront per SIMD (the operands and accumulators of four
// groups are 208 registers at 51 predictors); 2 = eight chains, half the registers, two wavefronts per SIMD -- an operand read then
// feeds two MFMAs instead of four.  Measured on cfg 4 (profiles/r5_cfg4): with ONE wavefront per SIMD the pipe idles half the time
// (LDS, MFMA -> VALU and barrier latencies have nothing to hide behind): 4.62 ms per launch against rh_grad_glm_kernel's 3.55.
#ifndef RH_GLM4R_JG
#define RH_GLM4R_JG 2
#endif
#define RH_GLM4R_WPS (RH_GLM4R_JG >= 4 ? 1 : 2)   /* wavefronts per SIMD the kernel is compiled for */
#if RH_FP_CONTRACT
#pragma clang fp contract(fast)
#endif
template <class GL>
struct rh_glm4r_map {
  static constexpr int P = GL::P, NC = GL::NCOLS;
  static constexpr int S4 = (P + 3) / 4;
  static constexpr bool is_pred_col(const int j) {
    for (int p = 0; p < P; p++) if (GL::pred_col[p] == j) return true;
    return false;
  }
  static constexpr int count_extra(const int upto) {   // columns below `upto` that no predictor reads
    int n = 0;
    for (int j = 0; j < upto; j++) if (!is_pred_col(j)) n++;
    return n;
  }
  static constexpr int NP = P + count_extra(NC);                 // LDS positions that hold data (or the constant 1)
  static constexpr int NPP = NP > 4 * S4 ? NP : 4 * S4;          // ... and zero pads up to the last (padded) predictor position
  static constexpr int ST = NPP | 1;                             // row stride in doubles (odd)
  struct Tab { int v[NPP]; };
  static constexpr Tab make_src() {   // position -> source column (-1: the constant 1, -2: a zero pad)
    Tab t{};
    for (int p = 0; p < P; p++) t.v[p] = GL::pred_col[p] >= 0 ? GL::pred_col[p] : -1;
    int n = P;
    for (int j = 0; j < NC; j++) if (!is_pred_col(j)) t.v[n++] = j;
    for (; n < NPP; n++) t.v[n] = -2;
    return t;
  }
  static constexpr int pos_of(const int j) {   // data column -> an LDS position that holds it
    for (int p = 0; p < P; p++) if (GL::pred_col[p] == j) return p;
    return P + count_extra(j);
  }
  struct ColTab { int v[NC > 0 ? NC : 1]; };
  static constexpr ColTab make_pos() {
    ColTab t{};
    for (int j = 0; j < NC; j++) t.v[j] = pos_of(j);
    return t;
  }
};
// (tables in constant memory: SRC is indexed by a wave-uniform position at run time, POS by the literal column numbers of the
//  generated scalar part -- which fold to constants once elem() is inlined)
template <class GL> struct rh_glm4r_tabs {
  static constexpr typename rh_glm4r_map<GL>::Tab SRC = rh_glm4r_map<GL>::make_src();
  static constexpr typename rh_glm4r_map<GL>::ColTab POS = rh_glm4r_map<GL>::make_pos();
};

typedef rh_glm4r_map<rh_glm<RH_GLM_TARGET>> rh_g4r;
// dynamic LDS the host has to give the kernel (read back from the module): one or two row tiles
extern "C" __device__ const int rh_glm4r_lds_bytes =
    ((2 * 64 * rh_g4r::ST * 8 + RH_LK_LDS * 4112 <= 160 * 1024) ? 2 : 1) * 64 * rh_g4r::ST * 8;
extern "C" __device__ const int rh_glm4r_waves = RH_GLM4R_W;
extern "C" __device__ const int rh_glm4r_chains_per_wave = 4 * RH_GLM4R_JG;

extern "C" __global__ void __launch_bounds__(64 * RH_GLM4R_W, RH_GLM4R_WPS)
rh_grad_glm4r_kernel(const rh_model_data d, const double *__restrict__ q, const int *__restrict__ active,
                     double *__restrict__ partial, int *__restrict__ err_out, int *__restrict__ n_running,
                     const int chains, const int nsplit, const int xcd_aware) {
  rh_lk_init();
  typedef rh_glm<RH_GLM_TARGET> GL;
  typedef rh_target<RH_GLM_TARGET> TG;
  typedef rh_g4r MP;
  constexpr int P = GL::P, W = RH_GLM4R_W, NP = MP::NP, NPP = MP::NPP, S4 = MP::S4, ST = MP::ST, JG = RH_GLM4R_JG;
  constexpr int MYP = (NP + W - 1) / W;   // LDS positions a wavefront stages per tile
  constexpr int NBUF = (2 * 64 * ST * 8 + RH_LK_LDS * 4112 <= 160 * 1024) ? 2 : 1;
  constexpr int NO = GL::NOTHER > 0 ? GL::NOTHER : 1, NU = GL::NTHU > 0 ? GL::NTHU : 1;
  extern __shared__ __attribute__((aligned(16))) double rh_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lk = lane >> 4, lb = (lane >> 2) & 3, le = lane & 3;
  const int b = blockIdx.x;
  if (b == 0 && threadIdx.x == 0) *n_running = 0;
  int bgroup, split;
  if (xcd_aware && (nsplit % 8) == 0) {
    const int xcd = b & 7, idx = b >> 3, spx = nsplit >> 3;
    split = xcd + 8 * (idx % spx);
    bgroup = idx / spx;
  } else { split = b % nsplit; bgroup = b / nsplit; }
  const int chain0 = (bgroup * W + wave) * (4 * JG);
  // this lane's JG chains: chain0 + 4 jg + le
  int cl[JG];
  bool cv[JG], mine = false;
#pragma unroll
  for (int jg = 0; jg < JG; jg++) {
    const int c = chain0 + 4 * jg + le;
    cv[jg] = c < chains;
    cl[jg] = cv[jg] ? c : chains - 1;
    mine = mine || (cv[jg] && active[cl[jg]] != 0);
  }
  const bool compute = __any(mine);
  if (!__syncthreads_or(compute ? 1 : 0)) return;   // no chain of this workgroup is waiting for a gradient
  // forward B operands: lane (k = lk, *, j = le) holds scale * theta[chain 4 jg + le][pred(lk, s)]
  double Bf[S4][JG];
#pragma unroll
  for (int s = 0; s < S4; s++) {
    const int pf = S4 * lk + s;
#pragma unroll
    for (int jg = 0; jg < JG; jg++)
      Bf[s][jg] = (pf < P) ? GL::pred_scale[pf < P ? pf : 0] * q[(size_t)cl[jg] * RH_NVARS + GL::pred_param[pf < P ? pf : 0]] : 0.0;
  }
  double thu[JG][NU];
#pragma unroll
  for (int jg = 0; jg < JG; jg++)
#pragma unroll
    for (int k = 0; k < GL::NTHU; k++) thu[jg][k] = q[(size_t)cl[jg] * RH_NVARS + GL::thu_param[k]];
  double G[S4][JG], oth[JG][NO];
#pragma unroll
  for (int s = 0; s < S4; s++)
#pragma unroll
    for (int jg = 0; jg < JG; jg++) G[s][jg] = 0.0;
#pragma unroll
  for (int jg = 0; jg < JG; jg++)
#pragma unroll
    for (int k = 0; k < NO; k++) oth[jg][k] = 0.0;
  int err = 0;

  const long long n = d.nrows[RH_GLM_TARGET];
  const long long per = (((n + 63) / 64) + nsplit - 1) / nsplit;
  long long r0 = (long long)split * per * 64, r1 = r0 + per * 64;
  if (r0 > n) r0 = n;
  if (r1 > n) r1 = n;
  const long long ntiles = (r1 - r0 + 63) / 64;
  // the constants of the tile: the 1 of an intercept, the zero pads (both buffers, once)
  for (int pos = wave; pos < NPP; pos += W) {
    const int src = rh_glm4r_tabs<GL>::SRC.v[pos];
    if (src < 0) {
#pragma unroll
      for (int bf = 0; bf < NBUF; bf++) rh_lds[(size_t)bf * 64 * ST + (size_t)lane * ST + pos] = src == -1 ? 1.0 : 0.0;
    }
  }
  double stage[MYP];
  auto fetch = [&](long long tile) {   // wave w fetches the positions w, w + W, ...; lane = row: coalesced 512 B per column
    long long row = r0 + tile * 64 + lane;   // (only called when the split has rows: n > 0)
    if (row >= n) row = n - 1;
#pragma unroll
    for (int m = 0; m < MYP; m++) {
      const int pos = wave + m * W;
      const int src = pos < NP ? rh_glm4r_tabs<GL>::SRC.v[pos < NP ? pos : 0] : -2;
      stage[m] = d.cols[TG::COL0 + (src >= 0 ? src : 0)][row];   // (no branch around the load: a constant / pad position re-reads column 0 and park drops it)
    }
  };
  auto park = [&](int buf) {           // ... and parks them row-major, in predictor order
#pragma unroll
    for (int m = 0; m < MYP; m++) {
      const int pos = wave + m * W;
      const int src = pos < NP ? rh_glm4r_tabs<GL>::SRC.v[pos < NP ? pos : 0] : -2;
      if (src >= 0) rh_lds[(size_t)buf * 64 * ST + (size_t)lane * ST + pos] = stage[m];
    }
  };
  // this lane's operand addresses inside a 16-row block (doubles): forward X[row 4 lb + le][pred(lk, s)], backward X[row 4 lb + lk][pred(le, s)]
  const int fbase = (4 * lb + le) * ST + S4 * lk, bbase = (4 * lb + lk) * ST + S4 * le;
  const int myrow = 4 * lb + lk;   // the row of this lane's four evaluations
  if (ntiles > 0) { fetch(0); park(0); }
  __syncthreads();
  // The operand reads are phased by hand (the compiler hoists all of a sub-tile's 2 S4 LDS reads to its top and waits for the lot): the
  // forward operands of the NEXT 16 rows are requested before the backward MFMAs of the current ones, the backward operands before
  // the scalar part -- every read has a phase of arithmetic to land behind, and the two sets share their registers' lifetime.
  double fa[S4];
  for (long long t = 0; t < ntiles; t++) {
    const int buf = NBUF == 2 ? (int)(t & 1) : 0;
    if (t + 1 < ntiles) fetch(t + 1);
    if (compute) {
      const double *tile = rh_lds + (size_t)buf * 64 * ST;
      const bool full = (r0 + t * 64 + 64 <= r1) && (chain0 + 4 * JG <= chains);
#pragma unroll
      for (int s = 0; s < S4; s++) fa[s] = tile[fbase + s];
#pragma unroll 1
      for (int sub = 0; sub < 4; sub++) {
        const double *ts = tile + (size_t)sub * 16 * ST;
        // forward: one operand per step feeds the JG chain groups; even and odd steps go to separate accumulators, so that 2 JG
        // independent MFMA chains are in flight (a dependent MFMA waits for the previous one to drain)
        double D0[JG], D1[JG];
#pragma unroll
        for (int jg = 0; jg < JG; jg++) { D0[jg] = 0.0; D1[jg] = 0.0; }
#pragma unroll
        for (int s = 0; s < S4; s++) {
#pragma unroll
          for (int jg = 0; jg < JG; jg++) {
            if (s & 1) D1[jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[s], Bf[s][jg], D1[jg], 0, 0, 0);
            else D0[jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[s], Bf[s][jg], D0[jg], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        double ba[S4];
#pragma unroll
        for (int s = 0; s < S4; s++) ba[s] = ts[bbase + s];
        __builtin_amdgcn_sched_barrier(0);
        // the scalar part: this lane's row, its JG chains
        const int rrow = sub * 16 + myrow;
        const bool row_ok = full || (r0 + t * 64 + rrow < r1);
        double Dv[JG];
#pragma unroll
        for (int jg = 0; jg < JG; jg++) {
          double w = 0.0, o[NO];
          GL::elem(thu[jg], D0[jg] + D1[jg], [&](int j) { return tile[(size_t)rrow * ST + rh_glm4r_tabs<GL>::POS.v[j]]; }, w, o, err);
          const bool valid = full || (row_ok && cv[jg]);
          Dv[jg] = valid ? w : 0.0;
#pragma unroll
          for (int k = 0; k < GL::NOTHER; k++) oth[jg][k] += valid ? o[k] : 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
        if (sub < 3) {
#pragma unroll
          for (int s = 0; s < S4; s++) fa[s] = ts[16 * ST + fbase + s];
        }
        __builtin_amdgcn_sched_barrier(0);
        // backward: S4 x JG independent accumulators
#pragma unroll
        for (int s = 0; s < S4; s++) {
#pragma unroll
          for (int jg = 0; jg < JG; jg++) G[s][jg] = __builtin_amdgcn_mfma_f64_4x4x4f64(ba[s], Dv[jg], G[s][jg], 0, 0, 0);
        }
      }
    }
    if (NBUF == 1) __syncthreads();
    if (t + 1 < ntiles) park(NBUF == 2 ? (buf ^ 1) : 0);
    __syncthreads();
  }
  if (compute) {
    // G[s][jg] at lane (i = lk, b = lb, j = le): block b's rows' share of sum x[pred(lk, s)] * w for chain 4 jg + le -- fold the blocks
#pragma unroll
    for (int s = 0; s < S4; s++)
#pragma unroll
      for (int jg = 0; jg < JG; jg++) {
        double v = G[s][jg];
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        const int pred = S4 * lk + s;
        if (lb == 0 && pred < P && cv[jg]) {
          double *out = partial + (((size_t)TG::ROWT * nsplit + split) * chains + cl[jg]) * RH_NACC_MAX;
          out[GL::pred_acc[pred < P ? pred : 0]] = v;
        }
      }
#pragma unroll
    for (int jg = 0; jg < JG; jg++)
#pragma unroll
      for (int k = 0; k < GL::NOTHER; k++) {   // a chain's evaluations sit in the 16 lanes that share le: fold lb and lk
        double v = oth[jg][k];
        v += __shfl_xor(v, 4, 64);
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lb == 0 && lk == 0 && cv[jg]) {
          double *out = partial + (((size_t)TG::ROWT * nsplit + split) * chains + cl[jg]) * RH_NACC_MAX;
          out[GL::other_acc[k]] = v;
        }
      }
  }
  if (__any(err != 0) && lane == 0) atomicOr(err_out, 1);
}
#pragma clang fp contract(off)
#endif  // RH_GLM_TARGET
