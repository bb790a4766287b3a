// fp64 VALU issue rate in CORE CYCLES (s_memtime), independent of the clock the part happens to run at.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fma64_cycles tools/ubench/fma64_cycles.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <map>

template <int NACC, int MODE>
__global__ void __launch_bounds__(64) k(double *out, unsigned long long *cyc, const double *in, int iters) {
  double acc[NACC];
  const double a = in[0];
  const double one = in[3000];
  const double va = in[2 + threadIdx.x];
  const double vb = in[130 + threadIdx.x];
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = in[threadIdx.x + i];
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) {
      if (MODE == 0) acc[i] = __builtin_fma(va, a, acc[i]);                                   // accumulate-form FMA
      else if (MODE == 1) acc[i] = (i % 9 < 7) ? __builtin_fma(va, a, acc[i]) : acc[i] + va;  // 7 fma : 2 add
      else if (MODE == 2) acc[i] = (i % 9 < 7) ? __builtin_fma(va, a, acc[i]) : __builtin_fma(va, one, acc[i]);
      else if (MODE == 4) {   // v_fmac_f64 with a DPP row_newbcast source: the lane-per-chain GLM formulation's broadcast FMA
        if (i % 4 == 0) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(va), "v"(vb));
        else if (i % 4 == 1) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(va), "v"(vb));
        else if (i % 4 == 2) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:10 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(va), "v"(vb));
        else asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:15 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(va), "v"(vb));
      }
      else acc[i] = acc[i] + va;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) s += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    cyc[4 * blockIdx.x] = t1 - t0; cyc[4 * blockIdx.x + 1] = r1 - r0;
    cyc[4 * blockIdx.x + 2] = ((unsigned long long)(xcc & 0xf) << 32) | (hw & 0xfffffff0u);  // everything but the wave slot
    cyc[4 * blockIdx.x + 3] = t0;
  }
}

// v_mfma_f64_16x16x4_f64: NT independent 16x16 accumulator tiles per wavefront (4 doubles per lane each)
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NT>
__global__ void __launch_bounds__(64) k_mfma(double *out, unsigned long long *cyc, const double *in, int iters) {
  d4 acc[NT];
  const double a = in[2 + threadIdx.x], b = in[70 + threadIdx.x];
#pragma unroll
  for (int i = 0; i < NT; i++) acc[i] = d4{in[i], in[i + 1], in[i + 2], in[i + 3]};
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NT; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NT; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    cyc[4 * blockIdx.x] = t1 - t0; cyc[4 * blockIdx.x + 1] = r1 - r0;
    cyc[4 * blockIdx.x + 2] = ((unsigned long long)(xcc & 0xf) << 32) | (hw & 0xfffffff0u);
    cyc[4 * blockIdx.x + 3] = t0;
  }
}
// v_mfma_f64_4x4x4_4b_f64: four 4x4x4 blocks per instruction (512 flop), one accumulator double per lane; NT independent accumulators
template <int NT>
__global__ void __launch_bounds__(64) k_mfma4(double *out, unsigned long long *cyc, const double *in, int iters) {
  double acc[NT];
  const double a = in[2 + threadIdx.x], b = in[70 + threadIdx.x];
#pragma unroll
  for (int i = 0; i < NT; i++) acc[i] = in[i];
  __builtin_amdgcn_s_barrier();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NT; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NT; i++) s += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    cyc[4 * blockIdx.x] = t1 - t0; cyc[4 * blockIdx.x + 1] = r1 - r0;
    cyc[4 * blockIdx.x + 2] = ((unsigned long long)(xcc & 0xf) << 32) | (hw & 0xfffffff0u);
    cyc[4 * blockIdx.x + 3] = t0;
  }
}
template <int NT, bool SMALL>
void run_mfma_shape(int w, double *out, unsigned long long *cyc, double *in) {
  const int iters = 20000, blocks = 256 * 4 * w;
  if (SMALL) k_mfma4<NT><<<blocks, 64>>>(out, cyc, in, iters); else k_mfma<NT><<<blocks, 64>>>(out, cyc, in, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(4 * blocks);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::map<unsigned long long, std::pair<int, double>> simd;
  std::vector<double> c(blocks), r(blocks);
  for (int i = 0; i < blocks; i++) {
    c[i] = (double)h[4 * i]; r[i] = (double)h[4 * i + 1];
    auto &e = simd[h[4 * i + 2]];
    e.first += 1; e.second = std::max(e.second, c[i]);
  }
  std::sort(c.begin(), c.end()); std::sort(r.begin(), r.end());
  const double instr = (double)iters * NT, flop = SMALL ? 512.0 : 2048.0;
  double rate = 0; int n = 0;
  for (auto &kv : simd) { rate += kv.second.first * instr / kv.second.second; n++; }
  printf("%-26s acc=%2d blocks/SIMD=%d  median cycles/mfma/wave %.2f  clock %.0f MHz | %.4f mfma/cycle/SIMD = %.1f flop/cycle/SIMD (%.0f flop each; VALU fma: 32)\n",
         SMALL ? "v_mfma_f64_4x4x4_4b_f64" : "v_mfma_f64_16x16x4_f64", NT, w, c[blocks / 2] / instr, c[blocks / 2] / r[blocks / 2] * 100.0, rate / n,
         rate / n * flop, flop);
}

template <int NT>
void run_mfma(int w, double *out, unsigned long long *cyc, double *in) {
  const int iters = 20000, blocks = 256 * 4 * w;
  k_mfma<NT><<<blocks, 64>>>(out, cyc, in, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(4 * blocks);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::map<unsigned long long, std::pair<int, double>> simd;
  std::vector<double> c(blocks), r(blocks);
  for (int i = 0; i < blocks; i++) {
    c[i] = (double)h[4 * i]; r[i] = (double)h[4 * i + 1];
    auto &e = simd[h[4 * i + 2]];
    e.first += 1; e.second = std::max(e.second, c[i]);
  }
  std::sort(c.begin(), c.end()); std::sort(r.begin(), r.end());
  const double instr = (double)iters * NT;
  double rate = 0; int n = 0;
  for (auto &kv : simd) { rate += kv.second.first * instr / kv.second.second; n++; }
  printf("v_mfma_f64_16x16x4_f64     tiles=%2d blocks/SIMD=%d  median cycles/mfma/wave %.2f  clock %.0f MHz | %.4f mfma/cycle/SIMD = %.1f flop/cycle/SIMD (2048 flop each; VALU fma: 64 lanes x 2 / 4 cycles = 32)\n",
         NT, w, c[blocks / 2] / instr, c[blocks / 2] / r[blocks / 2] * 100.0, rate / n, rate / n * 2048.0);
}

template <int NACC, int MODE>
void run(const char *name, int w, double *out, unsigned long long *cyc, double *in) {
  const int iters = 200000, blocks = 256 * 4 * w;
  k<NACC, MODE><<<blocks, 64>>>(out, cyc, in, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(4 * blocks);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::vector<double> c(blocks), r(blocks);
  std::map<unsigned long long, std::pair<int, double>> simd;  // SIMD -> (waves, longest wave)
  for (int i = 0; i < blocks; i++) {
    c[i] = (double)h[4 * i]; r[i] = (double)h[4 * i + 1];
    auto &e = simd[h[4 * i + 2]];
    e.first += 1; e.second = std::max(e.second, c[i]);
  }
  std::sort(c.begin(), c.end()); std::sort(r.begin(), r.end());
  const double med = c[blocks / 2], medr = r[blocks / 2];
  const double instr = (double)iters * NACC;
  int hist[16] = {0}; double rate[16] = {0};
  for (auto &kv : simd) { int n = std::min(15, kv.second.first); hist[n]++; rate[n] += n * instr / kv.second.second; }
  printf("%-22s nacc=%2d blocks/SIMD=%d  median cycles/instr/wave %.3f  clock %.0f MHz | SIMDs used %zu:", name, NACC, w, med / instr, med / medr * 100.0, simd.size());
  for (int n = 1; n < 16; n++) if (hist[n]) printf("  [%d waves: %d SIMDs, %.3f instr/cycle]", n, hist[n], rate[n] / hist[n]);
  printf("\n");
}

int main(int argc, char **argv) {
  double *in, *out; unsigned long long *cyc;
  hipMalloc(&in, 4096 * 8); hipMalloc(&out, 256 * 4 * 8 * 64 * 8); hipMalloc(&cyc, 256 * 4 * 8 * 4 * 8);
  std::vector<double> h(4096, 1.0000001); h[3000] = 1.0;
  hipMemcpy(in, h.data(), 4096 * 8, hipMemcpyHostToDevice);
  if (argc > 1 && argv[1][0] == 'm') {   // `fma64_cycles mfma`: the two fp64 matrix shapes only, two wavefronts per SIMD -- the run the
    // counters (SQ_INSTS_MFMA, SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES) are collected on: 4 dispatches, in this order
    run_mfma_shape<4, false>(2, out, cyc, in); run_mfma_shape<8, false>(2, out, cyc, in);
    run_mfma_shape<8, true>(2, out, cyc, in); run_mfma_shape<16, true>(2, out, cyc, in);
    return 0;
  }
  if (argc > 1) {   // `fma64_cycles dpp`: only the DPP-broadcast FMA against the plain one
    for (int w = 1; w <= 2; w++) {
      run<16, 0>("fma(v, s, acc)", w, out, cyc, in);
      run<16, 4>("fmac_dpp row_newbcast", w, out, cyc, in);
      run<40, 4>("fmac_dpp row_newbcast", w, out, cyc, in);
    }
    return 0;
  }
  for (int w = 1; w <= 4; w++) {
    run<16, 0>("fma(v, s, acc)", w, out, cyc, in);
    run<40, 0>("fma(v, s, acc)", w, out, cyc, in);
    run<18, 1>("7 fma : 2 add", w, out, cyc, in);
    run<18, 2>("7 fma : 2 fma(v,1,acc)", w, out, cyc, in);
    run<16, 3>("add only", w, out, cyc, in);
  }
  for (int w = 1; w <= 3; w++) { run_mfma<4>(w, out, cyc, in); run_mfma<8>(w, out, cyc, in); }
  return 0;
}
