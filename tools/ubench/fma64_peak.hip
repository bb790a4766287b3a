// fp64 VALU ceiling on MI355X: v_fma_f64 issue rate with different operand mixes and occupancies.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench/fma64_peak tools/ubench/fma64_peak.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int NACC, int MODE>
__global__ void __launch_bounds__(64) k_fma(double *out, const double *in, int iters) {
  double acc[NACC];
  const double a = in[0], b = in[1];          // wave-uniform (SGPR) operands
  const double one = in[3000];                // 1.0 at run time: keeps the fma from folding back into an add
  const double va = in[2 + threadIdx.x];      // per-lane (VGPR) operand
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = in[threadIdx.x + i];
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) {
      if (MODE == 0) acc[i] = __builtin_fma(acc[i], a, b);        // 2 SGPR-pair operands
      else if (MODE == 1) acc[i] = __builtin_fma(acc[i], a, va);  // 1 SGPR-pair operand
      else if (MODE == 2) acc[i] = __builtin_fma(acc[i], va, va); // VGPR only
      else if (MODE == 3) acc[i] = (i % 9 < 7) ? __builtin_fma(acc[i], a, va) : acc[i] + va; // 7 fma : 2 add
      else if (MODE == 5) acc[i] = acc[i] + va;                      // v_add_f64 only
      else if (MODE == 6) acc[i] = acc[i] * va;                      // v_mul_f64 only
      else if (MODE == 7) acc[i] = (i % 9 < 7) ? __builtin_fma(acc[i], a, va) : __builtin_fma(acc[i], one, va); // add as fma(x, 1, y)
      else acc[i] = __builtin_fma(va, a, acc[i]);                 // accumulate form
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) s += acc[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int NACC, int MODE>
void run(const char *name, int waves_per_simd, double *out, double *in) {
  const int iters = 20000, blocks = 256 * 4 * waves_per_simd;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k_fma<NACC, MODE><<<blocks, 64>>>(out, in, 100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_fma<NACC, MODE><<<blocks, 64>>>(out, in, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double instr = (double)blocks * iters * NACC;  // wave-instructions
  const double flops = instr * 64 * (MODE == 3 || MODE == 7 ? (7 * 2 + 2) / 9.0 : MODE == 5 || MODE == 6 ? 1.0 : 2.0);
  printf("%-28s nacc=%2d waves/SIMD=%d  %8.3f ms  %6.2f TFLOP/s  %.3f wave-instr/cycle/SIMD@2.4GHz\n", name, NACC, waves_per_simd,
         ms, flops / ms * 1e-9, instr / 1024.0 / (ms * 1e-3 * 2.4e9));
}

int main() {
  double *in, *out;
  hipMalloc(&in, 4096 * 8); hipMalloc(&out, 256 * 4 * 8 * 64 * 8);
  std::vector<double> h(4096, 1.0000001);
  hipMemcpy(in, h.data(), 4096 * 8, hipMemcpyHostToDevice);
  hipMemcpy(in + 3000, std::vector<double>(1, 1.0).data(), 8, hipMemcpyHostToDevice);
  for (int w = 2; w <= 4; w++) {
    run<16, 5>("add only", w, out, in);
    run<16, 6>("mul only", w, out, in);
    run<18, 7>("7 fma : 2 fma(x,1,y)", w, out, in);
    run<16, 0>("fma(acc, s, s)", w, out, in);
    run<16, 1>("fma(acc, s, v)", w, out, in);
    run<16, 2>("fma(acc, v, v)", w, out, in);
    run<18, 3>("7 fma : 2 add", w, out, in);
    run<16, 4>("fma(v, s, acc)", w, out, in);
    run<40, 4>("fma(v, s, acc) 40 acc", w, out, in);
  }
  return 0;
}
