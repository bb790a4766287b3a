// Lane layout of v_mfma_f64_4x4x4_4b_f64 (4 blocks of D[4x4] += A[4x4] . B[4x4], one f64 per lane for A, B and C/D), found
// empirically: one-hot A and B lanes, which D lanes light up.  The guide (cdna_hip_programming.md) gives the 16x16x4 f64 layout only.
// build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench/mfma4_layout tools/ubench/mfma4_layout.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double *a, const double *b, double *d, int n) {
  for (int t = 0; t < n; t++) {
    const double av = a[t * 64 + threadIdx.x], bv = b[t * 64 + threadIdx.x];
    d[t * 64 + threadIdx.x] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, 0.0, 0, 0, 0);
  }
}
int main() {
  // test t = la * 64 + lb: A one-hot at lane la (value 1 + la), B one-hot at lane lb (value 100 + lb)
  const int n = 64 * 64;
  std::vector<double> a((size_t)n * 64, 0.0), b((size_t)n * 64, 0.0), d((size_t)n * 64);
  for (int la = 0; la < 64; la++) for (int lb = 0; lb < 64; lb++) { const int t = la * 64 + lb; a[(size_t)t * 64 + la] = 1.0; b[(size_t)t * 64 + lb] = 1.0; }
  double *da, *db, *dd;
  hipMalloc(&da, a.size() * 8); hipMalloc(&db, b.size() * 8); hipMalloc(&dd, d.size() * 8);
  hipMemcpy(da, a.data(), a.size() * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice);
  k<<<1, 64>>>(da, db, dd, n);
  hipMemcpy(d.data(), dd, d.size() * 8, hipMemcpyDeviceToHost);
  // for every (la, lb) that produces output: the D lanes
  printf("# la lb : D lanes with a nonzero result\n");
  for (int la = 0; la < 64; la++) for (int lb = 0; lb < 64; lb++) {
    const int t = la * 64 + lb; bool any = false;
    for (int l = 0; l < 64; l++) if (d[(size_t)t * 64 + l] != 0.0) any = true;
    if (!any) continue;
    printf("%d %d :", la, lb);
    for (int l = 0; l < 64; l++) if (d[(size_t)t * 64 + l] != 0.0) printf(" %d", l);
    printf("\n");
  }
  return 0;
}
