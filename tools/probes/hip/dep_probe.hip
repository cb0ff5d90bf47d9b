// Probe (not part of the product): latency of DEPENDENT FP64 operations in one wave -- what bounds the pivot chain of the diagonal-tile
// factorisation (NOTES_r04 section 6e: "eleven dependent steps per pivot at ~28 cycles each").  One wave per workgroup, one workgroup.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/probes/hip/dep_probe tools/probes/hip/dep_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void k_dep(double* out, long long* cyc, int n)
{
  double x = out[threadIdx.x], y = out[64 + threadIdx.x];
  const long long t0 = clock64();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (MODE == 0) x = fma(x, y, 1e-9);                                  // dependent FMA chain
      if (MODE == 1) x = __builtin_amdgcn_rsq(x) + 1.0;                    // rsq + add, dependent
      if (MODE == 2) { const double s = __builtin_amdgcn_readfirstlane((int)__double2loint(x)) * 1e-300; x = fma(x, y, s); }  // through a scalar register
      if (MODE == 3) { x = fma(x, y, 1e-9); y = fma(y, 0.999999, 1e-9); } // two independent chains
      if (MODE == 4) x = __shfl(x, 5, 64) * y + 1e-9;                      // cross-lane broadcast (readlane) + FMA, dependent
    }
  }
  const long long t1 = clock64();
  out[threadIdx.x] = x + y;
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

int main()
{
  double* d; long long* c;
  CHECK(hipMalloc(&d, 128 * 8)); CHECK(hipMalloc(&c, 8));
  double h[128]; for (int i = 0; i < 128; ++i) h[i] = 1.0 + 1e-3 * i;
  const int n = 2000;
  const char* names[5] = {"dependent v_fma_f64", "dependent v_rsq_f64 + v_add_f64", "readfirstlane -> v_fma_f64, dependent", "two independent v_fma_f64 chains (per pair)",
                          "readlane broadcast + v_fma_f64, dependent"};
  for (int m = 0; m < 5; ++m) {
    CHECK(hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
      if (m == 0) hipLaunchKernelGGL(k_dep<0>, dim3(1), dim3(64), 0, 0, d, c, n);
      if (m == 1) hipLaunchKernelGGL(k_dep<1>, dim3(1), dim3(64), 0, 0, d, c, n);
      if (m == 2) hipLaunchKernelGGL(k_dep<2>, dim3(1), dim3(64), 0, 0, d, c, n);
      if (m == 3) hipLaunchKernelGGL(k_dep<3>, dim3(1), dim3(64), 0, 0, d, c, n);
      if (m == 4) hipLaunchKernelGGL(k_dep<4>, dim3(1), dim3(64), 0, 0, d, c, n);
      CHECK(hipDeviceSynchronize());
    }
    long long cy; CHECK(hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost));
    printf("%-48s %7.1f clock64 ticks per step\n", names[m], (double)cy / (16.0 * n));
  }
  return 0;
}
