// mfma_f64_peak.hip -- measures the sustained v_mfma_f64_16x16x4_f64 rate on this device (probe, not a test).
// The in-repo MI355X guide lists no FP64 matrix figure; this gives the number the roofline is priced against.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, int iters, double a0, double b0)
{
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void kfma(double* out, int iters, double a0, double b0)
{
  double acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = threadIdx.x;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = fma(acc[i], a, b);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
  double* d;
  hipMalloc(&d, sizeof(double) * 256 * 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wpc = 1; wpc <= 2; ++wpc) {
    const int blocks = 256 * wpc;  // wpc workgroups of 4 waves per CU
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0, 1.0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)blocks * 4 * iters * 4 * 2048.0;
      if (rep) printf("mfma_f64_16x16x4 NACC=4 blocks/CU=%d: %.2f TFLOP/s (%.3f ms)\n", wpc, flops / ms / 1e9, ms);
    }
  }
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kfma, dim3(1024), dim3(256), 0, 0, d, iters, 1.0000001, 1e-9);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = 1024.0 * 256 * iters * 8 * 2.0;
    if (rep) printf("v_fma_f64 vector: %.2f TFLOP/s (%.3f ms)\n", flops / ms / 1e9, ms);
  }
  return 0;
}
