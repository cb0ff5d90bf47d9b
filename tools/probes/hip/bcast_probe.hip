#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double* out, long long* cyc, int iters)
{
  double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double l = 1e-9 * threadIdx.x + 1e-7;
  // (A) readlane pair + fma, 8 independent accumulators
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#define RLF(acc, q) { int lo = __builtin_amdgcn_readlane(__double2loint(l), q), hi = __builtin_amdgcn_readlane(__double2hiint(l), q); double s = __hiloint2double(hi, lo); acc = fma(-l, s, acc); }
    RLF(a0, 1) RLF(a1, 2) RLF(a2, 3) RLF(a3, 4) RLF(a4, 5) RLF(a5, 6) RLF(a6, 7) RLF(a7, 8)
    __builtin_amdgcn_sched_barrier(0);
  }
  long long t1 = clock64();
  // (B) v_fmac_f64 dpp row_newbcast
  for (int it = 0; it < iters; ++it) {
#define FD(acc, q) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #q " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(l), "v"(l));
    FD(a0, 1) FD(a1, 2) FD(a2, 3) FD(a3, 4) FD(a4, 5) FD(a5, 6) FD(a6, 7) FD(a7, 8)
  }
  long long t2 = clock64();
  // (C) v_mov_b64 dpp + fma
  for (int it = 0; it < iters; ++it) {
#define MD(acc, q) { double s; asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #q " row_mask:0xf bank_mask:0xf" : "=v"(s) : "v"(l)); acc = fma(-l, s, acc); }
    MD(a0, 1) MD(a1, 2) MD(a2, 3) MD(a3, 4) MD(a4, 5) MD(a5, 6) MD(a6, 7) MD(a7, 8)
    __builtin_amdgcn_sched_barrier(0);
  }
  long long t3 = clock64();
  // (D) plain fma
  for (int it = 0; it < iters; ++it) {
    a0 = fma(-l, a1, a0); a1 = fma(-l, a2, a1); a2 = fma(-l, a3, a2); a3 = fma(-l, a4, a3); a4 = fma(-l, a5, a4); a5 = fma(-l, a6, a5); a6 = fma(-l, a7, a6); a7 = fma(-l, a0, a7);
    __builtin_amdgcn_sched_barrier(0);
  }
  long long t4 = clock64();
  out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; cyc[3] = t4 - t3; }
}
int main()
{
  double* o; long long* c; hipMalloc(&o, 64 * 8); hipMalloc(&c, 64);
  const int iters = 1000;
  for (int r = 0; r < 2; ++r) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, c, iters); hipDeviceSynchronize(); }
  long long h[4]; hipMemcpy(h, c, 32, hipMemcpyDeviceToHost);
  printf("per update (8 independent chains): readlane x2 + fma %.1f cycles; v_fmac_f64_dpp row_newbcast %.1f; v_mov_b64_dpp + fma %.1f; plain fma %.1f\n",
         h[0] / (8.0 * iters), h[1] / (8.0 * iters), h[2] / (8.0 * iters), h[3] / (8.0 * iters));
  return 0;
}
