// Probe (not part of the product): issue cost of the FP64 instructions of the diagonal tile's sweep -- one wave, 16 independent
// accumulators, cycles per instruction (clock64): v_fma_f64, v_fmac_f64_dpp row_newbcast, v_mov_b64_dpp, v_mul_f64, v_rsq_f64.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/probes/hip/issue_probe tools/probes/hip/issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ __launch_bounds__(64) void k(double* out, long long* cyc, int reps)
{
  double a[16], b = 1.0 + threadIdx.x * 1e-9, c = 1e-9;
  for (int i = 0; i < 16; ++i) a[i] = i + threadIdx.x;
  long long t = 0;
  for (int it = 0; it < reps; ++it) {
    const long long t0 = clock64();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        if (KIND == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c));
        if (KIND == 1) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(b), "v"(c));
        if (KIND == 2) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(a[i]) : "v"(b));
        if (KIND == 3) asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a[i]) : "v"(b), "v"(c));
        if (KIND == 4) asm volatile("v_rsq_f64 %0, %1" : "=v"(a[i]) : "v"(b));
        if (KIND == 5) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
        if (KIND == 6) asm volatile("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(reinterpret_cast<int*>(&a[i])[0]) : "v"(reinterpret_cast<int*>(&b)[0]));
        if (KIND == 7) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b));
        if (KIND == 8) asm volatile("v_fma_f64 %0, %1, %2, neg(0)" : "=v"(a[i]) : "v"(b), "v"(c));
        if (KIND == 9) asm volatile("v_add_f64 %0, %1, %2" : "=v"(a[i]) : "v"(b), "v"(c));
        if (KIND == 10) asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(a[i]) : "v"(b), "v"(c), "v"(b));
        if (KIND == 11) asm volatile("v_mul_f64 %0, %1, 0.5" : "=v"(a[i]) : "v"(b));
        if (KIND == 12) asm volatile("v_fma_f64 %0, -%1, %2, %3" : "=v"(a[i]) : "v"(b), "v"(c), "s"(1.5));
      }
    }
    t += clock64() - t0;
  }
  double s = 0; for (int i = 0; i < 16; ++i) s += a[i];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t;
}
typedef double pd4 __attribute__((ext_vector_type(4)));
// v_mfma_f64_16x16x4_f64 from ONE wave: NACC independent accumulators used round-robin (NACC = 1: a dependent chain)
template <int NACC>
__global__ __launch_bounds__(256) void km(double* out, long long* cyc, int reps)
{
  pd4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = pd4{0, 0, 0, 0};
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  long long t = 0;
  for (int it = 0; it < reps; ++it) {
    const long long t0 = clock64();
#pragma unroll
    for (int r = 0; r < 48 / NACC; ++r)
#pragma unroll
      for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    t += clock64() - t0;
  }
  double s = 0; for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[0] = t;
}
int main()
{
  double* o; long long* c; hipMalloc(&o, 64 * 8); hipMalloc(&c, 8);
  const char* names[] = {"v_fma_f64", "v_fmac_f64_dpp row_newbcast", "v_mov_b64_dpp row_newbcast", "v_mul_f64", "v_rsq_f64", "v_fmac_f64", "v_mov_b32_dpp row_newbcast", "v_mul_f64 in place", "v_fma_f64 a b neg(0)", "v_add_f64", "v_fma_f64 =v 3 inputs", "v_mul_f64 x 0.5", "v_fma_f64 -a b s[1.5]"};
  const int reps = 1000;
  for (int kind = 0; kind < 13; ++kind) {
    for (int w = 0; w < 2; ++w) {
      switch (kind) {
        case 0: hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, o, c, reps); break;
        case 1: hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, o, c, reps); break;
        case 2: hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, o, c, reps); break;
        case 3: hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, o, c, reps); break;
        case 4: hipLaunchKernelGGL(k<4>, dim3(1), dim3(64), 0, 0, o, c, reps); break;
        case 5: hipLaunchKernelGGL(k<5>, dim3(1), dim3(64), 0, 0, o, c, reps); break;
        case 6: hipLaunchKernelGGL(k<6>, dim3(1), dim3(64), 0, 0, o, c, reps); break;
        case 7: hipLaunchKernelGGL(k<7>, dim3(1), dim3(64), 0, 0, o, c, reps); break;
        case 8: hipLaunchKernelGGL(k<8>, dim3(1), dim3(64), 0, 0, o, c, reps); break;
        case 9: hipLaunchKernelGGL(k<9>, dim3(1), dim3(64), 0, 0, o, c, reps); break;
        case 10: hipLaunchKernelGGL(k<10>, dim3(1), dim3(64), 0, 0, o, c, reps); break;
        case 11: hipLaunchKernelGGL(k<11>, dim3(1), dim3(64), 0, 0, o, c, reps); break;
        default: hipLaunchKernelGGL(k<12>, dim3(1), dim3(64), 0, 0, o, c, reps); break;
      }
      hipDeviceSynchronize();
    }
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf("%-30s %.2f cycles per instruction (64 independent, one wave)\n", names[kind], (double)cy / reps / 64);
  }
  for (int nacc = 1; nacc <= 4; ++nacc) {
    for (int w = 0; w < 2; ++w) {
      if (nacc == 1) hipLaunchKernelGGL(km<1>, dim3(1), dim3(64), 0, 0, o, c, reps);
      else if (nacc == 2) hipLaunchKernelGGL(km<2>, dim3(1), dim3(64), 0, 0, o, c, reps);
      else if (nacc == 3) hipLaunchKernelGGL(km<3>, dim3(1), dim3(64), 0, 0, o, c, reps);
      else hipLaunchKernelGGL(km<4>, dim3(1), dim3(64), 0, 0, o, c, reps);
      hipDeviceSynchronize();
    }
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf("v_mfma_f64_16x16x4_f64, %d accumulator(s) in turn: %.1f cycles per instruction (one wave)\n", nacc, (double)cy / reps / 48);
  }
  for (int waves = 2; waves <= 4; waves += 2) {  // the same from 2 / 4 waves of ONE workgroup (a SIMD each): is the FP64 matrix rate per SIMD or per compute unit?
    for (int w = 0; w < 2; ++w) { hipLaunchKernelGGL(km<4>, dim3(1), dim3(64 * waves), 0, 0, o, c, reps); hipDeviceSynchronize(); }
    long long cy; hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
    printf("v_mfma_f64_16x16x4_f64, %d waves of one workgroup at once: %.1f cycles per instruction of wave 0\n", waves, (double)cy / reps / 48);
  }
  return 0;
}
