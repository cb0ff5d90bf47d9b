// Probe (not part of the product): what the ray-centric kernels (k_eval, k_lin_ray) pay for walking a ray's observations where they
// lie in ray-major order -- lane l reads record a0(l) + k, 8 + 4 bytes at a stride of one track (~7.5 records = 90 bytes) between
// lanes -- against a SLICED layout in which the k-th records of the 64 rays of a wave are contiguous (unit stride).  Thread = ray,
// 1024-thread workgroups with a 94 KB dynamic LDS block (one workgroup per compute unit, 16 waves, as k_eval at C4), FMA filler per
// observation to stand for the functor's arithmetic.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/probes/hip/rayloop_probe tools/probes/hip/rayloop_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool SLICED, int FILL>
__global__ __launch_bounds__(1024) void k_walk(const float2* __restrict__ uv, const int* __restrict__ cam, const int* __restrict__ ptr,
                                               const int* __restrict__ slice, int n_ray, double* __restrict__ out)
{
  extern __shared__ double tab[];
  for (int i = threadIdx.x; i < 200 * 35; i += 1024) tab[i] = 1.0 + 1e-9 * i;
  __syncthreads();
  const int j = blockIdx.x * 1024 + threadIdx.x;
  if (j >= n_ray) return;
  const int a0 = ptr[j], len = ptr[j + 1] - a0;
  const int base = SLICED ? slice[j >> 6] + (j & 63) : a0;
  constexpr int STEP = SLICED ? 64 : 1;
  double acc = 0;
  for (int pass = 0; pass < 2; ++pass) {
    float2 uvn = uv[base];
    int cn = cam[base];
    for (int k = 0; k < len; ++k) {
      const float2 p = uvn;
      const int c = cn;
      const int kn = min(k + 1, len - 1);
      uvn = uv[base + kn * STEP]; cn = cam[base + kn * STEP];
      const double* cb = tab + c * 35;
      double x = p.x * cb[0] + p.y * cb[1] + cb[2], y = p.x * cb[3] + p.y * cb[4] + cb[5];
#pragma unroll
      for (int f = 0; f < FILL; ++f) { x = fma(x, 1.0000001, y); y = fma(y, 0.9999999, x); }
      acc += x * y;
    }
  }
  if (acc == 123.456) out[j] = acc;
}

int main(int argc, char** argv)
{
  const int S = argc > 1 ? atoi(argv[1]) : 142, RAYS = 13432;
  const int n_ray = S * RAYS;
  // track lengths as on a C2 rig, sorted descending inside a scene (the library's ray order)
  const int hist[20] = {0, 0, 0, 0, 2032, 1937, 1750, 1583, 1526, 1466, 1175, 777, 582, 310, 161, 76, 39, 12, 3, 3};
  std::vector<int> len;
  for (int L = 19; L >= 4; --L) for (int i = 0; i < hist[L]; ++i) len.push_back(L);
  std::vector<int> ptr(n_ray + 1, 0), slice((n_ray + 63) / 64 + 1, 0);
  for (int s = 0; s < S; ++s) for (int j = 0; j < RAYS; ++j) ptr[s * RAYS + j + 1] = ptr[s * RAYS + j] + len[j];
  size_t tot_sl = 0;
  for (int w = 0; w < (n_ray + 63) / 64; ++w) { slice[w] = (int)tot_sl; const int j0 = w * 64; tot_sl += 64 * (size_t)(ptr[j0 + 1] - ptr[j0]); }
  const size_t n_obs = ptr[n_ray], cap = std::max(n_obs, tot_sl) + 4096;
  printf("%d scenes, %d rays, %zu observations, sliced layout %zu records (%.1f %% padding)\n", S, n_ray, n_obs, tot_sl, 100.0 * (tot_sl - n_obs) / n_obs);
  std::vector<int> cam(cap);
  for (size_t i = 0; i < cap; ++i) cam[i] = (int)((i * 2654435761ull >> 7) % 200);
  float2* d_uv; int *d_cam, *d_ptr, *d_slice; double* d_out;
  CHECK(hipMalloc(&d_uv, cap * 8)); CHECK(hipMemset(d_uv, 0, cap * 8));
  CHECK(hipMalloc(&d_cam, cap * 4)); CHECK(hipMemcpy(d_cam, cam.data(), cap * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&d_ptr, ptr.size() * 4)); CHECK(hipMemcpy(d_ptr, ptr.data(), ptr.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&d_slice, slice.size() * 4)); CHECK(hipMemcpy(d_slice, slice.data(), slice.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&d_out, (size_t)n_ray * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto kern) {
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3((n_ray + 1023) / 1024), dim3(1024), 94 * 1024, 0, (const float2*)d_uv, (const int*)d_cam, (const int*)d_ptr, (const int*)d_slice, n_ray, d_out);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 0 && ms < best) best = ms;
    }
    printf("%-44s %8.3f ms  (%.2f us per scene)\n", name, best, best * 1e3 / S);
  };
  run("ray-major records, no filler", k_walk<false, 0>);
  run("sliced records,    no filler", k_walk<true, 0>);
  run("ray-major records, 2 x 40 FMAs per observation", k_walk<false, 40>);
  run("sliced records,    2 x 40 FMAs per observation", k_walk<true, 40>);
  run("ray-major records, 2 x 100 FMAs per observation", k_walk<false, 100>);
  run("sliced records,    2 x 100 FMAs per observation", k_walk<true, 100>);
  return 0;
}
