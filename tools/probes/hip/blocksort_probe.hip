// Probe (not part of the product): one workgroup per view sorting its tracks' keys with rocprim::block_radix_sort (stable, value =
// track number) against the batch-wide rocprim::radix_sort_pairs the view build uses (22 launches for 24 bits).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/probes/hip/blocksort_probe tools/probes/hip/blocksort_probe.hip
#include <hip/hip_runtime.h>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <rocprim/block/block_radix_sort.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int CAP = 16384;
template <int IPT>
__global__ __launch_bounds__(1024) void k_sort(const unsigned* __restrict__ key_in, const int* __restrict__ n, int* __restrict__ val_out, int bits)
{
  using Sort = rocprim::block_radix_sort<unsigned, 1024, IPT, int>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typename Sort::storage_type& st = *reinterpret_cast<typename Sort::storage_type*>(smem);
  const int base = blockIdx.x * CAP, cnt = n[blockIdx.x];
  if (cnt > 1024 * IPT) return;
  unsigned k[IPT]; int v[IPT];
#pragma unroll
  for (int i = 0; i < IPT; ++i) { const int t = threadIdx.x * IPT + i; k[i] = t < cnt ? key_in[base + t] : 0xffffffffu; v[i] = t; }
  Sort().sort(k, v, st, 0, bits);
#pragma unroll
  for (int i = 0; i < IPT; ++i) { const int t = threadIdx.x * IPT + i; if (t < cnt) val_out[base + t] = v[i]; }
}
int main(int argc, char** argv)
{
  const int V = argc > 1 ? atoi(argv[1]) : 19, N = argc > 2 ? atoi(argv[2]) : 13432;
  std::vector<unsigned> key((size_t)V * CAP, 0); std::vector<int> cnt(V, N);
  for (int v = 0; v < V; ++v) for (int t = 0; t < N; ++t) key[(size_t)v * CAP + t] = (unsigned)(((t * 2654435761u) >> 9) % 3800u);  // (longest - len) n_cam + first: a few thousand buckets
  unsigned* d_key; int *d_cnt, *d_val; CHECK(hipMalloc(&d_key, key.size() * 4)); CHECK(hipMalloc(&d_cnt, V * 4)); CHECK(hipMalloc(&d_val, key.size() * 4));
  CHECK(hipMemcpy(d_key, key.data(), key.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_cnt, cnt.data(), V * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto launch) {
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) { CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1)); float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (r && ms < best) best = ms; }
    printf("%-52s %8.1f us\n", name, best * 1e3);
  };
  CHECK(hipFuncSetAttribute((const void*)k_sort<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  const size_t s16 = sizeof(rocprim::block_radix_sort<unsigned, 1024, 16, int>::storage_type), s4 = sizeof(rocprim::block_radix_sort<unsigned, 1024, 4, int>::storage_type);
  time("block sort, 16 items per thread, 13 bits", [&] { hipLaunchKernelGGL(k_sort<16>, dim3(V), dim3(1024), s16, 0, d_key, d_cnt, d_val, 13); });
  time("block sort, 16 items per thread, 23 bits", [&] { hipLaunchKernelGGL(k_sort<16>, dim3(V), dim3(1024), s16, 0, d_key, d_cnt, d_val, 23); });
  std::vector<int> val((size_t)V * CAP); CHECK(hipMemcpy(val.data(), d_val, val.size() * 4, hipMemcpyDeviceToHost));
  { std::vector<int> ref(N); for (int t = 0; t < N; ++t) ref[t] = t; std::stable_sort(ref.begin(), ref.end(), [&](int a, int b) { return key[a] < key[b]; });
    int bad = 0; for (int t = 0; t < N; ++t) bad += val[t] != ref[t]; printf("differences from std::stable_sort: %d\n", bad); }
  for (int v = 0; v < V; ++v) cnt[v] = 3000;
  CHECK(hipMemcpy(d_cnt, cnt.data(), V * 4, hipMemcpyHostToDevice));
  time("block sort, 4 items per thread (3000 tracks), 23 bits", [&] { hipLaunchKernelGGL(k_sort<4>, dim3(V), dim3(1024), s4, 0, d_key, d_cnt, d_val, 23); });
  // the batch-wide sort of the product: 64-bit keys, bits 24 .. 47 + view bits
  const size_t tot = (size_t)V * N;
  std::vector<unsigned long long> k64(tot); std::vector<int> v32(tot);
  for (int v = 0; v < V; ++v) for (int t = 0; t < N; ++t) { k64[(size_t)v * N + t] = ((unsigned long long)v << 47) | ((unsigned long long)key[(size_t)v * CAP + t] << 24) | (unsigned)t; v32[(size_t)v * N + t] = t; }
  unsigned long long *d_k0, *d_k1; int *d_v0, *d_v1; CHECK(hipMalloc(&d_k0, tot * 8)); CHECK(hipMalloc(&d_k1, tot * 8)); CHECK(hipMalloc(&d_v0, tot * 4)); CHECK(hipMalloc(&d_v1, tot * 4));
  CHECK(hipMemcpy(d_k0, k64.data(), tot * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(d_v0, v32.data(), tot * 4, hipMemcpyHostToDevice));
  unsigned end_bit = 47; while ((1u << (end_bit - 47)) < (unsigned)V) ++end_bit;
  size_t tmp_bytes = 0; CHECK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_k0, d_k1, d_v0, d_v1, tot, 24, end_bit, 0));
  void* tmp; CHECK(hipMalloc(&tmp, tmp_bytes + 256));
  time("rocprim::radix_sort_pairs over the batch", [&] { (void)rocprim::radix_sort_pairs(tmp, tmp_bytes, d_k0, d_k1, d_v0, d_v1, tot, 24, end_bit, 0); });
  return 0;
}
