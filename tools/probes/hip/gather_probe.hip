// Probe (not part of the product): at what rate can a camera-centric workgroup gather its rays' records?  k_schur_f's phase 1 in
// isolation -- 256 threads, three trips over a view of 514 observations: ray ids (4 B, unit stride), then NP 16-byte pieces per
// ray from (a) planes [NP][rays] or (b) records [rays][NP]; the ids of a view are clusters of the ray order as in a C2 rig (runs
// of ~25 rays with gaps of 1-3, a jump between runs); occupancy set by a dummy dynamic LDS allocation.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/probes/hip/gather_probe tools/probes/hip/gather_probe.hip
// run:   gather_probe [scenes=142]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int VIEW = 514, CAMS = 200, RAYS = 13432;

template <int NP, bool SOA, int PF, bool WIDE>
__global__ __launch_bounds__(256) void k_gather(const int* __restrict__ ids, const double2* __restrict__ rec, size_t n_ray, double* __restrict__ out)
{
  extern __shared__ double lds[];
  const int cam = blockIdx.x, sc = blockIdx.y;
  const int* my = ids + ((size_t)sc * CAMS + cam) * VIEW;
  double acc = 0;
  int gid[3];
#pragma unroll
  for (int t = 0; t < 3; ++t) gid[t] = my[min((int)threadIdx.x + t * 256, VIEW - 1)];
  if (PF == 3) {
    double2 v[3][NP];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < NP; ++k) v[t][k] = SOA ? rec[(size_t)k * n_ray + gid[t]] : rec[(size_t)gid[t] * NP + k];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int k = 0; k < NP; ++k) acc += v[t][k].x * v[t][k].y;
  }
  else {
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      double2 v[NP];
#pragma unroll
      for (int k = 0; k < NP; ++k) v[k] = SOA ? rec[(size_t)k * n_ray + gid[t]] : rec[(size_t)gid[t] * NP + k];
#pragma unroll
      for (int k = 0; k < NP; ++k) acc += v[k].x * v[k].y;
    }
  }
  if (threadIdx.x == 0) lds[0] = acc;
  if (acc == 123.456) out[blockIdx.x] = acc + lds[1];
}

// the same bytes as ONE coalesced stream per view (what a camera-major copy of the records would cost to read)
template <int NP>
__global__ __launch_bounds__(256) void k_stream(const double2* __restrict__ rec, double* __restrict__ out)
{
  extern __shared__ double lds[];
  const size_t base = ((size_t)blockIdx.y * CAMS + blockIdx.x) * VIEW * NP;
  double acc = 0;
  for (int i = threadIdx.x; i < VIEW * NP; i += 256) { const double2 v = rec[base + i]; acc += v.x * v.y; }
  if (threadIdx.x == 0) lds[0] = acc;
  if (acc == 123.456) out[blockIdx.x] = acc + lds[1];
}

int main(int argc, char** argv)
{
  const int S = argc > 1 ? atoi(argv[1]) : 142;
  const size_t n_ray = (size_t)S * RAYS;
  std::vector<int> ids((size_t)S * CAMS * VIEW);
  uint64_t st = 88172645463325252ull;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  for (int s = 0; s < S; ++s)
    for (int c = 0; c < CAMS; ++c) {
      int* v = &ids[((size_t)s * CAMS + c) * VIEW];
      // clusters: a camera sees ~20 runs of the ray order; its window moves with the camera
      int pos = (int)((int64_t)c * 40 % 600);
      int left = 0;
      for (int q = 0; q < VIEW; ++q) {
        if (left == 0) { pos += 200 + (int)(rnd() % 500); left = 15 + (int)(rnd() % 25); }
        const int r = (int)(rnd() % 100);
        pos += r < 60 ? 1 : (r < 85 ? 2 : (r < 95 ? 3 : 6));
        --left;
        v[q] = (int)((size_t)s * RAYS + (size_t)(pos % RAYS));
      }
    }
  int* d_ids; double2* d_rec; double* d_out;
  CHECK(hipMalloc(&d_ids, ids.size() * 4));
  CHECK(hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
  const size_t rec_bytes = std::max(n_ray * 8 * 16, (size_t)S * CAMS * VIEW * 8 * 16);
  CHECK(hipMalloc(&d_rec, rec_bytes));
  CHECK(hipMemset(d_rec, 0, rec_bytes));
  CHECK(hipMalloc(&d_out, 4096 * 8));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto run = [&](const char* name, auto kern, int np, size_t lds_bytes, auto... args) {
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      CHECK(hipEventRecord(e0));
      hipLaunchKernelGGL(kern, dim3(CAMS, S), dim3(256), lds_bytes, 0, args...);
      CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 0 && ms < best) best = ms;
    }
    const double bytes = (double)S * CAMS * VIEW * np * 16.0;
    printf("%-34s LDS %3zu KB: %7.3f ms  %7.1f GB/s useful  (%.2f us per scene)\n", name, lds_bytes >> 10, best, bytes / best * 1e-6, best * 1e3 / S);
  };
  for (size_t lds : {(size_t)76 << 10, (size_t)50 << 10, (size_t)38 << 10, (size_t)30 << 10, (size_t)16 << 10}) {
    run("planes 8 x 16 B, all trips ahead", k_gather<8, true, 3, false>, 8, lds, d_ids, d_rec, n_ray, d_out);
    run("records 128 B, all trips ahead", k_gather<8, false, 3, false>, 8, lds, d_ids, d_rec, n_ray, d_out);
    run("planes 6 x 16 B, all trips ahead", k_gather<6, true, 3, false>, 6, lds, d_ids, d_rec, n_ray, d_out);
    run("records 96 B, all trips ahead", k_gather<6, false, 3, false>, 6, lds, d_ids, d_rec, n_ray, d_out);
    run("planes 6 x 16 B, trip by trip", k_gather<6, true, 1, false>, 6, lds, d_ids, d_rec, n_ray, d_out);
    run("records 96 B, trip by trip", k_gather<6, false, 1, false>, 6, lds, d_ids, d_rec, n_ray, d_out);
    run("camera-major stream 96 B / obs", k_stream<6>, 6, lds, d_rec, d_out);
  }
  return 0;
}
