// Probe (not part of the product): what rocprofv3's FETCH_SIZE reports on gfx950 for the access patterns of this library's
// kernels, each over a known number of bytes of a buffer far larger than the 256 MB memory-side cache.  MI355X_MICROARCH.md:
// "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read (16 B/lane) ... other access widths are
// uncalibrated: calibrate on a known byte count in your own access pattern".  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- tests/probes/fetch_calib
// and divide FETCH_SIZE x 1024 by the byte count printed here (profiles/make_traffic.py does).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tests/probes/fetch_calib tests/probes/fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// 16 B per lane, unit stride (tile copies of the factorisation, camera-table staging)
__global__ void pat_stream16(const double2* __restrict__ p, size_t n, double* out)
{
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = p[i]; acc += v.x + v.y; }
  if (acc == 123.456) out[0] = acc;
}
// 8 B per lane, unit stride (pixel stream of the camera pass)
__global__ void pat_stream8(const double* __restrict__ p, size_t n, double* out)
{
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 123.456) out[0] = acc;
}
// 4 B per lane, unit stride (entry records, ray ids)
__global__ void pat_stream4(const float* __restrict__ p, size_t n, double* out)
{
  float acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += p[i];
  if (acc == 123.456f) out[0] = acc;
}
// one 96-byte row per lane, consecutive lanes consecutive rows (k_schur phase 1: a camera's W rows)
__global__ void pat_rows96(const double* __restrict__ p, size_t rows, double* out)
{
  double acc = 0;
  for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (size_t)gridDim.x * blockDim.x) {
    const double* q = p + r * 12;
#pragma unroll
    for (int k = 0; k < 12; ++k) acc += q[k];
  }
  if (acc == 123.456) out[0] = acc;
}
// one 96-byte row per lane at a pseudo-random row (k_schur phase 2: W_b rows of the camera-pair entries)
__global__ void pat_gather96(const double* __restrict__ p, size_t rows, size_t n, double* out)
{
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = (i * 2654435761ull + 12345ull) % rows;
    const double* q = p + r * 12;
#pragma unroll
    for (int k = 0; k < 12; ++k) acc += q[k];
  }
  if (acc == 123.456) out[0] = acc;
}
// one aligned 64-byte record per lane at a pseudo-random index (rayrec / (E, z) records of the camera passes)
__global__ void pat_gather64(const double* __restrict__ p, size_t recs, size_t n, double* out)
{
  double acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t r = (i * 2654435761ull + 12345ull) % recs;
    const double* q = p + r * 8;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc += q[k];
  }
  if (acc == 123.456) out[0] = acc;
}

int main()
{
  const size_t bytes = (size_t)2 << 30;
  void* buf; double* out;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 0, bytes);
  hipDeviceSynchronize();
  const dim3 grid(256 * 16), block(256);
  const size_t n_gather = (size_t)8 << 20;  // gathered rows / records
  hipLaunchKernelGGL(pat_stream16, grid, block, 0, 0, (const double2*)buf, bytes / 16, out);
  hipLaunchKernelGGL(pat_stream8, grid, block, 0, 0, (const double*)buf, bytes / 8, out);
  hipLaunchKernelGGL(pat_stream4, grid, block, 0, 0, (const float*)buf, bytes / 4, out);
  hipLaunchKernelGGL(pat_rows96, grid, block, 0, 0, (const double*)buf, bytes / 96, out);
  hipLaunchKernelGGL(pat_gather96, grid, block, 0, 0, (const double*)buf, bytes / 96, n_gather, out);
  hipLaunchKernelGGL(pat_gather64, grid, block, 0, 0, (const double*)buf, bytes / 64, n_gather, out);
  hipDeviceSynchronize();
  printf("{\"pat_stream16\": %zu, \"pat_stream8\": %zu, \"pat_stream4\": %zu, \"pat_rows96\": %zu, \"pat_gather96\": %zu, \"pat_gather64\": %zu}\n", bytes, bytes,
         bytes, (bytes / 96) * 96, n_gather * 96, n_gather * 64);
  return 0;
}
