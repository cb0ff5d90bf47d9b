// Probe (not part of the product): cycles of the diagonal-tile factorisation alone, one workgroup, tile resident in LDS.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/probes/hip/diag_probe tools/probes/hip/diag_probe.hip
// (phase stamps were used with an earlier single-wave version)
#include "../../../ptz-calib_amd/csrc/ptz_chol.hip"
#include <vector>
#include <cmath>
namespace ptz {
namespace {
__global__ __launch_bounds__(256) void diag_probe_kernel(CholBatch cb, const double* tile, int reps, long long* out)
{
  __shared__ __attribute__((aligned(16))) double As[NB * LD];
  __shared__ __attribute__((aligned(16))) double Dv[4][DB * LDD];
  __shared__ int ok;
  long long cyc = 0, wall = 0;
  for (int r = 0; r < reps; ++r) {
    tile_g2s<256, false>(tile, NB, As);
    __syncthreads();
    const long long c0 = clock64(), w0 = wall_clock64();
    diag_factor_tile(As, Dv, &ok, cb, 0, 0, NB);
    __syncthreads();
    cyc += clock64() - c0; wall += wall_clock64() - w0;
  }
  if (threadIdx.x == 0) { out[0] = cyc; out[1] = wall; }
}
}  // namespace
}  // namespace ptz
int main()
{
  using namespace ptz;
  const int n = NB;
  std::vector<double> M(n * (n + 8)), A(n * n);
  unsigned long long s = 12345;
  for (auto& v : M) { s = s * 6364136223846793005ull + 1442695040888963407ull; v = ((double)(s >> 11) / 9007199254740992.0) - 0.5; }
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double t = i == j ? 1e-3 : 0; for (int k = 0; k < n + 8; ++k) t += M[i * (n + 8) + k] * M[j * (n + 8) + k]; A[i * n + j] = t; }
  double *dA, *dL, *dD; int* dfail; long long* dout;
  hipMalloc(&dA, sizeof(double) * n * n); hipMalloc(&dL, sizeof(double) * n * n); hipMalloc(&dD, sizeof(double) * 4 * 256);
  hipMalloc(&dfail, 4); hipMalloc(&dout, 16); hipMemset(dfail, 0, 4);
  hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
  CholBatch cb; cb.count = 1; cb.np = NB; cb.A = dA; cb.Ldiag = dL; cb.Dinv = dD; cb.fail = dfail; cb.n = nullptr;
  const int reps = 200;
  for (int it = 0; it < 3; ++it) {
    hipLaunchKernelGGL(diag_probe_kernel, dim3(1), dim3(256), 0, 0, cb, dA, reps, dout);
    hipDeviceSynchronize();
  }
  long long h[2]; hipMemcpy(h, dout, 16, hipMemcpyDeviceToHost);
  std::vector<double> L(n * n); hipMemcpy(L.data(), dL, sizeof(double) * n * n, hipMemcpyDeviceToHost);
  // check: L L^T = A
  double err = 0, amax = 0;
  for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double t = 0; for (int k = 0; k <= j; ++k) t += L[i * n + k] * L[j * n + k]; err = fmax(err, fabs(t - A[i * n + j])); amax = fmax(amax, fabs(A[i * n + j])); }
  int fail; hipMemcpy(&fail, dfail, 4, hipMemcpyDeviceToHost);
  std::vector<double> Dh(4 * 256); hipMemcpy(Dh.data(), dD, sizeof(double) * 4 * 256, hipMemcpyDeviceToHost);
  unsigned long long hsh = 1469598103934665603ull;
  auto mix = [&](const void* ptr, size_t bytes) { const unsigned char* c = (const unsigned char*)ptr; for (size_t i = 0; i < bytes; ++i) { hsh ^= c[i]; hsh *= 1099511628211ull; } };
  mix(L.data(), sizeof(double) * n * n); mix(Dh.data(), sizeof(double) * 4 * 256);
  printf("bits of L and the block inverses: %016llx\n", hsh);
  printf("diag factor: %.0f cycles, %.2f us per tile (wall clock 100 MHz), effective clock %.2f GHz, |LL^T - A| / |A| = %.2e, fail %d\n",
         (double)h[0] / reps, (double)h[1] / reps * 0.01, (double)h[0] / ((double)h[1] * 10.0), err / amax, fail);
  return 0;
}
