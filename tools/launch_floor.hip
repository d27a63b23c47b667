// launch_floor.hip — what does it cost to launch one single-wave workgroup per pair?
//   hipcc -O3 --offload-arch=gfx950 tools/launch_floor.hip -o tools/launch_floor && tools/launch_floor
// The contact kernel runs one wave = one workgroup per pair (579 574 of them at the headline).  Kernels here: empty;
// one that reads a 320-byte record per workgroup (the contact kernel's prologue load); and a PERSISTENT form of the
// second (a grid that just fills the chip, each wave striding over the records).  Dynamic LDS as the contact kernel's.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(64) void k_empty(const double* rec, double* out, int n)
{
  extern __shared__ double sm[];
  if (n < 0) out[0] = sm[threadIdx.x];
}
__global__ __launch_bounds__(64) void k_record(const double* rec, double* out, int n)
{
  extern __shared__ double sm[];
  const int w = blockIdx.x;
  const double v = threadIdx.x < 40 ? rec[(size_t)40 * w + threadIdx.x] : 0.0;
  sm[threadIdx.x] = v;
  __builtin_amdgcn_wave_barrier();
  if (sm[(threadIdx.x + 1) & 63] == 12345.678) out[0] = v;
}
__global__ __launch_bounds__(64) void k_persistent(const double* rec, double* out, int n)
{
  extern __shared__ double sm[];
  for (int w = blockIdx.x; w < n; w += gridDim.x) {
    const double v = threadIdx.x < 40 ? rec[(size_t)40 * w + threadIdx.x] : 0.0;
    sm[threadIdx.x] = v;
    __builtin_amdgcn_wave_barrier();
    if (sm[(threadIdx.x + 1) & 63] == 12345.678) out[0] = v;
    __builtin_amdgcn_wave_barrier();
  }
}

template <typename K>
static float run(K kern, dim3 grid, int lds, const double* rec, double* out, int n)
{
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(kern, grid, dim3(64), lds, 0, rec, out, n);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 10; ++r) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, grid, dim3(64), lds, 0, rec, out, n);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  return best;
}

int main()
{
  const int n = 579574;
  double *rec, *out;
  CHECK(hipMalloc(&rec, (size_t)n * 40 * 8));
  CHECK(hipMemset(rec, 0, (size_t)n * 40 * 8));
  CHECK(hipMalloc(&out, 64));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  printf("# %d single-wave workgroups, %d CUs; best of 10, ms\n", n, cus);
  for (int lds : {0, 5344, 8512, 14112}) {
    const float a = run(k_empty, dim3(n), lds, rec, out, n);
    const float b = run(k_record, dim3(n), lds, rec, out, n);
    const int resident = lds ? (160 * 1024 / lds < 32 ? 160 * 1024 / lds : 32) : 32;
    const float c = run(k_persistent, dim3(cus * resident), lds, rec, out, n);
    printf("lds %5d B: empty %.3f   record load %.3f   persistent (%d waves/CU) %.3f\n", lds, a, b, resident, c);
  }
  return 0;
}
