// valu_mix.hip — what does a 32-bit vector instruction cost beside FP64 on one gfx950 SIMD?
//   hipcc -O3 --offload-arch=gfx950 tools/valu_mix.hip -o tools/valu_mix && tools/valu_mix
// Every wave runs ITER rounds of 8 independent v_fma_f64 chains; mode m adds, per v_fma_f64, one more independent
// vector instruction of another class.  Time per round against mode 0 (FP64 alone, 4 cycles per wave instruction at
// the FP64 peak) gives that instruction's cost in issue cycles at W waves per SIMD.  Used once to read the
// SQ_ACTIVE_INST_VALU counter of the contact kernel (it counts quad-cycles: a 2-cycle instruction still counts one).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITER = 1 << 15;
constexpr int ROUNDS = 8;   // workgroups per resident slot: the dispatcher evens the CUs out

template <int MODE>
__global__ __launch_bounds__(256) void mix_kernel(double* out, const double seed, const int iter)
{
  double a[8];
  unsigned u[8];
  float f[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    a[c] = seed + c + threadIdx.x;
    u[c] = threadIdx.x + c;
    f[c] = (float)(threadIdx.x + c);
  }
  const double m = 0.999999, b = 1e-9;
  for (int it = 0; it < iter; ++it) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(b));
      if (MODE == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) & 7]));
      if (MODE == 2) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[c]) : "v"(f[(c + 1) & 7]));
      if (MODE == 3) asm volatile("v_mov_b32 %0, %1" : "=v"(u[c]) : "v"(u[(c + 1) & 7]));
      if (MODE == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[(c + 1) & 7]));
      if (MODE == 5) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) & 7]));
      if (MODE == 6) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[c]));
      if (MODE == 7) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a[c]), "v"(m) : "vcc");
      if (MODE == 8) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) & 7]));
      if (MODE == 9) asm volatile("v_mov_b64 %0, %1" : "=v"(a[(c + 1) & 7]) : "v"(a[c]));
      if (MODE == 10) asm volatile("s_add_u32 s20, s20, 1\n s_and_b32 s21, s21, s20" : : : "s20", "s21", "scc");
      if (MODE == 11) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[c]) : "v"(u[(c + 1) & 7]));
    }
  }
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += a[c] + (double)u[c] + (double)f[c];
  if (s == 12345.678) out[0] = s;
}

// FP64 alone is mode 0; mode 12 = 32-bit adds alone (two per slot)
template <>
__global__ __launch_bounds__(256) void mix_kernel<12>(double* out, const double seed, const int iter)
{
  unsigned u[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) u[c] = threadIdx.x + c;
  for (int it = 0; it < iter; ++it) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) & 7]));
      asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 3) & 7]));
    }
  }
  unsigned s = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += u[c];
  if (s == 0x12345678u && seed == 17.0) out[0] = s;
}

template <int MODE>
static double run(double* out, const int waves_per_simd, const int cus)
{
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const dim3 grid(cus * waves_per_simd * ROUNDS), block(256);
  // residency is capped by LDS: W workgroups of four waves fit on a CU, one wave per SIMD each
  const int lds = (160 * 1024 / waves_per_simd) & ~1023;
  CHECK(hipFuncSetAttribute((const void*)mix_kernel<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipLaunchKernelGGL(mix_kernel<MODE>, grid, block, lds, 0, out, 1.0, ITER);
  CHECK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < 2; ++r) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(mix_kernel<MODE>, grid, block, lds, 0, out, 1.0, ITER);
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
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  double* out;
  CHECK(hipMalloc(&out, 64));
  for (int r = 0; r < 4; ++r) run<0>(out, 4, cus);   // warm-up: about a second of FP64 on every SIMD
  const char* names[] = {"v_fma_f64 alone", "+ v_add_u32", "+ v_fma_f32", "+ v_mov_b32", "+ v_cndmask_b32", "+ v_mul_lo_u32",
                         "+ v_rcp_f64", "+ v_cmp_gt_f64", "+ v_mul_u32_u24", "+ v_mov_b64", "+ 2 SALU", "+ v_lshl_add_u32",
                         "2 v_add_u32 alone"};
  printf("# %s, %d CUs; per slot = one v_fma_f64 (+ one instruction of the class); cycles at 2.4 GHz per slot and wave\n", prop.name, cus);
  printf("%-20s %10s %10s %10s %10s\n", "mode", "W=1", "W=2", "W=4", "W=5");
  for (int m = 0; m <= 12; ++m) {
    printf("%-20s", names[m]);
    for (int w : {1, 2, 4, 5}) {
      double ms = 0;
      switch (m) {
        case 0: ms = run<0>(out, w, cus); break;
        case 1: ms = run<1>(out, w, cus); break;
        case 2: ms = run<2>(out, w, cus); break;
        case 3: ms = run<3>(out, w, cus); break;
        case 4: ms = run<4>(out, w, cus); break;
        case 5: ms = run<5>(out, w, cus); break;
        case 6: ms = run<6>(out, w, cus); break;
        case 7: ms = run<7>(out, w, cus); break;
        case 8: ms = run<8>(out, w, cus); break;
        case 9: ms = run<9>(out, w, cus); break;
        case 10: ms = run<10>(out, w, cus); break;
        case 11: ms = run<11>(out, w, cus); break;
        case 12: ms = run<12>(out, w, cus); break;
      }
      // SIMD cycles per slot: time x clock / (slots per wave x waves per SIMD)
      const double cyc = ms * 1e-3 * 2.4e9 / ((double)ITER * 8 * w * ROUNDS);
      printf(" %10.2f", cyc);
    }
    printf("\n");
    fflush(stdout);
  }
  CHECK(hipFree(out));
  return 0;
}
