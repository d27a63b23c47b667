// mfma_mix.hip — does v_mfma_f64_16x16x4_f64 buy issue slots on gfx950?
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_mix.hip -o tools/mfma_mix && tools/mfma_mix
// The contact kernel is bound on VALU ISSUE: 40 % of its vector instructions are not FP64 and each costs ~3.8 cycles
// beside FP64 (tools/valu_mix.hip).  One v_mfma_f64_16x16x4_f64 carries the FLOP of 16 v_fma_f64 in one issue slot; the
// FP64 datapath is shared (profiles/r02_a_fp64_peak.json: MFMA and v_fma_f64 side by side add up to the single-pipe
// rate).  What this measures: while an FP64 MFMA is in flight, can OTHER instructions issue on the same SIMD —
//   (a) 32-bit integer / move / compare work of other waves,  (b) v_fma_f64 of other waves (control: expected not),
//   (c) independent 32-bit work of the SAME wave behind its own MFMA?
// One workgroup of 1024 threads per CU (the LDS request keeps a second one out): 16 waves, four per SIMD (wave w runs on
// SIMD w % 4; checked with HW_REG_HW_ID and printed).  Waves 0-7 (two per SIMD) take role A, waves 8-15 role B; a role
// is a loop over 8 independent chains of one instruction class, or idle.  Every wave times itself with s_memrealtime.
// If A and B overlap, the pair takes max(T_A, T_B); if they share the issue port / datapath, T_A + T_B.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

enum Role { IDLE = 0, MFMA = 1, FMA64 = 2, ADD32 = 3, MOV32 = 4, CMP64 = 5, MFMA_PLUS_ADD = 6, FMA64_PLUS_ADD = 7, CNDMASK = 8, LDSREAD = 9 };

template <int ROLE>
__device__ __forceinline__ void work(const int iter, double* sink, const double* lds)
{
  d4 acc[8];
  d2 ld[8];
  double a[8];
  unsigned u[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    acc[c] = d4{0.0, 0.0, 0.0, 0.0};
    ld[c] = d2{0.0, 0.0};
    a[c] = 1.0 + c + threadIdx.x * 1e-3;
    u[c] = threadIdx.x + c;
  }
  const double m = 0.999999, b = 1e-9;
  for (int it = 0; it < iter; ++it) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (ROLE == MFMA || ROLE == MFMA_PLUS_ADD)
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[c]) : "v"(m), "v"(b));
      if (ROLE == FMA64 || ROLE == FMA64_PLUS_ADD) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[c]) : "v"(m), "v"(b));
      if (ROLE == ADD32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[c]) : "v"(u[(c + 1) & 7]));
      if (ROLE == MOV32) asm volatile("v_mov_b32 %0, %1" : "=v"(u[c]) : "v"(u[(c + 1) & 7]));
      if (ROLE == CMP64) asm volatile("v_cmp_gt_f64 vcc, %0, %1" : : "v"(a[c]), "v"(m) : "vcc");
      if (ROLE == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[c]) : "v"(u[(c + 1) & 7]));
      if (ROLE == LDSREAD) asm volatile("ds_read_b128 %0, %1" : "=v"(ld[c]) : "v"((unsigned)(threadIdx.x * 16u + c * 16384u) & 0xffffu));
      if (ROLE == MFMA_PLUS_ADD || ROLE == FMA64_PLUS_ADD) {
        // the same wave: independent 32-bit work behind each FP64 instruction (MFMA: 8 of them, about its issue shadow)
#pragma unroll
        for (int r = 0; r < (ROLE == MFMA_PLUS_ADD ? 8 : 1); ++r)
          asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[(c + r) & 7]) : "v"(u[(c + r + 1) & 7]));
      }
    }
    if (ROLE == LDSREAD) asm volatile("s_waitcnt lgkmcnt(0)");
  }
  double s = 0.0;
#pragma unroll
  for (int c = 0; c < 8; ++c) s += a[c] + (double)u[c] + ld[c][0] + ld[c][1] + acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  if (s == 12345.678) sink[0] = s;
}

struct WaveRec {
  unsigned long long cycles;   // s_memrealtime ticks (100 MHz) of the wave's loop
  unsigned hwid;
  unsigned role;
};

template <int A, int B>
__global__ __launch_bounds__(1024) void mix_kernel(double* sink, WaveRec* rec, const int iter_a, const int iter_b)
{
  extern __shared__ double lds[];
  const int wave = threadIdx.x >> 6;
  const bool first = wave < 8;
  unsigned hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  const unsigned long long r0 = wall_clock64();
  if (first) work<A>(iter_a, sink, lds);
  else work<B>(iter_b, sink, lds);
  const unsigned long long r1 = wall_clock64();
  (void)t0;
  if ((threadIdx.x & 63) == 0) {
    WaveRec& w = rec[blockIdx.x * 16 + wave];
    w.cycles = r1 - r0;
    w.hwid = hwid;
    w.role = first ? A : B;
  }
}

template <int A, int B>
static void run(const char* name, double* sink, WaveRec* drec, const int cus, const int iter_a, const int iter_b, const bool show_simd = false)
{
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int lds = 96 * 1024;   // one workgroup per CU
  CHECK(hipFuncSetAttribute((const void*)mix_kernel<A, B>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  float best = 1e30f;
  std::vector<WaveRec> rec(cus * 16);
  for (int r = 0; r < 3; ++r) {
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((mix_kernel<A, B>), dim3(cus), dim3(1024), lds, 0, sink, drec, iter_a, iter_b);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CHECK(hipMemcpy(rec.data(), drec, sizeof(WaveRec) * rec.size(), hipMemcpyDeviceToHost));
  // median loop time of the waves of each role, in microseconds (s_memrealtime runs at 100 MHz)
  std::vector<double> ta, tb;
  for (auto& w : rec) ((w.role == (unsigned)A && (&w - rec.data()) % 16 < 8) ? ta : tb).push_back(w.cycles / 100.0);
  std::sort(ta.begin(), ta.end());
  std::sort(tb.begin(), tb.end());
  printf("%-44s kernel %8.3f ms   role A waves %9.1f us   role B waves %9.1f us\n", name, best, ta[ta.size() / 2], tb[tb.size() / 2]);
  if (show_simd) {
    printf("  SIMD of waves 0..15 of workgroup 0 (HW_ID bits 5:4):");
    for (int w = 0; w < 16; ++w) printf(" %u", (rec[w].hwid >> 4) & 3u);
    printf("\n");
  }
  fflush(stdout);
}

int main()
{
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  double* sink;
  WaveRec* drec;
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMalloc(&drec, sizeof(WaveRec) * cus * 16));
  printf("# %s, %d CUs; one 16-wave workgroup per CU; waves 0-7 role A, waves 8-15 role B (two of each per SIMD)\n", prop.name, cus);
  // equal nominal FP64-pipe time: one MFMA = 16 v_fma_f64 = 64 cycles; a 32-bit instruction ~4 cycles of issue
  const int NM = 1 << 11;            // MFMA rounds (8 MFMAs each)
  const int NF = NM * 16;            // v_fma_f64 rounds of the same FLOP
  const int NI = NM * 16;            // 32-bit rounds of the same nominal issue time
  for (int r = 0; r < 3; ++r) run<FMA64, FMA64>("warm-up", sink, drec, cus, NF, NF, r == 2);
  run<MFMA, IDLE>("A = MFMA alone", sink, drec, cus, NM, 0);
  run<FMA64, IDLE>("A = v_fma_f64 alone (same FLOP)", sink, drec, cus, NF, 0);
  run<ADD32, IDLE>("A = v_add_u32 alone", sink, drec, cus, NI, 0);
  run<MFMA, MFMA>("A = MFMA, B = MFMA", sink, drec, cus, NM, NM);
  run<FMA64, FMA64>("A = v_fma_f64, B = v_fma_f64", sink, drec, cus, NF, NF);
  run<MFMA, FMA64>("A = MFMA, B = v_fma_f64 (control)", sink, drec, cus, NM, NF);
  run<MFMA, ADD32>("A = MFMA, B = v_add_u32", sink, drec, cus, NM, NI);
  run<FMA64, ADD32>("A = v_fma_f64, B = v_add_u32 (control)", sink, drec, cus, NF, NI);
  run<MFMA, MOV32>("A = MFMA, B = v_mov_b32", sink, drec, cus, NM, NI);
  run<FMA64, MOV32>("A = v_fma_f64, B = v_mov_b32 (control)", sink, drec, cus, NF, NI);
  run<MFMA, CMP64>("A = MFMA, B = v_cmp_gt_f64", sink, drec, cus, NM, NI);
  run<FMA64, CMP64>("A = v_fma_f64, B = v_cmp_gt_f64 (control)", sink, drec, cus, NF, NI);
  run<MFMA, CNDMASK>("A = MFMA, B = v_cndmask_b32", sink, drec, cus, NM, NI);
  run<MFMA, LDSREAD>("A = MFMA, B = ds_read_b128", sink, drec, cus, NM, NI / 2);
  run<FMA64, LDSREAD>("A = v_fma_f64, B = ds_read_b128 (control)", sink, drec, cus, NF, NI / 2);
  run<MFMA_PLUS_ADD, IDLE>("A = MFMA + 8 v_add_u32 in the same wave", sink, drec, cus, NM, 0);
  run<FMA64_PLUS_ADD, IDLE>("A = v_fma_f64 + 1 v_add_u32, same wave", sink, drec, cus, NF, 0);
  run<MFMA_PLUS_ADD, MFMA_PLUS_ADD>("A = B = MFMA + 8 v_add_u32 in the same wave", sink, drec, cus, NM, NM);
  printf("# FLOP of a role-A run: %.3e (MFMA: 8 x 2048 per round and wave; v_fma_f64: 8 x 128)\n", (double)NM * 8 * 2048 * 8 * cus);
  CHECK(hipFree(sink));
  CHECK(hipFree(drec));
  return 0;
}
