// tools/fp64_peak.hip — the FP64 ceilings of the box as a standalone program:
//   hipcc -O3 --offload-arch=gfx950 tools/fp64_peak.hip -o tools/fp64_peak && tools/fp64_peak
// The same loops are in libshpair.so behind shpair_fp64_peak() (include/shpair.h), which bench.py calls so that
// every bench line carries the peak measured on the box it ran on.  Prints one JSON line.
#include <cstdio>

#include "../lammps-spherharm_amd/csrc/fp64_peak.hpp"

int main()
{
  if (hipSetDevice(0) != hipSuccess) {
    fprintf(stderr, "fp64_peak: no HIP device\n");
    return 2;
  }
  shp::Fp64PeakResult r[3];
  for (int mode = 0; mode < 3; ++mode) {
    const hipError_t e = shp::fp64_peak_run(mode, 50.0, 5, &r[mode]);
    if (e != hipSuccess) {
      fprintf(stderr, "fp64_peak: mode %d failed: %s\n", mode, hipGetErrorString(e));
      return 1;
    }
  }
  const double spec = (double)r[0].cus * 4 * 16 * 2 * r[0].clock_mhz * 1e6 / 1e12;  // CUs x SIMDs x 16 lanes/clk x 2 FLOP
  printf("{\"cus\": %d, \"clock_mhz\": %.0f, \"spec_valu_f64_tflops\": %.1f, \"valu_f64_tflops\": %.2f, "
         "\"mfma_f64_tflops\": %.2f, \"side_by_side\": {\"valu_tflops\": %.2f, \"mfma_tflops\": %.2f, \"sum\": %.2f}, "
         "\"ms\": [%.2f, %.2f, %.2f]}\n",
         r[0].cus, r[0].clock_mhz, spec, r[0].valu_tflops, r[1].mfma_tflops, r[2].valu_tflops, r[2].mfma_tflops,
         r[2].valu_tflops + r[2].mfma_tflops, r[0].ms, r[1].ms, r[2].ms);
  return 0;
}
