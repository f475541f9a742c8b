// orbcalib.hip -- measurement aids, NOT part of the product library (liborbhip.so): the known-traffic copy kernel that calibrates the
// rocprofv3 HBM counters in the product kernels' access pattern (tools/collect_traffic.py) and the vector-issue ceiling per opcode
// class (orb_calib.h, tools/collect_valu_calib.py -> profiles/valu_calib.json).  Built in-tree to tools/calib/liborbcalib.so by
// tools/calib/calib.py (hipcc --offload-arch=gfx950); nothing in the product path loads it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "orbcalib.h"
#include "orb_calib.h"

#define ORBX_E_ARG ORBCAL_E_ARG
#define ORBX_E_HIP ORBCAL_E_HIP

extern "C" {

// Known-traffic kernel for calibrating the HBM PMC counters in this library's access pattern (4 B per lane, the
// width k_fast / k_blur / k_resize load with): reads nbytes, writes nbytes.  See tools/collect_traffic.py.
__global__ __launch_bounds__(256) void k_calib_copy_u32(const uint32_t *src, uint32_t *dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

int orbx_calibration_copy(const void *d_src, void *d_dst, size_t nbytes, void *stream) {
  if (!d_src || !d_dst || nbytes < 4) return ORBX_E_ARG;
  hipLaunchKernelGGL(k_calib_copy_u32, dim3(2048), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)d_src, (uint32_t *)d_dst, nbytes / 4);
  return hipGetLastError() == hipSuccess ? 0 : ORBX_E_HIP;
}

// Vector-issue ceiling of one opcode class (orb_calib.h).  Synchronous; uses the device's default stream.
int orbx_calibration_valu_ops(void) { return CAL_NUM_OPS; }
const char *orbx_calibration_valu_name(int op) { return op >= 0 && op < CAL_NUM_OPS ? kCalibOpNames[op] : nullptr; }

int orbx_calibration_valu(int device, int op, int waves_per_simd, int trips, double *wave_instr_per_s, double *cycles_per_instr,
                          double *clock_ghz) {
  if (op < 0 || op >= CAL_NUM_OPS || trips < 1 || trips > (1 << 20)) return ORBX_E_ARG;
  if (waves_per_simd != 1 && waves_per_simd != 2 && waves_per_simd != 4 && waves_per_simd != 8) return ORBX_E_ARG;
  hipDeviceProp_t prop;
  if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&prop, device) != hipSuccess) return ORBX_E_HIP;
  const int cus = prop.multiProcessorCount;
  const int rounds = 4;                                               // every CU stays full until the last round drains
  const int grid = cus * waves_per_simd * rounds;                     // 256 threads = one wavefront per SIMD of a CU
  const size_t lds = (size_t)(160 * 1024 / waves_per_simd);           // => exactly waves_per_simd workgroups resident per CU
  uint32_t *sink = nullptr;
  unsigned long long *stamps = nullptr;
  if (hipMalloc(&sink, 256 * sizeof(uint32_t)) != hipSuccess) return ORBX_E_HIP;
  if (hipMalloc(&stamps, (size_t)grid * 2 * sizeof(unsigned long long)) != hipSuccess) { (void)hipFree(sink); return ORBX_E_HIP; }
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = 0;
  float best = 1e30f;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) rc = ORBX_E_HIP;
  for (int rep = 0; rep < 4 && rc == 0; rep++) {                      // rep 0 warms the code object and the clocks
    (void)hipEventRecord(e0, (hipStream_t)0);
    if (calib_dispatch<0>(op, grid, lds, sink, stamps, trips) != hipSuccess) { rc = ORBX_E_HIP; break; }
    (void)hipEventRecord(e1, (hipStream_t)0);
    if (hipEventSynchronize(e1) != hipSuccess) { rc = ORBX_E_HIP; break; }
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  if (rc == 0) {
    std::vector<unsigned long long> st((size_t)grid * 2);
    if (hipMemcpy(st.data(), stamps, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost) != hipSuccess) rc = ORBX_E_HIP;
    else {
      // median over the workgroups that ran with the CU full (all but the last round)
      std::vector<double> cyc, clk;
      for (int b = 0; b < grid; b++) {
        const double ticks = (double)st[2 * b], ref = (double)st[2 * b + 1];
        if (ref <= 0) continue;
        cyc.push_back(ticks / ((double)trips * CAL_INSTR_PER_TRIP * waves_per_simd));
        clk.push_back(ticks / ref * 0.1);                             // s_memrealtime runs at 100 MHz
      }
      std::sort(cyc.begin(), cyc.end());
      std::sort(clk.begin(), clk.end());
      const double total = (double)grid * 4.0 * trips * CAL_INSTR_PER_TRIP;
      if (wave_instr_per_s) *wave_instr_per_s = total / ((double)best * 1e-3);
      if (cycles_per_instr) *cycles_per_instr = cyc.empty() ? 0.0 : cyc[cyc.size() / 2];
      if (clock_ghz) *clock_ghz = clk.empty() ? 0.0 : clk[clk.size() / 2];
    }
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  (void)hipFree(stamps);
  return rc;
}

}  // extern "C"
