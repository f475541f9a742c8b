/* orbcalib.h -- C ABI of tools/calib/liborbcalib.so: measurement aids for profiles/ (not part of the product boundary, include/orbhip.h). */
#ifndef ORBCALIB_H
#define ORBCALIB_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORBCAL_E_ARG (-1)
#define ORBCAL_E_HIP (-2)

/* Copies nbytes device->device with 4-byte-per-lane accesses (known HBM traffic: nbytes read + nbytes written) so that the
 * rocprofv3 FETCH_SIZE/WRITE_SIZE counters can be calibrated in the product kernels' access pattern (tools/collect_traffic.py). */
int orbx_calibration_copy(const void *d_src, void *d_dst, size_t nbytes, void *stream);

/* The chip's vector-issue ceiling for one opcode class (orb_calib.h), behind bench.py's `valu_issue` object.  Runs a stream of
 * independent instructions of class `op` (0 <= op < orbx_calibration_valu_ops(); orbx_calibration_valu_name(op) names it) on every CU
 * with `waves_per_simd` (1, 2, 4 or 8) resident wavefronts per SIMD, `trips` x 128 instructions per wavefront.  Out: wave-instructions
 * per second of the whole chip (HIP events), shader cycles one wave-instruction occupies its SIMD (s_memtime, median over workgroups)
 * and the shader clock held meanwhile in GHz (s_memtime / s_memrealtime).  Synchronous, default stream. */
int orbx_calibration_valu_ops(void);
const char *orbx_calibration_valu_name(int op);
int orbx_calibration_valu(int device, int op, int waves_per_simd, int trips, double *wave_instr_per_s, double *cycles_per_instr,
                          double *clock_ghz);

#ifdef __cplusplus
}
#endif
#endif
