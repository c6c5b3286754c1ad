/*
 * mdx_debug.h — diagnostic entry points of libmdx.so: kernel timing experiments, launch-shape switches and per-kernel
 * time stamps used by the tools under tools/ and by bench.py's roofline figures. None of them has a counterpart in
 * the reference pipeline and none is needed to run a stage; they are kept apart from the run-time ABI of mdx.h.
 */
#ifndef MDX_DEBUG_H
#define MDX_DEBUG_H
#include "mdx.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Average time (ms) of `iters` back-to-back launches of the nonbonded kernel on the current pair list. */
int mdx_debug_time_nonbonded(mdx_system *sys, int energy, int iters, double *avg_ms);
/* Switches: key 2 diagnostic bits (0x400000 analytic nonbonded kernels instead of the tabulated ones, 0x1000000 force kernel that combines the
 * Lennard-Jones parameters per pair instead of reading them from the table by atom type, 512 library transform + stand-alone solve kernel, 0x2000000 y and z
 * transforms as separate launches, 0x4000000 bonded terms and pull always in launches of their own on the bonded stream, 0x8000000 pair search without the previous cluster order), 3 rounds of work units
 * handed out dynamically (0: static deal), 4 log2 of the parts a dynamically handed unit is cut into, 10 streams per
 * handle (1 or 3), 11 pair statistics on the energy steps of mdx_step, 12 fused update on or off, 14 list entries per
 * work unit. */
int mdx_debug_set(mdx_system *sys, int key, int value);
/* Diagnostic build (-DMDX_STAMPS) only: reset, or read, the per-kernel first-start / last-end time stamps. */
int mdx_debug_stamps(mdx_system *sys, int reset, long long *out48);
/* One launch of the force-only nonbonded kernel with per-wave clocks: 10 values per wave (shader clock
 * at start and end, 100 MHz wall clock at start and end, work units, list entries, HW_ID, XCC_ID,
 * 8x8 tiles tested, tiles evaluated). */
int mdx_debug_wave_times(mdx_system *sys, long long *out, int max_waves, int *nwaves);
/* Average time (ms) of `iters` back-to-back runs of one stage on its own stream: 1 pair search,
 * 2 PME chain, 3 bonded + exclusion kernels, 4 update + constraints, 5 prepare, 6 PME spread, 7 PME gather.
 * The state is not meaningful afterwards (diagnostic only). */
int mdx_debug_time_stage(mdx_system *sys, int stage, int iters, double *avg_ms);

#ifdef __cplusplus
}
#endif
#endif
