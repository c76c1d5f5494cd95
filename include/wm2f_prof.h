/*
 * wm2f_prof.h -- additions of the PROFILING build of the library (libwm2f_prof.so = the same sources compiled with
 * -DWM2F_PROFILING; `python -m weed_instance_segmentation_amd._build --prof`).  Used by tools/ only: never by the
 * product path, the tests' parity checks or bench.py.  It exports everything include/wm2f.h declares, plus:
 *
 *   - wm2f_msdeform_fwd_v accepts, beside the production variants 0 / 1 / 2 / 4:
 *     superseded kernels and measured negatives (valid outputs; the baselines DESIGN.md's numbers are quoted against)
 *       3                       phased quad kernel (superseded by the streaming form)
 *       5 / 6 / 7               streaming kernel with per-window flags instead of barriers / tiles in 2-wide strips / the
 *                               round-1 loader schedule
 *       8                       streaming kernel in its half-head form (two 77-KiB workgroups per CU)
 *       62                      LDS-window kernel in slab-major work order
 *     timing ablations, whose OUTPUTS ARE NOT VALID
 *       12 / 22 / 32 / 42 / 52  LDS-window kernel: staging only, gather only, no operand loads, no LDS reads, neither
 *       13 / 23 / 43            phased quad kernel: staging only, gather only, no LDS reads
 *       44                      streaming quad kernel (full-head form) without LDS reads
 *     stamped builds (valid outputs)
 *       73 / 74 / 84            phased / full-head streaming / half-head streaming kernel with in-kernel time stamps
 *   - wm2f_msdeform_fused_lanes_fwd reads WM2F_K1_MODE on every launch: 200 strip order, 300 round-1 loader schedule, 500
 *     Z-order, 600 / 700 static wave priority, 800 slab order, 801 / 802 / 803 slab order with non-temporal operand loads /
 *     output stores / both, 807 slab order stamped (tools/k1_slab_inmodel.py, tools/k1_stamps.py)
 *   - wm2f_msdeform_bwd reads WM2F_K1_BWD_OLD (1: the wave-per-query grad_value kernel); wm2f_msdeform_rows_bwd reads
 *     WM2F_K1_LW_THREADS: 512 / 768 = 8 / 12 waves per workgroup in the row-gradient kernel, 1 / 2 = that kernel without window
 *     staging / staging only (OUTPUTS NOT VALID), 3 = that kernel alone (grad_value not computed)   (tools/probes/k1_rows_bench.py)
 *   - K2 / K3 read their experiment knobs from the environment on every launch
 *       WM2F_K2_QTILES, WM2F_K2_WG_TARGET, WM2F_K2_FULL, WM2F_K2_QSPLIT, WM2F_K3_DBG   (tools/kbench.py)
 *   - the stamp buffer below: a __device__ global, i.e. the global mutable state the production library forbids.
 */
#ifndef WM2F_PROF_H
#define WM2F_PROF_H

#include "wm2f.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Variants 73 / 74 of wm2f_msdeform_fwd_v stamp s_memtime: 73 (phased kernel) wave 0 into slots 0-13 of its workgroup's
 * row; 74 (streaming kernel, second tile of every workgroup) EVERY wave into [wave][slot] -- gather waves slots 0-9,
 * loader waves slots 10-15.  This copies the stamps to HOST memory (int64 [8192 workgroups][160] = [8192][10 waves][16],
 * n_bytes <= 10 MiB).  Synchronous; no reference counterpart. */
int wm2f_debug_stamps(void* host_dst, int64_t n_bytes);

#ifdef __cplusplus
}
#endif
#endif /* WM2F_PROF_H */
