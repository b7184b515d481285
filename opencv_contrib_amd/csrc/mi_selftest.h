// Hardware self-test hooks: NOT part of the public C-ABI (include/miflow/c_api.h does not declare them).  They run the
// cross-lane primitives the kernels rely on (DPP wave shifts / scans / reductions, ballot tie-break) on known inputs so
// that tests/ can check their semantics on the real device.  Exported under their own prefix for the ctypes tests only.
#pragma once
#include "miflow/c_api.h"

#ifdef __cplusplus
extern "C" {
#endif
/* out_host[0..63] = wave-wide min of in_host[0..63] as seen by every lane, out_host[64] = lane picked by the reference's
 * tie-break rule among the minima (StereoBM winner-take-all) */
MI_API int miflow_selftest_wave_min(const unsigned *in_host, unsigned *out_host /*[65]*/);
/* in_host[k*64 + lane] = value k of lane `lane` (k < 16); out_host[lane] = max over all lanes of value (lane & 15) */
MI_API int miflow_selftest_tmax16(const unsigned *in_host /*[1024]*/, unsigned *out_host /*[64]*/);
/* out_host[i] = inclusive prefix sum of in_host[0..i] over the 64 lanes (DPP scan of the SURF integral) */
MI_API int miflow_selftest_wave_scan(const unsigned *in_host, unsigned *out_host /*[64]*/);
/* out_host[0..63] = value received from lane n-1, out_host[64..127] = from lane n+1 when every lane n contributes n+100
 * (DPP wave shifts of the blocked TV-L1 kernels) */
MI_API int miflow_selftest_lane_shift(int *out_host /*[128]*/);
/* Control slots of pair `pair` of the last convergence-checked calc: per launch 8 ints {scale, warp, S.x, S.y, X.x, X.y, X.z,
 * X.w} (tvl1_dev.h); returns the launch count or a negative status.  Used by tools/spec_trace.py to inspect the decisions of the
 * speculative steps; synchronises `stream`. */
/* *fault = 1 if a wave of a joined-wave blocked iteration kernel ever gave up waiting for its neighbour (never expected: the
 * results of that launch are invalid); synchronises the device */
MI_API int miflow_selftest_jw_fault(int *fault);
/* the RCCL binding of mi_tvl1_multi on one device: out_host = in_host after a grouped ncclSend / ncclRecv to self; *available = 0
 * (and nothing copied) where librccl is absent */
MI_API int miflow_selftest_rccl_self_copy(const unsigned char *in_host, unsigned char *out_host, size_t bytes, int *available);
/* fills the handle's whole scratch arena (every plane of every pair slot it was sized for) with NaNs on `stream`: a calc that reads a
 * plane it has not written first -- e.g. the coarsest level's flow, which is never cleared (zero_flow, csrc/fb_plan.h) -- then shows it.
 * MI_ERR_BAD_ARG before the handle's first calc (no arena yet). */
struct mi_farneback;
MI_API int miflow_selftest_farneback_poison(struct mi_farneback *h, void *stream);
struct mi_tvl1;
MI_API int miflow_selftest_tvl1_slots(struct mi_tvl1 *h, int pair, int *out_host, int cap_launches, void *stream);
#ifdef __cplusplus
}
#endif
