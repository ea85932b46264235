/*
 * cpm_profile.h -- measurement hooks of libcpm_hip.so (not part of the drop-in surface).
 *
 * When enabled, every kernel the library launches is bracketed by a hipEvent pair on the
 * launch stream; cpm_profile_collect() synchronises the device and folds the pairs into
 * per-kernel totals.  bench.py uses this to time the dominant kernel for the roofline
 * figure; rocprofv3 --kernel-trace --stats of the same command must agree.
 * Also here: the Woodcock iteration counter used to report Gsamples/s.
 */
#ifndef CPM_CPM_PROFILE_H
#define CPM_CPM_PROFILE_H
#include "cpm/cpm_ext.h"
#ifdef __cplusplus
extern "C" {
#endif
void cpm_profile_enable(cpm_ctx* ctx, int on);
void cpm_profile_reset(cpm_ctx* ctx);
int cpm_profile_collect(cpm_ctx* ctx);                 /* returns the number of kernels seen */
const char* cpm_profile_name(const cpm_ctx* ctx, int i);
double cpm_profile_total_ms(const cpm_ctx* ctx, int i);
long cpm_profile_calls(const cpm_ctx* ctx, int i);
/* Test / measurement hooks: state of the context they are given (another context -- another GPU, a concurrent frame --
 * is never affected). */
/* when non-NULL, cpm_trace launches add their Woodcock iteration counts to *dev_counter */
void cpm_debug_set_step_counter(cpm_ctx* ctx, unsigned long long* dev_counter);
/* when non-NULL, cpm_gather launches write (start, end [100 MHz ticks], records, XCC id) per 4x4x4 brick */
void cpm_debug_set_gather_stamps(cpm_ctx* ctx, unsigned long long* dev_stamps);
/* test hook: force the voxel-major gather kernel (default: record-major for 1 channel and r < 1.5 voxels) */
void cpm_debug_force_voxel_gather(cpm_ctx* ctx, int on);
/* r < 1 voxel gather: 1 (default) = cooperative kernel (4 waves share 4 bricks) up to 64 Ki bricks, one wave per brick
 * above; 0 = always one wave per brick; 2 / 4 / 8 = always cooperative with that many waves */
void cpm_debug_set_gather_coop(cpm_ctx* ctx, int on);
/* cpm_volume_minmax / cpm_volume_difference: 1 (default) = streaming brick-row kernels, 0 = one wave per brick */
void cpm_debug_set_brick_streaming(cpm_ctx* ctx, int on);
/* cpm_select_changed: 1 (default) = two-launch stable partition, 0 = one radix pass over a 1-bit flag */
void cpm_debug_set_select_partition(cpm_ctx* ctx, int on);
/* cpm_bin: 1 (default) = the last radix pass writes order / records / run starts itself, 0 = separate finalize launch */
void cpm_debug_set_bin_fused(cpm_ctx* ctx, int on);
/* test hook: radix sort pass structure: 0 = hist + rowscan + scatter (default), 1 = onesweep (one launch per pass) */
void cpm_debug_set_sort_mode(cpm_ctx* ctx, int mode);
/* radix tile: 0 = by size, 4 / 8 / 16 = keys per thread (256-thread tiles) */
void cpm_debug_set_sort_items(cpm_ctx* ctx, int items);
/* streaming kernels (temporal mix): workgroups per CU; 0 = one vector per lane, -1 = by size (default) */
void cpm_debug_set_stream_wg_per_cu(cpm_ctx* ctx, int n);
/* test hook: the next cpm_photon_importance_select / _equal_select / _retrace call fails AFTER it has appended its tiles (what a
 * refused launch or a failed allocation does): the selection must then publish a count of 0 */
void cpm_debug_fail_next_select(cpm_ctx* ctx, int on);
/* test hook: a cpm_trace_order's table (n_chunks = ceil(n_light_samples / 256) entries) and the costs gathered since its last
 * update (n_chunks + 1 entries, the last one = launches counted); either may be NULL.  Synchronises the device. */
struct cpm_trace_order;
int cpm_debug_trace_order_read(cpm_ctx* ctx, const struct cpm_trace_order* order, uint32_t* order_out, uint32_t* cost_out);
/* measurement hook: replace the order's table by `table` (n_chunks entries, a permutation of the chunks: checked).  Synchronises. */
int cpm_debug_trace_order_write(cpm_ctx* ctx, struct cpm_trace_order* order, const uint32_t* table);
/* test hook: the sender's launch of cpm_bricklist_pack_grid into a caller-made segment (a communicator of one rank has no segment to
 * pack into): the non-zero 4x4x4 bricks of `grid` (of the marked ones when `nonzero_bricks` is given) -> `segment`. */
int cpm_debug_pack_grid_segment(cpm_ctx* ctx, const cpm_bricklist_segment* segment, const cpm_grid_desc* grid_desc, const float* grid,
                                const uint8_t* nonzero_bricks, cpm_stream stream);
/* test / measurement hook: the ROOT's two launches of cpm_bricklist_reduce_exchange (brick -> slot tables, then the sum in the segments'
 * order) over n caller-made segments on this device (all of one ticket; a segment's `capacity` = the slots "received"; a segment whose
 * count exceeds it adds nothing, as at the root): grid += the segments.  slot_of: (n + 1) * (4x4x4 bricks of the grid) words of scratch,
 * zero-filled before the first call (the last block is the launches' who-lists-what words: zero again when the call's work is done). */
int cpm_debug_root_add_segments(cpm_ctx* ctx, const cpm_bricklist_segment* segments, int n, const cpm_grid_desc* grid_desc, float* grid,
                                uint32_t* slot_of, cpm_stream stream);
#ifdef __cplusplus
}
#endif
#endif
