/*
 * cpm_ext.h -- what libcpm_hip.so offers BEYOND the reference's call sites (cpm/cpm.h is the core: one entry point per call site).
 *
 * Same conventions as cpm.h (device pointers, the caller's stream, int status).  In the order below:
 *   photon-record layouts        the N * I records as two planes; per buffer (cpm_records_describe) or as a context's default
 *   tracer                       several lights in one launch, samples evaluated in the tracer, launch order from measured costs
 *   light volume                 splat from an N * I buffer, exact incremental gather, THE FAST FORMULATION (brick bin + fixed-point
 *                                LDS-tile gather: the product's default frame), its gather into a brick-list segment
 *   correlated update            time step in one pass, occupancy bits, the fused update without a host round trip
 *   sequences                    steps streamed from pinned host memory behind the step before (cpm_volume_stream)
 *   multi-GPU                    union-of-bricks reduce, brick lists to the display GPU, point-to-point bytes
 *   OpenGL                       buffer sharing for the light volume and the photon buffer
 */
#ifndef CPM_CPM_EXT_H
#define CPM_CPM_EXT_H
#include "cpm/cpm.h"
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ photon-record layouts */

/* How the N * I photon records lie in their buffer of 8 * N * I floats.
 * INTERLEAVED: the reference's float8 record (ref progressivephotonmapping/cl/photon.cl:49-63): record j at floats [8 j, 8 j + 8) =
 *   (x, y, z, powerR | powerG, powerB, theta, phi).  What every entry point takes unless it says otherwise, and what the
 *   `photons` port carries.
 * PLANAR: plane A = float4[N * I] of (x, y, z, powerR) at the start of the buffer, plane B = float4[N * I] of
 *   (powerG, powerB, theta, phi) behind it (record j = A[j], B[j]; the same bytes, the same records, `id = offset + k N + i` as
 *   in cl/photontracer.cl:166).  The tracer writes the same two 16-byte stores per record either way; what reads only position and
 *   the first power channel -- the brick bin of a one-channel light volume -- then streams 16 bytes per record instead of 32.
 *   Written by cpm_trace / cpm_trace_emitted / cpm_trace_lights with CPM_TRACE_PHOTONS_PLANAR, read by cpm_bin_fast_layout,
 *   converted by cpm_photons_convert -- call by call; or, for a whole context, cpm_set_photon_layout. */
enum { CPM_PHOTONS_INTERLEAVED = 0, CPM_PHOTONS_PLANAR = 1 };
/* The layout of this context's photon-record buffers (default CPM_PHOTONS_INTERLEAVED).  With CPM_PHOTONS_PLANAR every buffer of
 * N * I records -- `photons8` of every entry point, the snapshot of cpm_snapshot_selected_photons, `old_photons8` of
 * cpm_photon_importance_retrace[_lights] and of cpm_splat_delta with old_stride = 0 (all "laid out like photons8") -- is read and
 * written as two planes of N * I float4 (the tracers as if CPM_TRACE_PHOTONS_PLANAR were given, cpm_bin_fast as
 * cpm_bin_fast_layout(.., CPM_PHOTONS_PLANAR, ..)); results are the same bits, records sit at other addresses.  The compact,
 * index-ordered copies keep the float8 record whatever the context says: `aligned8` of cpm_copy_indexed_photons, `old_photons8` of
 * cpm_trace_selected and of cpm_splat_delta with old_stride > 0.  What the reference's host code never sees (it hands the buffers
 * from processor to processor); libcpm_host.so runs its network this way. */
int cpm_set_photon_layout(cpm_ctx* ctx, int layout);
int cpm_get_photon_layout(const cpm_ctx* ctx);
/* PER BUFFER, overriding the context's default: the record buffer that starts at `base` holds n_records (its N * I) records in `layout`.
 * Every entry point handed `base` as a photon-record buffer ("photons8", a snapshot, old_photons8 "laid out like photons8") from then on
 * reads and writes it that way whatever the context's default and whatever count the call itself is given: plane B lies n_records float4
 * behind plane A even when a call bins or splats only the first records of the buffer (with the context default alone the distance is
 * the call's own record count: a call over a prefix of a planar buffer then needs this description).  A tracer flag
 * (CPM_TRACE_PHOTONS_PLANAR) or cpm_bin_fast_layout's argument still win for their call.  At most 16 buffers per context; describing a
 * base again replaces its entry; cpm_records_forget drops it (do so before the memory is reused for something else). */
int cpm_records_describe(cpm_ctx* ctx, const float* base, int layout, size_t n_records);
int cpm_records_forget(cpm_ctx* ctx, const float* base);

/* ------------------------------------------------------------------ tracer: several lights per launch, samples evaluated in the tracer, launch order */

/* cpm_trace for SEVERAL lights in one launch: light l's samples go to photons photon_offset[l] .. + n_light_samples[l] exactly as
 * n_lights cpm_trace calls with those offsets leave them (same streams, same bits) -- one launch instead of one per light, whose
 * fixed part (a workgroup's chain of round trips, the ramp and the tail: about 10 us at any size) is paid once.
 * params->photon_offset / n_light_samples are ignored; at most CPM_MAX_TRACE_LIGHTS lights (more: call per light).
 * With a cpm_trace_order set it must have been created for the launch's chunks: sum over the lights of
 * 256 * ceil(n_light_samples / 256) samples (cpm_trace_lights_order_samples).
 * Replaces the loop over the light-sample inport's lights around PhotonTracerCL::tracePhotons
 * (ref processor/progressivephotontracercl.cpp:543-549). */
#define CPM_MAX_TRACE_LIGHTS 4
typedef struct cpm_light_span {
    const float* light_samples8;  /* device, float8[n_light_samples] */
    const float* isect2;          /* device, float2[n_light_samples] */
    int32_t n_light_samples;
    int32_t photon_offset;        /* first photon of this light in the photon / RNG arrays */
} cpm_light_span;
int cpm_trace_lights_order_samples(const cpm_light_span* lights, int n_lights);
int cpm_trace_lights(cpm_ctx* ctx, const cpm_volume* vol, const cpm_tf* tf, const cpm_tf* tf_scattering,
                     const float aabb[8], const cpm_trace_params* params, const cpm_light_span* lights, int n_lights,
                     uint32_t* rng_state, float* photons8, cpm_stream stream);

/* The same trace with the emission chain evaluated in the tracer's registers instead of read from buffers: thread i
 * takes lattice sample first_sample + i of cpm_uniform_samples_2d(nx, ny), turns it into the light sample of
 * cpm_directional_light_samples / cpm_point_light_samples and into the entry / exit of cpm_light_sample_box_intersection
 * against `aabb` -- the same device functions, bit-identical photons and RNG states -- and does not read 40 of the
 * 48 input bytes per photon (measured at 1 M photons the launch takes the same time either way: the tracer is bound
 * by instruction issue; what this saves is the buffers and their traffic).  For a tracer whose caller owns the light (a frame driver); a caller that is handed
 * LightSamples of unknown origin (PhotonTracerCL's inport, mesh-intersected samples) uses cpm_trace. */
enum { CPM_EMIT_DIRECTIONAL = 0, CPM_EMIT_POINT = 1 };
typedef struct cpm_emitter_desc {
    int32_t kind;                    /* CPM_EMIT_* */
    int32_t nx, ny;                  /* the emission lattice (cpm_uniform_samples_2d) */
    int32_t first_sample;            /* lattice index of light sample 0 of this call (a shard's first photon) */
    float radiance[4];
    float direction_or_position[4];  /* directional: travel direction; point: position */
    float plane_origin[4], tangent_u[4], tangent_v[4];  /* directional light plane (unused for a point light) */
    float plane_area;
} cpm_emitter_desc;
int cpm_trace_emitted(cpm_ctx* ctx, const cpm_volume* vol, const cpm_tf* tf, const cpm_tf* tf_scattering,
                      const float aabb[8], const cpm_trace_params* params, const cpm_emitter_desc* emitter,
                      const uint32_t* recompute_indices, int n_recompute,
                      uint32_t* rng_state, float* photons8, cpm_stream stream);

/* ---- launch order of a trace, from what its chunks cost the last time -------------------------------------------
 * A trace launch is short (tens of microseconds) and its 256-sample workgroups are unequal: the samples whose rays cross
 * a transparent pocket walk ten times longer than their neighbours, and they are neighbours of each other -- whole
 * workgroups.  Dealt out in lattice order, some of those start last and the launch ends with them alone (config 2:
 * 33.5 us; with every XCD's heaviest eighth started first 29.9; lightest first 35.7).  Frames repeat -- the same
 * light, a transfer function or time step that moved a little -- so the last launch's costs predict the next one's:
 * a cpm_trace_order keeps, per chunk, the sum of its waves' longest walks in the last launch it measured, and
 * cpm_trace_order_update turns that into the order of the following launches: of the chunks an XCD works on (whole
 * 4096-sample tiles, as before) the heaviest eighth first, the others in lattice order.
 * Photon i does not depend on the order: records, RNG states and importances are bit for bit those of the default
 * order.
 * Not in the reference (OpenCL enqueues its work-groups in order, ref processor/photontracercl.cpp:192-200). */
typedef struct cpm_trace_order cpm_trace_order;
/* For launches over exactly n_light_samples samples; starts out as the default order. */
int cpm_trace_order_create(cpm_ctx* ctx, int n_light_samples, cpm_trace_order** out);
void cpm_trace_order_destroy(cpm_ctx* ctx, cpm_trace_order* order);
/* The context's following cpm_trace / cpm_trace_emitted launches over all n_light_samples samples (no recompute
 * indices) take their chunks in `order`; with measure != 0 they also record what every chunk cost (one store per wave;
 * still 1 - 1.5 us of a 30 us launch -- a caller measures a launch now and again, not every one).  Launches of another size
 * are refused (CPM_ERR_INVALID_ARGUMENT).  NULL = the default order.  Several lights: one object each, set before each
 * launch. */
int cpm_trace_set_order(cpm_ctx* ctx, cpm_trace_order* order, int measure);
/* New order from the costs of the last measured launch (then cleared); two small launches.  Without costs -- no measured
 * launch since -- the order is left as it is. */
int cpm_trace_order_update(cpm_ctx* ctx, cpm_trace_order* order, cpm_stream stream);


/* ------------------------------------------------------------------ splat from a buffer of N * I records */

/* ... from a buffer of n_records >= total_photons records (N * I of them, of which the first N are splatted): what a planar context
 * (cpm_set_photon_layout) needs to find plane B; cpm_splat takes n_records = total_photons. */
int cpm_splat_records(cpm_ctx* ctx, const float* photons8, int n_records, int total_photons, const cpm_grid_desc* grid,
                      float radius, float relative_irradiance_scale, float* grid_out, cpm_stream stream);


/* ------------------------------------------------------------------ exact incremental gather; the fast formulation (the product's default frame) */

/* Exact incremental update of a light volume (the "delta gather restricted to touched cells"):
 * after n photons have been re-traced, only voxels inside the splat box of an old or a new position of
 * one of them can change.  cpm_mark_touched_bricks sets brick_mask[b] = 1 for every 4x4x4 voxel brick
 * (b = bx + ceil(dx/4) * (by + ceil(dy/4) * bz)) overlapped by the splat boxes of the selected photons --
 * call it once with the previous photon buffer and once with the new one, on a mask the caller has zeroed
 * (ceil(dx/4) * ceil(dy/4) * ceil(dz/4) bytes).  cpm_gather_bricks then recomputes exactly those bricks
 * from the re-binned photons and leaves every other voxel as it is: the result is bit-identical to a full
 * cpm_gather (unchanged photons keep their relative order, so untouched sums are the same additions).
 * The reference updates with two atomic splats instead (-old, +new: cpm_splat_selected), which is
 * cheaper but neither exact nor reproducible
 * (ref processor/photontolightvolumeprocessorcl.cpp:196-298). */
int cpm_mark_touched_bricks(cpm_ctx* ctx, const float* photons8, const uint32_t* indices, int n_indices,
                            int n_photons, int n_interactions, const cpm_grid_desc* grid, float radius,
                            uint8_t* brick_mask, cpm_stream stream);
int cpm_gather_bricks(cpm_ctx* ctx, const float* sorted_pos_power, const uint32_t* cell_start, int n,
                      const cpm_grid_desc* grid, float radius, float relative_irradiance_scale,
                      const uint8_t* brick_mask, float* grid_out, cpm_stream stream);

/* ---- MI355X formulation, tolerance mode: brick bin + LDS-tile gather (S6, G1-G3 restated)
 *
 * Same light volume as cpm_bin + cpm_gather within a stated fp32 tolerance -- per voxel rtol 2e-5 plus atol 1e-5 of
 * the larger of the volume's maximum and one full-weight contribution (max |power| * k * 0.75); the tolerance the
 * reference's own atomic splat is held to; the absolute part matters for voxels reached only from the rim of the
 * kernel, where the weight is a difference of nearly equal numbers in any formulation: the Epanechnikov weight is evaluated as
 * 0.75 * (1 - d^2 / r^2) for d^2 <= r^2 (no sqrt, no division -- ref cl/densityestimationkernel.cl:43-60 takes
 * x = d / r), and per-voxel sums are accumulated as 64-bit fixed-point integers, so they do not depend on any
 * order: the result is bitwise reproducible run to run although nothing is sorted inside a brick.
 * Photons are filed by BRICK (8x8x16 voxels; 16 voxels along the axes where the candidate box has 5 or more voxels once it has more
 * than 4 along some axis, at most 16x16x8; bigger for grids beyond 8 Ki bricks) with an unstable counting sort -- the
 * order of the records inside a brick is unspecified.  A photon whose candidate voxels (the integers within
 * r * textureToIndex + 1e-3 of its index-space coordinate, per axis, clipped to the grid) straddle a brick face is filed
 * under EVERY brick they lie in (at most 8; 0.3 % of the photons are filed twice at BASELINE config 2), so a brick's
 * voxels receive from that brick's records alone and the gather is one launch.  A photon with a WIDE box (4 - 8 candidates
 * along some axis) is filed ONCE, under the brick of its box's low corner: that brick's tile carries a halo, tiles are staged as
 * 64-bit sums and a second launch of cpm_gather_fast adds the tiles that cover a voxel before the one rounding (context scratch:
 * one tile per brick of the grid) -- the same bits.  Sentinel photons and photons that reach no voxel are dropped.
 * The reference adds with CAS float atomics in arrival order (ref cl/photonstolightvolume.cl:15-29,62-75). */

/* u32 entries of the brick table cpm_bin_fast fills on this grid (0 = bad arguments): brick starts, max |power|, radius. */
size_t cpm_fast_table_entries(const cpm_grid_desc* grid, int n);
/* 1 when the pair covers this grid / radius: at most 8 candidate voxels along every axis (floor(2 (r * textureToIndex + 1e-3)) + 1
 * <= 8: radius < 3.5 voxels of THAT axis -- the box follows an anisotropic grid: 6 x 6 x 2 on the workspace's 256 x 256 x 48
 * light volume; up to 3 per axis the record loops are unrolled, wider boxes take run-time loops and tiles with a halo), positive axis-aligned
 * textureToIndex; otherwise use cpm_bin + cpm_gather. */
int cpm_gather_fast_supported(const cpm_grid_desc* grid, float radius);
/* ... on the context's device: also that a brick's LDS tile fits what a workgroup may use THERE (the form above assumes gfx950's
 * 160 KiB; a wide box's tile with its halo needs 33 - 100 KiB).  What a host layer asks before it takes the fast pair. */
int cpm_gather_fast_supported_on(const cpm_ctx* ctx, const cpm_grid_desc* grid, float radius);
/* Records sorted_pos_power must hold for n photons at this radius (every photon in all its bricks: 8 n; n when a
 * candidate box is a single voxel wide, or wide -- filed once); 0 when unsupported. */
size_t cpm_fast_record_capacity(const cpm_grid_desc* grid, int n, float radius);

/* brick_table (device, cpm_fast_table_entries(grid, n) u32): brick starts (brick_table[bricks] = records written), max
 * |power|, the radius.  sorted_pos_power: cpm_fast_record_capacity(grid, n, radius) compact records, float4
 * (x, y, z, powerR) when channels == 1, 2 x float4 (x, y, z, powerR | powerG, powerB, 0, 0) when == 4.
 * radius: the photon radius (texture units) the gather will use -- it decides which bricks a photon is filed under. */
int cpm_bin_fast(cpm_ctx* ctx, const float* photons8, int n, const cpm_grid_desc* grid, float radius, uint32_t* brick_table,
                 float* sorted_pos_power, cpm_stream stream);
/* The same for records in either layout (CPM_PHOTONS_*; n = N * I records, for PLANAR also the distance between the planes):
 * the same table and the same records filed, bit for bit -- the planar form reads 16 instead of 32 bytes per photon when
 * channels == 1. */
int cpm_bin_fast_layout(cpm_ctx* ctx, const float* photons, int layout, int n, const cpm_grid_desc* grid, float radius,
                        uint32_t* brick_table, float* sorted_pos_power, cpm_stream stream);
/* dst (layout dst_layout) = the n records of src (layout src_layout); src != dst.  A pure copy: 64 bytes moved per record. */
int cpm_photons_convert(cpm_ctx* ctx, const float* src, int src_layout, float* dst, int dst_layout, size_t n_records, cpm_stream stream);

/* grid_out[v] = (accumulate ? grid_out[v] : 0) + float(sum over photons of fixed(power * k * w(v, photon))) with
 * k = relative_irradiance_scale / (4 pi) as in cpm_splat.  n, grid, radius: as given to cpm_bin_fast. */
int cpm_gather_fast(cpm_ctx* ctx, const float* sorted_pos_power, const uint32_t* brick_table, int n,
                    const cpm_grid_desc* grid, float radius, float relative_irradiance_scale, int accumulate,
                    float* grid_out, cpm_stream stream);
/* The same launch also says where the volume it wrote is not zero: nonzero_bricks[b] = 1 when the 4x4x4-voxel brick b
 * (b = bx + ceil(dx / 4) * (by + ceil(dy / 4) * bz): cpm_mark_touched_bricks' numbering) holds a non-zero value, else 0 --
 * every byte written.  What cpm_allreduce_grid_sparse(..., CPM_SPARSE_MASK_NONZERO) takes instead of reading the volume again
 * (multi-GPU full frames).  Not with accumulate (the marks would describe this launch's share only). */
int cpm_gather_fast_marked(cpm_ctx* ctx, const float* sorted_pos_power, const uint32_t* brick_table, int n,
                           const cpm_grid_desc* grid, float radius, float relative_irradiance_scale, int accumulate,
                           float* grid_out, uint8_t* nonzero_bricks, cpm_stream stream);


/* ------------------------------------------------------------------ time steps: difference + min/max in one pass; occupancy bits */

/* One time step of a sequence in one pass over the two volumes: mean_abs_diff as cpm_volume_difference(cur, next) and
 * next_minmax2 as cpm_volume_minmax(next) -- the same values; each volume is read once (8 / 16-bit voxels; float volumes take
 * the two separate launches).  What the importance processor's time-varying branch consumes per step
 * (ref importancesamplingcl/processors/minmaxuniformgrid3dimportanceclprocessor.cpp:149-190). */
int cpm_volume_step(cpm_ctx* ctx, const cpm_volume* cur, const cpm_volume* next, int region, float* mean_abs_diff,
                    uint16_t* next_minmax2, cpm_stream stream);

/* The same, and in the same launch the grid's occupancy bits: occupancy[(c >> 5)] bit (c & 31) set where importance[c] is
 * anything but +0.0f; 2 * ceil(n_cells / 64) u32 words (whole 64-cell groups are written).  What the selection's grid walk
 * tests before it loads a cell (cpm_selection_set_occupancy) -- otherwise the selection builds the bits with a launch of its
 * own.  occupancy may be NULL (= cpm_importance_tf). */
int cpm_importance_tf_occupancy(cpm_ctx* ctx, const uint16_t* minmax2, const uint16_t* prev_minmax2,
                                const float* volume_diff, int n_cells, const float* positions_host,
                                const float* colors4_host, int n_points, float* importance,
                                uint32_t* occupancy, cpm_stream stream);


/* ------------------------------------------------------------------ the fused correlated update */

/* ---- the correlated update without a host round trip (BASELINE configs 3 and 5)
 *
 * The reference's importance branch (ref processor/progressivephotontracercl.cpp:298-374) is a chain of small launches
 * around one host wait: detector kernel per light, threshold, reduce, iota, 31-bit sort, a 4-byte read-back the host
 * blocks on (:343-345,374), keys-only sort, tracer, and in the light-volume processor two selected splats and a
 * 32 MiB snapshot copy (ref processor/photontolightvolumeprocessorcl.cpp:196-298,343-352).  On this GPU each small launch
 * costs its latency and the wait drains the queue, so the chain cost more than re-tracing everything.  The entry points
 * below are the same computation with the count kept on the device:
 *
 *   cpm_selection_begin
 *   cpm_photon_importance_select   (per light)  detector kernel + threshold + per-tile count + tile-local index lists
 *   cpm_selection_finish                        tile lists -> ascending index list, count -> device word + host mailbox
 *   cpm_trace_selected             (per light)  -D PHOTON_RECOMPUTATION tracer over the DEVICE count; keeps the records it
 *                                               overwrites (what prevPhotons_ is for) and resets the photons' importance
 *   cpm_splat_delta                             - old + new atomic splat of the re-traced photons in one launch
 *
 * Results are those of cpm_photon_importance + cpm_select_changed + cpm_trace + cpm_reset_importance +
 * cpm_splat_selected(-1, snapshot) + cpm_splat_selected(+1) bit for bit (the splat sums within atomic-order tolerance).
 * It covers the case in which every changed photon is traced in this evaluation (maxIncrementalPhotonsToUpdate = 100 %,
 * the default); a smaller budget needs the ranking by importance and takes cpm_select_recompute with its host decision. */

typedef struct cpm_selection cpm_selection;

/* State of one selection over at most max_photons photons: tile counts, tile-local lists, the count word and its
 * host-visible mailbox (pinned host memory the compaction kernel writes; cpm_selection_count polls it, no stream sync). */
int cpm_selection_create(cpm_ctx* ctx, size_t max_photons, cpm_selection** out);
void cpm_selection_destroy(cpm_ctx* ctx, cpm_selection* sel);
/* Start a new selection (host bookkeeping only; nothing is enqueued). */
int cpm_selection_begin(cpm_ctx* ctx, cpm_selection* sel);

/* cpm_photon_importance for one light, fused with thresholdKernel and the count: importances[photon_offset + i] is
 * updated exactly as by cpm_photon_importance, and the photons left with a key < 0x7fffffff are listed per tile.
 * The importance grid is walked through a one-bit-per-cell occupancy mask staged in LDS (built by a small launch from the
 * grid handed in): cells with importance +0 are not loaded, the sums are the same floats.
 * Replaces photonRecomputationDetectorKernel + thresholdKernel + clogs::Reduce + indexToBufferKernel
 * (ref cl/photonrecomputationdetector.cl:92-157, cl/threshold.cl:33-40, cl/indextobuffer.cl:33-40,
 *  processor/progressivephotontracercl.cpp:298-356). */
int cpm_photon_importance_select(cpm_ctx* ctx, cpm_selection* sel, const float* importance_grid,
                                 const int32_t grid_dims[3], const float cell_size[3],
                                 const float texture_to_index[16], const float* photons8, int photon_offset,
                                 const float* light_samples8, const float* isect2, int n_light_samples,
                                 int max_interactions, int total_photons, int fix_exit_point,
                                 uint32_t* importances, cpm_stream stream);
/* Detector + threshold + tracer of one light in ONE launch: cpm_photon_importance_select, and the photons it selects are
 * re-traced on the spot by the lanes that found them (the device function cpm_trace runs: same RNG streams, same bits), their
 * importance keys reset.  params: as for cpm_trace (photon_offset, n_light_samples, max_interactions, total_photons, material,
 * flags without CPM_TRACE_PROGRESSIVE).  old_photons8 (float8[N * I], laid out like photons8): the records a re-traced photon had
 * before, written at the photon's OWN index -- the part of the reference's prevPhotons_ snapshot that the add-remove update
 * reads; pass it to cpm_splat_delta with old_stride = 0.  cpm_selection_finish then delivers the index list and the count as
 * for cpm_photon_importance_select.  Results: those of cpm_photon_importance_select + cpm_selection_finish +
 * cpm_trace_selected, bit for bit (photons, index list, count; every selected photon's key back at 0x7fffffff).
 * Replaces the importance branch of ProgressivePhotonTracerCL::process for a full budget
 * (ref processor/progressivephotontracercl.cpp:298-374,467-529). */
int cpm_photon_importance_retrace(cpm_ctx* ctx, cpm_selection* sel, const float* importance_grid,
                                  const int32_t grid_dims[3], const float cell_size[3], const float texture_to_index[16],
                                  const cpm_volume* vol, const cpm_tf* tf, const cpm_tf* tf_scattering, const float aabb[8],
                                  const cpm_trace_params* params, const float* light_samples8, const float* isect2,
                                  int fix_exit_point, uint32_t* importances, uint32_t* rng_state, float* photons8,
                                  float* old_photons8, cpm_stream stream);

/* cpm_photon_importance_retrace for SEVERAL lights in one launch (cpm_light_span: a light's buffers, count and photon offset;
 * params->photon_offset / n_light_samples are ignored): the tiles of the lights are appended to the selection one light after
 * the other and results are those of one call per light in that order, bit for bit -- the launch's fixed part is paid once.
 * At most CPM_MAX_TRACE_LIGHTS lights; through a mixed volume (cpm_volume_mix: stale footprint copy) the lights are launched one
 * by one.  Replaces the loop over the lights around the importance branch (ref processor/progressivephotontracercl.cpp:481-527). */
int cpm_photon_importance_retrace_lights(cpm_ctx* ctx, cpm_selection* sel, const float* importance_grid,
                                         const int32_t grid_dims[3], const float cell_size[3], const float texture_to_index[16],
                                         const cpm_volume* vol, const cpm_tf* tf, const cpm_tf* tf_scattering, const float aabb[8],
                                         const cpm_trace_params* params, const cpm_light_span* lights, int n_lights,
                                         int fix_exit_point, uint32_t* importances, uint32_t* rng_state, float* photons8,
                                         float* old_photons8, cpm_stream stream);

/* The equal-importance detector (ref ...detector.cl:160-194) in the same fused form. */
int cpm_photon_importance_equal_select(cpm_ctx* ctx, cpm_selection* sel, int photon_offset, int n_light_samples,
                                       int percentage, int iteration, uint32_t* importances, cpm_stream stream);

/* indices_out[0 .. count) = the selected photons of all lights, ascending (what cpm_select_changed leaves in the first
 * part of its list; the rest of indices_out is not written); count -> cpm_selection_count_device and the mailbox. */
int cpm_selection_finish(cpm_ctx* ctx, cpm_selection* sel, uint32_t* indices_out, cpm_stream stream);
/* The occupancy bits of `importance_grid` as cpm_importance_tf_occupancy left them: the select / retrace launches of this
 * selection over exactly that grid pointer use them instead of making their own (one launch less per light).  The caller
 * keeps bits and grid in step (the bits of the grid's LAST cpm_importance_tf_occupancy); both NULL = off.  Sticks until
 * changed. */
int cpm_selection_set_occupancy(cpm_ctx* ctx, cpm_selection* sel, const float* importance_grid, const uint32_t* occupancy);
const int32_t* cpm_selection_count_device(const cpm_selection* sel);
/* The count of the last cpm_selection_finish on the host: waits for the mailbox write of THAT launch (a poll of pinned
 * memory; later work in the stream keeps running), not for the stream.  Replaces the blocking wait on the reduce's
 * read-back (ref processor/progressivephotontracercl.cpp:343-345,374). */
int cpm_selection_count(cpm_ctx* ctx, cpm_selection* sel, int32_t* n_out);

/* cpm_trace's recompute variant with the number of indices read on the device: thread j < min(*n_indices_dev, max_indices)
 * traces light sample indices[j] - photon_offset (threads whose index falls outside this light's range do nothing, as in
 * cpm_trace).  old_photons8 (nullable): before a record is overwritten it is copied to old_photons8[k * max_indices + j]
 * (interaction k) -- the previous-photon snapshot of exactly the re-traced photons.  reset_importances (nullable):
 * reset_importances[indices[j]] = 0x7fffffff for every traced photon (resetPhotonImportance, ref
 * processor/progressivephotontracercl.cpp:529,607-611).
 * Replaces PhotonTracerCL::tracePhotons with indices (ref photontracercl.cpp:135-174; cl/photontracer.cl:97-106). */
int cpm_trace_selected(cpm_ctx* ctx, const cpm_volume* vol, const cpm_tf* tf, const cpm_tf* tf_scattering,
                       const float aabb[8], const cpm_trace_params* params, const float* light_samples8,
                       const float* isect2, const uint32_t* indices, const int32_t* n_indices_dev, int max_indices,
                       float* old_photons8, uint32_t* reset_importances, uint32_t* rng_state, float* photons8,
                       cpm_stream stream);

/* grid += splat(photons[indices[j] + k n_photons]) - splat(old_photons8[k old_stride + j]), j < *n_indices_dev
 * (old_stride = 0: the old record of photon indices[j] sits at old_photons8[k n_photons + indices[j]], as
 * cpm_photon_importance_retrace leaves it),
 * k < n_interactions, in one launch; a photon whose old and new records are the same bits adds nothing (the reference's
 * two splats cancel for it up to rounding).  Does nothing when *n_indices_dev >= apply_below (> 0): the caller then
 * rebuilds the volume (the incremental-or-full threshold, ref processor/photontolightvolumeprocessorcl.cpp:196,299,
 * evaluated where the count lives).  brick_mask (nullable): the 4x4x4-voxel bricks an old or new splat box overlaps
 * are marked as by cpm_mark_touched_bricks (multi-GPU delta reduce).
 * Replaces splatSelectedPhotonsToLightVolumeKernel x 2 (ref cl/photonstolightvolume.cl:168-202;
 * processor/photontolightvolumeprocessorcl.cpp:268-274). */
int cpm_splat_delta(cpm_ctx* ctx, const float* old_photons8, int old_stride, const float* photons8,
                    const uint32_t* indices, const int32_t* n_indices_dev, int max_indices, int apply_below,
                    const cpm_grid_desc* grid, float radius, float relative_irradiance_scale, int n_photons,
                    int n_interactions, uint8_t* brick_mask, float* grid_out, cpm_stream stream);


/* ------------------------------------------------------------------ sequences streamed from host memory */

/* ---- a sequence whose steps live in HOST memory: the upload inside the step, behind the step before it -------------------
 * The reference's players step a host-side sequence and upload whichever element is not resident (ref uniformgridcl/processors/
 * volumesequenceplayer.cpp:94-124; the difference analysis reads the same host elements, ref dynamicvolumedifferenceanalysis.h:96-151):
 * 16 MiB of PCIe per 256^3 step -- ~0.33 ms against a 0.07 ms correlated update.  A cpm_volume_stream is a ring of n_slots device volumes
 * (linear block + the tracer's footprint copy each) over such a sequence, filled by a copy stream the library owns:
 *   cpm_volume_stream_prefetch(tag, host voxels, consumer)   enqueues on the copy stream: wait for what `consumer` has enqueued so far (the
 *       slot it overwrites -- the one prefetched / handed out longest ago -- was last read there), H2D, footprint re-layout, an event.
 *       Returns at once; a step that is resident or under way is only marked as about to be used.
 *   cpm_volume_stream_acquire(tag, host voxels or NULL, consumer, &volume)   `consumer` waits for that event (uploading first when the
 *       step is absent) and the slot's volume is handed out: valid until n_slots - 1 OTHER steps have been prefetched / acquired after it.
 * A caller that prefetches step t + 1, then acquires step t and enqueues its update (difference against step t - 1, importance, re-trace,
 * delta splat) needs 3 slots and pays max(upload, update) per step instead of their sum.  Host buffers should be pinned
 * (cpm_pinned_alloc): a copy from pageable memory is staged by the runtime and holds the calling thread.  `tag` names a step (its
 * index); the library never reads host memory after the copy it enqueued has run -- keep a buffer alive until its step was acquired. */
typedef struct cpm_volume_stream cpm_volume_stream;
typedef struct cpm_volume_stream_info {
    uint64_t uploads;             /* H2D copies enqueued */
    uint64_t hits;                /* acquires that found their step resident or under way */
    uint64_t uploads_at_acquire;  /* acquires that had to start the upload themselves (not hidden) */
    uint64_t bytes_uploaded;
    uint64_t bytes_per_step;
    uint64_t uploads_timed;       /* finished uploads whose event pair has been read ... */
    double upload_ms_total;       /* ... and their H2D time in all (the copy alone, on the copy stream) */
} cpm_volume_stream_info;
int cpm_pinned_alloc(cpm_ctx* ctx, size_t bytes, void** out);
void cpm_pinned_free(cpm_ctx* ctx, void* p);
int cpm_volume_stream_create(cpm_ctx* ctx, const cpm_volume_desc* desc, int n_slots /* 2 .. 8 */, cpm_volume_stream** out);
void cpm_volume_stream_destroy(cpm_ctx* ctx, cpm_volume_stream* vs);
int cpm_volume_stream_prefetch(cpm_ctx* ctx, cpm_volume_stream* vs, uint64_t tag, const void* host_voxels, cpm_stream consumer);
int cpm_volume_stream_acquire(cpm_ctx* ctx, cpm_volume_stream* vs, uint64_t tag, const void* host_voxels, cpm_stream consumer, cpm_volume** out);
int cpm_volume_stream_stats(cpm_ctx* ctx, cpm_volume_stream* vs, cpm_volume_stream_info* info);


/* ------------------------------------------------------------------ multi-GPU: sparse and brick-list exchanges, point-to-point; OpenGL sharing */

/* Delta path: total[brick] = sum over ranks of partial[brick] for the UNION of the ranks' touched 4x4x4 bricks
 * (brick_mask as filled by cpm_mark_touched_bricks; replaced by the union), everything else of `total` untouched.
 * Two small collectives (mask, packed voxels) instead of the whole grid; dense fall-back beyond a quarter of the
 * bricks.  Synchronises the stream once (4-byte read-back of the union's size, returned in *n_union_out). */
int cpm_allreduce_grid_bricks(cpm_ctx* ctx, cpm_comm* comm, const float* partial, float* total,
                              const cpm_grid_desc* grid, uint8_t* brick_mask, uint32_t* n_union_out, cpm_stream stream);

/* dst[b] |= src[b] (non-zero -> 1) over n bricks: the union of two brick masks.  What a shard that REBUILDS its light volume passes to
 * cpm_allreduce_grid_sparse as its touched bricks -- the bricks it had lit before (old non-zero marks, plus everything its add-remove
 * updates touched since) or lights now -- so that shards may rebuild and update in the same frame and still agree on the sum. */
int cpm_brick_mask_or(cpm_ctx* ctx, uint8_t* dst, const uint8_t* src, size_t n, cpm_stream stream);

/* Full frames: the sum over the ranks of the per-rank light volumes WITHOUT moving the empty space and WITHOUT a host wait.
 * A rank's photons reach a fraction of the grid (config 2: 1 in 8 of the 4x4x4-voxel bricks holds a non-zero voxel), every
 * rank reaches nearly the same bricks (lattice tiles are dealt round-robin), and the dense sum -- 64 MiB per frame at
 * config 4 against ~40 us of per-rank compute -- would be the frame.  One call enqueues, on the caller's stream:
 *   this rank's non-zero bricks -> byte mask (or the caller's `brick_mask`: see below)
 *   -> ncclAllReduce(max) of the mask = the UNION, identical on every rank -> ascending brick list + count
 *   -> pack the union's bricks of `partial` -> ncclAllReduce / ncclReduce (sum) of capacity * 64 * channels floats -> unpack.
 * total[brick] = sum over ranks of partial[brick] for every brick of the union.  Elsewhere: total == partial (in place) is left
 * as it is (zero on every rank, or unchanged: delta path); a separate `total` is zero-filled there unless a TOUCHED mask was
 * given (then its other bricks still hold the previous sum).  root < 0: every rank receives; else only `root` writes `total`.
 *
 * No stream synchronisation, no read-back in the call: the collective's size is `capacity_bricks`, fixed by the host before
 * the launch.  capacity_bricks = 0 (the normal use) takes cpm_sparse_reduce_capacity_for(n_bricks, union count of the
 * call before the previous one): that count reaches the host through a pinned mailbox (a poll, no stream wait) and is the
 * same number on every rank, so all ranks size the collective alike as long as they make the same sequence of calls.
 * Should a union outgrow its capacity (the scene changed), pack / unpack do nothing and cpm_sparse_reduce_complete --
 * which the caller makes before reading `total` -- enqueues the dense sum instead; capacity == n_bricks means dense from the
 * start (a union beyond half of the bricks).  At most 8 tickets may be issued and not completed.  All calls of one
 * cpm_sparse_reduce go to one stream (or streams ordered by events): they share the mask, list and payload buffers.
 * Call site: where PhotonToLightVolumeProcessorCL::process hands the volume on
 * (ref processor/photontolightvolumeprocessorcl.cpp:356-412). */
typedef struct cpm_sparse_reduce cpm_sparse_reduce;
typedef struct cpm_sparse_reduce_info {
    uint64_t ticket;
    uint32_t n_bricks, n_union, capacity;
    int mode;               /* 0 sparse, 1 dense from the start (capacity policy), 2 dense after an overflow */
    uint64_t reduce_bytes;  /* elements handed to the collectives of this ticket, in bytes: mask + payload (+ dense) */
    uint64_t dense_bytes;   /* cells * channels * 4: what cpm_allreduce_grid hands over */
} cpm_sparse_reduce_info;
int cpm_sparse_reduce_create(cpm_ctx* ctx, cpm_comm* comm, const cpm_grid_desc* grid, cpm_sparse_reduce** out);
void cpm_sparse_reduce_destroy(cpm_sparse_reduce* sr);
uint32_t cpm_sparse_reduce_bricks(const cpm_sparse_reduce* sr);
/* Bricks of payload for a grid of n_bricks whose union two calls ago had previous_union bricks (< 0: not known yet):
 * previous_union * 1.25 + 64 rounded up to 64, a quarter of the bricks while unknown; n_bricks (= dense) beyond half of them.
 * A pure function: hosts that mirror the policy (sharding.py) call it. */
uint32_t cpm_sparse_reduce_capacity_for(uint32_t n_bricks, long long previous_union);
/* brick_mask (nullable = the library finds this rank's non-zero bricks with a pass over `partial`): n_bricks bytes, non-zero =
 * the brick takes part.  mask_kind says what it is: CPM_SPARSE_MASK_NONZERO -- every brick of `partial` holding a non-zero
 * voxel is marked (cpm_gather_fast_marked writes exactly that): same result as NULL without the pass;
 * CPM_SPARSE_MASK_TOUCHED -- the bricks an update changed (cpm_mark_touched_bricks / cpm_splat_delta): everything else of a
 * separate `total` keeps its value. */
enum { CPM_SPARSE_MASK_TOUCHED = 0, CPM_SPARSE_MASK_NONZERO = 1 };
int cpm_allreduce_grid_sparse(cpm_ctx* ctx, cpm_sparse_reduce* sr, const float* partial, float* total, const uint8_t* brick_mask,
                              int mask_kind, int root, uint32_t capacity_bricks, uint64_t* ticket_out, cpm_stream stream);
/* Before `total` of `ticket` is read (enqueue-wise: on `stream`, behind the call that issued the ticket): reads the ticket's
 * union count from the mailbox (waits for THAT launch only) and, after an overflow, enqueues the dense sum. */
int cpm_sparse_reduce_complete(cpm_ctx* ctx, cpm_sparse_reduce* sr, uint64_t ticket, cpm_stream stream, cpm_sparse_reduce_info* info_out);

/* Full frames whose shards light (nearly) DISJOINT parts of the volume: the north-star's "single RCCL reduce" to the display GPU,
 * carried as per-rank brick LISTS.  With contiguous photon ranges a rank's photons enter through a slab of the light plane and
 * its non-zero 4x4x4 bricks are its own; the union-of-bricks reduce above would still move the whole union over every link
 * (config 4 at 8 ranks: 13.8 MB per link and frame against ~55 us of compute per rank).  Here every rank other than `root` packs
 * ITS non-zero bricks -- (brick id, 64 * channels values) records behind a 16-byte header carrying their number -- and sends that one
 * segment to the root (ncclSend / ncclRecv: a rank's bytes cross one xGMI link once); the root adds the segments into its own
 * grid in rank order (a brick two ranks report is summed, in that fixed order: the result is reproducible; a rank whose list had
 * outgrown its segment is added last, at cpm_bricklist_reduce_complete).
 *     grid at the root    = sum over the ranks of their grids (in place)
 *     grid at other ranks = unchanged (read only)
 * A segment's size must be known on both sides when the send and the receive are enqueued: capacity = the sender's brick count of
 * the call before the previous one * 1.25 + 64 (rounded up to 64; a quarter of the bricks while unknown, never more than all) --
 * a number the sender has from its own pinned mailbox and the root from the header it received then (no host wait, no
 * collective for it: cpm_bricklist_capacity_for).  A rank whose count outgrows its capacity is not added from that segment; at
 * cpm_bricklist_reduce_complete -- which every rank calls before the grid is read or gathered into again -- that rank and the root
 * alone repeat the exchange at the exact size (both know the count by then).  No rank waits for a rank it does not exchange with.
 * Every rank makes the same sequence of calls; all calls of one cpm_bricklist_reduce go to one stream (or streams ordered by
 * events); at most 4 tickets issued and not completed.  nonzero_bricks (nullable): the marks cpm_gather_fast_marked left for `grid`
 * (else a pass over the grid finds them); every MARKED brick is listed -- a marked brick of zeros travels as zeros, an unmarked one not at
 * all: marks must cover every non-zero brick.  Communicators of size 1: nothing to do.
 * Call site: where PhotonToLightVolumeProcessorCL::process hands the volume on (ref processor/photontolightvolumeprocessorcl.cpp:356-412). */
typedef struct cpm_bricklist_reduce cpm_bricklist_reduce;
typedef struct cpm_bricklist_info {
    uint64_t ticket;
    uint32_t n_bricks;        /* 4x4x4 bricks of the grid */
    uint32_t n_own;           /* this rank's non-zero bricks */
    uint32_t capacity;        /* bricks its segment had room for (0 at the root) */
    int32_t resent;           /* segments exchanged again at exact size: 0 / 1 at a sender, their number at the root */
    uint64_t sent_bytes;      /* bytes this rank handed to ncclSend for the ticket (0 at the root) */
    uint64_t received_bytes;  /* bytes the root posted receives for (0 elsewhere) */
    uint64_t dense_bytes;     /* cells * channels * 4: what cpm_reduce_grid sends per rank */
    uint32_t listed_bricks;   /* root: bricks in all received lists together (their overlap with each other and the root's is summed) */
} cpm_bricklist_info;
int cpm_bricklist_reduce_create(cpm_ctx* ctx, cpm_comm* comm, const cpm_grid_desc* grid, int root, cpm_bricklist_reduce** out);
void cpm_bricklist_reduce_destroy(cpm_bricklist_reduce* br);
uint32_t cpm_bricklist_reduce_bricks(const cpm_bricklist_reduce* br);
/* Bricks a segment has room for when its sender listed previous_count bricks two calls ago (< 0: not known yet). */
uint32_t cpm_bricklist_capacity_for(uint32_t n_bricks, long long previous_count);
/* Bytes of a segment with room for `capacity` bricks: 16 + capacity * (16 + 256 * channels) -- a 16-byte header (count, capacity,
 * ticket, magic), then per brick a 16-byte head (its id) and its 64 * channels values. */
uint64_t cpm_bricklist_segment_bytes(uint32_t capacity, int channels);
int cpm_reduce_grid_bricklists(cpm_ctx* ctx, cpm_bricklist_reduce* br, float* grid, const uint8_t* nonzero_bricks, uint64_t* ticket_out,
                               cpm_stream stream);
int cpm_bricklist_reduce_complete(cpm_ctx* ctx, cpm_bricklist_reduce* br, uint64_t ticket, cpm_stream stream, cpm_bricklist_info* info_out);

/* The same exchange WITHOUT a dense grid on the senders (round 6).  cpm_reduce_grid_bricklists above is, step by step:
 *     cpm_bricklist_reduce_open      a ticket, its capacities (host only: no launch), and -- on a rank other than the root -- where that
 *                                    ticket's segment lies: `segment`
 *     [sender] the segment's slots   cpm_bricklist_pack_grid (from a dense grid: one launch over its bricks), or -- no dense grid, no
 *                                    zeros, no pass over it -- cpm_gather_fast_segment: the brick gather writes the non-zero 4x4x4
 *                                    bricks of every gather brick straight into slots taken from the segment's device counter; the
 *                                    launch's last workgroup writes the header and the count's mailbox word
 *     [root] its own light volume    cpm_gather_fast into `root_grid` as ever
 *     cpm_bricklist_reduce_exchange  sender: ONE ncclSend of the segment's first `capacity` slots; root: the N - 1 receives as one
 *                                    group, then TWO launches whatever N is: every received brick's slot goes into a per-sender
 *                                    brick -> slot table, then the lowest-ranked sender that lists a brick adds it and every higher
 *                                    rank's values for it, in rank order, into root_grid (a brick several ranks list is summed in
 *                                    that fixed order: reproducible)
 *     cpm_bricklist_reduce_complete  as above; a list that outgrew its segment goes again at exact size FROM THE SAME BUFFER (a
 *                                    sender's buffer has room for every brick of the grid: nothing is rebuilt) and is added after the
 *                                    others.
 * ONE PROCESS PER GPU (cpm_comm_create): a sender's ncclSend meets the root's ncclRecv only when both are enqueued, so a single host thread that
 * drives several communicators (cpm_comm_create_all: Inviwo's one process) must not use this exchange -- its sums go through the grouped
 * cpm_allreduce_grids / the union form, whose calls the library groups across the communicators itself.
 * open / fill / exchange of one ticket may go to different streams ordered by events (fill on the frame's stream, exchange on the
 * reduce's); a ticket's segment is written again four tickets later -- by then its exchange and completion have long been enqueued,
 * and the stream that fills it must have waited for the stream they went to (as it does for a dense grid it gathers into again). */
typedef struct cpm_bricklist_segment {
    void* segment;                /* device; NULL at the root (and with one rank): gather into the dense grid there */
    uint32_t capacity;            /* bricks this ticket's exchange carries */
    uint32_t room;                /* bricks the buffer holds: all of the grid's, rounded up to 64 */
    uint32_t ticket;              /* (low 32 bits: what the header carries) */
    uint32_t channels;
    uint32_t* control;            /* device, 2 words, zero between launches: slots handed out, workgroups done */
    unsigned long long* mailbox;  /* device address of the pinned word the count goes to */
} cpm_bricklist_segment;
int cpm_bricklist_reduce_open(cpm_ctx* ctx, cpm_bricklist_reduce* br, uint64_t* ticket_out, cpm_bricklist_segment* segment_out);
int cpm_bricklist_pack_grid(cpm_ctx* ctx, cpm_bricklist_reduce* br, uint64_t ticket, const float* grid, const uint8_t* nonzero_bricks,
                            cpm_stream stream);
int cpm_bricklist_reduce_exchange(cpm_ctx* ctx, cpm_bricklist_reduce* br, uint64_t ticket, float* root_grid, cpm_stream stream);
/* cpm_gather_fast whose output is a segment (fast formulation, any box the brick gather covers -- wide boxes included): what the
 * dense launch would store, as the non-zero 4x4x4 bricks alone.  No grid is read or written.  Replaces, on a shard that is not the
 * display GPU, the light volume PhotonToLightVolumeProcessorCL::process hands on (ref processor/photontolightvolumeprocessorcl.cpp:404-412). */
int cpm_gather_fast_segment(cpm_ctx* ctx, const float* sorted_pos_power, const uint32_t* brick_table, int n, const cpm_grid_desc* grid,
                            float radius, float scale, const cpm_bricklist_segment* segment, cpm_stream stream);
/* A segment's bricks added into a dense grid on the same device (tests, and a shard that wants its own light volume back):
 * grid[brick] += values for the first min(count, room) slots. */
int cpm_bricklist_segment_to_grid(cpm_ctx* ctx, const cpm_bricklist_segment* segment, const cpm_grid_desc* grid, float* grid_out, cpm_stream stream);
/* Point-to-point bytes over the communicator (ncclSend / ncclRecv on the caller's stream): what bench.py times at set-up to put measured
 * latency and link figures beside the exchange model's assumed ones. */
int cpm_comm_send(cpm_ctx* ctx, cpm_comm* comm, const void* buf, size_t bytes, int peer, cpm_stream stream);
int cpm_comm_recv(cpm_ctx* ctx, cpm_comm* comm, void* buf, size_t bytes, int peer, cpm_stream stream);

/* ---- OpenGL sharing: the consumer side of the light volume ---------------------------------------------------------
 * Replaces Inviwo's CL-GL sharing on this path (property `glsharing`, ref processor/progressivephotontracercl.cpp:93,
 * processor/photontolightvolumeprocessorcl.cpp:69): `SyncCLGL` + `BufferCLGL` for the photon buffer (ref
 * photontolightvolumeprocessorcl.cpp:184-194) and `VolumeCLGL` for the light volume the raycaster samples, filled by
 * `enqueueCopyBufferToImage` (ref :404-406).
 * CDNA GPUs have no image hardware (hipMalloc3DArray: "operation not supported" on gfx950), so a GL texture cannot be
 * mapped the way VolumeCLGL maps it; GL BUFFER objects can.  The light volume therefore reaches the raycaster's 3-D
 * texture through a pixel-unpack buffer of the HOST's context: registered once, acquired for the frame's launches --
 * the texels are written into it on the device -- released, and the host issues glTexSubImage3D from it (a copy inside
 * the GL driver).  Never through host memory.  The photon buffer can be shared the same way (BufferCLGL).
 * All calls need the host's OpenGL context current on the calling thread (as Inviwo's processors have it); without one
 * -- a headless process -- registration returns CPM_ERR_UNSUPPORTED and the caller keeps its own upload path.  The
 * library does not link OpenGL.  Never run against a live context in this repository's test environment (no display on
 * the GPU boxes): the device work is tested through cpm_light_volume_texels, the rest by argument / no-context tests. */

typedef struct cpm_gl_resource cpm_gl_resource;
enum { CPM_GL_TEXEL_F32 = 0, CPM_GL_TEXEL_F16 = 1 };

/* The light volume (n = cells * channels floats) as the texels of the four output formats the reference's processor knows
 * (two of them offered in its UI) -- Float32 / Vec4Float32 as they are, Float16 / Vec4Float16 rounded to nearest even
 * (ref photontolightvolumeprocessorcl.cpp:104-120)
 * -- into any device buffer (texels_out may be light_volume for CPM_GL_TEXEL_F32: nothing is done). */
int cpm_light_volume_texels(cpm_ctx* ctx, const float* light_volume, size_t n, int texel /* CPM_GL_TEXEL_* */, void* texels_out,
                            cpm_stream stream);

/* 1 when the calling thread has a current OpenGL context the library can see, else 0. */
int cpm_gl_available(cpm_ctx* ctx);
/* BufferCLGL: a GL buffer object (the photon buffer, or the pixel-unpack buffer behind the light-volume texture). */
int cpm_gl_register_buffer(cpm_ctx* ctx, unsigned gl_buffer, int read_only, cpm_gl_resource** out);
/* SyncCLGL: acquire before the first launch that touches the resources, release after the last (stream-ordered). */
int cpm_gl_acquire(cpm_ctx* ctx, cpm_gl_resource* const* resources, int n, cpm_stream stream);
int cpm_gl_release(cpm_ctx* ctx, cpm_gl_resource* const* resources, int n, cpm_stream stream);
/* Device address (and size in bytes) of an acquired buffer: valid until it is released.  A float32 light volume can be
 * gathered straight into it (pass it as grid_out). */
int cpm_gl_buffer_pointer(cpm_ctx* ctx, cpm_gl_resource* buffer, void** dev_ptr, size_t* bytes);
/* enqueueCopyBufferToImage's device half: cpm_light_volume_texels into the acquired buffer (size checked). */
int cpm_gl_copy_to_buffer(cpm_ctx* ctx, const float* light_volume, size_t n, int texel, cpm_gl_resource* buffer, cpm_stream stream);
void cpm_gl_unregister(cpm_ctx* ctx, cpm_gl_resource* resource);

#ifdef __cplusplus
}
#endif
#endif /* CPM_CPM_EXT_H */
