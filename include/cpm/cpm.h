/*
 * cpm.h -- C-ABI of libcpm_hip.so: the MI355X (gfx950) photon-mapping hot path.
 *
 * One data-parallel path, three stages:  trace -> sort/bin -> gather
 * (plus the correlated re-trace helpers that feed it), written as HIP kernels
 * for CDNA4 and exposed with plain pointers and sizes only.  Every entry point
 * names the reference interface it replaces ("ref:" = path under
 * /root/reference/modules, file:line).
 *
 * Conventions
 *   - All buffer arguments are DEVICE pointers unless the name ends in _host.
 *   - All work is enqueued on the caller's stream (a hipStream_t passed as
 *     void*; NULL = the null stream).  Nothing here synchronises the host
 *     unless the doc says so.
 *   - Return value: 0 = CPM_OK, negative = cpm_status.  The library never
 *     throws across the ABI (the reference catches cl::Error at every call
 *     site and logs it: ref progressivephotonmapping/photontracercl.cpp:79,128-130).
 *   - A context is bound to one device and is not thread-safe (Inviwo
 *     evaluates process() on one thread: ref processor/progressivephotontracercl.cpp:219).
 *   - Scratch memory is owned by the context, grows on demand and is reused;
 *     no allocation happens in steady state.  Calls on ONE context must
 *     therefore be ordered with respect to each other (one stream, or streams
 *     the caller chains with events): for frames in flight on several
 *     streams use one context per stream (bench.py's `pipelined` figure does).
 *   - float8 photons / light samples are 8 consecutive floats, 32-byte
 *     stride: (x, y, z, powerR, powerG, powerB, theta, phi)
 *     (ref progressivephotonmapping/photondata.h:47-56, cl/photon.cl:35,
 *      lightcl/cl/datastructures/lightsample.cl:75-101).
 *   - float8 buffers, the compact (pos, power) records of cpm_bin and 4-channel
 *     grids must be 16-byte aligned, (tStart, tEnd) pairs and RNG states
 *     8-byte aligned (every allocator block is); misaligned pointers are
 *     refused with CPM_ERR_INVALID_ARGUMENT.  u32 index / table buffers need
 *     only their natural 4 bytes.
 *   - 4x4 matrices are 16 floats, column-major (OpenCL float16 as uploaded by
 *     Inviwo: element [4*col + row]).
 *
 * THIS header is the core: one entry point per call site of the reference's host code on the path (create / destroy / error,
 * seed, volume, TF, emission, trace, splat +/-, sort, bin, gather, min/max, difference, importance, select, mix, communicator +
 * reduce) -- what a maintainer binding libcpm_hip.so into Inviwo needs to replace the OpenCL calls one for one (INTEGRATION.md,
 * call-site table).  Everything this build adds beyond those call sites -- the fast formulation that is the product's default
 * frame, photon-record layouts, several lights per launch, launch orders, the fused correlated update, streamed sequences, the
 * sparse / brick-list exchanges, OpenGL sharing -- is declared in cpm/cpm_ext.h (which includes this header); measurement hooks
 * in cpm/cpm_profile.h.  One library, one ABI version.
 */
#ifndef CPM_CPM_H
#define CPM_CPM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CPM_ABI_VERSION 2   /* 2: cpm.h (the reference's call sites) + cpm_ext.h; brick-list segments carry 16-byte slot heads */

typedef enum cpm_status {
    CPM_OK = 0,
    CPM_ERR_INVALID_ARGUMENT = -1,
    CPM_ERR_OUT_OF_MEMORY = -2,
    CPM_ERR_DEVICE = -3,       /* a HIP call failed; see cpm_last_error_string */
    CPM_ERR_UNSUPPORTED = -4,  /* e.g. a non axis-aligned textureToIndex matrix */
    CPM_ERR_NO_DEVICE = -5
} cpm_status;

typedef struct cpm_ctx cpm_ctx;
typedef struct cpm_volume cpm_volume;
typedef struct cpm_tf cpm_tf;
typedef void* cpm_stream; /* hipStream_t */

/* ------------------------------------------------------------------ context */

/* Create a context on HIP device `device`.  Fails with CPM_ERR_NO_DEVICE when
 * no GPU is visible: there is no CPU fallback in this library. */
int cpm_create(int device, cpm_ctx** out);
void cpm_destroy(cpm_ctx* ctx);
/* Last error text of this context ("" when none).  With ctx == NULL: the last
 * error of a failed cpm_create on this thread. */
const char* cpm_last_error_string(const cpm_ctx* ctx);
int cpm_abi_version(void);

/* ------------------------------------------------------------------ RNG (R1, R2) */

/* Host helper.  Fills bases_host[i] with glibc's rand() sequence after
 * srand(seed): the per-stream base offsets the reference draws on the host
 * (ref rndgenmwc64x/mwc64xseedgenerator.cpp:56-64).  Implemented as glibc's
 * TYPE_3 additive-feedback generator so results do not depend on the libc the
 * caller links. */
void cpm_glibc_rand_sequence(uint32_t seed, uint32_t* bases_host, size_t n);

/* MWC64X stream seeding: state[2i..2i+1] = split(BASEID * A^(state[2i] + i*gap) mod M).
 * In: state[2i] holds the base offset of stream i (state[2i+1] ignored).
 * Out: (x, c) per stream.  Replaces kernel MWC64X_GenerateRandomState
 * (ref rndgenmwc64x/cl/randstategen.cl:39-47, gap = 2^40) and
 * MWC64X_GeneratePerStreamRandomState (:52-60, gap = maxSamplesPerStream). */
int cpm_seed_streams(cpm_ctx* ctx, uint32_t* state, size_t n, uint64_t gap, cpm_stream stream);

/* out[i + k*n] = random_01(stream i), k = 0..draws-1; state is advanced and
 * written back.  Replaces randomNumberGeneratorKernel
 * (ref rndgenmwc64x/cl/randomnumbergenerator.cl:34-50); known-answer harness. */
int cpm_random_fill(cpm_ctx* ctx, uint32_t* state, size_t n, int draws, float* out, cpm_stream stream);

/* ------------------------------------------------------------------ volume / TF */

typedef enum cpm_dtype { CPM_U8 = 0, CPM_U16 = 1, CPM_F32 = 2 } cpm_dtype;

/* Mirrors the fields of Inviwo's VolumeParameters the path reads
 * (ref use sites progressivephotonmapping/cl/photontracer.cl:75,
 *  cl/photonstolightvolume.cl:37,46-47,57).  textureToIndex must be
 *  diag(dim) + translate(-0.5) (Inviwo's StructuredCoordinateTransformer),
 *  indexToTexture its inverse; anything else -> CPM_ERR_UNSUPPORTED. */
typedef struct cpm_volume_desc {
    int32_t dims[3];
    int32_t dtype;          /* cpm_dtype */
    float format_scaling;   /* VolumeParameters::formatScaling (0 for 8/16-bit UNORM, float) */
    float format_offset;    /* VolumeParameters::formatOffset */
    float texture_to_index[16];
    float index_to_texture[16];
} cpm_volume_desc;

/* Fill desc for a dims[0..2] volume of `dtype` with Inviwo's default matrices. */
void cpm_volume_desc_default(cpm_volume_desc* desc, const int32_t dims[3], int32_t dtype);

/* Upload (or adopt a device copy of) a scalar volume, x fastest.
 * Replaces Volume::getRepresentation<VolumeCL>() + getVolumeStruct
 * (ref progressivephotonmapping/photontracercl.cpp:111-118,139-140).
 * voxels_is_device != 0: `voxels` is a device pointer (device->device copy).
 * voxels == NULL: zero-filled storage (the output of cpm_volume_mix). */
int cpm_volume_create(cpm_ctx* ctx, const cpm_volume_desc* desc, const void* voxels,
                      int voxels_is_device, cpm_stream stream, cpm_volume** out);
/* Replace the voxel data (time-varying sequences; same desc).  From a device source this is one launch (copy +
 * the tracer's footprint copy, 6 x the volume's bytes of traffic); a sequence that stays on the device is better
 * kept as one cpm_volume per time step, which costs nothing per step. */
int cpm_volume_update(cpm_ctx* ctx, cpm_volume* vol, const void* voxels, int voxels_is_device,
                      cpm_stream stream);
void cpm_volume_destroy(cpm_ctx* ctx, cpm_volume* vol);
/* Device address and byte size of the voxel block, READ-ONLY (for consumers that read a volume produced on
 * the device, e.g. cpm_volume_mix's output): the tracer samples a second copy of the voxels laid out
 * by trilinear footprint (4 x the volume's bytes; one fetch per sample) that cpm_volume_create / _update keep in
 * step (cpm_volume_mix leaves it to the next trace that needs it) -- data written through this pointer does not reach it until cpm_volume_update(vol, that same pointer, 1)
 * re-derives it (no copy in that case); and a blocking device->host copy of the block
 * (Volume::getRepresentation<VolumeRAM>() of a volume whose valid representation is the device one). */
void* cpm_volume_device_data(const cpm_volume* vol, size_t* bytes);
int cpm_volume_download(cpm_ctx* ctx, const cpm_volume* vol, void* voxels_host, cpm_stream stream);

/* Transfer function LUT: `width` RGBA32F texels (Inviwo: 1024x1 layer,
 * ref photontracercl.cpp:118).  The tracer reads only alpha. */
int cpm_tf_create(cpm_ctx* ctx, const float* rgba, int width, int rgba_is_device,
                  cpm_stream stream, cpm_tf** out);
int cpm_tf_update(cpm_ctx* ctx, cpm_tf* tf, const float* rgba, int rgba_is_device, cpm_stream stream);
void cpm_tf_destroy(cpm_ctx* ctx, cpm_tf* tf);

/* ------------------------------------------------------------------ emission (E1, E3, E4, E5) */

/* samples[i] = ((0.5 + fmod(i, nx)) / nx, (0.5 + i / nx) / ny, 0, 1), i < nx*ny
 * (row coordinate NOT floored).
 * Replaces uniformSampleGenerator2DKernel
 * (ref importancesamplingcl/cl/uniformsamplegenerator2d.cl:35-52). */
int cpm_uniform_samples_2d(cpm_ctx* ctx, int nx, int ny, float* samples4, cpm_stream stream);

/* Replaces directionalLightSamplerKernel (ref lightcl/cl/directionallightsampler.cl:38-63). */
int cpm_directional_light_samples(cpm_ctx* ctx, const float* samples4, int n,
                                  const float radiance[4], const float direction[4],
                                  const float plane_origin[4], const float tangent_u[4],
                                  const float tangent_v[4], float plane_area,
                                  float* light_samples8, cpm_stream stream);

/* Build-defined point-light emitter (the reference ships no
 * processor that emits point-light samples): origin = position,
 * direction = uniform sphere direction from sample (u, v), power = radiance * 4*pi / pdf_uv. */
int cpm_point_light_samples(cpm_ctx* ctx, const float* samples4, int n, const float radiance[4],
                            const float position[4], float* light_samples8, cpm_stream stream);

/* (tStart, tEnd) of each light sample against the axis-aligned box
 * aabb = (min.xyz, 1, max.xyz, 1); miss -> (0, -1).
 * Replaces lightSampleMeshIntersectionKernel for the cube proxy geometry
 * (ref lightcl/cl/intersection/lightsamplemeshintersection.cl:37-58). */
int cpm_light_sample_box_intersection(cpm_ctx* ctx, const float* light_samples8, int n,
                                      const float aabb[8], float* isect2, cpm_stream stream);

/* Same against a triangle mesh (vertices xyz, `n_indices` ints, 3 per triangle):
 * nearest / farthest hit along the ray; miss -> (0, -1). */
int cpm_light_sample_mesh_intersection(cpm_ctx* ctx, const float* vertices3, const int32_t* indices,
                                       int n_indices, const float* light_samples8, int n,
                                       float* isect2, cpm_stream stream);

/* ------------------------------------------------------------------ trace (R3, R4, R5) */

enum {
    CPM_TRACE_PROGRESSIVE = 1,          /* -D PROGRESSIVE_PHOTON_MAPPING: write RNG state back */
    CPM_TRACE_NO_SINGLE_SCATTERING = 2, /* -D NO_SINGLE_SCATTERING */
    CPM_TRACE_PHOTONS_PLANAR = 4        /* write the records in the two-plane layout (CPM_PHOTONS_PLANAR, cpm_ext.h) */
};
enum { CPM_PHASE_HENYEY_GREENSTEIN = 0, CPM_PHASE_ISOTROPIC = 1 };

/* Scalar arguments of photonTracerKernel
 * (ref progressivephotonmapping/cl/photontracer.cl:69-95,
 *  marshalled at photontracercl.cpp:139-166). */
typedef struct cpm_trace_params {
    float material[4];          /* AdvancedMaterialProperty combined params; .x = anisotropy g */
    float step_size;            /* only the 0.5*step nudge after a scatter uses it */
    int32_t photon_offset;      /* first photon of this light in the photon / RNG arrays */
    int32_t n_light_samples;
    int32_t max_interactions;   /* I */
    int32_t total_photons;      /* N: stride between interactions (SoA by interaction) */
    int32_t shading_type;       /* CPM_PHASE_* */
    int32_t flags;              /* CPM_TRACE_* */
    int32_t iteration;          /* unused by the kernel (kept for signature parity) */
    int32_t batch;              /* unused by the kernel (kept for signature parity) */
} cpm_trace_params;

/* Woodcock-track light samples to <= I interaction points.
 * recompute_indices == NULL: thread i traces light sample i (i < n_light_samples).
 * recompute_indices != NULL (-D PHOTON_RECOMPUTATION variant): thread j traces
 *   light sample recompute_indices[j] - photon_offset, j < n_recompute.
 * tf_scattering may be NULL = the reference's behaviour of passing tf twice
 *   (ref photontracercl.cpp:150-151).
 * rng_state: 2 x uint32 per photon stream, indexed photon_offset + thread.
 * photons: float8[N * I].
 * Replaces PhotonTracerCL::tracePhotons (ref photontracercl.cpp:135-174). */
int cpm_trace(cpm_ctx* ctx, const cpm_volume* vol, const cpm_tf* tf, const cpm_tf* tf_scattering,
              const float aabb[8], const cpm_trace_params* params, const float* light_samples8,
              const float* isect2, const uint32_t* recompute_indices, int n_recompute,
              uint32_t* rng_state, float* photons8, cpm_stream stream);

/* ------------------------------------------------------------------ light volume (grid) */

/* Output light volume: dims, channels (1 = float32, 4 = 4xfloat32: rgb added,
 * a untouched) and its textureToIndex / indexToTexture
 * (ref volumeOutParams, cl/photonstolightvolume.cl:37,46-47,57). */
typedef struct cpm_grid_desc {
    int32_t dims[3];
    int32_t channels;
    float texture_to_index[16];
    float index_to_texture[16];
} cpm_grid_desc;

void cpm_grid_desc_default(cpm_grid_desc* desc, const int32_t dims[3], int32_t channels);

/* relativeIrradianceScale handed to the kernels:
 * (1/pi) / (4/3*pi*r^3 * nPhotons), in double, rounded to float
 * (ref processor/photontolightvolumeprocessorcl.cpp:388-390, photondata.cpp:38,79-81). */
float cpm_relative_irradiance_scale(double radius_relative_to_scene, double n_photons);

/* ---- reference formulation: atomic splat (G1, G3, G4) */

/* grid += splat of photons [0, total_photons) (interaction 0 only: SURVEY Q1).
 * The caller clears the grid (reference: enqueueFillBuffer, ...processorcl.cpp:307).
 * Replaces splatPhotonsToLightVolumeKernel (ref cl/photonstolightvolume.cl:139-166). */
int cpm_splat(cpm_ctx* ctx, const float* photons8, int total_photons, const cpm_grid_desc* grid,
              float radius, float relative_irradiance_scale, float* grid_out, cpm_stream stream);
/* grid += multiplier * splat of photons[indices[j] + k * n_photons], k < n_interactions.
 * Replaces splatSelectedPhotonsToLightVolumeKernel (ref cl/photonstolightvolume.cl:168-202). */
int cpm_splat_selected(cpm_ctx* ctx, const float* photons8, const uint32_t* indices, int n_indices,
                       const cpm_grid_desc* grid, float radius, float relative_irradiance_scale,
                       float multiplier, int n_photons, int n_interactions, float* grid_out,
                       cpm_stream stream);

/* aligned[out_offset + j + k*n_indices] = photons[indices[j] + k*n_photons] (power * multiplier).
 * Replaces copyIndexPhotonsKernel (ref cl/photonstolightvolume.cl:225-248). */
int cpm_copy_indexed_photons(cpm_ctx* ctx, const float* photons8, const uint32_t* indices,
                             int n_indices, float multiplier, int n_photons, int n_interactions,
                             float* aligned8, int out_offset, cpm_stream stream);

/* snapshot[id] = photons[id] for id = indices[j] + k*n_photons, k < n_interactions: refreshes the
 * previous-photons snapshot after a partial re-trace by moving only what changed.  The reference copies the
 * whole buffer every time (ref processor/photontolightvolumeprocessorcl.cpp:343-352: 64 MiB of traffic for
 * a 1 M photon frame of which 0.5 % changed); same result when snapshot held the photons before the re-trace. */
int cpm_snapshot_selected_photons(cpm_ctx* ctx, const float* photons8, const uint32_t* indices, int n_indices,
                                  int n_photons, int n_interactions, float* snapshot8, cpm_stream stream);

/* ---- MI355X formulation: sort/bin + per-cell gather (S6, G1-G3 restated) */

/* Stable LSD radix sort of (key, value) pairs / keys, ascending, `key_bits`
 * low bits significant (0 = all 32).  Result is left in keys/values.
 * Replaces clogs::Radixsort::enqueue
 * (ref radixsortcl/ext/clogs/src/radixsort.cpp:169-259; call sites
 *  processor/progressivephotontracercl.cpp:689-725). */
int cpm_sort_pairs(cpm_ctx* ctx, uint32_t* keys, uint32_t* values, size_t n, int key_bits,
                   cpm_stream stream);
int cpm_sort_keys(cpm_ctx* ctx, uint32_t* keys, size_t n, int key_bits, cpm_stream stream);

/* cpm_bin + cpm_gather are the BIT-EXACT formulation: every voxel's sum is taken in one defined order, so the light volume is
 * identical to the CPU oracle's word for word.  It is the verification path -- what the parity tests and `exactIncrementalUpdate`
 * use -- not the fast one: 0.169 ms per frame at BASELINE config 2 against 0.066 for cpm_bin_fast + cpm_gather_fast (tolerance
 * mode, below) and 0.124 for the reference's own atomic splat (cpm_splat) on the same GPU.
 *
 * Bin photons into light-volume cells.
 *   key(p) = cx + dims.x * (cy + dims.y * cz), c = clamp(floor(p * dims), 0, dims-1);
 *   sentinel photons (any position component == FLT_MAX) get key == cells: they sort behind every real
 *   cell, and cell_start[cells] is the number of stored photons.
 * Sorts (key, photon index) stably, then writes
 *   order[j]        = index (into photons8) of the j-th photon in cell order,
 *   cell_start[c]   = first j with key >= c, c = 0..cells (cells+1 entries),
 *   sorted_pos_power[4j..4j+3] = (x, y, z, powerR)            when grid->channels == 1,
 *   sorted_pos_power[8j..8j+7] = (x, y, z, powerR, powerG, powerB, 0, 0) when == 4.
 * n = number of float8 records to bin (N * I; interaction k of photon i at i + k*N). */
int cpm_bin(cpm_ctx* ctx, const float* photons8, int n, const cpm_grid_desc* grid,
            uint32_t* order, uint32_t* cell_start, float* sorted_pos_power, cpm_stream stream);

/* grid_out[v] = (accumulate ? grid_out[v] : 0) + sum over photons of the
 * cells within reach of voxel v, visited in (dz, dy, cell, sorted index) order,
 * of exactly the terms the reference splat would add to v
 * (same box test, same weight, same != 0 test: ref cl/photonstolightvolume.cl:42-75).
 * No atomics; the result is bitwise reproducible. */
int cpm_gather(cpm_ctx* ctx, const float* sorted_pos_power, const uint32_t* cell_start, int n,
               const cpm_grid_desc* grid, float radius, float relative_irradiance_scale,
               int accumulate, float* grid_out, cpm_stream stream);

/* ------------------------------------------------------------------ correlated re-trace (C1-C7, S2-S4) */

/* Per region^3 brick min/max of the normalised voxel value -> 2 x uint16
 * (value * 65535, round to nearest).  brick grid dims = ceil(dims / region).
 * Replaces volumeMinMaxKernel (ref uniformgridcl/cl/uniformgrid/volumeminmax.cl:33-61). */
int cpm_volume_minmax(cpm_ctx* ctx, const cpm_volume* vol, int region, uint16_t* minmax2,
                      cpm_stream stream);

/* Per-brick mean |v_next - v_cur| of two volumes with equal desc.
 * Replaces VolumeRAMDifferenceAnalysisDispatcher (CPU in the reference:
 * ref uniformgridcl/processors/dynamicvolumedifferenceanalysis.h:96-151). */
int cpm_volume_difference(cpm_ctx* ctx, const cpm_volume* cur, const cpm_volume* next, int region,
                          float* mean_abs_diff, cpm_stream stream);

/* importance[c] = sum of the 4 channel-wise (max - min) of the piecewise-linear
 * TF-difference colour over the brick's [min, max] data range
 * (-D INCREMENTAL_TF_IMPORTANCE).  positions/colors: n_points host arrays
 * (tens of points; consumed by the time the call returns: up to 48 ride in the kernel arguments, no copy is enqueued).
 * Replaces classifyMinMaxUniformGrid3DImportanceKernel
 * (ref importancesamplingcl/cl/minmaxuniformgrid3dimportance.cl:269-289).
 * prev_minmax2 / volume_diff non-NULL: the time-varying variant
 * classifyTimeVaryingMinMaxUniformGrid3DImportanceKernel (:291-330). */
int cpm_importance_tf(cpm_ctx* ctx, const uint16_t* minmax2, const uint16_t* prev_minmax2,
                      const float* volume_diff, int n_cells, const float* positions_host,
                      const float* colors4_host, int n_points, float* importance,
                      cpm_stream stream);
/* importances[photon_offset + i] -= min(0x7fffffff, sat_rtp_u32(100 * sum over the
 * stored poly-line of photon i of cellImportance * dt * |x2 - x1|)).
 * Replaces photonRecomputationDetectorKernel
 * (ref progressivephotonmapping/cl/photonrecomputationdetector.cl:92-157).
 * fix_exit_point != 0 applies the SURVEY Q8 fix (exit = origin + tEnd*dir). */
int cpm_photon_importance(cpm_ctx* ctx, const float* importance_grid, const int32_t grid_dims[3],
                          const float cell_size[3], const float texture_to_index[16],
                          const float* photons8, int photon_offset, const float* light_samples8,
                          const float* isect2, int n_light_samples, int max_interactions,
                          int total_photons, int fix_exit_point, uint32_t* importances,
                          cpm_stream stream);

/* Replaces photonRecomputationDetectorEqualImportanceKernel (ref ...detector.cl:160-194). */
int cpm_photon_importance_equal(cpm_ctx* ctx, int photon_offset, int n_light_samples,
                                int percentage, int iteration, uint32_t* importances,
                                cpm_stream stream);

/* importances[offset .. offset+n) = 0x7fffffff.
 * Replaces resetPhotonImportance (ref processor/progressivephotontracercl.cpp:607-611). */
int cpm_reset_importance(cpm_ctx* ctx, uint32_t* importances, size_t offset, size_t n,
                         cpm_stream stream);

/* Fused threshold + count + iota + sort-by-importance:
 *   *n_changed_dev = #{i : importances[i] < 0x7fffffff}
 *   indices_out    = 0..n-1 sorted stably by importances ascending (most important first);
 *   importances is left sorted too (as the reference's key buffer is).
 * Replaces thresholdKernel + clogs::Reduce + indexToBufferKernel + sortIndicesByImportance
 * (ref cl/threshold.cl:33-40, cl/indextobuffer.cl:33-40,
 *  processor/progressivephotontracercl.cpp:325-363,689-706,727-741). */
int cpm_select_recompute(cpm_ctx* ctx, uint32_t* importances, size_t n, uint32_t* indices_out,
                         int32_t* n_changed_dev, cpm_stream stream);

/* The changed photons only: indices_out = [photons with importance key < 0x7fffffff, ascending index |
 * the others, ascending index], *n_changed_dev = size of the first part; `importances` is not modified.
 * When every changed photon fits the update budget (the usual case after a small edit) their
 * importance ORDER is irrelevant -- the tracer re-sorts the batch by index anyway
 * (ref processor/progressivephotontracercl.cpp:467-473) -- so the caller can use this instead of
 * cpm_select_recompute's 31-bit sort and fall back to it only when n_changed exceeds the budget:
 * a two-launch stable partition (per-tile counts, then ballot ranks behind the counts of the earlier
 * tiles; no atomics), the count is the sum of the tile counts.
 * Replaces thresholdKernel + clogs::Reduce + indexToBufferKernel
 * (ref cl/threshold.cl:33-40, cl/indextobuffer.cl:33-40, ...tracercl.cpp:325-356). */
int cpm_select_changed(cpm_ctx* ctx, const uint32_t* importances, size_t n, uint32_t* indices_out,
                       int32_t* n_changed_dev, cpm_stream stream);

/* ------------------------------------------------------------------ temporal interpolation (time-varying data) */

typedef enum cpm_mix_type { CPM_MIX_F32 = 0, CPM_MIX_U16X2 = 1 } cpm_mix_type;

/* out[i] = mix(x[i], y[i], a) = x + (y - x) * a over device buffers (16-byte aligned).
 * CPM_MIX_F32: n_elements floats (any float/floatN grid).  CPM_MIX_U16X2: n_elements
 * (min, max) pairs of a MinMaxUniformGrid3D; integer vector formats go through
 * convert_float2 / convert_ushort2 (round toward zero) as BufferMixerCL compiles them.
 * Replaces mixKernel (ref uniformgridcl/cl/buffermixer.cl:37-48; uniformgridcl/buffermixercl.cpp:47-92,230-243)
 * as used by UniformGrid3DPlayerProcessor::process
 * (ref uniformgridcl/processors/uniformgrid3dplayerprocessor.cpp:87-115). */
int cpm_mix_buffers(cpm_ctx* ctx, const void* x, const void* y, float a, size_t n_elements,
                    int type, void* out, cpm_stream stream);

/* out = mix of two time steps of a volume sequence, voxel by voxel:
 * normalised values (v / 255, v / 65535, or the float itself) mixed as x * (1 - w) + y * w,
 * stored back in the same format (normalised integers: clamp, scale, round to nearest).
 * `out` is a volume of the same desc (it may alias neither input).
 * The tracer's footprint copy of `out` is NOT rebuilt here: the next trace over all the samples re-derives it first,
 * re-traces of selected photons (cpm_trace with indices, cpm_trace_selected, cpm_photon_importance_retrace) sample the
 * linear block meanwhile -- a time step served by a correlated update never pays the re-layout.
 * Replaces VolumeSequencePlayer::process + glsl/volume_mix.frag
 * (ref uniformgridcl/processors/volumesequenceplayer.cpp:87-140; uniformgridcl/glsl/volume_mix.frag:42-52). */
int cpm_volume_mix(cpm_ctx* ctx, const cpm_volume* v0, const cpm_volume* v1, float weight,
                   cpm_volume* out, cpm_stream stream);

/* ------------------------------------------------------------------ multi-GPU: photon shards + one reduce
 *
 * Photon i depends only on light sample i, RNG stream i and read-only data, so GPU g of D traces photons
 * [g N/D, (g+1) N/D) -- cpm_trace_params.photon_offset / n_light_samples address exactly that
 * (ref cl/photontracer.cl:102,123,166) -- bins and gathers them into its OWN full-size light volume, and the
 * volumes are summed with one collective per frame: RCCL over xGMI, enqueued on the caller's stream.
 * The call site in the reference is where PhotonToLightVolumeProcessorCL::process hands the volume on
 * (ref processor/photontolightvolumeprocessorcl.cpp:356-412).  RCCL is loaded on first use (dlopen): a single-GPU
 * host needs none; without it these return CPM_ERR_UNSUPPORTED. */

typedef struct cpm_comm cpm_comm;
#define CPM_COMM_ID_BYTES 128

/* One process per GPU: rank 0 obtains an id, the host distributes it (any channel), every rank creates its end.
 * cpm_comm_create is collective over the n_ranks participants (ncclCommInitRank). */
int cpm_comm_get_unique_id(cpm_ctx* ctx, uint8_t* id_out /* CPM_COMM_ID_BYTES */);
int cpm_comm_create(cpm_ctx* ctx, const uint8_t* id /* CPM_COMM_ID_BYTES */, int rank, int n_ranks, cpm_comm** out);
/* One process driving n devices (Inviwo's single process; one context per device): all ends at once (ncclCommInitAll). */
int cpm_comm_create_all(cpm_ctx* const* ctxs, int n, cpm_comm** comms_out /* n */);
void cpm_comm_destroy(cpm_comm* comm);
int cpm_comm_rank(const cpm_comm* comm);
int cpm_comm_size(const cpm_comm* comm);

/* recv[i] = sum over ranks of send[i], count floats (cells * channels); recv may be send (in place), or a separate
 * buffer when the rank keeps its partial light volume for incremental updates.
 * The sum's order differs from a single GPU's: tolerance, not bit equality (as in the reference's atomics). */
int cpm_allreduce_grid(cpm_ctx* ctx, cpm_comm* comm, const float* send, float* recv, size_t count, cpm_stream stream);
/* The same to one display GPU only (ncclReduce); recv is read on `root` only. */
int cpm_reduce_grid(cpm_ctx* ctx, cpm_comm* comm, const float* send, float* recv, size_t count, int root, cpm_stream stream);
/* Single-process form: the n ends of cpm_comm_create_all in one grouped call, streams[i] on device i (NULL = null streams). */
int cpm_allreduce_grids(cpm_ctx* const* ctxs, cpm_comm* const* comms, float* const* grids, size_t count,
                        const cpm_stream* streams, int n);

#ifdef __cplusplus
}
#endif
#endif /* CPM_CPM_H */
