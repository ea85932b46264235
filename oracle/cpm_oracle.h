/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
 * anything under oracle/.
 *
 * cpm_oracle.h -- CPU restatement (plain C) of the reference's
 * trace -> sort/bin -> gather path and of the correlated re-trace helpers.
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/modules).
 *
 * PARITY PINNING.  Pinned against the reference itself, compiled from its own
 * sources into oracle/_ref (oracle/Makefile): the MWC64X generator, its
 * skip-ahead seeding, random_01, the Epanechnikov density kernel, threshold
 * and iota kernels.  PARITY UNPINNED for everything that touches Inviwo's
 * shared .cl headers (samplers.cl, transformations.cl, shading/shading.cl,
 * intersection/ *.cl, image3d_write.cl), which are not part of the reference
 * tree: image sampling, transformPoint, encode/decodeDirection,
 * rayBoxIntersection, phase-function sampling.  Those are restated from the
 * OpenCL 1.2 specification and the reference's own call sites and host twins,
 * and flagged [INVIWO] below.
 */
#ifndef CPM_ORACLE_H
#define CPM_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { CPMO_U8 = 0, CPMO_U16 = 1, CPMO_F32 = 2 };
enum { CPMO_TRACE_PROGRESSIVE = 1, CPMO_TRACE_NO_SINGLE_SCATTERING = 2 };
enum { CPMO_PHASE_HENYEY_GREENSTEIN = 0, CPMO_PHASE_ISOTROPIC = 1 };

typedef struct cpmo_volume {
    int32_t dims[3];
    int32_t dtype;
    float format_scaling;
    float format_offset;
    float texture_to_index[16];
    float index_to_texture[16];
    const void* voxels;
} cpmo_volume;

typedef struct cpmo_trace_params {
    float material[4];
    float step_size;
    int32_t photon_offset;
    int32_t n_light_samples;
    int32_t max_interactions;
    int32_t total_photons;
    int32_t shading_type;
    int32_t flags;
    int32_t iteration;
    int32_t batch;
} cpmo_trace_params;

typedef struct cpmo_grid_desc {
    int32_t dims[3];
    int32_t channels;
    float texture_to_index[16];
    float index_to_texture[16];
} cpmo_grid_desc;

/* number of OpenMP threads used by the parallel loops (trace, gather, minmax); default 1 */
void cpmo_set_threads(int n);
int cpmo_get_threads(void);

/* math contract, exposed for unit tests */
float cpmo_log(float x);
void cpmo_sincos(float x, float* s, float* c);
float cpmo_acos(float x);
float cpmo_atan2(float y, float x);
void cpmo_encode_direction(const float d[3], float angles[2]);
void cpmo_decode_direction(const float angles[2], float d[3]);
float cpmo_density_kernel(float x);

/* RNG */
void cpmo_glibc_rand_sequence(uint32_t seed, uint32_t* out, size_t n);
void cpmo_seed_streams(uint32_t* state, size_t n, uint64_t gap);
void cpmo_random_fill(uint32_t* state, size_t n, int draws, float* out);
void cpmo_mwc64x_next(uint32_t* x, uint32_t* c, uint32_t* out_uint, float* out_01);

/* emission */
void cpmo_uniform_samples_2d(int nx, int ny, float* samples4);
void cpmo_directional_light_samples(const float* samples4, int n, const float radiance[4],
                                    const float direction[4], const float plane_origin[4],
                                    const float tangent_u[4], const float tangent_v[4],
                                    float plane_area, float* light_samples8);
void cpmo_point_light_samples(const float* samples4, int n, const float radiance[4],
                              const float position[4], float* light_samples8);
void cpmo_light_sample_box_intersection(const float* light_samples8, int n, const float aabb[8],
                                        float* isect2);
void cpmo_light_sample_mesh_intersection(const float* vertices3, const int32_t* indices,
                                         int n_indices, const float* light_samples8, int n,
                                         float* isect2);

/* trace; steps_out (nullable) receives the total number of Woodcock iterations */
void cpmo_trace(const cpmo_volume* vol, const float* tf_rgba, int tf_width,
                const float* tf_scattering_rgba, const float aabb[8],
                const cpmo_trace_params* params, const float* light_samples8, const float* isect2,
                const uint32_t* recompute_indices, int n_recompute, uint32_t* rng_state,
                float* photons8, uint64_t* steps_out);
/* statistics hook: non-null = cpmo_trace also stores each light sample's Woodcock iteration count (index = threadId) */
void cpmo_debug_set_step_array(uint32_t* per_photon_steps);
float cpmo_sample_volume(const cpmo_volume* vol, float x, float y, float z);
float cpmo_sample_tf_alpha(const float* tf_rgba, int width, float v);

/* light volume */
float cpmo_relative_irradiance_scale(double radius_relative_to_scene, double n_photons);
void cpmo_splat(const float* photons8, int total_photons, const cpmo_grid_desc* grid, float radius,
                float scale, float* grid_out);
void cpmo_splat_selected(const float* photons8, const uint32_t* indices, int n_indices,
                         const cpmo_grid_desc* grid, float radius, float scale, float multiplier,
                         int n_photons, int n_interactions, float* grid_out);
void cpmo_copy_indexed_photons(const float* photons8, const uint32_t* indices, int n_indices,
                               float multiplier, int n_photons, int n_interactions,
                               float* aligned8, int out_offset);
void cpmo_sort_pairs(uint32_t* keys, uint32_t* values, size_t n, int key_bits);
void cpmo_sort_keys(uint32_t* keys, size_t n, int key_bits);
void cpmo_bin(const float* photons8, int n, const cpmo_grid_desc* grid, uint32_t* order,
              uint32_t* cell_start, float* sorted_pos_power);
void cpmo_gather(const float* sorted_pos_power, const uint32_t* cell_start, int n,
                 const cpmo_grid_desc* grid, float radius, float scale, int accumulate,
                 float* grid_out);

/* tolerance-mode formulation (cpm_bin_fast + cpm_gather_fast), straight from the photon records: fixed-point sums */
void cpmo_gather_fast(const float* photons8, int n, const cpmo_grid_desc* grid, float radius, float scale,
                      int accumulate, float* grid_out);

/* correlated re-trace */
void cpmo_volume_minmax(const cpmo_volume* vol, int region, uint16_t* minmax2);
void cpmo_volume_difference(const cpmo_volume* cur, const cpmo_volume* next, int region,
                            float* mean_abs_diff);
void cpmo_importance_tf(const uint16_t* minmax2, const uint16_t* prev_minmax2,
                        const float* volume_diff, int n_cells, const float* positions,
                        const float* colors4, int n_points, float* importance);
void cpmo_photon_importance(const float* importance_grid, const int32_t grid_dims[3],
                            const float cell_size[3], const float texture_to_index[16],
                            const float* photons8, int photon_offset, const float* light_samples8,
                            const float* isect2, int n_light_samples, int max_interactions,
                            int total_photons, int fix_exit_point, uint32_t* importances);
void cpmo_photon_importance_equal(int photon_offset, int n_light_samples, int percentage,
                                  int iteration, uint32_t* importances);
void cpmo_select_recompute(uint32_t* importances, size_t n, uint32_t* indices_out,
                           int32_t* n_changed);

void cpmo_select_changed(const uint32_t* importances, size_t n, uint32_t* indices_out,
                         int32_t* n_changed);

/* temporal interpolation */
void cpmo_mix_f32(const float* x, const float* y, float a, size_t n, float* out);
void cpmo_mix_u16x2(const uint16_t* x, const uint16_t* y, float a, size_t n_pairs, uint16_t* out);
void cpmo_volume_mix(const cpmo_volume* v0, const cpmo_volume* v1, float weight, void* out_voxels);

/* ---- host arithmetic that decides device inputs (cpm_oracle_host.c) -------------------------------------- */
/* E2: lightcl/convexhull2d.cpp:38-130 (hull_xy: room for n + 2 points; returns the count) */
int cpmo_convex_hull_2d(const float* xy, int n, float* hull_xy);
/* E2: lightcl/orientedboundingbox2d.cpp:40-78 (out = origin.xy, u.xy, v.xy) */
void cpmo_minimum_bounding_rectangle(const float* hull_xy, int n, float out[6]);
/* E2: lightcl/orientedboundingbox2d.cpp:80-100 + pointplaneprojection.cpp:39-54 (out = origin.xyz, u.xyz, v.xyz) */
void cpmo_fit_obb(const float* points_xyz, int n, const float plane_point[3], const float plane_normal[3], float out[9]);
/* C2: importancesamplingcl/processors/minmaxuniformgrid3dimportanceclprocessor.cpp:364-524 (outputs: room for
 * n_tf + n_prev + 2 points; returns the count, -1 when exactly one function is empty) */
int cpmo_tf_difference_points(const double* tf_pos, const float* tf_rgba, int n_tf, const double* prev_pos, const float* prev_rgba, int n_prev,
                              float eps, int associated, float* out_pos, float* out_rgba);

#ifdef __cplusplus
}
#endif
#endif
