/* TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
 * A small end-to-end run of the oracle for the sanitizer build (make -C oracle san; tests/test_oracle_sanitizers.py):
 * every function is called once on small inputs with exact-size heap buffers, so that AddressSanitizer and
 * UndefinedBehaviorSanitizer see an out-of-bounds access, a misaligned load or an integer overflow in the checker. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cpm_oracle.h"

static void matrices(const int dims[3], float t2i[16], float i2t[16]) { /* Inviwo: index = tex * dim - 0.5 */
    memset(t2i, 0, 64); memset(i2t, 0, 64);
    for (int a = 0; a < 3; ++a) {
        t2i[5 * a] = (float)dims[a]; t2i[12 + a] = -0.5f;
        i2t[5 * a] = 1.0f / (float)dims[a]; i2t[12 + a] = 0.5f / (float)dims[a];
    }
    t2i[15] = i2t[15] = 1.0f;
}

int main(void) {
    const int nx = 48, ny = 40, n = nx * ny, vd = 20, I = 2;
    const int vdims[3] = { vd, vd + 3, vd - 5 }, gdims[3] = { 10, 12, 8 };
    size_t nvox = (size_t)vdims[0] * vdims[1] * vdims[2];
    unsigned char* vox = malloc(nvox); unsigned char* vox2 = malloc(nvox);
    for (size_t i = 0; i < nvox; ++i) { vox[i] = (unsigned char)((i * 2654435761u) >> 24); vox2[i] = (unsigned char)((i * 40503u + 17) >> 5); }
    cpmo_volume vol; memset(&vol, 0, sizeof vol);
    memcpy(vol.dims, vdims, sizeof vdims); vol.dtype = CPMO_U8; vol.voxels = vox;
    matrices(vdims, vol.texture_to_index, vol.index_to_texture);
    cpmo_volume vol2 = vol; vol2.voxels = vox2;
    cpmo_grid_desc grid; memset(&grid, 0, sizeof grid);
    memcpy(grid.dims, gdims, sizeof gdims); grid.channels = 1;
    matrices(gdims, grid.texture_to_index, grid.index_to_texture);
    const int W = 64;
    float* tf = malloc(sizeof(float) * 4 * W);
    for (int i = 0; i < W; ++i) { tf[4 * i] = 1.f; tf[4 * i + 1] = .5f; tf[4 * i + 2] = .25f; tf[4 * i + 3] = 0.02f + 0.5f * (float)i / W; }

    float* samples = malloc(sizeof(float) * 4 * n);
    cpmo_uniform_samples_2d(nx, ny, samples);
    const float radiance[4] = { 1, .8f, .6f, 1 }, dir[4] = { 0.2f, 0.3f, -0.93273790f, 0 }, org[4] = { -0.3f, -0.4f, 2.2f, 1 },
                tu[4] = { 1.6f, 0, 0.34f, 0 }, tv[4] = { 0, 1.7f, 0.55f, 0 }, pos[4] = { 0.5f, 0.5f, -1.f, 1 };
    float* ls = malloc(sizeof(float) * 8 * n); float* ls2 = malloc(sizeof(float) * 8 * n);
    cpmo_directional_light_samples(samples, n, radiance, dir, org, tu, tv, 2.72f, ls);
    cpmo_point_light_samples(samples, n, radiance, pos, ls2);
    const float aabb[8] = { 0, 0, 0, 1, 1, 1, 1, 1 };
    float* isect = malloc(sizeof(float) * 2 * n); float* isect2 = malloc(sizeof(float) * 2 * n);
    cpmo_light_sample_box_intersection(ls, n, aabb, isect);
    float verts[24]; int32_t idx[36]; int q = 0;
    for (int z = 0; z < 2; ++z) for (int y = 0; y < 2; ++y) for (int x = 0; x < 2; ++x) { verts[q++] = (float)x; verts[q++] = (float)y; verts[q++] = (float)z; }
    const int quads[6][4] = { {0,1,3,2}, {4,6,7,5}, {0,4,5,1}, {2,3,7,6}, {0,2,6,4}, {1,5,7,3} };
    q = 0;
    for (int f = 0; f < 6; ++f) { idx[q++] = quads[f][0]; idx[q++] = quads[f][1]; idx[q++] = quads[f][2]; idx[q++] = quads[f][0]; idx[q++] = quads[f][2]; idx[q++] = quads[f][3]; }
    cpmo_light_sample_mesh_intersection(verts, idx, 36, ls2, n, isect2);

    uint32_t* rng = malloc(sizeof(uint32_t) * 2 * n);
    uint32_t* bases = malloc(sizeof(uint32_t) * n);
    cpmo_glibc_rand_sequence(0, bases, n);
    for (int i = 0; i < n; ++i) { rng[2 * i] = bases[i]; rng[2 * i + 1] = 0; }
    cpmo_seed_streams(rng, n, 1ull << 40);
    float* draws = malloc(sizeof(float) * 3 * n);
    uint32_t* rng2 = malloc(sizeof(uint32_t) * 2 * n); memcpy(rng2, rng, sizeof(uint32_t) * 2 * n);
    cpmo_random_fill(rng2, n, 3, draws);

    cpmo_trace_params p; memset(&p, 0, sizeof p);
    p.material[0] = 0.4f; p.step_size = 1.0f / vd; p.n_light_samples = n; p.max_interactions = I; p.total_photons = n;
    p.flags = CPMO_TRACE_PROGRESSIVE;
    float* photons = malloc(sizeof(float) * 8 * n * I);
    uint64_t steps = 0;
    cpmo_set_threads(2);
    cpmo_trace(&vol, tf, W, tf, aabb, &p, ls, isect, NULL, 0, rng, photons, &steps);
    uint32_t sel[5] = { 3, 7, 100, 1000, (uint32_t)n - 1 };
    p.flags = 0;
    cpmo_trace(&vol, tf, W, NULL, aabb, &p, ls, isect, sel, 5, rng, photons, NULL);

    const float radius = 0.9f / 12.f, scale = cpmo_relative_irradiance_scale(radius, n);
    size_t cells = (size_t)gdims[0] * gdims[1] * gdims[2];
    float* lv = calloc(cells, sizeof(float)); float* lv2 = calloc(cells, sizeof(float));
    cpmo_splat(photons, n, &grid, radius, scale, lv);
    cpmo_splat_selected(photons, sel, 5, &grid, radius, scale, -1.f, n, I, lv);
    float* aligned = malloc(sizeof(float) * 8 * 5 * I);
    cpmo_copy_indexed_photons(photons, sel, 5, 1.f, n, I, aligned, 0);
    uint32_t* order = malloc(sizeof(uint32_t) * n * I); uint32_t* cs = malloc(sizeof(uint32_t) * (cells + 1));
    float* sorted = malloc(sizeof(float) * 4 * n * I);
    cpmo_bin(photons, n * I, &grid, order, cs, sorted);
    cpmo_gather(sorted, cs, n * I, &grid, radius, scale, 0, lv2);
    cpmo_gather(sorted, cs, n * I, &grid, radius, scale, 1, lv2);
    uint32_t* keys = malloc(sizeof(uint32_t) * n); uint32_t* vals = malloc(sizeof(uint32_t) * n);
    for (int i = 0; i < n; ++i) { keys[i] = bases[i] & 0xfffffu; vals[i] = (uint32_t)i; }
    cpmo_sort_pairs(keys, vals, n, 20);
    cpmo_sort_keys(keys, n, 0);

    const int region = 8;
    int o[3]; for (int a = 0; a < 3; ++a) o[a] = (vdims[a] + region - 1) / region;
    size_t nb = (size_t)o[0] * o[1] * o[2];
    uint16_t* mm = malloc(sizeof(uint16_t) * 2 * nb); uint16_t* mm2 = malloc(sizeof(uint16_t) * 2 * nb); uint16_t* mmx = malloc(sizeof(uint16_t) * 2 * nb);
    float* diff = malloc(sizeof(float) * nb); float* imp = malloc(sizeof(float) * nb); float* impx = malloc(sizeof(float) * nb);
    cpmo_volume_minmax(&vol, region, mm); cpmo_volume_minmax(&vol2, region, mm2);
    cpmo_volume_difference(&vol, &vol2, region, diff);
    const float pp[4] = { 0.f, 0.3f, 0.6f, 1.f };
    const float pc[16] = { 0,0,0,0, .1f,.2f,.3f,.4f, .5f,.1f,.2f,.9f, 0,0,0,0 };
    cpmo_importance_tf(mm, NULL, NULL, (int)nb, pp, pc, 4, imp);
    cpmo_importance_tf(mm2, mm, diff, (int)nb, pp, pc, 4, imp);
    const int32_t gd[3] = { o[0], o[1], o[2] }; const float cellsz[3] = { 8.f, 8.f, 8.f };
    uint32_t* pimp = malloc(sizeof(uint32_t) * n); uint32_t* pidx = malloc(sizeof(uint32_t) * n);
    for (int i = 0; i < n; ++i) pimp[i] = 2147483647u;
    cpmo_photon_importance(imp, gd, cellsz, vol.texture_to_index, photons, 0, ls, isect, n, I, n, 1, pimp);
    int32_t cnt = 0, cnt2 = 0;
    cpmo_select_changed(pimp, n, pidx, &cnt2);
    cpmo_select_recompute(pimp, n, pidx, &cnt);
    cpmo_photon_importance_equal(0, n, 25, 3, pimp);
    cpmo_mix_f32(imp, diff, 0.3f, nb, impx);
    cpmo_mix_u16x2(mm, mm2, 0.7f, nb, mmx);
    unsigned char* voxm = malloc(nvox);
    cpmo_volume_mix(&vol, &vol2, 0.4f, voxm);
    float ang[2], d3[3] = { 0.3f, -0.5f, 0.81240384f }, back[3];
    cpmo_encode_direction(d3, ang); cpmo_decode_direction(ang, back);

    double s1 = 0, s2 = 0;
    for (size_t i = 0; i < cells; ++i) { s1 += lv[i]; s2 += lv2[i]; }
    printf("selftest ok: steps %llu, changed %d/%d, splat sum %.6g, gather sum %.6g, dir %.3f %.3f %.3f\n",
           (unsigned long long)steps, cnt, cnt2, s1, s2, back[0], back[1], back[2]);
    free(vox); free(vox2); free(tf); free(samples); free(ls); free(ls2); free(isect); free(isect2); free(rng); free(bases); free(draws);
    free(rng2); free(photons); free(lv); free(lv2); free(aligned); free(order); free(cs); free(sorted); free(keys); free(vals);
    free(mm); free(mm2); free(mmx); free(diff); free(imp); free(impx); free(pimp); free(pidx); free(voxm);
    return (cnt == cnt2 && s2 > 0) ? 0 : 1;
}
