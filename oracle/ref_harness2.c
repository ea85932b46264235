/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
 *
 * ref_harness2.c -- driver for two more reference files that include nothing outside the reference tree, compiled
 * from where they lie by oracle/Makefile into a second library (_ref/libcpm_ref2.so: randomnumbergenerator.cl pulls in
 * random.cl again, whose functions libcpm_ref.so already defines):
 *     modules/progressivephotonmapping/cl/photon.cl             readPhoton / writePhoton (the float8 record, T1)
 *     modules/rndgenmwc64x/cl/randomnumbergenerator.cl          randomNumberGeneratorKernel: loadRandState ->
 *                                                               random_01 -> saveRandState (T5; the write-back
 *                                                               photontracer.cl:211-215 performs the same way)
 * Compiled with the same clang as the kernels (one ABI for float8).  Supplied here, as an OpenCL runtime would: the
 * work-item id and mad_hi (OpenCL 1.2 s6.12.3: mul_hi(a, b) + c).  randomnumbergenerator.cl's second kernel writes an
 * image (write_imagef); it is never called and the linker drops it (-ffunction-sections, hidden visibility,
 * --gc-sections), so no image built-in is needed.
 */
#include <stddef.h>
#include <stdint.h>

typedef float float8 __attribute__((ext_vector_type(8)));

static __thread size_t g_global_id;
__attribute__((visibility("hidden"))) size_t _Z13get_global_idj(unsigned dim) { return dim == 0 ? g_global_id : 0; }
__attribute__((visibility("hidden"))) unsigned _Z6mad_hijjj(unsigned a, unsigned b, unsigned c) {
    return (unsigned)(((uint64_t)a * (uint64_t)b) >> 32) + c;
}

extern float8 readPhoton(const float8* photonData, int photonId);
extern void writePhoton(float8 photon, float8* photonData, int photonId);
extern void randomNumberGeneratorKernel(uint32_t* randomSeeds /* uint2 per stream */, int size, float* generatedNumbers);

/* photonData[ids[i]] = photons[i] through the reference's writePhoton, then out[i] = readPhoton(photonData, ids[i]) */
__attribute__((visibility("default"))) void ref_photon_write_read(const float* photons8, const int* ids, int n, float* photonData, float* out8) {
    for (int i = 0; i < n; ++i) {
        float8 p;
        for (int k = 0; k < 8; ++k) p[k] = photons8[8 * i + k];
        writePhoton(p, (float8*)photonData, ids[i]);
    }
    for (int i = 0; i < n; ++i) {
        float8 p = readPhoton((const float8*)photonData, ids[i]);
        for (int k = 0; k < 8; ++k) out8[8 * i + k] = p[k];
    }
}

/* one launch of randomNumberGeneratorKernel: global size rounded up to 256 as in randomnumbergenerator.cpp */
__attribute__((visibility("default"))) void ref_random_number_kernel(uint32_t* seeds, int n, float* out) {
    int global = ((n + 255) / 256) * 256;
    for (int i = 0; i < global; ++i) { g_global_id = (size_t)i; randomNumberGeneratorKernel(seeds, n, out); }
}
