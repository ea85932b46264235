/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
 *
 * ref_harness.c -- host-side driver for the reference's own OpenCL C sources,
 * compiled for x86-64 by oracle/Makefile (clang -x cl) from where they lie
 * under /root/reference.  Only reference files that include nothing outside
 * the reference tree are built:
 *     modules/rndgenmwc64x/cl/randstategen.cl (+ random.cl, skip_mwc.cl)
 *     modules/progressivephotonmapping/cl/densityestimationkernel.cl
 *     modules/progressivephotonmapping/cl/threshold.cl
 *     modules/progressivephotonmapping/cl/indextobuffer.cl
 *     modules/uniformgridcl/cl/buffermixer.cl (driven by ref_harness_vec.c)
 *     modules/progressivephotonmapping/cl/photon.cl, modules/rndgenmwc64x/cl/randomnumbergenerator.cl
 *         (a second library, driven by ref_harness2.c)
 * No header of the reference or of Inviwo is stubbed.  What this file supplies
 * is the execution harness an OpenCL runtime would: the work-item id and the
 * two integer built-ins those kernels call, whose results the OpenCL 1.2
 * specification defines exactly: mad_hi(a, b, c) = mul_hi(a, b) + c (s6.12.3)
 * and convert_uint(int) = the C cast (s6.2.3).
 * Files that include Inviwo's shared .cl headers (photontracer.cl,
 * photonstolightvolume.cl, ...) are unbuildable here and are not attempted.
 */
#include <stddef.h>
#include <stdint.h>

static __thread size_t g_global_id;
void ref_set_global_id(size_t id) { g_global_id = id; }  /* for ref_harness_vec.c */

/* size_t get_global_id(uint) -- Itanium mangling used by clang's OpenCL C front end */
size_t _Z13get_global_idj(unsigned dim) { return dim == 0 ? g_global_id : 0; }
/* uint mad_hi(uint, uint, uint) */
unsigned _Z6mad_hijjj(unsigned a, unsigned b, unsigned c) {
    return (unsigned)(((uint64_t)a * (uint64_t)b) >> 32) + c;
}

/* uint convert_uint(int) (threshold.cl:39 applies it to a comparison result) */
unsigned _Z12convert_uinti(int v) { return (unsigned)v; }

/* symbols exported by the compiled reference objects */
typedef struct { uint32_t x, c; } random_state;
extern void MWC64X_GenerateRandomState(uint32_t* seeds, int size);
extern void MWC64X_GeneratePerStreamRandomState(uint32_t* seeds, uint64_t maxSamplesPerStream, int size);
extern float random_01(random_state* r);
extern uint32_t MWC64X_NextUint(random_state* r);
extern float densityEstimationKernel(float x);
extern void thresholdKernel(const uint32_t* data, uint32_t threshold, int nElements, uint32_t* out);
extern void indexToBufferKernel(uint32_t* indices, int nElements);

void ref_generate_random_state(uint32_t* seeds, int n) {
    /* launch shape of mwc64xseedgenerator.cpp:78-84: global size rounded up to 256 */
    int global = ((n + 255) / 256) * 256;
    for (int i = 0; i < global; ++i) { g_global_id = (size_t)i; MWC64X_GenerateRandomState(seeds, n); }
}
void ref_generate_per_stream_random_state(uint32_t* seeds, uint64_t gap, int n) {
    int global = ((n + 255) / 256) * 256;
    for (int i = 0; i < global; ++i) { g_global_id = (size_t)i; MWC64X_GeneratePerStreamRandomState(seeds, gap, n); }
}
void ref_random_fill(uint32_t* state, int n, int draws, float* out, uint32_t* out_uint) {
    for (int i = 0; i < n; ++i) {
        random_state s = { state[2 * i], state[2 * i + 1] };
        for (int k = 0; k < draws; ++k) {
            if (out_uint) {
                random_state t = s;
                out_uint[i + (size_t)k * n] = MWC64X_NextUint(&t);
            }
            out[i + (size_t)k * n] = random_01(&s);
        }
        state[2 * i] = s.x;
        state[2 * i + 1] = s.c;
    }
}
float ref_density_kernel(float x) { return densityEstimationKernel(x); }
void ref_threshold(const uint32_t* data, uint32_t threshold, int n, uint32_t* out) {
    int global = ((n + 127) / 128) * 128;
    for (int i = 0; i < global; ++i) { g_global_id = (size_t)i; thresholdKernel(data, threshold, n, out); }
}
void ref_index_to_buffer(uint32_t* idx, int n) {
    int global = ((n + 127) / 128) * 128;
    for (int i = 0; i < global; ++i) { g_global_id = (size_t)i; indexToBufferKernel(idx, n); }
}
