/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT.
 *
 * ref_harness_vec.c -- driver for the reference's modules/uniformgridcl/cl/buffermixer.cl (mixKernel), which
 * includes nothing and is compiled from where it lies, once as BufferMixerCL::compileKernel builds it for a float
 * grid (-DMIX_T=float) and once for the min/max grid (-DMIX_T=ushort2 with the convert_float2 / convert_ushort2
 * pair, buffermixercl.cpp:235-239).  Compiled with the same clang as the kernels so that the OpenCL vector types
 * have one ABI.  Supplied here, as an OpenCL runtime would: the work-item id (ref_harness.c) and the built-ins the
 * kernel calls, each with the definition the OpenCL 1.2 specification gives:
 *   mix(x, y, a)            "the linear blend of x and y implemented as x + (y - x) * a"   (s6.12.4)
 *   convert_float2(ushort2) exact (every u16 is a float)                                    (s6.2.3)
 *   convert_ushort2(float2) default rounding of float -> integer conversions: toward zero   (s6.2.3.3)
 * Built with -ffp-contract=off: the blend is three rounded operations.
 */
#include <stddef.h>
#include <stdint.h>

typedef float float2 __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2 __attribute__((ext_vector_type(2)));

extern void ref_set_global_id(size_t id);

float _Z3mixfff(float x, float y, float a) { return x + (y - x) * a; }
float2 _Z3mixDv2_fS_f(float2 x, float2 y, float a) { return x + (y - x) * a; }
float2 _Z14convert_float2Dv2_t(ushort2 v) { return (float2){ (float)v.x, (float)v.y }; }
ushort2 _Z15convert_ushort2Dv2_f(float2 v) { return (ushort2){ (unsigned short)(int)v.x, (unsigned short)(int)v.y }; }

extern void mixKernel_f32(const float* x, const float* y, float a, unsigned len, float* out);
extern void mixKernel_u16x2(const ushort2* x, const ushort2* y, float a, unsigned len, ushort2* out);

/* launch shape of BufferMixerCL::mix (buffermixercl.cpp:80-92): global size rounded up to the work-group size */
void ref_mix_f32(const float* x, const float* y, float a, unsigned n, float* out, unsigned wg) {
    size_t global = ((size_t)n + wg - 1) / wg * wg;
    for (size_t i = 0; i < global; ++i) { ref_set_global_id(i); mixKernel_f32(x, y, a, n, out); }
}
void ref_mix_u16x2(const uint16_t* x, const uint16_t* y, float a, unsigned n_pairs, uint16_t* out, unsigned wg) {
    size_t global = ((size_t)n_pairs + wg - 1) / wg * wg;
    for (size_t i = 0; i < global; ++i) {
        ref_set_global_id(i);
        mixKernel_u16x2((const ushort2*)x, (const ushort2*)y, a, n_pairs, (ushort2*)out);
    }
}
