// valu_rate.hip -- what a gfx950 SIMD issues per clock: wave-instructions per second for
//   * independent / dependent v_fma_f32, the quarter-rate classes (v_log_f32, v_rcp_f32, v_mul_lo_u32, v_mad_u64_u32),
//   * the Woodcock loop's arithmetic with its memory accesses removed (the contract's rand01_, log_, coord clamps, lerps),
// at 1, 2, 4 and 8 waves per SIMD.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I <pkg>/csrc -I include tools/valu_rate.hip -o build/valu_rate && build/valu_rate
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "cpm_math.hip.h"

using namespace cpm;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

enum { M_FMA_INDEP, M_FMA_DEP, M_LOG, M_RCP, M_MUL_LO, M_MAD64, M_CVT, M_LOOP_MIX, M_COUNT };
static const char* kNames[M_COUNT] = { "v_fma_f32 x16 independent", "v_fma_f32 x16 one chain", "v_log_f32 x16 independent", "v_rcp_f32 x16 independent",
                                       "v_mul_lo_u32 x16 independent", "v_mad_u64_u32 x16 independent", "v_cvt_f32_u32 x16 independent",
                                       "Woodcock loop arithmetic (no memory)" };
// wave-instructions per loop iteration (the loop-mix figure is read from the ISA: see main)
static int kPerIter[M_COUNT] = { 16, 16, 16, 16, 16, 16, 16, 0 };

template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters, float seed, unsigned long long* clocks) {
    extern __shared__ float pad[];  // sized by the host to cap the workgroups per CU
    const unsigned long long c0 = clock64(), w0 = wall_clock64();  // s_memtime (shader clock) / s_memrealtime (100 MHz)
    float a[16];
    uint32_t u[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = seed + (float)(threadIdx.x + i); u[i] = (uint32_t)(threadIdx.x * 16 + i) | 1u; }
    const float b = seed * 0.5f + 1.0f, c = seed + 0.25f;
    if (MODE == M_LOOP_MIX) {
        // the arithmetic of one Woodcock candidate (cpm_trace.hip woodcock + sample_volume + sample_alpha) with the footprint
        // fetch replaced by register values and the LUT read by a lerp of two registers: 2 draws, log, position, 3 + 1 clamps,
        // 7 + 1 lerps, conversions, accept test
        uint32_t rx = u[0], rc = u[1];
        float t = 0.f, acc = 0.f;
        const float dx = a[0] * 1e-3f, dy = a[1] * 1e-3f, dz = a[2] * 1e-3f;
        for (int it = 0; it < iters; ++it) {
            float u1 = rand01_(rx, rc);
            t = fma_(-log_(u1), 1.f / 150.f, t);
            float px = fma_(t, dx, b), py = fma_(t, dy, c), pz = fma_(t, dz, b);
            float fl[3], al[3];
            const float p3[3] = { px, py, pz };
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float uu = __builtin_amdgcn_fmed3f(fma_(p3[k], 256.f, -0.5f), 0.0f, 255.f);
                fl[k] = __builtin_amdgcn_fmed3f(__builtin_floorf(uu), 0.0f, 254.f);
                al[k] = uu - fl[k];
            }
            uint32_t idx = (uint32_t)(int)fl[0] + __umul24(256u, (uint32_t)(int)fl[1]) + __umul24(65536u, (uint32_t)(int)fl[2]);
            uint32_t w0 = idx * 2654435761u, w1 = idx ^ rx;  // stand-ins for the two fetched words
            float v[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { v[k] = (float)((w0 >> (8 * k)) & 0xffu); v[4 + k] = (float)((w1 >> (8 * k)) & 0xffu); }
            float c00 = lerp_(v[0], v[4], al[0]), c10 = lerp_(v[1], v[5], al[0]), c01 = lerp_(v[2], v[6], al[0]), c11 = lerp_(v[3], v[7], al[0]);
            float c0 = lerp_(c00, c10, al[1]), c1 = lerp_(c01, c11, al[1]);
            float vs = lerp_(c0, c1, al[2]) * (1.f / 255.f);
            vs = (vs + 0.f) * 1.f;
            const float uu = __builtin_amdgcn_fmed3f(fma_(vs, 1024.f, -0.5f), 0.0f, 1023.f);
            const float fi = __builtin_amdgcn_fmed3f(__builtin_floorf(uu), 0.0f, 1022.f);
            const float opacity = lerp_(pad[(int)fi & 63], pad[((int)fi + 1) & 63], uu - fi);  // (the LUT read is an LDS access here too)
            float u2 = rand01_(rx, rc);
            acc += (u2 >= opacity) ? 1.f : 0.f;
        }
        a[0] = acc + t + (float)rx;
    } else {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == M_FMA_INDEP) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
                else if (MODE == M_FMA_DEP) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(b), "v"(c));
                else if (MODE == M_LOG) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
                else if (MODE == M_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
                else if (MODE == M_MUL_LO) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 15]));
                else if (MODE == M_MAD64) {
                    uint64_t r;
                    asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(r) : "v"(u[i]), "v"(kMwcA), "v"((uint64_t)u[(i + 1) & 15]) : "vcc");
                    u[i] = (uint32_t)r ^ (uint32_t)(r >> 32);
                } else if (MODE == M_CVT) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[i]) : "v"(u[i]));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i] + (float)u[i];
    if (s == 12345.678f) out[threadIdx.x] = s + pad[0];  // keeps the loop alive, never true in practice
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = clock64() - c0; clocks[1] = wall_clock64() - w0; }
}

static double g_clock_ghz = 0;  // shader clock of the last run, from the kernel's own counters

template <int MODE>
static double run(float* out, int wps, int iters, int cus) {
    static unsigned long long* clocks = nullptr;
    if (!clocks) CHECK(hipHostMalloc((void**)&clocks, 16, hipHostMallocMapped));
    // wps waves per SIMD = wps workgroups of 256 threads per CU; LDS per workgroup caps the residency at that
    size_t lds = (size_t)(160 * 1024 / wps) - 1024;
    if (lds > 64 * 1024) {
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(rate_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    dim3 grid(cus * wps), block(256);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate_kernel<MODE>, grid, block, lds, 0, out, iters, 1.0f, clocks);  // warm-up of the same length: clocks ramp
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {  // the fastest of three
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(rate_kernel<MODE>, grid, block, lds, 0, out, iters, 1.0f, clocks);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) { best = ms; g_clock_ghz = clocks[1] ? (double)clocks[0] / ((double)clocks[1] / 100e6) * 1e-9 : 0; }
    }
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return (double)best * 1e-3;
}

int main(int argc, char** argv) {
    int loop_mix_insts = argc > 1 ? atoi(argv[1]) : 0;  // VALU instructions per iteration of the loop-mix kernel (from the ISA)
    kPerIter[M_LOOP_MIX] = loop_mix_insts;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("device %s, %d CUs, %.2f GHz nominal\n", prop.name, cus, ghz);
    float* out;
    CHECK(hipMalloc(&out, 4096));
    const int iters = 100000;
    printf("%-40s %5s %10s %14s %10s %14s\n", "stream", "w/SIMD", "ms", "Gwave-inst/s", "clock GHz", "inst/clk/SIMD");
    for (int mode = 0; mode < M_COUNT; ++mode) {
        if (mode == M_LOOP_MIX && loop_mix_insts <= 0) continue;
        for (int wps : { 1, 2, 4, 8 }) {
            double s = 0;
            switch (mode) {
                case M_FMA_INDEP: s = run<M_FMA_INDEP>(out, wps, iters, cus); break;
                case M_FMA_DEP: s = run<M_FMA_DEP>(out, wps, iters, cus); break;
                case M_LOG: s = run<M_LOG>(out, wps, iters, cus); break;
                case M_RCP: s = run<M_RCP>(out, wps, iters, cus); break;
                case M_MUL_LO: s = run<M_MUL_LO>(out, wps, iters, cus); break;
                case M_MAD64: s = run<M_MAD64>(out, wps, iters, cus); break;
                case M_CVT: s = run<M_CVT>(out, wps, iters, cus); break;
                default: s = run<M_LOOP_MIX>(out, wps, iters, cus); break;
            }
            const double waves = (double)cus * wps * 4;
            const double insts = waves * iters * kPerIter[mode];
            const double clk = g_clock_ghz > 0 ? g_clock_ghz : ghz;  // the launch's own shader clock (s_memtime against the 100 MHz s_memrealtime)
            printf("%-40s %5d %10.3f %14.1f %10.3f %14.3f\n", kNames[mode], wps, s * 1e3, insts / s * 1e-9, clk, insts / s / (cus * 4.0) / (clk * 1e9));
        }
    }
    CHECK(hipFree(out));
    return 0;
}
