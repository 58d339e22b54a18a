// mfma_peak.hip — register-only MFMA issue-rate probe for gfx950.
// Measures what the fp32 matrix pipe sustains on THIS box (clock under load included), so the
// roofline fractions of the convolution kernels can be read against the practical ceiling as well as
// the 157.3 TFLOP/s data-sheet figure.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC>
__global__ void __launch_bounds__(256) k_16x16x4(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) k_32x32x2(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 12345.678f) out[0] = s;
}

// same issue pattern with operands that toggle like real data: per-lane pseudo-random A/B values in [-1, 1), 8 of
// each rotated through the loop, accumulators doing a bounded random walk (no inf/NaN, full mantissa activity).
// Compares the matrix pipe's sustained clock under realistic switching power with the constant-operand run above.
__global__ void __launch_bounds__(256) k_16x16x4_rand(float* out, int iters, unsigned seed) {
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a[8], b[8];
    unsigned h = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u;
        a[i] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
        h = h * 1664525u + 1013904223u;
        b[i] = (float)(int)(h >> 8) * (1.0f / 8388608.0f) - 1.0f;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(i + r) & 7], b[(i * 3 + r) & 7], acc[i], 0, 0, 0);
        // sign flip keeps the walk bounded and the operands changing
#pragma unroll
        for (int i = 0; i < 8; ++i) a[i] = -a[i];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

// plain VALU fma chain (v_fma_f32 / v_pk_fma_f32 as the compiler chooses) for the fused ConvLSTM kernels' ceiling
__global__ void __launch_bounds__(256) k_valu(float* out, int iters, float a0, float b0) {
    float acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = (float)i;
    float a = a0 + threadIdx.x * 1e-9f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[i] = fmaf(acc[i], a, b);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i];
    if (s == 12345.678f) out[0] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename F>
static int time_it(const char* name, F launch, double flops_per_launch, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch();
    CK(hipDeviceSynchronize());
    float best = 1e30f, tot = 0.f;
    for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(e0));
        launch();
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        tot += ms;
    }
    printf("%-34s best %8.3f ms  %7.1f TFLOP/s   mean %8.3f ms  %7.1f TFLOP/s\n", name, best,
           flops_per_launch / best * 1e-9, tot / reps, flops_per_launch / (tot / reps) * 1e-9);
    return 0;
}

int main(int argc, char** argv) {
    int dev = 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    float* out;
    CK(hipMalloc(&out, 16));
    const int iters = argc > 1 ? atoi(argv[1]) : 200000;
    const int reps = argc > 2 ? atoi(argv[2]) : 5;
    for (int wps = 1; wps <= 2; ++wps) {   // waves per SIMD
        const int blocks = cus * wps;      // 256 threads = 4 waves = 1 per SIMD
        const double waves = (double)blocks * 4;
        char nm[96];
        snprintf(nm, sizeof nm, "mfma 16x16x4 f32, 4 acc, %d w/SIMD", wps);
        time_it(nm, [&] { hipLaunchKernelGGL(k_16x16x4<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f, 2.f); },
                waves * iters * 16.0 * 2048.0, reps);
        snprintf(nm, sizeof nm, "mfma 16x16x4 f32, 16 acc, %d w/SIMD", wps);
        time_it(nm, [&] { hipLaunchKernelGGL(k_16x16x4<16>, dim3(blocks), dim3(256), 0, 0, out, iters / 4, 1.f, 2.f); },
                waves * (iters / 4) * 64.0 * 2048.0, reps);
        snprintf(nm, sizeof nm, "mfma 32x32x2 f32, 4 acc, %d w/SIMD", wps);
        time_it(nm, [&] { hipLaunchKernelGGL(k_32x32x2<4>, dim3(blocks), dim3(256), 0, 0, out, iters / 2, 1.f, 2.f); },
                waves * (iters / 2) * 16.0 * 4096.0, reps);
    }
    for (int wps = 1; wps <= 2; ++wps) {
        const int blocks = cus * wps;
        const double waves = (double)blocks * 4;
        char nm[96];
        snprintf(nm, sizeof nm, "mfma 16x16x4 f32 RANDOM data, %d w/SIMD", wps);
        time_it(nm, [&] { hipLaunchKernelGGL(k_16x16x4_rand, dim3(blocks), dim3(256), 0, 0, out, iters / 4, 12345u); },
                waves * (iters / 4) * 64.0 * 2048.0, reps);
    }
    {
        const int blocks = cus * 8;
        const double threads = (double)blocks * 256;
        time_it("valu fma f32 chain (32 indep.)", [&] { hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); },
                threads * iters * 32.0 * 2.0, reps);
    }
    CK(hipFree(out));
    return 0;
}
