// Raw issue rate of the fp32 MFMA instructions on gfx950: independent / dependent chains, 1-4 waves per SIMD.
// Build: hipcc -O3 --offload-arch=gfx950 mfma_rate.hip -o mfma_rate   (GPU box only)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ __launch_bounds__(1024) void k16(float* out, int iters, float a, float b) {
    f32x4 acc[CH];
    for (int c = 0; c < CH; ++c) acc[c] = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) s += acc[c].x + acc[c].y + acc[c].z + acc[c].w;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CH>
__global__ __launch_bounds__(1024) void k32(float* out, int iters, float a, float b) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CH; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
static void run(const char* name, F launch, double flop_per_mfma, int ch, int threads, float* out) {
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)iters * 8 * ch * (threads / 64) / 4.0;       // one workgroup per CU
    const double tflops = 256.0 * (threads / 64) * iters * 8.0 * ch * flop_per_mfma / (ms * 1e-3) / 1e12;
    printf("%-10s chains=%d waves/SIMD=%d  %.3f ms  %.1f ns per MFMA per SIMD  %.1f TFLOP/s\n", name, ch, threads / 256, ms,
           ms * 1e6 / mfma_per_simd, tflops);
}
int main() {
    float* out; hipMalloc(&out, 256 * 1024 * sizeof(float));
    for (int threads : {256, 512, 1024}) {
        run("16x16x4", [&](int it) { hipLaunchKernelGGL(k16<1>, dim3(256), dim3(threads), 0, 0, out, it, 1.f, 2.f); }, 2048, 1, threads, out);
        run("16x16x4", [&](int it) { hipLaunchKernelGGL(k16<2>, dim3(256), dim3(threads), 0, 0, out, it, 1.f, 2.f); }, 2048, 2, threads, out);
        run("16x16x4", [&](int it) { hipLaunchKernelGGL(k16<4>, dim3(256), dim3(threads), 0, 0, out, it, 1.f, 2.f); }, 2048, 4, threads, out);
        run("32x32x2", [&](int it) { hipLaunchKernelGGL(k32<1>, dim3(256), dim3(threads), 0, 0, out, it, 1.f, 2.f); }, 4096, 1, threads, out);
        run("32x32x2", [&](int it) { hipLaunchKernelGGL(k32<2>, dim3(256), dim3(threads), 0, 0, out, it, 1.f, 2.f); }, 4096, 2, threads, out);
        run("32x32x2", [&](int it) { hipLaunchKernelGGL(k32<4>, dim3(256), dim3(threads), 0, 0, out, it, 1.f, 2.f); }, 4096, 4, threads, out);
    }
    hipFree(out);
    return 0;
}
