// A pure fp32 FMA loop on data that keeps changing, for a stated number of seconds: the reference point for what
// clock and what vector rate the device sustains when nothing but v_fma_f32 is issued (no LDS, no memory).
// Every lane runs x = fma(x, -1, b) on 16 independent registers (x toggles between x0 and b - x0 for ever: exact,
// bounded, every mantissa bit changes each time).  Prints the in-kernel shader clock (shader-clock counter against
// the constant 100 MHz counter) and the achieved TFLOP/s.
// Build: hipcc --offload-arch=gfx950 -O3 tools/fma_loop.hip -o exp_build/fma_loop ; run: exp_build/fma_loop SECONDS WAVES_PER_SIMD [MODE]
// MODE 0: the multiplier is the inline constant -1.0 (two VGPR reads per instruction); 1: it is a VGPR (three reads);
// 2: it is an SGPR (two VGPR reads and a scalar one); 3: v_fma_f64 on 16 independent doubles per lane, all three
// operands in VGPR pairs (the rate the fp64 pair pass issues against).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256) fma_loop(const float* seed, float* sink, long long* ticks, int iters, float multiplier) {
    float x[16], b[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        x[k] = seed[(threadIdx.x * 16 + k) & 4095];
        b[k] = seed[(threadIdx.x * 16 + k + 2048) & 4095] + 3.0f;
    }
    float mv = multiplier + (threadIdx.x == 100000 ? 1.0f : 0.0f);  // a per-lane register holding -1
    asm volatile("" : "+v"(mv));
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (MODE == 0) asm volatile("v_fma_f32 %0, %0, -1.0, %1" : "+v"(x[k]) : "v"(b[k]));
                else if (MODE == 1) asm volatile("v_fma_f32 %0, %0, %2, %1" : "+v"(x[k]) : "v"(b[k]), "v"(mv));
                else asm volatile("v_fma_f32 %0, %0, %2, %1" : "+v"(x[k]) : "v"(b[k]), "s"(multiplier));
            }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += x[k];
    if (s == 12345.678f) sink[0] = s;  // keeps the chain alive
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = w1 - w0; }
}

__global__ void __launch_bounds__(256) fma_loop_f64(const float* seed, float* sink, long long* ticks, int iters, double multiplier) {
    double x[16], b[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        x[k] = seed[(threadIdx.x * 16 + k) & 4095];
        b[k] = seed[(threadIdx.x * 16 + k + 2048) & 4095] + 3.0;
    }
    double mv = multiplier + (threadIdx.x == 100000 ? 1.0 : 0.0);
    asm volatile("" : "+v"(mv));
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_fma_f64 %0, %0, %2, %1" : "+v"(x[k]) : "v"(b[k]), "v"(mv));
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    double s = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += x[k];
    if (s == 12345.678) sink[0] = static_cast<float>(s);
    if (threadIdx.x == 0) { ticks[2 * blockIdx.x] = c1 - c0; ticks[2 * blockIdx.x + 1] = w1 - w0; }
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 5.0;
    const int waves_per_simd = argc > 2 ? atoi(argv[2]) : 3;
    const int mode = argc > 3 ? atoi(argv[3]) : 0;
    const int blocks = 256 * waves_per_simd, iters = 20000;  // 64 fma per iteration per lane
    std::vector<float> h(4096);
    srand(7);
    for (float& v : h) v = 0.5f + static_cast<float>(rand()) / RAND_MAX;
    float *d_seed, *d_sink;
    long long* d_ticks;
    CK(hipMalloc(&d_seed, h.size() * sizeof(float)));
    CK(hipMalloc(&d_sink, 64));
    CK(hipMalloc(&d_ticks, blocks * 2 * sizeof(long long)));
    CK(hipMemcpy(d_seed, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    const auto t0 = std::chrono::steady_clock::now();
    double kernel_ms = 0;
    long launches = 0;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<long long> t(blocks * 2);
    double mhz_last = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        CK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(fma_loop<0>, dim3(blocks), dim3(256), 0, 0, d_seed, d_sink, d_ticks, iters, -1.0f);
        else if (mode == 3) hipLaunchKernelGGL(fma_loop_f64, dim3(blocks), dim3(256), 0, 0, d_seed, d_sink, d_ticks, iters, -1.0);
        else if (mode == 1) hipLaunchKernelGGL(fma_loop<1>, dim3(blocks), dim3(256), 0, 0, d_seed, d_sink, d_ticks, iters, -1.0f);
        else hipLaunchKernelGGL(fma_loop<2>, dim3(blocks), dim3(256), 0, 0, d_seed, d_sink, d_ticks, iters, -1.0f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        kernel_ms += ms; ++launches;
    }
    CK(hipMemcpy(t.data(), d_ticks, t.size() * sizeof(long long), hipMemcpyDeviceToHost));
    double num = 0, den = 0;
    for (int k = 0; k < blocks; ++k) { num += t[2 * k]; den += t[2 * k + 1]; }
    mhz_last = 100.0 * num / den;
    const double flop = 2.0 * 64.0 * iters * 256.0 * blocks * launches;
    printf("fma_loop mode %d: %.1f s, %d waves/SIMD, %ld launches, in-kernel clock %.0f MHz (last launch), %.1f TFLOP/s over the kernels' time, %.2f cycles per FMA instruction per SIMD\n",
           mode, kernel_ms * 1e-3, waves_per_simd, launches, mhz_last, flop / (kernel_ms * 1e-3) / 1e12,
           (kernel_ms * 1e-3 * mhz_last * 1e6) / (64.0 * iters * waves_per_simd * launches));
    return 0;
}
