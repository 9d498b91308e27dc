// Micro-measurements that size the PairHMM kernel design on gfx950:
//  (1) does v_mov_b32_dpp wave_shr:1 move lane l-1 -> lane l across all 64 lanes?
//  (2) issue rate of v_fma_f32 vs v_pk_fma_f32 vs v_fma_f64 at 1..8 waves per SIMD
//  (3) issue rate of a dependent v_mov_dpp + fma chain (the anti-diagonal hand-off)
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o tools/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__device__ inline float shr1(float v, float old) {
    int r = __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false);
    return __builtin_bit_cast(float, r);
}

template <int CTRL>
__device__ inline float dpp_mov(float v) {
    int r = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(float, r);
}

__global__ void k_dpp_check(float* out) {
    float v = (float)threadIdx.x;
    out[threadIdx.x] = shr1(v, -1.0f);
}

typedef float float2_ __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k_rate(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    if (MODE == 0) {  // 8 independent v_fma_f32 chains
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
                x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
            }
        }
    } else if (MODE == 1) {  // 4 independent v_pk_fma_f32 chains (8 floats)
        float2_ p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7}, av = {a, a}, bv = {b, b};
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(av), "v"(bv));
            }
        }
        x0 = p0.x + p0.y; x1 = p1.x + p1.y; x2 = p2.x + p2.y; x3 = p3.x + p3.y;
    } else if (MODE == 2) {  // 8 independent v_fma_f64 chains
        double d0 = x0, d1 = x1, d2 = x2, d3 = x3, d4 = x4, d5 = x5, d6 = x6, d7 = x7, da = a, db = b;
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                d0 = __builtin_fma(d0, da, db); d1 = __builtin_fma(d1, da, db); d2 = __builtin_fma(d2, da, db); d3 = __builtin_fma(d3, da, db);
                d4 = __builtin_fma(d4, da, db); d5 = __builtin_fma(d5, da, db); d6 = __builtin_fma(d6, da, db); d7 = __builtin_fma(d7, da, db);
            }
        }
        x0 = (float)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    } else if (MODE == 3) {  // per "instruction": dpp shift then a dependent fma, 4 independent chains
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x0 = __builtin_fmaf(shr1(x0, 0.f), a, b); x1 = __builtin_fmaf(shr1(x1, 0.f), a, b);
                x2 = __builtin_fmaf(shr1(x2, 0.f), a, b); x3 = __builtin_fmaf(shr1(x3, 0.f), a, b);
            }
        }
    } else if (MODE == 4) {  // one dependent chain of v_fma_f32 (latency)
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 64; u++) x0 = __builtin_fmaf(x0, a, b);
        }
    } else if (MODE == 5) {  // 8 independent chains of v_mul + v_cndmask (select-heavy mix)
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x0 = (x1 > b ? a : b) * x0; x1 = (x2 > b ? a : b) * x1; x2 = (x3 > b ? a : b) * x2; x3 = (x0 > b ? a : b) * x3;
            }
        }
    }
    else if (MODE == 8) {  // row_shr:1 (within rows of 16 lanes) then dependent fma, 4 chains, VGPR operands
        float y0 = x4, y1 = x5, z0 = x6, z1 = x7;
        asm volatile("" : "+v"(y0), "+v"(y1), "+v"(z0), "+v"(z1));
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x0 = __builtin_fmaf(dpp_mov<0x111>(x0), y0, z0); x1 = __builtin_fmaf(dpp_mov<0x111>(x1), y1, z1);
                x2 = __builtin_fmaf(dpp_mov<0x111>(x2), y0, z1); x3 = __builtin_fmaf(dpp_mov<0x111>(x3), y1, z0);
            }
        }
    } else if (MODE == 9) {  // wave_shr:1 then dependent fma, 4 chains, VGPR operands
        float y0 = x4, y1 = x5, z0 = x6, z1 = x7;
        asm volatile("" : "+v"(y0), "+v"(y1), "+v"(z0), "+v"(z1));
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x0 = __builtin_fmaf(dpp_mov<0x138>(x0), y0, z0); x1 = __builtin_fmaf(dpp_mov<0x138>(x1), y1, z1);
                x2 = __builtin_fmaf(dpp_mov<0x138>(x2), y0, z1); x3 = __builtin_fmaf(dpp_mov<0x138>(x3), y1, z0);
            }
        }
    } else if (MODE == 10) {  // no shift: the same 4 dependent fma chains (baseline for 8 and 9)
        float y0 = x4, y1 = x5, z0 = x6, z1 = x7;
        asm volatile("" : "+v"(y0), "+v"(y1), "+v"(z0), "+v"(z1));
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x0 = __builtin_fmaf(x0, y0, z0); x1 = __builtin_fmaf(x1, y1, z1);
                x2 = __builtin_fmaf(x2, y0, z1); x3 = __builtin_fmaf(x3, y1, z0);
            }
        }
    } else if (MODE == 11) {  // ds_bpermute shift (LDS crossbar, no VALU) then dependent fma, 4 chains
        float y0 = x4, y1 = x5, z0 = x6, z1 = x7;
        const int src = (((int)threadIdx.x & 63) - 1) * 4;
        asm volatile("" : "+v"(y0), "+v"(y1), "+v"(z0), "+v"(z1));
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x0 = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, x0))), y0, z0);
                x1 = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, x1))), y1, z1);
                x2 = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, x2))), y0, z1);
                x3 = __builtin_fmaf(__builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, x3))), y1, z0);
            }
        }
    }
    else if (MODE == 12) {  // packed cell pattern: the 5-instruction cell on float2 values, 4 independent cell pairs
        typedef float v2 __attribute__((ext_vector_type(2)));
        v2 p0 = {x0, x1}, p1 = {x1, x2}, p2 = {x2, x3}, p3 = {x3, x4};
        v2 c0 = {a + x4 * 1e-6f, a + x5 * 1e-6f}, c1 = {b + x5 * 1e-6f, b + x6 * 1e-6f}, c2 = {a * 0.5f + x6 * 1e-6f, a * 0.5f}, c3 = {b * 0.5f + x7 * 1e-6f, b * 0.5f};
        v2 E0 = 1, E1 = 1, E2 = 1, E3 = 1, F0 = 1, F1 = 1, F2 = 1, F3 = 1, S0 = 1, S1 = 1, S2 = 1, S3 = 1, M0 = 1, M1 = 1, M2 = 1, M3 = 1;
        asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                v2 m0 = p0 * S0, m1 = p1 * S1, m2 = p2 * S2, m3 = p3 * S3;
                E0 = __builtin_elementwise_fma(E0, c0, M0); E1 = __builtin_elementwise_fma(E1, c1, M1); E2 = __builtin_elementwise_fma(E2, c2, M2); E3 = __builtin_elementwise_fma(E3, c3, M3);
                F0 = __builtin_elementwise_fma(F0, c1, m0); F1 = __builtin_elementwise_fma(F1, c2, m1); F2 = __builtin_elementwise_fma(F2, c3, m2); F3 = __builtin_elementwise_fma(F3, c0, m3);
                S0 = __builtin_elementwise_fma(E0, c2, __builtin_elementwise_fma(F0, c3, m0)); S1 = __builtin_elementwise_fma(E1, c3, __builtin_elementwise_fma(F1, c0, m1));
                S2 = __builtin_elementwise_fma(E2, c0, __builtin_elementwise_fma(F2, c1, m2)); S3 = __builtin_elementwise_fma(E3, c1, __builtin_elementwise_fma(F3, c2, m3));
                M0 = m0; M1 = m1; M2 = m2; M3 = m3;
            }
        }
        x0 = S0.x + E0.y + F0.x; x1 = S1.x + E1.y + F1.x; x2 = S2.y + E2.x + F2.y; x3 = S3.x + E3.x + F3.y;
    }
    else if (MODE == 6) {  // 8 independent v_fma_f32 chains, all three operands in VGPRs
        float y0 = x0 * 0.5f, y1 = x1 * 0.5f, y2 = x2 * 0.5f, y3 = x3 * 0.5f, z0 = a + x0, z1 = a + x1, z2 = b + x2, z3 = b + x3;
        asm volatile("" : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(z0), "+v"(z1), "+v"(z2), "+v"(z3));
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                x0 = __builtin_fmaf(x0, y0, z0); x1 = __builtin_fmaf(x1, y1, z1); x2 = __builtin_fmaf(x2, y2, z2); x3 = __builtin_fmaf(x3, y3, z3);
                x4 = __builtin_fmaf(x4, y0, z1); x5 = __builtin_fmaf(x5, y1, z2); x6 = __builtin_fmaf(x6, y2, z3); x7 = __builtin_fmaf(x7, y3, z0);
            }
        }
    } else if (MODE == 7) {  // the cell pattern: M = p*s; E = fma(E,c,Mo); F = fma(F,a,M); S = fma(E,c2,fma(F,b,M)), 4 independent cells
        float p0 = x0, p1 = x1, p2 = x2, p3 = x3, c0 = a + x4 * 1e-6f, c1 = b + x5 * 1e-6f, c2 = a * 0.5f + x6 * 1e-6f, c3 = b * 0.5f + x7 * 1e-6f;
        float E0 = 1, E1 = 1, E2 = 1, E3 = 1, F0 = 1, F1 = 1, F2 = 1, F3 = 1, S0 = 1, S1 = 1, S2 = 1, S3 = 1, M0 = 1, M1 = 1, M2 = 1, M3 = 1;
        asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3));
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                float m0 = p0 * S0, m1 = p1 * S1, m2 = p2 * S2, m3 = p3 * S3;
                E0 = __builtin_fmaf(E0, c0, M0); E1 = __builtin_fmaf(E1, c1, M1); E2 = __builtin_fmaf(E2, c2, M2); E3 = __builtin_fmaf(E3, c3, M3);
                F0 = __builtin_fmaf(F0, c1, m0); F1 = __builtin_fmaf(F1, c2, m1); F2 = __builtin_fmaf(F2, c3, m2); F3 = __builtin_fmaf(F3, c0, m3);
                S0 = __builtin_fmaf(E0, c2, __builtin_fmaf(F0, c3, m0)); S1 = __builtin_fmaf(E1, c3, __builtin_fmaf(F1, c0, m1));
                S2 = __builtin_fmaf(E2, c0, __builtin_fmaf(F2, c1, m2)); S3 = __builtin_fmaf(E3, c1, __builtin_fmaf(F3, c2, m3));
                M0 = m0; M1 = m1; M2 = m2; M3 = m3;
            }
        }
        x0 = S0 + E0 + F0; x1 = S1 + E1 + F1; x2 = S2 + E2 + F2; x3 = S3 + E3 + F3;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

// Shader clock under a full fp32 load: s_memtime ticks against the constant 100 MHz counter.
__global__ void __launch_bounds__(256) k_clock(float* out, long long* ticks, int iters, float a, float b) {
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (threadIdx.x == 0 && (blockIdx.x % 97) == 0) {
        ticks[2 * (blockIdx.x / 97)] = c1 - c0;
        ticks[2 * (blockIdx.x / 97) + 1] = w1 - w0;
    }
}

template <int MODE>
static int run_rate(const char* name, double instr_per_iter, double flop_per_instr, float* d_out) {
    int dev_cus = 256;
    for (int wavesPerSimd = 1; wavesPerSimd <= 8; wavesPerSimd *= 2) {
        int blocks = dev_cus * wavesPerSimd;  // 256 threads = 4 waves = 1 wave per SIMD per block
        int iters = 20000;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 100, 0.999f, 0.001f);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_rate<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.999f, 0.001f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        double waves = (double)blocks * 4;
        double winstr = waves * iters * instr_per_iter;           // wave-instructions
        double per_simd_per_s = winstr / (1024.0) / (ms * 1e-3);  // wave-instr per SIMD per second
        double cyc_per_instr = 2.4e9 / per_simd_per_s;            // at nominal 2.4 GHz
        printf("%-28s waves/SIMD=%d  %.3f ms  %.2f cyc/wave-instr(@2.4GHz)  %.1f TFLOP/s\n", name, wavesPerSimd, ms,
               cyc_per_instr, winstr * 64 * flop_per_instr / (ms * 1e-3) / 1e12);
    }
    return 0;
}

int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz  LDS/block=%zu\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.sharedMemPerBlock);
    float* d_out; CK(hipMalloc(&d_out, sizeof(float) * 256 * 8 * 256));
    hipLaunchKernelGGL(k_dpp_check, dim3(1), dim3(64), 0, 0, d_out);
    std::vector<float> h(64); CK(hipMemcpy(h.data(), d_out, 64 * 4, hipMemcpyDeviceToHost));
    int ok = (h[0] == -1.0f);
    for (int i = 1; i < 64; i++) ok &= (h[i] == (float)(i - 1));
    printf("dpp wave_shr:1 %s  (lane0=%g lane1=%g lane16=%g lane32=%g lane63=%g)\n", ok ? "OK" : "WRONG", h[0], h[1], h[16], h[32], h[63]);
    {
        long long* d_ticks; CK(hipMalloc(&d_ticks, sizeof(long long) * 64));
        for (int wavesPerSimd = 1; wavesPerSimd <= 4; wavesPerSimd *= 2) {
            const int blocks = 256 * wavesPerSimd, iters = 200000;
            hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(256), 0, 0, d_out, d_ticks, iters, 0.999f, 0.001f);
            long long t[8]; CK(hipMemcpy(t, d_ticks, sizeof t, hipMemcpyDeviceToHost));
            for (int k = 0; k < 2; k++)
                printf("clock under fp32 load, waves/SIMD=%d: s_memtime %lld ticks, 100MHz counter %lld ticks -> %.0f MHz if s_memtime counts shader cycles; %.2f ticks per wave-instr\n",
                       wavesPerSimd, t[2 * k], t[2 * k + 1], 100.0 * t[2 * k] / t[2 * k + 1], (double)t[2 * k] / (iters * 64.0 * wavesPerSimd));
        }
    }
    if (run_rate<0>("v_fma_f32 x8 indep", 64, 2, d_out)) return 1;
    if (run_rate<1>("v_pk_fma_f32 x4 indep", 32, 4, d_out)) return 1;
    if (run_rate<2>("v_fma_f64 x8 indep", 64, 2, d_out)) return 1;
    if (run_rate<3>("dpp_shr1+fma x4 indep", 64, 1, d_out)) return 1;
    if (run_rate<4>("v_fma_f32 dependent chain", 64, 2, d_out)) return 1;
    if (run_rate<5>("cmp+cndmask+mul x4", 96, 1.0 / 3, d_out)) return 1;
    if (run_rate<12>("packed cell pattern 5 pk x4", 80, 3.6, d_out)) return 1;
    if (run_rate<10>("4 chains fma (no shift)", 32, 2, d_out)) return 1;
    if (run_rate<8>("4 chains row_shr:1 + fma", 32, 2, d_out)) return 1;
    if (run_rate<9>("4 chains wave_shr:1 + fma", 32, 2, d_out)) return 1;
    if (run_rate<11>("4 chains ds_bpermute + fma", 32, 2, d_out)) return 1;
    if (run_rate<6>("v_fma_f32 x8 indep 3 VGPR ops", 64, 2, d_out)) return 1;
    if (run_rate<7>("cell pattern 5 instr x4 cells", 80, 1.8, d_out)) return 1;
    printf("done\n");
    return 0;
}
