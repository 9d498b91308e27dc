// Issue rates of the fp32 vector instruction forms the PairHMM cell is made of, on gfx950, written as
// inline assembly with explicit registers so that the compiler can neither pack nor move anything:
// VOP2 (v_mul_f32_e32, v_fmac_f32_e32) against VOP3 (v_fma_f32), packed (v_pk_fma_f32, v_pk_mul_f32),
// with source registers in different VGPR banks (register number mod 4) or all in one bank.
// Every pattern is 32 independent instructions (8 destinations in rotation); a wave runs the pattern
// `iters` times.  Reported: nanoseconds and shader-clock cycles per wave-instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_issue.hip -o exp_build/microbench_issue
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

#define CLOBBERS "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
                 "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47", \
                 "v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63"

// 8 instructions with destinations D0..D7; X(d, k) expands to the instruction text for destination d
#define REP8(X) X(16,0) X(17,1) X(18,2) X(19,3) X(20,4) X(21,5) X(22,6) X(23,7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)
// packed: destinations are register pairs v[16:17] .. v[30:31]
#define REP8P(X) X(16,17) X(18,19) X(20,21) X(22,23) X(24,25) X(26,27) X(28,29) X(30,31)
#define REP32P(X) REP8P(X) REP8P(X) REP8P(X) REP8P(X)

// sources: v32.. ; bank = register number mod 4
#define P_MUL_DIFF(d,k)   "v_mul_f32_e32 v" #d ", v33, v34\n"          /* banks 1,2 */
#define P_MUL_SAME(d,k)   "v_mul_f32_e32 v" #d ", v32, v36\n"          /* banks 0,0 */
#define P_FMAC_DIFF(d,k)  "v_fmac_f32_e32 v" #d ", v33, v34\n"
#define P_FMAC_SAME(d,k)  "v_fmac_f32_e32 v" #d ", v32, v36\n"
#define P_FMA_DIFF(d,k)   "v_fma_f32 v" #d ", v33, v34, v35\n"          /* banks 1,2,3 */
#define P_FMA_TWO(d,k)    "v_fma_f32 v" #d ", v33, v37, v34\n"          /* banks 1,1,2 */
#define P_FMA_SAME(d,k)   "v_fma_f32 v" #d ", v32, v36, v40\n"          /* banks 0,0,0 */
#define P_FMA_ACC(d,k)    "v_fma_f32 v" #d ", v33, v34, v" #d "\n"      /* the fmac form in VOP3 encoding */
#define P_FMA_DSTSRC(d,k) "v_fma_f32 v" #d ", v" #d ", v34, v35\n"      /* d = d * b + c */
#define P_FMA_REUSE(d,k)  "v_fma_f32 v" #d ", v33, v33, v34\n"          /* two operands the same register */
#define P_PKFMA_DIFF(d,e) "v_pk_fma_f32 v[" #d ":" #e "], v[34:35], v[36:37], v[38:39]\n"
#define P_PKFMA_ACC(d,e)  "v_pk_fma_f32 v[" #d ":" #e "], v[34:35], v[36:37], v[" #d ":" #e "]\n"
#define P_PKMUL(d,e)      "v_pk_mul_f32 v[" #d ":" #e "], v[34:35], v[36:37]\n"
#define P_MOV(d,k)        "v_mov_b32_e32 v" #d ", v33\n"
#define P_DPP(d,k)        "v_mov_b32_dpp v" #d ", v33 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
// the cell's mix: mul, fmac, fmac, fma, fma per cell -> 2 : 2 : 1
#define P_CELL(d,k)       "v_mul_f32_e32 v" #d ", v33, v34\n v_fmac_f32_e32 v24, v35, v34\n v_fma_f32 v25, v33, v38, v35\n v_fmac_f32_e32 v26, v37, v34\n v_fma_f32 v27, v37, v38, v39\n"

#define KERNEL(NAME, BODY)                                                                       \
    __global__ void __launch_bounds__(256) NAME(long long* ticks, int iters) {                   \
        const long long c0 = clock64(), w0 = wall_clock64();                                     \
        for (int i = 0; i < iters; i++) asm volatile(BODY ::: CLOBBERS);                         \
        const long long c1 = clock64(), w1 = wall_clock64();                                     \
        if (threadIdx.x == 0 && blockIdx.x == 0) { ticks[0] = c1 - c0; ticks[1] = w1 - w0; }     \
    }

KERNEL(k_mul_diff, REP32(P_MUL_DIFF))
KERNEL(k_mul_same, REP32(P_MUL_SAME))
KERNEL(k_fmac_diff, REP32(P_FMAC_DIFF))
KERNEL(k_fmac_same, REP32(P_FMAC_SAME))
KERNEL(k_fma_diff, REP32(P_FMA_DIFF))
KERNEL(k_fma_two, REP32(P_FMA_TWO))
KERNEL(k_fma_same, REP32(P_FMA_SAME))
KERNEL(k_fma_acc, REP32(P_FMA_ACC))
KERNEL(k_fma_dstsrc, REP32(P_FMA_DSTSRC))
KERNEL(k_fma_reuse, REP32(P_FMA_REUSE))
KERNEL(k_pkfma_diff, REP32P(P_PKFMA_DIFF))
KERNEL(k_pkfma_acc, REP32P(P_PKFMA_ACC))
KERNEL(k_pkmul, REP32P(P_PKMUL))
KERNEL(k_mov, REP32(P_MOV))
KERNEL(k_dpp, REP32(P_DPP))
KERNEL(k_cell, REP8(P_CELL))

template <typename K>
static int run(const char* name, K kernel, int instr_per_iter, long long* d_ticks) {
    for (int waves_per_simd : {1, 2, 3, 4, 8}) {
        const int blocks = 256 * waves_per_simd, iters = 40000;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_ticks, 200);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, d_ticks, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        long long t[2];
        CK(hipMemcpy(t, d_ticks, sizeof t, hipMemcpyDeviceToHost));
        const double mhz = 100.0 * t[0] / t[1];                                   // shader clock against the 100 MHz counter
        const double per_simd = (double)waves_per_simd * iters * instr_per_iter;  // wave-instructions one SIMD issued
        const double ns = ms * 1e6 / per_simd;
        printf("%-34s waves/SIMD=%d  %7.3f ms  %5.0f MHz  %5.2f ns/instr  %5.2f cycles/instr\n", name, waves_per_simd, ms, mhz, ns, ns * mhz * 1e-3);
        CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    }
    return 0;
}

int main() {
    long long* d_ticks;
    CK(hipMalloc(&d_ticks, 64));
    if (run("v_mul_f32_e32, banks differ", k_mul_diff, 32, d_ticks)) return 1;
    if (run("v_mul_f32_e32, one bank", k_mul_same, 32, d_ticks)) return 1;
    if (run("v_fmac_f32_e32, banks differ", k_fmac_diff, 32, d_ticks)) return 1;
    if (run("v_fmac_f32_e32, one bank", k_fmac_same, 32, d_ticks)) return 1;
    if (run("v_fma_f32, banks 1,2,3", k_fma_diff, 32, d_ticks)) return 1;
    if (run("v_fma_f32, banks 1,1,2", k_fma_two, 32, d_ticks)) return 1;
    if (run("v_fma_f32, banks 0,0,0", k_fma_same, 32, d_ticks)) return 1;
    if (run("v_fma_f32 d = a*b + d", k_fma_acc, 32, d_ticks)) return 1;
    if (run("v_fma_f32 d = d*b + c", k_fma_dstsrc, 32, d_ticks)) return 1;
    if (run("v_fma_f32 a, a, b", k_fma_reuse, 32, d_ticks)) return 1;
    if (run("v_pk_fma_f32, distinct pairs", k_pkfma_diff, 32, d_ticks)) return 1;
    if (run("v_pk_fma_f32 d = a*b + d", k_pkfma_acc, 32, d_ticks)) return 1;
    if (run("v_pk_mul_f32", k_pkmul, 32, d_ticks)) return 1;
    if (run("v_mov_b32", k_mov, 32, d_ticks)) return 1;
    if (run("v_mov_b32_dpp wave_shr:1", k_dpp, 32, d_ticks)) return 1;
    if (run("cell mix mul,fmac,fma,fmac,fma", k_cell, 40, d_ticks)) return 1;
    printf("done\n");
    return 0;
}
