// Shared by the driver and the per-instruction-set builds of the vectorised CPU baseline.
#pragma once
#include <cstdint>

constexpr int kSimdLanes = 16;  // haplotypes per block: one AVX-512 vector (wider GCC vector types are scalarised)

enum { matchToMatch = 0, indelToMatch = 1, matchToInsertion = 2, insertionToInsertion = 3, matchToDeletion = 4, deletionToDeletion = 5 };  // pairhmm/PairHMMModel.java:48-76

struct SimdReadConsts {  // rows 1..R
    const float* t;      // six transitions per row
    const float* pm;     // prior of a match
    const float* pmm_n;  // prior of a mismatch; the match prior in rows whose read base is 'N'
    const float* xs;     // read bases as floats
};

// One read against kSimdLanes haplotypes (hapT[j * kSimdLanes + lane], hlen[lane] = 0 for an unused
// lane); M, I, D: R + 1 vectors of 4 * kSimdLanes bytes each; acc_out[lane]: scaled sum of the last row.
typedef void (*simd_block_fn)(int R, const SimdReadConsts* rc, const uint8_t* hapT, const int32_t* hlen, int Hmax, void* M, void* I, void* D, float* acc_out);
extern "C" {
void simd_block_avx512(int, const SimdReadConsts*, const uint8_t*, const int32_t*, int, void*, void*, void*, float*);
void simd_block_avx2(int, const SimdReadConsts*, const uint8_t*, const int32_t*, int, void*, void*, void*, float*);
void simd_block_generic(int, const SimdReadConsts*, const uint8_t*, const int32_t*, int, void*, void*, void*, float*);
}
