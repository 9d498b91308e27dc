// C++ mirror of the reference's Smith-Waterman aligner interface over the C-ABI (gatk_smithwaterman.h):
//   new SWPairwiseAlignment(seq1, seq2, parameters, strategy)   smithwaterman/SWPairwiseAlignment.java:149-153
//   getCigar() / getAlignmentStart2wrt1()                        smithwaterman/SmithWaterman.java:44,55
// Header-only; link libgatk_pairhmm_hip.so.
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "gatk_smithwaterman.h"

namespace gatk {

struct SWParameters { int32_t w_match, w_mismatch, w_open, w_extend; };
constexpr SWParameters ORIGINAL_DEFAULT{3, -1, -4, -3};     // SWParameterSet.java:38
constexpr SWParameters STANDARD_NGS{25, -50, -110, -6};     // SWParameterSet.java:43

enum class OverhangStrategy : int32_t { SOFTCLIP = GATK_SW_SOFTCLIP, INDEL = GATK_SW_INDEL, LEADING_INDEL = GATK_SW_LEADING_INDEL, IGNORE = GATK_SW_IGNORE };

class SWPairwiseAlignment {
public:
    SWPairwiseAlignment(const std::vector<uint8_t>& seq1, const std::vector<uint8_t>& seq2, SWParameters parameters = ORIGINAL_DEFAULT,
                        OverhangStrategy strategy = OverhangStrategy::SOFTCLIP, int device = 0) {
        const uint8_t* r = seq1.data();
        const uint8_t* a = seq2.data();
        const int32_t n = static_cast<int32_t>(seq1.size()), m = static_cast<int32_t>(seq2.size());
        const gatk_sw_parameters p{parameters.w_match, parameters.w_mismatch, parameters.w_open, parameters.w_extend};
        constexpr int32_t cap = 256;
        int32_t ops[cap], lens[cap], count = 0;
        const int rc = gatk_sw_align(device, 1, &r, &n, &a, &m, &p, static_cast<int32_t>(strategy), cap, ops, lens, &count, &offset_);
        if (rc == GATK_SW_ILLEGAL_ARGUMENT) throw std::invalid_argument(gatk_sw_last_error());
        if (rc != GATK_SW_OK) throw std::runtime_error(gatk_sw_last_error());
        for (int32_t k = 0; k < count; ++k) cigar_ += std::to_string(lens[k]) + "MIDS"[ops[k]];
    }
    const std::string& getCigar() const { return cigar_; }
    int getAlignmentStart2wrt1() const { return offset_; }

private:
    std::string cigar_;
    int32_t offset_ = 0;
};

}  // namespace gatk
