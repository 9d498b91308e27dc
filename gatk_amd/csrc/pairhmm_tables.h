// Phred look-up tables shared by host planning and the device kernels.
#pragma once
#include <vector>

namespace pairhmm_hip {

constexpr int kMaxQual = 254;   // QualityUtils.java:63: the error-probability cache has 255 entries
constexpr int kGopLimit = 128;  // insertion/deletion/GCP bytes >= 128 are negative Java bytes: rejected

struct Tables {
    double err_prob[256];                // 10^(-q/10); entry 255 is a placeholder
    std::vector<double> match_to_match;  // [ins * 128 + del], linear scale
};

const Tables& tables();

}  // namespace pairhmm_hip
