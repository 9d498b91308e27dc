// C++ mirror of the reference's PairHMM plugin interface over the C-ABI (gatk_pairhmm.h).
//
// The reference's host side is Java (pairhmm/PairHMM.java); where no JDK is available this header
// is the host-side stand-in with the same method names, argument meaning and error behaviour:
//   initialize(readMaxLength, haplotypeMaxLength)            PairHMM.java:104
//   initialize(haplotypes, perSampleReadList, readMax, hapMax) PairHMM.java:137 (the region-level hook)
//   computeLikelihoods(matrix, reads, gcp)                   PairHMM.java:190
//   getLikelihoodArray()                                     PairHMM.java:360
//   finalizeRegion() / close()                               PairHMM.java:124 / :364
//   batchAdd(...) / batchGetResult()                         BatchPairHMM.java:33-40
//   submit(...) / fetch(ticket)                              computeLikelihoods in two halves (no counterpart in the reference)
// Java exceptions become C++ exceptions of the same name.  Header-only; link libgatk_pairhmm_hip.so.
#pragma once

#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "gatk_pairhmm.h"

namespace gatk {

struct IllegalArgumentException : std::invalid_argument { using std::invalid_argument::invalid_argument; };
struct IllegalStateException : std::logic_error { using std::logic_error::logic_error; };
struct ArrayIndexOutOfBoundsException : std::out_of_range { using std::out_of_range::out_of_range; };
struct DeviceException : std::runtime_error { using std::runtime_error::runtime_error; };

using Bytes = std::vector<uint8_t>;

// What the PairHMM reads from a GATKSAMRecord (PairHMM.java:210-213).
struct Read {
    Bytes bases, quals, insertionQuals, deletionQuals;  // equal lengths
};

// ReadLikelihoods.Matrix (genotyper/ReadLikelihoods.java:652-756): values[allele][read].
class LikelihoodMatrix {
public:
    LikelihoodMatrix(std::vector<Bytes> alleles, size_t readCount)
        : alleles_(std::move(alleles)), reads_(readCount), values_(alleles_.size() * readCount, 0.0) {}
    const std::vector<Bytes>& alleles() const { return alleles_; }
    size_t alleleCount() const { return alleles_.size(); }
    size_t readCount() const { return reads_; }
    void set(size_t alleleIndex, size_t readIndex, double v) { values_[alleleIndex * reads_ + readIndex] = v; }
    double get(size_t alleleIndex, size_t readIndex) const { return values_[alleleIndex * reads_ + readIndex]; }

private:
    std::vector<Bytes> alleles_;
    size_t reads_;
    std::vector<double> values_;
};

class HipLoglessPairHMM {
public:
    explicit HipLoglessPairHMM(int device = 0) { check(gatk_pairhmm_create(device, &h_)); }
    // One instance over several GPUs of the node (an empty list: every visible device): gatk_pairhmm_create_multi.
    explicit HipLoglessPairHMM(const std::vector<int32_t>& devices) {
        check(gatk_pairhmm_create_multi(devices.empty() ? nullptr : devices.data(), static_cast<int32_t>(devices.size()), &h_));
    }
    // PairHMM.doNotUseTristateCorrection(), PairHMM.java:83-84
    void doNotUseTristateCorrection() { check(gatk_pairhmm_set_option(h_, "do_not_use_tristate_correction", 1.0)); }
    ~HipLoglessPairHMM() { close(); }
    HipLoglessPairHMM(const HipLoglessPairHMM&) = delete;
    HipLoglessPairHMM& operator=(const HipLoglessPairHMM&) = delete;

    void initialize(int readMaxLength, int haplotypeMaxLength) { check(gatk_pairhmm_initialize(h_, readMaxLength, haplotypeMaxLength)); }

    // PairHMM.initialize(haplotypes, perSampleReadList, readMaxLength, haplotypeMaxLength), PairHMM.java:137: every
    // sample's reads and the region's haplotypes go to the device in one pass; returns without waiting for it.  The
    // computeLikelihoods calls that follow (one per sample, PairHMM.java:190) find their likelihoods there.
    void initialize(const std::vector<Bytes>& haplotypes, const std::map<std::string, std::vector<Read>>& perSampleReadList,
                    int readMaxLength, int haplotypeMaxLength) {
        std::vector<Flat> flats;
        flats.reserve(perSampleReadList.size());
        std::vector<gatk_pairhmm_batch> batches;
        for (const auto& sample : perSampleReadList) {
            flats.emplace_back(sample.second, nullptr, haplotypes);  // no GCP yet: it comes with computeLikelihoods
            batches.push_back(flats.back().batch());
        }
        check(gatk_pairhmm_initialize_region(h_, static_cast<int32_t>(batches.size()), batches.data(), readMaxLength, haplotypeMaxLength));
    }

    // gcp: per-read gap-continuation penalties (the Map<GATKSAMRecord, byte[]> of the reference),
    // indexed like `reads`.
    void computeLikelihoods(LikelihoodMatrix& likelihoods, const std::vector<Read>& reads, const std::vector<Bytes>& gcp) {
        if (reads.empty()) return;  // PairHMM.java:193
        const auto& alleles = likelihoods.alleles();
        const Flat f(reads, &gcp, alleles);
        const gatk_pairhmm_batch b = f.batch();
        mLikelihoodArray_.assign(reads.size() * alleles.size(), 0.0);
        check(gatk_pairhmm_compute_likelihoods(h_, &b, mLikelihoodArray_.data()));
        size_t idx = 0;
        for (size_t r = 0; r < reads.size(); ++r)
            for (size_t a = 0; a < alleles.size(); ++a) likelihoods.set(a, r, mLikelihoodArray_[idx++]);  // PairHMM.java:224-225
    }

    // computeLikelihoods in two halves: submit returns a ticket at once, fetch(ticket) fills the matrix.  The reads and
    // the GCP arrays may go away after submit; the matrix must live until fetch.
    struct Ticket {
        int64_t id = 0;
        LikelihoodMatrix* matrix = nullptr;
        std::vector<double> values;  // read-major, written by the engine
    };
    std::unique_ptr<Ticket> submit(LikelihoodMatrix& likelihoods, const std::vector<Read>& reads, const std::vector<Bytes>& gcp) {
        std::unique_ptr<Ticket> t(new Ticket());
        t->matrix = &likelihoods;
        t->values.assign(reads.size() * likelihoods.alleleCount(), 0.0);
        const Flat f(reads, &gcp, likelihoods.alleles());
        const gatk_pairhmm_batch b = f.batch();
        double* out = t->values.data();
        check(gatk_pairhmm_submit(h_, 1, &b, &out, &t->id));
        return t;
    }
    void fetch(Ticket& t) {
        check(gatk_pairhmm_fetch(h_, t.id));
        const size_t alleles = t.matrix->alleleCount(), reads = t.matrix->readCount();
        size_t idx = 0;
        for (size_t r = 0; r < reads; ++r)
            for (size_t a = 0; a < alleles; ++a) t.matrix->set(a, r, t.values[idx++]);
    }

    const std::vector<double>& getLikelihoodArray() const { return mLikelihoodArray_; }

    void batchAdd(const std::vector<Bytes>& haplotypes, const Bytes& readBases, const Bytes& readQuals, const Bytes& insertionGOP,
                  const Bytes& deletionGOP, const Bytes& overallGCP) {
        std::vector<const uint8_t*> hp;
        std::vector<int32_t> hl;
        for (const Bytes& h : haplotypes) { hp.push_back(h.data()); hl.push_back(static_cast<int32_t>(h.size())); }
        check(gatk_pairhmm_batch_add(h_, static_cast<int32_t>(hp.size()), hp.data(), hl.data(), readBases.data(), readQuals.data(),
                                     insertionGOP.data(), deletionGOP.data(), overallGCP.data(), static_cast<int32_t>(readBases.size())));
    }

    std::vector<double> batchGetResult() {
        const double* p = nullptr;
        int64_t n = 0;
        check(gatk_pairhmm_batch_get_result(h_, &p, &n));
        return std::vector<double>(p, p + n);
    }

    void finalizeRegion() { check(gatk_pairhmm_finalize_region(h_)); }

    void close() {
        if (h_) gatk_pairhmm_close(h_);
        h_ = nullptr;
    }

    gatk_pairhmm* handle() const { return h_; }

    // The concatenated arrays of one computeLikelihoods call (gatk_pairhmm_batch) with the checks of PairHMM.java:276-279.
    struct Flat {
        Bytes rb, rq, ri, rd, rg, hb;
        std::vector<int32_t> ro{0}, ho{0};
        bool has_gcp = false;
        Flat(const std::vector<Read>& reads, const std::vector<Bytes>* gcp, const std::vector<Bytes>& alleles) : has_gcp(gcp != nullptr) {
            for (size_t r = 0; r < reads.size(); ++r) {
                const Read& x = reads[r];
                const size_t n = x.bases.size();
                if (x.quals.size() != n) throw IllegalArgumentException("Read bases and read quals aren't the same size");
                if (x.insertionQuals.size() != n) throw IllegalArgumentException("Read bases and read insertion quals aren't the same size");
                if (x.deletionQuals.size() != n) throw IllegalArgumentException("Read bases and read deletion quals aren't the same size");
                if (gcp && gcp->at(r).size() != n) throw IllegalArgumentException("Read bases and overall GCP aren't the same size");
                rb.insert(rb.end(), x.bases.begin(), x.bases.end());
                rq.insert(rq.end(), x.quals.begin(), x.quals.end());
                ri.insert(ri.end(), x.insertionQuals.begin(), x.insertionQuals.end());
                rd.insert(rd.end(), x.deletionQuals.begin(), x.deletionQuals.end());
                if (gcp) rg.insert(rg.end(), (*gcp)[r].begin(), (*gcp)[r].end());
                ro.push_back(static_cast<int32_t>(rb.size()));
            }
            for (const Bytes& a : alleles) {
                hb.insert(hb.end(), a.begin(), a.end());
                ho.push_back(static_cast<int32_t>(hb.size()));
            }
            for (Bytes* v : {&rb, &rq, &ri, &rd, &rg, &hb}) v->push_back(0);  // data pointers stay non-null
        }
        gatk_pairhmm_batch batch() const {
            return gatk_pairhmm_batch{static_cast<int32_t>(ro.size()) - 1, static_cast<int32_t>(ho.size()) - 1, rb.data(), rq.data(), ri.data(),
                                      rd.data(), has_gcp ? rg.data() : nullptr, ro.data(), hb.data(), ho.data(), nullptr};
        }
    };

    static void check(int rc) {
        if (rc == GATK_PAIRHMM_OK) return;
        const std::string msg = gatk_pairhmm_last_error();
        switch (rc) {
            case GATK_PAIRHMM_ILLEGAL_ARGUMENT: throw IllegalArgumentException(msg);
            case GATK_PAIRHMM_ILLEGAL_STATE: throw IllegalStateException(msg);
            case GATK_PAIRHMM_INDEX_OUT_OF_BOUNDS: throw ArrayIndexOutOfBoundsException(msg);
            default: throw DeviceException(msg);
        }
    }
private:
    gatk_pairhmm* h_ = nullptr;
    std::vector<double> mLikelihoodArray_;
};

// The likelihoods of one sample's reads against a region's alleles, kept on the device between the
// PairHMM and the steps right behind it: the part of genotyper/ReadLikelihoods.java that works on the
// values (normalizeLikelihoods :353, filterPoorlyModeledReads :1059, marginalize :911), for one region.
class ResidentReadLikelihoods {
public:
    // Scores reads against alleles (PairHMM.computeLikelihoods) and keeps the result on the device.
    ResidentReadLikelihoods(HipLoglessPairHMM& hmm, const std::vector<Bytes>& alleles, const std::vector<Read>& reads,
                            const std::vector<Bytes>& gcp, int referenceAlleleIndex = -1)
        : hmm_(hmm), alleles_(alleles.size()), reads_(reads.size()), reference_(referenceAlleleIndex) {
        Bytes rb, rq, ri, rd, rg, hb;
        std::vector<int32_t> ro{0}, ho{0};
        for (size_t r = 0; r < reads.size(); ++r) {
            const Read& x = reads[r];
            rb.insert(rb.end(), x.bases.begin(), x.bases.end());
            rq.insert(rq.end(), x.quals.begin(), x.quals.end());
            ri.insert(ri.end(), x.insertionQuals.begin(), x.insertionQuals.end());
            rd.insert(rd.end(), x.deletionQuals.begin(), x.deletionQuals.end());
            rg.insert(rg.end(), gcp.at(r).begin(), gcp[r].end());
            ro.push_back(static_cast<int32_t>(rb.size()));
        }
        for (const Bytes& a : alleles) { hb.insert(hb.end(), a.begin(), a.end()); ho.push_back(static_cast<int32_t>(hb.size())); }
        for (Bytes* v : {&rb, &rq, &ri, &rd, &rg, &hb}) v->push_back(0);
        gatk_pairhmm_batch b{static_cast<int32_t>(reads.size()), static_cast<int32_t>(alleles.size()), rb.data(), rq.data(), ri.data(),
                             rd.data(), rg.data(), ro.data(), hb.data(), ho.data(), nullptr};
        HipLoglessPairHMM::check(gatk_pairhmm_upload(hmm.handle(), 1, &b, &r_));
        HipLoglessPairHMM::check(gatk_pairhmm_compute_resident(hmm.handle(), r_, 1));
    }
    ~ResidentReadLikelihoods() { if (r_) gatk_pairhmm_resident_free(hmm_.handle(), r_); }
    ResidentReadLikelihoods(const ResidentReadLikelihoods&) = delete;
    ResidentReadLikelihoods& operator=(const ResidentReadLikelihoods&) = delete;

    // ReadLikelihoods.normalizeLikelihoods, genotyper/ReadLikelihoods.java:353
    void normalizeLikelihoods(bool bestToZero, double maximumLikelihoodDifferenceCap) {
        HipLoglessPairHMM::check(gatk_pairhmm_normalize_resident(hmm_.handle(), r_, &reference_, bestToZero ? 1 : 0, maximumLikelihoodDifferenceCap));
    }
    // The reads ReadLikelihoods.filterPoorlyModeledReads would remove, :1059 (true = poorly modelled).
    std::vector<bool> poorlyModeledReads(double maximumErrorPerBase) {
        std::vector<uint8_t> f(reads_);
        uint8_t* p = f.data();
        HipLoglessPairHMM::check(gatk_pairhmm_poorly_modeled_resident(hmm_.handle(), r_, maximumErrorPerBase, &p));
        return std::vector<bool>(f.begin(), f.end());
    }
    // ReadLikelihoods.marginalize through an old-to-new allele index map (-1: dropped), :911-939.
    LikelihoodMatrix marginalize(const std::vector<int32_t>& oldToNewAlleleIndex, std::vector<Bytes> newAlleles) {
        if (oldToNewAlleleIndex.size() != alleles_) throw IllegalArgumentException("one new index per old allele is required");
        const int32_t n_new = static_cast<int32_t>(newAlleles.size());
        std::vector<double> out(reads_ * newAlleles.size());
        const int32_t* mp = oldToNewAlleleIndex.data();
        double* op = out.data();
        HipLoglessPairHMM::check(gatk_pairhmm_marginalize_resident(hmm_.handle(), r_, &mp, &n_new, &op));
        LikelihoodMatrix m(std::move(newAlleles), reads_);
        for (size_t r = 0; r < reads_; ++r)
            for (int32_t a = 0; a < n_new; ++a) m.set(a, r, out[r * n_new + a]);
        return m;
    }
    // values[allele][read] as they stand on the device
    LikelihoodMatrix matrix(std::vector<Bytes> alleles) {
        std::vector<double> out(reads_ * alleles_);
        double* op = out.data();
        HipLoglessPairHMM::check(gatk_pairhmm_download(hmm_.handle(), r_, &op));
        LikelihoodMatrix m(std::move(alleles), reads_);
        for (size_t r = 0; r < reads_; ++r)
            for (size_t a = 0; a < alleles_; ++a) m.set(a, r, out[r * alleles_ + a]);
        return m;
    }

private:
    HipLoglessPairHMM& hmm_;
    gatk_pairhmm_resident* r_ = nullptr;
    size_t alleles_, reads_;
    int32_t reference_;
};

}  // namespace gatk
