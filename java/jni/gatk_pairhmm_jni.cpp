// JNI veneer between org.broadinstitute.gatk.utils.pairhmm.HipLoglessPairHMM and the C-ABI of
// libgatk_pairhmm_hip.so (include/gatk_pairhmm.h).  It only moves bytes: pins or copies the Java
// arrays, calls one C entry point, maps the return code onto the exception the reference's own
// PairHMM would have thrown.
//
// Built only where a JDK is present (jni.h); the image this repository is built in has none, so
// the file compiles to nothing there.  Build on a JDK box with
//   g++ -O2 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
//       java/jni/gatk_pairhmm_jni.cpp -Lgatk_amd -lgatk_pairhmm_hip -o libgatk_pairhmm_hip_jni.so
#if __has_include(<jni.h>)
#include <jni.h>

#include <map>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "gatk_pairhmm.h"

namespace {

void throw_for(JNIEnv* env, int rc) {
    const char* cls = "java/lang/RuntimeException";
    switch (rc) {
        case GATK_PAIRHMM_ILLEGAL_ARGUMENT: cls = "java/lang/IllegalArgumentException"; break;
        case GATK_PAIRHMM_ILLEGAL_STATE: cls = "java/lang/IllegalStateException"; break;
        case GATK_PAIRHMM_INDEX_OUT_OF_BOUNDS: cls = "java/lang/ArrayIndexOutOfBoundsException"; break;
        case GATK_PAIRHMM_OUT_OF_MEMORY: cls = "java/lang/OutOfMemoryError"; break;
        case GATK_PAIRHMM_DEVICE_ERROR: cls = "java/lang/IllegalStateException"; break;
        default: break;
    }
    env->ThrowNew(env->FindClass(cls), gatk_pairhmm_last_error());
}

// Copies a byte[][] into one contiguous buffer with an offset table (the gatk_pairhmm_batch layout).
bool flatten(JNIEnv* env, jobjectArray arrays, std::vector<uint8_t>& data, std::vector<int32_t>& offsets) {
    const jsize n = env->GetArrayLength(arrays);
    offsets.assign(1, 0);
    for (jsize k = 0; k < n; ++k) {
        jbyteArray a = static_cast<jbyteArray>(env->GetObjectArrayElement(arrays, k));
        if (!a) return false;
        const jsize len = env->GetArrayLength(a);
        const size_t at = data.size();
        data.resize(at + len);
        env->GetByteArrayRegion(a, 0, len, reinterpret_cast<jbyte*>(data.data() + at));
        env->DeleteLocalRef(a);
        offsets.push_back(static_cast<int32_t>(data.size()));
    }
    data.push_back(0);  // never hand the C side a null data pointer
    return true;
}

gatk_pairhmm* H(jlong h) { return reinterpret_cast<gatk_pairhmm*>(h); }

// An outstanding jniSubmit: the engine's ticket and the array the engine writes the likelihoods into.
struct PendingResult {
    int64_t ticket = 0;
    std::vector<double> values;
};

// The results of a handle's tickets that have not been fetched yet: jniClose frees what the Java side never asked for.
std::mutex g_pending_mutex;
std::map<gatk_pairhmm*, std::set<PendingResult*>> g_pending;

// PairHMM.java:276-279: every per-read array must be as long as the read's bases.  The Java class checks this
// too; a caller that reaches the native method some other way gets the same exception instead of misaligned bytes.
bool same_lengths(JNIEnv* env, const std::vector<int32_t>& bases, const std::vector<int32_t>& other, const char* what) {
    if (bases == other) return true;
    const std::string msg = std::string("Read bases and ") + what + " aren't the same size";
    env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), msg.c_str());
    return false;
}

}  // namespace

extern "C" {

JNIEXPORT jint JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipLoglessPairHMM_jniDeviceCount(JNIEnv*, jclass) {
    int32_t n = 0;
    (void)gatk_pairhmm_device_count(&n);
    return n;
}

JNIEXPORT jlong JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipLoglessPairHMM_jniCreateMulti(JNIEnv* env, jclass, jintArray devices) {
    std::vector<int32_t> list;
    if (devices) {
        list.resize(static_cast<size_t>(env->GetArrayLength(devices)));
        if (!list.empty()) env->GetIntArrayRegion(devices, 0, static_cast<jsize>(list.size()), reinterpret_cast<jint*>(list.data()));
    }
    gatk_pairhmm* h = nullptr;
    if (const int rc = gatk_pairhmm_create_multi(list.empty() ? nullptr : list.data(), static_cast<int32_t>(list.size()), &h)) { throw_for(env, rc); return 0; }
    return reinterpret_cast<jlong>(h);
}

JNIEXPORT void JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipLoglessPairHMM_jniSetOption(JNIEnv* env, jclass, jlong h, jstring name, jdouble value) {
    const char* n = env->GetStringUTFChars(name, nullptr);
    if (!n) return;  // OutOfMemoryError is pending
    const int rc = gatk_pairhmm_set_option(H(h), n, value);
    env->ReleaseStringUTFChars(name, n);
    if (rc) throw_for(env, rc);
}

JNIEXPORT jlong JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipLoglessPairHMM_jniCreate(JNIEnv* env, jclass, jint device) {
    gatk_pairhmm* h = nullptr;
    if (const int rc = gatk_pairhmm_create(device, &h)) { throw_for(env, rc); return 0; }
    // one PairHMM object per map thread, one region per call: let the calls of all threads share device passes
    (void)gatk_pairhmm_set_option(h, "combine_calls", 1.0);
    return reinterpret_cast<jlong>(h);
}

JNIEXPORT void JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipLoglessPairHMM_jniInitialize(JNIEnv* env, jclass, jlong h, jint readMax, jint hapMax) {
    if (const int rc = gatk_pairhmm_initialize(H(h), readMax, hapMax)) throw_for(env, rc);
}

JNIEXPORT void JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipLoglessPairHMM_jniComputeLikelihoods(
    JNIEnv* env, jclass, jlong h, jobjectArray readBases, jobjectArray readQuals, jobjectArray insGOP, jobjectArray delGOP,
    jobjectArray gcp, jobjectArray hapBases, jdoubleArray likelihoodArray) {
    std::vector<uint8_t> rb, rq, ri, rd, rg, hb;
    std::vector<int32_t> ro, oq, oi, od, og, ho;
    if (!flatten(env, readBases, rb, ro) || !flatten(env, readQuals, rq, oq) || !flatten(env, insGOP, ri, oi) ||
        !flatten(env, delGOP, rd, od) || !flatten(env, gcp, rg, og) || !flatten(env, hapBases, hb, ho)) {
        env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "readBases cannot be null");
        return;
    }
    if (!same_lengths(env, ro, oq, "read quals") || !same_lengths(env, ro, oi, "read insertion quals") ||
        !same_lengths(env, ro, od, "read deletion quals") || !same_lengths(env, ro, og, "overall GCP"))
        return;
    gatk_pairhmm_batch b{static_cast<int32_t>(ro.size()) - 1, static_cast<int32_t>(ho.size()) - 1, rb.data(), rq.data(), ri.data(),
                         rd.data(), rg.data(), ro.data(), hb.data(), ho.data()};
    std::vector<double> out(static_cast<size_t>(b.num_reads) * b.num_haps);
    if (const int rc = gatk_pairhmm_compute_likelihoods(H(h), &b, out.data())) { throw_for(env, rc); return; }
    env->SetDoubleArrayRegion(likelihoodArray, 0, static_cast<jsize>(out.size()), out.data());
}

// PairHMM.initialize(haplotypes, perSampleReadList, readMaxLength, haplotypeMaxLength), PairHMM.java:137: the reads of
// all samples arrive sample after sample in one set of arrays with the per-sample counts beside them; every sample
// becomes one gatk_pairhmm_batch whose offset table is a window of the common one (offsets need not start at 0).
JNIEXPORT void JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipLoglessPairHMM_jniInitializeRegion(
    JNIEnv* env, jclass, jlong h, jintArray readsPerSample, jobjectArray readBases, jobjectArray readQuals, jobjectArray insGOP,
    jobjectArray delGOP, jbyteArray mappingQualities, jobjectArray hapBases, jint readMax, jint hapMax) {
    std::vector<uint8_t> rb, rq, ri, rd, hb, mq;
    std::vector<int32_t> ro, oq, oi, od, ho;
    if (!readsPerSample || !flatten(env, readBases, rb, ro) || !flatten(env, readQuals, rq, oq) || !flatten(env, insGOP, ri, oi) ||
        !flatten(env, delGOP, rd, od) || !flatten(env, hapBases, hb, ho)) {
        env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "readBases cannot be null");
        return;
    }
    if (!same_lengths(env, ro, oq, "read quals") || !same_lengths(env, ro, oi, "read insertion quals") || !same_lengths(env, ro, od, "read deletion quals"))
        return;
    std::vector<int32_t> counts(static_cast<size_t>(env->GetArrayLength(readsPerSample)) + 1);
    env->GetIntArrayRegion(readsPerSample, 0, static_cast<jsize>(counts.size()) - 1, reinterpret_cast<jint*>(counts.data()));
    counts.pop_back();
    const int64_t total = static_cast<int64_t>(ro.size()) - 1;
    if (mappingQualities) {
        if (env->GetArrayLength(mappingQualities) != total) {
            env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "one mapping quality per read is required");
            return;
        }
        mq.resize(static_cast<size_t>(total) + 1);
        env->GetByteArrayRegion(mappingQualities, 0, static_cast<jsize>(total), reinterpret_cast<jbyte*>(mq.data()));
    }
    std::vector<gatk_pairhmm_batch> samples;
    int64_t first = 0;
    for (const int32_t n : counts) {
        if (n < 0 || first + n > total) {
            env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "the per-sample read counts do not add up to the number of reads");
            return;
        }
        samples.push_back(gatk_pairhmm_batch{n, static_cast<int32_t>(ho.size()) - 1, rb.data(), rq.data(), ri.data(), rd.data(), nullptr,
                                             ro.data() + first, hb.data(), ho.data(), mq.empty() ? nullptr : mq.data() + first});
        first += n;
    }
    // the engine copies what it is given before it returns: the vectors above may go
    if (const int rc = gatk_pairhmm_initialize_region(H(h), static_cast<int32_t>(samples.size()), samples.data(), readMax, hapMax)) throw_for(env, rc);
}

JNIEXPORT jlong JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipLoglessPairHMM_jniSubmit(
    JNIEnv* env, jclass, jlong h, jobjectArray readBases, jobjectArray readQuals, jobjectArray insGOP, jobjectArray delGOP,
    jobjectArray gcp, jobjectArray hapBases) {
    std::vector<uint8_t> rb, rq, ri, rd, rg, hb;
    std::vector<int32_t> ro, oq, oi, od, og, ho;
    if (!flatten(env, readBases, rb, ro) || !flatten(env, readQuals, rq, oq) || !flatten(env, insGOP, ri, oi) ||
        !flatten(env, delGOP, rd, od) || !flatten(env, gcp, rg, og) || !flatten(env, hapBases, hb, ho)) {
        env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "readBases cannot be null");
        return 0;
    }
    if (!same_lengths(env, ro, oq, "read quals") || !same_lengths(env, ro, oi, "read insertion quals") ||
        !same_lengths(env, ro, od, "read deletion quals") || !same_lengths(env, ro, og, "overall GCP"))
        return 0;
    gatk_pairhmm_batch b{static_cast<int32_t>(ro.size()) - 1, static_cast<int32_t>(ho.size()) - 1, rb.data(), rq.data(), ri.data(),
                         rd.data(), rg.data(), ro.data(), hb.data(), ho.data(), nullptr};
    // the likelihoods land in memory this veneer owns until jniFetch hands them to the Java array
    PendingResult* pending = new PendingResult();
    pending->values.assign(static_cast<size_t>(b.num_reads) * static_cast<size_t>(b.num_haps), 0.0);
    double* out = pending->values.data();
    if (const int rc = gatk_pairhmm_submit(H(h), 1, &b, &out, &pending->ticket)) { delete pending; throw_for(env, rc); return 0; }
    {
        std::lock_guard<std::mutex> lock(g_pending_mutex);
        g_pending[H(h)].insert(pending);
    }
    return reinterpret_cast<jlong>(pending);
}

JNIEXPORT void JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipLoglessPairHMM_jniFetch(JNIEnv* env, jclass, jlong h, jlong ticket, jdoubleArray likelihoodArray) {
    PendingResult* pending = reinterpret_cast<PendingResult*>(ticket);
    {   // a ticket of this handle that has not been fetched before, or nothing is touched
        std::lock_guard<std::mutex> lock(g_pending_mutex);
        const auto of_handle = g_pending.find(H(h));
        if (!pending || of_handle == g_pending.end() || of_handle->second.erase(pending) == 0) pending = nullptr;
    }
    if (!pending) { env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "no such ticket"); return; }
    const int rc = gatk_pairhmm_fetch(H(h), pending->ticket);
    if (rc) throw_for(env, rc);
    else if (static_cast<size_t>(env->GetArrayLength(likelihoodArray)) != pending->values.size())
        env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "likelihoodArray has the wrong length for this ticket");
    else if (!pending->values.empty())
        env->SetDoubleArrayRegion(likelihoodArray, 0, static_cast<jsize>(pending->values.size()), pending->values.data());
    delete pending;
}

JNIEXPORT jdouble JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipLoglessPairHMM_jniComputePair(
    JNIEnv* env, jclass, jlong h, jbyteArray hap, jbyteArray bases, jbyteArray quals, jbyteArray ins, jbyteArray del, jbyteArray gcp) {
    auto get = [&](jbyteArray a) {
        std::vector<uint8_t> v(static_cast<size_t>(env->GetArrayLength(a)) + 1);
        env->GetByteArrayRegion(a, 0, static_cast<jsize>(v.size()) - 1, reinterpret_cast<jbyte*>(v.data()));
        return v;
    };
    const auto vh = get(hap), vb = get(bases), vq = get(quals), vi = get(ins), vd = get(del), vg = get(gcp);
    double lk = 0;
    const int rc = gatk_pairhmm_compute_pair(H(h), vh.data(), static_cast<int32_t>(vh.size()) - 1, vb.data(), vq.data(), vi.data(),
                                             vd.data(), vg.data(), static_cast<int32_t>(vb.size()) - 1, &lk);
    if (rc) throw_for(env, rc);
    return lk;
}

JNIEXPORT void JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipLoglessPairHMM_jniFinalizeRegion(JNIEnv* env, jclass, jlong h) {
    if (const int rc = gatk_pairhmm_finalize_region(H(h))) throw_for(env, rc);
}

JNIEXPORT void JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipLoglessPairHMM_jniClose(JNIEnv*, jclass, jlong h) {
    gatk_pairhmm_close(H(h));  // waits for the passes that still write into unfetched results, then drops their tickets
    std::set<PendingResult*> unfetched;
    {
        std::lock_guard<std::mutex> lock(g_pending_mutex);
        const auto of_handle = g_pending.find(H(h));
        if (of_handle != g_pending.end()) { unfetched.swap(of_handle->second); g_pending.erase(of_handle); }
    }
    for (PendingResult* p : unfetched) delete p;
}

// ---- HipResidentReadLikelihoods: the ReadLikelihoods steps on results that stay on the device ------------------------

namespace {

// One region per sample: the sample's reads against the collection's alleles.
struct ResidentRegions {
    gatk_pairhmm* handle = nullptr;
    gatk_pairhmm_resident* resident = nullptr;
    struct Sample {
        std::vector<uint8_t> rb, rq, ri, rd, rg, hb;
        std::vector<int32_t> ro, ho;
    };
    std::vector<Sample> samples;
    std::vector<gatk_pairhmm_batch> batches() const {
        std::vector<gatk_pairhmm_batch> b;
        for (const Sample& s : samples)
            b.push_back(gatk_pairhmm_batch{static_cast<int32_t>(s.ro.size()) - 1, static_cast<int32_t>(s.ho.size()) - 1, s.rb.data(), s.rq.data(),
                                           s.ri.data(), s.rd.data(), s.rg.data(), s.ro.data(), s.hb.data(), s.ho.data(), nullptr});
        return b;
    }
    int64_t pairs(size_t k) const { return (static_cast<int64_t>(samples[k].ro.size()) - 1) * (static_cast<int64_t>(samples[k].ho.size()) - 1); }
    int64_t reads(size_t k) const { return static_cast<int64_t>(samples[k].ro.size()) - 1; }
};

ResidentRegions* RR(jlong p) { return reinterpret_cast<ResidentRegions*>(p); }

jobjectArray to_double_arrays(JNIEnv* env, const std::vector<std::vector<double>>& v) {
    jobjectArray out = env->NewObjectArray(static_cast<jsize>(v.size()), env->FindClass("[D"), nullptr);
    for (size_t k = 0; out && k < v.size(); ++k) {
        jdoubleArray a = env->NewDoubleArray(static_cast<jsize>(v[k].size()));
        if (!a) return nullptr;
        if (!v[k].empty()) env->SetDoubleArrayRegion(a, 0, static_cast<jsize>(v[k].size()), v[k].data());
        env->SetObjectArrayElement(out, static_cast<jsize>(k), a);
        env->DeleteLocalRef(a);
    }
    return out;
}

}  // namespace

JNIEXPORT jlong JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipResidentReadLikelihoods_jniResidentOpen(JNIEnv*, jclass, jlong h) {
    ResidentRegions* r = new ResidentRegions();
    r->handle = H(h);
    return reinterpret_cast<jlong>(r);
}

JNIEXPORT void JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipResidentReadLikelihoods_jniResidentAddSample(
    JNIEnv* env, jclass, jlong p, jobjectArray readBases, jobjectArray readQuals, jobjectArray insGOP, jobjectArray delGOP, jobjectArray gcp,
    jobjectArray hapBases) {
    ResidentRegions::Sample s;
    std::vector<int32_t> oq, oi, od, og;
    if (!flatten(env, readBases, s.rb, s.ro) || !flatten(env, readQuals, s.rq, oq) || !flatten(env, insGOP, s.ri, oi) ||
        !flatten(env, delGOP, s.rd, od) || !flatten(env, gcp, s.rg, og) || !flatten(env, hapBases, s.hb, s.ho)) {
        env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "readBases cannot be null");
        return;
    }
    if (!same_lengths(env, s.ro, oq, "read quals") || !same_lengths(env, s.ro, oi, "read insertion quals") ||
        !same_lengths(env, s.ro, od, "read deletion quals") || !same_lengths(env, s.ro, og, "overall GCP"))
        return;
    RR(p)->samples.push_back(std::move(s));
}

JNIEXPORT void JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipResidentReadLikelihoods_jniResidentCompute(JNIEnv* env, jclass, jlong p) {
    ResidentRegions* r = RR(p);
    const std::vector<gatk_pairhmm_batch> b = r->batches();
    int rc = gatk_pairhmm_upload(r->handle, static_cast<int32_t>(b.size()), b.data(), &r->resident);
    if (rc == GATK_PAIRHMM_OK) rc = gatk_pairhmm_compute_resident(r->handle, r->resident, 1);
    if (rc) throw_for(env, rc);
}

JNIEXPORT void JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipResidentReadLikelihoods_jniResidentNormalize(
    JNIEnv* env, jclass, jlong p, jint referenceAlleleIndex, jboolean bestToZero, jdouble cap) {
    ResidentRegions* r = RR(p);
    const std::vector<int32_t> ref(r->samples.size(), referenceAlleleIndex);  // the samples share the collection's alleles
    if (const int rc = gatk_pairhmm_normalize_resident(r->handle, r->resident, ref.data(), bestToZero ? 1 : 0, cap)) throw_for(env, rc);
}

JNIEXPORT jobjectArray JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipResidentReadLikelihoods_jniResidentPoorlyModeled(
    JNIEnv* env, jclass, jlong p, jdouble maxErrorPerBase) {
    ResidentRegions* r = RR(p);
    std::vector<std::vector<uint8_t>> flags(r->samples.size());
    std::vector<uint8_t*> ptrs;
    for (size_t k = 0; k < flags.size(); ++k) { flags[k].assign(static_cast<size_t>(r->reads(k)) + 1, 0); ptrs.push_back(flags[k].data()); }
    if (const int rc = gatk_pairhmm_poorly_modeled_resident(r->handle, r->resident, maxErrorPerBase, ptrs.data())) { throw_for(env, rc); return nullptr; }
    jobjectArray out = env->NewObjectArray(static_cast<jsize>(flags.size()), env->FindClass("[B"), nullptr);
    for (size_t k = 0; out && k < flags.size(); ++k) {
        jbyteArray a = env->NewByteArray(static_cast<jsize>(r->reads(k)));
        if (!a) return nullptr;
        if (r->reads(k) > 0) env->SetByteArrayRegion(a, 0, static_cast<jsize>(r->reads(k)), reinterpret_cast<const jbyte*>(flags[k].data()));
        env->SetObjectArrayElement(out, static_cast<jsize>(k), a);
        env->DeleteLocalRef(a);
    }
    return out;
}

JNIEXPORT jobjectArray JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipResidentReadLikelihoods_jniResidentMarginalize(
    JNIEnv* env, jclass, jlong p, jintArray oldToNew, jint newAlleleCount) {
    ResidentRegions* r = RR(p);
    if (newAlleleCount < 0 || !oldToNew) {
        env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "the new allele count cannot be negative and the index map cannot be null");
        return nullptr;
    }
    std::vector<int32_t> map(static_cast<size_t>(env->GetArrayLength(oldToNew)) + 1);
    env->GetIntArrayRegion(oldToNew, 0, static_cast<jsize>(map.size()) - 1, reinterpret_cast<jint*>(map.data()));
    std::vector<std::vector<double>> outs(r->samples.size());
    std::vector<const int32_t*> maps(r->samples.size(), map.data());
    std::vector<int32_t> counts(r->samples.size(), newAlleleCount);
    std::vector<double*> ptrs;
    for (size_t k = 0; k < outs.size(); ++k) {
        if (static_cast<size_t>(env->GetArrayLength(oldToNew)) != r->samples[k].ho.size() - 1) {
            env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "the allele index map must have one entry per allele");
            return nullptr;
        }
        outs[k].assign(static_cast<size_t>(r->reads(k)) * static_cast<size_t>(newAlleleCount), 0.0);
        ptrs.push_back(outs[k].empty() ? nullptr : outs[k].data());
    }
    if (const int rc = gatk_pairhmm_marginalize_resident(r->handle, r->resident, maps.data(), counts.data(), ptrs.data())) { throw_for(env, rc); return nullptr; }
    return to_double_arrays(env, outs);
}

JNIEXPORT jobjectArray JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipResidentReadLikelihoods_jniResidentFetch(JNIEnv* env, jclass, jlong p) {
    ResidentRegions* r = RR(p);
    std::vector<std::vector<double>> outs(r->samples.size());
    std::vector<double*> ptrs;
    for (size_t k = 0; k < outs.size(); ++k) {
        outs[k].assign(static_cast<size_t>(r->pairs(k)), 0.0);
        ptrs.push_back(outs[k].empty() ? nullptr : outs[k].data());
    }
    if (const int rc = gatk_pairhmm_download(r->handle, r->resident, ptrs.data())) { throw_for(env, rc); return nullptr; }
    return to_double_arrays(env, outs);
}

JNIEXPORT void JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_HipResidentReadLikelihoods_jniResidentClose(JNIEnv*, jclass, jlong p) {
    ResidentRegions* r = RR(p);
    if (!r) return;
    if (r->resident) gatk_pairhmm_resident_free(r->handle, r->resident);
    delete r;
}

}  // extern "C"
#endif  // __has_include(<jni.h>)
