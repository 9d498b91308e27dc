// The JNI symbol set of GATK 3.x's own native PairHMM class, org.broadinstitute.gatk.utils.pairhmm.VectorLoglessPairHMM,
// routed to the C-ABI of libgatk_pairhmm_hip.so (include/gatk_pairhmm.h), so that an unmodified upstream
// VectorLoglessPairHMM.class that loads this library instead of libVectorLoglessPairHMM.so would run on the GPU.
//
// UNVERIFIABLE HERE, and marked so on purpose: neither that Java class nor its two holder classes
// (JNIReadDataHolderClass, JNIHaplotypeDataHolderClass) are in the reference tree, which mentions the implementation
// only in the selector enum (pairhmm/PairHMM.java:63-66).  Method names, argument lists and holder field names below
// are written from recollection of the upstream sources; before relying on them, compare with
// `javap -s -p` of the class in a real GATK 3.8 jar.  The in-tree facts they are consistent with: the read-major
// result array (PairHMM.java:206,225) and the hooks the reference reserves for native back ends
// (initialize(haplotypes, reads, ...) PairHMM.java:129-139, finalizeRegion :121-127, close :361-368).
//
// Built only where a JDK is present (jni.h), like gatk_pairhmm_jni.cpp:
//   g++ -O2 -fPIC -shared -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
//       java/jni/vector_logless_pairhmm_jni.cpp -Lgatk_amd -lgatk_pairhmm_hip -o libVectorLoglessPairHMM.so
#if __has_include(<jni.h>)
#include <jni.h>

#include <atomic>
#include <vector>

#include "gatk_pairhmm.h"

namespace {

// Field IDs of the holder classes, looked up once in jniInitializeClassFieldsAndMachineMask.
jfieldID g_read_bases, g_read_quals, g_insertion_gop, g_deletion_gop, g_overall_gcp, g_haplotype_bases;

// One engine handle per calling thread (upstream creates one PairHMM per map thread, PairHMM.java:297), the
// node's GPUs in turn; region-sized calls of all threads share device passes.
std::atomic<int> g_next_device{0};
struct ThreadHandle {
    gatk_pairhmm* h = nullptr;
    ~ThreadHandle() { gatk_pairhmm_close(h); }
};
thread_local ThreadHandle t_handle;

gatk_pairhmm* handle_of_this_thread() {
    if (!t_handle.h) {
        int32_t n = 0;
        (void)gatk_pairhmm_device_count(&n);
        if (n <= 0) n = 1;  // create reports the missing device
        if (gatk_pairhmm_create(g_next_device.fetch_add(1) % n, &t_handle.h) != GATK_PAIRHMM_OK) return nullptr;
        (void)gatk_pairhmm_set_option(t_handle.h, "combine_calls", 1.0);
    }
    return t_handle.h;
}

void throw_for(JNIEnv* env, int rc) {
    const char* cls = "java/lang/IllegalStateException";
    if (rc == GATK_PAIRHMM_ILLEGAL_ARGUMENT) cls = "java/lang/IllegalArgumentException";
    else if (rc == GATK_PAIRHMM_INDEX_OUT_OF_BOUNDS) cls = "java/lang/ArrayIndexOutOfBoundsException";
    else if (rc == GATK_PAIRHMM_OUT_OF_MEMORY) cls = "java/lang/OutOfMemoryError";
    env->ThrowNew(env->FindClass(cls), gatk_pairhmm_last_error());
}

// The byte[] fields of an array of holder objects, pinned for the duration of one call.
struct PinnedArrays {
    JNIEnv* env;
    std::vector<jbyteArray> arrays;
    std::vector<const uint8_t*> data;
    std::vector<int32_t> lengths;
    explicit PinnedArrays(JNIEnv* e) : env(e) {}
    bool add(jobject holder, jfieldID field) {
        jbyteArray a = static_cast<jbyteArray>(env->GetObjectField(holder, field));
        if (!a) return false;
        jbyte* p = env->GetByteArrayElements(a, nullptr);
        if (!p) return false;
        arrays.push_back(a);
        data.push_back(reinterpret_cast<const uint8_t*>(p));
        lengths.push_back(env->GetArrayLength(a));
        return true;
    }
    ~PinnedArrays() {
        for (size_t k = 0; k < arrays.size(); ++k) {
            env->ReleaseByteArrayElements(arrays[k], reinterpret_cast<jbyte*>(const_cast<uint8_t*>(data[k])), JNI_ABORT);
            env->DeleteLocalRef(arrays[k]);
        }
    }
};

}  // namespace

extern "C" {

// upstream: returns a bit mask of the vector instruction sets of the host; nothing on the GPU path depends on it
JNIEXPORT jlong JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_VectorLoglessPairHMM_jniGetMachineType(JNIEnv*, jobject) {
    return 0;
}

JNIEXPORT void JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_VectorLoglessPairHMM_jniInitializeClassFieldsAndMachineMask(
    JNIEnv* env, jobject, jclass readDataHolderClass, jclass haplotypeDataHolderClass, jlong /*mask*/) {
    g_read_bases = env->GetFieldID(readDataHolderClass, "readBases", "[B");
    g_read_quals = env->GetFieldID(readDataHolderClass, "readQuals", "[B");
    g_insertion_gop = env->GetFieldID(readDataHolderClass, "insertionGOP", "[B");
    g_deletion_gop = env->GetFieldID(readDataHolderClass, "deletionGOP", "[B");
    g_overall_gcp = env->GetFieldID(readDataHolderClass, "overallGCP", "[B");
    g_haplotype_bases = env->GetFieldID(haplotypeDataHolderClass, "haplotypeBases", "[B");
    // a missing field leaves NoSuchFieldError pending: the names above did not match this jar
}

// upstream sends a region's haplotypes ahead of its reads; here every jniComputeLikelihoods call carries its own
JNIEXPORT void JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_VectorLoglessPairHMM_jniInitializeHaplotypes(
    JNIEnv*, jobject, jint /*numHaplotypes*/, jobjectArray /*haplotypeDataArray*/) {}

JNIEXPORT void JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_VectorLoglessPairHMM_jniFinalizeRegion(JNIEnv* env, jobject) {
    if (gatk_pairhmm* h = t_handle.h)
        if (const int rc = gatk_pairhmm_finalize_region(h)) throw_for(env, rc);
}

// likelihoodArray[r * numHaplotypes + h], the order of mLikelihoodArray (PairHMM.java:206,225)
JNIEXPORT void JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_VectorLoglessPairHMM_jniComputeLikelihoods(
    JNIEnv* env, jobject, jint numReads, jint numHaplotypes, jobjectArray readDataArray, jobjectArray haplotypeDataArray,
    jdoubleArray likelihoodArray, jint /*maxNumThreadsToUse*/) {
    if (numReads <= 0) return;  // PairHMM.java:193
    gatk_pairhmm* h = handle_of_this_thread();
    if (!h) { throw_for(env, GATK_PAIRHMM_DEVICE_ERROR); return; }
    // every pinned byte[] keeps a local reference until the call ends: far more than the 16 a native frame is promised
    if (env->EnsureLocalCapacity(5 * numReads + numHaplotypes + 16) != 0) return;  // OutOfMemoryError is pending
    PinnedArrays bases(env), quals(env), ins(env), del(env), gcp(env), haps(env);
    for (jint r = 0; r < numReads; ++r) {
        jobject holder = env->GetObjectArrayElement(readDataArray, r);
        const bool ok = holder && bases.add(holder, g_read_bases) && quals.add(holder, g_read_quals) && ins.add(holder, g_insertion_gop) &&
                        del.add(holder, g_deletion_gop) && gcp.add(holder, g_overall_gcp);
        if (holder) env->DeleteLocalRef(holder);
        if (!ok) { env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "readBases cannot be null"); return; }
        // PairHMM.java:276-279
        const int32_t len = bases.lengths.back();
        if (quals.lengths.back() != len || ins.lengths.back() != len || del.lengths.back() != len || gcp.lengths.back() != len) {
            env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "Read bases and read quals aren't the same size");
            return;
        }
    }
    for (jint a = 0; a < numHaplotypes; ++a) {
        jobject holder = env->GetObjectArrayElement(haplotypeDataArray, a);
        const bool ok = holder && haps.add(holder, g_haplotype_bases);
        if (holder) env->DeleteLocalRef(holder);
        if (!ok) { env->ThrowNew(env->FindClass("java/lang/IllegalArgumentException"), "haplotypeBases cannot be null"); return; }
    }
    std::vector<double> out(static_cast<size_t>(numReads) * static_cast<size_t>(numHaplotypes));
    const int rc = gatk_pairhmm_compute_likelihoods_ptrs(h, numReads, numHaplotypes, bases.data.data(), quals.data.data(), ins.data.data(),
                                                         del.data.data(), gcp.data.data(), bases.lengths.data(), haps.data.data(),
                                                         haps.lengths.data(), out.data());
    if (rc) { throw_for(env, rc); return; }
    env->SetDoubleArrayRegion(likelihoodArray, 0, static_cast<jsize>(out.size()), out.data());
}

JNIEXPORT void JNICALL Java_org_broadinstitute_gatk_utils_pairhmm_VectorLoglessPairHMM_jniClose(JNIEnv*, jobject) {
    gatk_pairhmm_close(t_handle.h);
    t_handle.h = nullptr;
}

}  // extern "C"
#endif  // __has_include(<jni.h>)
