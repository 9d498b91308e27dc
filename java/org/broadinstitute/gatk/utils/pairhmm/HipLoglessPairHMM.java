/*
 * PairHMM implementation backed by the MI355X (gfx950) engine, libgatk_pairhmm_hip.so, through the
 * JNI veneer in java/jni/gatk_pairhmm_jni.cpp.
 *
 * NOT COMPILED IN THIS REPOSITORY'S BUILD: the build image has no JDK.  This file is the
 * reference-side half of the drop-in: it goes next to PairHMM.java in
 * public/gatk-utils/src/main/java/org/broadinstitute/gatk/utils/pairhmm/ of a GATK 3.x tree and is
 * selected by adding a HIP_LOGLESS_CACHING entry to PairHMM.HMM_IMPLEMENTATION (PairHMM.java:56-75)
 * and a case to the factory that instantiates PairHMMs from that enum.
 */
package org.broadinstitute.gatk.utils.pairhmm;

import htsjdk.variant.variantcontext.Allele;
import org.broadinstitute.gatk.utils.genotyper.ReadLikelihoods;
import org.broadinstitute.gatk.utils.haplotype.Haplotype;
import org.broadinstitute.gatk.utils.sam.GATKSAMRecord;

import java.util.List;
import java.util.Map;
import java.util.concurrent.atomic.AtomicInteger;

public class HipLoglessPairHMM extends PairHMM {

    static {
        System.loadLibrary("gatk_pairhmm_hip_jni");
    }

    /** gatk_pairhmm* of include/gatk_pairhmm.h; 0 after close(). */
    private long handle;

    /** Instances created without a device take the node's GPUs in turn: the reference creates one PairHMM per
     *  NanoScheduler map thread (PairHMM.java:297), so -nct threads spread over all GPUs of the node. */
    private static final AtomicInteger nextDevice = new AtomicInteger(0);

    // ---- natives: one per C-ABI entry point that the PairHMM life cycle needs -----------------
    private static native int jniDeviceCount();                                         // gatk_pairhmm_device_count
    private static native long jniCreate(int device);                                   // gatk_pairhmm_create
    private static native long jniCreateMulti(int[] devices);                           // gatk_pairhmm_create_multi (null: every GPU)
    private static native void jniSetOption(long handle, String name, double value);    // gatk_pairhmm_set_option
    private static native void jniInitialize(long handle, int readMax, int hapMax);     // gatk_pairhmm_initialize
    private static native void jniComputeLikelihoods(long handle,                       // gatk_pairhmm_compute_likelihoods (the veneer concatenates the arrays)
                                                     byte[][] readBases, byte[][] readQuals,
                                                     byte[][] insertionGOP, byte[][] deletionGOP, byte[][] overallGCP,
                                                     byte[][] haplotypeBases, double[] likelihoodArray);
    private static native double jniComputePair(long handle, byte[] haplotypeBases, byte[] readBases, byte[] readQuals,
                                                byte[] insertionGOP, byte[] deletionGOP, byte[] overallGCP); // gatk_pairhmm_compute_pair
    private static native void jniFinalizeRegion(long handle);                          // gatk_pairhmm_finalize_region
    private static native void jniClose(long handle);                                   // gatk_pairhmm_close

    /** One GPU per instance, the node's GPUs in turn. */
    public HipLoglessPairHMM() {
        this(nextDevice.getAndIncrement() % Math.max(1, jniDeviceCount()));
    }

    public HipLoglessPairHMM(final int device) {
        handle = jniCreate(device);  // throws IllegalStateException when no GPU is present: no CPU fallback
    }

    /** One instance over several GPUs (null: all of them): for a caller that hands over many regions per call
     *  from one thread; the native side shards them and writes every region's likelihoods in place. */
    public HipLoglessPairHMM(final int[] devices) {
        handle = jniCreateMulti(devices);
    }

    /** The native gatk_pairhmm* of this instance, for HipResidentReadLikelihoods. */
    long nativeHandle() {
        return handle;
    }

    /** PairHMM.java:83-84. */
    @Override
    protected void doNotUseTristateCorrection() {
        super.doNotUseTristateCorrection();
        jniSetOption(handle, "do_not_use_tristate_correction", 1.0);
    }

    @Override
    public void initialize(final int readMaxLength, final int haplotypeMaxLength) {
        super.initialize(readMaxLength, haplotypeMaxLength);
        jniInitialize(handle, readMaxLength, haplotypeMaxLength);
    }

    @Override
    public void initialize(final List<Haplotype> haplotypes, final Map<String, List<GATKSAMRecord>> perSampleReadList,
                           final int readMaxLength, final int haplotypeMaxLength) {
        // Nothing is sent ahead of time: each computeLikelihoods call carries its own bytes (a few MB
        // over PCIe against milliseconds of kernel time).
        initialize(readMaxLength, haplotypeMaxLength);
    }

    /** Same contract as PairHMM.computeLikelihoods (PairHMM.java:190-236): fills likelihoods.set(a, r, lk)
     *  and mLikelihoodArray in read-major order. */
    @Override
    public void computeLikelihoods(final ReadLikelihoods.Matrix<Haplotype> likelihoods,
                                   final List<GATKSAMRecord> processedReads,
                                   final Map<GATKSAMRecord, byte[]> gcp) {
        if (processedReads.isEmpty())
            return;
        if (doProfiling)
            startTime = System.nanoTime();
        final List<Haplotype> alleles = likelihoods.alleles();
        final int readCount = processedReads.size();
        final int alleleCount = alleles.size();
        final byte[][] readBases = new byte[readCount][];
        final byte[][] readQuals = new byte[readCount][];
        final byte[][] insGOP = new byte[readCount][];
        final byte[][] delGOP = new byte[readCount][];
        final byte[][] overallGCP = new byte[readCount][];
        int r = 0;
        for (final GATKSAMRecord read : processedReads) {
            readBases[r] = read.getReadBases();
            readQuals[r] = read.getBaseQualities();
            insGOP[r] = read.getBaseInsertionQualities();
            delGOP[r] = read.getBaseDeletionQualities();
            overallGCP[r] = gcp.get(read);
            // PairHMM.java:276-279
            if (readQuals[r].length != readBases[r].length) throw new IllegalArgumentException("Read bases and read quals aren't the same size: " + readBases[r].length + " vs " + readQuals[r].length);
            if (insGOP[r].length != readBases[r].length) throw new IllegalArgumentException("Read bases and read insertion quals aren't the same size: " + readBases[r].length + " vs " + insGOP[r].length);
            if (delGOP[r].length != readBases[r].length) throw new IllegalArgumentException("Read bases and read deletion quals aren't the same size: " + readBases[r].length + " vs " + delGOP[r].length);
            if (overallGCP[r].length != readBases[r].length) throw new IllegalArgumentException("Read bases and overall GCP aren't the same size: " + readBases[r].length + " vs " + overallGCP[r].length);
            r++;
        }
        final byte[][] hapBases = new byte[alleleCount][];
        for (int a = 0; a < alleleCount; a++)
            hapBases[a] = ((Allele) alleles.get(a)).getBases();

        mLikelihoodArray = new double[readCount * alleleCount];
        jniComputeLikelihoods(handle, readBases, readQuals, insGOP, delGOP, overallGCP, hapBases, mLikelihoodArray);

        int idx = 0;
        for (int readIndex = 0; readIndex < readCount; readIndex++)
            for (int a = 0; a < alleleCount; a++)
                likelihoods.set(a, readIndex, mLikelihoodArray[idx++]);
        initialized = true;
        if (doProfiling) {
            threadLocalPairHMMComputeTimeDiff = (System.nanoTime() - startTime);
            pairHMMComputeTime += threadLocalPairHMMComputeTimeDiff;
        }
    }

    @Override
    protected double subComputeReadLikelihoodGivenHaplotypeLog10(final byte[] haplotypeBases, final byte[] readBases,
                                                                 final byte[] readQuals, final byte[] insertionGOP,
                                                                 final byte[] deletionGOP, final byte[] overallGCP,
                                                                 final int hapStartIndex, final boolean recacheReadValues,
                                                                 final int nextHapStartIndex) {
        // every pair is a full recompute (PairHMM.java:217,284 always pass recacheReadValues = true)
        return jniComputePair(handle, haplotypeBases, readBases, readQuals, insertionGOP, deletionGOP, overallGCP);
    }

    @Override
    public void finalizeRegion() {
        jniFinalizeRegion(handle);
    }

    @Override
    public void close() {
        super.close();
        if (handle != 0) {
            jniClose(handle);
            handle = 0;
        }
    }
}
