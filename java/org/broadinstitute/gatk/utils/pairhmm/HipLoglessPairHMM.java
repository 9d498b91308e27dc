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
    private static native void jniInitializeRegion(long handle, int[] readsPerSample,    // gatk_pairhmm_initialize_region: the reads of all samples, sample after sample
                                                   byte[][] readBases, byte[][] readQuals, byte[][] insertionGOP, byte[][] deletionGOP,
                                                   byte[] mappingQualities, byte[][] haplotypeBases, int readMax, int hapMax);
    private static native long jniSubmit(long handle, byte[][] readBases, byte[][] readQuals,     // gatk_pairhmm_submit: returns a ticket
                                         byte[][] insertionGOP, byte[][] deletionGOP, byte[][] overallGCP, byte[][] haplotypeBases);
    private static native void jniFetch(long handle, long ticket, double[] likelihoodArray);     // gatk_pairhmm_fetch
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

    /** When set, the region-level hook sends each read's mapping quality along and the engine caps the base qualities
     *  by it and squashes the ones under BASE_QUALITY_SCORE_THRESHOLD (PairHMM.java:54), which is what the upstream
     *  likelihood engine does to a read between this call and computeLikelihoods: the reads it then passes compare
     *  equal, after the same treatment, to the ones sent ahead. */
    private boolean sendMappingQualities = false;

    public void setSendMappingQualities(final boolean value) {
        sendMappingQualities = value;
        jniSetOption(handle, "squash_base_quals", value ? 1.0 : 0.0);
    }

    /** PairHMM.java:129-139, "used by the JNI implementations to transfer all data once to the native code": every
     *  sample's reads and the region's haplotypes go to the GPU in ONE pass, which is started here and not waited for.
     *  Each computeLikelihoods call that follows (one per sample) finds its likelihoods there; a call whose reads
     *  differ from the ones sent (another GCP than the constant 10, qualities changed in a way the engine does not
     *  reproduce) is scored on its own, so the results never depend on this call having been made. */
    @Override
    public void initialize(final List<Haplotype> haplotypes, final Map<String, List<GATKSAMRecord>> perSampleReadList,
                           final int readMaxLength, final int haplotypeMaxLength) {
        super.initialize(readMaxLength, haplotypeMaxLength);
        int total = 0;
        for (final List<GATKSAMRecord> reads : perSampleReadList.values())
            total += reads.size();
        final int[] readsPerSample = new int[perSampleReadList.size()];
        final byte[][] readBases = new byte[total][];
        final byte[][] readQuals = new byte[total][];
        final byte[][] insGOP = new byte[total][];
        final byte[][] delGOP = new byte[total][];
        final byte[] mapq = sendMappingQualities ? new byte[total] : null;
        int s = 0, r = 0;
        for (final List<GATKSAMRecord> reads : perSampleReadList.values()) {
            readsPerSample[s++] = reads.size();
            for (final GATKSAMRecord read : reads) {
                readBases[r] = read.getReadBases();
                readQuals[r] = read.getBaseQualities();
                insGOP[r] = read.getBaseInsertionQualities();
                delGOP[r] = read.getBaseDeletionQualities();
                if (mapq != null) mapq[r] = (byte) Math.min(read.getMappingQuality(), 254);
                r++;
            }
        }
        final byte[][] hapBases = new byte[haplotypes.size()][];
        for (int a = 0; a < hapBases.length; a++)
            hapBases[a] = haplotypes.get(a).getBases();
        jniInitializeRegion(handle, readsPerSample, readBases, readQuals, insGOP, delGOP, mapq, hapBases, readMaxLength, haplotypeMaxLength);
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

    /** computeLikelihoods in two halves, for a caller that keeps the next region on the GPU while it consumes this one:
     *  submit returns at once; fetch fills the matrix and mLikelihoodArray as computeLikelihoods would have. */
    public static final class Ticket {
        private final long id;
        private final ReadLikelihoods.Matrix<Haplotype> likelihoods;
        private final int readCount, alleleCount;
        private Ticket(final long id, final ReadLikelihoods.Matrix<Haplotype> likelihoods, final int readCount, final int alleleCount) {
            this.id = id; this.likelihoods = likelihoods; this.readCount = readCount; this.alleleCount = alleleCount;
        }
    }

    public Ticket submit(final ReadLikelihoods.Matrix<Haplotype> likelihoods, final List<GATKSAMRecord> processedReads,
                         final Map<GATKSAMRecord, byte[]> gcp) {
        final List<Haplotype> alleles = likelihoods.alleles();
        final int readCount = processedReads.size(), alleleCount = alleles.size();
        final byte[][] readBases = new byte[readCount][], readQuals = new byte[readCount][], insGOP = new byte[readCount][],
                       delGOP = new byte[readCount][], overallGCP = new byte[readCount][];
        int r = 0;
        for (final GATKSAMRecord read : processedReads) {
            readBases[r] = read.getReadBases();
            readQuals[r] = read.getBaseQualities();
            insGOP[r] = read.getBaseInsertionQualities();
            delGOP[r] = read.getBaseDeletionQualities();
            overallGCP[r] = gcp.get(read);
            r++;
        }
        final byte[][] hapBases = new byte[alleleCount][];
        for (int a = 0; a < alleleCount; a++)
            hapBases[a] = ((Allele) alleles.get(a)).getBases();
        return new Ticket(jniSubmit(handle, readBases, readQuals, insGOP, delGOP, overallGCP, hapBases), likelihoods, readCount, alleleCount);
    }

    public void fetch(final Ticket ticket) {
        mLikelihoodArray = new double[ticket.readCount * ticket.alleleCount];
        jniFetch(handle, ticket.id, mLikelihoodArray);
        int idx = 0;
        for (int readIndex = 0; readIndex < ticket.readCount; readIndex++)
            for (int a = 0; a < ticket.alleleCount; a++)
                ticket.likelihoods.set(a, readIndex, mLikelihoodArray[idx++]);
        initialized = true;
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
