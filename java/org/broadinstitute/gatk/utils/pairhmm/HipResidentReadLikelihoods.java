/*
 * The three ReadLikelihoods steps that follow PairHMM.computeLikelihoods in the likelihood engine -
 * normalizeLikelihoods (genotyper/ReadLikelihoods.java:353-404), the test behind filterPoorlyModeledReads
 * (:1059-1093) and the marginal likelihoods of marginalize (:911-939) - run on the MI355X while the likelihoods of
 * a region are still in its memory, and only then fetched into the ReadLikelihoods object.
 *
 * NOT COMPILED IN THIS REPOSITORY'S BUILD (no JDK in the build image).  It goes next to HipLoglessPairHMM.java.
 * ReadLikelihoods keeps its arrays private, so this is a companion object rather than a subclass: it scores
 * every sample of a ReadLikelihoods collection in one device pass, applies the steps there, and writes the values
 * back through ReadLikelihoods.Matrix.set (the call PairHMM.computeLikelihoods itself uses, PairHMM.java:224).
 * Dropping the flagged reads and building the marginalized collection stay with the caller, as bookkeeping on
 * Java objects: ReadLikelihoods.filterPoorlyModeledReads / marginalize do that from the same numbers.
 */
package org.broadinstitute.gatk.utils.pairhmm;

import org.broadinstitute.gatk.utils.genotyper.ReadLikelihoods;
import org.broadinstitute.gatk.utils.haplotype.Haplotype;
import org.broadinstitute.gatk.utils.sam.GATKSAMRecord;

import java.util.List;
import java.util.Map;

public final class HipResidentReadLikelihoods {

    /** Native object holding the samples' bytes and the gatk_pairhmm_resident of include/gatk_pairhmm.h; 0 after close(). */
    private long resident;
    private final ReadLikelihoods<Haplotype> likelihoods;

    private static native long jniResidentOpen(long handle);                                    // new native holder
    private static native void jniResidentAddSample(long resident, byte[][] readBases, byte[][] readQuals,
                                                    byte[][] insertionGOP, byte[][] deletionGOP, byte[][] overallGCP,
                                                    byte[][] haplotypeBases);                   // one region per sample
    private static native void jniResidentCompute(long resident);                               // gatk_pairhmm_upload + gatk_pairhmm_compute_resident
    private static native void jniResidentNormalize(long resident, int referenceAlleleIndex,
                                                    boolean bestToZero, double cap);            // gatk_pairhmm_normalize_resident
    private static native byte[][] jniResidentPoorlyModeled(long resident, double maxErrorPerBase); // gatk_pairhmm_poorly_modeled_resident
    private static native double[][] jniResidentMarginalize(long resident, int[] oldToNewAlleleIndexMap,
                                                            int newAlleleCount);                // gatk_pairhmm_marginalize_resident, read-major per sample
    private static native double[][] jniResidentFetch(long resident);                          // gatk_pairhmm_download, read-major per sample
    private static native void jniResidentClose(long resident);                                 // gatk_pairhmm_resident_free

    /**
     * Scores every sample's reads against the collection's alleles (one computeLikelihoods call per sample in the
     * reference, PairHMM.java:190; one device pass here) and keeps the result on the device.
     *
     * @param gcp the per-read gap-continuation penalties, as passed to PairHMM.computeLikelihoods.
     */
    public HipResidentReadLikelihoods(final HipLoglessPairHMM hmm, final ReadLikelihoods<Haplotype> likelihoods,
                                      final Map<GATKSAMRecord, byte[]> gcp) {
        this.likelihoods = likelihoods;
        resident = jniResidentOpen(hmm.nativeHandle());
        final int alleleCount = likelihoods.alleleCount();
        final byte[][] hapBases = new byte[alleleCount][];
        for (int a = 0; a < alleleCount; a++)
            hapBases[a] = likelihoods.alleleAt(a).getBases();
        for (int s = 0; s < likelihoods.sampleCount(); s++) {
            final List<GATKSAMRecord> reads = likelihoods.sampleReads(s);
            final int n = reads.size();
            final byte[][] bases = new byte[n][], quals = new byte[n][], ins = new byte[n][], del = new byte[n][], g = new byte[n][];
            int r = 0;
            for (final GATKSAMRecord read : reads) {
                bases[r] = read.getReadBases();
                quals[r] = read.getBaseQualities();
                ins[r] = read.getBaseInsertionQualities();
                del[r] = read.getBaseDeletionQualities();
                g[r] = gcp.get(read);
                r++;
            }
            jniResidentAddSample(resident, bases, quals, ins, del, g, hapBases);  // length checks of PairHMM.java:276-279 in the veneer
        }
        jniResidentCompute(resident);
    }

    /** ReadLikelihoods.normalizeLikelihoods(bestToZero, maximumLikelihoodDifferenceCap), :353-404, on the device, in place. */
    public void normalizeLikelihoods(final boolean bestToZero, final double maximumLikelihoodDifferenceCap) {
        if (maximumLikelihoodDifferenceCap >= 0.0 || Double.isNaN(maximumLikelihoodDifferenceCap))
            throw new IllegalArgumentException("the minimum reference likelihood fall cannot be positive");
        int referenceAlleleIndex = -1;
        for (int a = 0; a < likelihoods.alleleCount(); a++)
            if (likelihoods.alleleAt(a).isReference())
                referenceAlleleIndex = a;
        jniResidentNormalize(resident, referenceAlleleIndex, bestToZero, maximumLikelihoodDifferenceCap);
    }

    /** result[s][r] is true for the reads ReadLikelihoods.filterPoorlyModeledReads(maximumErrorPerBase) removes, :1059-1093. */
    public boolean[][] poorlyModeledReads(final double maximumErrorPerBase) {
        if (likelihoods.alleleCount() == 0)
            throw new IllegalStateException("unsupported for read-likelihood collections with no alleles");
        if (Double.isNaN(maximumErrorPerBase) || maximumErrorPerBase <= 0.0)
            throw new IllegalArgumentException("the maximum error per base must be a positive number");
        final byte[][] flags = jniResidentPoorlyModeled(resident, maximumErrorPerBase);
        final boolean[][] result = new boolean[flags.length][];
        for (int s = 0; s < flags.length; s++) {
            result[s] = new boolean[flags[s].length];
            for (int r = 0; r < flags[s].length; r++)
                result[s][r] = flags[s][r] != 0;
        }
        return result;
    }

    /** The array ReadLikelihoods.marginalLikelihoods builds when it keeps every read, :911-939: result[s][newAllele][read]. */
    public double[][][] marginalLikelihoods(final int newAlleleCount, final int[] oldToNewAlleleIndexMap) {
        final double[][] flat = jniResidentMarginalize(resident, oldToNewAlleleIndexMap, newAlleleCount);
        final double[][][] result = new double[flat.length][newAlleleCount][];
        for (int s = 0; s < flat.length; s++) {
            final int readCount = newAlleleCount == 0 ? 0 : flat[s].length / newAlleleCount;
            for (int a = 0; a < newAlleleCount; a++) {
                result[s][a] = new double[readCount];
                for (int r = 0; r < readCount; r++)
                    result[s][a][r] = flat[s][r * newAlleleCount + a];  // the device keeps matrices read-major
            }
        }
        return result;
    }

    /** Writes the likelihoods, as they now stand on the device, into the ReadLikelihoods collection. */
    public void fetch() {
        final double[][] flat = jniResidentFetch(resident);
        final int alleleCount = likelihoods.alleleCount();
        for (int s = 0; s < flat.length; s++) {
            final ReadLikelihoods.Matrix<Haplotype> matrix = likelihoods.sampleMatrix(s);
            final int readCount = matrix.readCount();
            int idx = 0;
            for (int r = 0; r < readCount; r++)
                for (int a = 0; a < alleleCount; a++)
                    matrix.set(a, r, flat[s][idx++]);
        }
    }

    public void close() {
        if (resident != 0) {
            jniResidentClose(resident);
            resident = 0;
        }
    }
}
