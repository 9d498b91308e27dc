/*
 * Pins the parity oracle of the MI355X PairHMM engine to the reference's own Java: runs the in-tree Log10PairHMM
 * (exact and approximate log-sum, pairhmm/Log10PairHMM.java) over every (read, haplotype) pair of the engine's golden
 * inputs and prints the log10 likelihoods.
 *
 * NOT COMPILED IN THIS REPOSITORY'S BUILD (no JDK in the image).  On a machine with a GATK 3.8 source tree:
 *   1. python3 tools/export_golden_text.py            -> tests/golden/pairhmm_golden_inputs.txt
 *   2. copy this file to public/gatk-utils/src/test/java/org/broadinstitute/gatk/utils/pairhmm/ and run it with
 *      -Dpairhmm.golden.inputs=.../pairhmm_golden_inputs.txt -Dpairhmm.golden.outputs=.../reference_outputs.txt
 *      (mvn -Dtest=PinOracleTest test, or any TestNG runner)
 *   3. commit reference_outputs.txt as tests/golden/reference_outputs.txt of the engine's repository.
 * The class sits in the PairHMM's package because computeReadLikelihoodGivenHaplotypeLog10 (PairHMM.java:262) is protected.
 *
 * Output, one line per pair: NAME MODE READ_INDEX HAP_INDEX VALUE, MODE = exact | original (PairHMM.java:57-60),
 * VALUE printed with Double.toString (shortest text that reads back as the same double; -Infinity as such).
 */
package org.broadinstitute.gatk.utils.pairhmm;

import org.testng.Assert;
import org.testng.annotations.Test;

import java.io.BufferedReader;
import java.io.FileReader;
import java.io.PrintWriter;
import java.util.ArrayList;
import java.util.List;

public class PinOracleTest {

    private static byte[] bytesOfCsv(final String field) {
        if (field.equals("-")) return new byte[0];
        final String[] parts = field.split(",");
        final byte[] out = new byte[parts.length];
        for (int i = 0; i < parts.length; i++)
            out[i] = (byte) Integer.parseInt(parts[i]);  // unsigned byte values above 127 wrap, as they do in a BAM record
        return out;
    }

    private static byte[] bytesOfText(final String field) {
        return field.equals("-") ? new byte[0] : field.getBytes();
    }

    @Test
    public void printReferenceLikelihoods() throws Exception {
        final String in = System.getProperty("pairhmm.golden.inputs");
        final String out = System.getProperty("pairhmm.golden.outputs");
        Assert.assertNotNull(in, "-Dpairhmm.golden.inputs=FILE is required");
        Assert.assertNotNull(out, "-Dpairhmm.golden.outputs=FILE is required");
        final BufferedReader reader = new BufferedReader(new FileReader(in));
        final PrintWriter writer = new PrintWriter(out);
        try {
            String line = reader.readLine();
            while (line != null) {
                final String[] head = line.split(" ");
                Assert.assertEquals(head[0], "case");
                final String name = head[1];
                final int numReads = Integer.parseInt(head[2]), numHaps = Integer.parseInt(head[3]);
                final List<byte[][]> reads = new ArrayList<byte[][]>();
                final List<byte[]> haps = new ArrayList<byte[]>();
                int readMax = 1, hapMax = 1;
                for (int r = 0; r < numReads; r++) {
                    final String[] f = reader.readLine().split(" ");
                    Assert.assertEquals(f[0], "read");
                    reads.add(new byte[][]{bytesOfText(f[1]), bytesOfCsv(f[2]), bytesOfCsv(f[3]), bytesOfCsv(f[4]), bytesOfCsv(f[5])});
                    readMax = Math.max(readMax, reads.get(r)[0].length);
                }
                for (int a = 0; a < numHaps; a++) {
                    final String[] f = reader.readLine().split(" ");
                    Assert.assertEquals(f[0], "hap");
                    haps.add(bytesOfText(f[1]));
                    hapMax = Math.max(hapMax, haps.get(a).length);
                }
                for (final boolean exact : new boolean[]{true, false}) {
                    final Log10PairHMM hmm = new Log10PairHMM(exact);
                    hmm.initialize(readMax, hapMax);
                    for (int r = 0; r < numReads; r++) {
                        final byte[][] x = reads.get(r);
                        for (int a = 0; a < numHaps; a++) {
                            // recacheReadValues = true, as PairHMM.computeLikelihoods passes it (PairHMM.java:217,222)
                            final double lk = hmm.computeReadLikelihoodGivenHaplotypeLog10(haps.get(a), x[0], x[1], x[2], x[3], x[4], true,
                                    a == numHaps - 1 ? null : haps.get(a + 1));
                            writer.println(name + " " + (exact ? "exact" : "original") + " " + r + " " + a + " " + Double.toString(lk));
                        }
                    }
                }
                line = reader.readLine();
            }
        } finally {
            reader.close();
            writer.close();
        }
    }
}
