package com.verificatum.vmnhip;

import com.verificatum.crypto.RandomSource;
import com.verificatum.arithm.LargeInteger;

/** What vmn_random_source (include/vmnproofs.h) calls back into: the party's RandomSource, at the granularity the
 *  reference draws from it (PRing.randomElementArray / randomElement, LargeIntegerArray.random;
 *  src/java/com/verificatum/protocol/hvzk/PoSBasicTW.java:446, 465, 470-475, 583, 612, 667, 673, 687).  jni/vmnjni_rs.c
 *  looks the four methods up by name. */
public interface RandomSourceBridge {

    /** n rows of {@code expBytes} bytes, big-endian, each a uniform element of Z_q. */
    byte[] ringElements(long n);

    /** n rows of {@code expBytes} bytes holding {@code bits}-bit integers (reduced mod q when they can reach it). */
    byte[] integers(long n, int bits);

    /** 32 fresh bytes for ONE N-sized draw that the library expands on the GPU (PRGHeuristic over SHA-256). */
    byte[] arraySeed();

    /** false: N-sized draws come as host rows through the first two methods as well. */
    boolean deviceArrays();

    /** The bridge over VCR's {@code RandomSource} (the reference's {@code randomSource} field of a protocol,
     *  src/java/com/verificatum/protocol/elgamal/ProtocolElGamal.java:90-100).  Ring elements are drawn with
     *  {@code rbitlen} extra bits and reduced -- the sampling VCR documents for randomElement. */
    final class OverVCR implements RandomSourceBridge {
        private final RandomSource source;
        private final LargeInteger order;
        private final int expBytes;
        private final int rbitlen;

        public OverVCR(final RandomSource source, final LargeInteger order, final int expBytes, final int rbitlen) {
            this.source = source;
            this.order = order;
            this.expBytes = expBytes;
            this.rbitlen = rbitlen;
        }

        private void put(final byte[] rows, final long i, final LargeInteger x) {
            final byte[] b = x.toByteArray();                      // two's complement, minimal
            final int n = Math.min(b.length, expBytes);
            System.arraycopy(b, b.length - n, rows, (int) (i * expBytes) + expBytes - n, n);
        }

        @Override
        public byte[] ringElements(final long n) {
            final byte[] rows = new byte[(int) (n * expBytes)];
            for (long i = 0; i < n; i++) {
                put(rows, i, new LargeInteger(order.bitLength() + rbitlen, source).mod(order));
            }
            return rows;
        }

        @Override
        public byte[] integers(final long n, final int bits) {
            final byte[] rows = new byte[(int) (n * expBytes)];
            for (long i = 0; i < n; i++) {
                put(rows, i, new LargeInteger(bits, source).mod(order));
            }
            return rows;
        }

        @Override
        public byte[] arraySeed() {
            return source.getBytes(32);
        }

        @Override
        public boolean deviceArrays() {
            return true;
        }
    }
}
