package com.verificatum.vmnhip;

import java.nio.ByteBuffer;

import com.verificatum.arithm.PGroupElementArray;
import com.verificatum.arithm.PPGroupElementArray;
import com.verificatum.arithm.PRingElementArray;
import com.verificatum.arithm.Permutation;
import com.verificatum.eio.ByteTreeBasic;

/** Conversions between VCR's host objects and the device arrays: uploads go through the array's byte tree (the format
 *  the library parses on the GPU, SURVEY.md App. D), ciphertext arrays are split into their 2 * width component arrays
 *  (struct of arrays, the way {@code PPGroupElementArray.project} exposes them,
 *  src/java/com/verificatum/protocol/elgamal/DistrElGamalSession.java:377-378), permutations become gather tables. */
final class GPUArrays {
    private GPUArrays() { }

    /** The gather table of X.permute(pi): VCR puts X[i] at position pi.map(i), so out[k] = X[pi^-1(k)]. */
    static int[] gatherTable(final Permutation pi) {
        final Permutation inv = pi.inv();
        final int[] table = new int[inv.size()];
        for (int i = 0; i < table.length; i++) {
            table[i] = inv.map(i);
        }
        return table;
    }

    static ByteBuffer direct(final ByteTreeBasic bt) {
        final byte[] bytes = bt.toByteArray();
        final ByteBuffer buf = ByteBuffer.allocateDirect(bytes.length);
        buf.put(bytes);
        buf.flip();
        return buf;
    }

    /** The payload bytes of an element's byte tree: leaf -> its bytes; node(leaf, leaf) (a curve point) -> x || y. */
    static byte[] leafPayload(final ByteTreeBasic bt, final int elemBytes) {
        final byte[] raw = bt.toByteArray();
        final byte[] out = new byte[elemBytes];
        if (raw[0] == 1) {
            System.arraycopy(raw, 5, out, 0, elemBytes);
        } else {
            final int half = elemBytes / 2;
            System.arraycopy(raw, 5 + 5, out, 0, half);
            System.arraycopy(raw, 5 + 5 + half + 5, out, half, half);
        }
        return out;
    }

    static PGroupElementArrayGPU upload(final GPUGroup group, final PGroupElementArray a) {
        final ByteBuffer buf = direct(a.toByteTree());
        return PGroupElementArrayGPU.fromByteTree(group, buf, buf.remaining(), a.size());
    }

    static PRingElementArrayGPU upload(final GPUGroup group, final PRingElementArray a) {
        final ByteBuffer buf = direct(a.toByteTree());
        final long[] out = new long[1];
        final int[] formatOk = new int[1];
        final int[] inRange = new int[1];
        VMNException.check(VMNHip.vmn_rarray_from_bytetreeDirect(group.grp, buf, buf.remaining(), a.size(), out, formatOk, inRange));
        if (formatOk[0] == 0 || inRange[0] == 0) {
            throw new VMNException(VMNException.ERR_FORMAT);
        }
        return new PRingElementArrayGPU(group, out[0]);
    }

    /** A ciphertext array of width w as its 2w component arrays [u_1..u_w, v_1..v_w]. */
    static PGroupElementArrayGPU[] components(final GPUGroup group, final PGroupElementArray ciphertexts, final int width) {
        final PPGroupElementArray pp = (PPGroupElementArray) ciphertexts;
        final PGroupElementArrayGPU[] out = new PGroupElementArrayGPU[2 * width];
        for (int half = 0; half < 2; half++) {
            final PGroupElementArray part = pp.project(half);
            for (int c = 0; c < width; c++) {
                final PGroupElementArray col = width == 1 ? part : ((PPGroupElementArray) part).project(c);
                out[half * width + c] = upload(group, col);
            }
        }
        return out;
    }

    static long[] handles(final PGroupElementArrayGPU[] arrays) {
        final long[] h = new long[arrays.length];
        for (int i = 0; i < h.length; i++) {
            h[i] = arrays[i].handle;
        }
        return h;
    }

    static long[] handles(final PRingElementArrayGPU[] arrays) {
        final long[] h = new long[arrays.length];
        for (int i = 0; i < h.length; i++) {
            h[i] = arrays[i].handle;
        }
        return h;
    }

    static void free(final PGroupElementArrayGPU[] arrays) {
        for (final PGroupElementArrayGPU a : arrays) {
            a.free();
        }
    }
}
