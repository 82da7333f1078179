package com.verificatum.vmnhip;

import java.io.File;
import java.nio.ByteBuffer;

import com.verificatum.arithm.PGroupElement;
import com.verificatum.arithm.PPGroupElement;
import com.verificatum.arithm.PPRingElementArray;
import com.verificatum.arithm.PRingElementArray;
import com.verificatum.eio.ByteTree;
import com.verificatum.eio.ByteTreeReader;

/** What PoSGPU / PoSCGPU / CCPoSGPU share: file names of the proofs directory, the wide public key as 2 * width elements,
 *  product-ring arrays as their columns, byte trees in and out of direct buffers, and the trivial messages the reference
 *  substitutes for malformed ones. */
final class ProofSupport {
    private ProofSupport() { }

    /** proofs/<name>%02d.bt, the names of PoSTW.java:281-307, PoSCTW.java:221-235, CCPoSW.java:274-288. */
    static File file(final File nizkp, final String name, final int index) {
        return new File(nizkp, String.format("%s%02d.bt", name, index));
    }

    /** ((g,..,g),(y,..,y)) -> g..g, y..y as elem_bytes rows (ProtocolElGamal.java:785-800). */
    static byte[] wideKey(final GPUGroup group, final PGroupElement pkey, final int width) {
        final byte[] out = new byte[2 * width * group.elemBytes];
        final PPGroupElement pp = (PPGroupElement) pkey;
        for (int half = 0; half < 2; half++) {
            final PGroupElement part = pp.project(half);
            for (int c = 0; c < width; c++) {
                final PGroupElement el = width == 1 ? part : ((PPGroupElement) part).project(c);
                System.arraycopy(group.encode(el), 0, out, (half * width + c) * group.elemBytes, group.elemBytes);
            }
        }
        return out;
    }

    /** A ring array over Z_q^width as its width columns. */
    static PRingElementArrayGPU[] columns(final GPUGroup group, final PRingElementArray s, final int width) {
        final PRingElementArrayGPU[] out = new PRingElementArrayGPU[width];
        for (int c = 0; c < width; c++) {
            out[c] = GPUArrays.upload(group, width == 1 ? s : ((PPRingElementArray) s).project(c));
        }
        return out;
    }

    static ByteBuffer direct(final ByteTreeReader reader) {
        final byte[] bytes = reader.readRemaining();            // the whole subtree this reader stands on
        final ByteBuffer buf = ByteBuffer.allocateDirect(bytes.length);
        buf.put(bytes);
        buf.flip();
        return buf;
    }

    static ByteTree byteTree(final PGroupElementArrayGPU a) {
        final ByteBuffer buf = ByteBuffer.allocateDirect((int) a.byteTreeSize());
        a.toByteTree(buf);
        return ByteTree.wrap(buf);
    }

    static ByteTree byteTree(final GPUMessage m) {
        return ByteTree.wrap(m.toByteTree());
    }

    /** n copies of the unit: the trivial array the reference substitutes (PoSBasicTW.java:787-792). */
    static PGroupElementArrayGPU ones(final GPUGroup group, final int n) {
        final byte[] rows = new byte[n * group.elemBytes];
        final byte[] one = group.encode(group.pGroup.getONE());
        for (int i = 0; i < n; i++) {
            System.arraycopy(one, 0, rows, i * group.elemBytes, group.elemBytes);
        }
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_garray_from_be(group.grp, rows, n, out, null));
        return new PGroupElementArrayGPU(group, out[0]);
    }

    private static GPUMessage message(final GPUGroup group, final int n, final int[] layout, final long[] counts) {
        final long[] out = new long[1];
        VMNException.check(VMNProofs.vmn_msg_create(out));
        final byte[] one = group.encode(group.pGroup.getONE());
        for (int i = 0; i < layout.length; i++) {
            if (layout[i] == GPUMessage.GARRAY) {
                final PGroupElementArrayGPU a = ones(group, n);
                VMNException.check(VMNProofs.vmn_msg_push_garray(out[0], a.handle));     // ownership passes to the message
            } else {
                final byte[] rows = new byte[(int) counts[i] * group.elemBytes];
                for (int k = 0; k < counts[i]; k++) {
                    System.arraycopy(one, 0, rows, k * group.elemBytes, group.elemBytes);
                }
                VMNException.check(VMNProofs.vmn_msg_push_elements(out[0], rows, counts[i], group.elemBytes));
            }
        }
        return new GPUMessage(out[0]);
    }

    /** (B, A', B', C', D', F') of units: PoSBasicTW.java:794-815. */
    static GPUMessage trivialPoSCommitment(final GPUGroup group, final int n, final int width) {
        return message(group, n, new int[] {1, 3, 1, 3, 3, 3}, new long[] {n, 1, n, 1, 1, 2L * width});
    }

    /** (B, A', B', C', D') of units: PoSCBasicTW.java:560-580. */
    static GPUMessage trivialPoSCCommitment(final GPUGroup group, final int n) {
        return message(group, n, new int[] {1, 3, 1, 3, 3}, new long[] {n, 1, n, 1, 1});
    }

    /** (A', B') of units: CCPoSBasicW.java:414-426. */
    static GPUMessage trivialCCPoSCommitment(final GPUGroup group, final int width) {
        return message(group, 0, new int[] {3, 3}, new long[] {1, 2L * width});
    }
}
