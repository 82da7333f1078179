package com.verificatum.vmnhip;

import com.verificatum.arithm.ECqPGroup;
import com.verificatum.arithm.LargeInteger;
import com.verificatum.arithm.ModPGroup;
import com.verificatum.arithm.PGroup;
import com.verificatum.arithm.PGroupElement;

/** One GPU context (vmn_ctx) and one group on it (vmn_group) for a VCR {@code PGroup}: a safe-prime {@code ModPGroup} of up
 *  to 4096 bits or {@code ECqPGroup} over P-256 / P-384 (the reference's group shapes,
 *  src/java/com/verificatum/protocol/elgamal/ProtocolElGamal.java:738-800; default group P-256, demo/mixnet/.conf:153).
 *  The wire widths are the reference's: vmn_group_set_wire_bytes(0, 0) selects Java's BigInteger.toByteArray() lengths.
 *  The VCR accessors used here (getModulus, getElementOrder, getg, toByteArray of an element) are VCR 3.1.0 API
 *  [NOT-IN-REF: the classes are not in the reference tree]. */
public final class GPUGroup implements AutoCloseable {
    final long ctx;
    final long grp;
    final PGroup pGroup;
    final int elemBytes;
    final int expBytes;

    public GPUGroup(final int device, final PGroup pGroup) {
        this.pGroup = pGroup;
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_ctx_create(device, out));
        ctx = out[0];
        if (pGroup instanceof ModPGroup) {
            final ModPGroup mp = (ModPGroup) pGroup;
            final byte[] p = mp.getModulus().toByteArray();
            final int nb = p.length;
            VMNException.check(VMNHip.vmn_modp_group_create(ctx, p, fixed(mp.getElementOrder(), nb),
                                                            fixed(mp.getg().toLargeInteger(), nb), nb, out));
        } else if (pGroup instanceof ECqPGroup) {
            VMNException.check(VMNHip.vmn_ec_group_create(ctx, ((ECqPGroup) pGroup).getCurveName(), out));
        } else {
            throw new VMNException(VMNException.ERR_UNSUPPORTED);
        }
        grp = out[0];
        VMNException.check(VMNHip.vmn_group_set_wire_bytes(grp, 0, 0));
        elemBytes = (int) VMNHip.vmn_group_elem_bytes(grp);
        expBytes = (int) VMNHip.vmn_group_exp_bytes(grp);
    }

    static byte[] fixed(final LargeInteger x, final int nbytes) {
        final byte[] b = x.toByteArray();
        final byte[] out = new byte[nbytes];
        final int n = Math.min(b.length, nbytes);
        System.arraycopy(b, b.length - n, out, nbytes - n, n);
        return out;
    }

    /** One group element as the fixed-width bytes of the C ABI (the payload of its byte-tree leaf / leaves). */
    byte[] encode(final PGroupElement el) {
        return GPUArrays.leafPayload(el.toByteTree(), elemBytes);
    }

    /** Session setup: tables for the long-lived bases (the generator, the public key); include/vmnhip.h. */
    public void precomputeFixed(final PGroupElement base, final long nHint, final int usesHint) {
        VMNException.check(VMNHip.vmn_group_precompute_fixed(grp, encode(base), nHint, usesHint));
    }

    @Override
    public void close() {
        VMNHip.vmn_group_destroy(grp);
        VMNHip.vmn_ctx_destroy(ctx);
    }
}
