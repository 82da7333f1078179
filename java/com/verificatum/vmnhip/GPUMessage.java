package com.verificatum.vmnhip;

import java.nio.ByteBuffer;

/** A proof message (vmn_msg): commitment or reply of PoS / PoSC / CCPoS in the reference's item order
 *  (src/java/com/verificatum/protocol/hvzk/PoSBasicTW.java:694-699, 880-886; PoSCBasicTW.java:524-528, 629-634;
 *  CCPoSBasicW.java:395, 480-483).  toByteTree gives exactly the bytes the reference publishes on the bulletin board and
 *  writes to proofs/*.bt; parse is what setCommitment(ByteTreeReader) / verify(ByteTreeReader) do: framing, range and
 *  subgroup membership, all on the GPU -- null when malformed, and the caller substitutes trivial values as the reference does
 *  (PoSBasicTW.java:794-815). */
public final class GPUMessage {
    public static final int GARRAY = 1, RARRAY = 2, ELEMENTS = 3, RING = 4;
    long handle;

    GPUMessage(final long handle) {
        this.handle = handle;
    }

    public ByteBuffer toByteTree() {
        final ByteBuffer buf = ByteBuffer.allocateDirect((int) VMNProofs.vmn_msg_bytetree_size(handle));
        VMNException.check(VMNProofs.vmn_msg_to_bytetreeDirect(handle, buf));
        return buf;
    }

    public static GPUMessage parse(final GPUGroup group, final ByteBuffer direct, final long len, final int[] layout, final long[] counts) {
        final long[] out = new long[1];
        final int[] formatOk = new int[1];
        VMNException.check(VMNProofs.vmn_msg_from_bytetreeDirect(group.grp, direct, len, layout, counts, layout.length, out, formatOk));
        return formatOk[0] != 0 ? new GPUMessage(out[0]) : null;
    }

    public void free() {
        if (handle != 0) {
            VMNProofs.vmn_msg_free(handle);
            handle = 0;
        }
    }
}
