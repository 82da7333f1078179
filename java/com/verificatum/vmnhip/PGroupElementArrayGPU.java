package com.verificatum.vmnhip;

import java.nio.ByteBuffer;

import com.verificatum.arithm.LargeInteger;
import com.verificatum.arithm.PGroupElement;
import com.verificatum.arithm.PGroupElementArray;
import com.verificatum.arithm.Permutation;
import com.verificatum.eio.ByteTreeReader;

/** A device-resident {@code PGroupElementArray}: one vmn_garray handle.  The methods are the ones the reference calls on
 *  VCR's class (SURVEY.md App. B lists the call sites): exp / expProd / mul / prod / permute / shiftPush / copyOfRange /
 *  extract / get / size / equals / toByteTree / free -- each a single call of include/vmnhip.h over all N elements.
 *  This is the array type a storage model {@code arrays=gpu} hands out at the seam
 *  src/java/com/verificatum/protocol/elgamal/ProtocolElGamal.java:332-345 (next to "ram" and "file"). */
public final class PGroupElementArrayGPU {
    final GPUGroup group;
    long handle;

    PGroupElementArrayGPU(final GPUGroup group, final long handle) {
        this.group = group;
        this.handle = handle;
    }

    /** pGroup.toElementArray(size, reader): parse + range check on the GPU; subgroup membership is {@link #isMember()}. */
    public static PGroupElementArrayGPU fromByteTree(final GPUGroup group, final ByteBuffer direct, final long len, final long expectedSize) {
        final long[] out = new long[1];
        final int[] formatOk = new int[1];
        final int[] inRange = new int[1];
        VMNException.check(VMNHip.vmn_garray_from_bytetreeDirect(group.grp, direct, len, expectedSize, out, formatOk, inRange));
        if (formatOk[0] == 0 || inRange[0] == 0) {
            if (out[0] != 0) {
                VMNHip.vmn_garray_free(out[0]);
            }
            throw new VMNException(VMNException.ERR_FORMAT);          // ArithmFormatException / EIOException of the reference
        }
        return new PGroupElementArrayGPU(group, out[0]);
    }

    /** Upload of an array VCR holds in RAM or in a file (any {@code PGroupElementArray}). */
    public static PGroupElementArrayGPU of(final GPUGroup group, final PGroupElementArray a) {
        return GPUArrays.upload(group, a);
    }

    public int size() {
        return (int) VMNHip.vmn_garray_size(handle);
    }

    private PGroupElementArrayGPU wrap(final long[] out) {
        return new PGroupElementArrayGPU(group, out[0]);
    }

    /** X.exp(PRingElementArray): K1a. */
    public PGroupElementArrayGPU exp(final PRingElementArrayGPU e) {
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_garray_exp_array(handle, e.handle, 0, out));
        return wrap(out);
    }

    /** X.exp(PRingElement) / X.exp(LargeInteger): K1b, one shared exponent. */
    public PGroupElementArrayGPU exp(final LargeInteger e) {
        final long[] out = new long[1];
        final byte[] eb = e.toByteArray();
        VMNException.check(VMNHip.vmn_garray_exp_scalar(handle, eb, eb.length, out));
        return wrap(out);
    }

    /** X.expProd(E): K3, one element. */
    public byte[] expProd(final PRingElementArrayGPU e, final int bitLength) {
        final byte[] out = new byte[group.elemBytes];
        VMNException.check(VMNHip.vmn_garray_expprod(handle, e.handle, bitLength, out));
        return out;
    }

    public PGroupElementArrayGPU mul(final PGroupElementArrayGPU y) {
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_garray_mul(handle, y.handle, out));
        return wrap(out);
    }

    public PGroupElementArrayGPU inv() {
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_garray_inv(handle, out));
        return wrap(out);
    }

    public byte[] prod() {
        final byte[] out = new byte[group.elemBytes];
        VMNException.check(VMNHip.vmn_garray_prod(handle, out));
        return out;
    }

    /** X.permute(pi).  VCR puts X[i] at position pi.map(i) (the keep list of PermutationCommitment.java:398-405 pins it,
     *  include/vmnproofs.h vmn_permutation_shrink); the library gathers, out[i] = X[table[i]]: the table is pi's inverse. */
    public PGroupElementArrayGPU permute(final Permutation pi) {
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_garray_permute(handle, GPUArrays.gatherTable(pi), out));
        return wrap(out);
    }

    public PGroupElementArrayGPU shiftPush(final PGroupElement el) {
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_garray_shift_push(handle, group.encode(el), out));
        return wrap(out);
    }

    public PGroupElementArrayGPU copyOfRange(final int from, final int to) {
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_garray_copy_range(handle, from, to, out));
        return wrap(out);
    }

    public PGroupElementArrayGPU extract(final boolean[] keep) {
        final byte[] k = new byte[keep.length];
        for (int i = 0; i < keep.length; i++) {
            k[i] = (byte) (keep[i] ? 1 : 0);
        }
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_garray_extract(handle, k, out));
        return wrap(out);
    }

    public byte[] get(final int i) {
        final byte[] out = new byte[group.elemBytes];
        VMNException.check(VMNHip.vmn_garray_get(handle, i, out));
        return out;
    }

    public boolean equalsArray(final PGroupElementArrayGPU y) {
        final int[] eq = new int[1];
        VMNException.check(VMNHip.vmn_garray_equals(handle, y.handle, eq));
        return eq[0] != 0;
    }

    public boolean isMember() {
        final int[] ok = new int[1];
        VMNException.check(VMNHip.vmn_garray_is_member(handle, ok));
        return ok[0] != 0;
    }

    public long byteTreeSize() {
        return VMNHip.vmn_garray_bytetree_size(handle);
    }

    /** array.toByteTree(): framed on the GPU, downloaded straight into a direct (ideally page-locked) buffer. */
    public void toByteTree(final ByteBuffer direct) {
        VMNException.check(VMNHip.vmn_garray_to_bytetreeDirect(handle, direct));
    }

    /** PGroupElementArray.free(): explicit and early, as the reference does (PoSBasicTW.java:1088-1101). */
    public void free() {
        if (handle != 0) {
            VMNHip.vmn_garray_free(handle);
            handle = 0;
        }
    }
}
