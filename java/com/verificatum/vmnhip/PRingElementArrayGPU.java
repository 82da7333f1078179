package com.verificatum.vmnhip;

import java.nio.ByteBuffer;

import com.verificatum.arithm.LargeInteger;
import com.verificatum.arithm.PRingElementArray;
import com.verificatum.arithm.Permutation;

/** A device-resident {@code PRingElementArray} / {@code PFieldElementArray} over Z_q: one vmn_rarray handle, with the
 *  methods the reference calls (SURVEY.md App. B: mul, add, neg, mulAdd, recLin, prods, innerProduct, sum, prod, permute,
 *  shiftPush, copyOfRange; src/java/com/verificatum/protocol/hvzk/PoSBasicTW.java:583-604, 642-645, 861-878). */
public final class PRingElementArrayGPU {
    final GPUGroup group;
    long handle;

    PRingElementArrayGPU(final GPUGroup group, final long handle) {
        this.group = group;
        this.handle = handle;
    }

    public static PRingElementArrayGPU of(final GPUGroup group, final PRingElementArray a) {
        return GPUArrays.upload(group, a);
    }

    /** pRing.randomElementArray(size, randomSource, rbitlen): expanded on the GPU from 32 bytes of the source
     *  (ShufflerElGamalSession.java:408-409; PermutationCommitment.java:189-199). */
    public static PRingElementArrayGPU random(final GPUGroup group, final RandomSourceBridge rs, final long n, final int rbitlen) {
        final long[] out = new long[1];
        VMNException.check(VMNProofs.vmn_rarray_random(group.grp, rs, n, rbitlen, out));
        return new PRingElementArrayGPU(group, out[0]);
    }

    public int size() {
        return (int) VMNHip.vmn_rarray_size(handle);
    }

    private PRingElementArrayGPU wrap(final long[] out) {
        return new PRingElementArrayGPU(group, out[0]);
    }

    public PRingElementArrayGPU mul(final PRingElementArrayGPU y) {
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_rarray_mul(handle, y.handle, out));
        return wrap(out);
    }

    public PRingElementArrayGPU add(final PRingElementArrayGPU y) {
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_rarray_add(handle, y.handle, out));
        return wrap(out);
    }

    public PRingElementArrayGPU neg() {
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_rarray_neg(handle, out));
        return wrap(out);
    }

    /** x.mulAdd(v, y) = x v + y. */
    public PRingElementArrayGPU mulAdd(final LargeInteger v, final PRingElementArrayGPU y) {
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_rarray_mul_add(handle, GPUGroup.fixed(v, group.expBytes), y == null ? 0 : y.handle, out));
        return wrap(out);
    }

    /** b.recLin(e): (x, d) with x_0 = b_0, x_i = x_{i-1} e_i + b_i; {@code last} receives d. */
    public PRingElementArrayGPU recLin(final PRingElementArrayGPU e, final byte[] last) {
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_rarray_rec_lin(handle, e.handle, out, last));
        return wrap(out);
    }

    public PRingElementArrayGPU prods() {
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_rarray_prods(handle, out));
        return wrap(out);
    }

    public byte[] innerProduct(final PRingElementArrayGPU y) {
        final byte[] out = new byte[group.expBytes];
        VMNException.check(VMNHip.vmn_rarray_inner_product(handle, y.handle, out));
        return out;
    }

    public byte[] sum() {
        final byte[] out = new byte[group.expBytes];
        VMNException.check(VMNHip.vmn_rarray_sum(handle, out));
        return out;
    }

    public byte[] prod() {
        final byte[] out = new byte[group.expBytes];
        VMNException.check(VMNHip.vmn_rarray_prod(handle, out));
        return out;
    }

    public PRingElementArrayGPU permute(final Permutation pi) {
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_rarray_permute(handle, GPUArrays.gatherTable(pi), out));
        return wrap(out);
    }

    public PRingElementArrayGPU shiftPush(final LargeInteger el) {
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_rarray_shift_push(handle, GPUGroup.fixed(el, group.expBytes), out));
        return wrap(out);
    }

    public PRingElementArrayGPU copyOfRange(final int from, final int to) {
        final long[] out = new long[1];
        VMNException.check(VMNHip.vmn_rarray_copy_range(handle, from, to, out));
        return wrap(out);
    }

    /** Largest bit length among the entries: what a verifier uses for a received exponent array (include/vmnhip.h). */
    public int maxBits() {
        final int[] bits = new int[1];
        VMNException.check(VMNHip.vmn_rarray_max_bits(handle, bits));
        return bits[0];
    }

    public void toByteTree(final ByteBuffer direct) {
        VMNException.check(VMNHip.vmn_rarray_to_bytetreeDirect(handle, direct));
    }

    public void free() {
        if (handle != 0) {
            VMNHip.vmn_rarray_free(handle);
            handle = 0;
        }
    }
}
