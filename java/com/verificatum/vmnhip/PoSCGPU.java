package com.verificatum.vmnhip;

import java.io.File;
import java.nio.ByteBuffer;

import com.verificatum.arithm.LargeInteger;
import com.verificatum.arithm.PGroupElement;
import com.verificatum.arithm.PGroupElementArray;
import com.verificatum.arithm.PRingElementArray;
import com.verificatum.arithm.Permutation;
import com.verificatum.eio.ByteTree;
import com.verificatum.eio.ByteTreeContainer;
import com.verificatum.eio.ByteTreeReader;
import com.verificatum.protocol.elgamal.ProtocolElGamal;
import com.verificatum.protocol.hvzk.PoSC;
import com.verificatum.ui.Log;

/** Drop-in for {@code PoSCTW} (src/java/com/verificatum/protocol/hvzk/PoSCTW.java:73-134 prove, 137-212 verify) over
 *  vmn_posc_* (include/vmnproofs.h).  Factory seam: ShufflerElGamal.java:120-124. */
public final class PoSCGPU extends ProtocolElGamal implements PoSC {
    private static final int[] COM = {GPUMessage.GARRAY, GPUMessage.ELEMENTS, GPUMessage.GARRAY, GPUMessage.ELEMENTS, GPUMessage.ELEMENTS};
    private static final int[] REP = {GPUMessage.RING, GPUMessage.RARRAY, GPUMessage.RING, GPUMessage.RING, GPUMessage.RARRAY};
    private final GPUGroup group;

    public PoSCGPU(final String sid, final ProtocolElGamal protocol, final String rosid, final File nizkp, final GPUGroup group) {
        super(sid, protocol, rosid, nizkp);
        this.group = group;
    }

    private long create(final boolean prover) {
        final long[] out = new long[1];
        final RandomSourceBridge rs = prover
            ? new RandomSourceBridge.OverVCR(randomSource, group.pGroup.getElementOrder(), group.expBytes, rbitlen) : null;
        VMNException.check(VMNProofs.vmn_posc_create(group.grp, vbitlen(), ebitlen(), rbitlen, rs, out));
        return out[0];
    }

    private byte[] seed(final Log log, final PGroupElement g, final PGroupElementArray h, final PGroupElementArray u) {
        final ByteTreeContainer challengeData = new ByteTreeContainer(g.toByteTree(), h.toByteTree(), u.toByteTree());
        return challenger.challenge(log.newChildLog(), challengeData, 8 * prg.minNoSeedBytes(), rbitlen);
    }

    @Override
    public void prove(final Log log, final PGroupElement g, final PGroupElementArray h, final PGroupElementArray u,
                      final PRingElementArray r, final Permutation pi) {
        log.info("GPU path: proving our permutation commitment.");
        final Log tempLog = log.newChildLog();
        final PGroupElementArrayGPU H = PGroupElementArrayGPU.of(group, h);
        final PGroupElementArrayGPU U = PGroupElementArrayGPU.of(group, u);
        final PRingElementArrayGPU R = PRingElementArrayGPU.of(group, r);
        final long P = create(true);
        VMNException.check(VMNProofs.vmn_posc_set_instance(P, group.encode(g), H.handle, U.handle, R.handle, GPUArrays.gatherTable(pi)));
        VMNException.check(VMNProofs.vmn_posc_commit_prepare(P));          // beside the hashing of (g, h, u)

        tempLog.info("GPU path: seed of the batching vector from the random oracle; vector expanded on the device.");
        final byte[] prgSeed = seed(tempLog, g, h, u);
        VMNException.check(VMNProofs.vmn_posc_set_batch_vector_seed(P, prgSeed, prgSeed.length));

        tempLog.info("GPU path: commitment kernels.");
        final long[] msg = new long[1];
        VMNException.check(VMNProofs.vmn_posc_commit(P, msg));
        final GPUMessage commitment = new GPUMessage(msg[0]);
        final ByteTree commitmentTree = ProofSupport.byteTree(commitment);
        if (nizkp != null) {
            commitmentTree.unsafeWriteTo(ProofSupport.file(nizkp, "PoSCCommitment", j));
        }
        tempLog.info("GPU path: commitment framed on the device and posted.");
        bullBoard.publish("Commitment", commitmentTree, tempLog);

        tempLog.info("GPU path: challenge from the random oracle.");
        final byte[] challengeBytes = challenger.challenge(tempLog.newChildLog(), new ByteTreeContainer(new ByteTree(prgSeed), commitmentTree),
                                                           vbitlen(), rbitlen);
        final byte[] v = LargeInteger.toPositive(challengeBytes).toByteArray();

        tempLog.info("GPU path: reply kernels.");
        VMNException.check(VMNProofs.vmn_posc_reply(P, v, v.length, msg));
        final GPUMessage reply = new GPUMessage(msg[0]);
        final ByteTree replyTree = ProofSupport.byteTree(reply);
        if (nizkp != null) {
            replyTree.unsafeWriteTo(ProofSupport.file(nizkp, "PoSCReply", j));
        }
        tempLog.info("GPU path: reply framed on the device and posted.");
        bullBoard.publish("Reply", replyTree, tempLog);

        commitment.free();
        reply.free();
        VMNProofs.vmn_posc_free(P);
        H.free();
        U.free();
        R.free();
    }

    @Override
    public boolean verify(final Log log, final int l, final PGroupElement g, final PGroupElementArray h, final PGroupElementArray u) {
        log.info("GPU path: checking the permutation commitment of " + ui.getDescrString(l) + ".");
        final Log tempLog = log.newChildLog();
        final int n = h.size();
        final PGroupElementArrayGPU H = PGroupElementArrayGPU.of(group, h);
        final PGroupElementArrayGPU U = PGroupElementArrayGPU.of(group, u);
        final long V = create(false);
        VMNException.check(VMNProofs.vmn_posc_set_instance(V, group.encode(g), H.handle, U.handle, 0, null));

        tempLog.info("GPU path: seed of the batching vector from the random oracle; vector expanded on the device.");
        final byte[] prgSeed = seed(tempLog, g, h, u);
        VMNException.check(VMNProofs.vmn_posc_set_batch_vector_seed(V, prgSeed, prgSeed.length));

        tempLog.info("GPU path: parsing the commitment (range and membership checks on the device).");
        final ByteTreeReader cr = bullBoard.waitFor(l, "Commitment", tempLog);
        final ByteBuffer cb = ProofSupport.direct(cr);
        cr.close();
        GPUMessage commitment = GPUMessage.parse(group, cb, cb.remaining(), COM, new long[] {n, 1, n, 1, 1});
        boolean malformed = commitment == null;
        if (!malformed) {
            try {
                VMNException.check(VMNProofs.vmn_posc_set_commitment(V, commitment.handle));
            } catch (final VMNException e) {
                if (!e.isFormat()) {
                    throw e;
                }
                malformed = true;
                commitment.free();
            }
        }
        if (malformed) {
            commitment = ProofSupport.trivialPoSCCommitment(group, n);
            VMNException.check(VMNProofs.vmn_posc_set_commitment(V, commitment.handle));
        }
        final ByteTree commitmentTree = ProofSupport.byteTree(commitment);
        if (nizkp != null) {
            commitmentTree.unsafeWriteTo(ProofSupport.file(nizkp, "PoSCCommitment", l));
        }

        tempLog.info("GPU path: challenge from the random oracle.");
        final byte[] challengeBytes = challenger.challenge(tempLog.newChildLog(), new ByteTreeContainer(new ByteTree(prgSeed), commitmentTree),
                                                           vbitlen(), rbitlen);
        final byte[] v = LargeInteger.toPositive(challengeBytes).toByteArray();
        VMNException.check(VMNProofs.vmn_posc_set_challenge(V, v, v.length));

        tempLog.info("GPU path: parsing the reply (range checks on the device).");
        final ByteTreeReader rr = bullBoard.waitFor(l, "Reply", tempLog);
        final ByteBuffer rb = ProofSupport.direct(rr);
        rr.close();
        tempLog.info("GPU path: evaluating the verification equations.");
        final GPUMessage reply = GPUMessage.parse(group, rb, rb.remaining(), REP, new long[] {1, n, 1, 1, n});
        boolean verdict = false;
        if (reply != null && !malformed) {
            final int[] out = new int[1];
            VMNException.check(VMNProofs.vmn_posc_verify(V, reply.handle, out));
            verdict = out[0] != 0;
            if (verdict && nizkp != null) {
                ProofSupport.byteTree(reply).unsafeWriteTo(ProofSupport.file(nizkp, "PoSCReply", l));
            }
        }
        tempLog.info(verdict ? "GPU path: proof accepted." : "GPU path: proof rejected.");
        if (reply != null) {
            reply.free();
        }
        commitment.free();
        VMNProofs.vmn_posc_free(V);
        H.free();
        U.free();
        return verdict;
    }
}
