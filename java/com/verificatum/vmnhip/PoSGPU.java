package com.verificatum.vmnhip;

import java.io.File;
import java.nio.ByteBuffer;

import com.verificatum.arithm.LargeInteger;
import com.verificatum.arithm.PGroupElement;
import com.verificatum.arithm.PGroupElementArray;
import com.verificatum.arithm.PPGroup;
import com.verificatum.arithm.PRingElementArray;
import com.verificatum.arithm.Permutation;
import com.verificatum.eio.ByteTree;
import com.verificatum.eio.ByteTreeContainer;
import com.verificatum.eio.ByteTreeReader;
import com.verificatum.protocol.elgamal.ProtocolElGamal;
import com.verificatum.protocol.hvzk.PoS;
import com.verificatum.ui.Log;

/** Drop-in for {@code PoSTW} (src/java/com/verificatum/protocol/hvzk/PoSTW.java:95-165 prove, 177-272 verify): the same
 *  bulletin-board labels, file names, challenge derivations and message bytes, with PoSBasicTW's arithmetic behind
 *  include/vmnproofs.h (vmn_pos_*).  Swapped in by replacing {@code new PoSTWFactory()} with {@code new PoSGPUFactory()} in
 *  src/java/com/verificatum/protocol/mixnet/ShufflerElGamal.java:120-124. */
public final class PoSGPU extends ProtocolElGamal implements PoS {
    private static final int[] COM = {GPUMessage.GARRAY, GPUMessage.ELEMENTS, GPUMessage.GARRAY, GPUMessage.ELEMENTS,
                                      GPUMessage.ELEMENTS, GPUMessage.ELEMENTS};
    private static final int[] REP = {GPUMessage.RING, GPUMessage.RARRAY, GPUMessage.RING, GPUMessage.RING,
                                      GPUMessage.RARRAY, GPUMessage.RING};

    private final GPUGroup group;
    private long P;                       // vmn_pos of the prover
    private long V;                       // vmn_pos of the verifier
    private PGroupElement g;
    private PGroupElementArray h;
    private PGroupElementArrayGPU hGPU;

    public PoSGPU(final String sid, final ProtocolElGamal protocol, final String rosid, final File nizkp, final GPUGroup group) {
        super(sid, protocol, rosid, nizkp);
        this.group = group;
    }

    private long create(final boolean prover) {
        final long[] out = new long[1];
        final RandomSourceBridge rs = prover
            ? new RandomSourceBridge.OverVCR(randomSource, group.pGroup.getElementOrder(), group.expBytes, rbitlen) : null;
        VMNException.check(VMNProofs.vmn_pos_create(group.grp, vbitlen(), ebitlen(), rbitlen, rs, out));
        return out[0];
    }

    @Override
    public void precompute(final Log log, final PGroupElement g, final PGroupElementArray h, final Permutation pi) {
        this.g = g;
        this.h = h;
        hGPU = PGroupElementArrayGPU.of(group, h);
        P = create(true);
        log.info("GPU path: permutation commitment kernels.");
        VMNException.check(VMNProofs.vmn_pos_precompute(P, group.encode(g), hGPU.handle, GPUArrays.gatherTable(pi)));
    }

    @Override
    public void prove(final Log log, final PGroupElement pkey, final PGroupElementArray w, final PGroupElementArray wp,
                      final PRingElementArray s) {
        log.info("GPU path: proving our shuffle.");
        final Log tempLog = log.newChildLog();
        final int width = Math.max(1, ((PPGroup) pkey.getPGroup()).project(0).getWidth());
        final PGroupElementArrayGPU[] W = GPUArrays.components(group, w, width);
        final PGroupElementArrayGPU[] WP = GPUArrays.components(group, wp, width);
        final PRingElementArrayGPU[] S = ProofSupport.columns(group, s, width);
        VMNException.check(VMNProofs.vmn_pos_set_instance(P, ProofSupport.wideKey(group, pkey, width), width,
                                                          GPUArrays.handles(W), GPUArrays.handles(WP), GPUArrays.handles(S)));
        // everything of commit() that does not need the batching vector runs while the instance is hashed below
        VMNException.check(VMNProofs.vmn_pos_commit_prepare(P));

        tempLog.info("GPU path: permutation commitment framed on the device and posted.");
        final PGroupElementArrayGPU u = new PGroupElementArrayGPU(group, VMNProofs.vmn_pos_permutation_commitment(P));
        final ByteTree uTree = ProofSupport.byteTree(u);
        bullBoard.publish("PermutationCommitment", uTree, tempLog);
        if (nizkp != null) {
            uTree.unsafeWriteTo(ProofSupport.file(nizkp, "PermutationCommitment", j));
        }

        tempLog.info("GPU path: seed of the batching vector from the random oracle; vector expanded on the device.");
        final ByteTreeContainer challengeData = new ByteTreeContainer(g.toByteTree(), h.toByteTree(), uTree, pkey.toByteTree(),
                                                                      w.toByteTree(), wp.toByteTree());
        final byte[] prgSeed = challenger.challenge(tempLog.newChildLog(), challengeData, 8 * prg.minNoSeedBytes(), rbitlen);

        tempLog.info("GPU path: commitment kernels.");
        VMNException.check(VMNProofs.vmn_pos_set_batch_vector_seed(P, prgSeed, prgSeed.length));
        final long[] msg = new long[1];
        VMNException.check(VMNProofs.vmn_pos_commit(P, msg));
        final GPUMessage commitment = new GPUMessage(msg[0]);
        final ByteTree commitmentTree = ProofSupport.byteTree(commitment);
        if (nizkp != null) {
            commitmentTree.unsafeWriteTo(ProofSupport.file(nizkp, "PoSCommitment", j));
        }
        tempLog.info("GPU path: commitment framed on the device and posted.");
        bullBoard.publish("Commitment", commitmentTree, tempLog);

        tempLog.info("GPU path: challenge from the random oracle.");
        final byte[] challengeBytes = challenger.challenge(tempLog.newChildLog(), new ByteTreeContainer(new ByteTree(prgSeed), commitmentTree),
                                                           vbitlen(), rbitlen);
        final byte[] v = LargeInteger.toPositive(challengeBytes).toByteArray();

        tempLog.info("GPU path: reply kernels.");
        VMNException.check(VMNProofs.vmn_pos_reply(P, v, v.length, msg));
        final GPUMessage reply = new GPUMessage(msg[0]);
        final ByteTree replyTree = ProofSupport.byteTree(reply);
        if (nizkp != null) {
            replyTree.unsafeWriteTo(ProofSupport.file(nizkp, "PoSReply", j));
        }
        tempLog.info("GPU path: reply framed on the device and posted.");
        bullBoard.publish("Reply", replyTree, tempLog);

        commitment.free();
        reply.free();
        VMNProofs.vmn_pos_free(P);
        P = 0;
        GPUArrays.free(W);
        GPUArrays.free(WP);
        for (final PRingElementArrayGPU col : S) {
            col.free();
        }
    }

    @Override
    public void precompute(final Log log, final PGroupElement g, final PGroupElementArray h) {
        this.g = g;
        this.h = h;
        hGPU = PGroupElementArrayGPU.of(group, h);
        V = create(false);
        VMNException.check(VMNProofs.vmn_pos_precompute(V, group.encode(g), hGPU.handle, null));
    }

    @Override
    public boolean verify(final Log log, final int l, final PGroupElement pkey, final PGroupElementArray w, final PGroupElementArray wp) {
        log.info("GPU path: checking the proof of a shuffle by " + ui.getDescrString(l) + ".");
        final Log tempLog = log.newChildLog();
        final int width = Math.max(1, ((PPGroup) pkey.getPGroup()).project(0).getWidth());
        final int n = h.size();
        final PGroupElementArrayGPU[] W = GPUArrays.components(group, w, width);
        final PGroupElementArrayGPU[] WP = GPUArrays.components(group, wp, width);
        VMNException.check(VMNProofs.vmn_pos_set_instance(V, ProofSupport.wideKey(group, pkey, width), width,
                                                          GPUArrays.handles(W), GPUArrays.handles(WP), null));

        tempLog.info("GPU path: parsing the permutation commitment.");
        final ByteTreeReader ur = bullBoard.waitFor(l, "PermutationCommitment", tempLog);
        PGroupElementArrayGPU u;
        try {
            final ByteBuffer ub = ProofSupport.direct(ur);
            u = PGroupElementArrayGPU.fromByteTree(group, ub, ub.remaining(), n);
            if (!u.isMember()) {
                u.free();
                u = ProofSupport.ones(group, n);
            }
        } catch (final VMNException e) {                       // malformed: the trivial commitment, PoSBasicTW.java:780-792
            u = ProofSupport.ones(group, n);
        } finally {
            ur.close();
        }
        VMNException.check(VMNProofs.vmn_pos_set_permutation_commitment(V, u.handle));
        final ByteTree uTree = ProofSupport.byteTree(u);
        if (nizkp != null) {
            uTree.unsafeWriteTo(ProofSupport.file(nizkp, "PermutationCommitment", l));
        }

        tempLog.info("GPU path: seed of the batching vector from the random oracle; vector expanded on the device.");
        final ByteTreeContainer challengeData = new ByteTreeContainer(g.toByteTree(), h.toByteTree(), uTree, pkey.toByteTree(),
                                                                      w.toByteTree(), wp.toByteTree());
        final byte[] prgSeed = challenger.challenge(tempLog.newChildLog(), challengeData, 8 * prg.minNoSeedBytes(), rbitlen);
        VMNException.check(VMNProofs.vmn_pos_set_batch_vector_seed(V, prgSeed, prgSeed.length));

        tempLog.info("GPU path: multi-exponentiations over the batching vector.");
        VMNException.check(VMNProofs.vmn_pos_compute_af(V));      // in parallel with the prover computing the rest of the proof

        tempLog.info("GPU path: parsing the commitment (range and membership checks on the device).");
        final ByteTreeReader cr = bullBoard.waitFor(l, "Commitment", tempLog);
        final ByteBuffer cb = ProofSupport.direct(cr);
        cr.close();
        GPUMessage commitment = GPUMessage.parse(group, cb, cb.remaining(), COM, new long[] {n, 1, n, 1, 1, 2L * width});
        boolean malformed = commitment == null;
        if (malformed) {
            commitment = ProofSupport.trivialPoSCommitment(group, n, width);          // PoSBasicTW.java:794-815
        }
        try {
            VMNException.check(VMNProofs.vmn_pos_set_commitment(V, commitment.handle));
        } catch (final VMNException e) {
            if (!e.isFormat()) {
                throw e;
            }
            malformed = true;
            commitment.free();
            commitment = ProofSupport.trivialPoSCommitment(group, n, width);
            VMNException.check(VMNProofs.vmn_pos_set_commitment(V, commitment.handle));
        }
        final ByteTree commitmentTree = ProofSupport.byteTree(commitment);
        if (nizkp != null) {
            commitmentTree.unsafeWriteTo(ProofSupport.file(nizkp, "PoSCommitment", l));
        }

        tempLog.info("GPU path: challenge from the random oracle.");
        final byte[] challengeBytes = challenger.challenge(tempLog.newChildLog(), new ByteTreeContainer(new ByteTree(prgSeed), commitmentTree),
                                                           vbitlen(), rbitlen);
        final byte[] v = LargeInteger.toPositive(challengeBytes).toByteArray();
        VMNException.check(VMNProofs.vmn_pos_set_challenge(V, v, v.length));

        tempLog.info("GPU path: parsing the reply (range checks on the device).");
        final ByteTreeReader rr = bullBoard.waitFor(l, "Reply", tempLog);
        final ByteBuffer rb = ProofSupport.direct(rr);
        rr.close();
        tempLog.info("GPU path: evaluating the verification equations.");
        final GPUMessage reply = GPUMessage.parse(group, rb, rb.remaining(), REP, new long[] {1, n, 1, 1, n, width});
        boolean verdict = false;
        if (reply != null && !malformed) {                      // a malformed reply is rejected (PoSBasicTW.java:985-989)
            final int[] out = new int[1];
            VMNException.check(VMNProofs.vmn_pos_verify(V, reply.handle, out, null));
            verdict = out[0] != 0;
            if (verdict && nizkp != null) {
                ProofSupport.byteTree(reply).unsafeWriteTo(ProofSupport.file(nizkp, "PoSReply", l));
            }
        }
        tempLog.info(verdict ? "GPU path: proof accepted." : "GPU path: proof rejected.");
        if (reply != null) {
            reply.free();
        }
        commitment.free();
        VMNProofs.vmn_pos_free(V);
        V = 0;
        u.free();
        GPUArrays.free(W);
        GPUArrays.free(WP);
        return verdict;
    }

    @Override
    public void free() {
        if (P != 0) {
            VMNProofs.vmn_pos_free(P);
            P = 0;
        }
        if (V != 0) {
            VMNProofs.vmn_pos_free(V);
            V = 0;
        }
        if (hGPU != null) {
            hGPU.free();
            hGPU = null;
        }
    }
}
