package com.verificatum.vmnhip;

import java.io.File;
import java.nio.ByteBuffer;

import com.verificatum.arithm.LargeInteger;
import com.verificatum.arithm.PGroupElement;
import com.verificatum.arithm.PGroupElementArray;
import com.verificatum.arithm.PPGroup;
import com.verificatum.arithm.PRingElement;
import com.verificatum.arithm.PRingElementArray;
import com.verificatum.arithm.Permutation;
import com.verificatum.eio.ByteTree;
import com.verificatum.eio.ByteTreeContainer;
import com.verificatum.eio.ByteTreeReader;
import com.verificatum.protocol.elgamal.ProtocolElGamal;
import com.verificatum.protocol.hvzk.CCPoS;
import com.verificatum.ui.Log;

/** Drop-in for {@code CCPoSW} (src/java/com/verificatum/protocol/hvzk/CCPoSW.java:75-158 prove, 161-265 verify) over
 *  vmn_ccpos_* (include/vmnproofs.h): plain form when {@code raisedExponent == null}, the raised single-equation form
 *  otherwise (CCPoSBasicW.java:493-506, 571-580).  The commitment is exported by a helper thread as in the reference
 *  (:114-123): it announces itself with vmn_ctx_helper_begin, so its byte-tree export runs on the helper lane. */
public final class CCPoSGPU extends ProtocolElGamal implements CCPoS {
    private static final int[] COM = {GPUMessage.ELEMENTS, GPUMessage.ELEMENTS};
    private static final int[] REP = {GPUMessage.RING, GPUMessage.RING, GPUMessage.RARRAY};
    private final GPUGroup group;

    public CCPoSGPU(final String sid, final ProtocolElGamal protocol, final String rosid, final File nizkp, final GPUGroup group) {
        super(sid, protocol, rosid, nizkp);
        this.group = group;
    }

    private long create(final boolean prover) {
        final long[] out = new long[1];
        final RandomSourceBridge rs = prover
            ? new RandomSourceBridge.OverVCR(randomSource, group.pGroup.getElementOrder(), group.expBytes, rbitlen) : null;
        VMNException.check(VMNProofs.vmn_ccpos_create(group.grp, vbitlen(), ebitlen(), rbitlen, rs, out));
        return out[0];
    }

    private Thread export(final ByteTree tree, final File file) {
        if (nizkp == null) {
            return null;
        }
        final Thread t = new Thread() {
                @Override
                public void run() {
                    VMNException.check(VMNHip.vmn_ctx_helper_begin(group.ctx));
                    try {
                        tree.unsafeWriteTo(file);
                    } finally {
                        VMNHip.vmn_ctx_helper_end(group.ctx);
                    }
                }
            };
        VMNException.check(VMNHip.vmn_ctx_helper_mark(group.ctx));
        t.start();
        return t;
    }

    private static void join(final Thread t) {
        if (t != null) {
            try {
                t.join();
            } catch (final InterruptedException ie) {
                Thread.currentThread().interrupt();
            }
        }
    }

    private byte[] seed(final Log log, final PGroupElement g, final PGroupElementArray h, final PGroupElementArray u,
                        final PGroupElement pkey, final PGroupElementArray w, final PGroupElementArray wp) {
        final ByteTreeContainer challengeData = new ByteTreeContainer(g.toByteTree(), h.toByteTree(), u.toByteTree(), pkey.toByteTree(),
                                                                      w.toByteTree(), wp.toByteTree());
        return challenger.challenge(log.newChildLog(), challengeData, 8 * prg.minNoSeedBytes(), rbitlen);
    }

    @Override
    public void prove(final Log log, final PGroupElement g, final PGroupElementArray h, final PGroupElementArray u,
                      final PGroupElement pkey, final PGroupElementArray w, final PGroupElementArray wp, final PRingElementArray r,
                      final Permutation pi, final PRingElementArray s) {
        log.info("GPU path: proving our shuffle.");
        final Log tempLog = log.newChildLog();
        final int width = Math.max(1, ((PPGroup) pkey.getPGroup()).project(0).getWidth());
        final PGroupElementArrayGPU H = PGroupElementArrayGPU.of(group, h);
        final PGroupElementArrayGPU U = PGroupElementArrayGPU.of(group, u);
        final PGroupElementArrayGPU[] W = GPUArrays.components(group, w, width);
        final PGroupElementArrayGPU[] WP = GPUArrays.components(group, wp, width);
        final PRingElementArrayGPU R = PRingElementArrayGPU.of(group, r);
        final PRingElementArrayGPU[] S = ProofSupport.columns(group, s, width);
        final long P = create(true);
        VMNException.check(VMNProofs.vmn_ccpos_set_instance(P, group.encode(g), H.handle, U.handle, ProofSupport.wideKey(group, pkey, width),
                                                            width, GPUArrays.handles(W), GPUArrays.handles(WP), R.handle,
                                                            GPUArrays.gatherTable(pi), GPUArrays.handles(S)));
        VMNException.check(VMNProofs.vmn_ccpos_commit_prepare(P));         // A', B': beside the hashing of the instance

        tempLog.info("GPU path: seed of the batching vector from the random oracle; vector expanded on the device.");
        final byte[] prgSeed = seed(tempLog, g, h, u, pkey, w, wp);
        VMNException.check(VMNProofs.vmn_ccpos_set_batch_vector_seed(P, prgSeed, prgSeed.length));

        tempLog.info("GPU path: commitment kernels.");
        final long[] msg = new long[1];
        VMNException.check(VMNProofs.vmn_ccpos_commit(P, msg));
        final GPUMessage commitment = new GPUMessage(msg[0]);
        final ByteTree commitmentTree = ProofSupport.byteTree(commitment);
        final Thread exportThread = export(commitmentTree, ProofSupport.file(nizkp, "CCPoSCommitment", j));
        tempLog.info("GPU path: commitment framed on the device and posted.");
        bullBoard.publish("Commitment", commitmentTree, tempLog);

        tempLog.info("GPU path: challenge from the random oracle.");
        final byte[] challengeBytes = challenger.challenge(tempLog.newChildLog(), new ByteTreeContainer(new ByteTree(prgSeed), commitmentTree),
                                                           vbitlen(), rbitlen);
        final byte[] v = LargeInteger.toPositive(challengeBytes).toByteArray();

        tempLog.info("GPU path: reply kernels.");
        VMNException.check(VMNProofs.vmn_ccpos_reply(P, v, v.length, msg));
        final GPUMessage reply = new GPUMessage(msg[0]);
        final ByteTree replyTree = ProofSupport.byteTree(reply);
        if (nizkp != null) {
            replyTree.unsafeWriteTo(ProofSupport.file(nizkp, "CCPoSReply", j));
        }
        tempLog.info("GPU path: reply framed on the device and posted.");
        bullBoard.publish("Reply", replyTree, tempLog);

        join(exportThread);
        commitment.free();
        reply.free();
        VMNProofs.vmn_ccpos_free(P);
        H.free();
        U.free();
        R.free();
        GPUArrays.free(W);
        GPUArrays.free(WP);
        for (final PRingElementArrayGPU col : S) {
            col.free();
        }
    }

    @Override
    public boolean verify(final Log log, final int l, final PGroupElement g, final PGroupElementArray h, final PGroupElementArray u,
                          final PGroupElement pkey, final PGroupElementArray w, final PGroupElementArray wp,
                          final PGroupElementArray raisedu, final PGroupElementArray raisedh, final PRingElement raisedExponent) {
        log.info("GPU path: checking the proof of a shuffle by " + ui.getDescrString(l) + ".");
        final Log tempLog = log.newChildLog();
        final int width = Math.max(1, ((PPGroup) pkey.getPGroup()).project(0).getWidth());
        final int n = h.size();
        final PGroupElementArrayGPU H = PGroupElementArrayGPU.of(group, h);
        final PGroupElementArrayGPU U = PGroupElementArrayGPU.of(group, u);
        final PGroupElementArrayGPU[] W = GPUArrays.components(group, w, width);
        final PGroupElementArrayGPU[] WP = GPUArrays.components(group, wp, width);
        final PGroupElementArrayGPU RU = raisedu == null ? null : PGroupElementArrayGPU.of(group, raisedu);
        final PGroupElementArrayGPU RH = raisedh == null ? null : PGroupElementArrayGPU.of(group, raisedh);
        final long V = create(false);
        VMNException.check(VMNProofs.vmn_ccpos_set_instance(V, group.encode(g), H.handle, U.handle, ProofSupport.wideKey(group, pkey, width),
                                                            width, GPUArrays.handles(W), GPUArrays.handles(WP), 0, null, null));

        tempLog.info("GPU path: seed of the batching vector from the random oracle; vector expanded on the device.");
        final byte[] prgSeed = seed(tempLog, g, h, u, pkey, w, wp);
        VMNException.check(VMNProofs.vmn_ccpos_set_batch_vector_seed(V, prgSeed, prgSeed.length));

        tempLog.info("GPU path: multi-exponentiations over the batching vector.");
        VMNException.check(VMNProofs.vmn_ccpos_compute_ab(V, raisedExponent == null || RU == null ? 0 : RU.handle));

        tempLog.info("GPU path: parsing the commitment (range and membership checks on the device).");
        final ByteTreeReader cr = bullBoard.waitFor(l, "Commitment", tempLog);
        final ByteBuffer cb = ProofSupport.direct(cr);
        cr.close();
        GPUMessage commitment = GPUMessage.parse(group, cb, cb.remaining(), COM, new long[] {1, 2L * width});
        boolean malformed = commitment == null;
        if (!malformed) {
            try {
                VMNException.check(VMNProofs.vmn_ccpos_set_commitment(V, commitment.handle));
            } catch (final VMNException e) {
                if (!e.isFormat()) {
                    throw e;
                }
                malformed = true;
                commitment.free();
            }
        }
        if (malformed) {                                          // CCPoSBasicW.java:414-426
            commitment = ProofSupport.trivialCCPoSCommitment(group, width);
            VMNException.check(VMNProofs.vmn_ccpos_set_commitment(V, commitment.handle));
        }
        final ByteTree commitmentTree = ProofSupport.byteTree(commitment);
        final Thread exportThread = export(commitmentTree, ProofSupport.file(nizkp, "CCPoSCommitment", l));

        tempLog.info("GPU path: challenge from the random oracle.");
        final byte[] challengeBytes = challenger.challenge(tempLog.newChildLog(), new ByteTreeContainer(new ByteTree(prgSeed), commitmentTree),
                                                           vbitlen(), rbitlen);
        final byte[] v = LargeInteger.toPositive(challengeBytes).toByteArray();
        VMNException.check(VMNProofs.vmn_ccpos_set_challenge(V, v, v.length));

        tempLog.info("GPU path: parsing the reply (range checks on the device).");
        final ByteTreeReader rr = bullBoard.waitFor(l, "Reply", tempLog);
        final ByteBuffer rb = ProofSupport.direct(rr);
        rr.close();
        tempLog.info("GPU path: evaluating the verification equations.");
        final GPUMessage reply = GPUMessage.parse(group, rb, rb.remaining(), REP, new long[] {1, width, n});
        boolean verdict = false;
        if (reply != null && !malformed) {                        // malformed reply: false (CCPoSBasicW.java:533-544)
            final int[] out = new int[1];
            final boolean raised = raisedExponent != null && RH != null;
            final byte[] rho = raised ? raisedExponent.toLargeInteger().toByteArray() : null;
            VMNException.check(VMNProofs.vmn_ccpos_verify(V, reply.handle, raised ? RH.handle : 0, rho, raised ? rho.length : 0, out));
            verdict = out[0] != 0;
            if (nizkp != null) {
                ProofSupport.byteTree(reply).unsafeWriteTo(ProofSupport.file(nizkp, "CCPoSReply", l));
            }
        }
        tempLog.info(verdict ? "GPU path: proof accepted." : "GPU path: proof rejected.");
        join(exportThread);
        if (reply != null) {
            reply.free();
        }
        commitment.free();
        VMNProofs.vmn_ccpos_free(V);
        H.free();
        U.free();
        if (RU != null) {
            RU.free();
        }
        if (RH != null) {
            RH.free();
        }
        GPUArrays.free(W);
        GPUArrays.free(WP);
        return verdict;
    }
}
