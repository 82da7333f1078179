package com.verificatum.vmnhip;

import java.io.File;

import com.verificatum.protocol.elgamal.ProtocolElGamal;
import com.verificatum.protocol.hvzk.CCPoS;
import com.verificatum.protocol.hvzk.CCPoSFactory;

/** Factory of {@link CCPoSGPU}: replaces the hard-wired factory of the reference at
 *  src/java/com/verificatum/protocol/mixnet/ShufflerElGamal.java:120-124 (copied per session at
 *  ShufflerElGamalSession.java:145-150).  One GPU group per party (per protocol thread). */
public final class CCPoSGPUFactory implements CCPoSFactory {
    private final int device;

    public CCPoSGPUFactory(final int device) {
        this.device = device;
    }

    @Override
    public CCPoS newCCPoS(final String sid, final ProtocolElGamal protocol, final String rosid, final File nizkp) {
        return new CCPoSGPU(sid, protocol, rosid, nizkp, GPUGroups.of(device, protocol.getPGroup()));
    }
}
