package com.verificatum.vmnhip;

import java.io.File;

import com.verificatum.protocol.elgamal.ProtocolElGamal;
import com.verificatum.protocol.hvzk.PoS;
import com.verificatum.protocol.hvzk.PoSFactory;

/** Factory of {@link PoSGPU}: replaces the hard-wired factory of the reference at
 *  src/java/com/verificatum/protocol/mixnet/ShufflerElGamal.java:120-124 (copied per session at
 *  ShufflerElGamalSession.java:145-150).  One GPU group per party (per protocol thread). */
public final class PoSGPUFactory implements PoSFactory {
    private final int device;

    public PoSGPUFactory(final int device) {
        this.device = device;
    }

    @Override
    public PoS newPoS(final String sid, final ProtocolElGamal protocol, final String rosid, final File nizkp) {
        return new PoSGPU(sid, protocol, rosid, nizkp, GPUGroups.of(device, protocol.getPGroup()));
    }
}
