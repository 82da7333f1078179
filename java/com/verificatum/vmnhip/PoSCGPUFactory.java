package com.verificatum.vmnhip;

import java.io.File;

import com.verificatum.protocol.elgamal.ProtocolElGamal;
import com.verificatum.protocol.hvzk.PoSC;
import com.verificatum.protocol.hvzk.PoSCFactory;

/** Factory of {@link PoSCGPU}: replaces the hard-wired factory of the reference at
 *  src/java/com/verificatum/protocol/mixnet/ShufflerElGamal.java:120-124 (copied per session at
 *  ShufflerElGamalSession.java:145-150).  One GPU group per party (per protocol thread). */
public final class PoSCGPUFactory implements PoSCFactory {
    private final int device;

    public PoSCGPUFactory(final int device) {
        this.device = device;
    }

    @Override
    public PoSC newPoSC(final String sid, final ProtocolElGamal protocol, final String rosid, final File nizkp) {
        return new PoSCGPU(sid, protocol, rosid, nizkp, GPUGroups.of(device, protocol.getPGroup()));
    }
}
