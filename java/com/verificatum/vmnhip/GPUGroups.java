package com.verificatum.vmnhip;

import java.util.HashMap;
import java.util.Map;

import com.verificatum.arithm.PGroup;

/** One {@link GPUGroup} (context + group + fixed-base tables) per (thread, device, group): the reference runs one
 *  protocol thread per party, and k parties may share a JVM in the demo (src/java/com/verificatum/protocol/demo/Demo.java:282-291);
 *  contexts are not shared between threads. */
final class GPUGroups {
    private static final ThreadLocal<Map<String, GPUGroup>> CACHE = ThreadLocal.withInitial(HashMap::new);

    private GPUGroups() { }

    static GPUGroup of(final int device, final PGroup pGroup) {
        final String key = device + "/" + pGroup.toString();
        return CACHE.get().computeIfAbsent(key, k -> new GPUGroup(device, pGroup));
    }
}
