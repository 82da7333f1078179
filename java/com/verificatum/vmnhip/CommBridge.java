package com.verificatum.vmnhip;

/** The one exchange a sharded proof needs (vmn_comm, include/vmnproofs.h): a fixed-size all-gather between the processes
 *  of one party, one per GPU (RCCL over xGMI under a small native helper, or any transport the party already has between
 *  its processes).  Payloads are a few hundred bytes: partial products, partial sums, scan carries, verdict bits. */
public interface CommBridge {
    int rank();

    int world();

    /** Every rank contributes {@code mine.length} bytes; the result is world * mine.length bytes in rank order. */
    byte[] allGather(byte[] mine);
}
