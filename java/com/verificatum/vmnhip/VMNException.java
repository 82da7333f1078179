package com.verificatum.vmnhip;

/** A negative status of the C ABI (include/vmnhip.h: VMN_ERR_*), with the library's message for the calling thread.
 *  Unchecked, like the reference's ProtocolError for misuse (src/java/com/verificatum/protocol/hvzk/PoSBasicTW.java:842-845);
 *  {@link #isFormat()} marks malformed / out-of-range input, which callers catch and replace by trivial values the way
 *  the reference catches ArithmFormatException (PoSBasicTW.java:794-815). */
public final class VMNException extends RuntimeException {
    private static final long serialVersionUID = 1L;
    public static final int ERR_ARG = -1, ERR_DEVICE = -2, ERR_NOMEM = -3, ERR_FORMAT = -4, ERR_UNSUPPORTED = -5;
    private final int status;

    public VMNException(final int status) {
        super("vmnhip status " + status + ": " + VMNHip.vmn_last_error());
        this.status = status;
    }

    public int status() {
        return status;
    }

    public boolean isFormat() {
        return status == ERR_FORMAT;
    }

    /** Throws unless the status is VMN_OK. */
    public static void check(final int status) {
        if (status != 0) {
            throw new VMNException(status);
        }
    }
}
