#!/usr/bin/env python3
"""What one small library call costs on the host and on the device, untraced: draws of random ring arrays (the prover's N-sized
draws are made of them), element-wise products, single-element uploads.  GPU box only.  usage: python tools/call_latency_probe.py [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
vmn = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ctx = vmn.Context(0)
G = vmn.ECqPGroup(ctx, "P-256")
seed = bytes(range(32))


def timed(label, f, reps=200):
    f(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    print(f"{label:60s} issue {(t1 - t0) / reps * 1e6:8.1f} us/call   issue + drain {(t2 - t0) / reps * 1e6:8.1f} us/call")


keep = []
timed(f"ringArrayFromPRG n={n} 256 bits (one part)", lambda: G.ringArrayFromPRG(seed, n, 256).free())
timed(f"ringArrayFromPRG n={n} 356 bits (two parts)", lambda: G.ringArrayFromPRG(seed, n, 356).free())
timed(f"ringArrayFromPRG n={n} 612 bits (three parts)", lambda: G.ringArrayFromPRG(seed, n, 612).free())
A = G.ringArrayFromPRG(seed, n, 256)
B = G.ringArrayFromPRG(seed[::-1], n, 256)
timed(f"ring array product n={n}", lambda: A.mul(B).free())
X = G.exp(G.g, A)
timed(f"point array sum n={n}", lambda: X.mul(X).free())
timed(f"scalar exponent on an array (upload of 32 bytes) n={n}", lambda: X.exp(12345678901234567890).free(), reps=50)
