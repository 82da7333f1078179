// sha256.h — SHA-256 for host and device: the hash under the reference's PRGHeuristic / RandomOracle
// (P/distr/IndependentGeneratorsRO.java:117-130, P/hvzk/PoSBasicTW.java:533-538, elgamal/ProtocolElGamal.java:357).
// Device use: one compression per PRG block (a 32-byte seed and a 4-byte counter fit one 64-byte block).
#pragma once
#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define VMN_HD __host__ __device__
#else
#define VMN_HD
#endif

namespace vmn {
namespace sha256 {

VMN_HD inline uint32_t rotr(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

// state' = compress(state, 16 big-endian message words)
VMN_HD inline void compress(uint32_t (&st)[8], const uint32_t (&msg)[16]) {
    const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
        0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
        0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
        0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
        0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
        0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = msg[i];
    for (int i = 16; i < 64; ++i) {
        uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
        uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; ++i) {
        uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = h + S1 + ch + K[i] + w[i];
        uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
VMN_HD inline void init(uint32_t (&st)[8]) {
    st[0] = 0x6a09e667; st[1] = 0xbb67ae85; st[2] = 0x3c6ef372; st[3] = 0xa54ff53a;
    st[4] = 0x510e527f; st[5] = 0x9b05688c; st[6] = 0x1f83d9ab; st[7] = 0x5be0cd19;
}

// PRGHeuristic block: digest = SHA-256(seed || uint32_be(counter)), seed given as 8 big-endian words (32 bytes)
VMN_HD inline void prg_block(uint32_t (&digest)[8], const uint32_t (&seed)[8], uint32_t counter) {
    uint32_t msg[16];
    for (int i = 0; i < 8; ++i) msg[i] = seed[i];
    msg[8] = counter;
    msg[9] = 0x80000000u;
    for (int i = 10; i < 15; ++i) msg[i] = 0;
    msg[15] = 36 * 8;
    init(digest);
    compress(digest, msg);
}

// plain host-side hashing of a byte string (RandomOracle, small inputs)
inline void hash(const uint8_t* data, size_t len, uint8_t (&out)[32]) {
    uint32_t st[8];
    init(st);
    uint32_t msg[16];
    size_t full = len / 64;
    for (size_t b = 0; b < full; ++b) {
        for (int i = 0; i < 16; ++i) {
            const uint8_t* p = data + b * 64 + 4 * i;
            msg[i] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
        }
        compress(st, msg);
    }
    uint8_t tail[128] = {0};
    size_t rem = len - full * 64;
    for (size_t i = 0; i < rem; ++i) tail[i] = data[full * 64 + i];
    tail[rem] = 0x80;
    size_t tl = rem + 9 <= 64 ? 64 : 128;
    uint64_t bits = (uint64_t)len * 8;
    for (int i = 0; i < 8; ++i) tail[tl - 1 - i] = (uint8_t)(bits >> (8 * i));
    for (size_t b = 0; b < tl / 64; ++b) {
        for (int i = 0; i < 16; ++i) {
            const uint8_t* p = tail + b * 64 + 4 * i;
            msg[i] = ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3];
        }
        compress(st, msg);
    }
    for (int i = 0; i < 8; ++i) {
        out[4 * i] = (uint8_t)(st[i] >> 24);
        out[4 * i + 1] = (uint8_t)(st[i] >> 16);
        out[4 * i + 2] = (uint8_t)(st[i] >> 8);
        out[4 * i + 3] = (uint8_t)st[i];
    }
}

}  // namespace sha256
}  // namespace vmn
