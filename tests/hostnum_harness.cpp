// Test harness for verificatum-vmn_amd/csrc/hostnum64.h (the host-side scalars of the proof drivers): reads
// "n a b e" as hex words from argv and prints a*b mod n, a^e mod n, a^-1 mod n, (e mod n) as hex lines.
// Built and run by tests/test_hostnum.py against Python integers.
#include <stdio.h>
#include <string.h>

#include <string>

#include "../verificatum-vmn_amd/csrc/hostcurve.h"

using namespace vmn::num64;

static Bytes from_hex(const char* s) {
    std::string h(s);
    if (h.size() % 2) h = "0" + h;
    Bytes out;
    for (size_t i = 0; i < h.size(); i += 2) out.push_back((uint8_t)strtoul(h.substr(i, 2).c_str(), nullptr, 16));
    return out;
}
static void print(const Num& a, size_t nbytes) {
    Bytes b = to_bytes(a, nbytes);
    for (uint8_t c : b) printf("%02x", c);
    printf("\n");
}

static void print_bytes(const Bytes& b) {
    for (uint8_t c : b) printf("%02x", c);
    printf("\n");
}

// "ec p a b e": a, b = points as x||y hex (all ff = infinity); prints a^e, a*b, a^-1 in the same encoding
static int ec_main(char** argv) {
    Bytes pb = from_hex(argv[2]), a = from_hex(argv[3]), b = from_hex(argv[4]), e = from_hex(argv[5]);
    size_t nl = (pb.size() + 7) / 8;
    Mod F(from_be(pb.data(), pb.size(), nl));
    HostCurve C;
    C.F = &F;
    C.cb = pb.size();
    C.fl = nl;
    print_bytes(C.exp(a, e.data(), e.size()));
    print_bytes(C.add(a, b));
    print_bytes(C.negate(a));
    return 0;
}

// "jac n a [a ...]": the Jacobi symbols (a / n)
static int jac_main(int argc, char** argv) {
    Bytes nb = from_hex(argv[2]);
    size_t nl = (nb.size() + 7) / 8;
    Mod M(from_be(nb.data(), nb.size(), nl));
    for (int i = 3; i < argc; ++i) {
        Bytes ab = from_hex(argv[i]);
        printf("%d\n", M.jacobi(from_be(ab.data(), ab.size(), nl)));
    }
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 4 && !strcmp(argv[1], "jac")) return jac_main(argc, argv);
    if (argc == 6 && !strcmp(argv[1], "ec")) return ec_main(argv);
    if (argc != 5) return 2;
    Bytes nb = from_hex(argv[1]), ab = from_hex(argv[2]), bb = from_hex(argv[3]), eb = from_hex(argv[4]);
    size_t nl = (nb.size() + 7) / 8, nbytes = nb.size();
    Mod M(from_be(nb.data(), nb.size(), nl));
    Num a = from_be(ab.data(), ab.size(), nl), b = from_be(bb.data(), bb.size(), nl);
    print(M.mul(a, b), nbytes);
    print(M.pow(a, eb.data(), eb.size()), nbytes);
    print(M.inv(a), nbytes);
    print(M.reduce(eb.data(), eb.size()), nbytes);
    print(M.add(a, b), nbytes);
    print(M.neg(a), nbytes);
    return 0;
}
