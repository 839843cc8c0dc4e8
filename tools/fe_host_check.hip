// tools/fe_host_check.hip -- the fused-ladder arithmetic (csrc/fe26.h, csrc/fe28.h) compiled for the HOST and run
// against the CPU oracle (oracle/liboracle.so, test infrastructure).  The device kernels wrap the very functions
// called here (x25519_fe26_one, x448_fe28_one), so a change to the limb arithmetic / ladder step can be checked in
// the build container, without a GPU, before it goes to the box.  Usage: fe_host_check [n]  (exit code 0 = all equal)
//   hipcc -O2 -std=c++17 tools/fe_host_check.hip -o /tmp/fe_host_check -Loracle -l:liboracle.so -Wl,-rpath,$PWD/oracle
#define MA_DEV __host__ __device__ inline
#include "../modarith_amd/csrc/fe26.h"
#include "../modarith_amd/csrc/fe28.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

extern "C" void rfc7748_X25519(const char* bk, const char* bu, char* bv);
extern "C" void rfc7748_X448(const char* bk, const char* bu, char* bv);

static uint64_t sm_state = 0x1234567;
static uint64_t sm() {
    uint64_t z = (sm_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

template <int NW, class Fn, class Ref>
static int run(const char* name, int n, Fn fn, Ref ref) {
    int bad = 0;
    for (int it = 0; it < n; it++) {
        uint64_t k[NW], u[NW], got[NW], want[NW];
        for (int i = 0; i < NW; i++) { k[i] = sm(); u[i] = sm(); }
        // corner records first: u = 0, 1, all ones, p-1 .. p+1 region (top words all ones), scalar all zeros / all ones
        if (it == 0) memset(u, 0, sizeof u);
        if (it == 1) { memset(u, 0, sizeof u); u[0] = 1; }
        if (it == 2) memset(u, 0xff, sizeof u);
        if (it == 3) { memset(u, 0xff, sizeof u); u[0] = ~(uint64_t)18; if (NW == 4) u[3] >>= 1; }
        if (it == 4) { memset(u, 0xff, sizeof u); u[0] = ~(uint64_t)19; if (NW == 4) u[3] >>= 1; }
        if (it == 5) memset(k, 0, sizeof k);
        if (it == 6) memset(k, 0xff, sizeof k);
        if (it == 7) { memset(u, 0, sizeof u); u[0] = (NW == 4) ? 9 : 5; }
        fn(k, u, got);
        ref((const char*)k, (const char*)u, (char*)want);
        if (memcmp(got, want, sizeof got) != 0) {
            if (bad < 4) printf("%s: record %d differs\n", name, it);
            bad++;
        }
    }
    printf("%s: %d records, %d differ from the oracle\n", name, n, bad);
    return bad;
}

int main(int argc, char** argv) {
    int n = argc > 1 ? atoi(argv[1]) : 2000;
    int bad = run<4>("x25519_fe26_one", n, [](const uint64_t* k, const uint64_t* u, uint64_t* o) { ma::x25519_fe26_one(k, u, o); }, rfc7748_X25519);
    bad += run<7>("x448_fe28_one", n / 4 + 8, [](const uint64_t* k, const uint64_t* u, uint64_t* o) { ma::x448_fe28_one(k, u, o); }, rfc7748_X448);
    return bad ? 1 : 0;
}
