// Host-only fuzz driver of the `.onnx` reader (rover-slam_amd/csrc/onnx_load.hip), built by tests/test_onnx_fuzz.py with AddressSanitizer +
// UBSan (no GPU code in that file; GPU sanitizers are not available on the pool, the CPU build is where the reader is sanitised).
//   onnx_fuzz_driver <seed.onnx> <kind 1|2> <script> <scratch-file>
// script: one mutation per line, applied to a fresh copy of the seed --
//   T pos            truncate to pos bytes
//   F pos val        byte at pos := val
//   L pos v          write the 10-byte varint of the 64-bit value v at pos (a length field that lies)
//   V pos v          write the minimal varint of v at pos
//   I pos n seed     insert n pseudo-random bytes at pos
//   R path           replace the whole file by the bytes of `path` (hand-built reproducers)
// Every case is converted twice (hyper-parameters + weights, weights only) under alarm(): a spinning reader is a failure with the case named.
// Output: one line per case "<index> <rc> <rc_weights_only>", then "done <cases> <refused>".
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <string>
#include <vector>
#include "../../include/rover_fe.h"

// the two C-ABI helpers rfe_k_onnx_convert leans on live in rfe_api.hip (GPU code); the driver links the reader alone
extern "C" rfe_hparams rfe_default_hparams(void) { rfe_hparams h; memset(&h, 0, sizeof h); return h; }
extern "C" int64_t rfe_weight_count(int kind) { return kind == RFE_KIND_SUPERPOINT ? 1300865 : kind == RFE_KIND_LIGHTGLUE ? 11321153 : -1; }
extern "C" int rfe_k_onnx_convert(const char* path, int kind, int weights_only, float* blob, rfe_hparams* hp, char* err, int errlen);

static volatile int g_case = -1;
static void on_alarm(int) {
    char b[96];
    const int n = snprintf(b, sizeof b, "HANG case %d\n", (int)g_case);
    if (write(1, b, (size_t)n) < 0) {}
    _exit(3);
}

static std::vector<unsigned char> slurp(const char* p) {
    std::vector<unsigned char> d;
    FILE* f = fopen(p, "rb");
    if (!f) return d;
    unsigned char c[1 << 16];
    size_t g;
    while ((g = fread(c, 1, sizeof c, f)) > 0) d.insert(d.end(), c, c + g);
    fclose(f);
    return d;
}

int main(int argc, char** argv) {
    if (argc != 5) { fprintf(stderr, "usage: %s seed.onnx kind script scratch\n", argv[0]); return 2; }
    const std::vector<unsigned char> seed = slurp(argv[1]);
    const int kind = atoi(argv[2]);
    if (seed.empty()) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
    FILE* sc = fopen(argv[3], "r");
    if (!sc) { fprintf(stderr, "cannot read %s\n", argv[3]); return 2; }
    std::vector<float> blob((size_t)rfe_weight_count(kind));
    signal(SIGALRM, on_alarm);
    char line[4096];
    int cases = 0, refused = 0;
    while (fgets(line, sizeof line, sc)) {
        std::vector<unsigned char> d = seed;
        char op = 0;
        unsigned long long a = 0, b = 0, c = 0;
        char path[3000];
        if (sscanf(line, " %c", &op) != 1) continue;
        if (op == 'R') {
            if (sscanf(line, " R %2999s", path) != 1) continue;
            d = slurp(path);
        } else {
            const int got = sscanf(line, " %c %llu %llu %llu", &op, &a, &b, &c);
            if (got < 2) continue;
            if (a > d.size()) a = d.size();
            if (op == 'T') d.resize((size_t)a);
            else if (op == 'F') { if (a < d.size()) d[(size_t)a] = (unsigned char)b; }
            else if (op == 'L' || op == 'V') {
                unsigned char v[10]; int n = 0; uint64_t x = b;
                if (op == 'L') { for (n = 0; n < 10; ++n) { v[n] = (unsigned char)((x & 0x7F) | (n < 9 ? 0x80 : 0)); x >>= 7; } }
                else { do { v[n] = (unsigned char)(x & 0x7F); x >>= 7; if (x) v[n] |= 0x80; ++n; } while (x); }
                for (int q = 0; q < n && a + q < d.size(); ++q) d[(size_t)a + q] = v[q];
            } else if (op == 'I') {
                std::vector<unsigned char> ins((size_t)b);
                uint64_t s = c * 6364136223846793005ull + 1442695040888963407ull;
                for (auto& x : ins) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = (unsigned char)(s >> 56); }
                d.insert(d.begin() + (ptrdiff_t)a, ins.begin(), ins.end());
            } else continue;
        }
        FILE* o = fopen(argv[4], "wb");
        if (!o) { fprintf(stderr, "cannot write %s\n", argv[4]); return 2; }
        if (!d.empty() && fwrite(d.data(), 1, d.size(), o) != d.size()) { fclose(o); return 2; }
        fclose(o);
        g_case = cases;
        char err[1024];
        int rc[2];
        for (int wo = 0; wo < 2; ++wo) {
            rfe_hparams hp = rfe_default_hparams();
            err[0] = 0;
            alarm(30);
            rc[wo] = rfe_k_onnx_convert(argv[4], kind, wo, blob.data(), &hp, err, (int)sizeof err);
            alarm(0);
            if (rc[wo] != RFE_OK && rc[wo] != RFE_ERR_IO) { printf("BADRC case %d rc %d\n", cases, rc[wo]); return 4; }
            if (rc[wo] == RFE_ERR_IO && !err[0]) { printf("NOREASON case %d\n", cases); return 5; }
        }
        refused += rc[0] != RFE_OK;
        printf("%d %d %d\n", cases, rc[0], rc[1]);
        ++cases;
    }
    fclose(sc);
    printf("done %d %d\n", cases, refused);
    return 0;
}
