// Sanitizer harness of the host-side file I/O (hostio.hip + hosterr.hip built with -fsanitize=address,undefined by `make asan`).
// Test infrastructure: tests/test_host_fuzz.py generates the malformed corpus and drives this binary; nothing here ships.
//
//   hostio_fuzz wav <file>...                 probe + decode each file like frontend.load_wav_16k does; one line per file:
//                                             "<frames-or-error> <sample_rate> <channels> <fnv1a of the decoded floats>"
//   hostio_fuzz mutate <seed.wav> <n> <seed> <scratch>   n random mutations of a valid file (byte flips in the first 64 bytes, 32-bit
//                                             fields set to edge values, truncations), each decoded in process
//   hostio_fuzz pt <out.pt> <rows> <cols>     ser_pt_write_f32 of a deterministic matrix
#include "../../include/ser_hip.h"
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

static uint64_t fnv(const void* p, size_t n) {
    const unsigned char* b = (const unsigned char*)p;
    uint64_t h = 1469598103934665603ULL;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ULL; }
    return h;
}

static int decode(const char* path, bool print) {
    int32_t sr = -1, ch = -1;
    const int64_t n = ser_wav_read_f32(path, nullptr, 0, &sr, &ch);
    if (n < 0) {
        if (print) printf("%lld %d %d 0 %s\n", (long long)n, sr, ch, ser_last_error());
        return 0;
    }
    std::vector<float> buf((size_t)n);                                   // EXACTLY n floats: one past the end is an ASan report
    const int64_t m = ser_wav_read_f32(path, buf.data(), n, &sr, &ch);
    if (n > 1) {                                                         // a buffer one frame short must be refused, not overrun
        std::vector<float> small((size_t)(n - 1));
        if (ser_wav_read_f32(path, small.data(), n - 1, nullptr, nullptr) != -6) { fprintf(stderr, "capacity check failed: %s\n", path); return 2; }
    }
    if (m != n) { fprintf(stderr, "probe said %lld frames, decode %lld: %s\n", (long long)n, (long long)m, path); return 2; }
    if (print) printf("%lld %d %d %016llx\n", (long long)n, sr, ch, (unsigned long long)fnv(buf.data(), buf.size() * 4));
    return 0;
}

static uint64_t rng_state;
static uint32_t rnd() {                                                  // xorshift64*
    rng_state ^= rng_state >> 12; rng_state ^= rng_state << 25; rng_state ^= rng_state >> 27;
    return (uint32_t)((rng_state * 2685821657736338717ULL) >> 32);
}

int main(int argc, char** argv) {
    if (argc >= 3 && !strcmp(argv[1], "wav")) {
        int rc = 0;
        for (int i = 2; i < argc; ++i) rc |= decode(argv[i], true);
        return rc;
    }
    if (argc == 6 && !strcmp(argv[1], "mutate")) {
        FILE* f = fopen(argv[2], "rb");
        if (!f) return 3;
        std::vector<unsigned char> seed;
        unsigned char tmp[4096];
        size_t k;
        while ((k = fread(tmp, 1, sizeof tmp, f)) > 0) seed.insert(seed.end(), tmp, tmp + k);
        fclose(f);
        const int n = atoi(argv[3]);
        rng_state = 0x9E3779B97F4A7C15ULL ^ (uint64_t)strtoull(argv[4], nullptr, 10);
        const std::string scratch = std::string(argv[5]) + "/mut.wav";
        static const uint32_t edge[] = {0u, 1u, 2u, 3u, 0x7fffu, 0x8000u, 0xffffu, 0x10000u, 0x7fffffffu, 0x80000000u, 0xfffffffeu, 0xffffffffu};
        int rc = 0;
        for (int it = 0; it < n && !rc; ++it) {
            std::vector<unsigned char> m = seed;
            const int edits = 1 + rnd() % 4;
            for (int e = 0; e < edits; ++e) {
                const uint32_t kind = rnd() % 4;
                const size_t head = m.size() < 64 ? m.size() : 64;
                if (!head) break;
                if (kind == 0) m[rnd() % head] = (unsigned char)rnd();
                else if (kind == 1 && head >= 4) { const size_t at = rnd() % (head - 3); const uint32_t v = edge[rnd() % 12]; memcpy(&m[at], &v, 4); }
                else if (kind == 2 && head >= 2) { const size_t at = rnd() % (head - 1); const uint16_t v = (uint16_t)edge[rnd() % 12]; memcpy(&m[at], &v, 2); }
                else m.resize(rnd() % (m.size() + 1));
            }
            f = fopen(scratch.c_str(), "wb");
            if (!f) return 3;
            if (!m.empty()) fwrite(m.data(), 1, m.size(), f);
            fclose(f);
            rc = decode(scratch.c_str(), false);
        }
        printf("mutations %d rc %d\n", n, rc);
        return rc;
    }
    if (argc == 5 && !strcmp(argv[1], "pt")) {
        const int64_t rows = atoll(argv[3]), cols = atoll(argv[4]);
        std::vector<float> x;
        if (rows > 0 && cols > 0 && rows < (1 << 24) && cols < (1 << 24) && rows * cols < (1 << 24)) {
            x.resize((size_t)(rows * cols));
            for (size_t i = 0; i < x.size(); ++i) x[i] = (float)((int)(i % 2001) - 1000) * 0.125f;
        }
        const int rc = ser_pt_write_f32(argv[2], x.empty() ? nullptr : x.data(), rows, cols);
        printf("%d %s\n", rc, rc ? ser_last_error() : "");
        return 0;
    }
    fprintf(stderr, "usage: hostio_fuzz wav <file>... | mutate <seed.wav> <n> <seed> <scratch dir> | pt <out> <rows> <cols>\n");
    return 64;
}
