// Host-side file I/O of the extraction path, off the Python GIL (no device code in this file).
//
//   ser_wav_read_f32   row a5 : what librosa.load(path, sr=16000) -> soundfile does for a RIFF/WAVE file
//                               (preprocessing/preprocess_speech.py:47): integer PCM / 2^(bits-1), channels averaged.
//   ser_pt_write_f32   row a21: the file torch.save(feats, "<name>.pt") leaves for the downstream heads
//                               (preprocess_speech.py:69-71; read back by torch.load(path),
//                               bin/train_cat_bimodal_lazy_1head.py:227).
//
// Why native: the end-to-end rate of the driver was bounded by Python-level work under one GIL -- wave parsing and
// torch.save's pickler + zip writer -- not by the GPU.  Both calls are plain C, re-entrant, and
// are entered through ctypes.CDLL, which drops the GIL: the driver's worker threads decode and write in parallel.
//
// .pt layout written here (a valid torch zipfile archive, checked against torch.load in tests/test_host_logic.py):
//   <stem>/data.pkl   pickle protocol 2: torch._utils._rebuild_tensor_v2(FloatStorage '0' on 'cpu', numel), offset 0,
//                     size (rows, cols), stride (cols, 1), requires_grad False, OrderedDict())
//   <stem>/byteorder  "little"
//   <stem>/data/0     rows*cols little-endian fp32, payload aligned to 64 bytes (extra field "FB", as torch does)
//   <stem>/version    "3\n"
// stored (no compression), CRC-32 of every member in its local header and in the central directory.
// Plain C++ on purpose (no HIP header): `make asan` compiles this file and hosterr.hip with g++ -fsanitize=address,undefined into
// the fuzz harness tests/test_host_fuzz.py drives (malformed RIFF headers, short files, huge lengths) -- no GPU needed.
#include "../../include/ser_hip.h"
#include <stdint.h>
#include <errno.h>
#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <immintrin.h>

int ser_fail(int code, const char* fmt, ...);          // hosterr.hip

namespace {

struct Crc32 {
    uint32_t t[8][256];
    Crc32() {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; ++i)
            for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
    }
    // slice-by-8 (fallback, short inputs and tails): ~1 GB/s per thread
    uint32_t run(const void* data, size_t n, uint32_t crc = 0) const {
        const unsigned char* p = (const unsigned char*)data;
        crc = ~crc;
        while (n && ((uintptr_t)p & 7)) { crc = t[0][(crc ^ *p++) & 0xff] ^ (crc >> 8); --n; }
        while (n >= 8) {
            uint64_t v;
            memcpy(&v, p, 8);
            v ^= crc;
            crc = t[7][v & 0xff] ^ t[6][(v >> 8) & 0xff] ^ t[5][(v >> 16) & 0xff] ^ t[4][(v >> 24) & 0xff] ^
                  t[3][(v >> 32) & 0xff] ^ t[2][(v >> 40) & 0xff] ^ t[1][(v >> 48) & 0xff] ^ t[0][(v >> 56) & 0xff];
            p += 8;
            n -= 8;
        }
        while (n--) crc = t[0][(crc ^ *p++) & 0xff] ^ (crc >> 8);
        return ~crc;
    }
};
const Crc32& crc_table() {
    static const Crc32 c;          // thread-safe static initialisation
    return c;
}

// CRC-32 (IEEE 802.3, the zip polynomial) by carry-less multiplication: 64 bytes per iteration folded with PCLMULQDQ,
// then 128 -> 64 -> 32 bits and a Barrett reduction (Gopal et al., "Fast CRC Computation for Generic Polynomials Using
// PCLMULQDQ Instruction", Intel 2009; constants for the bit-reflected polynomial 0xEDB88320).  ~10x the table walk:
// the CRC of a 2 MB hidden state drops from ~2 ms to ~0.2 ms, which is what a writer thread spends per file.
// `len` must be a multiple of 16 and at least 64; `crc` is the running (pre-inverted) register.
__attribute__((target("pclmul,sse4.1")))
uint32_t crc32_clmul(const unsigned char* buf, size_t len, uint32_t crc) {
    static const uint64_t __attribute__((aligned(16))) k1k2[2] = {0x0154442bd4ULL, 0x01c6e41596ULL};
    static const uint64_t __attribute__((aligned(16))) k3k4[2] = {0x01751997d0ULL, 0x00ccaa009eULL};
    static const uint64_t __attribute__((aligned(16))) k5k0[2] = {0x0163cd6124ULL, 0x0000000000ULL};
    static const uint64_t __attribute__((aligned(16))) poly[2] = {0x01db710641ULL, 0x01f7011641ULL};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i*)k1k2);
    buf += 64;
    len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
        x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
        x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
        y6 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
        y8 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64;
        len -= 64;
    }
    x0 = _mm_load_si128((const __m128i*)k3k4);                       // fold the four lanes into one
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {                                                // remaining 16-byte blocks
        x2 = _mm_loadu_si128((const __m128i*)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16;
        len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);                           // 128 -> 64 bits
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i*)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i*)poly);                         // Barrett reduction to 32 bits
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}

uint32_t crc32_of(const void* data, size_t n) {
    static const bool fast = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    const unsigned char* p = (const unsigned char*)data;
    uint32_t reg = 0xffffffffu;                                        // running register (pre-inverted)
    if (fast && n >= 64) {
        const size_t body = n & ~(size_t)15;
        reg = crc32_clmul(p, body, reg);
        p += body;
        n -= body;
    }
    return crc_table().run(p, n, ~reg);                                // table walk for short inputs and the tail
}

void put16(std::string& s, uint32_t v) { s.push_back((char)(v & 0xff)); s.push_back((char)((v >> 8) & 0xff)); }
void put32(std::string& s, uint32_t v) { put16(s, v & 0xffff); put16(s, v >> 16); }

struct Member { std::string name; uint32_t crc, size, offset; };

// local file header (+ optional "FB" padding so that the payload starts on a 64-byte boundary)
std::string local_header(const std::string& name, uint32_t crc, uint32_t size, size_t at, bool align64) {
    std::string h;
    size_t extra = 0;
    if (align64) {
        const size_t payload = at + 30 + name.size() + 4;             // with an empty FB field
        extra = 4 + ((64 - (payload & 63)) & 63);
    }
    put32(h, 0x04034b50u); put16(h, 20); put16(h, 0); put16(h, 0);    // version, flags, method = stored
    put16(h, 0); put16(h, 0x21);                                      // time, date (1980-01-01)
    put32(h, crc); put32(h, size); put32(h, size);
    put16(h, (uint32_t)name.size()); put16(h, (uint32_t)extra);
    h += name;
    if (extra) {
        h += "FB";
        put16(h, (uint32_t)(extra - 4));
        h.append(extra - 4, 'Z');
    }
    return h;
}

std::string pickle_f32_matrix(int64_t rows, int64_t cols) {
    auto J = [](std::string& s, int64_t v) {                          // BININT (4 bytes): valid for any value < 2^31
        s.push_back('J');
        for (int i = 0; i < 4; ++i) s.push_back((char)((v >> (8 * i)) & 0xff));
    };
#define SER_LIT(x) std::string(x, sizeof(x) - 1)                       /* literals with embedded NULs */
    std::string p = SER_LIT("\x80\x02" "ctorch._utils\n_rebuild_tensor_v2\nq\x00((X\x07\x00\x00\x00storageq\x01"
                            "ctorch\nFloatStorage\nq\x02X\x01\x00\x00\x00" "0q\x03X\x03\x00\x00\x00" "cpuq\x04");
    J(p, rows * cols);
    p += SER_LIT("tq\x05QK\x00");
    J(p, rows); J(p, cols);
    p += SER_LIT("\x86q\x06");
    J(p, cols);
    p += SER_LIT("K\x01\x86q\x07\x89" "ccollections\nOrderedDict\nq\x08)Rq\x09tq\x0aRq\x0b.");
#undef SER_LIT
    return p;
}

}  // namespace

static int pt_write_f32(const char* path, const float* host_data, int64_t rows, int64_t cols) {
    if (!path || rows < 0 || cols <= 0 || (!host_data && rows > 0)) return ser_fail(-1, "ser_pt_write_f32: bad arguments");
    if (rows >= (1LL << 29) || cols >= (1LL << 29) || rows * cols >= (1LL << 29))       // (each factor first: the product cannot wrap)
        return ser_fail(-2, "ser_pt_write_f32: tensor too large for a 32-bit zip member");
    const int64_t numel = rows * cols;
    // archive prefix = file stem, as torch.save names it
    std::string stem(path);
    const size_t slash = stem.find_last_of('/');
    if (slash != std::string::npos) stem = stem.substr(slash + 1);
    // the host writes "<name>.pt.<pid>.tmp" (round 4: "<name>.pt.tmp") and renames it when complete (frontend.save_feature): the archive is <name>.pt's
    if (stem.size() > 4 && stem.compare(stem.size() - 4, 4, ".tmp") == 0) {
        stem.resize(stem.size() - 4);
        const size_t d = stem.find_last_of('.');
        if (d != std::string::npos && d + 1 < stem.size() && stem.find_first_not_of("0123456789", d + 1) == std::string::npos) stem.resize(d);
    }
    const size_t dot = stem.find_last_of('.');
    if (dot != std::string::npos && dot > 0) stem = stem.substr(0, dot);
    if (stem.empty()) stem = "archive";

    const std::string pkl = pickle_f32_matrix(rows, cols);
    struct Part { std::string name; const void* data; size_t size; bool align; };
    const Part parts[4] = {
        {stem + "/data.pkl", pkl.data(), pkl.size(), false},
        {stem + "/byteorder", "little", 6, false},
        {stem + "/data/0", host_data, (size_t)numel * 4, true},
        {stem + "/version", "3\n", 2, false},
    };
    FILE* f = fopen(path, "wb");
    if (!f) return ser_fail(-errno, "ser_pt_write_f32: cannot open %s: %s", path, strerror(errno));
    std::vector<Member> members;
    size_t at = 0;
    bool ok = true;
    for (const Part& pt : parts) {
        const uint32_t crc = crc32_of(pt.data, pt.size);
        const std::string h = local_header(pt.name, crc, (uint32_t)pt.size, at, pt.align);
        members.push_back({pt.name, crc, (uint32_t)pt.size, (uint32_t)at});
        ok = ok && fwrite(h.data(), 1, h.size(), f) == h.size();
        ok = ok && (pt.size == 0 || fwrite(pt.data, 1, pt.size, f) == pt.size);
        at += h.size() + pt.size;
    }
    std::string cd;
    for (const Member& m : members) {
        put32(cd, 0x02014b50u); put16(cd, 20); put16(cd, 20); put16(cd, 0); put16(cd, 0);
        put16(cd, 0); put16(cd, 0x21);
        put32(cd, m.crc); put32(cd, m.size); put32(cd, m.size);
        put16(cd, (uint32_t)m.name.size()); put16(cd, 0); put16(cd, 0); put16(cd, 0); put16(cd, 0);
        put32(cd, 0); put32(cd, m.offset);
        cd += m.name;
    }
    std::string end;
    put32(end, 0x06054b50u); put16(end, 0); put16(end, 0); put16(end, (uint32_t)members.size()); put16(end, (uint32_t)members.size());
    put32(end, (uint32_t)cd.size()); put32(end, (uint32_t)at); put16(end, 0);
    ok = ok && fwrite(cd.data(), 1, cd.size(), f) == cd.size();
    ok = ok && fwrite(end.data(), 1, end.size(), f) == end.size();
    const int saved = errno;
    if (fclose(f) != 0) ok = false;
    if (!ok) return ser_fail(-(saved ? saved : EIO), "ser_pt_write_f32: short write to %s", path);
    return 0;
}

extern "C" int ser_pt_write_f32(const char* path, const float* host_data, int64_t rows, int64_t cols) {
    try {                                                             // no C++ exception may cross the C ABI (std::string / vector growth)
        return pt_write_f32(path, host_data, rows, cols);
    } catch (const std::bad_alloc&) {
        return ser_fail(-ENOMEM, "ser_pt_write_f32: out of memory");
    }
}

// ---------------------------------------------------------------------------------------------------------------- a5
namespace {
uint32_t rd32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint32_t rd16(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }
}  // namespace

// Returns the number of frames (samples per channel) of the file, or < 0 on error (ser_last_error() says why).
// With host_dst != NULL the mono float32 samples are written there (capacity in samples; a file with more frames
// is an error, not a truncation).  *sample_rate / *channels receive the header values when not NULL.
// Formats: RIFF/WAVE PCM 8/16/24/32-bit integer and 32-bit IEEE float, plain or WAVE_FORMAT_EXTENSIBLE.
static int64_t wav_read_f32(const char* path, float* host_dst, int64_t capacity, int32_t* sample_rate, int32_t* channels) {
    if (!path) return ser_fail(-1, "ser_wav_read_f32: null path");
    if (host_dst && capacity < 0) return ser_fail(-1, "ser_wav_read_f32: negative capacity");
    FILE* f = fopen(path, "rb");
    if (!f) return ser_fail(-2, "ser_wav_read_f32: cannot open %s: %s", path, strerror(errno));
    struct Closer { FILE* f; ~Closer() { fclose(f); } } closer{f};
    unsigned char hdr[12];
    if (fread(hdr, 1, 12, f) != 12 || memcmp(hdr, "RIFF", 4) || memcmp(hdr + 8, "WAVE", 4))
        return ser_fail(-3, "ser_wav_read_f32: %s is not a RIFF/WAVE file", path);
    int fmt = 0, ch = 0, bits = 0;
    uint32_t sr = 0;
    int64_t data_bytes = -1;
    for (;;) {
        unsigned char ck[8];
        if (fread(ck, 1, 8, f) != 8) break;
        const uint32_t len = rd32(ck + 4);
        if (!memcmp(ck, "fmt ", 4)) {
            unsigned char b[40] = {0};
            const uint32_t take = len < 40 ? len : 40;
            if (len < 16 || fread(b, 1, take, f) != take) break;
            fmt = (int)rd16(b); ch = (int)rd16(b + 2); sr = rd32(b + 4); bits = (int)rd16(b + 14);
            if (fmt == 0xFFFE && len >= 26) fmt = (int)rd16(b + 24);             // extensible: sub-format GUID starts with the tag
            if (fseeko(f, (off_t)(len - take) + (off_t)(len & 1), SEEK_CUR) != 0) break;
        } else if (!memcmp(ck, "data", 4)) {
            data_bytes = len;
            break;
        } else if (fseeko(f, (off_t)len + (off_t)(len & 1), SEEK_CUR) != 0) {
            break;
        }
    }
    if (data_bytes < 0 || ch <= 0 || bits <= 0)
        return ser_fail(-4, "ser_wav_read_f32: %s has no usable fmt/data chunks", path);
    if (ch > 1024)                                                               // libsndfile's SF_MAX_CHANNELS: soundfile refuses more
        return ser_fail(-4, "ser_wav_read_f32: %s declares %d channels", path, ch);
    const int bps = bits / 8;
    const bool is_float = fmt == 3;
    if (!((fmt == 1 && (bps == 1 || bps == 2 || bps == 3 || bps == 4) && bits % 8 == 0) || (is_float && bits == 32)))
        return ser_fail(-5, "ser_wav_read_f32: %s: format tag %d with %d bits is not supported", path, fmt, bits);
    // The frame size is channels x bytes per sample, whatever the header's blockAlign field says -- what libsndfile (behind
    // librosa.load, preprocess_speech.py:47) does for PCM / float data: it logs a mismatch and decodes with channels x width.
    // Round 3 sized the buffer by the header's blockAlign and read channels x width bytes per frame: a header with blockAlign = 2
    // and 60 000 channels read 120 KB past a 64-byte buffer.
    const int64_t align = (int64_t)bps * ch;
    {   // a streaming writer may leave 0xFFFFFFFF (or more than the file holds) in the data length: trust the file size then
        const off_t here = ftello(f);
        if (here < 0 || fseeko(f, 0, SEEK_END) != 0) return ser_fail(-7, "ser_wav_read_f32: cannot seek in %s", path);
        const off_t end = ftello(f);
        if (end < here || fseeko(f, here, SEEK_SET) != 0) return ser_fail(-7, "ser_wav_read_f32: cannot seek in %s", path);
        const int64_t rest = (int64_t)(end - here);
        if (data_bytes == 0xFFFFFFFFLL || data_bytes > rest) data_bytes = rest;
    }
    const int64_t frames = data_bytes / align;
    if (sample_rate) *sample_rate = (int32_t)sr;
    if (channels) *channels = ch;
    if (!host_dst) return frames;
    if (frames > capacity)
        return ser_fail(-6, "ser_wav_read_f32: %s has %lld frames, buffer holds %lld", path, (long long)frames, (long long)capacity);
    std::vector<unsigned char> raw((size_t)(frames * align));
    const size_t got = raw.empty() ? 0 : fread(raw.data(), 1, raw.size(), f);
    if (got != raw.size()) return ser_fail(-7, "ser_wav_read_f32: short read from %s", path);
    for (int64_t i = 0; i < frames; ++i) {
        const unsigned char* p = raw.data() + i * align;             // ch * bps = align bytes per frame: never past raw's end
        float acc = 0.f;
        for (int c = 0; c < ch; ++c, p += bps) {
            float v;
            if (is_float) { memcpy(&v, p, 4); }
            else if (bps == 2) v = (float)(int16_t)rd16(p) / 32768.0f;
            else if (bps == 1) v = ((float)p[0] - 128.0f) / 128.0f;
            else if (bps == 3) {
                int32_t x = (int32_t)(p[0] | (p[1] << 8) | (p[2] << 16));
                if (x & 0x800000) x -= 0x1000000;
                v = (float)((double)x / 8388608.0);
            } else v = (float)((double)(int32_t)rd32(p) / 2147483648.0);
            acc = c == 0 ? v : acc + v;                                  // numpy's float32 mean over the channel axis
        }
        host_dst[i] = ch == 1 ? acc : acc / (float)ch;               // add, then one true division, like np.mean
    }
    return frames;
}

extern "C" int64_t ser_wav_read_f32(const char* path, float* host_dst, int64_t capacity, int32_t* sample_rate,
                                    int32_t* channels) {
    try {
        return wav_read_f32(path, host_dst, capacity, sample_rate, channels);
    } catch (const std::bad_alloc&) {
        return ser_fail(-ENOMEM, "ser_wav_read_f32: out of memory reading %s", path ? path : "");
    }
}
