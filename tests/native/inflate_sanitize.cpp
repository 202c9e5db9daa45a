// Sanitizer driver of the DEFLATE decoder (svim_asm_amd/csrc/svx_inflate.h): streams made by zlib's deflate over
// the kinds of bytes a BAM holds, whole / cut / bit-flipped / overwritten / pure noise, decoded into heap buffers of
// exactly the size the decoder is told (so that any write or read past them is caught), whole and in resumed prefix
// steps, and compared with zlib's inflate: same verdict, same bytes.
//   inflate_sanitize N_ROUNDS SEED
#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <random>
#include <vector>

#include "svx_inflate.h"

static std::vector<uint8_t> deflate_raw(const std::vector<uint8_t>& in, int level, int strategy, int mem) {
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (deflateInit2(&z, level, Z_DEFLATED, -15, mem, strategy) != Z_OK) abort();
    std::vector<uint8_t> out(deflateBound(&z, in.size()) + 64);
    z.next_in = const_cast<Bytef*>(in.data());
    z.avail_in = (uInt)in.size();
    z.next_out = out.data();
    z.avail_out = (uInt)out.size();
    if (deflate(&z, Z_FINISH) != Z_STREAM_END) abort();
    out.resize(z.total_out);
    deflateEnd(&z);
    return out;
}

// zlib's verdict: the whole stream decodes to at most cap bytes and ends
static bool zlib_inflate(const std::vector<uint8_t>& s, size_t cap, std::vector<uint8_t>* out) {
    z_stream z;
    memset(&z, 0, sizeof(z));
    if (inflateInit2(&z, -15) != Z_OK) abort();
    out->assign(cap + 1, 0);
    z.next_in = const_cast<Bytef*>(s.data());
    z.avail_in = (uInt)s.size();
    z.next_out = out->data();
    z.avail_out = (uInt)out->size();
    const int rc = inflate(&z, Z_FINISH);
    const size_t n = z.total_out;
    inflateEnd(&z);
    if (rc != Z_STREAM_END || n > cap) return false;
    out->resize(n);
    return true;
}

static std::vector<uint8_t> make_data(std::mt19937_64& rng, size_t n) {
    std::vector<uint8_t> d(n);
    static const uint8_t base[4] = {1, 2, 4, 8};
    switch (rng() % 7) {
        case 0: for (auto& b : d) b = (uint8_t)((base[rng() & 3] << 4) | base[(rng() >> 8) & 3]); break;   // SEQ
        case 1: for (auto& b : d) b = (uint8_t)rng(); break;
        case 2: break;                                                                                      // zeros
        case 3: for (size_t i = 0; i < n; ++i) d[i] = (uint8_t)("svim_asm.DEL.\tPASS\tSVTYPE=DEL;END="[i % 33] + (rng() % 97 == 0)); break;
        case 4: {  // repeated words at many distances
            size_t i = 0;
            while (i < n) {
                if (i > 8 && rng() % 3) {
                    const size_t dist = 1 + rng() % std::min<size_t>(i, 32768), len = 3 + rng() % 300;
                    for (size_t k = 0; k < len && i < n; ++k, ++i) d[i] = d[i - dist];
                } else {
                    d[i++] = (uint8_t)rng();
                }
            }
            break;
        }
        case 5: for (auto& b : d) b = (rng() % 10) ? 0 : (uint8_t)rng(); break;                              // skewed
        default: for (size_t i = 0; i < n; ++i) d[i] = (uint8_t)((i * 2654435761u) >> (rng() % 2 ? 13 : 24)); break;
    }
    return d;
}

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 300;
    std::mt19937_64 rng(argc > 2 ? strtoull(argv[2], nullptr, 10) : 1);
    std::unique_ptr<svx_inflate::Stream> st(new svx_inflate::Stream()), st2(new svx_inflate::Stream());
    std::vector<uint8_t> prev_stream, prev_want;  // the previous case: the partner of the side-by-side run
    bool prev_ok = false;
    size_t prev_cap = 0, pairs = 0;
    size_t accepted = 0, refused = 0, long_codes = 0;
    static const int strategies[5] = {Z_DEFAULT_STRATEGY, Z_FILTERED, Z_HUFFMAN_ONLY, Z_RLE, Z_FIXED};
    static const size_t sizes[8] = {0, 1, 7, 300, 321, 5000, 40000, 65536};
    for (int r = 0; r < rounds; ++r) {
        const std::vector<uint8_t> data = make_data(rng, sizes[rng() % 8]);
        std::vector<uint8_t> s = deflate_raw(data, (int)(rng() % 10), strategies[rng() % 5], 1 + (int)(rng() % 9));
        for (int m = 0; m < 12; ++m) {
            std::vector<uint8_t> t = s;
            const int how = m == 0 ? -1 : (int)(rng() % 4);
            if (how == 0 && !t.empty()) {
                for (int k = 0, nk = 1 + (int)(rng() % 3); k < nk; ++k) t[rng() % t.size()] ^= (uint8_t)(1u << (rng() % 8));
            } else if (how == 1 && !t.empty()) {
                t.resize(rng() % t.size());
            } else if (how == 2 && !t.empty()) {
                const size_t k = rng() % t.size();
                for (size_t j = k; j < t.size() && j < k + 4; ++j) t[j] = (uint8_t)rng();
            } else if (how == 3) {
                t.resize(1 + rng() % 300);
                for (auto& b : t) b = (uint8_t)rng();
            }
            static const size_t slack[4] = {0, 0, 5, 400};
            const size_t cap = data.size() + slack[rng() % 4];
            std::vector<uint8_t> want;
            const bool z_ok = zlib_inflate(t, cap, &want);
            // exact-size heap copies: the sanitizer sees every byte outside
            std::unique_ptr<uint8_t[]> in(new uint8_t[t.size() ? t.size() : 1]);
            if (!t.empty()) memcpy(in.get(), t.data(), t.size());
            for (int mode = 0; mode < 2; ++mode) {
                std::unique_ptr<uint8_t[]> out(new uint8_t[cap ? cap : 1]);
                st->begin(in.get(), t.size());
                bool ok = true;
                if (mode == 1) {
                    size_t stop = 0;
                    for (int k = 0; k < 4 && ok; ++k) {
                        stop += cap ? rng() % (cap + 1 - stop) : 0;
                        ok = st->run(out.get(), cap, stop, false);
                        if (ok && st->produced() < stop) { fprintf(stderr, "stopped short of the stop\n"); return 1; }
                    }
                }
                ok = ok && st->run(out.get(), cap, 0, true);
                if (st->produced() > cap) { fprintf(stderr, "produced more than cap\n"); return 1; }
                if (ok != z_ok && !(mode == 1 && z_ok == true && !ok)) {
                    // (prefix steps ask for bytes a damaged stream that zlib still accepts may not have: refusing is fine)
                    fprintf(stderr, "verdicts differ: own %d zlib %d (round %d mutation %d how %d mode %d)\n", ok, z_ok, r, m, how, mode);
                    return 1;
                }
                if (ok && (st->produced() != want.size() || (!want.empty() && memcmp(out.get(), want.data(), want.size()) != 0))) {
                    fprintf(stderr, "bytes differ (round %d mutation %d)\n", r, m);
                    return 1;
                }
                if (mode == 0) (ok ? accepted : refused)++;
            }
            // the same stream decoded side by side with the previous case's: each as if alone
            if (!prev_stream.empty() || prev_cap == 0) {
                std::unique_ptr<uint8_t[]> in2(new uint8_t[prev_stream.size() ? prev_stream.size() : 1]);
                if (!prev_stream.empty()) memcpy(in2.get(), prev_stream.data(), prev_stream.size());
                std::unique_ptr<uint8_t[]> oa(new uint8_t[cap ? cap : 1]), ob(new uint8_t[prev_cap ? prev_cap : 1]);
                st->begin(in.get(), t.size());
                st2->begin(in2.get(), prev_stream.size());
                bool ok_a = false, ok_b = false;
                svx_inflate::Stream::run_pair(*st, oa.get(), cap, 0, true, &ok_a, *st2, ob.get(), prev_cap, 0, true, &ok_b);
                if (ok_a != z_ok || ok_b != prev_ok ||
                    (ok_a && (st->produced() != want.size() || (!want.empty() && memcmp(oa.get(), want.data(), want.size()) != 0))) ||
                    (ok_b && (st2->produced() != prev_want.size() || (!prev_want.empty() && memcmp(ob.get(), prev_want.data(), prev_want.size()) != 0)))) {
                    fprintf(stderr, "side by side differs from alone (round %d mutation %d)\n", r, m);
                    return 1;
                }
                ++pairs;
            }
            prev_stream = t;
            prev_want = want;
            prev_ok = z_ok;
            prev_cap = cap;
        }
        long_codes += s.size() > 100;
    }
    printf("inflate_sanitize ok: %zu accepted, %zu refused, %zu side by side\n", accepted, refused, pairs);
    return accepted > 0 && refused > 0 ? 0 : 1;
}
