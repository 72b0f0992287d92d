// row_sample.cpp -- a11: which LP rows survive `--sample N` (sample_sorted, profile.rs:1287-1295):
//     StdRng::seed_from_u64(42); valid_nodes.choose_multiple(&mut rng, N); sort
// The chosen set depends only on (number of valid rows n, N): it is a set of RANKS among the valid rows.
//
// The algorithm lives in the reference's dependencies, which are not vendored under /root/reference
// (Cargo.lock: rand 0.9.2, rand_chacha 0.9.0, rand_core 0.9.x).  This file restates their published algorithm:
//   * rand_core  SeedableRng::seed_from_u64 : the u64 is expanded to the 32-byte seed by a PCG32 stream
//   * rand_chacha ChaCha12Rng (= StdRng)    : ChaCha, 12 rounds, 64-bit block counter (words 12-13) starting at 0,
//                                             stream id 0 (words 14-15); the u32 output is the keystream in order
//   * rand::seq::index::sample              : amount >= 163: in-place partial Fisher-Yates when
//                                             length < C[j]*amount (C = {270, 330/9}, j = length >= 500000), else
//                                             rejection sampling; amount < 163: Floyd's / in-place by the f32 rule
//   * Rng::random_range(lo..hi) for u32     : Canon's method, single extra draw (UniformInt::sample_single_inclusive)
//   * Uniform<u32>::sample                  : widening multiply with rejection of lo < (2^32 - range) % range
// PARITY UNPINNED: no Rust toolchain or crate source is available here to run the reference; the ChaCha core is
// checked against the published ChaCha20/ChaCha12 zero-key keystreams (tests/test_host_io.py), the rest is a
// restatement.  DESIGN.md §8 carries the same note.
#include <cstdint>
#include <cstring>
#include <vector>
#include "row_sample.hpp"

namespace ptx {

namespace {
inline uint32_t rotl(uint32_t x, int k) { return (x << k) | (x >> (32 - k)); }
inline void quarter(uint32_t *s, int a, int b, int c, int d) {
    s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 16);
    s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 12);
    s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 8);
    s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 7);
}
}  // namespace

void chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16]) {
    uint32_t in[16] = {0x61707865u, 0x3320646eu, 0x79622d32u, 0x6b206574u, key[0], key[1], key[2], key[3], key[4], key[5], key[6], key[7],
                       (uint32_t)counter, (uint32_t)(counter >> 32), (uint32_t)stream, (uint32_t)(stream >> 32)};
    uint32_t s[16];
    std::memcpy(s, in, sizeof(s));
    for (int r = 0; r < rounds; r += 2) {
        quarter(s, 0, 4, 8, 12); quarter(s, 1, 5, 9, 13); quarter(s, 2, 6, 10, 14); quarter(s, 3, 7, 11, 15);
        quarter(s, 0, 5, 10, 15); quarter(s, 1, 6, 11, 12); quarter(s, 2, 7, 8, 13); quarter(s, 3, 4, 9, 14);
    }
    for (int i = 0; i < 16; ++i) out[i] = s[i] + in[i];
}

StdRng::StdRng(uint64_t seed) {
    // rand_core seed_from_u64: PCG32 (XSH-RR) steps, four little-endian bytes each
    uint64_t state = seed;
    for (int w = 0; w < 8; ++w) {
        state = state * 6364136223846793005ull + 11634580027462260723ull;
        const uint32_t xorshifted = (uint32_t)(((state >> 18) ^ state) >> 27);
        const uint32_t rot = (uint32_t)(state >> 59);
        key[w] = (xorshifted >> rot) | (xorshifted << ((32 - rot) & 31));
    }
}

uint32_t StdRng::next_u32() {
    if (idx >= 16) { chacha_block(key, counter++, 0, 12, buf); idx = 0; }
    return buf[idx++];
}

// Rng::random_range(low..high), u32
static uint32_t range_u32(StdRng &rng, uint32_t low, uint32_t high_excl) {
    const uint32_t range = high_excl - low;           // (high - 1) - low + 1
    if (range == 0) return rng.next_u32();            // full range (cannot happen for i < length)
    uint64_t m = (uint64_t)rng.next_u32() * range;
    uint32_t result = (uint32_t)(m >> 32);
    const uint32_t lo_order = (uint32_t)m;
    if (lo_order > (uint32_t)(0u - range)) {          // the sample may be biased: one more draw decides the carry
        const uint32_t new_hi = (uint32_t)(((uint64_t)rng.next_u32() * range) >> 32);
        result += ((uint64_t)lo_order + new_hi) >> 32 ? 1u : 0u;
    }
    return low + result;
}

void sample_ranks(uint64_t length, uint64_t amount, uint64_t seed, std::vector<uint32_t> &bits) {
    bits.assign((length + 31) / 32, 0u);
    if (amount >= length) {   // the caller samples only when length > amount
        for (uint64_t i = 0; i < length; ++i) bits[i >> 5] |= 1u << (i & 31);
        return;
    }
    StdRng rng(seed);
    const uint32_t len = (uint32_t)length, amt = (uint32_t)amount;
    auto inplace = [&]() {
        std::vector<uint32_t> idx(len);
        for (uint32_t i = 0; i < len; ++i) idx[i] = i;
        for (uint32_t i = 0; i < amt; ++i) { const uint32_t j = range_u32(rng, i, len); std::swap(idx[i], idx[j]); }
        for (uint32_t i = 0; i < amt; ++i) bits[idx[i] >> 5] |= 1u << (idx[i] & 31);
    };
    auto rejection = [&]() {   // the bitmap doubles as the "seen" set
        const uint32_t thresh = (uint32_t)(0u - len) % len;
        for (uint32_t k = 0; k < amt; ++k) {
            for (;;) {
                uint64_t m;
                do m = (uint64_t)rng.next_u32() * len; while ((uint32_t)m < thresh);
                const uint32_t pos = (uint32_t)(m >> 32);
                if (bits[pos >> 5] & (1u << (pos & 31))) continue;
                bits[pos >> 5] |= 1u << (pos & 31);
                break;
            }
        }
    };
    auto floyd = [&]() {
        // for j in length-amount .. length: t = random_range(..=j); if t already chosen, take j instead
        for (uint32_t j = len - amt; j < len; ++j) {
            const uint32_t t = j == 0xFFFFFFFFu ? rng.next_u32() : range_u32(rng, 0, j + 1);
            const uint32_t pick = (bits[t >> 5] & (1u << (t & 31))) ? j : t;
            bits[pick >> 5] |= 1u << (pick & 31);
        }
    };
    const int j = len >= 500000u ? 1 : 0;
    if (amt < 163) {
        static const float C[2][2] = {{1.6f, 8.0f / 45.0f}, {10.0f, 70.0f / 9.0f}};
        const float af = (float)amt, m4 = C[0][j] * af;
        if (amt > 11 && (float)len < (C[1][j] + m4) * af) inplace(); else floyd();
    } else {
        static const float C[2] = {270.0f, 330.0f / 9.0f};
        if ((float)len < C[j] * (float)amt) inplace(); else rejection();
    }
}

}  // namespace ptx

extern "C" int pantax_hip_sample_ranks(uint64_t n_valid, uint64_t sample_nodes, uint64_t seed, uint32_t *bits_out) {
    if (!bits_out || n_valid > 0xFFFFFFFFull) return -1;
    std::vector<uint32_t> bits;
    ptx::sample_ranks(n_valid, sample_nodes, seed, bits);
    std::memcpy(bits_out, bits.data(), bits.size() * sizeof(uint32_t));
    return 0;
}

extern "C" int pantax_hip_chacha_block(const uint32_t *key8, uint64_t counter, int rounds, uint32_t *out16) {
    if (!key8 || !out16 || rounds <= 0 || (rounds & 1)) return -1;
    ptx::chacha_block(key8, counter, 0, rounds, out16);
    return 0;
}
