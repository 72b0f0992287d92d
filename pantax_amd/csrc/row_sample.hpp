// row_sample.hpp -- a11 (sample_sorted, profile.rs:1287-1295): restatement of rand 0.9.2's
// StdRng::seed_from_u64 + slice::choose_multiple; see row_sample.cpp for what is pinned and what is not.
#pragma once
#include <cstdint>
#include <vector>

namespace ptx {

void chacha_block(const uint32_t key[8], uint64_t counter, uint64_t stream, int rounds, uint32_t out[16]);

struct StdRng {   // rand 0.9 StdRng = ChaCha12Rng
    uint32_t key[8];
    uint64_t counter = 0;
    uint32_t buf[16];
    int idx = 16;
    explicit StdRng(uint64_t seed);
    uint32_t next_u32();
};

// bit r of `bits` = the r-th valid row (in node order) is among the `amount` rows chosen out of `length`
void sample_ranks(uint64_t length, uint64_t amount, uint64_t seed, std::vector<uint32_t> &bits);

}  // namespace ptx
