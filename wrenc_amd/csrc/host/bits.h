// bits.h -- MSB-first bit accumulator of the host bitstream writer.
//
// Does the job of the reference's Bins (bins.rs:3-140: a u64 head plus a Vec<u64> tail) with a plain
// byte vector; u(n), ue(v), se(v) as bool_coder.rs:62-84 writes them.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace wrenc_host {

class BitWriter {
public:
    void bit(bool b) {
        acc_ = (uint8_t)((acc_ << 1) | (b ? 1 : 0));
        if (++n_ == 8) {
            bytes_.push_back(acc_);
            acc_ = 0;
            n_ = 0;
        }
    }
    // the low `len` bits of v, most significant first (Bins::push_bins_with_size, bins.rs:58-80)
    void put(uint64_t v, int len) {
        for (int i = len - 1; i >= 0; --i) bit((v >> i) & 1);
    }
    // one whole byte at a byte boundary
    void byte(uint8_t v) {
        if (n_ == 0)
            bytes_.push_back(v);
        else
            put(v, 8);
    }
    void reserve(size_t n) { bytes_.reserve(n); }
    // bool_coder.rs:62-73
    void ue(uint64_t v) {
        if (v == 0) {
            bit(true);
            return;
        }
        int nbits = 0;
        for (uint64_t t = v + 1; t; t >>= 1) ++nbits;
        const int n = nbits - 1;
        put(0, n);
        bit(true);
        put(v - ((1ull << n) - 1), n);
    }
    // bool_coder.rs:76-84
    void se(int64_t v) {
        if (v == 0) {
            bit(true);
            return;
        }
        const uint64_t a = (uint64_t)(v < 0 ? -v : v);
        ue((a - 1) * 2 + 1 + (v < 0 ? 1 : 0));
    }
    // zero bits up to the next byte boundary (Bins::byte_align, bins.rs:128-134)
    void align() {
        while (n_) bit(false);
    }
    bool aligned() const { return n_ == 0; }
    size_t bit_count() const { return bytes_.size() * 8 + (size_t)n_; }
    const std::vector<uint8_t>& bytes() const { return bytes_; } // complete bytes only

private:
    std::vector<uint8_t> bytes_;
    uint8_t acc_ = 0;
    int n_ = 0;
};

} // namespace wrenc_host
