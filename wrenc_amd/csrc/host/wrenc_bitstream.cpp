// wrenc_bitstream.cpp -- the C ABI of include/wrenc_bitstream.h.
#include "slice_data.h"

#include <cstring>

using namespace wrenc_host;

namespace {

thread_local long long g_last_slice_data_bits = 0;

bool size_ok(int width, int height, int qp) {
    return width >= 32 && height >= 32 && width % 32 == 0 && height % 32 == 0 && width <= 16384 &&
           height <= 16384 && qp >= 0 && qp <= 63;
}

int hand_over(const std::vector<uint8_t>& bytes, uint8_t* out, size_t cap, size_t* len) {
    if (len) *len = bytes.size();
    if (!out || cap < bytes.size()) return WRENC_BS_ENOSPC;
    memcpy(out, bytes.data(), bytes.size());
    return WRENC_BS_OK;
}

} // namespace

extern "C" {

size_t wrenc_bs_picture_bound(int width, int height) {
    // A level costs at most 32 escape bins + sign, everything else is far below one bit per sample on
    // top of that; emulation prevention adds at most one byte per two.
    if (width <= 0 || height <= 0) return 0;
    return (size_t)width * (size_t)height * 12 + 4096;
}

int wrenc_bs_write_parameter_sets(int width, int height, int qp, uint8_t* out, size_t cap, size_t* len) {
    if (!size_ok(width, height, qp)) return WRENC_BS_EINVAL;
    std::vector<uint8_t> stream;
    {
        BitWriter bw;
        write_vps(bw, width, height);
        append_nal(stream, 1, NAL_VPS, 0, bw.bytes()); // main.rs:232
    }
    {
        BitWriter bw;
        write_sps(bw, width, height);
        append_nal(stream, 9, NAL_SPS, 0, bw.bytes()); // main.rs:245
    }
    {
        BitWriter bw;
        write_pps(bw, width, height, qp);
        append_nal(stream, 9, NAL_PPS, 0, bw.bytes()); // main.rs:257
    }
    return hand_over(stream, out, cap, len);
}

int wrenc_bs_write_picture(int width, int height, int qp, int poc, const wrenc_bs_record* rec, uint8_t* out,
                           size_t cap, size_t* len) {
    if (!size_ok(width, height, qp) || poc < 0 || !rec || !rec->cu_log2_size || !rec->luma_mode ||
        !rec->chroma_mode || !rec->lev_y || !rec->lev_cb || !rec->lev_cr)
        return WRENC_BS_EINVAL;
    std::vector<uint8_t> stream;
    {
        BitWriter bw;
        write_picture_header(bw, poc);
        append_nal(stream, 9, NAL_PH, 0, bw.bytes()); // main.rs:307-313
    }
    {
        // main.rs:380-382: let bins = slice_encoder.encode(&slice, &sh)
        const Slice slice = {width, height, rec};
        const SliceHeader sh = {qp};
        SliceEncoder slice_encoder;
        int rc = WRENC_BS_OK;
        const Bins bins = slice_encoder.encode(slice, sh, &rc);
        if (rc) return rc;
        g_last_slice_data_bits = slice_encoder.slice_data_bits();
        append_nal(stream, 9, NAL_IDR_W_RADL, 0, bins.bytes()); // main.rs:377-383
    }
    return hand_over(stream, out, cap, len);
}

int wrenc_bs_write_picture_tokens(int width, int height, int qp, int poc, const wrenc_bs_tokens* tok, uint8_t* out, size_t cap,
                                  size_t* len) {
    if (!size_ok(width, height, qp) || poc < 0 || !tok || !tok->cu_log2_size || !tok->luma_mode || !tok->chroma_mode ||
        !tok->pool || !tok->first_page)
        return WRENC_BS_EINVAL;
    std::vector<uint8_t> stream;
    {
        BitWriter bw;
        write_picture_header(bw, poc);
        append_nal(stream, 9, NAL_PH, 0, bw.bytes());
    }
    {
        Bins bins;
        write_slice_header(bins, qp);
        const size_t header_bits = bins.bit_count();
        const int rc = write_slice_data_tokens(width, height, qp, *tok, bins);
        if (rc) return rc;
        g_last_slice_data_bits = (long long)(bins.bit_count() - header_bits);
        bins.align();
        append_nal(stream, 9, NAL_IDR_W_RADL, 0, bins.bytes());
    }
    return hand_over(stream, out, cap, len);
}

long long wrenc_bs_last_slice_data_bits(void) { return g_last_slice_data_bits; }

} // extern "C"
