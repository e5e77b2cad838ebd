// slice_data.h -- CTU data of one picture (see slice_data.cpp) and the headers around it (headers.cpp).
#pragma once
#include "../../../include/wrenc_bitstream.h"
#include "cabac.h"

namespace wrenc_host {

// The reference's call surface for this part, kept by name and argument meaning:
//   SliceEncoder::encode(&mut self, slice: &Slice, sh: &SliceHeader) -> Bins          slice_encoder.rs:343
//   CtuEncoder::encode(&mut self, bins: &mut Bins, ctu: .., sh: &SliceHeader)         ctu_encoder.rs:33
// In the reference CtuEncoder::encode first runs the search of its CTU (split_ct, ctu_encoder.rs:53-54) and then
// writes the CTU's syntax; here the search of the whole picture has already run on the device (it needs the CTU
// wavefront of whole pictures), so a Ctu is a position plus the picture's record and encode() writes syntax only.
struct SliceHeader {
    int slice_qp; // sh_qp_delta + pps_init_qp (slice_header.rs:95-99)
};
struct Slice {    // one slice = one tile = the picture (slice_splitter.rs:11-20, tile_splitter.rs:10)
    int width, height;
    const wrenc_bs_record* record; // what split_ct left in the reference's CT / CU / TU graph
};
struct Ctu {
    int x, y; // luma position of the 32x32 CTU
};
typedef BitWriter Bins; // bins.rs: the bit accumulator the coder writes to

class PictureCoder; // the syntax walk over the flat record (slice_data.cpp)

class CtuEncoder {
public:
    explicit CtuEncoder(PictureCoder& coder) : coder_(coder) {}
    // initialises CABAC at the picture's first CTU (ctu_encoder.rs:38-47), then coding_tree of the CTU
    // (ctu_encoder.rs:172-201).  Returns WRENC_BS_OK or WRENC_BS_EDATA.
    int encode(Bins& bins, const Ctu& ctu, const SliceHeader& sh);

private:
    PictureCoder& coder_;
};

class SliceEncoder {
public:
    // slice header bits, byte alignment, the CTUs in raster order through CtuEncoder::encode, end_of_slice_one_bit
    // (slice_encoder.rs:343-427).  *status receives WRENC_BS_OK or the first error.
    Bins encode(const Slice& slice, const SliceHeader& sh, int* status);
    long long slice_data_bits() const { return slice_data_bits_; }

private:
    long long slice_data_bits_ = 0;
};

// CABAC-coded CTUs of the whole picture followed by end_of_slice_one_bit; bw must be byte aligned.
int write_slice_data(int width, int height, int qp, const wrenc_bs_record& rec, BitWriter& bw);
int write_slice_data_tokens(int width, int height, int qp, const wrenc_bs_tokens& tok, BitWriter& bw);

// Raw byte sequence payloads (headers.cpp)
void write_vps(BitWriter& bw, int width, int height);
void write_sps(BitWriter& bw, int width, int height);
void write_pps(BitWriter& bw, int width, int height, int qp);
void write_picture_header(BitWriter& bw, int poc);
void write_slice_header(BitWriter& bw, int qp);

enum NalType { NAL_IDR_W_RADL = 7, NAL_VPS = 14, NAL_SPS = 15, NAL_PPS = 16, NAL_PH = 19 };

// nal.rs:186-298: 00 00 00 00 00 01, two header bytes, payload with the reference's emulation prevention
void append_nal(std::vector<uint8_t>& out, int layer_id, NalType type, int temporal_id,
                const std::vector<uint8_t>& payload);

} // namespace wrenc_host
