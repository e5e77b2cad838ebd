// slice_data.h -- CTU data of one picture (see slice_data.cpp) and the headers around it (headers.cpp).
#pragma once
#include "../../../include/wrenc_bitstream.h"
#include "cabac.h"

namespace wrenc_host {

// CABAC-coded CTUs of the whole picture followed by end_of_slice_one_bit; bw must be byte aligned.
int write_slice_data(int width, int height, int qp, const wrenc_bs_record& rec, BitWriter& bw);

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
