// vvc_parse.h -- TEST INFRASTRUCTURE ONLY (see vvc_parse.cpp): parse a byte stream written by the host
// bitstream writer back into the record of wrenc_oracle.h.
#pragma once
#include "wrenc_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wro_stream_info {
    int width, height; // sps_pic_width/height_max_in_luma_samples
    int init_qp;       // 26 + pps_init_qp_minus26
    int n_pictures;    // picture header + IDR slice pairs
} wro_stream_info;

// Parameter sets of the stream (every field is checked against what the reference writes).
// Returns 0 or a negative code naming the structure that failed (vvc_parse.cpp).
int wro_parse_stream_info(const uint8_t* stream, size_t len, wro_stream_info* info);

// CABAC-decode picture `index`: fills rec->cu_log2_size, luma_mode, chroma_mode, lev_y/cb/cr (rec_* and
// ctu_cost are not touched; rec may be NULL to read the headers only).  Also checks end_of_slice_one_bit
// and the trailing bits.
int wro_parse_picture(const uint8_t* stream, size_t len, int index, int* poc_lsb, int* slice_qp,
                      wro_picture_out* rec);

// The parser's up-right diagonal scan of a (1 << lg) square, lg 0..3: 2 bytes (x, y) per position
// (tests pin it to the reference's DIAG_SCAN_ORDER table, ctu.rs:14-81).
void wro_parse_debug_scan(int lg, uint8_t* xy);

#ifdef __cplusplus
}
#endif
