/* wrenc_bitstream.h -- C ABI of the host bitstream writer that follows the RD search.
 *
 * the CABAC + CTU syntax writer and the parameter-set / NAL framing
 * of hjmkt/wrenc, restated as pure host functions over the record the device search returns
 * (include/wrenc_gpu.h, wrenc_gpu_picture).  Entropy coding is serial per picture and stays on
 * the CPU, as in the reference; pictures are independent (CABAC is re-initialised at the first
 * CTU of every picture, ctu_encoder.rs:38-47), so a caller runs one of these per host thread.
 *
 * What each entry point replaces (paths relative to the reference's src/):
 *   wrenc_bs_write_parameter_sets  main.rs:223-260 (VPS, SPS, PPS through vps_encoder.rs,
 *                                  sps_encoder.rs, pps_encoder.rs, ptl/gci/dpbp/rpl encoders)
 *   wrenc_bs_write_picture         main.rs:294-315 (PH NAL, ph_encoder.rs) + main.rs:358-385
 *                                  (slice NAL: slice_encoder.rs:32-427, ctu_encoder.rs:227-2269,
 *                                  bool_coder.rs, cabac_contexts.rs) + nal.rs:186-298 framing
 *
 * Plain C: pointers and sizes only, no allocation across the boundary.  Functions return 0 or a
 * negative wrenc_bs_status; nothing aborts (the reference asserts in release builds where this
 * returns WRENC_BS_EDATA).  No state is kept between calls.
 */
#ifndef WRENC_BITSTREAM_H
#define WRENC_BITSTREAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum wrenc_bs_status {
    WRENC_BS_OK = 0,
    WRENC_BS_EINVAL = -1,  /* bad argument (size not a multiple of 32, QP outside 0..63, NULL plane) */
    WRENC_BS_ENOSPC = -2,  /* output buffer too small; *len holds the size that is needed */
    WRENC_BS_EDATA = -3    /* record inconsistent with the syntax: a level whose parity contradicts the
                              dependent-quantisation state (ctu_encoder.rs:1975-1978,2258-2262), a
                              coding-unit size map that is not a quadtree, a chroma mode the chroma
                              syntax cannot express */
};

/* One picture's search result, exactly the planes wrenc_gpu_download fills (same layout). */
typedef struct wrenc_bs_record {
    const uint8_t* cu_log2_size; /* (width/4)*(height/4): log2 size of the luma CU covering the 4x4 */
    const uint8_t* luma_mode;    /* (width/4)*(height/4): 0 planar, 1 DC, 2..66 angular */
    const uint8_t* chroma_mode;  /* (width/8)*(height/8): 0..66 or 81/82/83 (LT/L/T CCLM) */
    const int16_t* lev_y;        /* width*height TransCoeffLevel, each TB at its own position */
    const int16_t* lev_cb;       /* (width/2)*(height/2) */
    const int16_t* lev_cr;
} wrenc_bs_record;

/* Size that is always enough for wrenc_bs_write_picture at this picture size. */
size_t wrenc_bs_picture_bound(int width, int height);

/* VPS + SPS + PPS NAL units of a sequence (byte-stream format, each with the reference's
 * 00 00 00 00 00 01 prefix).  *len = bytes written (or needed, on WRENC_BS_ENOSPC). */
int wrenc_bs_write_parameter_sets(int width, int height, int qp, uint8_t* out, size_t cap, size_t* len);

/* Picture header NAL + one IDR_W_RADL slice NAL holding every CTU of picture `poc`. */
int wrenc_bs_write_picture(int width, int height, int qp, int poc, const wrenc_bs_record* rec,
                           uint8_t* out, size_t cap, size_t* len);

/* The same picture from the device's TOKEN record (include/wrenc_gpu.h, wrenc_gpu_download_tokens: residual_coding done
 * on the device): the maps, the page pool the call filled, and this picture's table of first pages.  Writes the bytes
 * wrenc_bs_write_picture writes from the level planes of the same search result; the host then runs the CU-level syntax
 * and the arithmetic coder only.  WRENC_BS_EDATA on a token stream that ends early or names a context that does not exist. */
#define WRENC_BS_TOKEN_PAGE 64
typedef struct wrenc_bs_tokens {
    const uint8_t* cu_log2_size;
    const uint8_t* luma_mode;
    const uint8_t* chroma_mode;
    const uint32_t* pool;        /* the pages of the download call */
    size_t pool_words;           /* words of `pool` that are valid */
    const uint32_t* first_page;  /* (width / 32) * (height / 32): where each CTU's tokens start */
} wrenc_bs_tokens;
int wrenc_bs_write_picture_tokens(int width, int height, int qp, int poc, const wrenc_bs_tokens* tok, uint8_t* out,
                                  size_t cap, size_t* len);

/* Bits the CABAC engine produced for the CTU data of the last wrenc_bs_write_picture call made by
 * this thread (slice data without headers and framing): the true rate the search's estimate models. */
long long wrenc_bs_last_slice_data_bits(void);

#ifdef __cplusplus
}
#endif
#endif /* WRENC_BITSTREAM_H */
