/* wrenc_gpu.h -- C ABI of the MI355X (gfx950) all-intra RD-search path.
 *
 * Drop-in boundary for the ONE hot path of hjmkt/wrenc: the per-CTU RD search
 * (BlockSplitter::split_ct and everything it calls) plus the per-TU final pass.
 * The reference has no FFI; the seam is two call sites (paths relative to the
 * reference's src/):
 *   - search:     BlockSplitter::new + split_ct            ctu_encoder.rs:53-54
 *   - final pass: predict/transform/quantize/dequantize/
 *                 inverse_transform/recon per TU component  ctu_encoder.rs:1421-1461
 * Both are entered once per CTU from CtuEncoder::encode (ctu_encoder.rs:33) inside
 * SliceEncoder::encode's raster CTU loop (slice_encoder.rs:352-379).  Nothing in
 * them reads CABAC state, so this ABI is picture-granular: the device runs the
 * search of whole pictures (CTU wavefront inside a picture, many pictures in
 * flight) and hands back what the host entropy coder needs.
 *
 * Plain C: pointers and sizes only.  All functions return 0 on success or a
 * negative wrenc_gpu_status; wrenc_gpu_last_error() gives the text.  Nothing
 * aborts across this boundary (the reference panics or exit(0)s, main.rs:127-133).
 * One context per GPU, one host thread per context.
 */
#ifndef WRENC_GPU_H
#define WRENC_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wrenc_gpu_ctx wrenc_gpu_ctx;

enum wrenc_gpu_status {
    WRENC_GPU_OK = 0,
    WRENC_GPU_EINVAL = -1,   /* bad argument (size not a multiple of 32: picture.rs:178-181) */
    WRENC_GPU_ENODEV = -2,   /* no usable HIP device / wrong architecture */
    WRENC_GPU_ENOMEM = -3,
    WRENC_GPU_EHIP = -4,     /* HIP runtime error, see last_error */
    WRENC_GPU_ESTATE = -5,   /* slot not submitted / still in flight */
    WRENC_GPU_ELEVEL = -6    /* a quantised level reached 1024 (reference would panic,
                                block_splitter.rs:453) */
};
/* WRENC_GPU_ELEVEL and the internal WRENC_GPU_EHIP "a team member never reached a meeting point" are reported by sync /
 * download / download_compact and are STICKY, like the reference's panic: the pictures of the call that raised them --
 * and of every call in flight next to it -- are undefined (a team that times out stops storing, its CTU and everything
 * that depends on it are garbage; the slots' zero-block bookkeeping no longer matches their level planes), every later
 * sync / download of the context fails the same way, and the context must be destroyed.  The wrenc_gpu_test_* entries
 * keep a word of their own and never poison a context. */

/* Resolved configuration.  The RD-model constants are resolved on the host
 * (libm pow/powf, exactly as block_splitter.rs:29-53,187-375 and
 * quantizer.rs:16-25,650-683 do) and passed as tables so that device code does
 * no transcendental math.  wrenc_gpu_default_config() fills every table from the
 * reference's defaults for a given QP. */
typedef struct wrenc_gpu_config {
    int32_t width;            /* luma, multiple of 32 (CLI --output-size, main.rs:176-191) */
    int32_t height;
    int32_t qp;               /* CLI --qp (main.rs:193-198), 26 when absent (ctu.rs:382) */
    int32_t max_split_depth;  /* CLI --max-split-depth 0..3 (main.rs:108-109) */
    int32_t device;           /* HIP device ordinal */
    int32_t n_slots;          /* pictures resident on the device at once (>= 1) */
    int64_t lv_table[1024];   /* lv_dq_trellis_table, block_splitter.rs:51-52 */
    int64_t dq_table[1024];   /* quantizer.rs:20-21 */
    int64_t lambda_q;         /* quantizer.rs:683 */
    float lambda_rd;          /* block_splitter.rs:472 */
    float lambda_rd_chroma;   /* block_splitter.rs:775-778 (extra-param "a") */
    /* header bits ((x*16384.0) as i64), block_splitter.rs:377-406.
     * [tree: 0 SINGLE_TREE, 1 DUAL_TREE_LUMA]
     * [cclm class: 0 not CCLM, 1..3 = cclm_mode_idx 0..2]
     * [mode class: 0 planar, 1..5 = mpm_idx 0..4, 6..66 = mpm_remainder 0..60] */
    int64_t header_bits_luma[2][4][67];
    /* block_splitter.rs:695-712. [0 not CCLM, 1..3 = cclm_mode_idx 0..2] */
    int64_t header_bits_chroma[4];
} wrenc_gpu_config;

/* Per-picture result (caller-allocated; any pointer may be NULL to skip it).
 * Replaces what the reference keeps in its Rust object graph:
 *   CT tree ctu.rs:1794-1821, CodingUnit.intra_pred_mode ctu.rs:1241,
 *   TransformUnit.cu_intra_pred_mode ctu.rs:361,
 *   TransformUnit.quantized_transformed_coeffs ctu.rs:340 (TransCoeffLevel,
 *   written by the final pass), Tile.reconst_pixels tile.rs:17. */
typedef struct wrenc_gpu_picture {
    uint8_t* rec_y;         /* width*height */
    uint8_t* rec_cb;        /* (width/2)*(height/2) */
    uint8_t* rec_cr;
    int16_t* lev_y;         /* TransCoeffLevel planes, each TB at its own position */
    int16_t* lev_cb;
    int16_t* lev_cr;
    uint8_t* cu_log2_size;  /* (width/4)*(height/4): log2 size of the luma CU covering the 4x4 */
    uint8_t* luma_mode;     /* (width/4)*(height/4): intra_pred_mode[0] of that CU */
    uint8_t* chroma_mode;   /* (width/8)*(height/8): chroma prediction mode of the chroma block
                               (TU array: 0..66, or 81 LT_CCLM / 82 L_CCLM / 83 T_CCLM) */
    float* ctu_cost;        /* one f32 per CTU: value split_ct returns at ctu_encoder.rs:54 */
} wrenc_gpu_picture;

/* Fill cfg (all tables) from the reference's default constants for this QP. */
int wrenc_gpu_default_config(wrenc_gpu_config* cfg, int width, int height, int qp,
                             int max_split_depth);

/* Re-resolve cfg's tables with the reference's --extra-params string "K1=V1,K2=V2" (main.rs:202-217):
 * the RD-model tuning knobs of block_splitter.rs:21-53,187-375,594-693,775 and quantizer.rs:16-19,650-683.
 * Keys of the code path that is live (dependent quantisation + trellis) take effect; other keys are accepted
 * and ignored, as in the reference.  WRENC_GPU_EINVAL (text in wrenc_gpu_last_error(NULL)) for an item that
 * is not KEY=VALUE or a live key whose value is not a number.  NULL or "" gives the defaults again. */
int wrenc_gpu_config_extra_params(wrenc_gpu_config* cfg, const char* extra_params);

/* WRENC_GPU_EINVAL for a geometry that is not whole CTUs, a QP outside 0..63 -- and for a quantiser rate model the device's
 * arithmetic does not cover: the trellis keeps its path costs in 32 bits, exact while 128 * 65535 + lambda_q * dq_table[i]
 * < 2^25 for every i.  That holds for the reference's defaults at every QP and for tuned models near them; it does not
 * for values such as quant_lv_pow = 2.5 or quant_qp_div_trellis = 1.5, which are refused here (the reference computes those
 * products in i64) instead of being searched with other results.  WRENC_GPU_ENODEV without an MI355X. */
int wrenc_gpu_create(const wrenc_gpu_config* cfg, wrenc_gpu_ctx** out);
void wrenc_gpu_destroy(wrenc_gpu_ctx* ctx);
const char* wrenc_gpu_last_error(const wrenc_gpu_ctx* ctx); /* ctx may be NULL: create errors */

/* Copy one picture's planes (host memory, given strides in bytes) into slot `slot`
 * (asynchronous on the context's copy stream, behind any search still reading the slot;
 * the host buffers must stay valid until wrenc_gpu_sync or wrenc_gpu_download of that
 * slot). Uploads overlap the search of other slots.  Replaces main.rs:318-350 +
 * picture.rs:169-196. */
int wrenc_gpu_upload(wrenc_gpu_ctx* ctx, int slot, const uint8_t* y, const uint8_t* cb,
                     const uint8_t* cr, size_t stride_y, size_t stride_c);

/* Run search + final pass for slots [first_slot, first_slot + n_pictures) whose
 * planes are resident.  Asynchronous.  This is the call a C++ SliceEncoder::encode
 * makes in place of the per-CTU split_ct loop. */
int wrenc_gpu_encode(wrenc_gpu_ctx* ctx, int first_slot, int n_pictures);

/* Wait for everything queued on the context. */
int wrenc_gpu_sync(wrenc_gpu_ctx* ctx);

/* Copy a slot's results to host memory (blocking).  Waits for the wrenc_gpu_encode call
 * that searched the slot and for nothing queued after it, so with two sets of slots the
 * read-back of one set overlaps the search of the other. */
int wrenc_gpu_download(wrenc_gpu_ctx* ctx, int slot, wrenc_gpu_picture* out);

/* Compact read-back.  The level planes of a picture are mostly zeros (6 bytes per luma sample on the bus for a few
 * hundred coded bytes); a device pass behind the search packs them: one bit per 4x4 BLOCK OF LEVELS (picture-aligned,
 * luma plane, then Cb, then Cr, each in raster order of its 4x4 blocks: (W/4)(H/4) + 2 (W/8)(H/8) bits, bit b of word
 * b / 32) and, for the blocks with a non-zero level, the 16 levels (row-major) back to back in mask order.
 * wrenc_gpu_download_compact reads back `n` slots in one call: per picture the mask, its count of coded blocks, their
 * levels (up to payload_cap blocks: WRENC_GPU_ENOMEM with n_blocks set to the need if a picture has more), the maps,
 * and the reconstruction when asked for.  Blocking; waits for the encode call of the slots and nothing queued later.
 * wrenc_gpu_expand_levels is the host-side inverse (no device involved): dense level planes as wrenc_gpu_download
 * fills them, for consumers of the plane layout (wrenc_bs_write_picture). */
typedef struct wrenc_gpu_compact {
    uint32_t* mask;         /* wrenc_gpu_compact_mask_words(width, height) words */
    int16_t* payload;       /* payload_cap x 16 levels */
    size_t payload_cap;     /* in blocks of 16 levels */
    size_t n_blocks;        /* out: coded blocks of this picture */
    uint8_t* cu_log2_size;  /* as in wrenc_gpu_picture; any may be NULL */
    uint8_t* luma_mode;
    uint8_t* chroma_mode;
    uint8_t* rec_y;
    uint8_t* rec_cb;
    uint8_t* rec_cr;
} wrenc_gpu_compact;
size_t wrenc_gpu_compact_mask_words(int width, int height);
int wrenc_gpu_download_compact(wrenc_gpu_ctx* ctx, int first_slot, int n, wrenc_gpu_compact* out);
void wrenc_gpu_expand_levels(int width, int height, const uint32_t* mask, const int16_t* payload, int16_t* lev_y,
                             int16_t* lev_cb, int16_t* lev_cr);

/* Token read-back (round 4): residual_coding done on the device.  Behind the search a device pass walks every CTU's
 * transform blocks in coding order and turns their levels into the tokens the host's arithmetic coder consumes as they
 * come (ctu_encoder.rs:1786-2269, context selection bool_coder.rs:2053-2400, binarisations :1133-1465):
 *     context-coded bin   (ctx << 1) | bin                      ctx: index into the flat context array of wrenc_amd/csrc/host/cabac.h
 *     bypass group        1 << 31 | (nbits - 1) << 25 | value   nbits <= 25
 * Per transform unit (CU luma + Cb + Cr; a 4x4 luma block of a split 8x8 CU; the 4x4 Cb + Cr pair of such a CU) one header
 * word per component -- bit 31: coded, bits 0..23: tokens of the component that follow, bit 30 (luma): the last
 * significant position is not the DC position (MtsDcOnly = 0), bit 29 (luma): a level outside the 16x16 low-frequency
 * corner (MtsZeroOutSigCoeffFlag = 0) -- then the components' tokens in order.  Tokens live in PAGES of
 * WRENC_GPU_TOKEN_PAGE words (the last word of a page is the index of the CTU's next page, 0xFFFFFFFF at its end) taken
 * from `pool`; first_page[ctu] (CTUs in raster order) is where a CTU starts.  The CU-level syntax stays on the host: it
 * needs only the maps.  wrenc_bs_write_picture_tokens (wrenc_bitstream.h) writes the same bytes from this record as
 * wrenc_bs_write_picture writes from the level planes, with the residual syntax walk gone from the host.
 * wrenc_gpu_download_tokens reads back `n` slots in one call: pool_cap_words of room in `pool` for all of them together
 * (WRENC_GPU_ENOMEM if it does not suffice: fall back to wrenc_gpu_download_compact), every picture's first_page table
 * and maps, its reconstruction when asked for.  The pages in use are spread over the WHOLE pool (it is cut into up to 64
 * sub-pools that fill side by side, each needing room for its share of the CTUs): page indices are valid up to
 * pool_cap_words -- that is wrenc_bs_tokens::pool_words -- and *pool_words_used is the words of the pages in use, i.e.
 * what crossed the bus (with WRENC_GPU_ENOMEM: of the pages asked for until the pass gave up). */
#define WRENC_GPU_TOKEN_PAGE 64
typedef struct wrenc_gpu_tokens {
    uint32_t* first_page;   /* (width / 32) * (height / 32) entries */
    uint8_t* cu_log2_size;  /* as in wrenc_gpu_picture; any may be NULL */
    uint8_t* luma_mode;
    uint8_t* chroma_mode;
    uint8_t* rec_y;
    uint8_t* rec_cb;
    uint8_t* rec_cr;
} wrenc_gpu_tokens;
int wrenc_gpu_download_tokens(wrenc_gpu_ctx* ctx, int first_slot, int n, wrenc_gpu_tokens* out, uint32_t* pool,
                              size_t pool_cap_words, size_t* pool_words_used);
/* Test entry: put an arbitrary record (maps + level planes, as wrenc_gpu_download fills them) into a slot as if a search
 * had produced it, so that the token pass can be compared with the host-only writer on records no search emits. */
int wrenc_gpu_test_load_record(wrenc_gpu_ctx* ctx, int slot, const wrenc_gpu_picture* rec);

/* Page-locked host memory for the planes handed to wrenc_gpu_upload / wrenc_gpu_download: transfers from
 * and to it run at PCIe rate and truly asynchronously (pageable buffers are staged by the runtime at a
 * fraction of that).  Optional: any host memory works.  Free with wrenc_gpu_free_host before destroy. */
void* wrenc_gpu_alloc_host(wrenc_gpu_ctx* ctx, size_t bytes);
void wrenc_gpu_free_host(wrenc_gpu_ctx* ctx, void* p);

/* Convenience: upload + encode + download of a single picture through slot 0. */
int wrenc_gpu_encode_picture(wrenc_gpu_ctx* ctx, const uint8_t* y, const uint8_t* cb,
                             const uint8_t* cr, wrenc_gpu_picture* out);

/* How an encode call maps CTUs to wavefronts.  Results are identical (bit-exact) either way.
 *   WAVE: one wavefront per CTU, a workgroup = the same CTU of 4 pictures.  Highest throughput, but it needs
 *         hundreds of pictures in flight to fill the GPU (a picture offers only one anti-diagonal of CTUs at a time).
 *   TEAM: four wavefronts per CTU.  After they have searched the CTU's 32x32 candidate together, each takes one level of
 *         the quad-tree and runs ahead (a node's unsplit search and the search of everything below it read only neighbours
 *         outside the node, block_splitter.rs:1081-1123); the levels meet at the split decisions, and the wavefront left
 *         without a level serves candidate packs to the others.  Shorter CTU latency, for calls with few pictures.
 *   AUTO (default): decided per anti-diagonal of CTUs: TEAM while pictures x CTUs of the diagonal cannot fill the GPU
 *         with one wave each, WAVE beyond. */
enum wrenc_gpu_schedule { WRENC_GPU_SCHEDULE_AUTO = 0, WRENC_GPU_SCHEDULE_WAVE = 1, WRENC_GPU_SCHEDULE_TEAM = 2 };
int wrenc_gpu_set_schedule(wrenc_gpu_ctx* ctx, int schedule);
int wrenc_gpu_last_schedule(const wrenc_gpu_ctx* ctx); /* what the most recent encode call used (AUTO = both) */
/* What the library found and how it was built: wavefronts of the search kernel the device holds at once (CUs x 20) and
 * the HIP streams an encode call deals its pictures to.  For measurement tools (bench.py, tools/fill_probe.py). */
int wrenc_gpu_device_info(const wrenc_gpu_ctx* ctx, long long* wave_slots, int* encode_lanes);
/* Test entry: overrides the number of wave slots AUTO compares a diagonal's CTUs x pictures with (default: what the
 * device holds, CUs x waves per CU), so that a SMALL encode call mixes TEAM and WAVE diagonals as a big one does on
 * the real figure (tests/test_gpu_groups.py).  slots <= 0 restores the device's value.  Results do not depend on it. */
int wrenc_gpu_test_set_wave_slots(wrenc_gpu_ctx* ctx, long long slots);
/* Test entry: how many workgroups since context creation found the scratch partition of their XCD full and took a
 * region of the shared overflow partition instead (wrenc_gpu.hip, acquire_scratch).  0 on MI355X: the saved
 * reconstructions of the search then stay behind the L2 of the XCD that wrote them.  Waits for the device. */
int wrenc_gpu_test_scratch_overflows(wrenc_gpu_ctx* ctx, long long* count);
/* Test entry: the dependent quantiser's head proof decides "this coefficient ends the region" and "quotient >= 2" by range
 * tests whose bounds the host derives per context (QP) and block size; this holds them against the device's own
 * formulas over every 16-bit coefficient, 4 block sizes, DC and other positions.  counts[0]: coefficients a range admits
 * that the formula does not (must be 0: results depend on it); counts[1]: the opposite (allowed, costs speed; 0
 * expected); counts[2]: quotient differences (must be 0); counts[3]: differences of the two alpha formulas (must be 0).
 * ranges (may be NULL): the 4 x 6 bounds.  Waits for the device. */
int wrenc_gpu_test_head_ranges(wrenc_gpu_ctx* ctx, int counts[4], int ranges[24]);
/* Test entry: which reference segments of a block are available (below-left, left, corner, above, above-right) comes
 * from a table of the block's place in its CTU plus the picture's edges; this counts the blocks -- every CTU of the
 * context's picture, every size and position, luma and chroma spacing -- at which that differs from the reference's rules
 * (ctu.rs:2083-2188, encoder_context.rs:918-956) evaluated directly.  Must be 0.  Waits for the device. */
int wrenc_gpu_test_avail_tab(wrenc_gpu_ctx* ctx, int* differences);

/* Per-launch timing (two HIP events around every kernel launch) is OFF by default: the product path
 * (CLI, native program) never reads it.  bench.py / profiling switch it on.  While it is on, an encode
 * call first waits for the previous call's end event (the events are re-recorded), so consecutive calls
 * do not overlap: measurement mode only. */
int wrenc_gpu_stats_enable(wrenc_gpu_ctx* ctx, int on);

/* Device time (ms, HIP events on the context's own stream) and launch count of
 * the search kernel during the last wrenc_gpu_encode call (waits for its end).  WRENC_GPU_ESTATE unless
 * timing was on (wrenc_gpu_stats_enable) when that call was queued. */
int wrenc_gpu_last_encode_stats(wrenc_gpu_ctx* ctx, float* total_ms, float* kernel_ms_sum,
                                int* n_launches);
/* The same per kernel: out[0] = ctu_search_kernel (one wavefront per CTU), out[1] = ctu_search_team_kernel (four per
 * CTU): summed HIP-event durations of its launches in the last encode call, their number, and the CTU-pictures
 * (CTUs x pictures) they searched -- what a roofline per launch of ONE kernel needs. */
typedef struct wrenc_gpu_kernel_stats {
    float ms_sum;
    int32_t launches;
    int64_t ctu_pictures;
} wrenc_gpu_kernel_stats;
int wrenc_gpu_last_encode_kernel_stats(wrenc_gpu_ctx* ctx, wrenc_gpu_kernel_stats out[2]);

/* Transform blocks (a luma block, or a Cb + Cr pair) whose final-pass reconstruction differed from what
 * the search left in their place (expected 0; SURVEY.md 3.4 "treat as a property to test"), accumulated
 * since context creation.  Compared through a position-weighted checksum of the block's samples: a
 * change of any one sample is always seen. */
int wrenc_gpu_final_pass_mismatches(wrenc_gpu_ctx* ctx, long long* count);

/* ---- kernel-level entry points (parity tests of the building blocks) ----
 * Each runs `count` independent square blocks of side 1<<log2n (2..5), row-major
 * int16, host pointers.  Same arithmetic as the picture path.
 * Precondition of the 32x32 forward transform (log2n = 5, and wrenc_gpu_test_fwd_dct32 with use_mfma = 1): every
 * residual lies within +-255, as original minus prediction always does -- the i8-MFMA code splits it into two
 * base-256 digits.  Inputs outside that range return WRENC_GPU_EINVAL (nothing is computed). */
int wrenc_gpu_test_fwd_dct(wrenc_gpu_ctx* ctx, const int16_t* res, int log2n, int count,
                           int16_t* coef);                              /* transformer.rs:2040 */
/* The 32x32 forward transform of `count` blocks (residuals within +-255), by the v_dot2 code the search kernel used
 * until round 2 (use_mfma = 0) or by the i8-MFMA version it runs now (use_mfma = 1; wrenc_amd/csrc/dev_transform.h); the kernel repeats the
 * transform `reps` times per block and *kernel_ms receives its HIP-event duration: parity gate and micro-benchmark
 * of north_star's MFMA question in one entry point (transformer.rs:2040-2378). */
int wrenc_gpu_test_fwd_dct32(wrenc_gpu_ctx* ctx, const int16_t* res, int count, int16_t* coef,
                             int use_mfma, int reps, float* kernel_ms);
/* The same for the inverse 32x32 transform (transformer.rs:2380-2737): `count` dequantised blocks in, residuals out;
 * use_mfma = 1 is what the search kernel runs (four v_mfma_i32_32x32x32_i8), 0 the v_dot2 version it replaced. */
int wrenc_gpu_test_inv_dct32(wrenc_gpu_ctx* ctx, const int16_t* deq, int count, int16_t* res,
                             int use_mfma, int reps, float* kernel_ms);
/* The packed quantiser of the 8x8 / 16x16 leaf searches (quantize_pk, wrenc_amd/csrc/dev_quant.h): `n_packs` packs of
 * `nc` candidates (log2n = 3: nc 1..3, log2n = 4: nc 1..2); a pack is nc luma blocks of side 1 << log2n, then per candidate
 * its Cb and Cr block of half that side, all row-major int16 coefficients back to back (nc * 1.5 * 4^log2n values).
 * levels: the same layout; level_cost: per pack and candidate {luma block, chroma pair} (block_splitter.rs:436-458). */
int wrenc_gpu_test_quantize_pk(wrenc_gpu_ctx* ctx, const int16_t* coef, int log2n, int nc, int n_packs, int16_t* levels,
                               int64_t* level_cost);
int wrenc_gpu_test_inv_dct(wrenc_gpu_ctx* ctx, const int16_t* deq, int log2n, int count,
                           int16_t* res);                               /* transformer.rs:2380 */
int wrenc_gpu_test_quantize(wrenc_gpu_ctx* ctx, const int16_t* coef, int log2n, int count,
                            int16_t* levels, int64_t* level_cost);      /* quantizer.rs:519 +
                                                                           block_splitter.rs:415-460 */
/* The same quantiser as the packed 4x4 leaf search runs it: `count` 4x4 blocks, up to four per wavefront at once
 * (quantizer.rs:519 + block_splitter.rs:415-460; wrenc_amd/csrc/dev_quant.h quantize_p16). */
int wrenc_gpu_test_quantize_p16(wrenc_gpu_ctx* ctx, const int16_t* coef, int count, int16_t* levels,
                                int64_t* level_cost);
int wrenc_gpu_test_dequantize(wrenc_gpu_ctx* ctx, const int16_t* levels, int log2n, int count,
                              int16_t* deq);                            /* quantizer.rs:761 */

/* Intra prediction of single blocks (intra_predictor.rs:56-144: reference-sample build :146-353, PLANAR
 * :759-1146, DC :1148-1285, ANGULAR 2..66 :1287-1602, PDPC :355-757, CCLM :1604-2055) in the environment of
 * given reconstruction planes of the context's picture size.  items: n_items x 5 int32 {x, y (luma, picture
 * coordinates, multiples of the size), log2 luma size 2..5, comp (0 = luma block, 1 = Cb+Cr pair of the block,
 * log2 size >= 3, 2 = the 4x4 luma block through the row-parallel predictor of the packed 4x4 leaf search, log2 size 2),
 * mode (0..66, or 81..83 for comp 1)}.  out: the predicted samples, item after item (luma
 * n x n; pair: Cb (n/2)^2 then Cr (n/2)^2); out_bytes must equal their total.
 * comp 4 / 5 / 6: not a prediction but a SAD LIST of the search (get_intra_pred_aux_cost, block_splitter.rs:64-108, of each
 * entry) over the luma block / the chroma pair (log2 size >= 3) / both, against the block's own samples in the planes as
 * originals: mode = first mode (2..66) | entries (1..13) << 8 | stride (1..64) << 16, entry j = first mode + j * stride (an
 * entry beyond 66 is not evaluated and reads 0).  comp 7 (log2 size >= 3, mode 0): the CCLM SAD list of the chroma pair,
 * entries LT_CCLM, T_CCLM, L_CCLM (get_chroma_intra_pred_aux_cost, :476-522).  A list item's output is 16 uint32 (64 bytes):
 * the SAD of entry j at index j. */
int wrenc_gpu_test_predict(wrenc_gpu_ctx* ctx, const uint8_t* rec_y, const uint8_t* rec_cb,
                           const uint8_t* rec_cr, int n_items, const int32_t* items, uint8_t* out,
                           size_t out_bytes);

#ifdef __cplusplus
}
#endif
#endif /* WRENC_GPU_H */
