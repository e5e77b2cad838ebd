// wrenc_oracle.h -- TEST INFRASTRUCTURE ONLY.
//
// CPU restatement ("oracle") of the per-CTU RD-search hot path of hjmkt/wrenc
// (reference paths below are relative to the reference tree's src/):
//   block_splitter.rs  (split_ct, get_intra_pred_cost, ...)
//   intra_predictor.rs (PLANAR / DC / ANGULAR / CCLM / PDPC, reference samples)
//   transformer.rs     (integer DCT-2 4/8/16/32 forward + inverse)
//   quantizer.rs       (dependent quantisation trellis, dequantisation)
//   ctu.rs             (CT split, MPM derivation, availability walkers)
//   ctu_encoder.rs:1421-1461 (final pass)
//
// PARITY UNPINNED: the reference is Rust; no Rust toolchain exists in the build
// image and the reference ships no golden vectors for this path, so this
// restatement cannot be checked against reference output.  It follows the
// reference source line by line (scalar arms, release-profile wrap semantics).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may link
// or call this code.  The product path (wrenc_amd/) never does.
#pragma once
#include <cstdint>
#include <cstddef>

#ifdef __cplusplus
extern "C" {
#endif

// Resolved RD-model constants (defaults of block_splitter.rs:187-375,594-693 and
// quantizer.rs:16-25,650-683; every live call site passes trellis=true and
// dep_quant_used_flag=true, so only the *_dq_trellis variants are live).
typedef struct wro_params {
    int width;            // luma, multiple of 32 (picture.rs:178-181 panics otherwise)
    int height;
    int qp;               // fixed QP (main.rs:193-198); 26 when --qp absent
    int max_split_depth;  // 0..3 (main.rs:108-109)
} wro_params;

// Output of one picture.  All buffers caller-allocated.
typedef struct wro_picture_out {
    uint8_t* rec_y;        // width*height
    uint8_t* rec_cb;       // (width/2)*(height/2)
    uint8_t* rec_cr;
    int16_t* lev_y;        // TransCoeffLevel planes (final pass, ctu_encoder.rs:1437)
    int16_t* lev_cb;
    int16_t* lev_cr;
    uint8_t* cu_log2_size; // (width/4)*(height/4): log2 of the luma CU covering each 4x4
    uint8_t* luma_mode;    // (width/4)*(height/4): CU intra_pred_mode[0]
    uint8_t* chroma_mode;  // (width/8)*(height/8): TU cu_intra_pred_mode[1] of the chroma block
    float* ctu_cost;       // one per CTU (return value of split_ct at ctu_encoder.rs:54)
} wro_picture_out;

// Encode (search + final pass) one picture. Returns 0 on success.
int wro_encode_picture(const wro_params* p, const uint8_t* y, const uint8_t* cb,
                       const uint8_t* cr, wro_picture_out* out);

// Decoder-side reconstruction of a picture from its output record (cu_log2_size, luma_mode,
// chroma_mode, lev_*): predict -> dequantize -> inverse transform -> clip per CU in coding order.
// Equality with rec_* is the in-repo form of the reference's integration test (decoded == recon).
int wro_reconstruct_from_record(const wro_params* p, const wro_picture_out* record, uint8_t* out_y,
                                uint8_t* out_cb, uint8_t* out_cr);

// Kernel-level parity of the prediction (intra_predictor.rs:56-144): predicted blocks in the environment of
// given reconstruction planes.  Item = 6 ints {x, y, log2 luma size, tree type, component, mode}; out =
// the component's blocks back to back.  See the definition for the details.
int wro_predict_blocks(const wro_params* p, const uint8_t* rec_y, const uint8_t* rec_cb, const uint8_t* rec_cr,
                       int n_items, const int32_t* items, uint8_t* out);

// Final-pass consistency: number of samples where the final pass recon differs
// from the recon the search left in the planes (expected 0).
long wro_last_final_pass_mismatches(void);
// debug: ssd and level (incl. header bits) of the most recent get_intra_pred_cost call
void wro_debug_last_cost(unsigned long long* ssd, long long* level);

// ---- building blocks exposed for kernel-level parity tests ----
// transformer.rs:2040-2378 (DCT-2 only). in/out: n*n row-major, n = 1<<log2n.
void wro_fwd_dct(const int16_t* res, int log2n, int16_t* coef);
// transformer.rs:2380-2737
void wro_inv_dct(const int16_t* deq, int log2n, int16_t* res);
// quantizer.rs:519-759 with the literal memoised DFS search_dq (:338-517)
void wro_quantize(const int16_t* coef, int log2n, int qp, int16_t* levels);
// same result via backward 4-state Viterbi; used to prove the
// equivalence the GPU kernel relies on
void wro_quantize_viterbi(const int16_t* coef, int log2n, int qp, int16_t* levels);
// MODEL of the device quantiser's shortcuts (round 4; see quantize_viterbi_sc in the .cpp): the same Viterbi with the
// "head proven zero" (use_head) and "all-quotient-zero sub-block in closed form" (use_z bit 0) exits and the long linear part
// of a chain walked as four segments side by side (use_z bit 1).  Must equal wro_quantize.
void wro_quantize_viterbi_sc(const int16_t* coef, int log2n, int qp, int16_t* levels, int use_head, int use_z);
// while enabled, every quantiser call of wro_encode_picture also runs the model on the same coefficients; read returns
// the number of blocks whose levels differed (expected 0) and 6 x 11 counters by log2n: blocks, non-zero blocks, their
// sub-blocks, head sub-blocks skipped, head tests, head tests failed, (Z)-eligible sub-blocks, (Z) passed, walked; and of a
// second run with the segmented walk alone: sub-blocks of its segments 0..2, those of them not walked a second time; blocks
// proven all zero without any walk (the head proof over the whole block; use_z bit 2 switches that off)
void wro_dq_sc_stats_enable(int on);
long long wro_dq_sc_stats_read(long long* out72);
// quantizer.rs:761-1079
void wro_dequantize(const int16_t* levels, int log2n, int qp, int16_t* deq);
// block_splitter.rs:415-460 level-cost walk of one TB (no header bits)
int64_t wro_level_cost(const int16_t* levels, int log2n);

// Constant tables (block_splitter.rs:29-53, quantizer.rs:16-25,650-683,
// block_splitter.rs:472). Used by tests to compare with the product's resolver.
void wro_tables(int qp, int64_t* lv_table1024, int64_t* dq_table1024,
                int64_t* lambda_q, float* lambda_rd);
// header bits ( (x*16384.0) as i64 ) for the luma/single-tree cost function
// (block_splitter.rs:377-406). tree: 0 single, 1 dual luma, 2 dual chroma.
// non_planar: 0/1; mpm_flag; mpm_idx; mpm_rem; cclm_flag; cclm_idx
int64_t wro_header_bits(int tree, int non_planar, int mpm_flag, int mpm_idx,
                        int mpm_rem, int cclm_flag, int cclm_idx);
// chroma cost function header bits (block_splitter.rs:695-712)
int64_t wro_chroma_header_bits(int cclm_flag, int cclm_idx);

// --extra-params of the reference (main.rs:202-217; the RD-model knobs of block_splitter.rs:21-53,187-375,
// 594-693,775 and quantizer.rs:16-19,650-683): applies to every later call in this process; NULL or ""
// restores the defaults.  Returns -1 on an item that is not KEY=VALUE.
int wro_set_extra_params(const char* text);
float wro_lambda_rd_chroma(int qp);

// Test hook: make the oracle misread ONE constant on purpose (0 none; 1 PDPC rounding offset 32 -> 31,
// intra_predictor.rs:747-752; 2 CCLM down-sampling offset 4 -> 3, :1855-1868; 3 the level-scale entry + 1,
// quantizer.rs:8,617-622; 4 inverse-transform first-stage offset 64 -> 63, transformer.rs:2569-2580).  The
// perturbed oracle stays self-consistent (its own decoder-side reconstruction still agrees with it); the
// independent spec decoder (spec_decoder.cpp) must disagree.  tests/test_spec_decoder.py.
void wro_debug_perturb(int which);

// Trace of every candidate evaluation of the search (block_splitter.rs:64-108, 110-474, 476-522,
// 524-780): enable, run wro_encode_picture, read.  A record is 8 int32 words: x, y (luma, picture
// coordinates of the CU), log2 size, tree type (0 single, 1 dual luma, 2 dual chroma), kind
// (0 aux cost, 1 full cost, 2 chroma aux cost, 3 chroma full cost), luma mode (0 for kinds 2, 3),
// chroma mode, and the bits of the f32 the function returned.  Returns the number of records made.
void wro_trace_enable(int on);
long wro_trace_read(int32_t* out, long max);

// 64x64 DCT-2 matrix rows (transformer.rs:934-1191 after symmetric extension)
void wro_dct64(int16_t* m64x64);

#ifdef __cplusplus
}
#endif
