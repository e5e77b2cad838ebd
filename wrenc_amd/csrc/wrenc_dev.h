// wrenc_dev.h -- CDNA4 (gfx950) device code of the all-intra RD-search path.
//
// Execution model: ONE 64-lane wavefront owns one 32x32 CTU; a workgroup is WPB = 4 waves working on
// the SAME CTU position of 4 different pictures, so the waves make the same sequence of full
// evaluations and pool their one serial stage (the Viterbi walk) in one wave.  A wave keeps its
// working set in LDS (transform buffers, cached reference samples, the reconstruction tile with its
// neighbour border, trellis decisions, decision maps, search state: 7.2 KB, sized so that five workgroups
// = 20 waves fit a CU: the kernel is latency-bound per wave and its throughput follows the waves in
// flight); originals come from the picture's CTU tile (L1 / L2) or a staged copy in LDS, saved
// reconstructions live in a small pool of global scratch.  Search control
// is a scalar state machine that hands evaluation requests to one inlined evaluator: no device function
// calls, no scratch, every branch scalar.
//
// What is computed follows the reference function by function (paths relative to the reference's
// src/, cited at each function); how it is computed is wave-parallel:
//   dev_predict.h    intra_predictor.rs:56-2055     lane = sample; SAD lists of many modes in one loop
//   dev_transform.h  transformer.rs:2040-2737       lane = basis row, v_dot2 / v_mad_i24
//   dev_quant.h      quantizer.rs:338-759           backward 4-state Viterbi, one lane per state, DPP
//   dev_search.h     block_splitter.rs:64-1154, ctu_encoder.rs:1421-1461
//   dev_bins.h       ctu_encoder.rs:1786-2269       residual_coding as a token stream for the host's arithmetic coder
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev_common.h"
#include "dev_predict.h"
#include "dev_transform.h"
#include "dev_quant.h"
#include "dev_search.h"
#define WRENC_TOKENS_KERNEL_TU
#include "dev_bins.h"
