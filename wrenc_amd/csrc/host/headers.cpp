// headers.cpp -- parameter sets, picture header, slice header and NAL framing.
//
// The reference fills its parameter-set structs with fixed defaults (vps.rs:87-127, sps.rs:186-347,
// pps.rs:147-197, picture_header.rs:86-142, slice_header.rs:63-122), so every payload is a function of
// (width, height, QP, POC).  Each writer below emits the syntax elements the reference's encoders emit
// for those defaults, in their order (vps_encoder.rs, sps_encoder.rs, pps_encoder.rs, ph_encoder.rs,
// slice_encoder.rs:32-341 with ptl/gci/dpbp/rpl_encoder.rs); branches that the defaults never take are
// left out, the value each skipped condition has is noted where it matters.
#include "slice_data.h"

namespace wrenc_host {

namespace {

// ptl_encoder.rs:31-79 with ProfileTierLevel::new(true) (ptl.rs:17-31): profile/tier/level 0, no GCI,
// max_num_sublayers 1 so no sub-layer loop
void profile_tier_level(BitWriter& bw) {
    bw.put(0, 7); // general_profile_idc
    bw.bit(0);    // general_tier_flag
    bw.put(0, 8); // general_level_idc
    bw.bit(0);    // ptl_frame_only_constraint_flag
    bw.bit(0);    // ptl_multilayer_enabled_flag
    bw.bit(0);    // gci_present_flag (gci_encoder.rs:27-28), then byte_align (:110)
    bw.align();
    bw.align();   // ptl_encoder.rs:66
    bw.put(0, 8); // ptl_num_sub_profiles
}

// dpbp_encoder.rs:26-51 with DpbParameter::new (dpb.rs:11-19), one sub-layer
void dpb_parameters(BitWriter& bw) {
    bw.ue(8); // max_dec_pic_buffering
    bw.ue(4); // max_num_reorder_pics
    bw.ue(1); // max_latency_increase
}

// rpl_encoder.rs:76-126 with RefPicListStruct::new(lx) (reference_picture.rs:13-25): three short-term
// entries, abs_delta_poc_st {0, 2, 3}, sign flag = (lx == 0)
void ref_pic_list_struct(BitWriter& bw, int lx) {
    static const int kAbsDeltaPocSt[3] = {0, 2, 3};
    bw.ue(3);
    for (int i = 0; i < 3; ++i) {
        bw.ue((uint64_t)kAbsDeltaPocSt[i]);
        bw.bit(lx == 0); // abs_delta_poc_st + 1 > 0 always (no weighted prediction)
    }
}

void trailing_bits(BitWriter& bw) {
    bw.bit(1);
    bw.align();
}

} // namespace

// vps_encoder.rs:28-279, VideoParameterSet::new(8, ..) (main.rs:223)
void write_vps(BitWriter& bw, int, int) {
    bw.put(8, 4); // vps id
    bw.put(0, 6); // max_layers - 1
    bw.put(0, 3); // max_sublayers - 1
    bw.put(9, 6); // layer id of layer 0 (vps.rs:98)
    bw.align();   // :110 (one PTL, default_ptl_dpb_hrd_max_tid_flag 1)
    profile_tier_level(bw);
    // each_layer_is_an_ols is false in the struct (vps.rs:99), so the DPB part is written although
    // max_layers is 1 (:147-205): one dpb_parameters, no multi-layer OLS entries
    bw.ue(0); // num_dpb_params - 1
    dpb_parameters(bw);
    bw.bit(0); // general_timing_hrd_parameters absent (:207-208)
    bw.bit(0); // extension data absent (:264-265)
    trailing_bits(bw);
}

// sps_encoder.rs:29-545, SequenceParameterSet::new(1, 8, w, h, 8) (main.rs:236)
void write_sps(BitWriter& bw, int width, int height) {
    bw.put(1, 4); // sps id
    bw.put(8, 4); // vps id
    bw.put(0, 3); // max_sublayers - 1
    bw.put(1, 2); // chroma_format 4:2:0
    bw.put(0, 2); // log2_ctu_size - 5
    bw.bit(1);    // ptl_dpb_hrd_params_present_flag
    profile_tier_level(bw);
    bw.bit(0); // gdr_enabled_flag
    bw.bit(0); // ref_pic_resampling_enabled_flag
    bw.ue((uint64_t)width);
    bw.ue((uint64_t)height);
    bw.bit(0);    // conformance window
    bw.bit(0);    // subpic info
    bw.ue(0);     // bitdepth - 8
    bw.bit(0);    // entropy_coding_sync_enabled_flag
    bw.bit(0);    // entry_point_offsets_present_flag
    bw.put(0, 4); // log2_max_pic_order_cnt_lsb - 4
    bw.bit(0);    // poc_msb_cycle_flag
    bw.put(0, 2); // num_extra_ph_bytes
    bw.put(0, 2); // num_extra_sh_bytes
    dpb_parameters(bw);
    bw.ue(0);  // log2_min_luma_coding_block_size - 2
    bw.bit(0); // partition_constraints_override_enabled_flag
    bw.ue(0);  // log2_diff_min_qt_min_cb_intra_slice_luma
    bw.ue(0);  // max_mtt_hierarchy_depth_intra_slice_luma
    bw.bit(0); // qtbtt_dual_tree_intra_flag
    bw.ue(0);  // log2_diff_min_qt_min_cb_inter_slice
    bw.ue(0);  // max_mtt_hierarchy_depth_inter_slice
    // CtbSizeY is 32: no max_luma_transform_size_64_flag (:265-267)
    bw.bit(1); // transform_skip_enabled_flag
    bw.ue(5);  // log2_transform_skip_max_size, written without the minus2 (:270-271)
    bw.bit(0); // bdpcm_enabled_flag
    bw.bit(1); // mts_enabled_flag
    bw.bit(1); // explicit_mts_intra_enabled_flag
    bw.bit(1); // explicit_mts_inter_enabled_flag
    bw.bit(0); // lfnst_enabled_flag
    bw.bit(0); // joint_cbcr_enabled_flag
    bw.bit(1); // same_qp_table_for_chroma_flag
    // one identity chroma QP table, QpTable::new(bit_depth, 63, 0) (sps.rs:33-56, :291-311)
    bw.se(0 - 26); // qp_table_start - 26
    bw.ue(62);     // num_points_in_qp_table - 1
    for (int j = 0; j < 63; ++j) {
        bw.ue(0); // delta_qp_in_val - 1
        bw.ue(1); // delta_qp_diff_val
    }
    bw.bit(0); // sao_enabled_flag
    bw.bit(0); // alf_enabled_flag
    bw.bit(0); // lmcs_enabled_flag
    bw.bit(0); // weighted_pred_flag
    bw.bit(0); // weighted_bipred_flag
    bw.bit(0); // long_term_ref_pics_flag
    bw.bit(0); // inter_layer_prediction_enabled_flag (vps id > 0)
    bw.bit(0); // idr_rpl_present_flag
    bw.bit(0); // rpl1_same_as_rpl0_flag
    for (int lx = 0; lx < 2; ++lx) {
        bw.ue(1); // num_ref_pic_list
        ref_pic_list_struct(bw, lx);
    }
    bw.bit(0); // ref_wraparound_enabled_flag
    bw.bit(0); // temporal_mvp_enabled_flag
    bw.bit(0); // amvr_enabled_flag
    bw.bit(0); // bdof_enabled_flag
    bw.bit(0); // smvd_enabled_flag
    bw.bit(0); // dmvr_enabled_flag
    bw.bit(0); // mmvd_enabled_flag
    bw.ue(0);  // six_minus_max_num_merge_cand
    bw.bit(0); // sbt_enabled_flag
    bw.bit(0); // affine_enabled_flag
    bw.bit(0); // bcw_enabled_flag
    bw.bit(0); // ciip_enabled_flag
    bw.bit(0); // gpm_enabled_flag (MaxNumMergeCand 6 >= 2)
    bw.ue(0);  // log2_parallel_merge_level - 2
    bw.bit(0); // isp_enabled_flag
    bw.bit(0); // mrl_enabled_flag
    bw.bit(0); // mip_enabled_flag
    bw.bit(1); // cclm_enabled_flag
    bw.bit(0); // chroma_horizontal_collocated_flag
    bw.bit(0); // chroma_vertical_collocated_flag
    bw.bit(0); // palette_enabled_flag
    bw.ue(0);  // min_qp_prime_ts (transform skip enabled)
    bw.bit(0); // ibc_enabled_flag
    bw.bit(0); // ladf parameters
    bw.bit(0); // explicit_scaling_list_enabled_flag
    bw.bit(1); // dep_quant_enabled_flag
    bw.bit(0); // sign_data_hiding_enabled_flag
    bw.bit(0); // virtual_boundaries_enabled_flag
    bw.bit(0); // timing_hrd_params_present_flag
    bw.bit(0); // field_seq_flag
    bw.bit(0); // vui_parameters_present_flag
    bw.bit(0); // extension data
    trailing_bits(bw);
}

// pps_encoder.rs:25-350, PictureParameterSet::new(1, sps, qp) (main.rs:248): init_qp = max(qp, 26)
void write_pps(BitWriter& bw, int width, int height, int qp) {
    bw.put(1, 6); // pps id
    bw.put(1, 4); // sps id
    bw.bit(0);    // mixed_nalu_types_in_pic_flag
    bw.ue((uint64_t)width);
    bw.ue((uint64_t)height);
    bw.bit(0); // conformance window
    bw.bit(0); // scaling_window_explicit_signalling_flag
    bw.bit(0); // output_flag_present_flag
    bw.bit(1); // no_pic_partition_flag
    bw.bit(0); // subpic_id_mapping_present_flag
    bw.bit(0); // cabac_init_present_flag
    bw.ue(2);  // num_ref_idx_default_active[0] - 1
    bw.ue(2);  // num_ref_idx_default_active[1] - 1
    bw.bit(0); // rpl1_idx_present_flag
    bw.bit(0); // weighted_pred_flag
    bw.bit(0); // weighted_bipred_flag
    bw.bit(0); // ref_wraparound_enabled_flag
    bw.se((qp > 26 ? qp : 26) - 26); // init_qp - 26
    bw.bit(1); // cu_qp_delta_enabled_flag
    bw.bit(0); // chroma_tool_offsets_present_flag
    bw.bit(1); // deblocking_filter_control_present_flag
    bw.bit(0); // deblocking_filter_override_enabled_flag
    bw.bit(1); // deblocking_filter_disabled_flag
    bw.bit(0); // picture_header_extension_present_flag
    bw.bit(0); // slice_header_extension_present_flag
    bw.bit(0); // extension data
    trailing_bits(bw);
}

// ph_encoder.rs:30-425, PictureHeader::new(pps, intra = true, poc) (main.rs:297)
void write_picture_header(BitWriter& bw, int poc) {
    bw.bit(1);         // gdr_or_irap_pic_flag
    bw.bit(0);         // non_ref_pic_flag
    bw.bit(0);         // gdr_pic_flag
    bw.bit(0);         // inter_slice_allowed_flag
    bw.ue(1);          // pps id
    bw.put((uint64_t)(poc & 15), 4); // pic_order_cnt_lsb (picture_header.rs:99)
    bw.ue(0);          // cu_qp_delta_subdiv_intra_slice (cu_qp_delta_enabled_flag)
    trailing_bits(bw);
}

// slice_encoder.rs:32-341, SliceHeader::new (slice_header.rs:63-122): qp_delta = qp - init_qp
void write_slice_header(BitWriter& bw, int qp) {
    bw.bit(0); // picture_header_in_slice_header_flag
    bw.bit(0); // no_output_of_prior_pics_flag (IDR_W_RADL)
    bw.se(qp - (qp > 26 ? qp : 26));
    bw.bit(1); // dep_quant_used_flag
    bw.bit(1); // byte_alignment: bit equal to one, then zeros (:339-341)
    bw.align();
}

void append_nal(std::vector<uint8_t>& out, int layer_id, NalType type, int temporal_id,
                const std::vector<uint8_t>& p) {
    static const uint8_t kPrefix[6] = {0, 0, 0, 0, 0, 1};
    out.insert(out.end(), kPrefix, kPrefix + 6);
    // forbidden_zero_bit, nuh_reserved_zero_bit, nuh_layer_id u(6) | nal_unit_type u(5), temporal_id_plus1 u(3)
    out.push_back((uint8_t)(layer_id & 63));
    out.push_back((uint8_t)(((int)type << 3) | ((temporal_id + 1) & 7)));
    // the reference's emulation prevention (nal.rs:156-182,273-297): a window of three payload bytes,
    // the last three bytes of the payload are copied without being examined
    size_t i = 0;
    while (i + 3 < p.size()) {
        if (p[i] == 0 && p[i + 1] == 0 && p[i + 2] <= 3) {
            out.push_back(0);
            out.push_back(0);
            out.push_back(3);
            i += 2;
        } else {
            out.push_back(p[i++]);
        }
    }
    for (; i < p.size(); ++i) out.push_back(p[i]);
}

} // namespace wrenc_host
