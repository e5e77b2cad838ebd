#!/usr/bin/env python3
"""Extract the I-slice context initialisation data of the live syntax elements from the reference
(src/cabac_contexts.rs:243-917, ctx_table[ctx][0][0] = initValue, ctx_table[ctx][1][0] = shiftIdx;
:919 c_rice_params) into tests/golden/cabac_ctx_init.json.  Data only (the numbers of the VVC
specification's tables as the reference holds them); run once where /root/reference exists."""
import ast
import json
import os
import re

SRC = "/root/reference/src/cabac_contexts.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cabac_ctx_init.json")
# CabacContext discriminants (cabac_contexts.rs:16-127) of the elements the all-intra path codes
LIVE = [("split_cu_flag", 16), ("intra_luma_mpm_flag", 34), ("intra_luma_not_planar_flag", 35),
        ("cclm_mode_flag", 40), ("cclm_mode_idx", 41), ("intra_chroma_pred_mode", 42), ("mts_idx", 67),
        ("tu_y_coded_flag", 87), ("tu_cb_coded_flag", 88), ("tu_cr_coded_flag", 89),
        ("cu_qp_delta_abs", 90), ("transform_skip_flag", 94), ("last_sig_coeff_x_prefix", 96),
        ("last_sig_coeff_y_prefix", 97), ("sb_coded_flag", 100), ("sig_coeff_flag", 101),
        ("par_level_flag", 102), ("abs_level_gtx_flag", 103)]


def main():
    s = open(SRC).read()
    i = s.index("pub static ref ctx_table")
    j = s.index("= vec![", i) + 2
    k = s.index("];", j)
    body = re.sub(r"//[^\n]*", "", s[j:k + 1]).replace("vec![", "[")
    table = ast.literal_eval(body)
    m = re.search(r"c_rice_params: \[usize; 32\] = \[([^\]]*)\]", s)
    rice = [int(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]
    out = {"order": [n for n, _ in LIVE], "contexts": {}, "c_rice_params": rice}
    for name, idx in LIVE:
        out["contexts"][name] = {"init_value": table[idx][0][0], "shift_idx": table[idx][1][0]}
    json.dump(out, open(OUT, "w"), indent=1)
    print("wrote", OUT, sum(len(v["init_value"]) for v in out["contexts"].values()), "contexts")


if __name__ == "__main__":
    main()
