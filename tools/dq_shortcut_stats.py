"""How often the quantiser's shortcuts (oracle/wrenc_oracle.cpp, quantize_viterbi_sc) apply on the search's own blocks.

Runs the CPU oracle over a crop of the bench contents with the model enabled: every quantiser call of the search is also
run through the model (levels must agree) and the sub-blocks are counted: skipped by the head proof, taken in closed form
(all quotients zero), walked.  CPU only.   usage: dq_shortcut_stats.py [W H [DEPTH [QP ...]]]
"""
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from wrenc_amd import synth  # noqa: E402


def main():
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    depth = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    qps = [int(a) for a in sys.argv[4:]] or [32]
    for qp in qps:
        for name, fn in (("smooth", synth.synth_frame), ("textured", synth.synth_textured_frame)):
            # a crop of the full-size picture keeps the content's scale (the smooth pattern depends on the picture size)
            y, cb, cr = fn(1920, 1088, 0)
            y, cb, cr = y[:h, :w].copy(), cb[:h // 2, :w // 2].copy(), cr[:h // 2, :w // 2].copy()
            po.dq_sc_stats_enable(True)
            po.encode_picture(y, cb, cr, qp, depth)
            mism, st = po.dq_sc_stats_read()
            po.dq_sc_stats_enable(False)
            print("QP %d depth %d %s %dx%d: model mismatches %d" % (qp, depth, name, w, h, mism))
            for l in (2, 3, 4, 5):
                s = st[l]
                if not s["blocks"]:
                    continue
                sb = max(s["sub_blocks"], 1)
                print("  %2dx%-2d blocks %7d non-zero %5.1f %%, proven all zero without a walk %5.1f %% of those | sub-blocks of those: head-skipped %5.1f %%, closed form %5.1f %% "
                      "(eligible %5.1f %%), walked %5.1f %% | head tests %d failed %d | four segments: %5.1f %% of the sub-blocks in "
                      "segments 0..2, %5.1f %% of those not walked again"
                      % (1 << l, 1 << l, s["blocks"], 100.0 * s["nz_blocks"] / s["blocks"], 100.0 * s["whole_zero"] / max(s["nz_blocks"], 1),
                         100.0 * s["head_sb_skipped"] / sb,
                         100.0 * s["z_pass"] / sb, 100.0 * s["z_eligible"] / sb, 100.0 * s["walked"] / sb, s["head_tests"],
                         s["head_fail"], 100.0 * s["seg_sb"] / sb, 100.0 * s["seg_kept"] / max(s["seg_sb"], 1)))


if __name__ == "__main__":
    main()
