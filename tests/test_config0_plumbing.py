"""BASELINE.json configs[0] / SURVEY.md 8d config 1 (plumbing, no GPU): 30 pictures of synthetic 352x288 at QP32,
max-split-depth 0, through the CPU oracle, the host bitstream writer and back through the stream parser.  This is
the reference's own CPU-runnable case (bus CIF needs ffmpeg, so the synthetic sequence of the bench stands in)."""
import numpy as np


def test_cif_30_pictures_depth0_stream(built):
    from oracle import pyoracle as po
    from wrenc_amd import bitstream as bs, synth
    w, h, qp, n = 352, 288, 32, 30
    recs = []
    stream = bs.write_parameter_sets(w, h, qp)
    for f in range(n):
        rec = po.encode_picture(*synth.synth_frame(w, h, f), qp, 0)
        assert rec["final_pass_mismatches"] == 0 and np.all(rec["cu_log2_size"] == 5)
        recs.append(rec)
        stream += bs.write_picture(w, h, qp, f, rec)
    info = po.parse_stream_info(stream)
    assert info == {"width": w, "height": h, "init_qp": qp, "n_pictures": n}
    for f in (0, 7, 15, 16, 29):
        back = po.parse_picture(stream, f)
        assert back["poc_lsb"] == f & 15
        for k in ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr"):
            assert np.array_equal(back[k], recs[f][k]), (f, k)
        ry, rcb, rcr = po.spec_decode_record(back, qp)      # independent decoder (oracle/spec_decoder.cpp)
        assert np.array_equal(ry, recs[f]["rec_y"]) and np.array_equal(rcb, recs[f]["rec_cb"])
        assert np.array_equal(rcr, recs[f]["rec_cr"])
    assert 20_000 < len(stream) < 3_000_000
