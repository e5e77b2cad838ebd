import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np
from content import content
from wrenc_amd import gpu, bitstream as bs, synth
from oracle import pyoracle as po
cases = [("noise", 64, 64, 63, 3), ("extremes", 64, 64, 63, 3), ("cclm", 96, 64, 4, 2), ("noise", 64, 64, 12, 3), ("checker", 64, 64, 8, 3), ("stripes45", 64, 64, 60, 1),
         ("tex", 512, 32, 32, 3), ("tex", 32, 512, 22, 3), ("tex", 32, 32, 51, 3), ("noise", 128, 32, 16, 3)]
for kind, w, h, qp, depth in cases:
    y, cb, cr = synth.synth_textured_frame(w, h, 3) if kind == "tex" else content(kind, w, h, 17)
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth)
    try:
        got = enc.encode_picture(y, cb, cr)
    except gpu.WrencGpuError as e:
        print(kind, w, h, qp, "device:", repr(e)[:90]); enc.close(); continue
    pool, pics = enc.download_tokens(0, 1)
    enc.close()
    a = bs.write_picture(w, h, qp, 0, got)
    b = bs.write_picture_tokens(w, h, qp, 0, pool, pics[0])
    stream = bs.write_parameter_sets(w, h, qp) + a
    back = po.parse_picture(stream, 0)
    ok_rec = all(np.array_equal(x, got[k]) for x, k in zip(po.spec_decode_record(back, qp), ("rec_y", "rec_cb", "rec_cr")))
    ok_lev = all(np.array_equal(back[k], got[k]) for k in ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr"))
    print(kind, w, h, "qp", qp, "depth", depth, "bytes", len(a), "tokens==planes", a == b, "parsed==record", ok_lev, "spec decode==recon", ok_rec, flush=True)
