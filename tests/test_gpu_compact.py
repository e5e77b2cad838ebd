"""The compact read-back (include/wrenc_gpu.h: wrenc_gpu_download_compact + wrenc_gpu_expand_levels): the mask of coded
4x4 blocks and their levels, packed on the device behind the search, expand on the host to exactly the level planes the
plain read-back copies -- several pictures per call, pictures with no coded block at all, a picture that needs more
payload room than the caller gave."""
import numpy as np
import pytest

from content import content

pytestmark = pytest.mark.gpu


def test_compact_readback_expands_to_the_level_planes(built):
    from wrenc_amd import bitstream as bs, gpu, synth
    w, h, qp, depth = 160, 96, 30, 3
    grey = (np.full((h, w), 128, np.uint8), np.full((h // 2, w // 2), 128, np.uint8), np.full((h // 2, w // 2), 128, np.uint8))
    frames = [synth.synth_textured_frame(w, h, 1), grey, synth.synth_frame(w, h, 2), content("noise", w, h, 5),
              content("extremes", w, h, 6)]
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=len(frames) + 1)
    for s, f in enumerate(frames):
        enc.upload(1 + s, *f)
    enc.encode(1, len(frames))
    recs = [enc.download(1 + s) for s in range(len(frames))]
    packs = enc.download_compact(1, len(frames))
    blocks_total = (w // 4) * (h // 4) * 3 // 2
    seen_empty = False
    for s, (mask, payload, maps) in enumerate(packs):
        ly, lcb, lcr = enc.expand_levels(mask, payload)
        assert np.array_equal(ly, recs[s]["lev_y"]) and np.array_equal(lcb, recs[s]["lev_cb"]) and np.array_equal(lcr, recs[s]["lev_cr"]), s
        for k in ("cu_log2_size", "luma_mode", "chroma_mode"):
            assert np.array_equal(maps[k], recs[s][k]), (s, k)
        bits = int(sum(bin(int(x)).count("1") for x in mask))
        assert bits == payload.shape[0] <= blocks_total
        assert all(np.any(payload[i]) for i in range(payload.shape[0]))     # only coded blocks travel
        seen_empty = seen_empty or bits == 0
        # the stream written from the expanded planes is the stream written from the plain read-back
        rec = dict(maps, lev_y=ly, lev_cb=lcb, lev_cr=lcr)
        assert bs.write_picture(w, h, qp, s, rec) == bs.write_picture(w, h, qp, s, recs[s])
    assert seen_empty                                                        # the mid-grey picture codes nothing
    with pytest.raises(gpu.WrencGpuError) as e:                              # too little room: reported, not truncated
        enc.download_compact(1, 1, payload_cap=3)
    assert e.value.code == -3
    enc.close()
