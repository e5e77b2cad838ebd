"""The multi-group, multi-lane split of one encode call against the oracle: an encode call
deals its pictures to workgroups of WPB = 4 pictures and the workgroups to up to 4 HIP streams ("lanes");
the last workgroup is padded with waves that compute and never store.  19 pictures = 5 workgroups on 4 lanes
with 1 padding wave; 33 pictures = 9 workgroups on 4 lanes with 3 padding waves; 9 pictures = 3 workgroups with 3.
The pictures of a workgroup are of DIFFERENT content kinds, so its waves walk the pooled Viterbi's barriers with
different data.  Every slot's full record (ctu_cost included) must equal the oracle's and the stream bytes the CPU path's."""
import numpy as np
import pytest

from content import content

pytestmark = pytest.mark.gpu

KEYS = ("cu_log2_size", "luma_mode", "chroma_mode", "lev_y", "lev_cb", "lev_cr", "rec_y", "rec_cb", "rec_cr", "ctu_cost")
KINDS = ["flat", "ramp", "stripes20", "stripes110", "checker", "noise", "cclm", "extremes", "stripes65"]


def _frame(i, w, h):
    from wrenc_amd import synth
    if i % 3 == 2:
        return synth.synth_textured_frame(w, h, i)
    if i % 11 == 10:
        return synth.synth_frame(w, h, i)
    return content(KINDS[i % len(KINDS)], w, h, 1000 + i)


@pytest.mark.parametrize("schedule", [1, 2])    # wave: groups of 4 pictures; team: one picture x 4 waves per workgroup
@pytest.mark.parametrize("n_pictures,w,h,qp,depth,first_slot", [(19, 96, 64, 32, 2, 0), (33, 64, 64, 27, 3, 2), (9, 64, 96, 37, 1, 1)])
def test_every_slot_of_a_multi_group_call_equals_the_oracle(built, n_pictures, w, h, qp, depth, first_slot, schedule):
    from wrenc_amd import bitstream as bs, gpu
    from oracle import pyoracle as po
    frames = [_frame(i, w, h) for i in range(n_pictures)]
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=first_slot + n_pictures, schedule=schedule)
    for s, f in enumerate(frames):
        enc.upload(first_slot + s, *f)
    enc.encode(first_slot, n_pictures)
    enc.sync()
    assert enc.final_pass_mismatches() == 0
    for s, f in enumerate(frames):
        got = enc.download(first_slot + s)
        ref = po.encode_picture(*f, qp, depth)
        for k in KEYS:
            assert np.array_equal(got[k], ref[k]), (s, k)
        assert bs.write_picture(w, h, qp, s, got) == bs.write_picture(w, h, qp, s, ref), s
    enc.close()


def test_two_calls_in_flight_on_disjoint_slots(built):
    """Two encode calls queued back to back (the double-buffered front ends do this): each keeps its own
    workgroups' scratch regions and its slots' results."""
    from wrenc_amd import gpu
    from oracle import pyoracle as po
    w, h, qp, depth = 64, 64, 32, 2
    frames = [_frame(i, w, h) for i in range(20)]
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=20)
    for s, f in enumerate(frames):
        enc.upload(s, *f)
    enc.encode(0, 11)
    enc.encode(11, 9)
    enc.sync()
    for s in (0, 7, 10, 11, 19):
        got = enc.download(s)
        ref = po.encode_picture(*frames[s], qp, depth)
        for k in KEYS:
            assert np.array_equal(got[k], ref[k]), (s, k)
    enc.close()


@pytest.mark.parametrize("n_pictures,w,h,qp,depth", [(9, 160, 96, 32, 2), (6, 128, 128, 27, 3)])
def test_auto_schedule_mixing_team_and_wave_diagonals(built, n_pictures, w, h, qp, depth):
    """AUTO decides per anti-diagonal: with the device's real wave-slot count a test-sized call is all
    TEAM, so the count is overridden (wrenc_gpu_test_set_wave_slots) until the thin first / last diagonals run as teams
    and the wide ones as waves inside ONE call -- the border records, the scratch-region bitmap and the picture-to-lane
    mapping all cross the switch.  Every slot's full record equals the oracle's; mixed content, padding waves."""
    from wrenc_amd import bitstream as bs, gpu
    from oracle import pyoracle as po
    frames = [_frame(i, w, h) for i in range(n_pictures)]
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=n_pictures, schedule=0)
    # diagonals hold 1 .. min(cols, rows-ish) CTUs: team while pictures x count x 100 <= slots x pct (pct: 65 at
    # max-split-depth 3, 50 below: wrenc_gpu.hip, kTeamBelowSlotsPct)
    pct = 65 if depth == 3 else 50
    enc.test_set_wave_slots((200 * n_pictures - 1) // pct)   # count = 1 -> team, count >= 2 -> wave
    for s, f in enumerate(frames):
        enc.upload(s, *f)
    enc.encode(0, n_pictures)
    enc.sync()
    assert enc.last_schedule() == 0                          # AUTO = both kernels ran in this call
    assert enc.final_pass_mismatches() == 0
    for s, f in enumerate(frames):
        got = enc.download(s)
        ref = po.encode_picture(*f, qp, depth)
        for k in KEYS:
            assert np.array_equal(got[k], ref[k]), (s, k)
        assert bs.write_picture(w, h, qp, s, got) == bs.write_picture(w, h, qp, s, ref), s
    enc.test_set_wave_slots(0)
    enc.encode(0, n_pictures)
    enc.sync()
    assert enc.last_schedule() == 2                          # the device's own figure: all teams at this size
    enc.close()


@pytest.mark.parametrize("schedule", [1, 2])
@pytest.mark.parametrize("w,h,qp,depth", [(96, 64, 32, 2), (64, 64, 27, 3), (64, 96, 37, 0)])
def test_slots_reused_by_pictures_with_other_zero_blocks(built, w, h, qp, depth, schedule):
    """The final pass writes the levels of the transform blocks that have any and zeroes only the blocks that held
    levels of the slot's previous picture (PicBufs::lev_dirty): rounds of pictures with very different sets of zero
    blocks (flat, noise, stripes, smooth, textured, rotated over the slots; compact read-back in between) must each
    read back, dense and compact, exactly as the oracle's record -- a level left over from an earlier round, or a
    block cleared that the round did code, shows as a difference."""
    from wrenc_amd import gpu, synth
    from oracle import pyoracle as po
    kinds = [lambda i: content("flat", w, h, 7 + i), lambda i: content("noise", w, h, 11 + i),
             lambda i: synth.synth_frame(w, h, i), lambda i: synth.synth_textured_frame(w, h, i),
             lambda i: content("stripes20", w, h, 3 + i), lambda i: content("flat", w, h, 90 + i)]
    n = 5
    enc = gpu.Encoder(w, h, qp=qp, max_split_depth=depth, n_slots=n, schedule=schedule)
    for rnd in range(len(kinds) + 1):
        frames = [kinds[(rnd + s) % len(kinds)](rnd) for s in range(n)]
        for s, f in enumerate(frames):
            enc.upload(s, *f)
        enc.encode(0, n)
        enc.sync()
        assert enc.final_pass_mismatches() == 0
        comp = enc.download_compact(0, n)
        for s, f in enumerate(frames):
            ref = po.encode_picture(*f, qp, depth)
            got = enc.download(s)
            for k in KEYS:
                assert np.array_equal(got[k], ref[k]), (rnd, s, k)
            ly, lcb, lcr = enc.expand_levels(comp[s][0], comp[s][1])
            assert np.array_equal(ly, ref["lev_y"]) and np.array_equal(lcb, ref["lev_cb"]) and np.array_equal(lcr, ref["lev_cr"]), (rnd, s)
    enc.close()
