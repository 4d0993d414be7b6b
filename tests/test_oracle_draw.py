"""CPU tests pinning the oracle's debug overlays (Drawing.cl:22-39,75-105) by hand-computable cases."""
import numpy as np


def test_grid_known_pattern(oracle):
    img = np.zeros((12, 16, 3), np.uint8)
    out = oracle.draw_grid(img, (2, 2), (9, 8, 7), thickness=1)          # cells 8 x 6
    on = (out == (9, 8, 7)).all(axis=2)
    # fmod(x, 8) < 1 -> x in {0, 8}; fmod(x, 8) > 8 - 1 - 1 = 6 -> x in {7, 15}; rows likewise with 6: {0, 6} and {5, 11}
    cols_on = {0, 7, 8, 15}; rows_on = {0, 5, 6, 11}
    for y in range(12):
        for x in range(16):
            assert on[y, x] == (x in cols_on or y in rows_on), (x, y)
    assert (out[~on] == 0).all()


def test_grid_single_cell_draws_the_frame_border_only(oracle):
    img = np.full((10, 14, 3), 50, np.uint8)
    out = oracle.draw_grid(img, (1, 1), (1, 2, 3), thickness=1)          # 2x2 motion mesh -> 1x1 grid
    on = (out == (1, 2, 3)).all(axis=2)
    want = np.zeros((10, 14), bool); want[0] = want[-1] = True; want[:, 0] = want[:, -1] = True
    assert np.array_equal(on, want)


def test_cross_known_pattern_and_clipping(oracle):
    img = np.zeros((40, 40, 3), np.uint8)
    out = oracle.draw_crosses(img, [(20, 20)], (255, 1, 2), cross_size=7, thickness=1)       # kernel size = (7 + 1) / 2 = 4
    on = (out == (255, 1, 2)).all(axis=2)
    want = np.zeros((40, 40), bool)
    for k in range(9):                                                   # x, y from 16 to 24; the back diagonal starts at max_x - 1 = 24
        want[16 + k, 16 + k] = True; want[16 + k, 24 - k] = True
    assert np.array_equal(on, want)
    # thickness widens every stroke to the right; points outside the frame draw nothing; rounding is half-to-even after scaling
    out = oracle.draw_crosses(img, [(-30, 5), (100, 100), (2.5, 3.5)], (5, 5, 5), cross_size=3, thickness=2, scaling=(1.0, 1.0))
    on = (out == 5).all(axis=2)
    ys, xs = np.nonzero(on)
    assert ys.min() == 2 and ys.max() == 6                               # centre (2, 4): y = 4 +- 2, x clipped at 0
    assert xs.min() == 0 and xs.max() <= 2 + 2 + 1 + 1
    assert not on[:, 10:].any()


def test_stab_overlays_modify_the_queued_frame(oracle):
    from tests import oracle_lib, synth
    frames, _ = synth.make_clip(180, 320, 6, seed=2)
    s = oracle_lib.preset("field", predictive_samples=2)
    st = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); st.configure(s)
    plain = oracle_lib.OracleStabilizer(oracle, oracle_lib.preset("default")); plain.configure(s)
    outs, refs = [], []
    for i, f in enumerate(frames):
        o, _ = st.push(f, ts=i); st.draw_motion_mesh(); st.draw_trackers()
        r, _ = plain.push(f, ts=i)
        if o is not None:
            outs.append(o); refs.append(r)
    st.close(); plain.close()
    assert len(outs) == 4
    # the first emitted frame was drawn on when it entered the queue: it differs from the plain output, mostly by blue grid pixels
    d = (outs[0] != refs[0]).any(axis=2)
    assert d.mean() > 0.02
