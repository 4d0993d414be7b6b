"""Row a7 against GROUND TRUTH (no second implementation involved): sparse pyramidal LK on image pairs whose motion is known exactly.

* sub-pixel TRANSLATIONS without any resampling error: a 4x finer texture is box-averaged to the 480 x 270 tracking frame at two integer
  fine-grid offsets -- the coarse frames differ by an exact multiple of 0.25 px;
* small AFFINE motions (rotation + zoom about the frame centre) rendered by SURVEY 8d's clip generator, ground truth = its homography.
Points: FAST corners of the first frame (all four pyramid levels take part), plus the border band.  Bar: median error below
0.05 px.  The CPU suite holds the specification (oracle) to it, the GPU suite the HIP kernel (which equals the oracle bit for bit)."""
import numpy as np
import pytest

from tests import clipgen, synth


def _box4(fine, oy, ox, rows=270, cols=480):
    v = fine[oy:oy + 4 * rows, ox:ox + 4 * cols].astype(np.float64)
    return np.clip(np.rint(v.reshape(rows, 4, cols, 4).mean(axis=(1, 3))), 0, 255).astype(np.uint8)


def _translation_cases():
    fine = synth.textured_frame(4 * 270 + 64, 4 * 480 + 64, seed=99, channels=1)
    # blur a little so that the coarse frame is not aliased (box filter of the fine texture is the only low-pass otherwise)
    for (qx, qy) in [(1, 0), (2, 3), (5, -6), (-9, 7), (13, 10)]:                  # quarter pixels: 0.25 .. 3.25 px
        prev = _box4(fine, 32, 32)
        nxt = _box4(fine, 32 - qy, 32 - qx)                                        # content moves by (+qx / 4, +qy / 4)
        yield f"translation ({qx / 4:+.2f}, {qy / 4:+.2f})", prev, nxt, (lambda p, dx=qx / 4.0, dy=qy / 4.0: p + np.array([dx, dy]))


def _affine_cases():
    clip = clipgen.Clip(540, 960, 40, jitter=1.0)
    for i in (3, 11, 22, 31):
        a = clip.render444(i - 1).numpy()[..., 0]; b = clip.render444(i).numpy()[..., 0]
        prev = a.reshape(270, 2, 480, 2).astype(np.float64).mean(axis=(1, 3)).round().astype(np.uint8)
        nxt = b.reshape(270, 2, 480, 2).astype(np.float64).mean(axis=(1, 3)).round().astype(np.uint8)
        S = np.array([[2, 0, 0.5], [0, 2, 0.5], [0, 0, 1.0]])
        H = np.linalg.inv(S) @ clip.motion(i) @ S

        def truth(p, H=H):
            q = np.c_[p, np.ones(len(p))] @ H.T
            return q[:, :2] / q[:, 2:]
        yield f"clip frame {i}", prev, nxt, truth


def _check(track, oracle):
    print()
    worst = 0.0
    for name, prev, nxt, truth in list(_translation_cases()) + list(_affine_cases()):
        kp = oracle.fast(prev, 20)
        pts = kp[:, :2].astype(np.float32)
        assert len(pts) > 150, name
        out, st = track(prev, nxt, pts)
        ok = st == 1
        assert ok.mean() > 0.9, (name, ok.mean())
        err = np.linalg.norm(out[ok] - truth(pts[ok].astype(np.float64)), axis=1)
        border = ((pts[ok, 0] < 12) | (pts[ok, 0] > 467) | (pts[ok, 1] < 12) | (pts[ok, 1] > 257))
        print(f"  {name:28s} {len(pts):5d} corners  tracked {100 * ok.mean():5.1f} %   median {np.median(err):.4f}  p90 {np.percentile(err, 90):.4f} px"
              + (f"   border band ({border.sum()}): median {np.median(err[border]):.4f}" if border.sum() >= 5 else ""))
        assert np.median(err) < 0.05, (name, float(np.median(err)))
        worst = max(worst, float(np.median(err)))
    return worst


def test_specification_tracks_known_motion(oracle):
    _check(lambda a, b, p: oracle.pyrlk(a, b, p), oracle)


@pytest.mark.gpu
def test_hip_kernel_tracks_known_motion(ctx, oracle):
    import torch
    _check(lambda a, b, p: ctx.pyrlk(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda(), p), oracle)
