"""Host-side schedules of the fused launches (cirkit_amd/fusion.py): CPU tests."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
def test_backward_walk_segments_cover_every_tile_once_and_use_every_workgroup():
    """`balanced_segments` (the schedule of `ck_leaf_walk_bwd`): whatever form it picks -- whole roots cut into equal pieces
    dealt longest first, or the flat (root, wave round) list cut into equal stretches -- every (root, tile) is in exactly one
    segment, segments never straddle roots, workgroup g takes rows g, g + num_wg, ..., and at the north-star shape (196 nodes
    x 128 tiles, 8 waves) all 256 workgroups have work (whole roots would leave 60 idle)."""
    import numpy as np

    from cirkit_amd.fusion import balanced_segments

    for roots, tiles, wg, waves in [(196, 128, 256, 8), (49, 128, 256, 8), (196, 128, 256, 4), (49, 128, 256, 4), (3, 5, 256, 8),
                                    (1, 128, 256, 8), (392, 32, 256, 8), (7, 1000, 64, 8)]:
        seg = balanced_segments(roots, tiles, wg, waves=waves)
        assert seg.dtype == np.int32 and seg.shape[1] == 4
        cover = np.zeros((roots, tiles), dtype=np.int64)
        for r, t0, t1, _ in seg:
            assert 0 <= r < roots and 0 <= t0 <= t1 <= tiles
            cover[r, t0:t1] += 1
        assert (cover == 1).all(), (roots, tiles, wg, waves)
    seg = balanced_segments(196, 128, 256, waves=8)
    busy = {j % 256 for j, (_, t0, t1, _) in enumerate(seg) if t1 > t0}
    assert len(busy) == 256
    rounds = np.zeros(256)
    for j, (_, t0, t1, _) in enumerate(seg):
        rounds[j % 256] += -(-(t1 - t0) // 8)
    assert rounds.max() <= 13  # (12.25 on average)
