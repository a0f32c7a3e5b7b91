"""Shared helpers of the parity tests: bounded comparison of NMS keep lists / detection sets when a
consulted IoU sits within libm distance of the threshold (instead of skipping such cases)."""
import numpy as np

IOU_TOL = 2e-5          # device cosf / sinf / atan2f against glibc: IoU agrees to this (fp32, abs)


def assert_keep_lists_match(got, want, boxes, thr, tol=10 * IOU_TOL):
    """Keep lists (positions into the score-sorted `boxes`, pcdet frame) must be identical, except
    that a pair whose oracle IoU lies within `tol` of `thr` may fall on either side.  Every position
    in the symmetric difference must then be explained: it belongs to such a borderline pair, or it
    overlaps (IoU > thr - tol) an earlier explained position -- a flipped decision changes what that
    box suppresses further down the list.  Anything else is a real mismatch."""
    got, want = np.asarray(got, np.int64), np.asarray(want, np.int64)
    if np.array_equal(got, want):
        return 0
    from oracle import iou3d
    iou = iou3d.iou_matrix(boxes, boxes)
    n = boxes.shape[0]
    near = np.abs(iou - np.float32(thr)) < tol
    np.fill_diagonal(near, False)
    borderline = set(np.flatnonzero(near.any(1)).tolist())
    assert borderline, (f"keep lists differ ({np.setxor1d(got, want)[:10]}...) although no pair of the {n} boxes "
                        f"has an IoU within {tol:g} of {thr}")
    diff = sorted(np.setxor1d(got, want).tolist())
    explained = []
    for d in diff:
        ok = d in borderline or any(iou[e, d] > thr - tol for e in explained if e < d)
        assert ok, f"position {d} differs between the keep lists and no borderline pair explains it"
        explained.append(d)
    assert len(diff) <= 8 * len(borderline), (len(diff), len(borderline))
    return len(diff)


def assert_detections_match(got, want, min_gap, gap_tol=2e-4, max_flips=6, score_atol=2e-6, box_atol=1e-5):
    """Detections of one sample (dicts of numpy arrays: box3d_lidar, scores, label_preds) from the
    same head maps.  Exact (to the stated float tolerances) unless the oracle's NMS consulted an IoU
    within `gap_tol` of its threshold (`min_gap`): then at most `max_flips` detections may be present
    on one side only, and all others must still agree."""
    gb, wb = np.asarray(got["box3d_lidar"]), np.asarray(want["box3d_lidar"])
    gs, ws = np.asarray(got["scores"]), np.asarray(want["scores"])
    gl, wl = np.asarray(got["label_preds"]), np.asarray(want["label_preds"])
    if gb.shape == wb.shape and np.array_equal(gl, wl) and np.allclose(gs, ws, rtol=0, atol=score_atol) \
            and np.allclose(gb, wb, rtol=1e-5, atol=box_atol):
        return 0
    if min_gap >= gap_tol:
        what = f"detections differ ({gb.shape[0]} vs {wb.shape[0]})"
        if gb.shape == wb.shape:
            bad = np.flatnonzero((gl != wl) | (np.abs(gs - ws) > score_atol) |
                                 (np.abs(gb - wb) > box_atol + 1e-5 * np.abs(wb)).any(1))
            i = int(bad[0])
            what += (f": {bad.size} rows, first at {i}: labels {gl[i]} / {wl[i]}, scores {gs[i]!r} / {ws[i]!r}, "
                     f"boxes {gb[i]} / {wb[i]}; max |dbox| {np.abs(gb - wb).max():g}, max |dscore| {np.abs(gs - ws).max():g}")
        raise AssertionError(what + f"; the oracle's closest IoU is {min_gap:g} from its threshold: not a borderline case")
    # match on (label, centre): every oracle detection but a few must have an identical partner
    unmatched_w, used = 0, np.zeros(gb.shape[0], bool)
    for i in range(wb.shape[0]):
        ok = (gl == wl[i]) & (np.abs(gb[:, :2] - wb[i, :2]).max(1) <= box_atol) & (np.abs(gs - ws[i]) <= score_atol) & ~used
        j = np.flatnonzero(ok)
        if j.size:
            used[j[0]] = True
            np.testing.assert_allclose(gb[j[0]], wb[i], rtol=1e-5, atol=box_atol)
        else:
            unmatched_w += 1
    flips = unmatched_w + int((~used).sum())
    assert flips <= max_flips, f"{flips} detections differ; a borderline IoU explains at most {max_flips}"
    return flips


def partner_rate(got, want, centre_tol=5e-2, score_tol=2e-2):
    """Fraction of `want`'s detections (dict of numpy arrays: box3d_lidar, scores, label_preds) that have a partner in
    `got`: same label, centre (x, y) within `centre_tol` metres, score within `score_tol`; each row of `got` serves once.
    -> (rate, matched, total)."""
    gb, gs, gl = np.asarray(got["box3d_lidar"]), np.asarray(got["scores"]), np.asarray(got["label_preds"])
    wb, ws, wl = np.asarray(want["box3d_lidar"]), np.asarray(want["scores"]), np.asarray(want["label_preds"])
    used, hit = np.zeros(gb.shape[0], bool), 0
    for i in np.argsort(-ws):
        ok = (gl == wl[i]) & (np.abs(gb[:, :2] - wb[i, :2]).max(1) <= centre_tol) & (np.abs(gs - ws[i]) <= score_tol) & ~used
        j = np.flatnonzero(ok)
        if j.size:
            j = j[np.argmin(np.abs(gb[j, :2] - wb[i, :2]).max(1))]
            used[j] = True
            hit += 1
    n = wb.shape[0]
    return (hit / n if n else 1.0), hit, n
