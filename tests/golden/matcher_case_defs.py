"""Case table shared by make_matcher_golden.py (runs the reference) and tests/test_matcher.py (runs ours)."""
CASES = {
    # name: (ctor kwargs, B, Q, num_boxes, repeats, repeat_batch, use_out_valid, use_tgt_valid)
    "cli_focal": (dict(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal=True), 3, 20, [2, 0, 5], 1, 1, False, False),
    "cli_focal_big": (dict(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal=True), 4, 200, [7, 1, 3, 12], 1, 1, False, False),
    "more_targets_than_queries": (dict(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal=True), 2, 4, [6, 3], 1, 1, False, False),
    "masked": (dict(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal=True), 3, 12, [4, 2, 3], 1, 1, True, True),
    "one_to_many": (dict(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal=True), 2, 30, [3, 4], 3, 1, False, False),
    "aux_batched": (dict(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal=True), 2, 10, [2, 3], 1, 3, False, False),
    "plain": (dict(cost_class=1.0, cost_bbox=1.0, cost_giou=1.0, focal=False), 2, 10, [3, 2], 1, 1, False, False),
    "stable": (dict(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal=True, stable=True), 2, 10, [3, 2], 1, 1, False, False),
    "keep_empty": (dict(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal=True, remove_samples_with_0_gt=False), 3, 8, [2, 0, 1], 1, 1, False, False),
    "all_empty": (dict(cost_class=2.0, cost_bbox=5.0, cost_giou=2.0, focal=True), 2, 8, [0, 0], 1, 1, False, False),
}


