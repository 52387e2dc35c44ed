"""Decision margins of a whole-step fixture (test infrastructure; data and plain arithmetic only).

A training step of the reference takes discrete decisions on its model outputs: one Hungarian assignment per decoder output
(final + auxiliary) and per auxiliary one-to-many twin (``sam3/train/matcher.py:481-668``, called with
``use_o2m_matcher_on_o2m_aux=False``: ``sam3/train/loss/sam3_loss.py:112-125``), and one threshold decision for the final
output's one-to-many twin (``BinaryOneToManyMatcher``, ``matcher.py:671-810``: ``alpha p + (1 - alpha) IoU`` in the per-target
top-k AND above the threshold).  A fixture whose decisions sit on near-ties pins nothing about a mixed-precision build: rounding
noise decides them.  The model's forward does not depend on the ground-truth boxes, so the generator
(``make_e2e_golden.py <case> --search-boxes``) chooses the boxes such that every decision is taken with a margin; this module
is the arithmetic of "margin", shared by the generator (on the reference's own cost matrices) and by the CPU test that holds
the committed fixtures to it (``tests/test_sam3_e2e.py::test_fixture_decisions_have_margins``).
"""
import numpy as np
from scipy.optimize import linear_sum_assignment

HUNGARIAN_MARGIN = 0.3      # every optimum beats the best assignment that differs from it in at least one pair by this much
O2M_MARGIN = 0.05           # no one-to-many score within this distance of the decision it hangs on
FORBID = 1e6


def lsap_gap(cost):
    """cost [Q, T] (T <= Q) -> (optimal total, second-best total - optimal total).  The second-best assignment differs from the
    optimum in at least one pair, so it is the best of the T problems that each forbid one of the optimum's pairs (the first
    level of Murty's partition)."""
    cost = np.asarray(cost, np.float64)
    if cost.shape[1] == 0:
        return 0.0, float("inf")
    r, c = linear_sum_assignment(cost)
    best = float(cost[r, c].sum())
    second = float("inf")
    for i, j in zip(r, c):
        alt = cost.copy()
        alt[i, j] = FORBID
        r2, c2 = linear_sum_assignment(alt)
        second = min(second, float(alt[r2, c2].sum()))
    return best, second - best


def o2m_margin(score, num_boxes, threshold, topk):
    """score [B, Q, T] = alpha p + (1 - alpha) IoU (higher is better), num_boxes [B] -> (margin, number of positive pairs).
    A pair is positive iff it is among its target's top-k scores AND above the threshold.  margin = the smaller of: the distance
    of any valid pair's score to the threshold; per target, the distance between the k-th and the (k+1)-th score when the
    (k+1)-th is not safely below the threshold (otherwise the order of those two decides nothing)."""
    score = np.asarray(score, np.float64)
    B, Q, T = score.shape
    margin, positives = float("inf"), 0
    for b in range(B):
        for t in range(int(num_boxes[b])):
            s = np.sort(score[b, :, t])[::-1]
            margin = min(margin, float(np.abs(s - threshold).min()))
            if Q > topk and s[topk] > threshold - O2M_MARGIN:
                margin = min(margin, float(s[topk - 1] - s[topk]))
            positives += int((s[:topk] > threshold).sum())
    return margin, positives


def candidate_boxes(rng, n, predicted=None):
    """n ground-truth boxes (cx, cy, w, h), normalised, inside the image, three decimals.  Half of the draws are uniform, half
    are a predicted box of the model (``predicted`` [K, 4]) with a little jitter -- a target that one query already covers is
    what separates that query's cost from everybody else's."""
    out = []
    for _ in range(n):
        if predicted is not None and len(predicted) and rng.random() < 0.5:
            cx, cy, w, h = (float(v) for v in predicted[rng.integers(len(predicted))] + rng.normal(0, 0.01, 4))
        else:
            w, h = rng.uniform(0.10, 0.45, 2)
            cx, cy = rng.uniform(0.25, 0.75, 2)
        w, h = min(max(w, 0.06), 0.9), min(max(h, 0.06), 0.9)
        cx = min(max(cx, w / 2 + 0.01), 1 - w / 2 - 0.01)
        cy = min(max(cy, h / 2 + 0.01), 1 - h / 2 - 0.01)
        out.append(tuple(round(float(v), 3) for v in (cx, cy, w, h)))
    return out
