"""Stated fp32 tolerances of the parity tests, in one place (profiles/r02_lg_tolerance.md).

SuperPoint has none: keypoints, scores and descriptors are bit-exact against the oracle.

LightGlue match scores are probabilities exp(log-assignment); the log-assignment is a difference of O(30-100)
similarity logits and their log-sum-exps, so a few fp32 ulps there (6e-6 each at 50) are 1e-4-level in the score.
Measured at K = 1024 over 40 (weight seed, input) cases, max |score difference| over common matches:
  oracle (fp32) vs float64 evaluation of the same graph   1.6e-4
  HIP path vs oracle                                       2.1e-4 (Wo folded into ffn.0, the default) / 1.7e-4 (unfolded)
  HF transformers (fp32, torch CPU) vs float64             1.2e-4
i.e. any two fp32 evaluation orders of this graph differ by 1-2e-4; 1e-4 is below the noise floor of fp32 itself at
this size, and the fold does not change the picture.  Final token states agree to 1e-5 (bar 1e-4).
"""
LG_SCORE_TOL = 5e-4         # |match score difference|, any keypoint count up to 1024
LG_SCORE_TOL_SMALL = 1e-4   # <= 256 keypoints per side
LG_STATE_TOL = 1e-4         # final token states x0 / x1
LG_LOGSCORE_TOL = 2e-3      # log-domain assignment matrix, |values| up to ~1e2


def lists_agree(pairs_a, ms_a, pairs_b, ms_b, filter_thr=0.1, slack=5e-4):
    """Two match lists are the same up to matches whose score sits within `slack` of the 0.1 filter (either side may
    drop them).  Returns (ok, max |score difference| over the common matches)."""
    da = {(int(i), int(j)): float(s) for (i, j), s in zip(pairs_a, ms_a)}
    db = {(int(i), int(j)): float(s) for (i, j), s in zip(pairs_b, ms_b)}
    only = [(k, da.get(k, db.get(k))) for k in da.keys() ^ db.keys()]
    ok = all(abs(s - filter_thr) <= slack for _, s in only)
    dev = max((abs(da[k] - db[k]) for k in da.keys() & db.keys()), default=0.0)
    return ok, dev
