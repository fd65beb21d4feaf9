"""The parity RULES: how a result of the HIP path is held against the oracle's — top-k lists, ranked lists of
arbitrary items, and lists of beam hypotheses under a noisy arithmetic (the bf16 precision mode).

Test infrastructure like everything under oracle/: imported by tests/ (through tests/conftest.py), by bench.py's parity /
cpu_baseline legs and by __graft_entry__.smoke() only — never by the product (gdr_amd/).  Every rule raises AssertionError
on a violation.
"""
import numpy as np


def order_insensitive_topk_match(ref_vals, ref_idx, got_vals, got_idx, tol):
    """Top-k parity rule: values within tol; ids exact wherever neighbouring reference scores are
    more than 2*tol apart, and the same id *set* inside a tolerance-tie group.  Returns number of permuted slots."""
    ref_vals = np.asarray(ref_vals, dtype=np.float64)
    got_vals = np.asarray(got_vals, dtype=np.float64)
    assert ref_vals.shape == got_vals.shape
    np.testing.assert_allclose(got_vals, ref_vals, rtol=tol, atol=tol)
    permuted = 0
    for r in range(ref_vals.shape[0]):
        if np.array_equal(ref_idx[r], got_idx[r]):
            continue
        k = ref_vals.shape[1]
        j = 0
        while j < k:
            e = j
            while e + 1 < k and abs(ref_vals[r, e] - ref_vals[r, e + 1]) <= 2 * tol * (1 + abs(ref_vals[r, e])):
                e += 1
            a, b = ref_idx[r, j:e + 1], got_idx[r, j:e + 1]
            if e == k - 1:
                # last group may be cut by k: ids in `got` must score within tol of the boundary — checked by values
                common = len(set(a.tolist()) & set(b.tolist()))
                permuted += (e - j + 1) - common
            else:
                assert set(a.tolist()) == set(b.tolist()), (r, j, e, a, b)
                permuted += int((a != b).sum())
            j = e + 1
    return permuted


def ranked_lists_match(ref_items, ref_scores, got_items, tol):
    """The same rule for ranked lists of arbitrary hashable items (cluster strings, doc-id strings): positions agree
    exactly wherever neighbouring reference scores are more than 2*tol apart; inside a tolerance-tie group the same
    item set in any order; the last group may be cut by the list length.  Returns the number of permuted slots."""
    assert len(ref_items) == len(got_items), (len(ref_items), len(got_items))
    if list(ref_items) == list(got_items):
        return 0
    sc = np.asarray(ref_scores, dtype=np.float64)
    k, j, permuted = len(ref_items), 0, 0
    while j < k:
        e = j
        while e + 1 < k and abs(sc[e] - sc[e + 1]) <= 2 * tol * (1 + abs(sc[e])):
            e += 1
        a, b = list(ref_items[j:e + 1]), list(got_items[j:e + 1])
        if e == k - 1:
            permuted += (e - j + 1) - len(set(a) & set(b))
        else:
            assert sorted(a) == sorted(b), (j, e, a, b)
            permuted += sum(1 for x, y in zip(a, b) if x != y)
        j = e + 1
    return permuted


def beam_cut_explains_absence(trace, prefix_trace, q, R, Vd, row, tol, lp=0.8, eos=1, final_cut=None):
    """Beam search prunes at EVERY step, so a hypothesis can be missing from the reference's final list although its final
    score would have ranked well: its prefix fell at an intermediate cut.  That is legitimate under a noisy arithmetic only if it
    fell by a TIE.  Replays the reference's selection (generation_utils.py:800-829) from the oracle's per-step trace
    (beam_ref.beam_search(trace=, prefix_trace=)) for the token row `row` (START, tokens..., [EOS, PAD...]) of query q and
    returns a short description of the cut it tied with — or None when it fell by more than `tol` (absolute, on the final-score
    scale; per-step scores are sums of log-probabilities, i.e. final score x cur_len^lp) or cannot be found among the 2R
    ranked candidates of the step at all."""
    toks = [int(t) for t in row]
    if eos in toks[1:]:
        toks = toks[:1 + toks[1:].index(eos)]
    n = len(toks) - 1                                           # tokens after START
    for s in range(min(n + 1, len(trace))):
        sc, tk = trace[s][0][q].tolist(), trace[s][1][q].tolist()
        pref = prefix_trace[s][q * R:(q + 1) * R].tolist()
        if toks[:s + 1] not in pref:
            return None                                         # the prefix is not a beam although no cut explained it
        j = pref.index(toks[:s + 1])
        want = j * Vd + (toks[s + 1] if s < n else eos)
        if want not in tk:
            return None
        r = tk.index(want)
        tol_s = tol * float(s + 1) ** lp                        # the step's scores are not length-normalised
        if s == n:                                              # the EOS candidate: counted only at rank < R (:811-813)
            if r >= R:
                return f"EOS at rank {r} of step {s} ties the rank-{R - 1} candidate" if abs(sc[r] - sc[R - 1]) <= 2 * tol_s else None
            if final_cut is not None and abs(sc[r] / float(s + 1) ** lp - final_cut) <= 2 * tol:
                return f"finished with a score that ties the final cut"
            return None
        non_eos = [i for i, t in enumerate(tk) if t % Vd != eos]
        p = non_eos.index(r)
        if p >= R:                                              # not among the R continued beams: it fell here
            cut = sc[non_eos[R - 1]]
            return f"pruned at step {s} (rank {p} of the non-EOS candidates) in a tie with the cut" if abs(sc[r] - cut) <= 2 * tol_s else None
    # survived every recorded step: an open beam at max_length — it is in the final list unless it ties the final cut
    if final_cut is not None:
        last = len(trace) - 1
        return "open beam tying the final cut" if abs(sc[r] / float(last + 2) ** lp - final_cut) <= 2 * tol else None
    return None


def hypothesis_lists_match(ref_items, ref_scores, got_items, tol, explain_foreign=None):
    """The rule for lists of beam HYPOTHESES under a noisy arithmetic (the bf16 precision mode), with teeth.
    `tol` is ABSOLUTE and must come from the measured score gap between the two implementations (the caller asserts that
    gap first): two hypotheses whose reference scores differ by more than 2*tol cannot legitimately swap.  Neighbours of the
    reference list closer than 2*tol chain into one tie group.  For every position p of `got`:
      * the item is the reference's item at p: fine;
      * the item sits at another reference position r: p and r must lie in the SAME tie group (asserted in every group, the
        last one included — it is only 'cut' for items that left the list, next case);
      * the item is not in the reference list at all: either its slot p lies in the LAST tie group (it crossed the final cut at k
        in a tie), or — beam search prunes at every step — `explain_foreign(item)` (beam_cut_explains_absence on the oracle's
        per-step trace) names the intermediate cut it fell at in the reference, by a tie.  Anything else fails.
    Returns (moved, foreign, group_sizes) so that the caller can print that the rule is not one big group."""
    k = len(ref_items)
    assert len(got_items) == k, (len(got_items), k)
    sc = np.asarray(ref_scores, dtype=np.float64)
    group, sizes = np.zeros(k, np.int64), [1]
    for i in range(1, k):
        if abs(sc[i - 1] - sc[i]) <= 2 * tol:
            group[i] = group[i - 1]
            sizes[-1] += 1
        else:
            group[i] = group[i - 1] + 1
            sizes.append(1)
    pos = {x: i for i, x in enumerate(ref_items)}
    assert len(pos) == k, "reference hypotheses are distinct"
    moved = foreign = 0
    for p, x in enumerate(got_items):
        r = pos.get(x)
        if r is None:
            why = None if group[p] == group[k - 1] or explain_foreign is None else explain_foreign(x)
            assert group[p] == group[k - 1] or why, ("a hypothesis outside the reference list: its slot does not tie with the final "
                                                     "cut and no intermediate cut of the reference's search explains it by a tie",
                                                     p, x, sc[p], sc[k - 1], tol)
            foreign += 1
        elif r != p:
            assert group[p] == group[r], ("two hypotheses swapped outside a tolerance-tie group", p, r, sc[p], sc[r], tol)
            moved += 1
    return moved, foreign, sizes
