"""The round protocol of the speculative tracker (r04), checked as a protocol: every pattern of failing milliseconds over small
blocks through the Python statement of it (tests/spec_rounds_model.py, which cites the device functions it mirrors).  What must
hold whatever fails where:
  * a channel that ends `ok` has, for every sub-block, a last tracking pass that was verified good and that started from the end
    state of the last pass over the sub-block before it -- no stale state anywhere in the chain;
  * a channel that is handed to the transform kernel (`bad_from = b`) has exactly that property for every sub-block before b, and
    the checkpoint of b is the end state of b - 1's last (good) pass (or the block's initial state for b = 0);
  * a verification failure costs a re-tracked sub-block, not the block: with one bad millisecond, every other sub-block is tracked
    once (plus at most one stale pass over the sub-block after the failing one)."""
import itertools

import spec_rounds_model as model


def _check(n_sub, sub_len, bad_ms, rounds=None):
    ok, bad_from, passes, ctl, ckpt, state_end = model.run_channel(n_sub, sub_len, set(bad_ms), rounds)
    upto = n_sub if ok else bad_from
    last = {}
    for p in passes:
        last[p.sub] = p                                   # (passes are in round order)
    prev = None
    for s in range(upto):
        p = last[s]
        assert p.good, (bad_ms, s, "last pass not verified good")
        assert p.parent is prev, (bad_ms, s, "started from a stale state")
        prev = p
    if not ok:
        assert bad_from in ckpt and ckpt[bad_from] is prev, (bad_ms, bad_from, "checkpoint of the hand-over is not the last good state")
    return ok, bad_from, passes, ctl


def test_every_pattern_of_up_to_three_bad_milliseconds_in_four_sub_blocks():
    n_sub, sub_len = 4, 3
    all_ms = range(n_sub * sub_len)
    n_ok = n_bad = 0
    for k in range(0, 4):
        for bad in itertools.combinations(all_ms, k):
            ok, bad_from, passes, ctl = _check(n_sub, sub_len, bad)
            n_ok += ok
            n_bad += not ok
            if k == 0:
                assert ok and len(passes) == n_sub and ctl.redos == 0
            if k == 1:
                s = bad[0] // sub_len
                assert ok and ctl.redos == 1                 # one re-tracked sub-block, never the transform kernel
                counts = [sum(p.sub == t for p in passes) for t in range(n_sub)]
                assert counts[s] == 2
                for t in range(n_sub):
                    if t != s:
                        assert counts[t] <= (2 if t == s + 1 else 1)     # only the sub-block tracked while the report was on its way is repeated
    assert n_ok > 0 and n_bad > 0                            # both outcomes occur (three failures do not always fit the rounds)


def test_eight_sub_blocks_with_failures_everywhere_it_can_hurt():
    n_sub, sub_len = 8, 2
    for first in range(n_sub):                               # a failure in every sub-block in turn, alone and with a second one later
        _check(n_sub, sub_len, [first * sub_len])
        for second in range(first, n_sub):
            ok, _, _, ctl = _check(n_sub, sub_len, [first * sub_len, second * sub_len + 1])
            assert ok and ctl.redos == 2
    # the last sub-block failing: only spec_finalize_kernel can notice -> handed over from that sub-block
    ok, bad_from, _, _ = _check(n_sub, sub_len, [(n_sub - 1) * sub_len], rounds=n_sub)          # no spare rounds at all
    assert not ok and bad_from == n_sub - 1
    ok, bad_from, _, _ = _check(n_sub, sub_len, [(n_sub - 1) * sub_len])                          # with the spare rounds it is re-done
    assert ok


def test_a_hopeless_channel_runs_out_of_slots_or_rounds_and_is_handed_over_cleanly():
    n_sub, sub_len = 12, 16
    every = list(range(0, n_sub * sub_len))                  # every millisecond fails unless forced
    ok, bad_from, passes, ctl = _check(n_sub, sub_len, every)
    assert not ok and bad_from == 0 and ctl.dead and len(ctl.force) == model.K_MAX_FORCE
    ok, bad_from, passes, ctl = _check(4, 16, every)         # short block: the rounds run out first
    assert not ok and bad_from == 0 and len(ctl.force) < model.K_MAX_FORCE
    # a channel that is bad only in its second half keeps its first half
    ok, bad_from, _, _ = _check(n_sub, sub_len, every[n_sub * sub_len // 2:])
    assert not ok and bad_from == n_sub // 2
