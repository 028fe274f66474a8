"""Python statement of the speculative tracker's round protocol -- `spec_ctl_begin` / `track_verify_kernel`'s report /
`spec_finalize_kernel` in gypsum_amd/csrc/kernels_track_block.hpp and kernels_dll_exact.hpp, `track_block_speculative` in
gypsum_hip.hip -- with the tracking and the verification replaced by bookkeeping: a channel is a set of "bad" milliseconds (those
whose window does not hold the profile's arg-max); tracking a sub-block with a millisecond on the forced-transform list makes
that millisecond good.  tests/test_spec_rounds_model.py runs it over every small failure pattern and checks what the protocol
promises; nothing here touches the GPU or the oracle."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Optional, Set, Tuple

K_MAX_FORCE = 8          # kMaxForce
NO_FAIL = None           # kNoFail


@dataclass
class Ctl:               # SpecCtl
    cursor: int = 0
    rb_round: int = 0    # (memset 0; only ever compared with R - 1 for R >= 2)
    dead: bool = False
    redos: int = 0
    force: List[int] = field(default_factory=list)


@dataclass
class Pass:              # one tracking pass over one sub-block
    sub: int
    round: int
    forced: Tuple[int, ...]
    parent: Optional["Pass"]          # the pass over sub - 1 whose end state this one started from (None: the block's initial state)
    from_checkpoint: bool
    good: Optional[bool] = None       # set by the verification of its round


def rounds_for(n_sub: int) -> int:
    return n_sub + 2 + (6 if n_sub >= 8 else 2)


def run_channel(n_sub: int, sub_len: int, bad_ms: Set[int], rounds: Optional[int] = None):
    """Returns (ok, bad_from, passes, ctl).  bad_from is None when ok."""
    T = rounds_for(n_sub) if rounds is None else rounds
    c = Ctl()
    trk: List[int] = [-1] * T
    fail: List[Optional[int]] = [NO_FAIL] * T
    passes: List[Pass] = []
    state_end: Dict[int, Pass] = {}           # end state of the most recent pass over each sub-block = what `states` / the next checkpoint hold
    ckpt: Dict[int, Optional[Pass]] = {}      # ckpt[s] = the pass whose end state was saved when sub-block s was started (None: initial state)
    current: Optional[Pass] = None            # the pass whose end state `states[ch]` holds right now
    for R in range(T):
        # ---- spec_ctl_begin
        restore = False
        if not c.dead and R >= 2 and c.rb_round != R - 1:
            s = trk[R - 2]
            x = fail[R - 2] if s >= 0 else NO_FAIL
            if x is not NO_FAIL:
                if len(c.force) < K_MAX_FORCE:
                    c.force.append(x); c.cursor = s; c.rb_round = R; c.redos += 1; restore = True
                else:
                    c.dead = True; c.cursor = s
        sub = c.cursor if (not c.dead and c.cursor < n_sub) else -1
        trk[R] = sub
        if sub < 0:
            continue
        c.cursor = sub + 1
        # ---- the tracking kernel's prologue: checkpoint taken, or gone back to
        if restore:
            parent = ckpt[sub]
        else:
            ckpt[sub] = current
            parent = current
        p = Pass(sub, R, tuple(c.force), parent, restore)
        passes.append(p)
        current = p
        state_end[sub] = p
        # ---- track_verify_kernel of round R (its report is read in round R + 2)
        lo, hi = sub * sub_len, (sub + 1) * sub_len
        failing = sorted(m for m in bad_ms if lo <= m < hi and m not in p.forced)
        fail[R] = failing[0] if failing else NO_FAIL
        p.good = not failing
    # ---- spec_finalize_kernel
    for R in (T, T + 1):
        if c.dead or R < 2 or c.rb_round == R - 1:
            continue
        s = trk[R - 2]
        if s >= 0 and fail[R - 2] is not NO_FAIL:
            c.dead = True; c.cursor = s
    ok = (not c.dead) and c.cursor >= n_sub
    bad_from = None if ok else c.cursor
    if not ok and not c.dead:
        ckpt[c.cursor] = current          # ran out of rounds in front of a sub-block it never started: its present state IS that checkpoint
    return ok, bad_from, passes, c, ckpt, state_end
