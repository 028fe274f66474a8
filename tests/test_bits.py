"""Navigation-bit integrator: the oracle restatement and the native host integrator
(gyp_bits_*) against the reference's own outputs in tests/golden/bits.npz."""
from __future__ import annotations

import numpy as np
import pytest

import golden_util
from oracle import gypsum_oracle as oracle

from gypsum_amd import _lib
from gypsum_amd.navigation_bit_intergrator import (EmitNavigationBitEvent, NavigationBitIntegrator,
                                                    NavigationBitIntegratorBank)
from gypsum_amd.tracker import BitValue, EmittedPseudosymbol, NavigationBitPseudosymbol

GOLD = golden_util.load("bits.npz")
NAMES = sorted({k.split("__")[0] for k in GOLD.files})
STATE_FIELDS = ["determined_bit_phase", "previous_bit_phase_decision", "failed_bit_count", "emitted_bit_count",
                "processed_pseudosymbol_count", "sequential_unknown_bit_value_counter",
                "pseudosymbol_cursor_within_queue", "slide", "queued_pseudosymbols"]


def gold(name: str, key: str) -> np.ndarray:
    return GOLD[f"{name}__{key}"]


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference(name: str) -> None:
    integ = oracle.BitIntegrator()
    events, cursors = [], []
    for v, st, en in zip(gold(name, "symbols"), gold(name, "start"), gold(name, "end")):
        at, ev = integ.process(float(st), float(st), float(en), int(v))
        events += [(a, b, c, len(cursors)) for a, b, c in ev]
        cursors.append(at)
    assert np.array_equal(np.array(events, dtype=np.float64).reshape(-1, 4), gold(name, "events"))
    assert np.array_equal(np.array(cursors), gold(name, "cursors"))
    none = lambda v: -1 if v is None else v
    state = [none(integ.determined_bit_phase), none(integ.previous_bit_phase_decision), integ.failed_bit_count,
             integ.emitted_bit_count, integ.processed, integ.sequential_unknown, integ.cursor, integ.slide, len(integ.queued)]
    assert state == list(gold(name, "state"))
    assert list(integ.last_bits) == list(gold(name, "last_bits"))


@pytest.mark.parametrize("name", NAMES)
def test_native_per_symbol_matches_reference(name: str) -> None:
    """The drop-in class, one pseudosymbol per call like pipeline.py:79."""
    integ = NavigationBitIntegrator(satellite_id=1)
    events, cursors = [], []
    for v, st, en in zip(gold(name, "symbols"), gold(name, "start"), gold(name, "end")):
        ps = EmittedPseudosymbol(float(st), float(en), NavigationBitPseudosymbol.from_val(int(v)), 0)
        for e in integ.process_pseudosymbol(float(st), ps):
            assert isinstance(e, EmitNavigationBitEvent)
            code = {BitValue.ZERO: 0, BitValue.ONE: 1, BitValue.UNKNOWN: 2}[e.bit_value]
            events.append((e.receiver_timestamp, e.trailing_edge_receiver_timestamp, code, len(cursors)))
        cursors.append(ps.cursor_at_emit_time)
    assert np.array_equal(np.array(events, dtype=np.float64).reshape(-1, 4), gold(name, "events"))
    assert np.array_equal(np.array(cursors), gold(name, "cursors"))
    h = integ.history
    none = lambda v: -1 if v is None else v
    state = [none(h.determined_bit_phase), none(h.previous_bit_phase_decision), h.failed_bit_count, h.emitted_bit_count,
             h.processed_pseudosymbol_count, h.sequential_unknown_bit_value_counter, h.pseudosymbol_cursor_within_queue,
             integ.slide, int(integ._native.state(0)["queued_pseudosymbols"])]
    assert state == list(gold(name, "state"))
    code = {BitValue.ZERO: 0, BitValue.ONE: 1, BitValue.UNKNOWN: 2}
    assert [code[b] for b in h.last_emitted_bits] == list(gold(name, "last_bits"))
    assert len(h.last_seen_pseudosymbols) == min(1000, len(cursors))


def test_native_block_path_matches_reference() -> None:
    """All scenarios of equal start time as channels of one gyp_track_rec block, fed in uneven block sizes."""
    names = [n for n in NAMES if gold(n, "start")[0] == 0.0]
    n_ms = min(len(gold(n, "symbols")) for n in names if len(gold(n, "symbols")) >= 1500)
    names = [n for n in names if len(gold(n, "symbols")) >= n_ms]
    recs = np.zeros((len(names), n_ms), dtype=_lib.TRACK_REC)
    for c, n in enumerate(names):
        recs[c]["pseudosymbol"] = gold(n, "symbols")[:n_ms]
    start, end = gold(names[0], "start")[:n_ms], gold(names[0], "end")[:n_ms]
    bank = NavigationBitIntegratorBank(len(names))
    got = []
    cuts = [0, 1, 20, 21, 100, 777, 1000, n_ms]
    for a, b in zip(cuts[:-1], cuts[1:]):
        got.append(bank.push_block(recs[:, a:b], start[a:b], end[a:b]))
    got = np.concatenate(got)
    for c, n in enumerate(names):
        ev = gold(n, "events")
        ev = ev[ev[:, 3] < n_ms]
        mine = got[got["channel"] == c]
        assert np.array_equal(mine["receiver_timestamp"], ev[:, 0]), n
        assert np.array_equal(mine["trailing_edge_receiver_timestamp"], ev[:, 1]), n
        assert np.array_equal(mine["bit_value"], ev[:, 2].astype(np.int32)), n
    # emission order: by the millisecond that completed the bit, then channel (receiver.py:244-247)
    want = sorted((int(e[3]), c) for c, n in enumerate(names) for e in gold(n, "events") if e[3] < n_ms)
    assert [c for _, c in want] == list(got["channel"])


def test_native_block_stops_feeding_a_lost_channel() -> None:
    name = "clean_phase7"
    n_ms = 1000
    recs = np.zeros((2, n_ms), dtype=_lib.TRACK_REC)
    recs["pseudosymbol"] = gold(name, "symbols")[:n_ms]
    recs[1, 500:]["status"] = 2
    recs[1, 500]["status"] = 1
    recs[1, 500:]["pseudosymbol"] = 0          # a lost channel's records carry no pseudosymbol
    bank = NavigationBitIntegratorBank(2)
    bank.push_block(recs, gold(name, "start")[:n_ms], gold(name, "end")[:n_ms])
    assert int(bank.state(0)["processed_pseudosymbol_count"]) == 1000
    assert int(bank.state(1)["processed_pseudosymbol_count"]) == 500


def test_native_block_applies_code_phase_delay() -> None:
    name = "clean_phase7"
    n_ms = 400
    recs = np.zeros((1, n_ms), dtype=_lib.TRACK_REC)
    recs["pseudosymbol"] = gold(name, "symbols")[:n_ms]
    recs["code_phase"] = (np.arange(n_ms) * 37) % 2046
    start, end = gold(name, "start")[:n_ms], gold(name, "end")[:n_ms]
    ev = NavigationBitIntegratorBank(1).push_block(recs, start, end)
    ref = gold(name, "events")
    ref = ref[ref[:, 3] < n_ms]
    assert len(ev) == len(ref) > 10
    first = np.searchsorted(start, ref[:, 0])             # ms index of each bit's first / last pseudosymbol
    last = np.searchsorted(end, ref[:, 1])
    delay = lambda ms: (int(recs[0, ms]["code_phase"]) / 2046) * 0.001          # tracker.py:319
    assert [float(e) for e in ev["receiver_timestamp"]] == [start[i] + delay(i) for i in first]
    assert [float(e) for e in ev["trailing_edge_receiver_timestamp"]] == [end[i] + delay(i) for i in last]


def test_native_rejects_bad_input() -> None:
    bank = NavigationBitIntegratorBank(1)
    recs = np.zeros((1, 4), dtype=_lib.TRACK_REC)    # pseudosymbol 0: NavigationBitPseudosymbol.from_val -> KeyError upstream
    with pytest.raises(_lib.GypsumHipError):
        bank.push_block(recs, np.zeros(4), np.zeros(4))
    with pytest.raises(_lib.GypsumHipError):
        NavigationBitIntegratorBank(0)
    with pytest.raises(_lib.GypsumHipError):
        bank.push_block(np.zeros((2, 4), dtype=_lib.TRACK_REC), np.zeros(4), np.zeros(4))
