"""Recording file -> native ingest -> device-resident tracking -> native bit integrator, all through the C ABI.

Checks (i) the ingest path feeds the kernels exactly what the host-buffer path feeds them (records bit-identical),
(ii) the decoded navigation bits are the transmitted ones (up to the Costas loop's 180-degree ambiguity), and
(iii) the bit events equal the oracle integrator run on the same pseudosymbols."""
from __future__ import annotations

import numpy as np
import pytest

import golden_util as gu
from oracle import gypsum_oracle as orc

from gypsum_amd import _lib, synth
from gypsum_amd.engine import default_engine
from gypsum_amd.ingest import IqFileIngest
from gypsum_amd.navigation_bit_intergrator import NavigationBitIntegratorBank

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dtype", [np.float32, np.int8])
def test_file_to_navigation_bits(tmp_path, dtype):
    fs, n = 2_046_000, 2046
    n_ms = 2600
    scene = synth.random_scene(fs, n_ms, 4, 20261001, noise_sigma=0.02)
    iq = synth.render(scene)
    words = np.empty(2 * len(iq), dtype=np.float32)
    words[0::2], words[1::2] = iq.real, iq.imag
    scale = np.float32(1.0)
    if dtype is np.int8:                                   # an 8-bit front end: noise sigma = 2 LSB, saturating
        words = np.clip(np.rint(words * 100), -127, 127).astype(np.int8)
        scale = np.float32(0.01)                            # bring the amplitudes back where the loop constants expect them
    words.tofile(tmp_path / "rec")
    # what the reference's provider yields (antenna_sample_provider.py:119), times the ingest scale
    iq_file = ((words[0::2] * scale) + (1j * (words[1::2] * scale))).astype(np.complex64)

    eng = default_engine(fs, n)
    sat_ids = [s.sat_id for s in scene.sats]
    acq = eng.acquire(iq_file[:10 * n], 1, 10, sat_ids)
    assert all(a["strength"] > 3 for a in acq)
    inits = np.zeros(len(sat_ids), dtype=_lib.CHAN_INIT)
    for i, a in enumerate(acq):
        inits[i] = (0, a["sat_id"], a["doppler_hz"], a["carrier_phase"], a["code_phase"], 0)

    # host-buffer path: the block the C ABI copies itself
    ing = IqFileIngest(tmp_path / "rec", fs, dtype, block_ms=250, depth=3, engine=eng)
    if dtype is np.int8:
        ing.set_scale(scale)
    total = ing.total_ms
    assert total == n_ms - 1                                # the chunk ending exactly at EOF is refused upstream
    start_all, end_all = ing.times(0, total)
    bank_host = eng.create_bank(inits)
    want = bank_host.track_block(iq_file[9 * n:total * n], 1, total - 9, start_all[9:])

    # ingest path
    bank = eng.create_bank(inits)
    bits = NavigationBitIntegratorBank(len(sat_ids))
    ing.seek(9)
    d_times = eng.alloc(250 * 8)
    d_rec = eng.alloc(len(sat_ids) * 250 * _lib.TRACK_REC.itemsize)
    recs, events = [], []
    while (blk := ing.next_device_block()) is not None:
        first, count, dev = blk
        t0, t1 = ing.times(first, count)
        d_times.upload(t0)
        bank.track_block_dev(dev, 0, count, d_times.ptr.value, d_rec.ptr.value)
        r = d_rec.download(_lib.TRACK_REC, len(sat_ids) * count).reshape(len(sat_ids), count)
        recs.append(r)
        events.append(bits.push_block(r, t0, t1))
    got = np.concatenate(recs, axis=1)
    events = np.concatenate(events)
    assert got.shape == want.shape
    assert got.tobytes() == want.tobytes()                  # same samples reached the kernels

    for c, sat in enumerate(scene.sats):
        sym = got[c]["pseudosymbol"].astype(int)
        assert not got[c]["status"].any()
        # (iii) oracle integrator on the same pseudosymbols
        o = orc.BitIntegrator()
        expect = []
        for k in range(total - 9):
            delay = (int(got[c, k]["code_phase"]) / 2046) * 0.001
            _, ev = o.process(float(start_all[9 + k]), float(start_all[9 + k]) + delay, float(end_all[9 + k]) + delay, int(sym[k]))
            expect += ev
        mine = events[events["channel"] == c]
        assert [(float(a), float(b), int(v)) for a, b, v in zip(mine["receiver_timestamp"], mine["trailing_edge_receiver_timestamp"], mine["bit_value"])] == expect
        # (ii) transmitted bits: bit k of the events starts at the ms where its first pseudosymbol started
        assert len(mine) > 100
        known = mine[mine["bit_value"] != _lib.GYP_BIT_UNKNOWN][-100:]
        ms_of_bit = np.rint(known["receiver_timestamp"] * 1000).astype(int)
        sent = np.array([synth.nav_symbol_at(sat, int(m) + 10) for m in ms_of_bit])      # mid-bit
        decoded = known["bit_value"] * 2 - 1
        agree = np.mean(decoded == sent)
        print(f"sv{sat.sat_id}: {len(mine)} bits, {int((mine['bit_value'] == 2).sum())} unknown, agreement {max(agree, 1 - agree):.3f}")
        assert max(agree, 1 - agree) == 1.0
    ing.close()
