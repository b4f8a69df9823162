"""AGC on (DSP_Fn.cpp:504-631): the oracle's restatement against the independent
float64 model (tests/f64_model.agc_f64) on fading signals that walk through every state of the
gain law, plus the derived constants.  CPU only."""
import numpy as np
import pytest

import f64_model as M
import oracle_lib as O
import siggen

L = 2048


def _agc_dict(c):
    return dict(zip(O.AGC_NAMES, np.ctypeslib.as_array(c.agc).astype(np.float64)))


def test_agc_constants_follow_AGCLoadValues(built):
    g = {m: _agc_dict(O.design(O.default_params(AGCMode=m))) for m in range(5)}
    assert all(v == 0.0 for v in g[0].values())
    for m in (1, 2, 3, 4):
        # (int)ceil(24000.f * 4 * 0.001f): the f32 product is 96.0000076, so the look-ahead is 97
        assert g[m]["attack_buffsize"] == 97.0
        assert abs(g[m]["attack_mult"] - (1 - np.exp(-1 / 24.0))) < 1e-7
        assert abs(g[m]["out_target"] - (1 - np.exp(-4.0)) * 0.9999) < 1e-7
        assert abs(g[m]["min_volts"] - g[m]["out_target"] / 15.0) < 1e-8  # var_gain 1.5 * 10^(20/20)
        assert g[m]["pop_ratio"] == 5.0 and g[m]["inv_max_input"] == 1.0
    assert [g[m]["hang_count"] for m in (1, 2, 3, 4)] == [48000.0, 24000.0, 0.0, 0.0]
    tau = {1: 2.0, 2: 0.5, 3: 0.25, 4: 0.05}
    for m, t in tau.items():
        assert abs(g[m]["decay_mult"] - (1 - np.exp(-1 / (24000 * t)))) < 1e-7
    # hang_thresh 0.25 (modes 1, 2) vs 1.0 (modes 3, 4): DSP_Fn.cpp:388, 395, 426-427
    assert abs(g[1]["hang_level"] - (1e-6 + g[1]["min_volts"] * (1 - 1e-6)) * 0.637) < 1e-7
    assert abs(g[3]["hang_level"] - 0.637) < 1e-7
    # AGC_thresh moves max_gain = 10^(thresh/20)
    g30 = _agc_dict(O.design(O.default_params(AGCMode=1, AGC_thresh=30)))
    assert abs(g30["min_volts"] * 10 ** 1.5 - g[1]["min_volts"] * 10) < 1e-6


# (AGCMode, demod mode, frames, fading envelope, states the f64 model must visit)
SCENARIOS = [
    (2, 0, 140, [(0.15, 2.5), (0.75, 0.05), (0.1, 2.5)], {0, 1, 2, 3, 4}),
    (1, 0, 60, [(0.5, 2.5), (0.3, 0.05), (0.2, 1.5)], {0, 1, 2}),
    (3, 1, 40, [(0.4, 2.5), (0.3, 0.05), (0.3, 1.6)], {0, 1, 3}),
    (4, 2, 30, [(0.4, 2.0), (0.3, 0.05), (0.3, 1.6)], {0, 3}),
    (1, 3, 12, [(0.4, 1.0), (0.3, 0.3), (0.3, 1.0)], {0}),
]


@pytest.mark.parametrize("agcmode,mode,nfr,segs,must_visit", SCENARIOS)
def test_oracle_agc_matches_f64_model(built, agcmode, mode, nfr, segs, must_visit):
    nco = 7350
    if mode == 3:
        I, Q = siggen.make_fm(1, nfr * L, [nco], seed=3)
    else:
        I, Q = siggen.make_iq(1, nfr * L, [nco], mode=mode, seed=5, audio_hz=(400.0, 2500.0))
    I, Q = siggen.fade(I, Q, segs)
    flo, fhi = {0: (200, 3000), 1: (-3000, -200), 2: (-3000, 3000), 3: (200, 3000)}[mode]
    p = O.default_params(mode=mode, AGCMode=agcmode, FLoCut=flo, FHiCut=fhi)
    ob = O.OracleBatch(p, [nco])
    out = ob.process(I, Q)[0]
    trace = []
    ref = M.run(I[0], Q[0], nco, O.coeff_arrays(ob.c, 512), mode=mode, FLoCut=flo, FHiCut=fhi,
                agc=_agc_dict(ob.c), agc_trace=trace)
    visited = {s for s, _ in trace}
    assert must_visit <= visited, visited
    err = siggen.block_rel_err(out[None], ref[None], L)
    # the f32 decay `volts += (ring_max - volts) * decay_mult * .05` moves volts by a few ulps per
    # sample, so its rounding is a visible fraction of every step: the f32 restatement drifts
    # ~1e-5..1e-4 from a float64 evaluation over a long decay (inherent to the reference arithmetic)
    assert err.max() < 3e-4, (err.max(), int(err.argmax()))


# fading / burst scenarios that together walk every transition the gain law has
EDGE_SCENARIOS = [
    (2, 140, [(0.15, 2.5), (0.75, 0.05), (0.1, 2.5)]),                                   # 0>1 0>2 0>3 1>0 2>0 2>4 3>0 4>0
    (2, 160, [(0.1, 1.5), (0.05, 0.3), (0.003, 2.5), (0.1, 0.3), (0.003, 2.5), (0.55, 0.3), (0.003, 2.5), (0.191, 0.3)]),  # 1>4
    (2, 60, [(0.25, 0.6), (0.15, 0.2), (0.006, 2.5), (0.594, 0.06)]),                    # 1>3
    (1, 100, [(0.2, 1.5), (0.4, 0.2), (0.003, 2.5), (0.397, 0.2)]),                      # 1>2
]
ALL_EDGES = {(0, 1), (0, 2), (0, 3), (1, 0), (1, 2), (1, 3), (1, 4), (2, 0), (2, 4), (3, 0), (4, 0)}


def agc_edges(agcmode, nfr, segs, seed=5):
    nco = [7350]
    I, Q = siggen.make_iq(1, nfr * L, nco, mode=0, seed=seed, audio_hz=(400.0, 2500.0))
    I, Q = siggen.fade(I, Q, segs)
    ob = O.OracleBatch(O.default_params(mode=0, AGCMode=agcmode), nco)
    out = ob.process(I, Q)
    e = ob.tap(0, O.TAP_AGC_EDGES, 25).reshape(5, 5)
    return {(a, b) for a in range(5) for b in range(5) if a != b and e[a, b] > 0}, I, Q, nco, ob, out


def test_agc_scenarios_walk_every_transition(built):
    """DSP_Fn.cpp:545-626 has eleven state changes; the scenarios the parity tests use hit all of them
    (the GPU's slow path is the only code that takes them, so this is what its tests stand on)"""
    seen = set()
    for agcmode, nfr, segs in EDGE_SCENARIOS:
        edges, I, Q, nco, ob, out = agc_edges(agcmode, nfr, segs)
        seen |= edges
        # and on each of them the restatement still agrees with the float64 model
        ref = M.run(I[0], Q[0], nco[0], O.coeff_arrays(ob.c, 512), mode=0, agc=_agc_dict(ob.c))
        assert siggen.block_rel_err(out, ref[None], L).max() < 3e-4
    assert seen == ALL_EDGES, (ALL_EDGES - seen, seen - ALL_EDGES)


def test_agc_streaming_split_equals_whole(built):
    nco = [-3000, 12000]
    I, Q = siggen.make_iq(2, 12 * L, nco, mode=0, seed=21)
    I, Q = siggen.fade(I, Q, [(0.3, 2.0), (0.4, 0.05), (0.3, 1.0)])
    p = O.default_params(AGCMode=2)
    whole = O.OracleBatch(p, nco).process(I, Q)
    ob = O.OracleBatch(p, nco)
    parts = [ob.process(np.ascontiguousarray(I[:, a * L:b * L]), np.ascontiguousarray(Q[:, a * L:b * L]))
             for a, b in ((0, 1), (1, 5), (5, 12))]
    assert np.array_equal(np.concatenate(parts, axis=1), whole)


def test_agc_volts_tap_and_delay(built):
    """the gain follows the 97-sample look-ahead: a step up in level is met by the gain before it
    reaches the output, so the output never overshoots out_target by much"""
    nco = [5000]
    I, Q = siggen.make_iq(1, 6 * L, nco, mode=0, seed=8)
    I, Q = siggen.fade(I, Q, [(0.5, 0.1), (0.5, 2.5)])
    p = O.default_params(AGCMode=3, audioVolume=100)
    ob = O.OracleBatch(p, nco)
    out = ob.process(I, Q)[0]
    v = ob.tap(0, O.TAP_AGC_VOLTS, 256)
    assert v.min() >= _agc_dict(ob.c)["min_volts"] and np.all(np.isfinite(v))
    # audio = AGC output through two interpolators without make-up gain (1/8) times volume 40
    assert np.abs(out).max() < 40.0 / 8.0 * 1.3
