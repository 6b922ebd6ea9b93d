"""Single-carrier modem row: the CPU restatement (oracle/sc_oracle.py) against vectors produced by the
reference's own `single_carrier` class (oracle/gen_golden_sc.py)."""
import os
import numpy as np
import pytest

from oracle import sc_oracle as SC

GOLD = os.path.join(os.path.dirname(__file__), "golden")
RX_CASES = ["clean", "noisy_foff", "drift", "drift_pos", "lose_sync"]


def test_rrc_taps_match_reference():
    g = np.load(os.path.join(GOLD, "sc_tx_bpsk_1500.npz"))
    assert np.abs(SC.rrc_coeffs(0.25, 2400, 9600) - g["rrc"]).max() < 1e-12


@pytest.mark.parametrize("name", ["bpsk_1500", "analog_0"])
def test_tx_matches_reference(name):
    g = np.load(os.path.join(GOLD, f"sc_tx_{name}.npz"))
    m = SC.SingleCarrier(fcentreHz=float(g["fcentre"]))
    tx = np.stack([m.tx(s) for s in g["symbs"]])
    assert np.abs(tx - g["tx"]).max() < 2e-6          # complex64 samples; the LO phasor is evaluated in closed form


@pytest.mark.parametrize("name", RX_CASES)
def test_rx_matches_reference(name):
    g = np.load(os.path.join(GOLD, f"sc_rx_{name}.npz"))
    m = SC.SingleCarrier(fcentreHz=float(g["fcentre"]))
    out, consumed = SC.run_rx_stream(m, g["rx_in"])
    assert consumed == int(g["consumed"])
    for k in ("state", "nin", "fs_s"):
        assert np.array_equal(out[k], g[k]), k                                   # discrete outputs: exact
    assert np.abs(out["norm_rx_timing"] - g["norm_rx_timing"]).max() < 1e-6
    assert np.abs(out["phase_ambiguity"] - g["phase_ambiguity"]).max() == 0
    assert np.abs(out["g"] - g["g"]).max() < 1e-5 * np.abs(g["g"]).max()
    assert np.abs(out["max_Cs"] - g["max_Cs"]).max() < 1e-5
    assert np.abs(out["payload"] - g["payload"]).max() < 1e-5 * max(1.0, np.abs(g["payload"]).max())
