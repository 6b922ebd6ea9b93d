#!/usr/bin/env python3
"""Golden vectors for the single-carrier modem row, made by importing the reference here.

    PYTHONPATH=/root/reference python oracle/gen_golden_sc.py

Runs /root/reference/radae/dsp.py's `single_carrier` (tx at 1500 Hz and at baseband; rx over streams impaired the way
its own run_test() does: 4x oversampling + sample-clock offset, frequency / phase / gain offsets, AWGN) and stores
inputs and per-frame outputs in tests/golden/sc_*.npz.  The reference is only imported, never copied.
"""
import os
import sys
import numpy as np

sys.path.insert(0, "/root/reference")
import matplotlib
matplotlib.use("Agg")
from radae.dsp import single_carrier, sample_clock_offset   # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def tx_case(name, fcentre, nframes, seed, analog):
    rng = np.random.default_rng(seed)
    m = single_carrier(fcentreHz=fcentre)
    if analog:
        symbs = np.clip(rng.standard_normal((nframes, 80)), -1, 1).astype(np.float32)       # BBFM latents are real numbers in [-1, 1]
    else:
        symbs = (1 - 2 * (rng.random((nframes, 80)) > 0.5)).astype(np.float32)
    tx = np.stack([m.tx(symbs[f]) for f in range(nframes)])
    np.savez_compressed(os.path.join(OUT, f"sc_tx_{name}.npz"), fcentre=fcentre, symbs=symbs, tx=tx, rrc=m.rrc)
    return symbs, tx.reshape(-1)


def rx_case(name, fcentre, nframes, seed, EbNodB, phase_off, freq_off, mag, ppm, lead, noise_tail=0):
    rng = np.random.default_rng(seed)
    t = single_carrier(fcentreHz=fcentre)
    symbs = (1 - 2 * (rng.random((nframes, 80)) > 0.5)).astype(np.float32)
    tx = np.concatenate([t.tx(symbs[f]) for f in range(nframes)])
    if ppm != 0:                                                       # run_test(): 4x oversample, drift, back to Fs
        zp = np.zeros(4 * len(tx), np.csingle); zp[0::4] = tx
        rx = sample_clock_offset(t.lpf.bpf(zp), ppm)[0::4]
    else:
        rx = tx.copy()
    rx = np.concatenate([np.zeros(lead, np.csingle), rx])
    ph = 2 * np.pi * freq_off * np.arange(len(rx)) / t.Fs + phase_off
    rx = rx * np.exp(1j * ph)
    sigma = np.sqrt(1 / (t.M * 10 ** (EbNodB / 10)))
    noise = (sigma / np.sqrt(2)) * (rng.standard_normal(len(rx)) + 1j * rng.standard_normal(len(rx)))
    rx = (mag * (rx + noise)).astype(np.csingle)
    if noise_tail:                                                     # signal disappears: frame-sync errors drop the modem back to search
        k = noise_tail * 384
        rx[-k:] = (0.7 * (rng.standard_normal(k) + 1j * rng.standard_normal(k))).astype(np.csingle)
    m = single_carrier(fcentreHz=fcentre)
    log = {k: [] for k in ("state", "nin", "fs_s", "norm_rx_timing", "g", "max_Cs", "phase_ambiguity", "payload")}
    n = 0
    m.max_Cs = 0
    while len(rx) - n >= m.nin:
        nin = m.nin
        pay = m.rx(rx[n:n + nin]); n += nin
        log["state"].append(1 if m.state == "sync" else 0); log["nin"].append(m.nin); log["fs_s"].append(m.fs_s)
        log["norm_rx_timing"].append(m.norm_rx_timing); log["g"].append(m.g); log["max_Cs"].append(complex(m.max_Cs))
        log["phase_ambiguity"].append(m.phase_ambiguity); log["payload"].append(np.asarray(pay, np.complex128))
    np.savez_compressed(os.path.join(OUT, f"sc_rx_{name}.npz"), fcentre=fcentre, symbs=symbs, rx_in=rx, consumed=n,
                        **{k: np.array(v) for k, v in log.items()})
    st = np.array(log["state"])
    print(name, "frames", len(st), "synced", int(st.sum()), "nin values", sorted(set(log["nin"])), "fs_s", sorted(set(log["fs_s"])))


def wire_case():
    """The reference's own filters end to end: 12 frames of latents | sc_tx.py | sc_rx.py (int16 real samples at 1500 Hz)."""
    import subprocess
    rng = np.random.default_rng(11)
    z = np.clip(rng.standard_normal((12, 80)), -1, 1).astype(np.float32)
    env = dict(os.environ, PYTHONPATH="/root/reference", MPLBACKEND="Agg")
    t = subprocess.run([sys.executable, "/root/reference/sc_tx.py"], input=z.tobytes(), capture_output=True, env=env, check=True).stdout
    zh = subprocess.run([sys.executable, "/root/reference/sc_rx.py", "-v", "0"], input=t, capture_output=True, env=env, check=True).stdout
    np.savez_compressed(os.path.join(OUT, "sc_wire.npz"), z=z, t_int16=np.frombuffer(t, np.int16), zhat=np.frombuffer(zh, np.float32).reshape(-1, 80))


if __name__ == "__main__":
    wire_case()
    tx_case("bpsk_1500", 1500.0, 6, 1, False)
    tx_case("analog_0", 0.0, 5, 2, True)
    rx_case("clean", 1500.0, 12, 3, 100.0, 0.0, 0.0, 1.0, 0, 37)
    rx_case("noisy_foff", 1500.0, 30, 4, 6.0, 0.7, 1.5, 0.5, 0, 101)
    rx_case("drift", 0.0, 40, 5, 12.0, -2.0, -1.0, 2.0, -100, 0)
    rx_case("drift_pos", 1500.0, 40, 6, 20.0, 1.0, 0.5, 1.0, 200, 13)
    rx_case("lose_sync", 1500.0, 16, 7, 100.0, 0.0, 0.0, 1.0, 0, 5, noise_tail=8)
