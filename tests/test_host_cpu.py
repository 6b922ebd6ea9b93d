"""CPU-only tests of the product's host side: the C ABI library loads and exports every symbol the
headers declare, the plain-C model preparation (tables, DNNw reader, weight packing) matches the golden
vectors, the library refuses to run without a GPU, and the host tools (loss, channel generators)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from radae_amd import engine
    return engine.load_library()


def test_built_kernels_do_not_spill_ahead_of_exec_restore(lib):
    """Code-generation guard for the shipped library (tools/check_spill_exec.py): a VGPR spill placed in a control-flow
    join block before `s_or_b64 exec, exec, ...` is lost when the wavefront skipped the branch; the checker itself is
    exercised on a hand-made disassembly of the bad and the good placement."""
    import sys
    sys.path.insert(0, os.path.join(REPO, "tools"))
    import check_spill_exec as chk
    from radae_amd import engine
    assert chk.scan(chk.disassemble(engine.LIB_PATH)) == []
    bad = """
0000000000001000 <k>:
\ts_and_saveexec_b64 s[2:3], s[14:15]  // 000000001000: BE82200E
\ts_cbranch_execz 3  // 000000001004: BF880003 <k+0x14>
\tds_write_b32 v4, v3  // 000000001008: D81A0000 00000304
\ts_nop 0  // 000000001010: BF800000
\ts_movk_i32 s34, 0x1bf  // 000000001014: B02201BF
\tscratch_store_dwordx2 off, v[190:191], off offset:212  // 000000001018: DC7440D4 007FBE00
\ts_or_b64 exec, exec, s[2:3]  // 000000001020: 87FE027E
\ts_endpgm  // 000000001024: BF810000
"""
    hits = chk.scan(bad)
    assert len(hits) == 1 and hits[0][0] == "k" and hits[0][1] == 0x1020
    good = bad.replace("\tscratch_store_dwordx2 off, v[190:191], off offset:212  // 000000001018: DC7440D4 007FBE00\n\ts_or_b64 exec, exec, s[2:3]  // 000000001020: 87FE027E",
                       "\ts_or_b64 exec, exec, s[2:3]  // 000000001018: 87FE027E\n\tscratch_store_dwordx2 off, v[190:191], off offset:212  // 00000000101C: DC7440D4 007FBE00")
    assert good != bad and chk.scan(good) == []


def test_library_exports_every_declared_symbol(lib):
    declared = set()
    for hdr in ("rade_api.h", "rade_batch.h", "rade_core.h"):
        src = open(os.path.join(REPO, "include", hdr)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        declared |= set(re.findall(r"\b((?:rade|init_rade)[a-zA-Z0-9_]+)\s*\(", src))
    declared -= {"rade_batch", "rade_batch_config", "rade_channel_params", "rade_rx_status", "rade_rx_trace"}
    assert len(declared) >= 35
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    from radae_amd import engine
    assert set(engine.EXPORTED_SYMBOLS) <= declared


def test_core_level_model_init_on_cpu(lib):
    """include/rade_core.h without a GPU: the DNNw walker (Opus parse_weights() role) lists the blob's arrays, init_radeenc /
    init_radedec accept exactly the dimension the blob was exported with (README.md:582-588: 80 <-> 84 at run time)."""
    from radae_amd import core, dnnw
    blob19 = open(os.path.join(REPO, "weights", "model19_check3.bin"), "rb").read()
    arrays = core.parse_weights(blob19)
    ref = dnnw.read_records(os.path.join(REPO, "weights", "model19_check3.bin"))
    assert [a[0] for a in arrays] == list(ref) and all(a[2] == ref[a[0]].nbytes for a in arrays)
    with pytest.raises(ValueError):
        core.parse_weights(b"DNNx" + blob19[4:])
    with pytest.raises(ValueError):
        core.parse_weights(blob19[:1000])
    for path, good, bad in (("model19_check3.bin", 84, 80), ("model05.bin", 80, 84)):
        L = core._lib()
        buf = C.create_string_buffer(open(os.path.join(REPO, "weights", path), "rb").read())
        lst = C.POINTER(core.WeightArray)()
        assert L.rade_parse_weights(C.byref(lst), C.cast(buf, C.c_void_p), len(buf) - 1) > 100
        m = core._Model()
        assert L.init_radeenc(C.byref(m), lst, good) == 0 and m.dim == good and m.nb_z == 80
        assert L.init_radeenc(C.byref(m), lst, bad) != 0 and L.init_radedec(C.byref(m), lst, bad) != 0 and L.init_radedec(C.byref(m), lst, 64) != 0
        assert L.init_radedec(C.byref(m), lst, good) == 0
        # the reference harnesses free(list) after init (test_rade_enc.c:115, test_rade_dec.c:115): *list must be the malloc'ed pointer
        # itself (glibc aborts on an interior pointer), and a model initialised from it keeps working from the blob alone
        libc = C.CDLL(None); libc.free.argtypes = [C.c_void_p]; libc.free.restype = None
        n_names = 0
        while lst[n_names].name:
            n_names += 1
        assert n_names > 100
        libc.free(C.cast(lst, C.c_void_p))
        assert m.dim == good


def test_multi_gpu_sharding_rule(lib):
    """rade_multi_shard: contiguous balanced shards (no device left empty when there are at least as many streams as devices: 9 or 10
    streams on 8 GPUs) -- config 4: 2048 utterances, GPU g owns [256 g, 256 g + 256) -- and the same rule as the torchrun path's
    radae_amd.parallel.shard_range; rade_multi_open refuses to run without a GPU."""
    from radae_amd.parallel import shard_range
    lib.rade_multi_shard.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]; lib.rade_multi_shard.restype = None
    for total, ndev in ((2048, 8), (256, 1), (10, 4), (7, 8), (1000, 3), (9, 8), (10, 8), (8, 8), (2047, 8)):
        got = []
        for i in range(ndev):
            lo, n = C.c_int(), C.c_int()
            lib.rade_multi_shard(total, ndev, i, C.byref(lo), C.byref(n))
            got.append((lo.value, lo.value + n.value))
        assert got == [shard_range(total, r, ndev) for r in range(ndev)]
        assert sum(b - a for a, b in got) == total and got[0][0] == 0 and all(got[i][1] == got[i + 1][0] for i in range(ndev - 1))
        sizes = [b - a for a, b in got]
        assert max(sizes) - min(sizes) <= 1 and (total < ndev or min(sizes) >= 1)
    lo, n = C.c_int(), C.c_int()
    lib.rade_multi_shard(2048, 8, 5, C.byref(lo), C.byref(n))
    assert (lo.value, n.value) == (1280, 256)
    import torch
    if not torch.cuda.is_available():
        lib.rade_multi_open.restype = C.c_void_p; lib.rade_multi_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_ulonglong, C.c_int]
        from radae_amd import engine
        assert not lib.rade_multi_open(engine.DEFAULT_BLOB.encode(), 16, 1, 1, 0)


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from radae_amd import engine
    cfg = engine.BatchConfig(1, 1, 0, 0, 0, 0.0)
    h = lib.rade_batch_open(engine.DEFAULT_BLOB.encode(), C.byref(cfg))
    assert not h                                     # fails loudly (message on stderr), never computes on the CPU
    with pytest.raises(RuntimeError):
        engine.BatchEngine(1)


class Tables(C.Structure):
    _fields_ = [("Winv", C.c_float * (30 * 160 * 2)), ("Wfwd", C.c_float * (160 * 30 * 2)), ("P", C.c_float * 30), ("Pend", C.c_float * 30),
                ("p", C.c_float * 320), ("pend", C.c_float * 320), ("eoo", C.c_float * 2304), ("Pmat", C.c_float * 360), ("eq_rot", C.c_float * 60),
                ("bpf_h", C.c_float * 104), ("bpf_E", C.c_float * 2304), ("p_w", C.c_float * 12800), ("fcoarse", C.c_double * 40),
                ("pilot_gain", C.c_float), ("snr_c1", C.c_float), ("snr_c2", C.c_float), ("pad", C.c_float)]


def _c(a):
    return np.ctypeslib.as_array(a).view(np.complex64)


def test_host_tables_match_reference_constants(lib, golden):
    c = golden("consts")
    T = Tables()
    lib.rd_tables_fill.argtypes = [C.POINTER(Tables)]
    lib.rd_tables_fill(C.byref(T))
    assert np.abs(_c(T.Winv) - c["Winv"].ravel()).max() < 2e-9
    assert np.abs(_c(T.Wfwd) - c["Wfwd"].ravel()).max() < 2e-7
    assert np.array_equal(np.ctypeslib.as_array(T.P), c["P"].real) and np.array_equal(np.ctypeslib.as_array(T.Pend), c["Pend"].real)
    assert np.abs(_c(T.p) - c["p"]).max() < 5e-8 and np.abs(_c(T.pend) - c["pend"]).max() < 5e-8
    assert np.abs(_c(T.eoo) - c["eoo_default"]).max() < 1e-6
    assert np.abs(_c(T.Pmat) - c["Pmat"].ravel()).max() < 2e-6
    assert np.abs(np.ctypeslib.as_array(T.bpf_h)[:101] - c["bpf_h"].real).max() < 5e-9
    assert np.abs(_c(T.bpf_E)[:1120] - c["bpf_phase_vec_exp"]).max() < 2e-7
    assert np.abs(_c(T.p_w) - c["acq_p_w"].ravel()).max() < 5e-8
    assert np.array_equal(np.ctypeslib.as_array(T.fcoarse), c["acq_fcoarse"])
    assert T.pilot_gain == pytest.approx(float(c["pilot_gain"]), rel=1e-7)


def test_demodulator_dft_matrix_as_matrix_core_operands(lib, golden):
    """rd_wfwd16_table_fill: the forward DFT matrix as rows (wr, -wi) in two binary16 planes, in the A-operand order of v_mfma_f32_16x16x32_f16 (k_rx_sync2's
    demodulator).  Un-permuted and applied (on the CPU, in float64) to a random window as (xr, xi) it must give the real part of the reference's Wfwd product,
    applied to (xi, -xr) the imaginary part, to the planes' 22 bits."""
    c = golden("consts")
    T = Tables()
    lib.rd_tables_fill.argtypes = [C.POINTER(Tables)]; lib.rd_tables_fill(C.byref(T))
    out = np.zeros(2 * 10 * 2 * 64 * 8, np.uint16)
    lib.rd_wfwd16_table_fill.argtypes = [C.POINTER(Tables), C.c_void_p]; lib.rd_wfwd16_table_fill.restype = None
    lib.rd_wfwd16_table_fill(C.byref(T), out.ctypes.data_as(C.c_void_p))
    tab = out.view(np.float16).astype(np.float64).reshape(2, 10, 2, 64, 8)
    R = np.zeros((32, 320))
    for tile in range(2):
        for s in range(10):
            for lane in range(64):
                R[16 * tile + (lane & 15), 32 * s + 8 * (lane >> 4):32 * s + 8 * (lane >> 4) + 8] = (tab[tile, s, 0, lane] + tab[tile, s, 1, lane]) / 4096.0
    assert not R[30:].any()                                       # rows 30, 31: padding
    rng = np.random.default_rng(5)
    x = rng.standard_normal(160) + 1j * rng.standard_normal(160)
    v0 = np.empty(320); v0[0::2] = x.real; v0[1::2] = x.imag
    v1 = np.empty(320); v1[0::2] = x.imag; v1[1::2] = -x.real
    ref = x @ c["Wfwd"].astype(np.complex128)                     # sym[c] = sum_n x[n] Wfwd[n][c] (dsp.py:501)
    assert np.abs(((R @ v0)[:30] + 1j * (R @ v1)[:30]) - ref).max() < 3e-6 * np.abs(ref).max()


def test_two_stage_pilot_correlator_tables(lib, golden):
    """rd_corrq16_table_fill / rd_corra16_table_fill (rade_host.c): acquisition.p_w (dsp.py:166-173) factored as 16 polynomial moments x their expansion to the 40
    coarse frequencies.  Un-permuted from the matrix-core operand order (stage 2's K axis in the accumulator order of stage 1) and multiplied back together on the
    CPU, the two tables give the reference's p_w to the planes' 22 bits; applied in two stages to a random window they give the reference's Dt
    (dsp.py:207-208) -- and the exact factorisation (before the binary16 split) reproduces p_w to 1e-8."""
    c = golden("consts")
    T = Tables()
    lib.rd_tables_fill.argtypes = [C.POINTER(Tables)]; lib.rd_tables_fill(C.byref(T))
    lib.rd_corr_tables_check.argtypes = [C.POINTER(Tables)]; lib.rd_corr_tables_check.restype = C.c_double
    assert lib.rd_corr_tables_check(C.byref(T)) < 1e-8            # (p_w itself is rounded to float32: 0.124 x 6e-8)
    q16 = np.zeros(2 * 10 * 2 * 64 * 8, np.uint16); a16 = np.zeros(5 * 2 * 64 * 8, np.uint16)
    for fn, buf in (("rd_corrq16_table_fill", q16), ("rd_corra16_table_fill", a16)):
        getattr(lib, fn).argtypes = [C.POINTER(Tables), C.c_void_p]; getattr(lib, fn).restype = None
        getattr(lib, fn)(C.byref(T), buf.ctypes.data_as(C.c_void_p))
    tq = q16.view(np.float16).astype(np.float64).reshape(2, 10, 2, 64, 8); ta = a16.view(np.float16).astype(np.float64).reshape(5, 2, 64, 8)
    Q1 = np.zeros((32, 320)); A2 = np.zeros((80, 32))
    for tile in range(2):
        for s in range(10):
            for lane in range(64):
                Q1[16 * tile + (lane & 15), 32 * s + 8 * (lane >> 4):32 * s + 8 * (lane >> 4) + 8] = (tq[tile, s, 0, lane] + tq[tile, s, 1, lane]) / 4096.0
    for tile in range(5):
        for lane in range(64):
            g = lane >> 4
            for j in range(8):
                A2[16 * tile + (lane & 15), (4 * g + j) if j < 4 else (16 + 4 * g + j - 4)] = (ta[tile, 0, lane, j] + ta[tile, 1, lane, j]) / 1024.0
    # the product of the two real matrices is the realified p_w: rows (f, re | im), columns (m, re | im) as in rd_corr16_table_fill
    pw = c["acq_p_w"].astype(np.complex128)                        # [m][f]
    P = np.zeros((80, 320)); P[0::2, 0::2] = pw.real.T; P[0::2, 1::2] = pw.imag.T; P[1::2, 0::2] = pw.imag.T; P[1::2, 1::2] = -pw.real.T
    assert np.abs(A2 @ Q1 - P).max() < 4e-7 * np.abs(P).max()
    rng = np.random.default_rng(9)
    x = rng.standard_normal(160) + 1j * rng.standard_normal(160)
    v = np.empty(320); v[0::2] = x.real; v[1::2] = x.imag
    d = A2 @ (Q1 @ v)
    ref = np.conj(x) @ pw                                          # Dt[t, :] = matmul(conj(rx[t:t+M]), p_w)
    assert np.abs((d[0::2] + 1j * d[1::2]) - ref).max() < 4e-7 * np.abs(ref).max()
    assert np.abs(A2[:, 28:]).max() < 1e-4 * np.abs(A2).max()     # the highest moments carry next to nothing: 16 are plenty


def test_bandpass_taps_as_matrix_core_operands(lib, golden):
    """rd_bpf16_table_fill: complex_bpf's 101 taps as the Toeplitz A operand of the matrix-core FIR (k_rx_bpf), two binary16 planes in the lane order of
    v_mfma_f32_16x16x32_f16.  Un-permuted and applied (CPU, float64) to the Hankel matrix of a random window it must reproduce the direct FIR sums of
    the reference's taps to the planes' 22 bits."""
    c = golden("consts")
    T = Tables()
    lib.rd_tables_fill.argtypes = [C.POINTER(Tables)]; lib.rd_tables_fill(C.byref(T))
    out = np.zeros(4 * 2 * 64 * 8, np.uint16)
    lib.rd_bpf16_table_fill.argtypes = [C.POINTER(Tables), C.c_void_p]; lib.rd_bpf16_table_fill.restype = None
    lib.rd_bpf16_table_fill(C.byref(T), out.ctypes.data_as(C.c_void_p))
    tab = out.view(np.float16).astype(np.float64).reshape(4, 2, 64, 8)
    Tm = np.zeros((16, 128))
    for ks in range(4):
        for lane in range(64):
            Tm[lane & 15, 32 * ks + 8 * (lane >> 4):32 * ks + 8 * (lane >> 4) + 8] = (tab[ks, 0, lane] + tab[ks, 1, lane]) / 1024.0
    h = c["bpf_h"].real.astype(np.float64)
    for r in range(16):
        assert np.abs(Tm[r, r:r + 101] - h).max() < 2.0 ** -22 * np.abs(h).max() and not Tm[r, :r].any() and not Tm[r, r + 101:].any()
    rng = np.random.default_rng(6)
    w = rng.standard_normal(256 + 127)
    U = np.stack([w[16 * q:16 * q + 128] for q in range(16)], axis=1)          # U[m][q] = w[16 q + m]
    Y = Tm @ U                                                                  # Y[r][q] = y[16 q + r]
    ref = np.array([np.dot(h, w[i:i + 101]) for i in range(256)])
    assert np.abs(Y.T.ravel() - ref).max() < 1e-6 * np.abs(ref).max()


class Lin(C.Structure):
    _fields_ = [("n_in", C.c_int), ("n_out", C.c_int), ("w", C.POINTER(C.c_float)), ("b", C.POINTER(C.c_float)), ("row_scale", C.POINTER(C.c_float))]


class Gru(C.Structure):
    _fields_ = [("n_in", C.c_int), ("hid", C.c_int), ("w_ih", C.POINTER(C.c_float)), ("w_hh", C.POINTER(C.c_float)), ("b_ih", C.POINTER(C.c_float)), ("b_hh", C.POINTER(C.c_float)),
                ("s_ih", C.POINTER(C.c_float)), ("s_hh", C.POINTER(C.c_float))]


class Model(C.Structure):
    _fields_ = [("enc_dense1", Lin), ("enc_zdense", Lin), ("dec_dense1", Lin), ("dec_output", Lin), ("enc_gru", Gru * 5), ("dec_gru", Gru * 5),
                ("enc_conv", Lin * 5), ("dec_conv", Lin * 5), ("dec_glu", Lin * 5)]


def test_host_blob_reader_and_packing(lib, golden):
    from radae_amd import dnnw, engine
    blob = open(engine.DEFAULT_BLOB, "rb").read()
    m = Model()
    lib.rd_model_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Model)]
    assert lib.rd_model_parse(blob, len(blob), C.byref(m)) == 0
    ref = dnnw.load_model(engine.DEFAULT_BLOB)
    arr = lambda p, n: np.ctypeslib.as_array(p, shape=(n,))
    assert np.array_equal(arr(m.enc_dense1.w, 64 * 84).reshape(64, 84), ref.enc_dense1.w)
    assert np.array_equal(arr(m.dec_output.w, 84 * 736).reshape(84, 736), ref.dec_output.w)
    for i in range(5):
        g, r = m.enc_gru[i], ref.enc_gru[i]
        assert np.array_equal(arr(g.w_ih, 192 * g.n_in).reshape(192, -1), r.w_ih) and np.array_equal(arr(g.w_hh, 192 * 64).reshape(192, 64), r.w_hh)
        assert np.array_equal(arr(g.b_ih, 192), r.b_ih) and np.array_equal(arr(g.b_hh, 192), r.b_hh)
        g, r = m.dec_gru[i], ref.dec_gru[i]
        assert np.array_equal(arr(g.w_ih, 288 * g.n_in).reshape(288, -1), r.w_ih) and np.array_equal(arr(g.w_hh, 288 * 96).reshape(288, 96), r.w_hh)
        cw = ref.enc_conv[i].w.transpose(0, 2, 1).reshape(96, -1)         # [out][tap][in]
        assert np.array_equal(arr(m.enc_conv[i].w, cw.size).reshape(cw.shape), cw)
        assert np.array_equal(arr(m.dec_glu[i].w, 96 * 96).reshape(96, 96), ref.dec_glu[i].w)
    assert np.array_equal(arr(m.enc_zdense.w, 80 * 864).reshape(80, 864), ref.enc_zdense.w)       # (pinned against the reference's exporter: tests/test_dnnw_export.py)
    # truncated / corrupt blobs are rejected, not mis-parsed
    assert lib.rd_model_parse(blob[:100000], 100000, C.byref(Model())) != 0
    assert lib.rd_model_parse(b"XXXX" + blob[4:], len(blob), C.byref(Model())) != 0
    # int8-exact layers: one binary16 plane of integers + row scales reproduces the de-quantised weights bit for bit
    lib.rd_pack_weights_q16_a16.restype = C.c_long
    lib.rd_pack_weights_q16_a16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    for lin, N, K in ((m.dec_conv[3], 32, 1152), (m.dec_glu[0], 96, 96)):
        assert bool(lin.row_scale)
        plane = np.zeros((K // 32) * ((N + 15) // 16) * 64 * 8, np.float16); sc = np.zeros(((N + 15) // 16) * 16, np.float32)
        assert lib.rd_pack_weights_q16_a16(lin.w, lin.row_scale, N, K, plane.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p)) == plane.size
        W = arr(lin.w, N * K).reshape(N, K)
        q = plane.reshape(K // 32, (N + 15) // 16, 4, 16, 8)                     # [ks][ct][k-slice][n][j]
        Wr = (q.transpose(1, 3, 0, 2, 4).reshape(-1, K).astype(np.float32) * sc[:, None])[:N]
        assert np.array_equal(Wr, W) and np.abs(plane).max() <= 127 and np.array_equal(plane, np.round(plane))
    g = m.dec_gru[2]
    assert bool(g.s_ih) and not bool(m.dec_output.row_scale)                     # float layers keep two planes
    bad = arr(m.dec_glu[0].w, 96 * 96).copy(); bad[5] *= 1.0000001
    assert lib.rd_pack_weights_q16_a16(bad.ctypes.data_as(C.c_void_p), m.dec_glu[0].row_scale, 96, 96, plane.ctypes.data_as(C.c_void_p), sc.ctypes.data_as(C.c_void_p)) < 0
    # packing: every weight lands exactly once where k_gemm's lane expects it
    lib.rd_packed_size.restype = C.c_long; lib.rd_packed_size.argtypes = [C.c_int, C.c_int]
    lib.rd_pack_weights.restype = C.c_long; lib.rd_pack_weights.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    N, K = 84, 88
    W = np.arange(N * K, dtype=np.float32).reshape(N, K) + 1
    n = lib.rd_packed_size(N, K)
    out = np.zeros(n, np.float32)
    assert lib.rd_pack_weights(W.ctypes.data_as(C.c_void_p), N, K, out.ctypes.data_as(C.c_void_p)) == n
    ntt = 3
    for kb, nt, lane, s in [(0, 0, 0, 0), (3, 2, 45, 1), (10, 1, 63, 3), (5, 2, 31, 2)]:
        nn, k = nt * 32 + (lane & 31), kb * 8 + 4 * (lane >> 5) + s
        exp = W[nn, k] if nn < N else 0.0
        assert out[((kb * ntt + nt) * 64 + lane) * 4 + s] == exp
    assert np.sort(out[out != 0]).tolist() == np.sort(W.ravel()).tolist()


def test_chunk_major_packing_for_the_single_stream_kernel(lib):
    """rd_chunkmajor_q16 / rd_chunkmajor_f32 (rade_core_step.hip's operand layout): out[(c * Npad + n) * 8 + j] = W[n][8 c + j], rows and K zero-padded;
    the binary16 plane holds the integers of an int8 x scale layer exactly and refuses anything else."""
    rng = np.random.default_rng(11)
    N, K, Kpad, Npad = 80, 84, 88, 128
    q = rng.integers(-127, 128, size=(N, K)).astype(np.float32)
    sc = (rng.uniform(0.001, 0.02, size=N)).astype(np.float32)
    W = (q * sc[:, None]).astype(np.float32)
    out = np.zeros(Npad * Kpad, np.uint16)
    lib.rd_chunkmajor_q16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]; lib.rd_chunkmajor_q16.restype = C.c_int
    assert lib.rd_chunkmajor_q16(W.ctypes.data, sc.ctypes.data, N, K, Kpad, Npad, out.ctypes.data) == 0
    got = out.view(np.float16).astype(np.float32).reshape(Kpad // 8, Npad, 8)
    exp = np.zeros((Kpad // 8, Npad, 8), np.float32)
    for c in range(Kpad // 8):
        for j in range(8):
            if 8 * c + j < K: exp[c, :N, j] = q[:, 8 * c + j]
    assert np.array_equal(got, exp)
    W2 = W.copy(); W2[3, 5] *= np.float32(1.0001)                  # not an integer multiple of its row scale any more
    assert lib.rd_chunkmajor_q16(W2.ctypes.data, sc.ctypes.data, N, K, Kpad, Npad, out.ctypes.data) != 0
    outf = np.zeros(Npad * Kpad, np.float32)
    lib.rd_chunkmajor_f32.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]; lib.rd_chunkmajor_f32.restype = None
    lib.rd_chunkmajor_f32(W.ctypes.data, N, K, Kpad, Npad, outf.ctypes.data)
    gf = outf.reshape(Kpad // 8, Npad, 8)
    assert all(np.array_equal(gf[c, :N, j], W[:, 8 * c + j]) for c in range(Kpad // 8) for j in range(8) if 8 * c + j < K) and not gf[:, N:, :].any() and not gf[-1, :, 4:].any()


def test_host_blob_reader_rejects_hostile_headers(lib):
    """The blob path is caller-controlled (rade_open argument / $RADE_MODEL_FILE): record headers with negative or
    inconsistent sizes, wild sparse-index entries and zero-sized arrays must be refused before anything is written."""
    import struct
    from radae_amd import dnnw, engine
    blob = bytearray(open(engine.DEFAULT_BLOB, "rb").read())
    lib.rd_model_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Model)]
    recs, off = {}, 0                                    # name -> (header offset, size, block)
    while off + 64 <= len(blob):
        ver, typ, size, block = struct.unpack_from("<iiii", blob, off + 4)
        recs[bytes(blob[off + 20:off + 64]).split(b"\0")[0].decode()] = (off, size, block)
        off += 64 + block
    assert "enc_gru1_input_weights_idx" in recs and "dec_conv2_bias" in recs

    def rejected(mut):
        b = bytearray(blob); mut(b)
        return lib.rd_model_parse(bytes(b), len(b), C.byref(Model())) != 0

    o, size, block = recs["enc_gru1_input_weights_idx"]
    assert rejected(lambda b: struct.pack_into("<i", b, o + 16, -64))                 # negative block: wraps the bounds check
    assert rejected(lambda b: struct.pack_into("<i", b, o + 12, -4))                  # negative size
    assert rejected(lambda b: struct.pack_into("<i", b, o + 12, block + 4096))        # size beyond its block
    assert rejected(lambda b: struct.pack_into("<i", b, o + 64, 1 << 28))             # first group's block count runs off the index array
    assert rejected(lambda b: struct.pack_into("<i", b, o + 68, 1 << 30))             # a column index far outside the matrix
    assert rejected(lambda b: struct.pack_into("<i", b, o + 68, -8))
    o2, size2, _ = recs["dec_conv2_bias"]
    assert rejected(lambda b: struct.pack_into("<i", b, o2 + 12, 0))                  # n_out = 0: no division by zero
    assert rejected(lambda b: struct.pack_into("<i", b, o2 + 12, size2 - 4))          # bias / scale / weight sizes disagree
    o3, size3, _ = recs["enc_conv1_weights_int8"]
    assert rejected(lambda b: struct.pack_into("<i", b, o3 + 12, size3 - 96))         # weight count not n_in x n_out
    o4, size4, _ = recs["enc_gru2_input_weights_int8"]
    assert rejected(lambda b: struct.pack_into("<i", b, o4 + 12, size4 - 32))         # fewer 8x4 blocks than the index walk needs
    # self-consistent but SMALLER dense layers (a foreign model): the engine's upload hard-codes 64 / 96 / 736, so the parser refuses them
    def shrink(b, name, n_in, n_out):
        struct.pack_into("<i", b, recs[name + "_weights_float"][0] + 12, n_in * n_out * 4); struct.pack_into("<i", b, recs[name + "_bias"][0] + 12, n_out * 4)
    assert not rejected(lambda b: shrink(b, "enc_dense1", 84, 64))                    # (the unchanged shape parses)
    assert rejected(lambda b: shrink(b, "enc_dense1", 84, 32))
    assert rejected(lambda b: shrink(b, "dec_dense1", 80, 48))
    assert rejected(lambda b: shrink(b, "dec_output", 368, 84))
    rng = np.random.default_rng(5)                                                    # random header bytes: refuse or parse, never crash
    for _ in range(200):
        b = bytearray(blob)
        o5 = list(recs.values())[int(rng.integers(0, len(recs)))][0]
        struct.pack_into("<i", b, o5 + int(rng.choice([8, 12, 16])), int(rng.integers(-2**31, 2**31 - 1)))
        lib.rd_model_parse(bytes(b), len(b), C.byref(Model()))


def test_host_wait_policy_switches_to_blocking_sync_under_a_cpu_quota(lib):
    """rade_batch_rx spins on its stream while every engine's host thread has a CPU and sleeps on a blocking event otherwise: 8 GPUs x 3 batches
    in flight = 24 engines under the 16-core quota seen on the GPU lease must block; one GPU x 3 must not.  rade_host_cpu_quota reads what
    bench.py's cpu_quota() reads (affinity mask, cgroup cpu.max)."""
    import ctypes as C
    lib.rade_host_cpu_quota.restype = C.c_double
    lib.rade_sync_policy.argtypes = [C.c_int, C.c_double]
    assert lib.rade_sync_policy(24, 16.0) == 1 and lib.rade_sync_policy(3, 16.0) == 0 and lib.rade_sync_policy(16, 16.0) == 0 and lib.rade_sync_policy(17, 16.0) == 1
    assert lib.rade_sync_policy(2, 1.5) == 1 and lib.rade_sync_policy(1, 1.5) == 0
    import bench
    ncpu, quota = bench.cpu_quota()
    q = lib.rade_host_cpu_quota()
    assert abs(q - min(ncpu, quota if quota else ncpu)) < 1e-9 and q >= 1.0 - 1e-9


def test_loss_tool_matches_reference(golden):
    from radae_amd.loss import distortion_loss, find_loss
    g = golden("dec_loss")
    assert distortion_loss(g["la"], g["lb"]) == pytest.approx(float(g["loss20"]), rel=2e-6)
    assert distortion_loss(g["la21"], g["lb21"]) == pytest.approx(float(g["loss21"]), rel=2e-6)
    t = golden("rxtrace_awgn")
    l, s = find_loss(t["features_in"], t["features_out"].reshape(-1, 36))
    assert 0 < l < 2.0 and s % 12 in range(12)


def test_channel_tools_are_deterministic_and_normalised():
    from radae_amd.channel_tools import multipath_g, synth_features
    a, b = multipath_g("mpp", 8000, 16000, 3), multipath_g("mpp", 8000, 16000, 3)
    assert np.array_equal(a, b) and a.shape == (16000, 2) and a.dtype == np.complex64
    assert np.var(a[:, 0]) + np.var(a[:, 1]) == pytest.approx(1.0, rel=1e-3)       # hf_gain (multipath_samples.m:31)
    f = synth_features(5, 24)
    assert f.shape == (24, 36) and not f[:, 20:].any() and np.abs(f[:, 19]).max() <= 0.5


def test_blob_writer_reproduces_the_reference_format(tmp_path, lib):
    """dnnw.write_blob emits the record set / sizes of the reference's own blobs, and what it writes is read
    back identically by the Python reader and by the engine's C reader."""
    from radae_amd import dnnw, engine
    ref_path = os.path.join(REPO, "weights", "model05.bin")
    ref = dnnw.read_records(ref_path)
    m = dnnw.load_model(ref_path)
    out = str(tmp_path / "rt.bin")
    dnnw.write_blob(m, out)
    got = dnnw.read_records(out)
    assert list(got) == list(ref)                                            # same records, same order
    assert all(got[k].shape == ref[k].shape and got[k].dtype == ref[k].dtype for k in ref)
    assert os.path.getsize(out) == os.path.getsize(ref_path)
    for k in ("enc_dense1_weights_float", "dec_output_bias", "enc_gru3_input_weights_idx"):
        assert np.array_equal(got[k], ref[k])
    m2 = dnnw.load_model(out)                                                # re-quantisation error is within half a step
    for a, b in ((m.enc_conv[2].w, m2.enc_conv[2].w), (m.dec_gru[1].w_ih, m2.dec_gru[1].w_ih), (m.dec_glu[4].w, m2.dec_glu[4].w)):
        assert np.abs(a - b).max() <= 0.51 * np.abs(a).max() / 127
    blob = open(out, "rb").read()
    cm = Model()
    lib.rd_model_parse.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(Model)]
    assert lib.rd_model_parse(blob, len(blob), C.byref(cm)) == 0
    w = np.ctypeslib.as_array(cm.dec_gru[1].w_ih, shape=(288 * 224,)).reshape(288, 224)
    assert np.array_equal(w, m2.dec_gru[1].w_ih)


def test_wire_format_converters_match_reference_scripts(golden):
    """radae_amd.wire restates int16tof32.py / f32toint16.py; tests/golden/wire.npz holds the reference scripts' outputs."""
    from radae_amd import wire
    g = golden("wire")
    i16, f32 = g["i16"].tobytes(), g["f32"].tobytes()
    assert wire.int16_to_f32(i16) == g["i2f"].tobytes()
    assert wire.int16_to_f32(i16, zeropad=True) == g["i2f_zp"].tobytes()
    assert wire.f32_to_int16(f32) == g["f2i"].tobytes()
    assert wire.f32_to_int16(f32, real=True) == g["f2i_real"].tobytes()
    assert wire.f32_to_int16(f32, scale=8192.0) == g["f2i_scale"].tobytes()


def test_refine_moment_expansion_is_complex128_accurate():
    """The in-sync refine() of the HIP receiver evaluates the 20 or 21 fine frequencies (np.arange(fmax - 1, fmax + 1, 0.1)) from eight moments of
    the 160-sample window about the centre frequency and the centre of the window (rade_kernels.hip: refine_moments) instead of
    twenty direct sums (dsp.py:233-270, complex128 dots stored as complex64).  This pins the mathematics in NumPy: against a
    long-double evaluation the expansion is as accurate as the complex128 sum itself (a few 1e-16 of sum|y|), and every value rounded
    to complex64 -- what the reference stores and compares -- equals the direct sum's."""
    import math
    rng = np.random.default_rng(2)
    N = 160; n = np.arange(N)
    worst_t = worst_d = 0.0; differing = 0
    for trial in range(200):
        y = (rng.standard_normal(N) + 1j * rng.standard_normal(N)) * rng.uniform(0.01, 100)
        if trial % 3 == 0:                                      # a coherent component near the centre, as in sync
            y = y + 5 * np.exp(1j * rng.uniform(0, 6.28)) * np.exp(1j * 2 * np.pi * rng.uniform(-1, 1) / 8000 * n)
        fm = rng.uniform(-45, 45); fstart = fm - 1.0; fstep = 0.1
        nf = int(math.ceil((fm + 1.0 - fstart) / fstep)); delta = (fstart + fstep) - fstart
        assert nf in (20, 21)                                   # np.arange's length depends on how fm rounds: both occur
        w = 2 * np.pi * (fstart + np.arange(nf) * delta) / 8000.0
        exact = np.array([np.sum(y.astype(np.clongdouble) * np.exp(-1j * (np.longdouble(w[k]) * n.astype(np.longdouble)))) for k in range(nf)])
        direct = np.array([np.sum(y * np.exp(-1j * w[k] * n)) for k in range(nf)])
        wc = 0.5 * (w[0] + w[nf - 1]); nu = (n - 79.5) / 80.0
        mom = np.array([np.sum(nu ** m * (y * np.exp(-1j * wc * n))) for m in range(8)])
        out = np.empty(nf, np.complex128)
        for k in range(nf):
            dw = w[k] - wc; c = 1.0 + 0j; s = 0j
            for m in range(8):
                s += c * mom[m]; c = c * (-1j * dw * 80.0) / (m + 1)
            out[k] = s * np.exp(-1j * dw * 79.5)
        sy = np.sum(np.abs(y))
        worst_t = max(worst_t, float(np.abs(out - exact).max() / sy)); worst_d = max(worst_d, float(np.abs(direct - exact).max() / sy))
        differing += int(np.sum(out.astype(np.complex64) != direct.astype(np.complex64)))
    assert worst_t < 1e-15 and worst_t < 20 * worst_d, (worst_t, worst_d)
    assert differing == 0


def test_doppler_filter_design_against_scipy_firwin2():
    """the reference designs its Doppler-spread filter with Octave's fir2 (doppler_spread.m:27-29: `fir2(Ntaps-1, x/(lowFs/2), y)`), which
    cannot be run here.  scipy's firwin2 is an independent implementation of the same frequency-sampling algorithm: the module's restatement of fir2's recipe
    equals it to rounding for the channel presets' spreads (mpg, mpp, mpd, lmr60), and it is the design IN USE: doppler_plan (the device generator's taps) returns exactly these
    taps, and doppler_spread / multipath_g (every golden G, bench.py's workload) filter with them."""
    import math
    import scipy.signal as ss
    from radae_amd import channel_tools as ct
    for name, (spread, _) in ct.PRESETS.items():
        low_fs = math.ceil(10 * spread); m = 8000 / low_fs
        if m != math.floor(m):
            m = math.floor(m); low_fs = 8000 / m
        m = int(m)
        sigma = spread / 2.0
        x = np.arange(51) * low_fs / 100.0
        y = (1.0 / (sigma * math.sqrt(2 * math.pi))) * np.exp(-(x ** 2) / (2 * sigma * sigma))
        assert y[-1] < 1e-20                      # (firwin2 insists on exactly zero gain at Nyquist for an even number of taps)
        y[-1] = 0.0
        ref = ss.firwin2(100, x / (low_fs / 2.0), y, nfreqs=513, window="hamming")
        ours = ct.fir2_from_gaussian_psd(spread, low_fs, 100)
        assert np.abs(ours - ref).max() < 1e-12 * np.abs(ref).max(), name
        taps, ratio, n_low = ct.doppler_plan(spread, 8000, 16000)
        assert np.array_equal(taps, ours) and ratio == m
        # doppler_spread() filters with the same taps: rebuild its output from its own noise draw
        rng = np.random.default_rng(11); g = ct.doppler_spread(spread, 8000, 16000, rng)
        rng = np.random.default_rng(11); xs = rng.standard_normal(n_low + 100) + 1j * rng.standard_normal(n_low + 100)
        ylow = np.convolve(xs, ref)[: n_low + 100][100:]
        pos = np.arange(16000) / ratio; i0 = np.minimum(np.floor(pos).astype(np.int64), n_low - 2)
        want = ylow[i0] + (ylow[i0 + 1] - ylow[i0]) * (pos - i0)
        assert np.abs(g - want).max() < 1e-9 * np.abs(want).max(), name


def test_doppler_process_statistics():
    """multipath_samples.m:10-31 / doppler_spread.m:7-50: the filter's AMPLITUDE response is the Gaussian exp(-f^2 / (2 sigma^2)), sigma = spread / 2
    (doppler_spread.m:20-27), so the generated path gains have the power spectrum exp(-f^2 / sigma^2) and the autocorrelation R(tau) = exp(-(pi sigma tau)^2);
    hf_gain normalises var G1 + var G2 to 1."""
    from radae_amd import channel_tools as ct
    spread = ct.PRESETS["mpd"][0]; n = 8000 * 400                      # 400 s at 2 Hz spread: ~800 coherence times
    G = ct.multipath_g("mpd", 8000, n, 5).astype(np.complex128)
    assert np.var(G[:, 0]) + np.var(G[:, 1]) == pytest.approx(1.0, rel=1e-3)
    g = G[::80, 0]; g = g - g.mean()                                    # 100 Hz is plenty for a 2 Hz process
    for lag_s in (0.05, 0.1, 0.2, 0.3):
        k = int(round(lag_s * 100))
        r = np.real(np.vdot(g[:-k], g[k:]) / np.vdot(g, g))
        want = np.exp(-(np.pi * (spread / 2.0) * lag_s) ** 2)
        assert abs(r - want) < 0.03, (lag_s, r, want)


def test_lmr60_preset_and_rate_rs_h():
    """multipath_samples.m:17-21 (lmr60: fd = 450e6 * (60e3 / 3600 / 3e8) = 25 Hz, spread 2 fd, 200 us) and :33-40 (H from G): host restatement."""
    from radae_amd.channel_tools import PRESETS, doppler_plan, multipath_g, multipath_h
    spread, delay = PRESETS["lmr60"]
    assert abs(spread - 50.0) < 1e-9 and delay == 200e-6
    taps, ratio, n_low = doppler_plan(spread, 8000, 80000)
    # the preset's spread is 50.00000000000001 in IEEE doubles (the script's own expression, same in Octave): lowFs = ceil(10 * spread) = 501, M = floor(8000 / 501) = 15,
    # lowFs = 8000 / 15 (doppler_spread.m:12-19) -- not the 500 Hz / M = 16 of an exact 50
    assert spread > 50.0 and ratio == 15 and n_low == 5334 and len(taps) == 100
    H = multipath_h("lmr60", 8000, 2000, 1, 20000, 3)
    G = multipath_g("lmr60", 8000, 19999 * 4 + 1, 3)
    assert H.shape == (20000, 1) and H.dtype == np.float32
    assert np.abs(H[:, 0] - np.abs(G[::4, 0] + G[::4, 1])).max() < 1e-6  # Nc = 1: omega = 0, H = hf_gain |G1 + G2| at every M-th sample
    Hc = multipath_h("lmr60", 8000, 2000, 4, 500, 3, complex_=True)
    G4 = multipath_g("lmr60", 8000, 499 * 4 + 1, 3)                      # (hf_gain is the variance over the generated length: same length, same G)
    ph = np.exp(-2j * np.pi * np.arange(4) * 200e-6 * 2000)
    assert np.abs(Hc - (G4[::4, 0][:, None] + G4[::4, 1][:, None] * ph[None])).max() < 1e-6
    pav = np.mean(H.astype(np.float64) ** 2)
    lcr = np.sum((H[:-1, 0] ** 2 < 1.0) & (H[1:, 0] ** 2 > 1.0)) / 10.0      # the script's own check (:48-61)
    assert 0.8 < pav < 1.2 and abs(lcr - np.sqrt(2 * np.pi / pav) * 25.0 * np.exp(-1.0 / pav)) < 4.0


def test_sleeping_wait_estimate_recovers_from_an_outlier(lib):
    """rade_engine.c: sleep_until_event's estimate (it fed on its own nap -- est' = 0.9375 est per call after one outlier, ~12 s of cumulative oversleep after a
    1 s hiccup).  The update rule, driven with synthetic waits: steady 3 ms launches, one 1 s outlier, steady again -> the cumulative oversleep after the outlier stays
    below 40 ms and the estimate is back within 2x of the launch time in under 10 calls; a cold first wait does not seed it."""
    lib.rade_wait_model_step.restype = C.c_double
    lib.rade_wait_model_step.argtypes = [C.POINTER(C.c_double), C.c_double, C.c_int]
    est = C.c_double(0.0)
    lib.rade_wait_model_step(C.byref(est), 250000.0, 1)                      # first wait: 250 ms of lazy module load
    assert est.value == 0.0
    for _ in range(20):
        over = lib.rade_wait_model_step(C.byref(est), 3000.0, 0)
    assert 2500.0 < est.value < 3500.0 and over < 50.0
    lib.rade_wait_model_step(C.byref(est), 1e6, 0)                           # the hiccup
    assert est.value < 8.0 * 3500.0
    total, calls_to_recover = 0.0, None
    for k in range(60):
        total += lib.rade_wait_model_step(C.byref(est), 3000.0, 0)
        if calls_to_recover is None and est.value < 6000.0:
            calls_to_recover = k + 1
    assert total < 40000.0 and calls_to_recover is not None and calls_to_recover < 10, (total, calls_to_recover)
    assert 1400.0 < est.value < 3500.0


def test_loss_command_line(golden, tmp_path, capsys):
    """`python -m radae_amd.loss` = the reference's loss.py command line (:36-112): aligned loss, acquisition time, PASS / FAIL, --compare; on the awgn trace's features
    (the reference's own receiver output) the printed loss is find_loss's and the thresholds behave like the ctests use them (CMakeLists.txt:300-420)."""
    from radae_amd import loss
    g = golden("rxtrace_awgn")
    fi, fo = g["features_in"].astype(np.float32), g["features_out"].reshape(-1, 36).astype(np.float32)
    a, b = str(tmp_path / "features_in.f32"), str(tmp_path / "features_out.f32")
    fi.tofile(a); fo.tofile(b)
    want, start = loss.find_loss(fi, fo)
    assert loss.main([a, b, "--loss_test", "10.0", "--acq_time_test", "10.0"]) == 0
    out = capsys.readouterr().out
    assert f"  loss: {want:5.3f} start: {start:d} acq_time: {start * 0.01:5.2f} s" in out and out.strip().endswith("PASS")
    loss.main([a, b, "--loss_test", "1e-6"]); assert capsys.readouterr().out.strip().endswith("FAIL")
    if start > 0:
        loss.main([a, b, "--acq_time_test", str(start * 0.01 / 2)]); assert capsys.readouterr().out.strip().endswith("FAIL")
    loss.main([a, b, "--features_hat2", b, "--compare", "--clip_end", "12"]); out = capsys.readouterr().out
    assert "delta: 0.0" in out and out.strip().endswith("PASS")


def test_eoo_ber_tool(golden, tmp_path, capsys):
    """`python -m radae_amd.wire eoo_ber eoo_tx.f32 eoo_rx.f32` = the reference's eoo_ber.py on the five end-of-over frames the reference receiver decoded through MPP
    (rxtrace_eoo_mpp.npz; ctest radae_eoo_data_mpp: one frame below 5 % is a PASS)."""
    from radae_amd import wire
    g = golden("rxtrace_eoo_mpp")
    a, b = str(tmp_path / "eoo_tx.f32"), str(tmp_path / "eoo_rx.f32")
    g["tx_bits"].astype(np.float32).tofile(a); g["eoo_out"].astype(np.float32).tofile(b)
    bers, n_ok = wire.eoo_ber(g["tx_bits"], g["eoo_out"].ravel())
    assert np.allclose(bers, g["eoo_ber"]) and n_ok == int((g["eoo_ber"] < 0.05).sum()) >= 1
    assert wire.main(["eoo_ber", a, b]) == 0
    cap = capsys.readouterr()
    assert cap.out.count("frame received!") == 5 and "EOO frames  received: 5 n_ok_frames: 1" in cap.err and cap.err.strip().endswith("PASS")
