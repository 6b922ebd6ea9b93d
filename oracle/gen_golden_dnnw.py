#!/usr/bin/env python3
"""Pin the DNNw weight blob against the reference's OWN exporter (this container only).

TEST INFRASTRUCTURE -- never imported by the product; /root/reference does not exist on the GPU box, only the fixtures written here travel.

What runs, and whose code it is:
  1. oracle/dnnw_synth.py makes two seeded checkpoints of the model19 architecture (torch orientation): A "lossless" (int8-exact rows, power-of-two
     row scales, genuinely sparse GRU input blocks) and B "generic" (Gaussian floats, same sparsity).  They are loaded into the reference's
     `RADAE(21, 80, EbNodB=100)` exactly as `/root/reference/export_rade_weights.py:208-212` builds it, through `state_dict` keys only.
  2. THE REFERENCE'S EXPORTER: `export_rade_weights.py` is executed with runpy (its argparse runs at import) and its `c_export(args, model)`
     (:54-172) is called on that model; it drives `wexchange.torch.dump_torch_weights` -> `wexchange/c_export/common.py` (`print_gru_layer` :346-382,
     `print_dense_layer` :279-294, `print_conv1d_layer` :297-321, `print_linear_layer` :200-277, `print_sparse_weight` :140-176, `compute_scaling` :180-194,
     `quantize_weight` :132-137) and writes rade_enc_data.c / rade_dec_data.c into a scratch directory.  (The script's own `__main__` block cannot run
     on this torch: `load_state_dict(..., weights_only=True)` at :212 is a TypeError in torch 2.10, so steps :208-220 are restated below.)
  3. `common.print_linear_layer` is wrapped to RECORD what the exporter computed per layer -- the matrix it was handed (after its gate swap / transposes),
     `compute_scaling`'s scale, `quantize_weight`'s q, subias -- by calling the reference's own functions on the reference's own arguments.
  4. The emitted C is parsed (array initialisers, the `radeenc_arrays[]` / `radedec_arrays[]` tables, `#ifndef DISABLE_DEBUG_FLOAT` sections dropped as in the build that
     produced bin/model19_check3.bin: that blob has no `*_weights_float` for quantised layers) and packed record by record as
     `/root/reference/src/write_rade_weights.c:51-74` does (64-byte WeightHead, payload, zero padding to a multiple of 64; encoder table then decoder table).
     Decimal literals become floats the way a C compiler reads them (decimal -> double -> float).
  5. Reference-module outputs on checkpoint A: `CoreEncoderStatefull` / `CoreDecoderStatefull` (radae_base.py:223-286, :358-430; `n()` clamp-only) stepped one
     40 ms step per call on seeded features / latents -- the float modules holding the checkpoint itself, no de-quantisation of mine involved.

Fixtures: tests/golden/dnnw_export_A.bin (the reference-exported blob of checkpoint A, 2.4 MB), tests/golden/dnnw_export.npz (per-layer exporter records for A,
sha256 + per-record digests of the exported blob of B, module outputs for A), tests/golden/weights_check.npz (per-tensor statistics of CHECKPOINT A itself:
size, sum, sum of magnitudes, first 8 and last 4 values -- what a correct reader returns for the lossless blob; rounds 1-5 wrote this file from radae_amd/dnnw.py's own reading of model19_check3.bin).

Run:  python3 oracle/gen_golden_dnnw.py     (~1 min)
"""
import hashlib
import os
import re
import runpy
import struct
import sys
import tempfile

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(REF, "weight-exchange"))
sys.path.insert(0, REPO)

import numpy as np
import torch

torch.set_num_threads(1)
os.chdir(REF)

import radae.radae_base as rb

rb.n = lambda x: torch.clamp(x, min=-1.0, max=1.0)
from radae import RADAE  # noqa: E402
import wexchange.c_export.common as wx_common  # noqa: E402

from oracle import dnnw_synth  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")
SEED_A, SEED_B = 19001, 19002


def load_checkpoint_into(model, m):
    """synthetic tensors -> the reference's RADAE module (the role of torch.load + load_state_dict, export_rade_weights.py:208-212)."""
    sd = model.state_dict()
    T = torch.tensor
    new = {}
    for side, enc in (("encoder", True), ("decoder", False)):
        p = f"core_{side}.module."
        new[p + "dense_1.weight"] = T((m.enc_dense1 if enc else m.dec_dense1).w); new[p + "dense_1.bias"] = T((m.enc_dense1 if enc else m.dec_dense1).b)
        last = "z_dense" if enc else "output"
        L = m.enc_zdense if enc else m.dec_output
        new[p + last + ".weight"] = T(L.w); new[p + last + ".bias"] = T(L.b)
        for i in range(5):
            G = (m.enc_gru if enc else m.dec_gru)[i]; Cv = (m.enc_conv if enc else m.dec_conv)[i]
            new[p + f"gru{i+1}.weight_ih_l0"] = T(G.w_ih); new[p + f"gru{i+1}.weight_hh_l0"] = T(G.w_hh)
            new[p + f"gru{i+1}.bias_ih_l0"] = T(G.b_ih); new[p + f"gru{i+1}.bias_hh_l0"] = T(G.b_hh)
            new[p + f"conv{i+1}.conv.weight"] = T(Cv.w); new[p + f"conv{i+1}.conv.bias"] = T(Cv.b)
            if not enc:
                W = T(m.dec_glu[i].w)
                new[p + f"glu{i+1}.gate.parametrizations.weight.original1"] = W
                # g = ||v|| as the parametrisation's own kernel computes it, so that g / ||v|| == 1.0 exactly and gate.weight == W bit for bit
                # (a checkpoint whose weight-normalised layer holds exactly the numbers the test regenerates from the seed); that kernel's
                # summation order is not norm_except_dim's, so each row's g is searched within a few float32 steps of it
                g = torch.norm_except_dim(W, 2, 0).clone()
                for step in (0, 1, -1, 2, -2, 3, -3, 4, -4):
                    gt = torch.tensor(np.frombuffer((g.numpy().view(np.int32) + step).tobytes(), dtype=np.float32).reshape(g.shape).copy())
                    ok = (torch._weight_norm(W, gt, 0) == W).all(dim=1)
                    if step == 0:
                        best, done = g.clone(), ok.clone()
                    best[ok & ~done] = gt[ok & ~done]; done |= ok
                assert bool(done.all()), "no g reproduces the GLU weight exactly"
                new[p + f"glu{i+1}.gate.parametrizations.weight.original0"] = best
    for k, v in new.items():
        assert k in sd and sd[k].shape == v.shape, k
    missing = model.load_state_dict(new, strict=False)
    assert not missing.unexpected_keys
    own = [k for k in missing.missing_keys if k.startswith("core_encoder.") or k.startswith("core_decoder.")]
    assert not own, own
    for i in range(5):
        assert torch.equal(model.core_decoder.module.get_submodule(f"glu{i+1}.gate").weight, T(m.dec_glu[i].w))


def run_reference_exporter(m, outdir):
    """Steps 2 + 3 of the module docstring.  Returns the per-layer records of the exporter."""
    argv = sys.argv
    sys.argv = ["export_rade_weights.py", "synthetic_checkpoint", outdir]
    try:
        script = runpy.run_path(os.path.join(REF, "export_rade_weights.py"), run_name="export_rade_weights")
    finally:
        sys.argv = argv
    args = script["args"]
    model = RADAE(21, args.latent_dim, EbNodB=100)                      # export_rade_weights.py:211
    load_checkpoint_into(model, m)

    def _remove_weight_norm(mod):                                       # :214-219 (a no-op for parametrised modules, as in the reference)
        try:
            torch.nn.utils.remove_weight_norm(mod)
        except ValueError:
            return
    model.apply(_remove_weight_norm)

    records = {}
    orig = wx_common.print_linear_layer

    def recording(writer, name, weight, bias, scale=None, sparse=False, diagonal=False, quantize=True):
        rec = {"weight": np.array(weight, copy=True), "bias": None if bias is None else np.array(bias, copy=True), "sparse": sparse, "quantize": quantize}
        if quantize:
            s = wx_common.compute_scaling(weight, quantize) if scale is None else scale
            q = wx_common.quantize_weight(weight, s)
            rec["scale"] = np.array(s, copy=True); rec["q"] = q
            rec["subias"] = (np.zeros(weight.shape[1]) if bias is None else bias) - np.sum(q * s, axis=0)      # common.py:264
        records[name] = rec
        return orig(writer, name, weight, bias, scale=scale, sparse=sparse, diagonal=diagonal, quantize=quantize)

    wx_common.print_linear_layer = recording
    os.makedirs(outdir, exist_ok=True)
    try:
        script["c_export"](args, model)                                 # :54-172
    finally:
        wx_common.print_linear_layer = orig
    return records, model


_CTYPE = {"float": ("<f4", 0), "int": ("<i4", 1), "opus_int8": ("i1", 3)}          # Opus dnn/nnet.h WEIGHT_TYPE_float / _int / _int8


def parse_c_arrays(path):
    """The arrays a build with DISABLE_DEBUG_FLOAT defined and USE_WEIGHTS_FILE undefined compiles in, in the order of the `*_arrays[]` table."""
    src = open(path).read()
    src = re.sub(r"#ifndef DISABLE_DEBUG_FLOAT\n.*?#endif /\*DISABLE_DEBUG_FLOAT\*/\n", "", src, flags=re.S)
    arrays = {}
    for mt in re.finditer(r"static const (\w+) (\w+)\[(\d+)\] = \{(.*?)\};", src, flags=re.S):
        ctype, name, n, body = mt.group(1), mt.group(2), int(mt.group(3)), mt.group(4)
        toks = body.replace("\n", " ").split(",")
        assert len(toks) == n, name
        if ctype == "float":
            a = np.array([float(t) for t in toks], dtype=np.float64).astype(np.float32)        # decimal -> double -> float, as a C compiler
        else:
            a = np.array([int(t) for t in toks], dtype=np.int64).astype(_CTYPE[ctype][0])
        arrays[name] = (ctype, a)
    table = re.search(r"const WeightArray \w+_arrays\[\] = \{(.*?)\{NULL, 0, 0, NULL\}", src, flags=re.S).group(1)
    order = re.findall(r'\{"(\w+)",\s+WEIGHTS_\w+_TYPE', table)
    return [(nm, ) + arrays[nm] for nm in order if nm in arrays]           # `#ifdef WEIGHTS_<name>_DEFINED`: only compiled-in arrays


def pack_blob(entries):
    """write_rade_weights.c:51-74."""
    out = []
    for name, ctype, a in entries:
        payload = a.astype(_CTYPE[ctype][0]).tobytes()
        block = (len(payload) + 63) // 64 * 64
        assert len(name) < 43
        out.append(struct.pack("<4siiii", b"DNNw", 0, _CTYPE[ctype][1], len(payload), block) + name.encode().ljust(44, b"\0"))
        out.append(payload + b"\0" * (block - len(payload)))
    return b"".join(out)


def export(m, tag):
    with tempfile.TemporaryDirectory() as d:
        records, model = run_reference_exporter(m, d)
        entries = parse_c_arrays(os.path.join(d, "rade_enc_data.c")) + parse_c_arrays(os.path.join(d, "rade_dec_data.c"))
    blob = pack_blob(entries)
    print(f"checkpoint {tag}: {len(entries)} records, {len(blob)} bytes, sha256 {hashlib.sha256(blob).hexdigest()[:16]}", file=sys.stderr)
    return records, model, entries, blob


def stats(a):
    a = np.asarray(a, dtype=np.float64).ravel()
    return np.array([a.size, a.sum(), np.abs(a).sum()] + list(a[:8]) + list(a[-4:]))


def main():
    fx = {}
    # ---- checkpoint A: lossless ----
    mA = dnnw_synth.synth_model(SEED_A, lossless=True)
    recA, modelA, entA, blobA = export(mA, "A")
    open(os.path.join(OUT, "dnnw_export_A.bin"), "wb").write(blobA)
    fx["A_sha256"] = np.array(hashlib.sha256(blobA).hexdigest())
    fx["A_record_names"] = np.array([e[0] for e in entA])
    for name, r in recA.items():
        if not r["quantize"]:
            continue
        w, q, s = r["weight"], r["q"], r["scale"]
        assert np.array_equal(q.astype(np.float32) * s[None, :].astype(np.float32), w), f"{name}: checkpoint A is not lossless under the exporter"
        fx[f"A/{name}/scale"] = s.astype(np.float32)                            # compute_scaling's value (n_out,), exporter column order (gates z,r,n)
        fx[f"A/{name}/subias"] = r["subias"].astype(np.float64)
        fx[f"A/{name}/q_sha256"] = np.array(hashlib.sha256(q.astype(np.int8).tobytes()).hexdigest())      # (n_in, n_out) row-major, exporter orientation
        fx[f"A/{name}/q_head"] = q[:8, :16].astype(np.int8)
        if r["sparse"]:
            blocks = np.abs(w).reshape(w.shape[0] // 4, 4, w.shape[1] // 8, 8).sum(axis=(1, 3)) > 1e-10
            fx[f"A/{name}/kept_blocks"] = np.packbits(blocks.T)                 # [out group][in block]
            fx[f"A/{name}/n_kept"] = np.int64(blocks.sum())
            assert blocks.sum() < blocks.size, "the GRU input must be genuinely sparse"
    # per-tensor statistics of the CHECKPOINT (what a correct reader returns for a lossless blob): weights_check's format
    tA = dnnw_synth.tensors(mA)
    np.savez(os.path.join(OUT, "weights_check.npz"), **{k: stats(v.transpose(0, 2, 1) if v.ndim == 3 else v) for k, v in tA.items()})   # conv as [out][tap][in]: the C layout
    # ---- reference-module outputs on checkpoint A ----
    modelA.core_encoder_statefull_load_state_dict()
    modelA.core_decoder_statefull_load_state_dict()
    rng = np.random.Generator(np.random.PCG64(77))
    T = 24
    feats = np.clip(0.6 * rng.standard_normal((T, 84)), -2, 2).astype(np.float32)
    feats[:, 20::21] = -1.0                                                     # aux symbol (radae_txe.py:117)
    zin = (0.8 * rng.standard_normal((T, 80))).astype(np.float32)
    enc, dec = modelA.core_encoder_statefull, modelA.core_decoder_statefull
    enc_m = enc.module if hasattr(enc, "module") else enc
    assert enc_m.bottleneck == 1                                                  # RADAE's default, as export_rade_weights.py:211 builds it: z = tanh(z_dense(.))
    with torch.no_grad():
        z_b1 = np.stack([enc(torch.tensor(feats[t].reshape(1, 4, 21))).numpy().reshape(80) for t in range(T)])
        for g in (enc_m.gru1, enc_m.gru2, enc_m.gru3, enc_m.gru4, enc_m.gru5, enc_m.conv1, enc_m.conv2, enc_m.conv3, enc_m.conv4, enc_m.conv5):
            g.reset()
        enc_m.bottleneck = 3; modelA.core_encoder.module.bottleneck = 3           # model19_check3's setting (radae_txe.py:57-63): linear z (radae_base.py:281-284)
        z = np.stack([enc(torch.tensor(feats[t].reshape(1, 4, 21))).numpy().reshape(80) for t in range(T)])
        fo = np.stack([dec(torch.tensor(zin[t].reshape(1, 1, 80))).numpy().reshape(84) for t in range(T)])
        # the stateless modules over the whole sequence (CoreEncoder / CoreDecoder, radae_base.py:157-221, :291-356) as a cross-check of the stepping
        z_sl = modelA.core_encoder.module(torch.tensor(feats.reshape(1, 4 * T, 21))).numpy().reshape(T, 80)
        f_sl = modelA.core_decoder.module(torch.tensor(zin.reshape(1, T, 80))).numpy().reshape(T, 84)
    print(f"stateful vs stateless: enc {np.abs(z - z_sl).max():.2e} dec {np.abs(fo - f_sl).max():.2e}; |z| rms {np.sqrt((z**2).mean()):.3f} |f| rms {np.sqrt((fo**2).mean()):.3f}", file=sys.stderr)
    assert np.abs(np.tanh(z) - z_b1).max() < 1e-6
    fx["A/run/features"] = feats; fx["A/run/z"] = z.astype(np.float32); fx["A/run/z_bottleneck1"] = z_b1.astype(np.float32)
    fx["A/run/z_hat"] = zin; fx["A/run/features_out"] = fo.astype(np.float32)
    # ---- checkpoint B: generic floats -> digests of the reference-exported blob (the writer's byte-identity target) ----
    mB = dnnw_synth.synth_model(SEED_B, lossless=False)
    recB, _, entB, blobB = export(mB, "B")
    fx["B_sha256"] = np.array(hashlib.sha256(blobB).hexdigest())
    fx["B_record_names"] = np.array([e[0] for e in entB])
    fx["B_record_sha256"] = np.array([hashlib.sha256(e[2].tobytes()).hexdigest()[:16] for e in entB])
    for name in ("enc_gru3_input", "dec_conv2", "dec_glu4", "dec_gru1_recurrent"):
        r = recB[name]
        fx[f"B/{name}/scale"] = r["scale"].astype(np.float32); fx[f"B/{name}/subias"] = r["subias"].astype(np.float64)
        fx[f"B/{name}/q_head"] = r["q"][:8, :16].astype(np.int8)
    fx["seeds"] = np.array([SEED_A, SEED_B])
    np.savez_compressed(os.path.join(OUT, "dnnw_export.npz"), **fx)
    print("dnnw_export ok")


if __name__ == "__main__":
    main()
