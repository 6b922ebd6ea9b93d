import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from radae_amd.engine import BatchEngine as Engine
from radae_amd.channel_tools import synth_features
B, n_mf = 72, 84
dev = torch.device("cuda:0")
feats = torch.tensor(np.stack([synth_features(900 + b, 12 * (2 * n_mf + 3)) for b in range(B)]), device=dev)
cuts = ((0, n_mf), (n_mf, n_mf + 3), (n_mf + 3, 2 * n_mf + 3))
def run():
    eng = Engine(B, max_tx_mf=n_mf)
    out = [eng.tx(feats[:, 12 * a:12 * b].contiguous(), want_z=True) for a, b in cuts]
    eng.close()
    return out
frag = run()
os.environ["RADE_ENC_ROWS"] = "1"
rows = run()
for i, ((iq_f, z_f), (iq_r, z_r)) in enumerate(zip(frag, rows)):
    d = (z_f - z_r).abs()
    print("call", i, "z max", z_r.abs().max().item(), "diff max", d.max().item(), "n diff", (d > 0).sum().item(), "of", d.numel())
    dt = d.amax(dim=(0, 2)).cpu().numpy()
    print("  per step first 12:", np.array2string(dt[:12], precision=3), " argmax step", int(dt.argmax()))
    db = d.amax(dim=(1, 2)).cpu().numpy()
    print("  per stream first 6:", np.array2string(db[:6], precision=3))
    dc = d.amax(dim=(0, 1)).cpu().numpy()
    print("  per channel:", np.array2string(dc[:16], precision=3))
    print("  iq diff", (iq_f - iq_r).abs().max().item())
