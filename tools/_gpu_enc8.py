import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from radae_amd.engine import BatchEngine as Engine
from radae_amd.channel_tools import synth_features
B = 5632
dev = torch.device("cuda:0")
base = np.stack([synth_features(40 + b, 12 * 5) for b in range(64)])
feats = torch.tensor(np.tile(base, (B // 64, 1, 1)) * (1.0 + 0.001 * (np.arange(B) // 64))[:, None, None].astype(np.float32), device=dev)
def run():
    eng = Engine(B, max_tx_mf=1)
    out = [eng.tx(feats[:, 12 * k:12 * k + 12].contiguous(), want_z=True) for k in range(5)]
    eng.close()
    return out
frag = run()
os.environ["RADE_ENC_ROWS"] = "1"
rows = run()
for i, ((iq_f, z_f), (iq_r, z_r)) in enumerate(zip(frag, rows)):
    d = (z_f - z_r).abs()
    print("call", i, "z max", z_r.abs().max().item(), "diff max", d.max().item(), "n diff", (d > 0).sum().item(), "of", d.numel())
    print("  per step:", d.amax(dim=(0, 2)).cpu().numpy())
    db = d.amax(dim=(1, 2)).cpu().numpy()
    print("  streams with diff:", int((db > 0).sum()), "first few", np.nonzero(db)[0][:10])
    print("  per channel:", np.array2string(d.amax(dim=(0, 1)).cpu().numpy()[:12], precision=3))
