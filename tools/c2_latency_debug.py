import sys, time, numpy as np
sys.path.insert(0, '.')
from radae_amd import core
from radae_amd.channel_tools import synth_features
from radae_amd.engine import DEFAULT_BLOB
T=1008; n=T//4
f = synth_features(1000, T)
rows = np.concatenate([f[:, :20], -np.ones((T, 1), np.float32)], axis=1).reshape(n, 84).astype(np.float32)
enc = core.CoreEncoder(DEFAULT_BLOB); dec = core.CoreDecoder(DEFAULT_BLOB)
for i in range(32): dec.step(enc.step(rows[i]))
for rep in range(3):
    enc.reset(); dec.reset()
    te=np.zeros(n); td=np.zeros(n)
    for i in range(n):
        a=time.perf_counter(); z=enc.step(rows[i]); b=time.perf_counter(); dec.step(z); c=time.perf_counter(); te[i]=b-a; td[i]=c-b
    print(rep, 'enc top', [(int(i), round(1e3*te[i],3)) for i in np.argsort(te)[-3:]], 'dec top', [(int(i), round(1e3*td[i],3)) for i in np.argsort(td)[-3:]], 'medians', round(1e3*np.median(te),4), round(1e3*np.median(td),4), 'mean step ms', round(1e3*(te.sum()+td.sum())/n,4))
