import os, sys, faulthandler
faulthandler.enable()
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import e_alphazero_amd as eaz
from e_alphazero_amd import engine
engine.tune(persist=0, shards=2, fc_tile=int(os.environ.get("TILE", "0")), fused_leaf=int(os.environ.get("FUSED", "1")),
            tree_block=int(os.environ.get("TBLOCK", "64")), tree_spread=int(os.environ.get("TSPREAD", "0")))
G, B, S, T = int(os.environ.get("G", "8")), 256, int(os.environ.get("S", "16")), 3
am = (np.random.default_rng(0).random((G, G)) < 0.5).astype(np.uint8)
params = eaz.params.init_fc_params(G * G, 256, 2, seed=0)
env = engine.EnvHandle.deepsea(G, am, device=0)
net = engine.FCNet(env.ctx, G * G, 256, 2, params, eaz.params.init_hash_state())
search = engine.Search(env, net, B, S)
st = env.init(B, key=1)
print("eager...", flush=True)
o1 = search.selfplay(st.clone(), T, key=7, use_graph=False); torch.cuda.synchronize()
print("eager ok; graph...", flush=True)
o2 = search.selfplay(st.clone(), T, key=7, use_graph=True); torch.cuda.synchronize()
print("graph ok", all(torch.equal(o1[k], o2[k]) for k in o1), flush=True)
