import sys, os, numpy as np, torch
R = os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
os.chdir(R)
import oryon_amd; oryon_amd.configure()
import test_gpu_pointdsc as T
for name in T.names("g4_pointdsc_"):
    g = T.load(name); m = T.build(g)
    src, tgt, nn_, n = T.padded(g)
    feat, conf = m.encode(src, tgt, nn_)
    feat, conf = feat[0, :n].cpu().numpy(), conf[0, :n].cpu().numpy()
    print(name, "feat err/scale %.2e" % (np.abs(feat - g["feat"]).max() / np.abs(g["feat"]).max()), "conf err %.2e" % (np.abs(conf - g["confidence"]).max() / max(1.0, np.abs(g["confidence"]).max())))
