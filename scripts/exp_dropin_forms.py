import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from torch.utils.data import DataLoader
import test_gpu_dropin_r4 as t
from hypad_amd import train as ht, _C
for (S,B,hyper,n) in [(100,64,True,1916),(100,64,False,256),(150,256,True,2*256+9)]:
    ds = t.Windows(n, S); loader = DataLoader(ds, batch_size=B, drop_last=True, shuffle=True, num_workers=0)
    runs = {}
    for form in ("epoch","epoch_pi","call"):
        mods = t.build(S, hyper, 5)
        np.random.seed(21); torch.manual_seed(21)
        P = t.P_(B,S,hyper, per_iteration=(form=="call"), stage_samples=False)
        if form=="epoch_pi": P.epoch_flags = _C.EPOCH_PER_ITERATION
        import io, contextlib
        with contextlib.redirect_stdout(io.StringIO()):
            hist = ht.train_tadgan(loader, *mods, n_epochs=2, params=P, path="/tmp")
        torch.cuda.synchronize()
        runs[form] = (hist, t.weights(mods))
    for a,b in (("epoch","epoch_pi"),("epoch_pi","call"),("epoch","call")):
        mx = max(float((wa[k]-wb[k]).abs().max()) for wa,wb in zip(runs[a][1],runs[b][1]) for k in wa)
        hd = max(abs(x-y) for nm in ("cx","cz","dec") for x,y in zip(getattr(runs[a][0],nm),getattr(runs[b][0],nm)))
        rel = max(float((wa[k]-wb[k]).norm() / wb[k].norm().clamp_min(1e-12)) for wa,wb in zip(runs[a][1],runs[b][1]) for k in wa if "weight_hh" not in k)
        worst = max(((float((wa[k]-wb[k]).norm() / wb[k].norm().clamp_min(1e-12)), k) for wa,wb in zip(runs[a][1],runs[b][1]) for k in wa if "weight_hh" not in k))
        print(S,B,hyper,n,a,b,"max weight diff",mx,"max hist diff",hd, "max rel L2", rel, worst[1])
