import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, hf_amd, common
if os.environ.get('HF_LIB'):
    from hf_amd import _capi, build
    build.LIB_PATH = os.environ['HF_LIB']; _capi._build.LIB_PATH = os.environ['HF_LIB']
from oracle import hf_oracle as O
W,H,kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
rng = np.random.default_rng(W * 1000 + H)
h = common.heights(kind, W, H, rng)
fo = O.OracleField(h, max_height=0.5); fg = hf_amd.Heightfield(heightfield=torch.from_numpy(h), max_height=0.5)
r = np.concatenate([common.random_rays(4000, rng), common.inside_rays(2000, rng)], 1)
xs = np.array([fo.vertex(0, j)[0] for j in range(W)]); ys = np.array([fo.vertex(i, 0)[1] for i in range(H)])
r = np.concatenate([r, common.structured_rays(xs, ys)],1)
t,u,v,prim = fo.ray_intersect_preliminary(r, naive=True)
rt = torch.from_numpy(r).cuda()
pi = fg.ray_intersect_preliminary(hf_amd.Ray3f(rt[0:3].contiguous(), rt[3:6].contiguous(), rt[6].contiguous()))
tg, pg = pi.t.cpu().numpy(), pi.prim_index.cpu().numpy().view(np.uint32)
bad = np.where((t != tg) | (prim != pg))[0]
print('n', r.shape[1], 'bad', len(bad))
for i in bad[:12]:
    print(i, 'ray', r[:,i], 'oracle t %g prim %d'%(t[i],prim[i]), 'gpu t %g prim %d'%(tg[i],pg[i]))
