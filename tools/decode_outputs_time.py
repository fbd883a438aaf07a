import os, sys, torch
sys.path.insert(0, os.getcwd())
from dad_3dheads_amd import _lib, landmarks, synthetic
from dad_3dheads_amd.head_mesh import HeadMesh
st = synthetic.load_static()
hm = HeadMesh(flame_model=synthetic.synthetic_flame_model(0, st), landmarks=landmarks.canonical("445", st), static=st, device=0)
p = torch.from_numpy(synthetic.synthetic_params(64, seed=0)).cuda()
def t(**kw):
    out = {}
    for _ in range(300): hm.flame.decode(p, out=out, **kw)
    torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(2000): hm.flame.decode(p, out=out, **kw)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 2000 * 1000)
    return best
print("RD headline (v3d+proj2+lmk_px)", round(t(verts3d=True, proj=True, to_2d=True, landmarks_px=True), 2))
print("RD proj3 flip only", round(t(proj=True, to_2d=False, flip_z=True), 2))
print("RD proj3 only", round(t(proj=True, to_2d=False), 2))
print("RD proj2 only", round(t(proj=True, to_2d=True), 2))
print("RD v3d only", round(t(verts3d=True), 2))
print("RD v3d+proj3", round(t(verts3d=True, proj=True, to_2d=False), 2))
