import numpy as np, torch, sys
sys.path.insert(0,'.')
from oracle import nerf_oracle as orc
from nerf_amd import ops
from nerf_amd.run_nerf_helpers import init_nerf_model
from tests.util import synthetic_rays, trained_like_weights
g = torch.device('cuda')
rng = np.random.default_rng(5)
wc, wf = trained_like_weights(rng), trained_like_weights(rng)
m = init_nerf_model(8,256,63,27,4,[4],True); m.set_weights(wc)
n=256
rb = synthetic_rays(n, seed=11)
rays = torch.tensor(rb, device=g)
for trial in range(3):
    zc = ops.sample_z(rays, 64, False)
    raw = ops.mlp_forward_rays(m.packed(), rays, zc)
    out = ops.composite_fwd(raw, zc, rays[:,3:6], 11, True)
    z = zc.cpu().numpy().astype(np.float64); r=rb.astype(np.float64)
    pts = r[:,None,0:3]+r[:,None,3:6]*z[:,:,None]
    ref = orc.run_network(pts, r[:,8:11], wc, dtype=np.float64)
    rawn = raw.cpu().numpy()
    sig = lambda v: 1/(1+np.exp(-v))
    e_rgb = np.abs(sig(rawn[...,:3])-sig(ref[...,:3])).max(-1)
    e_s = np.abs(rawn[...,3]-ref[...,3])
    print('trial',trial,'max rgb err',e_rgb.max(),'max sigma err',e_s.max(),'sigma range',ref[...,3].min(),ref[...,3].max())
    refc = orc.raw2outputs(ref, z, r[:,3:6], 0., True, dtype=np.float64)
    e = np.abs(out[0].cpu().numpy()-refc[0]).max(-1)
    bad = np.argsort(-e)[:3]
    print(' composite rgb err worst rays', bad, e[bad])
    # composite of gpu raw in oracle
    refg = orc.raw2outputs(rawn.astype(np.float64), z, r[:,3:6], 0., True, dtype=np.float64)
    print(' gpu-composite vs oracle-composite-of-gpu-raw', np.abs(out[0].cpu().numpy()-refg[0]).max())
    b=bad[0]; print(' ray',b,'norm d',np.linalg.norm(r[b,3:6]),'sigma ref',ref[b,:,3][:8],'gpu',rawn[b,:,3][:8]); print(' weights ref', refc[3][b][:10])
