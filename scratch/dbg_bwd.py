import numpy as np, torch, sys
sys.path.insert(0,'.')
from tests.test_gpu_train import _mlp_grads_reference, _t
from tests.util import synthetic_rays, trained_like_weights
from nerf_amd import ops
from nerf_amd.run_nerf_helpers import init_nerf_model
gpu=torch.device('cuda')
n,S=8,64
rng = np.random.default_rng(100 + n)
ws = trained_like_weights(rng)
m = init_nerf_model(8, 256, 63, 27, 4, [4], True, device=gpu); m.set_weights(ws)
rb = synthetic_rays(n, seed=n); rays=_t(rb,gpu); z=ops.sample_z(rays,S)
d_raw = (rng.normal(size=(n, S, 4)) * 1e-4).astype(np.float32)
act = torch.zeros((ops.act_bytes(n * S),), dtype=torch.uint8, device=gpu)
ops.mlp_forward_train(m.packed(), rays, z, act)
pb = torch.empty((ops.packed_bwd_bytes(),), dtype=torch.uint8, device=gpu)
ops.mlp_pack_bwd([p.detach() for p in m.params], pb)
gbuf = torch.zeros((ops.grad_bytes(n * S),), dtype=torch.uint8, device=gpu)
part = torch.empty((ops.partial_bytes(),), dtype=torch.uint8, device=gpu)
grads = [torch.full_like(p, 7.0) for p in m.params]
for gs in (4096.0, 65536.0):
    ops.mlp_backward(pb, act, _t(d_raw, gpu), gs, gbuf, part, grads)
    ref,_ = _mlp_grads_reference(ws, rb, z.cpu().numpy(), d_raw)
    print('gscale',gs)
    for i,(g,r) in enumerate(zip(grads,ref)):
        g=g.cpu().numpy().astype(np.float64)
        print(i, r.shape, 'rel %.3e'%(np.linalg.norm(g-r)/np.linalg.norm(r)), 'max|r| %.2e'%np.abs(r).max())
# W0 per input row error
g=grads[0].cpu().numpy().astype(np.float64); r=ref[0]
print('W0 row rel:', np.round(np.linalg.norm(g-r,axis=1)/np.linalg.norm(r,axis=1),3))

# ---- compare the chain's saved d v_pre / d bott / d pre7 with fp64 autograd, wave tile 0
W=[torch.tensor(w,dtype=torch.float64) for w in ws]
r=torch.tensor(rb,dtype=torch.float64); zz=z.cpu().double()
pts=(r[:,None,0:3]+r[:,None,3:6]*zz[:,:,None]).reshape(-1,3); dirs=r[:,None,8:11].expand(n,S,3).reshape(-1,3)
from oracle import torch_restatement as trs
e=torch.cat([trs.embed(pts,10),trs.embed(dirs,4)],-1)
p_,v_=e[:,:63],e[:,63:]
h=p_; pres=[]
for i in range(8):
    if i==5: h=torch.cat([p_,h],-1)
    pre=(h@W[2*i]+W[2*i+1]).requires_grad_(True); pres.append(pre); h=torch.relu(pre)
alpha=h@W[22]+W[23]; bott=(h@W[16]+W[17]).requires_grad_(True)
prev=(torch.cat([bott,v_],-1)@W[18]+W[19]).requires_grad_(True); v=torch.relu(prev)
rgb=v@W[20]+W[21]
raw=torch.cat([rgb,alpha],-1)
# manual chain in fp64
dr=torch.tensor(d_raw,dtype=torch.float64).reshape(-1,4)
dv=(dr[:,:3]@W[20].T)*(prev>0)
dbott=dv@W[18][:256].T
dh7=dbott@W[16].T+dr[:,3:4]@W[22].T
dpre7=dh7*(pres[7]>0)
gs=65536.0
gb=gbuf.view(-1,154,1024)[0]
def decode(blk0,nk):
    t=gb[blk0:blk0+nk].contiguous().view(torch.float16).view(nk,64,8).double().cpu().numpy()
    out=np.zeros((32,nk*16))
    for k in range(nk):
        for lane in range(64):
            c,hh=lane&31,lane>>5
            for j in range(8):
                f=32*(k>>1)+16*(k&1)+8*(j>>2)+4*hh+(j&3)
                out[c,f]=t[k,lane,j]
    return out
for name,blk0,nk,refv in [('dv',0,8,dv),('dbott',8,16,dbott),('dpre7',24+112,16,dpre7)]:
    got=decode(blk0,nk)/gs; rf=refv[:32].numpy()
    print(name,'rel',np.linalg.norm(got-rf)/np.linalg.norm(rf),'nonzero mismatch',np.sum((got!=0)!=(rf!=0)),'of',rf.size)
    bad=np.argwhere(np.abs(got-rf)>0.05*np.abs(rf).max())[:8]
    for (pp,ff) in bad: print('   pt',pp,'feat',ff,'got',got[pp,ff],'ref',rf[pp,ff], 'prev' if name=='dv' else '', float(prev[pp,ff]) if name=='dv' else '')

# ---- dW kernel in isolation: recompute dW7 = h6^T dpre7 from the GPU's own buffers
def decode_all(buf, nblk_total, blk0, nk):
    b = buf.view(-1, nblk_total, 1024)[:, blk0:blk0+nk].contiguous().view(torch.float16).view(-1, nk, 64, 8).double().cpu().numpy()
    T = b.shape[0]
    out = np.zeros((T*32, nk*16))
    k = np.arange(nk)[:,None,None]; lane=np.arange(64)[None,:,None]; j=np.arange(8)[None,None,:]
    f = 32*(k>>1)+16*(k&1)+8*(j>>2)+4*(lane>>5)+(j&3)
    c = np.broadcast_to(lane&31, f.shape)
    for t in range(T):
        out[t*32 + c, f] = b[t]
    return out
h6 = decode_all(act, 158, 4+16*6, 16); dp7 = decode_all(gbuf, 154, 24+112, 16)
dW7 = h6.T @ dp7 / gs
g14 = grads[14].cpu().numpy().astype(np.float64)
print('dW7 gpu vs from-buffers rel', np.linalg.norm(g14-dW7)/np.linalg.norm(dW7), ' from-buffers vs autograd', np.linalg.norm(dW7-ref[14])/np.linalg.norm(ref[14]))
db7 = dp7.sum(0)/gs
print('db7 gpu vs from-buffers rel', np.linalg.norm(grads[15].cpu().numpy()-db7)/np.linalg.norm(db7), ' vs autograd', np.linalg.norm(db7-ref[15])/np.linalg.norm(ref[15]))
# per wave tile check of dpre7 against fp64
rf = dpre7.numpy()*gs
for t in range(16):
    print('tile',t,'dpre7 rel', np.linalg.norm(dp7[t*32:(t+1)*32]-rf[t*32:(t+1)*32])/np.linalg.norm(rf[t*32:(t+1)*32]))

print('---- tile 1 detail')
dvb = decode_all(gbuf,154,0,8); dbb = decode_all(gbuf,154,8,16)
for name,got,refv in [('dv',dvb,dv.numpy()*gs),('dbott',dbb,dbott.numpy()*gs),('dpre7',dp7,rf)]:
    for t in (1,2):
        g_=got[t*32:(t+1)*32]; r_=refv[t*32:(t+1)*32]
        err=np.abs(g_-r_)
        print(name,'tile',t,'rel',np.linalg.norm(g_-r_)/np.linalg.norm(r_),'bad points',np.where(err.max(1)>0.02*np.abs(r_).max())[0],'bad feats',np.where(err.max(0)>0.02*np.abs(r_).max())[0][:40])
