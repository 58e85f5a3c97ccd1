import numpy as np, torch, sys
sys.path.insert(0,'.')
from oracle import nerf_oracle as orc
from nerf_amd.run_nerf_helpers import sample_pdf
nb, ns, det = 63, 128, True
rng = np.random.default_rng(nb * 7 + ns)
n = 301
bins = np.sort(rng.uniform(2., 6., size=(n, nb)), -1).astype(np.float32)
w = (rng.uniform(size=(n, nb - 1)) ** 4).astype(np.float32)
w[::7] = 0.
g = torch.device('cuda')
s = sample_pdf(torch.tensor(bins, device=g), torch.tensor(w, device=g), ns, det=det).cpu().numpy()
ref64 = orc.sample_pdf(bins, w, ns, det=det, dtype=np.float64)
ref32 = orc.sample_pdf(bins, w, ns, det=det, dtype=np.float32)
bad = np.argwhere(np.abs(s-ref64) > 2e-5*4)
print(len(bad), 'bad vs f64;', np.sum(np.abs(s-ref32)>8e-5), 'bad vs f32;', np.sum(np.abs(ref32-ref64)>8e-5), 'f32 vs f64')
for (r,k) in bad[:10]:
    ww = w[r].astype(np.float64)+1e-5; pdf = ww/ww.sum(); cdf = np.concatenate([[0],np.cumsum(pdf)])
    u = k/(ns-1)
    i = np.searchsorted(cdf, u, side='right')
    print(r,k,'u',u,'inds',i,'cdf around',cdf[max(0,i-2):i+2],'pdf',pdf[max(0,i-2):i+1],'gpu',s[r,k],'f64',ref64[r,k],'f32',ref32[r,k], 'bins', bins[r,max(0,i-2):i+2])
