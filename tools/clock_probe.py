"""In-kernel clock of the inference MLP kernel: s_memtime / s_memrealtime around the whole
persistent workgroup, after >= 2 s of back-to-back launches on random data."""
import ctypes, sys, time, torch, numpy as np
L = ctypes.CDLL(sys.argv[1])
NPT = int(sys.argv[2]) if len(sys.argv) > 2 else 3   # point tiles per wave of the build under test
L.nerf_mlp_forward_rays.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
L.nerf_mlp_packed_bytes.restype = ctypes.c_size_t
L.nerf_mlp_pack.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
L.nerf_debug_set_fwd_stamps.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0"); torch.manual_seed(0)

dims = [(63,256)]+[(256,256)]*4+[(319,256)]+[(256,256)]*2+[(256,256),(283,128),(128,3),(256,1)]
ws = []
for i,o in dims:
    lim = (6.0/(i+o))**0.5
    ws.append(((torch.rand((i,o),device=dev)*2-1)*lim).contiguous()); ws.append(torch.zeros((o,),device=dev))
ptrs = (ctypes.c_void_p*24)(*[w.data_ptr() for w in ws])
packed = torch.empty((L.nerf_mlp_packed_bytes(),), dtype=torch.uint8, device=dev)
assert L.nerf_mlp_pack(ptrs, packed.data_ptr(), None) == 0
n_rays, S = 131072, 128
o = torch.randn((n_rays,3),device=dev); o = 4*o/o.norm(dim=-1,keepdim=True)
d = (torch.rand((n_rays,3),device=dev)*2-1 - o)
vd = d/d.norm(dim=-1,keepdim=True)
rays = torch.cat([o,d,torch.full((n_rays,1),2.,device=dev),torch.full((n_rays,1),6.,device=dev),vd],-1).contiguous()
z = (torch.rand((n_rays,S),device=dev)*4+2).sort(-1).values.contiguous()
raw = torch.empty((n_rays,S,4),device=dev)
stamps = torch.zeros((4*256+32*8*16,),dtype=torch.int64,device=dev)
L.nerf_debug_set_fwd_stamps(stamps.data_ptr())
t0=time.time(); n=0
while time.time()-t0 < 3.0:
    for _ in range(10):
        assert L.nerf_mlp_forward_rays(packed.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), n_rays, S, raw.data_ptr(), None)==0
    torch.cuda.synchronize(); n+=10
e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    L.nerf_mlp_forward_rays(packed.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), n_rays, S, raw.data_ptr(), None)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)/10
full = stamps.cpu().numpy(); s = full[:1024].reshape(-1,4); ls = full[1024:].reshape(32,8,16)
clk = (s[:,1]-s[:,0])/((s[:,3]-s[:,2])*10.0)  # cycles per ns = GHz
span = (s[:,3]-s[:,2])*0.01
pts = n_rays*S
tf = pts*1186816/(ms*1e-3)/1e12
mfma_cyc_per_simd = pts/32/1024 * 2344*16   # tiles per CU x cycles per tile per SIMD
print(f"launch {ms:.3f} ms = {tf:.1f} TFLOP/s; in-kernel clock median {np.median(clk):.3f} GHz (p10 {np.percentile(clk,10):.3f}, p90 {np.percentile(clk,90):.3f}); WG span median {np.median(span):.1f} us")
print(f"MFMA pipe busy at that clock: {mfma_cyc_per_simd/(np.median(s[:,1]-s[:,0])):.3f}; finite raw: {bool(torch.isfinite(raw).all())}; mean rgb pre-act {raw[...,:3].mean().item():.3f}")

names = ["rebase+PE","start(sync0)","L0","L1","L2","L3","L4","L5","L6","L7","alpha+bott","view+fetch","rgb+store","finish"]
d = np.diff(ls[:,:,:15],axis=2)  # [blk][wave][14]
print("segment: median cycles over 32 WGs x 8 waves (MFMA-only time of a SIMD's two waves per point tile: hidden layer 4096, L0 1024, L5 5120, alpha+bott 4224, view 2304; this build has %d point tiles)" % NPT)
for i in range(14):
    print(f"  {names[i]:>14s}: {np.median(d[:,:,i]):8.0f}  (p10 {np.percentile(d[:,:,i],10):8.0f} p90 {np.percentile(d[:,:,i],90):8.0f})")
tot = ls[:,:,14]-ls[:,:,0]
print("pass total median", np.median(tot), "cycles; MFMA-only time of a SIMD's two waves:", 2*2344*8*NPT)
print("wave skew at L1 start within WG 0 (cycles rel. to wave 0):", (ls[0,:,3]-ls[0,0,3]).tolist())
