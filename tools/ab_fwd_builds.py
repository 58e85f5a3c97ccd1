"""A/B of two builds of the inference kernel: bit equality of raw on ragged shapes, then timing."""
import ctypes, sys, time, torch, numpy as np
libs = [ctypes.CDLL(p) for p in sys.argv[1:]]
for L in libs:
    L.nerf_mlp_forward_rays.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.nerf_mlp_forward_pts.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    L.nerf_mlp_packed_bytes.restype = ctypes.c_size_t
    L.nerf_mlp_pack.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device("cuda:0"); torch.manual_seed(0)
dims = [(63,256)]+[(256,256)]*4+[(319,256)]+[(256,256)]*2+[(256,256),(283,128),(128,3),(256,1)]
ws = []
for i,o in dims:
    lim = (6.0/(i+o))**0.5
    ws.append(((torch.rand((i,o),device=dev)*2-1)*lim).contiguous()); ws.append(((torch.rand((o,),device=dev)*2-1)*0.1).contiguous())
ptrs = (ctypes.c_void_p*24)(*[w.data_ptr() for w in ws])
L0 = libs[0]
packed = torch.empty((L0.nerf_mlp_packed_bytes(),), dtype=torch.uint8, device=dev)
assert L0.nerf_mlp_pack(ptrs, packed.data_ptr(), None) == 0
def batch(n_rays, S):
    o = torch.randn((n_rays,3),device=dev); o = 4*o/o.norm(dim=-1,keepdim=True)
    d = (torch.rand((n_rays,3),device=dev)*2-1 - o)
    vd = d/d.norm(dim=-1,keepdim=True)
    rays = torch.cat([o,d,torch.full((n_rays,1),2.,device=dev),torch.full((n_rays,1),6.,device=dev),vd],-1).contiguous()
    z = (torch.rand((n_rays,S),device=dev)*4+2).sort(-1).values.contiguous()
    return rays, z
ok = True
for n_rays, S in ((1,1),(3,5),(7,64),(100,192),(5,37),(1000,64),(4096,192),(33,1000),(20000,7)):
    rays, z = batch(n_rays, S)
    outs = []
    for L in libs:
        raw = torch.full((n_rays,S,4), float('nan'), device=dev)
        assert L.nerf_mlp_forward_rays(packed.data_ptr(), rays.data_ptr(), 11, z.data_ptr(), n_rays, S, raw.data_ptr(), None)==0
        torch.cuda.synchronize(); outs.append(raw)
    pts = (rays[:,None,0:3] + rays[:,None,3:6]*z[...,None]).contiguous(); vd = rays[:,8:11].contiguous()
    for L in libs:
        raw = torch.full((n_rays,S,4), float('nan'), device=dev)
        assert L.nerf_mlp_forward_pts(packed.data_ptr(), pts.data_ptr(), vd.data_ptr(), n_rays, S, raw.data_ptr(), None)==0
        torch.cuda.synchronize(); outs.append(raw)
    same = all(torch.equal(outs[0], o) for o in outs[1:len(libs)]) and all(torch.equal(outs[len(libs)], o) for o in outs[len(libs)+1:])
    fin = all(bool(torch.isfinite(o).all()) for o in outs)
    print(f"{n_rays}x{S}: bit-equal across builds {same}, finite {fin}")
    ok &= same and fin
print("ALL EQUAL" if ok else "MISMATCH")
rays, z = batch(131072, 128); raw = torch.empty((131072,128,4),device=dev)
rays2, z2 = batch(4096, 192); raw2 = torch.empty((4096,192,4),device=dev)
def timeit(L, r, zz, rw, n, S, reps):
    for _ in range(3): L.nerf_mlp_forward_rays(packed.data_ptr(), r.data_ptr(), 11, zz.data_ptr(), n, S, rw.data_ptr(), None)
    torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    for _ in range(reps): L.nerf_mlp_forward_rays(packed.data_ptr(), r.data_ptr(), 11, zz.data_ptr(), n, S, rw.data_ptr(), None)
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps
for rnd in range(3):
    for i, L in enumerate(libs):
        ms = timeit(L, rays, z, raw, 131072, 128, 20); ms2 = timeit(L, rays2, z2, raw2, 4096, 192, 50)
        print(f"round {rnd} {sys.argv[1+i].split('_')[-1]}: 131072x128 {ms:.3f} ms = {131072*128*1186816/ms/1e9:.1f} TFLOP/s; 4096x192 {ms2*1e3:.1f} us = {4096*192*1186816/ms2/1e9:.1f} TFLOP/s")
