import sys, os, torch
ROOT="/root/repo"; sys.path[:0]=[ROOT, os.path.join(ROOT,"dlmc-quant_amd")]
from dlmc import _native as N
from dlmc.quantization.scalar import kernels as K
dev="cuda:0"; g=torch.Generator(device=dev).manual_seed(1)
shapes=[(512,256,14,256),(512,512,7,512),(512,128,28,128),(256,256,14,256),(256,512,7,512),(256,128,28,128)]
if "--stamps" in sys.argv:
    shapes=shapes[:1]
for (n,c,h,k) in shapes:
    nset=6
    xs=[torch.randint(-128,128,(n,c,h,h),generator=g,device=dev,dtype=torch.int16).to(torch.int8).contiguous(memory_format=torch.channels_last) for _ in range(nset)]
    wq=torch.randint(-127,128,(k,3,3,c),generator=g,device=dev,dtype=torch.int8)
    wsum=wq.to(torch.int32).sum(dim=(1,2,3)).to(torch.int32).contiguous()
    s_w=torch.full((k,),1e-4,device=dev); bias=torch.randn(k,device=dev)
    one=torch.full((1,),0.02,device=dev); zp=torch.full((1,),-128.0,device=dev)
    emit=K.EmitCodes(torch.full((1,),0.05,device=dev),None,0,255,N.FORM_ZEROPOINT,shift128=True)
    res={}
    for name,kw in (("pipe",{"pipelined":True}),("halo",{})):
        ts=[]
        for it in range(12):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record(); K.conv2d_i8(xs[it%nset],wq,wsum,bias,one,zp,s_w,padding=1,relu=True,emit=emit,want_out=False,**kw); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1)*1e3)
        res[name]=sorted(ts)[len(ts)//2]
    ops=2*n*h*h*k*c*9
    print(f"N{n} C{c} {h}^2 K{k}: pipe {res['pipe']:7.1f} us ({ops/res['pipe']/1e6:6.0f} TOP/s)   halo {res['halo']:7.1f} us ({ops/res['halo']/1e6:6.0f} TOP/s)   ratio {res['pipe']/res['halo']:.3f}",flush=True)
    if "--stamps" in sys.argv and hasattr(N.lib, "dlmcq_x_pipe_stamps"):
        import ctypes
        buf=(ctypes.c_uint64*256)()
        torch.cuda.synchronize()
        N.lib.dlmcq_x_pipe_stamps(buf)
        st=list(buf)
        ghz=(st[2]-st[0])/max(st[3]-st[1],1)*0.1
        steps=[st[i+1]-st[i] for i in range(4,4+71)]
        print(f"stamps: workgroup 0 lives {st[2]-st[0]} clocks = {(st[3]-st[1])*0.01:.1f} us, in-kernel clock {ghz:.2f} GHz; clocks per K step, tile 0: "
              f"{sorted(steps[:36])[18]} median (min {min(steps[:36])}, max {max(steps[:36])}); tile 1: {sorted(steps[36:71])[17]} median; first ten: {steps[:10]}")
        wg=(ctypes.c_uint64*1024)()
        N.lib.dlmcq_x_pipe_wg(wg)
        w=list(wg); st_=[w[2*i] for i in range(256)]; en=[w[2*i+1] for i in range(256)]
        t0=min(st_); life=sorted((e-s_)*0.01 for s_,e in zip(st_,en))
        print(f"stamps: 256 workgroups: starts {0:.1f} .. {(max(st_)-t0)*0.01:.1f} us, ends {(min(en)-t0)*0.01:.1f} .. {(max(en)-t0)*0.01:.1f} us; lifetimes min {life[0]:.1f} median {life[128]:.1f} max {life[-1]:.1f} us")
