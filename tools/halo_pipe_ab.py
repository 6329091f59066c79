import sys, os, torch
ROOT="/root/repo"; sys.path[:0]=[ROOT, os.path.join(ROOT,"dlmc-quant_amd")]
from dlmc import _native as N
from dlmc.quantization.scalar import kernels as K
dev="cuda:0"; g=torch.Generator(device=dev).manual_seed(1)
for (n,c,h,k) in [(512,256,14,256),(512,512,7,512),(512,128,28,128),(256,256,14,256),(256,512,7,512),(256,128,28,128)]:
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
