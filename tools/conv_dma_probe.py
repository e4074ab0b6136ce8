import os, sys, torch
sys.path.insert(0, os.getcwd())
from framepack_amd import ops
dev="cuda:0"
g=torch.Generator(device=dev).manual_seed(0)
def run(T,H,W,cin,cout,res):
    x=torch.randn(T,H,W,cin,generator=g,device=dev).half()
    w=(torch.randn(cout,27*cin,generator=g,device=dev)*(27*cin)**-0.5).half()
    b=torch.randn(cout,generator=g,device=dev).half()
    r=torch.randn(T,H,W,cout,generator=g,device=dev).half() if res else None
    for _ in range(2): ops.conv3d_causal(x,w,b,resid=r)
    torch.cuda.synchronize()
    a,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5): ops.conv3d_causal(x,w,b,resid=r)
    e.record(); torch.cuda.synchronize()
    ms=a.elapsed_time(e)/5
    print(f"conv {cin}->{cout} at {T}x{H}x{W} res={res}: {ms:.3f} ms  {2*27*cin*cout*T*H*W/ms/1e9:.0f} TF/s", flush=True)
run(17,640,640,128,128,False)
run(17,640,640,128,128,True)
run(17,640,640,256,128,False)
run(17,640,640,256,256,False)
