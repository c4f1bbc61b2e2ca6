import sys, time, torch, numpy as np
sys.path.insert(0,'.')
from passiveradar_amd import engine, _lib
dev=torch.device('cuda')
n=200000; L=1024
for ns in (1, 256, 1024, 1026, 2048, 3072, 4096):
    g=torch.Generator(device=dev); g.manual_seed(1)
    ref=torch.view_as_complex(torch.randn((ns*n,2),generator=g,device=dev))
    srv=torch.roll(ref,2)+0.01*torch.view_as_complex(torch.randn((ns*n,2),generator=g,device=dev))
    out=torch.empty_like(srv)
    s=_lib.torch_stream_ptr()
    engine.nlms_execute(ref,srv,out,n,L,0.02,10,None,None,ns,n,n,s); torch.cuda.synchronize()
    t=time.perf_counter(); engine.nlms_execute(ref,srv,out,n,L,0.02,10,None,None,ns,n,n,s); torch.cuda.synchronize(); dt=time.perf_counter()-t
    print(f"streams {ns}: {dt*1e3:.1f} ms  -> {dt/(n-L-10)*1e9:.0f} ns/step/stream-batch, {ns*(n-L-10)/dt/1e6:.1f} Msamples/s total")
