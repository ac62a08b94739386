import sys, ctypes as C, torch, numpy as np
sys.path.insert(0,'.')
from floodplanet_code_amd import _lib
from floodplanet_code_amd._lib import check, ptr
lib=_lib.load()
raw=C.CDLL(_lib.LIB_PATH)
raw.fu_debug_set_conv_stamps.argtypes=[C.c_void_p]
DEV='cuda:0'
def run(B,C0,Cout,H,W,bn=True,mode=0):
    x=torch.randn(B,H,W,C0,device=DEV).to(torch.bfloat16); a=torch.rand(C0,device=DEV)+0.5; b=torch.randn(C0,device=DEV)*0.1
    w=torch.randn(Cout,C0,3,3,device=DEV)/10; bias=torch.zeros(Cout,device=DEV)
    y=torch.empty(B,H,W,Cout,device=DEV,dtype=torch.bfloat16)
    nwg=B*((H+15)//16)*((W+15)//16)*((Cout+63)//64)
    dbg=torch.zeros(nwg*10+64,dtype=torch.int64,device=DEV)
    for it in range(3):
        raw.fu_debug_set_conv_stamps(dbg.data_ptr() if it==2 else None)
        check(lib.fu_op_conv3x3_fwd(1,ptr(x),C0,ptr(a),ptr(b),None,0,ptr(w),ptr(bias),ptr(y),Cout,B,H,W,None,None,torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize(); raw.fu_debug_set_conv_stamps(None)
    d=dbg[:nwg*10].view(nwg,10).cpu().numpy().astype(np.float64)
    d=d[d[:,3]>0]
    t0=d[:,0].min()
    nch=(C0+31)//32
    print(f"   per-chunk: barrier-wait {np.median(d[:,4])/nch:.0f} | load-wait {np.median(d[:,5])/nch:.0f} | store+barrier {np.median(d[:,6])/nch:.0f} | mfma block {np.median(d[:,7])/nch:.0f} || epilogue: math+stores issued {np.median(d[:,8]-d[:,2]):.0f} | stats {np.median(d[:,9]-d[:,8]):.0f} | drain {np.median(d[:,3]-d[:,9]):.0f}")
    print(f"{C0}->{Cout} @{H} B={B}: wgs={len(d)} | prologue(load+store chunk0) {np.median(d[:,1]-d[:,0]):.0f} | main loop {np.median(d[:,2]-d[:,1]):.0f} | epilogue+drain {np.median(d[:,3]-d[:,2]):.0f} | WG lifetime {np.median(d[:,3]-d[:,0]):.0f} cycles (100MHz ticks?) | kernel span {(d[:,3].max()-t0):.0f}")
run(16,64,64,256,256)
run(16,128,128,128,128)
run(16,512,512,32,32)
run(16,256,256,64,64)

