# microbench: layer-3 conv1 (M=12544,N=256,K=1024): plain variants vs dual-source variants (bn3 + add + ReLU folded in)
import importlib, sys, os, torch, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sat = importlib.import_module("show-and-tell_amd"); L = sat._lib; lib = L.load()
N,H,W,Cin,Cout=64,14,14,1024,256; M=N*H*W
x=torch.randn(M,Cin,device="cuda").bfloat16(); r=torch.randn(M,Cin,device="cuda").bfloat16(); y=torch.empty(M,Cin,device="cuda",dtype=torch.bfloat16)
w=(torch.randn(Cout,Cin,device="cuda")/32).bfloat16(); out=torch.empty(M,Cout,device="cuda",dtype=torch.bfloat16)
acc=torch.zeros(2,2,Cout,dtype=torch.int64,device="cuda"); sc,sh=torch.ones(Cin,device="cuda"),torch.zeros(Cin,device="cuda")
big=torch.empty(64*1024*1024,device="cuda")   # 256 MB: flush MALL between reps when FLUSH=1
def run(v,dual):
    o=L.SatOp(); o.kind,o.dtype=L.OP_CONV,L.SAT_BF16
    o.in0,o.w,o.out=x.data_ptr(),w.data_ptr(),out.data_ptr()
    o.N,o.Hin,o.Win,o.Cin,o.Hout,o.Wout,o.Cout=N,H,W,Cin,H,W,Cout
    o.KH,o.KW,o.stride,o.pad=1,1,1,0; o.sN,o.sH,o.sW=H*W*Cin,W*Cin,Cin
    o.stat_acc=acc.data_ptr(); o.variant=v
    if dual:
        o.scale0,o.shift0=sc.data_ptr(),sh.data_ptr(); o.in1=r.data_ptr(); o.out1=y.data_ptr()
    ops=(L.SatOp*1)(o)
    for _ in range(3): L.check(lib.sat_run_ops_parity(ops,1,0,L.stream()))
    torch.cuda.synchronize()
    tot=0.0
    for _ in range(10):
        if os.environ.get("FLUSH"): big.zero_()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); L.check(lib.sat_run_ops_parity(ops,1,0,L.stream())); e1.record(); torch.cuda.synchronize()
        tot+=e0.elapsed_time(e1)
    print("dbg=%s flush=%s dual=%d variant %2d: %6.1f us"%(os.environ.get("SAT_CONV_DBG","0"),os.environ.get("FLUSH","0"),dual,v,tot/10*1e3))
for v in [1,2,16,17]: run(v,False)
for v in [22,24,26,27]: run(v,True)
