# microbench: sat_lstm_fwd at cfg-2 dims (B=64, T=19, E=256, H=512): persistent launch vs one launch per step
import importlib, sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sat = importlib.import_module("show-and-tell_amd"); L = sat._lib; lib = L.load()
B,T,In,H = 64,19,256,512
pi = sat.PackInfo.get([T]*B, "cuda"); N = pi.N
X=torch.randn(N,In,device="cuda"); k=1/H**0.5
w_ih=torch.empty(4*H,In,device="cuda").uniform_(-k,k); w_hh=torch.empty(4*H,H,device="cuda").uniform_(-k,k)
b=torch.zeros(4*H,device="cuda")
GA=torch.empty(N,4*H,device="cuda"); CS=torch.empty(N,H,device="cuda"); HS=torch.empty(N,H,device="cuda"); HP=torch.empty(N,H,device="cuda"); cst=torch.empty(B,H,device="cuda")
wsb=lib.sat_lstm_fwd_ws_bytes(B,H); ws=torch.empty(wsb,dtype=torch.uint8,device="cuda")
def run(with_ws, reps=50):
    args=(X.data_ptr(),w_ih.data_ptr(),w_hh.data_ptr(),b.data_ptr(),b.data_ptr(),pi.bs_c,T,In,H,GA.data_ptr(),CS.data_ptr(),HS.data_ptr(),HP.data_ptr(),cst.data_ptr(), ws.data_ptr() if with_ws else None, wsb if with_ws else 0, L.stream())
    for _ in range(5): L.check(lib.sat_lstm_fwd(*args))
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): L.check(lib.sat_lstm_fwd(*args))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
print("sat_lstm_fwd (x-gate GEMM + recurrence) persistent: %.1f us   per-step launches: %.1f us" % (run(True), run(False)))

# backward chain: fused step launches (SAT_LSTM_BWD_FUSED, default) vs pointwise + split-K skinny GEMM per step (env read once per process)
dHS=torch.randn(N,H,device="cuda"); DG=torch.empty(N,4*H,device="cuda")
dw_ih=torch.empty(4*H,In,device="cuda"); dw_hh=torch.empty(4*H,H,device="cuda"); db1=torch.empty(4*H,device="cuda"); db2=torch.empty(4*H,device="cuda"); dX=torch.empty(N,In,device="cuda")
bws=lib.sat_lstm_bwd_ws_bytes_full(N,B,In,H); bw=torch.empty(bws//4,device="cuda")
L.check(lib.sat_lstm_fwd(X.data_ptr(),w_ih.data_ptr(),w_hh.data_ptr(),b.data_ptr(),b.data_ptr(),pi.bs_c,T,In,H,GA.data_ptr(),CS.data_ptr(),HS.data_ptr(),HP.data_ptr(),cst.data_ptr(),ws.data_ptr(),wsb,L.stream()))
def runb(reps=50):
    args=(dHS.data_ptr(),X.data_ptr(),w_ih.data_ptr(),w_hh.data_ptr(),GA.data_ptr(),CS.data_ptr(),HP.data_ptr(),pi.bs_c,T,In,H,DG.data_ptr(),dw_ih.data_ptr(),dw_hh.data_ptr(),db1.data_ptr(),db2.data_ptr(),dX.data_ptr(),bw.data_ptr(),bws,L.stream())
    for _ in range(5): L.check(lib.sat_lstm_bwd(*args))
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): L.check(lib.sat_lstm_bwd(*args))
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/reps*1e3
print("sat_lstm_bwd (chain + batched weight-gradient GEMMs) SAT_LSTM_BWD_FUSED=%s: %.1f us" % (os.environ.get("SAT_LSTM_BWD_FUSED","1"), runb()))
