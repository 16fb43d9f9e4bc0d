# microbench: sat_lstm_fwd / sat_lstm_bwd at the configs[3] layer shapes (B = 64, T = 19, In = 512 / 1024, H = 1024)
import importlib, sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sat = importlib.import_module("show-and-tell_amd"); L = sat._lib; lib = L.load()
B, T, H = 64, 19, 1024
for In in (512, 1024):
    pi = sat.PackInfo.get([T] * B, "cuda"); N = pi.N
    X = torch.randn(N, In, device="cuda"); k = 1 / H ** 0.5
    w_ih = torch.empty(4 * H, In, device="cuda").uniform_(-k, k); w_hh = torch.empty(4 * H, H, device="cuda").uniform_(-k, k)
    b = torch.zeros(4 * H, device="cuda")
    GA = torch.empty(N, 4 * H, device="cuda"); CS = torch.empty(N, H, device="cuda"); HS = torch.empty(N, H, device="cuda"); HP = torch.empty(N, H, device="cuda"); cst = torch.empty(B, H, device="cuda")
    wsb = lib.sat_lstm_fwd_ws_bytes(B, H); ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda")
    def run(with_ws, reps=30):
        args = (X.data_ptr(), w_ih.data_ptr(), w_hh.data_ptr(), b.data_ptr(), b.data_ptr(), pi.bs_c, T, In, H, GA.data_ptr(), CS.data_ptr(), HS.data_ptr(), HP.data_ptr(), cst.data_ptr(), ws.data_ptr() if with_ws else None, wsb if with_ws else 0, L.stream())
        for _ in range(3): L.check(lib.sat_lstm_fwd(*args))
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): L.check(lib.sat_lstm_fwd(*args))
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    p_, s_ = run(True), run(False)
    err = int(ws[wsb - 64:wsb - 60].view(torch.int32).item())
    print("In=%d H=%d: sat_lstm_fwd persistent %.1f us (status %d)   per-step launches %.1f us" % (In, H, p_, err, s_))
